/* The elementary functions of the shading arithmetic, spelled out.
 *
 * The reference calls sinf / cosf / powf / acosf / atan2f / exp2f of the CUDA
 * math library under --use_fast_math (CMakeLists.txt:52-58): approximations whose
 * results are a property of that toolchain, not of the algorithm.  A host libm and
 * a device math library round these functions differently in the last place, and a
 * path integrator turns one such bit into another path wherever a random decision
 * sits on the result.  So the engine's device code and the CPU oracle both take
 * them from this header: every function below is a fixed sequence of IEEE-754
 * binary64 additions, multiplications, divisions, square roots, floor and
 * conversions - operations that are correctly rounded on the host and on the GPU -
 * rounded once to binary32 at the end.  Compiled without contraction
 * (-ffp-contract=off) the results are the same bits everywhere, and within one
 * unit in the last place of the exact value (tests/test_shared_math.py checks that
 * against the host's libm over dense samples).
 *
 * Argument reduction and polynomial coefficients follow the published fdlibm
 * algorithms (Sun Microsystems, 1993: k_sin.c, k_cos.c, s_atan.c, e_log.c).
 *
 * Domain notes: rtm::sine / rtm::cosine give up (NaN for infinities and NaN, 0 / 1
 * for finite arguments) beyond |x| = 2^30, where a binary32 argument no longer
 * resolves a period anyway; rtm::power is defined for x >= 0 (a negative base gives
 * NaN: the callers clamp their bases), with power(x, 0) = power(1, y) = 1. */
#ifndef RT_MATH_H
#define RT_MATH_H

#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__) || defined(__CUDACC__)
#define RT_MATH_FN __host__ __device__ inline
#else
#include <math.h>
#define RT_MATH_FN inline
#endif

namespace rtm {

RT_MATH_FN uint64_t bitsOf(double v) {
    uint64_t b;
    memcpy(&b, &v, sizeof(b));
    return b;
}

RT_MATH_FN double fromBits(uint64_t b) {
    double v;
    memcpy(&v, &b, sizeof(v));
    return v;
}

RT_MATH_FN double absolute(double v) { return fromBits(bitsOf(v) & 0x7fffffffffffffffULL); }
RT_MATH_FN bool negative(double v) { return (bitsOf(v) >> 63) != 0; }
RT_MATH_FN double quietNaN() { return fromBits(0x7ff8000000000000ULL); }
RT_MATH_FN double infinity() { return fromBits(0x7ff0000000000000ULL); }

/* sine and cosine of r, |r| <= pi / 4 (k_sin.c, k_cos.c) */
RT_MATH_FN double sinKernel(double r) {
    const double z = r * r;
    const double p = -1.66666666666666324348e-01 +
                     z * (8.33333333332248946124e-03 +
                          z * (-1.98412698298579493134e-04 +
                               z * (2.75573137070700676789e-06 + z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10))));
    return r + r * (z * p);
}

RT_MATH_FN double cosKernel(double r) {
    const double z = r * r;
    const double p = 4.16666666666666019037e-02 +
                     z * (-1.38888888888741095749e-03 +
                          z * (2.48015872894767294178e-05 +
                               z * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11))));
    return (1.0 - 0.5 * z) + z * (z * p);
}

/* x = k * pi/2 + r with |r| <= pi/4; returns k mod 4.  pi/2 in two pieces, the first 33 bits long, so that
 * k * piece is exact for |k| < 2^20 (|x| < 1.6e6); beyond that the remainder loses bits one by one. */
RT_MATH_FN int reduceQuarterTurns(double x, double& r) {
    const double k = floor(x * 6.36619772367581382433e-01 + 0.5);
    r = (x - k * 1.57079632673412561417e+00) - k * 6.07710050650619224932e-11;
    return static_cast<int>(k - 4.0 * floor(k * 0.25));
}

RT_MATH_FN float sine(float x) {
    const double v = x;
    if (!(absolute(v) < 1073741824.0))
        return static_cast<float>(v - v);  /* NaN for NaN and infinities, 0 for what is left */
    double r;
    const int quarter = reduceQuarterTurns(v, r);
    const double s = (quarter & 1) ? cosKernel(r) : sinKernel(r);
    return static_cast<float>((quarter & 2) ? -s : s);
}

RT_MATH_FN float cosine(float x) {
    const double v = x;
    if (!(absolute(v) < 1073741824.0))
        return static_cast<float>((v - v) + 1.0);
    double r;
    const int quarter = reduceQuarterTurns(v, r);
    const double c = (quarter & 1) ? sinKernel(r) : cosKernel(r);
    return static_cast<float>(((quarter + 1) & 2) ? -c : c);
}

/* 2^x for |x| <= 1100, in binary64: the integer part goes into the exponent, the rest through the series of
 * e^t, t = f ln 2, |t| <= 0.347, up to t^11 / 11! (the next term is below 2^-47, a millionth of the last place
 * of the binary32 result). */
RT_MATH_FN double exp2Double(double x) {
    const double k = floor(x + 0.5);
    const double t = (x - k) * 6.93147180559945286227e-01;
    double p = 1.0 / 39916800.0;
    p = p * t + 1.0 / 3628800.0;
    p = p * t + 1.0 / 362880.0;
    p = p * t + 1.0 / 40320.0;
    p = p * t + 1.0 / 5040.0;
    p = p * t + 1.0 / 720.0;
    p = p * t + 1.0 / 120.0;
    p = p * t + 1.0 / 24.0;
    p = p * t + 1.0 / 6.0;
    p = p * t + 0.5;
    p = p * t + 1.0;
    p = p * t + 1.0;
    /* two factors, so that the scale stays a normal number down to 2^-1100 */
    const int64_t whole = static_cast<int64_t>(k);
    const int64_t first = whole / 2, second = whole - first;
    return p * fromBits(static_cast<uint64_t>(first + 1023) << 52) * fromBits(static_cast<uint64_t>(second + 1023) << 52);
}

RT_MATH_FN float exp2(float x) {
    const double v = x;
    if (v != v)
        return x;
    if (v > 200.0)
        return static_cast<float>(infinity());
    if (v < -200.0)
        return 0.0f;
    return static_cast<float>(exp2Double(v));
}

/* log2 of a positive, finite, normal binary64 (every positive binary32 is one): exponent plus the series of
 * log((1 + s) / (1 - s)) = 2 (s + s^3 / 3 + ...), s = (m - 1) / (m + 1), m in [sqrt(1/2), sqrt(2)). */
RT_MATH_FN double log2Double(double x) {
    const uint64_t bits = bitsOf(x);
    int64_t exponent = static_cast<int64_t>((bits >> 52) & 0x7ffU) - 1023;
    double m = fromBits((bits & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL);
    if (m > 1.41421356237309514547) {
        m = m * 0.5;
        exponent = exponent + 1;
    }
    const double s = (m - 1.0) / (m + 1.0);
    const double z = s * s;
    double p = 1.0 / 23.0;
    p = p * z + 1.0 / 21.0;
    p = p * z + 1.0 / 19.0;
    p = p * z + 1.0 / 17.0;
    p = p * z + 1.0 / 15.0;
    p = p * z + 1.0 / 13.0;
    p = p * z + 1.0 / 11.0;
    p = p * z + 1.0 / 9.0;
    p = p * z + 1.0 / 7.0;
    p = p * z + 1.0 / 5.0;
    p = p * z + 1.0 / 3.0;
    p = p * z + 1.0;
    return static_cast<double>(exponent) + (2.0 * s) * p * 1.44269504088896338700e+00;
}

RT_MATH_FN float power(float x, float y) {
    const double base = x, exponent = y;
    if (base == 1.0 || exponent == 0.0)
        return 1.0f;
    if (base != base || exponent != exponent || base < 0.0)
        return static_cast<float>(quietNaN());
    if (exponent == 5.0) {  /* Schlick's weight, at every shaded point: four products, exact to the last place */
        const double square = base * base, fourth = square * square;
        return static_cast<float>(fourth * base);
    }
    if (base == 0.0)
        return exponent > 0.0 ? 0.0f : static_cast<float>(infinity());
    if (base == infinity())
        return exponent > 0.0 ? static_cast<float>(infinity()) : 0.0f;
    double t = exponent * log2Double(base);
    if (t != t)
        return static_cast<float>(quietNaN());  /* an infinite exponent on a base the logarithm of which is 0 cannot happen: base != 1 */
    if (t > 200.0)
        t = 200.0;  /* beyond the binary32 range either way */
    if (t < -200.0)
        t = -200.0;
    return static_cast<float>(exp2Double(t));
}

/* arc tangent of t, 0 <= t <= 1 (s_atan.c: three of its five intervals) */
RT_MATH_FN double atanUnit(double t) {
    double hi = 0.0, lo = 0.0, x = t;
    int reduced = 0;
    if (t >= 0.4375) {
        reduced = 1;
        if (t < 0.6875) {
            hi = 4.63647609000806093515e-01;
            lo = 2.26987774529616870924e-17;
            x = (2.0 * t - 1.0) / (2.0 + t);
        } else {
            hi = 7.85398163397448278999e-01;
            lo = 3.06161699786838301793e-17;
            x = (t - 1.0) / (t + 1.0);
        }
    }
    const double z = x * x;
    const double w = z * z;
    const double s1 = z * (3.33333333333329318027e-01 +
                           w * (1.42857142725034663711e-01 +
                                w * (9.09088713343650656196e-02 +
                                     w * (6.66107313738753120669e-02 + w * (4.97687799461593236017e-02 + w * 1.62858201153657823623e-02)))));
    const double s2 = w * (-1.99999999998764832476e-01 +
                           w * (-1.11111104054623557880e-01 +
                                w * (-7.69187620504482999495e-02 + w * (-5.83357013379057348645e-02 + w * -3.65315727442169155270e-02))));
    if (!reduced)
        return x - x * (s1 + s2);
    return hi - ((x * (s1 + s2) - lo) - x);
}

RT_MATH_FN double atan2Double(double y, double x) {
    if (x != x || y != y)
        return quietNaN();
    const double ax = absolute(x), ay = absolute(y);
    double angle;
    if (ax == 0.0 && ay == 0.0) {
        angle = 0.0;
    } else if (ax == infinity() && ay == infinity()) {
        angle = 7.85398163397448278999e-01;
    } else {
        const bool steep = ay > ax;
        angle = atanUnit(steep ? ax / ay : ay / ax);
        if (steep)
            angle = 1.57079632679489655800e+00 - angle;
    }
    if (negative(x))
        angle = 3.14159265358979311600e+00 - angle;
    return negative(y) ? -angle : angle;
}

RT_MATH_FN float arcTangent2(float y, float x) { return static_cast<float>(atan2Double(y, x)); }

/* arc cosine through the tangent: the angle of (x, sqrt((1 - x)(1 + x))); both factors are exact in binary64 */
RT_MATH_FN float arcCosine(float x) {
    const double v = x;
    if (!(absolute(v) <= 1.0))
        return static_cast<float>(quietNaN());
    return static_cast<float>(atan2Double(sqrt((1.0 - v) * (1.0 + v)), v));
}

}  /* namespace rtm */

#endif /* RT_MATH_H */
