// Device-side shading math of the path integrator and the photon tracer: the
// helper functions of the reference's device programs (src/device.cu:12-621,
// 623-712), written for the wavefront kernels of wavefront.hip.  IEEE
// arithmetic, no fast-math; random-number draw order and every constant follow
// the cited lines.
#pragma once

#include "bvh.h"
#include "light_index.h"
#include "rt_math.h"

namespace rt {

// Sine, cosine, power, arc functions and 2^x come from rt_math.h, sequences of correctly rounded binary64
// operations that give the same bits on the host, where the CPU restatement of this code uses them too:
// a device math library and a host libm differ in the last place of these functions, and one such bit sends
// a path another way.  -DRT_LIBM_MATH takes the device library's functions instead, for comparison.
#ifdef RT_LIBM_MATH
__device__ inline float mathSin(float x) { return sinf(x); }
__device__ inline float mathCos(float x) { return cosf(x); }
__device__ inline float mathPow(float x, float y) { return powf(x, y); }
__device__ inline float mathAtan2(float y, float x) { return atan2f(y, x); }
__device__ inline float mathAcos(float x) { return acosf(x); }
__device__ inline float mathExp2(float x) { return exp2f(x); }
#else
__device__ inline float mathSin(float x) { return rtm::sine(x); }
__device__ inline float mathCos(float x) { return rtm::cosine(x); }
__device__ inline float mathPow(float x, float y) { return rtm::power(x, y); }
__device__ inline float mathAtan2(float y, float x) { return rtm::arcTangent2(y, x); }
__device__ inline float mathAcos(float x) { return rtm::arcCosine(x); }
__device__ inline float mathExp2(float x) { return rtm::exp2(x); }
#endif

struct V3 {
    float x, y, z;
};

// device.cu:12-75.  Division by a scalar multiplies by the reciprocal (:40-42).
__device__ inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ inline V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
__device__ inline V3 operator*(V3 a, V3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
__device__ inline V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ inline V3 operator*(float s, V3 a) { return {a.x * s, a.y * s, a.z * s}; }
__device__ inline V3 operator/(V3 a, float s) { return a * (1.0f / s); }
__device__ inline V3 operator/(V3 a, V3 b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }
__device__ inline V3 operator+(V3 a, float s) { return {a.x + s, a.y + s, a.z + s}; }
__device__ inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ inline V3 cross(V3 a, V3 b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__device__ inline V3 normalize(V3 v) { return v / sqrtf(dot(v, v)); }
__device__ inline V3 v3(const rt_float3& v) { return {v.x, v.y, v.z}; }
__device__ inline V3 rgb(const rt_float4& v) { return {v.x, v.y, v.z}; }
__device__ inline V3 rgb(const float4& v) { return {v.x, v.y, v.z}; }
__device__ inline rt_float3 r3(V3 v) { return {v.x, v.y, v.z}; }
__device__ inline float maximum(V3 v) { return fmaxf(fmaxf(v.x, v.y), v.z); }  // :184-186
__device__ inline float saturate(float v) { return fminf(fmaxf(v, 0.0f), 1.0f); }  // :188-190

// device.cu:220-233
__device__ inline uint32_t seeded(uint32_t a, uint32_t b) {
    uint32_t value = a * 0x9e3779b9U ^ b * 0x85ebca6bU;
    value ^= value >> 16U;
    value *= 0x7feb352dU;
    value ^= value >> 15U;
    value *= 0x846ca68bU;
    return value ^ (value >> 16U);
}

__device__ inline float random01(uint32_t& state) {
    state = state * 747796405U + 2891336453U;
    const uint32_t word = ((state >> ((state >> 28U) + 4U)) ^ state) * 277803737U;
    return static_cast<float>((word >> 22U) ^ word) * 0x1p-32f;
}

// device.cu:81-84
__device__ inline float srgbToLinear(float value) {
    return value <= 0.04045f ? value / 12.92f : mathPow((value + 0.055f) / 1.055f, 2.4f);
}

// value mod n for n > 0 as ((value % n) + n) % n gives it, with one division instead of two, and none for a
// power of two (the low bits of a two's complement number are its residue, negative or not).
__device__ inline int floorMod(int value, int n) {
    if ((n & (n - 1)) == 0)
        return value & (n - 1);
    const uint32_t un = static_cast<uint32_t>(n);
    return static_cast<int>(value >= 0 ? static_cast<uint32_t>(value) % un
                                       : un - 1U - static_cast<uint32_t>(~value) % un);
}

// device.cu:86-98 (glTF wrap modes: 33071 clamp, 33648 mirror, otherwise repeat) for a texel and its upper
// neighbour together: the neighbour of a wrapped coordinate follows from it without another division.
__device__ inline void wrapCoordinates(int value, uint32_t size, uint32_t mode, int& at, int& next) {
    const int n = static_cast<int>(size);
    if (mode == 33071U) {
        at = min(max(value, 0), n - 1);
        next = min(max(value + 1, 0), n - 1);
    } else if (mode == 33648U) {
        const int period = n * 2;
        const int first = floorMod(value, period);
        const int second = first + 1 == period ? 0 : first + 1;
        at = first < n ? first : period - first - 1;
        next = second < n ? second : period - second - 1;
    } else {
        at = floorMod(value, n);
        next = at + 1 == n ? 0 : at + 1;
    }
}

// byte / 255.0f, correctly rounded, without the division: a product with the rounded reciprocal, its
// remainder and one correction, every step one IEEE operation.  Equal to the quotient for each of the 256
// bytes (tests/test_abi.py evaluates both in exact arithmetic).
constexpr float kInverse255 = 1.0f / 255.0f;
__device__ inline float unitOfByte(uint8_t byte) {
    const float value = static_cast<float>(byte);
    const float q = value * kInverse255;
    return fmaf(fmaf(-q, 255.0f, value), kInverse255, q);
}

// device.cu:100-107
__device__ inline float4 fetchTexel(const rt_gpu_texture& texture, int x, int y) {
    const rt_uchar4 value = texture.pixels[y * texture.width + x];  // wrapped coordinates
    return make_float4(unitOfByte(value.x), unitOfByte(value.y), unitOfByte(value.z), unitOfByte(value.w));
}

// device.cu:109-147: bilinear lookup through the KHR_texture_transform of the reference
__device__ inline float4 sampleTexture(const rt_gpu_texture* textures, const rt_gpu_texture_ref& reference,
                                       const rt_hit& hit, bool srgb) {
    if (reference.texture < 0)
        return make_float4(1.0f, 1.0f, 1.0f, 1.0f);
    const rt_gpu_texture source = textures[reference.texture];
    const rt_float2 input = reference.texCoord == 1U ? hit.uv1 : hit.uv;
    const float sx = input.x * reference.scale.x, sy = input.y * reference.scale.y;
    // a rotation of zero (either sign) has the cosine 1 and itself as its sine, exactly
    const bool unrotated = reference.rotation == 0.0f;
    const float cosine = unrotated ? 1.0f : mathCos(reference.rotation);
    const float sine = unrotated ? reference.rotation : mathSin(reference.rotation);
    const float u = reference.offset.x + cosine * sx - sine * sy;
    const float v = reference.offset.y + sine * sx + cosine * sy;
    const float x = u * source.width - 0.5f;
    const float y = v * source.height - 0.5f;
    const int x0 = static_cast<int>(floorf(x));
    const int y0 = static_cast<int>(floorf(y));
    const float tx = x - floorf(x);
    const float ty = y - floorf(y);
    int xa, xb, ya, yb;
    wrapCoordinates(x0, source.width, source.wrapU, xa, xb);
    wrapCoordinates(y0, source.height, source.wrapV, ya, yb);
    const float4 a = fetchTexel(source, xa, ya);
    const float4 b = fetchTexel(source, xb, ya);
    const float4 c = fetchTexel(source, xa, yb);
    const float4 d = fetchTexel(source, xb, yb);
    float4 value = make_float4((a.x * (1.0f - tx) + b.x * tx) * (1.0f - ty) + (c.x * (1.0f - tx) + d.x * tx) * ty,
                               (a.y * (1.0f - tx) + b.y * tx) * (1.0f - ty) + (c.y * (1.0f - tx) + d.y * tx) * ty,
                               (a.z * (1.0f - tx) + b.z * tx) * (1.0f - ty) + (c.z * (1.0f - tx) + d.z * tx) * ty,
                               (a.w * (1.0f - tx) + b.w * tx) * (1.0f - ty) + (c.w * (1.0f - tx) + d.w * tx) * ty);
    if (srgb) {
        value.x = srgbToLinear(value.x);
        value.y = srgbToLinear(value.y);
        value.z = srgbToLinear(value.z);
    }
    return value;
}

// device.cu:149-166
__device__ inline rt_gpu_material applyTextures(const rt_gpu_texture* textures, rt_gpu_material material,
                                                const rt_hit& hit) {
    const float4 base = sampleTexture(textures, material.baseColorTexture, hit, true);
    material.baseColor.x *= base.x;
    material.baseColor.y *= base.y;
    material.baseColor.z *= base.z;
    material.baseColor.w *= base.w;
    const float4 pbr = sampleTexture(textures, material.metallicRoughnessTexture, hit, false);
    material.roughness *= pbr.y;
    material.metallic *= pbr.z;
    const float4 emissive = sampleTexture(textures, material.emissiveTexture, hit, true);
    material.emissive.x *= emissive.x;
    material.emissive.y *= emissive.y;
    material.emissive.z *= emissive.z;
    material.transmission *= sampleTexture(textures, material.transmissionTexture, hit, false).x;
    material.thickness *= sampleTexture(textures, material.thicknessTexture, hit, false).y;
    return material;
}

// device.cu:168-182
__device__ inline V3 shadingNormal(const rt_gpu_texture* textures, const rt_hit& hit,
                                   const rt_gpu_material& material) {
    if (material.normalTexture.texture < 0)
        return v3(hit.normal);
    const float4 sample = sampleTexture(textures, material.normalTexture, hit, false);
    const V3 tangent = normalize(V3{hit.tangent.x, hit.tangent.y, hit.tangent.z});
    const V3 local = normalize(V3{(sample.x * 2.0f - 1.0f) * material.normalTexture.strength,
                                  (sample.y * 2.0f - 1.0f) * material.normalTexture.strength,
                                  sample.z * 2.0f - 1.0f});
    return normalize(tangent * local.x + cross(v3(hit.normal), tangent) * (local.y * hit.tangent.w) +
                     v3(hit.normal) * local.z);
}

// device.cu:192-205
__device__ inline V3 reflectAbout(V3 direction, V3 normal) {
    return direction - 2.0f * dot(direction, normal) * normal;
}

__device__ inline bool refractThrough(V3 direction, V3 normal, float eta, V3& result) {
    const float cosine = fminf(dot(-direction, normal), 1.0f);
    const V3 perpendicular = eta * (direction + cosine * normal);
    const float parallelSquared = 1.0f - dot(perpendicular, perpendicular);
    if (parallelSquared < 0.0f)
        return false;
    result = perpendicular - sqrtf(parallelSquared) * normal;
    return true;
}

// device.cu:239-271: latitude/longitude environment lookup, constant grey when absent
__device__ inline void environmentCoordinates(const rt_launch_params& p, V3 direction, float& u, float& v) {
    u = mathAtan2(direction.z, direction.x) * 0.15915494f + 0.5f + p.environmentRotation * 0.15915494f;
    u -= floorf(u);
    v = mathAcos(fminf(fmaxf(direction.y, -1.0f), 1.0f)) * 0.31830989f;
}

__device__ inline V3 environmentPixel(const rt_launch_params& p, int x, int y) {
    x = floorMod(x, static_cast<int>(p.environmentWidth));
    y = min(max(y, 0), static_cast<int>(p.environmentHeight) - 1);
    return rgb(p.environment[y * p.environmentWidth + x]);
}

__device__ inline V3 environmentRadiance(const rt_launch_params& p, V3 direction) {
    if (!p.environment)
        return {0.1f, 0.1f, 0.1f};  // :235-237
    float u, v;
    environmentCoordinates(p, direction, u, v);
    const float x = u * p.environmentWidth - 0.5f;
    const float y = v * p.environmentHeight - 0.5f;
    const int x0 = static_cast<int>(floorf(x));
    const int y0 = static_cast<int>(floorf(y));
    const float tx = x - floorf(x);
    const float ty = y - floorf(y);
    const V3 a = environmentPixel(p, x0, y0) * (1.0f - tx) + environmentPixel(p, x0 + 1, y0) * tx;
    const V3 b = environmentPixel(p, x0, y0 + 1) * (1.0f - tx) + environmentPixel(p, x0 + 1, y0 + 1) * tx;
    return (a * (1.0f - ty) + b * ty) * p.environmentStrength;
}

// the tangent every local frame starts from (:278-280, 295-297, 634-636, 645-647)
__device__ inline V3 frameTangent(V3 normal) {
    return normalize(fabsf(normal.x) > 0.5f ? cross(V3{0.0f, 1.0f, 0.0f}, normal)
                                            : cross(V3{1.0f, 0.0f, 0.0f}, normal));
}

// device.cu:273-284: draws phi, then the radius
__device__ inline V3 sampleCosine(V3 normal, uint32_t& rng) {
    const float phi = 6.2831853f * random01(rng);
    const float radius = sqrtf(random01(rng));
    const float z = sqrtf(fmaxf(0.0f, 1.0f - radius * radius));
    const V3 tangent = frameTangent(normal);
    const V3 bitangent = cross(normal, tangent);
    return normalize(tangent * (radius * mathCos(phi)) + bitangent * (radius * mathSin(phi)) + normal * z);
}

__device__ inline float alphaOf(float roughness) { return fmaxf(roughness * roughness, 1e-4f); }  // :286-288

// device.cu:293-323: visible-normal sampling (Heitz); draws the radius, then phi
__device__ inline V3 sampleVisibleNormal(V3 view, V3 normal, float alpha, uint32_t& rng) {
    const V3 tangent = frameTangent(normal);
    const V3 bitangent = cross(normal, tangent);
    const V3 local{dot(view, tangent), dot(view, bitangent), dot(view, normal)};
    const V3 stretched = normalize(V3{alpha * local.x, alpha * local.y, local.z});
    const float lengthSquared = stretched.x * stretched.x + stretched.y * stretched.y;
    const V3 basisX = lengthSquared > 0.0f ? V3{-stretched.y, stretched.x, 0.0f} / sqrtf(lengthSquared)
                                           : V3{1.0f, 0.0f, 0.0f};
    const V3 basisY = cross(stretched, basisX);
    const float radius = sqrtf(random01(rng));
    const float phi = 6.2831853f * random01(rng);
    const float x = radius * mathCos(phi);
    float y = radius * mathSin(phi);
    const float lerp = 0.5f * (1.0f + stretched.z);
    y = (1.0f - lerp) * sqrtf(fmaxf(0.0f, 1.0f - x * x)) + lerp * y;
    const V3 hemisphere = basisX * x + basisY * y + stretched * sqrtf(fmaxf(0.0f, 1.0f - x * x - y * y));
    const V3 micro = normalize(V3{alpha * hemisphere.x, alpha * hemisphere.y, fmaxf(0.0f, hemisphere.z)});
    return normalize(tangent * micro.x + bitangent * micro.y + normal * micro.z);
}

// device.cu:325-337
__device__ inline float smithLambda(float cosine, float alpha) {
    const float squared = cosine * cosine;
    const float tangentSquared = fmaxf(1.0f - squared, 0.0f) / fmaxf(squared, 1e-8f);
    return 0.5f * (sqrtf(1.0f + alpha * alpha * tangentSquared) - 1.0f);
}

__device__ inline float maskingRatio(float viewCosine, float lightCosine, float alpha) {
    const float view = smithLambda(viewCosine, alpha);
    return (1.0f + view) / (1.0f + view + smithLambda(lightCosine, alpha));
}

// device.cu:339-358
__device__ inline float dielectricFresnel(float cosine, float ior) {
    const float r = (1.0f - ior) / (1.0f + ior);
    const float r2 = r * r;
    return r2 + (1.0f - r2) * mathPow(fmaxf(1.0f - cosine, 0.0f), 5.0f);
}

__device__ inline V3 schlickFresnel(V3 f0, float cosine) {
    const float scale = mathPow(fmaxf(1.0f - cosine, 0.0f), 5.0f);
    return {f0.x + (1.0f - f0.x) * scale, f0.y + (1.0f - f0.y) * scale, f0.z + (1.0f - f0.z) * scale};
}

__device__ inline V3 normalReflectance(const rt_gpu_material& m) {
    const V3 color = rgb(m.baseColor);
    return {0.04f + (color.x - 0.04f) * m.metallic, 0.04f + (color.y - 0.04f) * m.metallic,
            0.04f + (color.z - 0.04f) * m.metallic};
}

// device.cu:360-370: Beer-Lambert
__device__ inline V3 mediumAbsorption(const rt_gpu_material& m, float distance) {
    if (!isfinite(m.attenuationDistance) || m.attenuationDistance <= 0.0f)
        return {1.0f, 1.0f, 1.0f};
    const float scale = distance / m.attenuationDistance;
    return {mathPow(fmaxf(m.attenuationColor.x, 1e-4f), scale), mathPow(fmaxf(m.attenuationColor.y, 1e-4f), scale),
            mathPow(fmaxf(m.attenuationColor.z, 1e-4f), scale)};
}

// device.cu:379-386
__device__ inline V3 acesTonemap(V3 color) {
    color = color * (2.51f * color + 0.03f) / (color * (2.43f * color + 0.59f) + 0.14f);
    return {mathPow(fminf(fmaxf(color.x, 0.0f), 1.0f), 1.0f / 2.2f), mathPow(fminf(fmaxf(color.y, 0.0f), 1.0f), 1.0f / 2.2f),
            mathPow(fminf(fmaxf(color.z, 0.0f), 1.0f), 1.0f / 2.2f)};
}

// device.cu:399-413
__device__ inline float powerHeuristic(float a, float b) {
    const float a2 = a * a;
    const float b2 = b * b;
    return a2 / fmaxf(a2 + b2, 1e-12f);
}

__device__ inline float specularChoice(const rt_gpu_material& m, V3 view, V3 normal) {
    const V3 reflectance = schlickFresnel(normalReflectance(m), fmaxf(dot(view, normal), 0.0f));
    return fminf(fmaxf(maximum(reflectance), 0.05f), 0.95f);
}

// device.cu:433-447: the light at which the running remainder of the weights stops being positive;
// by the thresholds of the light index where there is one, else by the walk itself - the same light
// for the same random number either way.
__device__ inline const rt_gpu_light* pickLight(const rt_launch_params& p, const LightIndex& lightIndex, uint32_t& rng,
                                                float& probability) {
    if (p.lightCount == 0U || p.lightWeight <= 0.0f)
        return nullptr;
    float target = random01(rng) * p.lightWeight;
    const rt_gpu_light* selected = &p.lights[p.lightCount - 1U];
    if (lightIndex.thresholds && lightIndex.count == p.lightCount && p.lightCount > kLightWalkLimit) {
        selected = &p.lights[lightByThreshold(lightIndex, target)];
    } else {
        for (uint32_t index = 0; index < p.lightCount; ++index) {
            selected = &p.lights[index];
            target -= selected->weight;
            if (target <= 0.0f)
                break;
        }
    }
    probability = selected->weight / p.lightWeight;
    return selected;
}

struct EnvironmentDraw {
    V3 direction;
    V3 radiance;
    float pdf;
};

// device.cu:449-478
__device__ inline EnvironmentDraw drawEnvironment(const rt_launch_params& p, uint32_t& rng) {
    const uint32_t count = p.environmentWidth * p.environmentHeight;
    const float target = random01(rng);
    uint32_t first = 0;
    uint32_t last = count - 1U;
    while (first < last) {
        const uint32_t middle = (first + last) / 2U;
        if (p.environmentCdf[middle] < target)
            first = middle + 1U;
        else
            last = middle;
    }
    const uint32_t index = first;
    const float probability = p.environmentCdf[index] - (index ? p.environmentCdf[index - 1U] : 0.0f);
    const float u = (static_cast<float>(index % p.environmentWidth) + random01(rng)) / p.environmentWidth;
    const float v = (static_cast<float>(index / p.environmentWidth) + random01(rng)) / p.environmentHeight;
    const float theta = 3.14159265f * v;
    const float phi = 6.2831853f * (u - 0.5f) - p.environmentRotation;
    const float sine = mathSin(theta);
    const V3 direction{mathCos(phi) * sine, mathCos(theta), mathSin(phi) * sine};
    const float solidAngle = 19.7392088f * fmaxf(sine, 1e-6f) / static_cast<float>(count);
    return {direction, environmentRadiance(p, direction), probability / solidAngle};
}

struct LightDraw {  // device.cu:415-425
    V3 direction{0.0f, 0.0f, 0.0f};
    V3 radiance{0.0f, 0.0f, 0.0f};
    float distance = 0.0f;
    float pdf = 0.0f;
    uint32_t instance = 0;
    uint32_t primitive = 0;
    bool delta = false;
    bool environment = false;
    bool valid = false;
};

// device.cu:480-546
__device__ inline LightDraw drawLight(const rt_launch_params& p, const LightIndex& lightIndex, V3 position, uint32_t& rng) {
    LightDraw sample;
    float choice = 0.0f;
    const rt_gpu_light* selected = pickLight(p, lightIndex, rng, choice);
    if (!selected)
        return sample;
    const rt_gpu_light light = *selected;
    if (light.type == 4U) {
        const EnvironmentDraw source = drawEnvironment(p, rng);
        sample.direction = source.direction;
        sample.radiance = source.radiance;
        sample.distance = 1e16f;
        sample.pdf = choice * source.pdf;
        sample.environment = true;
        sample.valid = true;
        return sample;
    }
    if (light.type == 3U) {
        const float root = sqrtf(random01(rng));
        const float u = 1.0f - root;
        const float v = random01(rng) * root;
        const V3 point = v3(light.a) * u + v3(light.b) * v + v3(light.c) * (1.0f - u - v);
        const V3 offset = point - position;
        const float distanceSquared = dot(offset, offset);
        sample.distance = sqrtf(distanceSquared);
        sample.direction = offset / sample.distance;
        const float cosine = fabsf(dot(v3(light.normal), -sample.direction));
        if (cosine <= 1e-6f)
            return sample;
        sample.pdf = choice * distanceSquared / (light.area * cosine);
        sample.radiance = v3(light.emission);
        sample.instance = light.instance;
        sample.primitive = light.primitive;
        sample.valid = true;
        return sample;
    }
    sample.delta = true;
    sample.pdf = choice;
    sample.radiance = v3(light.emission);
    if (light.type == 0U) {
        sample.direction = -v3(light.b);
        sample.distance = 1e16f;
    } else {
        const V3 offset = v3(light.a) - position;
        const float distanceSquared = dot(offset, offset);
        sample.distance = sqrtf(distanceSquared);
        if (sample.distance >= light.range)
            return sample;
        sample.direction = offset / sample.distance;
        sample.radiance = sample.radiance / distanceSquared;
        if (light.type == 1U) {
            const float cone = dot(v3(light.b), -sample.direction);
            const float outer = mathCos(light.outerCone);
            const float inner = mathCos(light.innerCone);
            const float falloff = saturate((cone - outer) / fmaxf(inner - outer, 1e-5f));
            sample.radiance = sample.radiance * (falloff * falloff);
            if (falloff <= 0.0f)
                return sample;
        }
    }
    sample.valid = true;
    return sample;
}

// device.cu:561-578
__device__ inline float environmentDensity(const rt_launch_params& p, V3 direction) {
    if (!p.environment || p.lightWeight <= 0.0f)
        return 0.0f;
    float u, v;
    environmentCoordinates(p, direction, u, v);
    const uint32_t x = min(static_cast<uint32_t>(u * p.environmentWidth), p.environmentWidth - 1U);
    const uint32_t y = min(static_cast<uint32_t>(v * p.environmentHeight), p.environmentHeight - 1U);
    const uint32_t index = y * p.environmentWidth + x;
    const float probability = p.environmentCdf[index] - (index ? p.environmentCdf[index - 1U] : 0.0f);
    const float theta = 3.14159265f * (static_cast<float>(y) + 0.5f) / p.environmentHeight;
    const float solidAngle =
        19.7392088f * fmaxf(mathSin(theta), 1e-6f) / static_cast<float>(p.environmentWidth * p.environmentHeight);
    return p.environmentWeight / p.lightWeight * probability / solidAngle;
}

// device.cu:607-621: density of having sampled this emissive triangle from `origin`
__device__ inline float emissiveDensity(const rt_launch_params& p, const LightIndex& lightIndex, V3 origin, const rt_hit& hit) {
    const V3 offset = v3(hit.position) - origin;
    const float distanceSquared = dot(offset, offset);
    const V3 direction = offset / sqrtf(distanceSquared);
    if (lightIndex.keys && lightIndex.count == p.lightCount && p.lightCount > kLightWalkLimit) {
        // the first light of this triangle in table order, which is the one the walk below returns at
        const uint32_t index = lightOfTriangle(lightIndex, hit.instance, hit.primitive);
        if (index >= p.lightCount)
            return 0.0f;
        const rt_gpu_light& light = p.lights[index];
        const float cosine = fabsf(dot(v3(light.normal), -direction));
        return light.weight / p.lightWeight * distanceSquared / fmaxf(light.area * cosine, 1e-8f);
    }
    for (uint32_t index = 0; index < p.lightCount; ++index) {
        const rt_gpu_light& light = p.lights[index];
        if (light.type != 3U || light.instance != hit.instance || light.primitive != hit.primitive)
            continue;
        const float cosine = fabsf(dot(v3(light.normal), -direction));
        return light.weight / p.lightWeight * distanceSquared / fmaxf(light.area * cosine, 1e-8f);
    }
    return 0.0f;
}

// device.cu:630-650
__device__ inline V3 sampleAround(V3 axis, float cosine, uint32_t& rng) {
    const float sine = sqrtf(fmaxf(0.0f, 1.0f - cosine * cosine));
    const float angle = 6.2831853f * random01(rng);
    const V3 tangent = frameTangent(axis);
    return normalize(axis * cosine + tangent * (sine * mathCos(angle)) + cross(axis, tangent) * (sine * mathSin(angle)));
}

__device__ inline V3 sampleDisk(V3 normal, float radius, uint32_t& rng) {
    const float distance = sqrtf(random01(rng)) * radius;
    const float angle = 6.2831853f * random01(rng);
    const V3 tangent = frameTangent(normal);
    return tangent * (distance * mathCos(angle)) + cross(normal, tangent) * (distance * mathSin(angle));
}

// The rough-dielectric event shared by the camera path (device.cu:954-997) and
// the photon tracer (device.cu:798-841): dispersion channel, visible-normal
// draw, Fresnel choice between reflection and refraction.  `weight` is the path
// throughput or the photon power.  Returns false when the path ends.
__device__ inline bool dielectricEvent(const rt_gpu_material& m, const rt_hit& hit, uint32_t& rng, V3& origin,
                                       V3& direction, V3& weight, int& medium) {
    float ior = m.ior;
    if (m.dispersion > 0.0f) {
        const uint32_t channel = static_cast<uint32_t>(random01(rng) * 3.0f) % 3U;
        const float spread = (m.ior - 1.0f) * m.dispersion * 0.5f;
        ior += (static_cast<float>(channel) - 1.0f) * spread;
        weight = weight * (channel == 0U ? V3{3.0f, 0.0f, 0.0f}
                                         : channel == 1U ? V3{0.0f, 3.0f, 0.0f} : V3{0.0f, 0.0f, 3.0f});
    }
    const V3 view = -direction;
    const V3 normal = v3(hit.normal);
    const float alpha = alphaOf(m.roughness);
    const V3 microNormal = sampleVisibleNormal(view, normal, alpha, rng);
    const float eta = hit.frontFace ? 1.0f / ior : ior;
    const float cosine = fminf(fmaxf(dot(view, microNormal), 0.0f), 1.0f);
    V3 transmitted{0.0f, 0.0f, 0.0f};
    const bool totalReflection = !refractThrough(direction, microNormal, eta, transmitted);
    float reflectance = 1.0f;
    if (!totalReflection) {
        transmitted = normalize(transmitted);
        reflectance = dielectricFresnel(eta > 1.0f ? fabsf(dot(transmitted, microNormal)) : cosine, ior);
    }
    const float viewCosine = fabsf(dot(view, normal));
    if (totalReflection || random01(rng) < reflectance) {
        direction = reflectAbout(direction, microNormal);
        if (dot(direction, normal) <= 0.0f)
            return false;
        origin = v3(hit.position) + normal * 0.001f;
    } else {
        direction = transmitted;
        origin = v3(hit.position) - normal * 0.001f;
        medium = hit.frontFace ? static_cast<int>(hit.material) : -1;
    }
    weight = weight * maskingRatio(viewCosine, fabsf(dot(direction, normal)), alpha);
    return true;
}

// The closest-hit program, device.cu:1081-1145: attribute interpolation at
// (t, u, v) of triangle `primitive` of instance `instance`.  The object->world
// normal transform multiplies by the transpose of the world->object 3x3, the
// tangent by the object->world 3x3, each row as (a*x + b*y) + c*z.  The tangent has one reader, the normal
// map (shadingNormal): for a material without one it is left at zero and its arithmetic is not done.
__device__ inline void reconstructHit(const ShadeInstance* instances, const rt_hit_group_data* geometries,
                                      const rt_gpu_material* materials, V3 origin, V3 ray, float t, float bx, float by,
                                      uint32_t primitive, uint32_t instance, rt_hit& hit) {
    const ShadeInstance& record = instances[instance];
    struct { const rt_gpu_vertex* vertices; const rt_uint3* indices; uint32_t material; } geometry{
        record.vertices, record.indices, record.material};  // = geometries[record.geometry]
    const bool wantsTangent = materials[geometry.material].normalTexture.texture >= 0;
    const rt_uint3 triangle = geometry.indices[primitive];
    const rt_gpu_vertex v0 = geometry.vertices[triangle.x];
    const rt_gpu_vertex v1 = geometry.vertices[triangle.y];
    const rt_gpu_vertex v2 = geometry.vertices[triangle.z];
    const float b0 = 1.0f - bx - by;
    V3 normal = v3(v0.normal) * b0 + v3(v1.normal) * bx + v3(v2.normal) * by;
    if (dot(normal, normal) < 1e-12f)
        normal = cross(v3(v1.position) - v3(v0.position), v3(v2.position) - v3(v0.position));
    const V3 objectNormalSum = normal;
    const float* w2o = record.toObject;
    normal = normalize(V3{(w2o[0] * normal.x + w2o[4] * normal.y) + w2o[8] * normal.z,
                          (w2o[1] * normal.x + w2o[5] * normal.y) + w2o[9] * normal.z,
                          (w2o[2] * normal.x + w2o[6] * normal.y) + w2o[10] * normal.z});
    const bool frontFace = dot(normal, ray) < 0.0f;
    const V3 facing = frontFace ? normal : -normal;
    float tw = 1.0f;
    V3 tangent{0.0f, 0.0f, 0.0f};
    if (wantsTangent) {
        tw = v0.tangent.w * b0 + v1.tangent.w * bx + v2.tangent.w * by;
        tangent = V3{v0.tangent.x * b0 + v1.tangent.x * bx + v2.tangent.x * by,
                     v0.tangent.y * b0 + v1.tangent.y * bx + v2.tangent.y * by,
                     v0.tangent.z * b0 + v1.tangent.z * bx + v2.tangent.z * by};
        if (dot(tangent, tangent) < 1e-12f) {
            const V3 objectNormal = normalize(objectNormalSum);
            const float d1x = v1.uv.x - v0.uv.x, d1y = v1.uv.y - v0.uv.y;
            const float d2x = v2.uv.x - v0.uv.x, d2y = v2.uv.y - v0.uv.y;
            const float determinant = d1x * d2y - d1y * d2x;
            if (fabsf(determinant) > 1e-8f)
                tangent = ((v3(v1.position) - v3(v0.position)) * d2y - (v3(v2.position) - v3(v0.position)) * d1y) /
                          determinant;
            else
                tangent = cross(fabsf(objectNormal.x) > 0.5f ? V3{0.0f, 1.0f, 0.0f} : V3{1.0f, 0.0f, 0.0f}, objectNormal);
            tw = 1.0f;
        }
        const float* o2w = record.toWorld;
        tangent = V3{(o2w[0] * tangent.x + o2w[1] * tangent.y) + o2w[2] * tangent.z,
                     (o2w[4] * tangent.x + o2w[5] * tangent.y) + o2w[6] * tangent.z,
                     (o2w[8] * tangent.x + o2w[9] * tangent.y) + o2w[10] * tangent.z};
        tangent = normalize(tangent - facing * dot(tangent, facing));
    }
    hit.position = r3(origin + ray * t);
    hit.normal = r3(facing);
    hit.tangent.x = tangent.x;
    hit.tangent.y = tangent.y;
    hit.tangent.z = tangent.z;
    hit.tangent.w = tw < 0.0f ? -1.0f : 1.0f;
    hit.uv.x = v0.uv.x * b0 + v1.uv.x * bx + v2.uv.x * by;
    hit.uv.y = v0.uv.y * b0 + v1.uv.y * bx + v2.uv.y * by;
    hit.uv1.x = v0.uv1.x * b0 + v1.uv1.x * bx + v2.uv1.x * by;
    hit.uv1.y = v0.uv1.y * b0 + v1.uv1.y * bx + v2.uv1.y * by;
    hit.distance = t;
    hit.material = geometry.material;
    hit.instance = instance;
    hit.primitive = primitive;
    hit.frontFace = frontFace ? 1 : 0;
    hit.found = 1;
}

}  // namespace rt
