// Per-lane traversal of the two-level 8-wide BVH, shared by every kernel that traces rays.
//
// Replaces the optixTrace call of the reference (device.cu:388-397) and the
// occlusion use of it in visible() (device.cu:548-559).  One ray per lane; the
// traversal stack's first entries live in LDS (one column per lane, so a wave's
// accesses never share a bank), the rest in private memory; entries carry the
// child's entry distance so a subtree that fell behind the current hit is
// dropped without touching memory.
//
// The ray/triangle arithmetic is the watertight test of Woop, Benthin and Wald
// (2013) in a fixed operation order, every operation individually rounded, so
// that t, the barycentrics and the winning primitive equal a scalar evaluation
// bit for bit:
//   object-space ray: o' = ((m0*ox + m1*oy) + m2*oz) + m3 per row, d' alike;
//   kz = largest |d'| component (first wins ties), kx, ky the next two in
//   cyclic order, swapped when d'[kz] < 0; Sx = d'[kx]/d'[kz], Sy = d'[ky]/d'[kz],
//   Sz = 1/d'[kz];
//   A = v0 - o' (B, C alike); Ax = A[kx] - Sx*A[kz]; Ay = A[ky] - Sy*A[kz];
//   U = Cx*By - Cy*Bx; V = Ax*Cy - Ay*Cx; W = Bx*Ay - By*Ax, recomputed with
//   binary64 products and difference when any of them is zero;
//   miss when U, V, W have mixed signs; det = (U + V) + W, miss when zero;
//   T = (U*(Sz*A[kz]) + V*(Sz*B[kz])) + W*(Sz*C[kz]); t = T * (1/det);
//   hit when tmin < t <= tmax; u = V/det, v = W/det as products with 1/det.
//   Closest hit: smallest t, exact ties to the lowest (instance, primitive).
#pragma once

#include "bvh.h"

namespace rt {

constexpr int kTraceBlock = 256;
constexpr int kLdsStack = 12;       // entries per lane kept in LDS
constexpr int kPrivateStack = 116;  // overflow entries per lane in private memory
constexpr int kMaxStack = kLdsStack + kPrivateStack;
constexpr uint32_t kLeafBit = 0x80000000U;
constexpr uint32_t kSentinel = 0x7fffffffU;  // marks the return from an instance

struct Ray {
    float ox, oy, oz;
    float dx, dy, dz;
};

struct Shear {
    int kx, ky, kz;
    float Sx, Sy, Sz;
};

__device__ inline float pick(int k, float x, float y, float z) {
    return k == 0 ? x : (k == 1 ? y : z);
}

__device__ inline Shear makeShear(const Ray& r) {
    Shear s;
    int kz = 0;
    if (fabsf(r.dy) > fabsf(pick(kz, r.dx, r.dy, r.dz)))
        kz = 1;
    if (fabsf(r.dz) > fabsf(pick(kz, r.dx, r.dy, r.dz)))
        kz = 2;
    int kx = kz == 2 ? 0 : kz + 1;
    int ky = kx == 2 ? 0 : kx + 1;
    const float dz = pick(kz, r.dx, r.dy, r.dz);
    if (dz < 0.0f) {
        const int tmp = kx;
        kx = ky;
        ky = tmp;
    }
    s.kx = kx;
    s.ky = ky;
    s.kz = kz;
    s.Sx = pick(kx, r.dx, r.dy, r.dz) / dz;
    s.Sy = pick(ky, r.dx, r.dy, r.dz) / dz;
    s.Sz = 1.0f / dz;
    return s;
}

// Returns true and t, u, v when the triangle is hit inside (tmin, tmax].
__device__ inline bool intersectTriangle(const Ray& r, const Shear& s, const uint4& p0, const uint4& p1,
                                         const uint4& p2, float tmin, float tmax, float& t, float& u, float& v) {
    const float ax = __uint_as_float(p0.x) - r.ox, ay = __uint_as_float(p0.y) - r.oy, az = __uint_as_float(p0.z) - r.oz;
    const float bx = __uint_as_float(p1.x) - r.ox, by = __uint_as_float(p1.y) - r.oy, bz = __uint_as_float(p1.z) - r.oz;
    const float cx = __uint_as_float(p2.x) - r.ox, cy = __uint_as_float(p2.y) - r.oy, cz = __uint_as_float(p2.z) - r.oz;
    const float Akz = pick(s.kz, ax, ay, az), Bkz = pick(s.kz, bx, by, bz), Ckz = pick(s.kz, cx, cy, cz);
    const float Ax = pick(s.kx, ax, ay, az) - s.Sx * Akz;
    const float Ay = pick(s.ky, ax, ay, az) - s.Sy * Akz;
    const float Bx = pick(s.kx, bx, by, bz) - s.Sx * Bkz;
    const float By = pick(s.ky, bx, by, bz) - s.Sy * Bkz;
    const float Cx = pick(s.kx, cx, cy, cz) - s.Sx * Ckz;
    const float Cy = pick(s.ky, cx, cy, cz) - s.Sy * Ckz;
    float U = Cx * By - Cy * Bx;
    float V = Ax * Cy - Ay * Cx;
    float W = Bx * Ay - By * Ax;
    if (U == 0.0f || V == 0.0f || W == 0.0f) {
        U = static_cast<float>(static_cast<double>(Cx) * static_cast<double>(By) -
                               static_cast<double>(Cy) * static_cast<double>(Bx));
        V = static_cast<float>(static_cast<double>(Ax) * static_cast<double>(Cy) -
                               static_cast<double>(Ay) * static_cast<double>(Cx));
        W = static_cast<float>(static_cast<double>(Bx) * static_cast<double>(Ay) -
                               static_cast<double>(By) * static_cast<double>(Ax));
    }
    if ((U < 0.0f || V < 0.0f || W < 0.0f) && (U > 0.0f || V > 0.0f || W > 0.0f))
        return false;
    const float det = (U + V) + W;
    if (det == 0.0f)
        return false;
    const float Az = s.Sz * Akz;
    const float Bz = s.Sz * Bkz;
    const float Cz = s.Sz * Ckz;
    const float T = (U * Az + V * Bz) + W * Cz;
    const float rcp = 1.0f / det;
    const float tt = T * rcp;
    if (!(tt > tmin && tt <= tmax))
        return false;
    t = tt;
    u = V * rcp;
    v = W * rcp;
    return true;
}

__device__ inline uint32_t byteOf(uint32_t lo, uint32_t hi, int i) {
    const unsigned long long both = static_cast<unsigned long long>(hi) << 32 | lo;
    return static_cast<uint32_t>(both >> (8 * i)) & 0xffU;
}

// Workgroups that share an XCD (and its L2) get neighbouring ray ranges.
__device__ inline uint32_t xcdRemap(uint32_t block, uint32_t blocks) {
    const uint32_t xcd = block & 7U, q = blocks >> 3, r = blocks & 7U;
    return (xcd < r ? xcd * (q + 1U) : r * (q + 1U) + (xcd - r) * q) + (block >> 3);
}

struct TraceCounters {
    unsigned long long nodes = 0, triangles = 0, instances = 0;
};

struct TraceResult {
    float t, u, v;
    uint32_t primitive, instance;  // kNoHit on a miss
};

// Traces one ray.  stackRef / stackT point at this lane's column of two LDS
// arrays of kLdsStack rows and `stride` columns.  With anyHit the walk stops at
// the first accepted triangle (the result then only says hit or miss).
template <bool COUNT>
__device__ inline TraceResult traceRay(const WideNode* __restrict__ topNodes,
                                       const InstanceRecord* __restrict__ instances, const Ray& world, float tmin,
                                       float tmax, bool anyHit, uint32_t* stackRef, float* stackT, int stride,
                                       TraceCounters& counters) {
    uint32_t privateRef[kPrivateStack];
    float privateT[kPrivateStack];
    float bestT = tmax, bestU = 0.0f, bestV = 0.0f;
    uint32_t bestPrim = kNoHit, bestInst = kNoHit;

    Ray cur = world;
    Shear shear{};
    float idx = 1.0f / cur.dx, idy = 1.0f / cur.dy, idz = 1.0f / cur.dz;
    const WideNode* nodes = topNodes;
    const PackedTriangle* triangles = nullptr;
    uint32_t instanceId = kNoHit;
    bool inBottom = false;
    bool done = false;

    int sp = 0;
    auto push = [&](uint32_t ref, float t) {
        if (sp < kLdsStack) {
            stackRef[sp * stride] = ref;
            stackT[sp * stride] = t;
        } else if (sp < kMaxStack) {
            privateRef[sp - kLdsStack] = ref;
            privateT[sp - kLdsStack] = t;
        }
        ++sp;
    };

    uint32_t node = 0;  // current reference; the top-level root is node 0
    while (!done) {
        if (!(node & kLeafBit)) {
            // ---- wide node: test the eight quantised child boxes ----
            if (COUNT)
                ++counters.nodes;
            const uint4* np = reinterpret_cast<const uint4*>(nodes + node);
            const uint4 n0 = np[0], n1 = np[1], n2 = np[2], n3 = np[3], n4 = np[4];
            const uint32_t imask = n0.w >> 24;
            const float ax = __uint_as_float((n0.w & 0xffU) << 23) * idx;
            const float ay = __uint_as_float(((n0.w >> 8) & 0xffU) << 23) * idy;
            const float az = __uint_as_float(((n0.w >> 16) & 0xffU) << 23) * idz;
            const float bx = (__uint_as_float(n0.x) - cur.ox) * idx;
            const float by = (__uint_as_float(n0.y) - cur.oy) * idy;
            const float bz = (__uint_as_float(n0.z) - cur.oz) * idz;
            // near / far byte planes per axis, chosen by the direction sign
            const bool nx = idx < 0.0f, ny = idy < 0.0f, nz = idz < 0.0f;
            const uint32_t xlo0 = nx ? n3.z : n2.x, xlo1 = nx ? n3.w : n2.y;
            const uint32_t xhi0 = nx ? n2.x : n3.z, xhi1 = nx ? n2.y : n3.w;
            const uint32_t ylo0 = ny ? n4.x : n2.z, ylo1 = ny ? n4.y : n2.w;
            const uint32_t yhi0 = ny ? n2.z : n4.x, yhi1 = ny ? n2.w : n4.y;
            const uint32_t zlo0 = nz ? n4.z : n3.x, zlo1 = nz ? n4.w : n3.y;
            const uint32_t zhi0 = nz ? n3.x : n4.z, zhi1 = nz ? n3.y : n4.w;
            float tnear[kWide];
            uint32_t hitMask = 0;
    #pragma unroll
            for (int i = 0; i < kWide; ++i) {
                const int sh = 8 * (i & 3);
                const float qxn = static_cast<float>(((i < 4 ? xlo0 : xlo1) >> sh) & 0xffU);
                const float qxf = static_cast<float>(((i < 4 ? xhi0 : xhi1) >> sh) & 0xffU);
                const float qyn = static_cast<float>(((i < 4 ? ylo0 : ylo1) >> sh) & 0xffU);
                const float qyf = static_cast<float>(((i < 4 ? yhi0 : yhi1) >> sh) & 0xffU);
                const float qzn = static_cast<float>(((i < 4 ? zlo0 : zlo1) >> sh) & 0xffU);
                const float qzf = static_cast<float>(((i < 4 ? zhi0 : zhi1) >> sh) & 0xffU);
                // fmaxf / fminf drop a NaN operand (0 * inf on an axis the ray
                // is parallel to), which leaves that axis unconstrained
                float tn = fmaxf(fmaxf(fmaf(qxn, ax, bx), fmaf(qyn, ay, by)), fmaxf(fmaf(qzn, az, bz), tmin));
                float tf = fminf(fminf(fmaf(qxf, ax, bx), fmaf(qyf, ay, by)), fminf(fmaf(qzf, az, bz), bestT));
                const uint32_t meta = ((i < 4 ? n1.z : n1.w) >> sh) & 0xffU;
                const bool present = ((imask >> i) & 1U) || (meta >> 5) != 0U;
                // widened by 2^-20 relative so rounding never rejects a box
                // whose triangle the intersection test accepts
                const bool hit = present && tn * (1.0f - 0x1p-20f) <= tf * (1.0f + 0x1p-20f);
                tnear[i] = tn;
                hitMask |= hit ? (1U << i) : 0U;
            }
            auto refOf = [&](int i) -> uint32_t {
                const uint32_t meta = byteOf(n1.z, n1.w, i);
                if ((imask >> i) & 1U)
                    return n1.x + meta;
                return kLeafBit | (meta >> 5) << 28 | (n1.y + (meta & 31U));
            };
            if (hitMask == 0U) {
                node = kSentinel;  // nothing to enter: pop below
            } else {
                // push all but the nearest, farthest first
                while (hitMask & (hitMask - 1U)) {
                    float far = -INFINITY;
                    int which = 0;
    #pragma unroll
                    for (int i = 0; i < kWide; ++i)
                        if (((hitMask >> i) & 1U) && tnear[i] >= far) {
                            far = tnear[i];
                            which = i;
                        }
                    push(refOf(which), far);
                    hitMask &= ~(1U << which);
                }
                node = refOf(__ffs(static_cast<int>(hitMask)) - 1);
                continue;
            }
        } else if (!inBottom) {
            // ---- top-level leaf: enter the instance ----
            if (COUNT)
                ++counters.instances;
            const InstanceRecord* record = instances + (node & 0x0fffffffU);
            const float4* m = reinterpret_cast<const float4*>(record->toObject);
            const float4 r0 = m[0], r1 = m[1], r2 = m[2];
            push(kSentinel, 0.0f);
            cur.ox = ((r0.x * world.ox + r0.y * world.oy) + r0.z * world.oz) + r0.w;
            cur.oy = ((r1.x * world.ox + r1.y * world.oy) + r1.z * world.oz) + r1.w;
            cur.oz = ((r2.x * world.ox + r2.y * world.oy) + r2.z * world.oz) + r2.w;
            cur.dx = (r0.x * world.dx + r0.y * world.dy) + r0.z * world.dz;
            cur.dy = (r1.x * world.dx + r1.y * world.dy) + r1.z * world.dz;
            cur.dz = (r2.x * world.dx + r2.y * world.dy) + r2.z * world.dz;
            shear = makeShear(cur);
            idx = 1.0f / cur.dx;
            idy = 1.0f / cur.dy;
            idz = 1.0f / cur.dz;
            nodes = record->nodes;
            triangles = record->triangles;
            instanceId = record->instanceId;
            inBottom = true;
            node = 0;
            continue;
        } else {
            // ---- bottom-level leaf: triangles ----
            const uint32_t count = (node >> 28) & 7U;
            const uint32_t first = node & 0x0fffffffU;
            for (uint32_t k = 0; k < count; ++k) {
                if (COUNT)
                    ++counters.triangles;
                const uint4* tp = reinterpret_cast<const uint4*>(triangles + first + k);
                const uint4 p0 = tp[0], p1 = tp[1], p2 = tp[2];
                float t, u, v;
                if (intersectTriangle(cur, shear, p0, p1, p2, tmin, bestT, t, u, v)) {
                    const uint32_t prim = p0.w;
                    if (t < bestT || instanceId < bestInst || (instanceId == bestInst && prim < bestPrim)) {
                        bestT = t;
                        bestU = u;
                        bestV = v;
                        bestPrim = prim;
                        bestInst = instanceId;
                        if (anyHit)
                            done = true;
                    }
                }
            }
            if (anyHit && done)
                break;
        }
        // ---- pop ----
        for (;;) {
            if (sp == 0) {
                done = true;
                break;
            }
            --sp;
            if (sp >= kMaxStack)
                continue;  // never pushed (guarded against on the host as well)
            uint32_t ref;
            float t;
            if (sp < kLdsStack) {
                ref = stackRef[sp * stride];
                t = stackT[sp * stride];
            } else {
                ref = privateRef[sp - kLdsStack];
                t = privateT[sp - kLdsStack];
            }
            if (ref == kSentinel) {
                cur = world;
                idx = 1.0f / cur.dx;
                idy = 1.0f / cur.dy;
                idz = 1.0f / cur.dz;
                nodes = topNodes;
                inBottom = false;
                continue;
            }
            if (t * (1.0f - 0x1p-20f) > bestT * (1.0f + 0x1p-20f))
                continue;  // fell behind the closest hit found since
            node = ref;
            break;
        }
    }
    return {bestT, bestU, bestV, bestPrim, bestInst};
}

}  // namespace rt
