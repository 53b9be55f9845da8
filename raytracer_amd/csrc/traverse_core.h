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

// Tuning knobs (compile-time): minimum waves per SIMD the register allocator must
// leave room for, and whether children are pushed in distance order.
#ifndef RT_TRACE_MIN_WAVES
#define RT_TRACE_MIN_WAVES 6
#endif
#ifndef RT_TRACE_SORTED_PUSH
#define RT_TRACE_SORTED_PUSH 1
#endif

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

// Global-memory views: the node and triangle pointers of an instance come out of
// memory, so without the address space the compiler falls back to flat loads.
typedef unsigned int NativeUint4 __attribute__((ext_vector_type(4)));
struct GlobalUint4 {
    const NativeUint4 __attribute__((address_space(1))) * base;
    __device__ uint4 operator[](int i) const {
        const NativeUint4 v = base[i];
        return make_uint4(v.x, v.y, v.z, v.w);
    }
};
__device__ inline GlobalUint4 globalView(const void* p) {
    return GlobalUint4{(const NativeUint4 __attribute__((address_space(1))) *)(reinterpret_cast<uintptr_t>(p))};
}

// LDS views of the stack columns (a generic pointer would make every stack access a flat access).
typedef uint32_t __attribute__((address_space(3))) * LdsUint;
typedef float __attribute__((address_space(3))) * LdsFloat;
__device__ inline LdsUint ldsView(uint32_t* p) { return (LdsUint)p; }
__device__ inline LdsFloat ldsView(float* p) { return (LdsFloat)p; }

constexpr uint32_t kDone = 0xffffffffU;  // "no more work" reference (has the leaf bit set)

// Traces one ray.  stackRef / stackT point at this lane's column of two LDS
// arrays of kLdsStack rows and `stride` columns.  With anyHit the walk stops at
// the first accepted triangle (the result then only says hit or miss).  A hit
// is accepted when it beats (tmax, initInstance, initPrimitive) in the order
// (t, instance, primitive); callers without a prior candidate pass kNoHit ids.
//
// Shape of the loop: an inner loop walks wide nodes until the lane holds a leaf,
// then the leaf is processed; the lanes of a wave therefore run the node code
// together and the triangle code together instead of alternating per step.
template <bool COUNT>
__device__ inline TraceResult traceRay(const WideNode* __restrict__ topNodes,
                                       const InstanceRecord* __restrict__ instances, const Ray& world, float tmin,
                                       float tmax, uint32_t initInstance, uint32_t initPrimitive, bool anyHit,
                                       LdsUint stackRef, LdsFloat stackT, int stride, TraceCounters& counters) {
    uint32_t privateRef[kPrivateStack];
    float privateT[kPrivateStack];
    float bestT = tmax, bestU = 0.0f, bestV = 0.0f;
    uint32_t bestPrim = initPrimitive, bestInst = initInstance;
    bool found = false;

    constexpr float kShrink = 1.0f - 0x1p-20f, kGrow = 1.0f + 0x1p-20f;
    Ray cur = world;
    Shear shear{};
    float idx = __builtin_amdgcn_rcpf(cur.dx), idy = __builtin_amdgcn_rcpf(cur.dy), idz = __builtin_amdgcn_rcpf(cur.dz);
    const WideNode* nodes = topNodes;
    const PackedTriangle* triangles = nullptr;
    uint32_t instanceId = kNoHit;
    bool inBottom = false;
    const float tminShrunk = tmin * kShrink;

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
    // Next entry that can still matter, or kDone.  Leaving an instance restores
    // the world-space ray.
    auto pop = [&]() -> uint32_t {
        for (;;) {
            if (sp == 0)
                return kDone;
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
                idx = __builtin_amdgcn_rcpf(cur.dx);
                idy = __builtin_amdgcn_rcpf(cur.dy);
                idz = __builtin_amdgcn_rcpf(cur.dz);
                nodes = topNodes;
                inBottom = false;
                continue;
            }
            if (t > bestT * kGrow)
                continue;  // fell behind the closest hit found since
            return ref;
        }
    };

    uint32_t node = 0;  // current reference; the top-level root is node 0
    for (;;) {
        // ---- wide nodes, until this lane holds a leaf ----
        while (!(node & kLeafBit)) {
            if (COUNT)
                ++counters.nodes;
            GlobalUint4 np = globalView(nodes + node);
            const uint4 n0 = np[0], n1 = np[1], n2 = np[2], n3 = np[3], n4 = np[4];
            const uint32_t imask = n0.w >> 24;
            // child plane distance = q * a + b per axis; the near planes are shrunk
            // and the far planes grown by 2^-20 relative, so rounding never rejects
            // a box whose triangle the intersection test accepts
            const float ax = __uint_as_float((n0.w & 0xffU) << 23) * idx;
            const float ay = __uint_as_float(((n0.w >> 8) & 0xffU) << 23) * idy;
            const float az = __uint_as_float(((n0.w >> 16) & 0xffU) << 23) * idz;
            const float bx = (__uint_as_float(n0.x) - cur.ox) * idx;
            const float by = (__uint_as_float(n0.y) - cur.oy) * idy;
            const float bz = (__uint_as_float(n0.z) - cur.oz) * idz;
            const float axn = ax * kShrink, ayn = ay * kShrink, azn = az * kShrink;
            const float bxn = bx * kShrink, byn = by * kShrink, bzn = bz * kShrink;
            const float axf = ax * kGrow, ayf = ay * kGrow, azf = az * kGrow;
            const float bxf = bx * kGrow, byf = by * kGrow, bzf = bz * kGrow;
            const float limit = bestT * kGrow;
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
                const float tn = fmaxf(fmaxf(fmaf(qxn, axn, bxn), fmaf(qyn, ayn, byn)), fmaxf(fmaf(qzn, azn, bzn), tminShrunk));
                const float tf = fminf(fminf(fmaf(qxf, axf, bxf), fmaf(qyf, ayf, byf)), fminf(fmaf(qzf, azf, bzf), limit));
                const uint32_t meta = ((i < 4 ? n1.z : n1.w) >> sh) & 0xffU;
                const bool present = ((imask >> i) & 1U) || (meta >> 5) != 0U;
                tnear[i] = tn;
                hitMask |= (present && tn <= tf) ? (1U << i) : 0U;
            }
            auto refOf = [&](int i) -> uint32_t {
                const uint32_t meta = byteOf(n1.z, n1.w, i);
                if ((imask >> i) & 1U)
                    return n1.x + meta;
                return kLeafBit | (meta >> 5) << 28 | (n1.y + (meta & 31U));
            };
            if (hitMask == 0U) {
                node = pop();
            } else {
#if RT_TRACE_SORTED_PUSH
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
#else
                // enter the nearest, push the others in slot order
                float nearest = INFINITY;
                int which = 0;
#pragma unroll
                for (int i = 0; i < kWide; ++i)
                    if (((hitMask >> i) & 1U) && tnear[i] < nearest) {
                        nearest = tnear[i];
                        which = i;
                    }
                hitMask &= ~(1U << which);
                while (hitMask) {
                    const int i = __ffs(static_cast<int>(hitMask)) - 1;
                    hitMask &= hitMask - 1U;
                    float t = tnear[0];
#pragma unroll
                    for (int k = 1; k < kWide; ++k)
                        t = i == k ? tnear[k] : t;
                    push(refOf(i), t);
                }
                node = refOf(which);
#endif
            }
        }
        if (node == kDone)
            break;
        if (!inBottom) {
            // ---- top-level leaf: enter the instance ----
            if (COUNT)
                ++counters.instances;
            const InstanceRecord* record = instances + (node & 0x0fffffffU);
            GlobalUint4 m = globalView(record);
            const uint4 q0 = m[0], q1 = m[1], q2 = m[2], q3 = m[3], q4 = m[4];
            const float4 r0 = make_float4(__uint_as_float(q0.x), __uint_as_float(q0.y), __uint_as_float(q0.z), __uint_as_float(q0.w));
            const float4 r1 = make_float4(__uint_as_float(q1.x), __uint_as_float(q1.y), __uint_as_float(q1.z), __uint_as_float(q1.w));
            const float4 r2 = make_float4(__uint_as_float(q2.x), __uint_as_float(q2.y), __uint_as_float(q2.z), __uint_as_float(q2.w));
            push(kSentinel, 0.0f);
            cur.ox = ((r0.x * world.ox + r0.y * world.oy) + r0.z * world.oz) + r0.w;
            cur.oy = ((r1.x * world.ox + r1.y * world.oy) + r1.z * world.oz) + r1.w;
            cur.oz = ((r2.x * world.ox + r2.y * world.oy) + r2.z * world.oz) + r2.w;
            cur.dx = (r0.x * world.dx + r0.y * world.dy) + r0.z * world.dz;
            cur.dy = (r1.x * world.dx + r1.y * world.dy) + r1.z * world.dz;
            cur.dz = (r2.x * world.dx + r2.y * world.dy) + r2.z * world.dz;
            shear = makeShear(cur);
            idx = __builtin_amdgcn_rcpf(cur.dx);
            idy = __builtin_amdgcn_rcpf(cur.dy);
            idz = __builtin_amdgcn_rcpf(cur.dz);
            nodes = reinterpret_cast<const WideNode*>(static_cast<uintptr_t>(q3.x) | static_cast<uintptr_t>(q3.y) << 32);
            triangles = reinterpret_cast<const PackedTriangle*>(static_cast<uintptr_t>(q3.z) |
                                                                static_cast<uintptr_t>(q3.w) << 32);
            instanceId = q4.x;
            inBottom = true;
            node = 0;
            continue;
        }
        // ---- bottom-level leaf: triangles ----
        const uint32_t count = (node >> 28) & 7U;
        const uint32_t first = node & 0x0fffffffU;
        for (uint32_t k = 0; k < count; ++k) {
            if (COUNT)
                ++counters.triangles;
            GlobalUint4 tp = globalView(triangles + first + k);
            const uint4 p0 = tp[0], p1 = tp[1], p2 = tp[2];
            float t, u, v;
            if (intersectTriangle(cur, shear, p0, p1, p2, tmin, bestT, t, u, v)) {
                const uint32_t prim = p0.w;
                // intersectTriangle only returns t <= bestT
                if (t < bestT || instanceId < bestInst || (instanceId == bestInst && prim < bestPrim)) {
                    bestT = t;
                    bestU = u;
                    bestV = v;
                    bestPrim = prim;
                    bestInst = instanceId;
                    found = true;
                }
            }
        }
        if (anyHit && found)
            break;
        node = pop();
    }
    if (!found)
        return {tmax, 0.0f, 0.0f, kNoHit, kNoHit};
    return {bestT, bestU, bestV, bestPrim, bestInst};
}

}  // namespace rt
