// Per-lane traversal of the two-level 8-wide BVH, shared by every kernel that traces rays.
//
// Replaces the optixTrace call of the reference (device.cu:388-397) and the
// occlusion use of it in visible() (device.cu:548-559).  One ray per lane at a
// time, in persistent waves that hand a lane a new ray when its ray is done; the
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

#include <atomic>

namespace rt {

// Tuning knob (compile-time): minimum waves per SIMD the register allocator must
// leave room for.
#ifndef RT_TRACE_MIN_WAVES
#define RT_TRACE_MIN_WAVES 5
#endif
// Persistent launches: how many idle lanes make a wave fetch new rays.
#ifndef RT_TRACE_REFILL
#define RT_TRACE_REFILL 16
#endif
// Triangles a lane may test per turn.
#ifndef RT_TRACE_LEAF_STEPS
#define RT_TRACE_LEAF_STEPS 1
#endif
// Workgroups launched per resident workgroup slot (1 = exactly one resident set).
// Whether a lane takes a stack entry a second time per turn, behind the leaf code.
#ifndef RT_TRACE_POP_AFTER_LEAF
#define RT_TRACE_POP_AFTER_LEAF 1
#endif
// Whether a refill rides along a turn (the atomic that draws ray indices travels during the node
// step, the ray loads during the stack and leaf steps) instead of stalling the wave twice.
#ifndef RT_TRACE_RIDING_REFILL
#define RT_TRACE_RIDING_REFILL 1
#endif
#ifndef RT_TRACE_GRID_SCALE
#define RT_TRACE_GRID_SCALE 1
#endif
// Whether a lane whose node step found no child takes its next stack entry before the leaf part of the same turn
// (a leaf it finds there is then tested in this turn) or only at the end of the turn.
// 1: the child box test takes the near and the far plane of an axis through one packed multiply-add (measured
// slower: a packed multiply-add costs more issue time than the two plain ones it replaces, 1080p step 1.77 against 1.73 ms)
#ifndef RT_TRACE_PACKED_FMA
#define RT_TRACE_PACKED_FMA 0
#endif
#ifndef RT_TRACE_POP_AFTER_NODE
#define RT_TRACE_POP_AFTER_NODE 1
#endif
#ifndef RT_TRACE_BLOCK
#define RT_TRACE_BLOCK 256
#endif
constexpr int kTraceBlock = RT_TRACE_BLOCK;  // lanes of a ray-kernel workgroup (128 and 512 measured no better)
#ifndef RT_TRACE_LDS_STACK
#define RT_TRACE_LDS_STACK 15
#endif
constexpr int kLdsStack = RT_TRACE_LDS_STACK;  // entries per lane kept in LDS
constexpr int kPrivateStack = 128 - kLdsStack;  // overflow entries per lane in private memory
constexpr int kMaxStack = kLdsStack + kPrivateStack;
static_assert(kMaxStack == static_cast<int>(kTraversalStackEntries), "fitsTraversalStack() assumes this stack size");
constexpr uint32_t kLeafBit = 0x80000000U;

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
typedef float Pair __attribute__((ext_vector_type(2)));  // two floats in a register pair, for packed arithmetic
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
// "take the next stack entry": has the leaf bit set and is no real leaf, whose
// item index (bits 0-27) stays below 2^28 - 2
constexpr uint32_t kPop = 0xfffffffeU;

// Development aid (-DRT_TRACE_PROFILE): per code site, how many times a wave
// issued it and with how many lanes, summed over every traversal kernel.
#ifdef RT_TRACE_PROFILE
static __device__ float gTraceProfile[64];  // one copy per translation unit
static __device__ unsigned int gTraceHistogram[64];  // per kernel: rays by log2(steps), 16 bins
// launch timeline in 100 MHz ticks: [0] earliest wave start, [1] earliest sight of an empty feed,
// [2] latest wave end; [3] wave turns after the feed emptied, [4] lanes active in them
static __device__ unsigned long long gTraceClock[8];
#define RT_SITE(k)                                                             \
    do {                                                                       \
        if ((k) == 0 || (k) == 3)                                              \
            ++raySteps;                                                        \
        const float lanes = static_cast<float>(__popcll(__ballot(true)));      \
        profile[2 * (k)] += 1.0f / lanes;                                      \
        profile[2 * (k) + 1] += 1.0f;                                          \
    } while (0)
#else
#define RT_SITE(k) ((void)0)
#endif

// Every index the walk takes out of memory (node, instance and triangle
// references) is checked against the size of the array it points into before
// it is used; a reference that fails ends the ray as a miss and is counted in
// `violations`, so a damaged acceleration structure shows up as a number
// instead of a memory fault.
#define RT_GUARD(ok)                                                           \
    do {                                                                       \
        if (!(ok)) {                                                           \
            ++violations;                                                      \
            found = false;                                                     \
            node = kDone;                                                      \
            return;                                                            \
        }                                                                      \
    } while (0)

// Traversal state of one lane.  The walk is resumable: begin() takes a ray and
// nodeStep() / enterInstance() / triangleStep() each do one unit of work and
// leave the next reference in `node` (kDone when the ray is finished).  That lets a wave hand a new ray to a lane whose ray ended early instead of
// idling it until the slowest ray of the wave is done (see RayFeed below).
//
// stackRef / stackT point at this lane's column of two LDS arrays of kLdsStack
// rows and `stride` columns; deeper entries spill to private memory.  With anyHit
// (per ray) the walk stops at the first accepted triangle (the result then only says hit
// or miss).  A hit is accepted when it beats (tmax, initInstance, initPrimitive)
// in the order (t, instance, primitive); callers without a prior candidate pass
// kNoHit ids.
template <bool COUNT>
struct Traversal {
    static constexpr float kShrink = 1.0f - 0x1p-20f, kGrow = 1.0f + 0x1p-20f;

    const WideNode* topNodes;
    const InstanceRecord* instances;
    LdsUint stackRef;
    LdsFloat stackT;
    int stride;

    Ray world, cur;
    Shear shear;
    // shear constants of the ray in the top level (where folded triangles are tested), packed: the walk goes
    // back to them when it leaves an instance
    float topSx, topSy, topSz;
    int topAxes;
    float idx, idy, idz;
    const WideNode* nodes;
    const PackedTriangle* triangles;
    uint32_t instanceId;
    bool found, anyHit;
    // Stack height at which the ray entered the instance it is in, -1 while it is in the top level: entries
    // below it belong to the top level, so taking one of them off means the instance is behind the ray.
    int floor;
    __device__ __forceinline__ bool inBottom() const { return floor >= 0; }
    float tmin, tminShrunk, tmax;
    float bestT, bestU, bestV;
    uint32_t bestPrim, bestInst;
    int sp;
    uint32_t node;
    // overflow stack: arrays owned by the kernel (kept out of this struct so that
    // the rest of the state is promoted to registers)
    uint32_t (&privateRef)[kPrivateStack];
    float (&privateT)[kPrivateStack];
    TraceCounters counters;
    uint32_t topNodeLimit, instanceLimit;  // sizes of the top-level arrays
    uint32_t nodeLimit, itemLimit;         // sizes of the node and item arrays of the level the ray is in
    uint32_t violations;
    bool hasFolded;  // the same for every lane: the top level holds folded triangles, so the walk keeps shear constants for it
#ifdef RT_TRACE_PROFILE
    float profile[16] = {};
    int profileBase = 0;
    uint32_t raySteps = 0;
    __device__ void finishRay() {
        const int bin = raySteps ? min(31 - __clz(static_cast<int>(raySteps)), 15) : 0;
        atomicAdd(&gTraceHistogram[profileBase + bin], 1U);
        raySteps = 0;
    }
    __device__ void flushProfile() {
        for (int k = 0; k < 16; ++k) {
            float v = profile[k];
            for (int delta = 32; delta >= 1; delta >>= 1)
                v += __shfl_down(v, delta, 64);
            if ((threadIdx.x & 63) == 0)
                atomicAdd(&gTraceProfile[profileBase + k], v);
        }
    }
#endif

    // topCount / instanceCount: entries of `top` and `inst` (see itemCountWord: the top bit of instanceCount
    // says whether folded triangles are among them).  Every member starts defined.
    __device__ __forceinline__ Traversal(const WideNode* top, uint32_t topCount, const InstanceRecord* inst,
                                         uint32_t instanceCount, LdsUint refs, LdsFloat ts, int s,
                                         uint32_t (&spillRef)[kPrivateStack], float (&spillT)[kPrivateStack])
        : topNodes(top), instances(inst), stackRef(refs), stackT(ts), stride(s), world{}, cur{}, shear{}, topSx(0.0f), topSy(0.0f), topSz(0.0f), topAxes(0), idx(0.0f),
          idy(0.0f), idz(0.0f), nodes(top), triangles(nullptr), instanceId(kNoHit), found(false),
          anyHit(false), floor(-1), tmin(0.0f), tminShrunk(0.0f), tmax(0.0f), bestT(0.0f), bestU(0.0f), bestV(0.0f),
          bestPrim(kNoHit), bestInst(kNoHit), sp(0), node(kDone), privateRef(spillRef), privateT(spillT),
          topNodeLimit(topCount), instanceLimit(instanceCount & 0x7fffffffU), nodeLimit(topCount),
          itemLimit(instanceCount & 0x7fffffffU), violations(0), hasFolded((instanceCount >> 31) != 0U) {}

    // The ray in the top level: the world-space ray through the identity transform, operation by operation as
    // enterInstance() applies a transform, because that is what a folded triangle's instance would have made of
    // it (the same values; a negative zero or a component that is not finite come out as they would there).  The
    // translation such an instance adds last is -0 (rt_scene_build folds no other), which changes nothing.
    __device__ __forceinline__ void toTopLevel() {
        cur.ox = (world.ox + 0.0f * world.oy) + 0.0f * world.oz;
        cur.oy = (0.0f * world.ox + world.oy) + 0.0f * world.oz;
        cur.oz = (0.0f * world.ox + 0.0f * world.oy) + world.oz;
        cur.dx = (world.dx + 0.0f * world.dy) + 0.0f * world.dz;
        cur.dy = (0.0f * world.dx + world.dy) + 0.0f * world.dz;
        cur.dz = (0.0f * world.dx + 0.0f * world.dy) + world.dz;
    }

    __device__ __forceinline__ void begin(const Ray& ray, float rayTmin, float rayTmax, uint32_t initInstance,
                                          uint32_t initPrimitive, bool stopAtFirstHit) {
        anyHit = stopAtFirstHit;
        world = ray;
        cur = ray;
        shear = Shear{};
        if (hasFolded) {
            toTopLevel();
            shear = makeShear(cur);
            topSx = shear.Sx;
            topSy = shear.Sy;
            topSz = shear.Sz;
            topAxes = shear.kx | shear.ky << 2 | shear.kz << 4;
        }
        idx = __builtin_amdgcn_rcpf(cur.dx);
        idy = __builtin_amdgcn_rcpf(cur.dy);
        idz = __builtin_amdgcn_rcpf(cur.dz);
        nodes = topNodes;
        triangles = nullptr;
        nodeLimit = topNodeLimit;
        itemLimit = instanceLimit;
        instanceId = kNoHit;
        floor = -1;
        found = false;
        // a ray starts at its origin at the earliest: a tmin below zero, or one that is no number, counts as zero
        // (fmaxf drops the NaN).  Distances behind the origin then never decide a box test, whose margins
        // are factors and would lean the wrong way there.
        tmin = fmaxf(rayTmin, 0.0f);
        tminShrunk = tmin * kShrink;
        tmax = rayTmax;
        bestT = rayTmax;
        bestU = 0.0f;
        bestV = 0.0f;
        bestPrim = initPrimitive;
        bestInst = initInstance;
        sp = 0;
        node = 0;  // the top-level root
    }

    __device__ __forceinline__ TraceResult result() const {
        TraceResult r;
        r.t = found ? bestT : tmax;
        r.u = found ? bestU : 0.0f;
        r.v = found ? bestV : 0.0f;
        r.primitive = found ? bestPrim : kNoHit;
        r.instance = found ? bestInst : kNoHit;
        return r;
    }

    __device__ __forceinline__ void push(uint32_t ref, float t) {
        if (sp < kLdsStack) {
            stackRef[sp * stride] = ref;
            stackT[sp * stride] = t;
        } else if (sp < kMaxStack) {
            privateRef[sp - kLdsStack] = ref;
            privateT[sp - kLdsStack] = t;
        }
        ++sp;
    }

    // Takes one entry off the stack: the reference to go to, kPop when the entry
    // needs no visit (it fell behind the closest hit) or kDone when nothing is left.
    // An entry from below the height at which the ray entered its instance is a
    // top-level one: the ray goes back to world space with it.
    // One entry per call: a loop here would hold the wave's other lanes up.
    __device__ __forceinline__ uint32_t pop() {
        RT_SITE(2);
        if (sp == 0)
            return kDone;
        --sp;
        if (sp < floor) {
            cur = world;
            if (hasFolded) {
                toTopLevel();
                shear.Sx = topSx;
                shear.Sy = topSy;
                shear.Sz = topSz;
                shear.kx = topAxes & 3;
                shear.ky = (topAxes >> 2) & 3;
                shear.kz = topAxes >> 4;
            }
            idx = __builtin_amdgcn_rcpf(cur.dx);
            idy = __builtin_amdgcn_rcpf(cur.dy);
            idz = __builtin_amdgcn_rcpf(cur.dz);
            nodes = topNodes;
            nodeLimit = topNodeLimit;
            itemLimit = instanceLimit;
            floor = -1;
        }
        const uint32_t otherwise = sp == 0 ? kDone : kPop;
        if (sp >= kMaxStack)
            return otherwise;  // never pushed (guarded against on the host as well)
        uint32_t ref;
        float t;
        if (sp < kLdsStack) {
            ref = stackRef[sp * stride];
            t = stackT[sp * stride];
        } else {
            ref = privateRef[sp - kLdsStack];
            t = privateT[sp - kLdsStack];
        }
        return t > bestT * kGrow ? otherwise : ref;
    }

    // One wide node: intersect the eight child boxes, continue with the nearest
    // and leave the others on the stack.
    __device__ __forceinline__ void nodeStep() {
        RT_SITE(0);
        RT_GUARD(node < nodeLimit);
        if (COUNT)
            ++counters.nodes;
        GlobalUint4 np = globalView(nodes + node);
        const uint4 n0 = np[0];
        uint4 n1 = np[1], n2 = np[2], n3 = np[3], n4 = np[4];
        uint32_t imask = n0.w >> 24;
        // A ray parallel to an axis (a direction component of exactly zero) is unconstrained on that axis
        // below: its plane distances are 0 * inf or inf - inf, the NaN that fmaxf / fminf drop.  That is
        // safe but lets the ray into every child along the axis - a hundred times the visits of its
        // neighbours, alone in its wave.  Such a ray can only enter a child whose slab holds its origin,
        // which is tested on the planes themselves, with a margin: corner - origin and the multiply-add are two
        // roundings, so a child is only dropped when the origin is outside its slab by more than 2^-20 of the
        // magnitudes involved (the margin the general path gives its boxes); an origin on a plane, or within
        // rounding of it, keeps the child.
        if ((cur.dx * cur.dy) * cur.dz == 0.0f) {
            auto outsideSlab = [&](float direction, float origin, uint32_t corner, uint32_t exponent, uint32_t lo0,
                                   uint32_t lo1, uint32_t hi0, uint32_t hi1) -> uint32_t {
                if (direction != 0.0f)
                    return 0U;
                const float scale = __uint_as_float((exponent & 0xffU) << 23);
                const float offset = __uint_as_float(corner) - origin;
                const float margin = 0x1p-20f * ((fabsf(__uint_as_float(corner)) + fabsf(origin)) + 255.0f * scale);
                uint32_t outside = 0U;
#pragma unroll
                for (int i = 0; i < kWide; ++i) {
                    const int sh = 8 * (i & 3);
                    const float qlo = static_cast<float>(((i < 4 ? lo0 : lo1) >> sh) & 0xffU);
                    const float qhi = static_cast<float>(((i < 4 ? hi0 : hi1) >> sh) & 0xffU);
                    if (fmaf(qlo, scale, offset) > margin || fmaf(qhi, scale, offset) < -margin)
                        outside |= 1U << i;
                }
                return outside;
            };
            const uint32_t outside = outsideSlab(cur.dx, cur.ox, n0.x, n0.w, n2.x, n2.y, n3.z, n3.w) |
                                     outsideSlab(cur.dy, cur.oy, n0.y, n0.w >> 8, n2.z, n2.w, n4.x, n4.y) |
                                     outsideSlab(cur.dz, cur.oz, n0.z, n0.w >> 16, n3.x, n3.y, n4.z, n4.w);
            // a child that is dropped is struck from the node: no inner bit, no triangle count, and the
            // inverted box of an empty slot on every axis
            auto bytesOf = [](uint32_t four) -> uint32_t {
                return ((four & 1U) ? 0xffU : 0U) | ((four & 2U) ? 0xff00U : 0U) | ((four & 4U) ? 0xff0000U : 0U) |
                       ((four & 8U) ? 0xff000000U : 0U);
            };
            const uint32_t b0 = bytesOf(outside & 15U), b1 = bytesOf(outside >> 4);
            imask &= ~outside;
            n1.z &= ~b0;
            n1.w &= ~b1;
            n2.x |= b0; n2.y |= b1; n2.z |= b0; n2.w |= b1; n3.x |= b0; n3.y |= b1;
            n3.z &= ~b0; n3.w &= ~b1; n4.x &= ~b0; n4.y &= ~b1; n4.z &= ~b0; n4.w &= ~b1;
        }
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
        // per child: the entry distance with the slot number in its low three
        // bits (all ones when the ray misses the child), so that sorting the
        // words sorts the children by distance
        uint32_t key[kWide];
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
#if RT_TRACE_PACKED_FMA
            // near and far plane of an axis in one packed multiply-add: the same two roundings as two fmaf
            const Pair tx = __builtin_elementwise_fma(Pair{qxn, qxf}, Pair{axn, axf}, Pair{bxn, bxf});
            const Pair ty = __builtin_elementwise_fma(Pair{qyn, qyf}, Pair{ayn, ayf}, Pair{byn, byf});
            const Pair tz = __builtin_elementwise_fma(Pair{qzn, qzf}, Pair{azn, azf}, Pair{bzn, bzf});
            const float tn = fmaxf(fmaxf(tx.x, ty.x), fmaxf(tz.x, tminShrunk));
            const float tf = fminf(fminf(tx.y, ty.y), fminf(tz.y, limit));
#else
            const float tn = fmaxf(fmaxf(fmaf(qxn, axn, bxn), fmaf(qyn, ayn, byn)), fmaxf(fmaf(qzn, azn, bzn), tminShrunk));
            const float tf = fminf(fminf(fmaf(qxf, axf, bxf), fmaf(qyf, ayf, byf)), fminf(fmaf(qzf, azf, bzf), limit));
#endif
            // an empty slot needs no test of its own: its box is inverted (see WideNode)
            const bool hit = tn <= tf;
            key[i] = hit ? ((__float_as_uint(tn) & ~7U) | static_cast<uint32_t>(i)) : 0xffffffffU;
            hitMask |= hit ? (1U << i) : 0U;
        }
        // reference of the child in `slot`; its meta byte is byte `slot` of n1.z : n1.w
        auto refOf = [&](uint32_t slot) -> uint32_t {
            const uint32_t meta = __builtin_amdgcn_perm(n1.w, n1.z, 0x0c0c0c00U | slot);
            const uint32_t inner = n1.x + meta;
            const uint32_t leaf = kLeafBit | (meta >> 5) << 28 | (n1.y + (meta & 31U));
            return ((imask >> slot) & 1U) ? inner : leaf;
        };
        if (hitMask == 0U) {
            node = kPop;
            return;
        }
        if ((hitMask & (hitMask - 1U)) == 0U) {
            node = refOf(static_cast<uint32_t>(__ffs(static_cast<int>(hitMask)) - 1));
            return;
        }
        RT_SITE(1);
        // several children: sort them (19 compare-exchanges), enter the nearest
        // and leave the others on the stack, farthest first
#define RT_EXCHANGE(a, b)                                                      \
    do {                                                                       \
        const uint32_t lo = min(key[a], key[b]), hi = max(key[a], key[b]);     \
        key[a] = lo;                                                           \
        key[b] = hi;                                                           \
    } while (0)
        RT_EXCHANGE(0, 1); RT_EXCHANGE(2, 3); RT_EXCHANGE(4, 5); RT_EXCHANGE(6, 7);
        RT_EXCHANGE(0, 2); RT_EXCHANGE(1, 3); RT_EXCHANGE(4, 6); RT_EXCHANGE(5, 7);
        RT_EXCHANGE(1, 2); RT_EXCHANGE(5, 6); RT_EXCHANGE(0, 4); RT_EXCHANGE(3, 7);
        RT_EXCHANGE(1, 5); RT_EXCHANGE(2, 6);
        RT_EXCHANGE(1, 4); RT_EXCHANGE(3, 6);
        RT_EXCHANGE(2, 4); RT_EXCHANGE(3, 5);
        RT_EXCHANGE(3, 4);
#undef RT_EXCHANGE
#pragma unroll
        for (int j = kWide - 1; j >= 1; --j) {
            if (key[j] != 0xffffffffU) {
                // the distance lost its low bits, which only lowers it; a negative
                // one (ray interval starting behind the origin) would be raised, so
                // it is replaced by one that never drops the entry
                const float t = static_cast<int>(key[j]) < 0 ? -INFINITY : __uint_as_float(key[j] & ~7U);
                push(refOf(key[j] & 7U), t);
            }
        }
        node = refOf(key[0] & 7U);
    }


    // Top-level leaf: move the ray into the instance's object space.
    __device__ __forceinline__ void enterInstance() {
        RT_SITE(4);
        RT_GUARD((node & 0x0fffffffU) < itemLimit);
        if (COUNT)
            ++counters.instances;
        const InstanceRecord* record = instances + (node & 0x0fffffffU);
        GlobalUint4 m = globalView(record);
        const uint4 q0 = m[0], q1 = m[1], q2 = m[2], q3 = m[3], q4 = m[4];
        const float4 r0 = make_float4(__uint_as_float(q0.x), __uint_as_float(q0.y), __uint_as_float(q0.z), __uint_as_float(q0.w));
        const float4 r1 = make_float4(__uint_as_float(q1.x), __uint_as_float(q1.y), __uint_as_float(q1.z), __uint_as_float(q1.w));
        const float4 r2 = make_float4(__uint_as_float(q2.x), __uint_as_float(q2.y), __uint_as_float(q2.z), __uint_as_float(q2.w));
        floor = sp;
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
        triangles = reinterpret_cast<const PackedTriangle*>(static_cast<uintptr_t>(q3.z) | static_cast<uintptr_t>(q3.w) << 32);
        instanceId = q4.x;
        nodeLimit = q4.z;
        itemLimit = q4.w;
        node = 0;
    }

    // Bottom-level leaf: tests its next triangle.  The leaf reference counts down;
    // after the last triangle the lane moves on to the next stack entry.
    __device__ __forceinline__ void triangleStep() {
        RT_SITE(3);
        if (COUNT)
            ++counters.triangles;
        static_assert(kMaxLeaf <= 3, "the third bit of a leaf's count field marks a folded triangle");
        const uint32_t count = (node >> 28) & 3U;
        const uint32_t first = node & 0x0fffffffU;
        RT_GUARD(first < itemLimit);
        // in the top level the item is a folded triangle: a record of the instance array, its instance's id in it
        const bool folded = !inBottom();
        GlobalUint4 tp = folded ? globalView(instances + first) : globalView(triangles + first);
        const uint4 p0 = tp[0], p1 = tp[1], p2 = tp[2];
        float t, u, v;
        if (intersectTriangle(cur, shear, p0, p1, p2, tmin, bestT, t, u, v)) {
            const uint32_t prim = p0.w, instance = folded ? p1.w : instanceId;
            // intersectTriangle only returns t <= bestT
            if (t < bestT || instance < bestInst || (instance == bestInst && prim < bestPrim)) {
                bestT = t;
                bestU = u;
                bestV = v;
                bestPrim = prim;
                bestInst = instance;
                found = true;
            }
        }
        if (anyHit && found)
            node = kDone;
        else if (count > 1U)
            node = (node & (kLeafBit | kFoldedBit)) | (count - 1U) << 28 | (first + 1U);
        else
            node = kPop;
    }
};

// Hands out ray indices [0, n) to the waves of a persistent launch.  The range
// is cut into contiguous chunks, kFeedSplit per XCD (workgroups are dealt to the
// XCDs round-robin, so blockIdx & 7 names the XCD): a wave drains a chunk of
// its own XCD first, which keeps neighbouring rays behind one L2, and then
// helps with the others.  `cursors` holds one counter per chunk, kFeedStride
// words apart, zeroed before the launch.  Four chunks per XCD rather than one: every draw is an atomic that
// returns a value, those on one word are served one after the other (about 11 ns apiece), and a launch of
// sixteen samples makes some 900,000 draws (1080p step 1.78 -> 1.74 ms).
struct RayFeed {
    uint32_t* cursors;
    uint32_t n, chunkSize;
    uint32_t chunk;      // chunk this wave draws from (wave-uniform)
    uint32_t exhausted;  // chunks found empty so far (wave-uniform)

    __device__ __forceinline__ RayFeed(uint32_t* c, uint32_t count) : cursors(c), n(count) {
        constexpr uint32_t chunks = kFeedChunks;
        chunkSize = ((count + chunks - 1U) / chunks + 63U) & ~63U;
        // the chunks of an XCD are neighbours in the index range; its workgroups start on them in turn
        chunk = (blockIdx.x & 7U) * kFeedSplit + (blockIdx.x >> 3) % kFeedSplit;
        exhausted = 0;  // chunk and exhausted go through readfirstlane wherever they change: scalar registers
    }
    __device__ __forceinline__ bool drained() const { return exhausted >= static_cast<uint32_t>(kFeedChunks); }

    // True for a wave the launch has no use for: waves of an XCD beyond those whose
    // first draw (64 rays each) empties the XCD's chunk.  Such a wave leaves at once.
    // A launch with few rays (a late bounce) is then over when the waves that got
    // rays are done, instead of after every wave of the GPU has found out, one
    // atomic per chunk, that there is nothing left: atomics on one word queue up
    // behind each other, which put a floor of about 130 us under every launch.
    __device__ __forceinline__ bool surplus() const {
        const uint32_t rank = ((blockIdx.x >> 3) / kFeedSplit) * (blockDim.x >> 6) + (threadIdx.x >> 6);
        const uint32_t begin = min(chunk * chunkSize, n), end = min(begin + chunkSize, n);
        return rank >= (end - begin + 63U) / 64U + 2U;
    }

    // take() in two halves, so that the answer of the atomic can travel while the wave does other
    // work.  ask(): one atomic for `want` indices of the current chunk; what it returns goes to
    // answer(), which hands the indices to the lanes of `idleMask` or, when the chunk had nothing
    // left, moves on to the next chunk and returns false (the caller asks again next time).
    __device__ __forceinline__ uint32_t ask(uint32_t want) {
        uint32_t base = 0;
        if ((threadIdx.x & 63) == 0)
            base = atomicAdd(&cursors[chunk * kFeedStride], want);
        return base;
    }
    __device__ __forceinline__ bool answer(uint32_t ticket, unsigned long long idleMask, bool idle, uint32_t& index) {
        const int lane = threadIdx.x & 63;
        const uint32_t begin = min(chunk * chunkSize, n), end = min(begin + chunkSize, n);
        const uint32_t base = __builtin_amdgcn_readfirstlane(ticket);
        if (base < end - begin) {
            const uint32_t mine = begin + base + __popcll(idleMask & ((1ULL << lane) - 1ULL));
            index = mine;
            return idle && mine < end;
        }
        exhausted = __builtin_amdgcn_readfirstlane(exhausted + 1U);
        chunk = __builtin_amdgcn_readfirstlane((chunk + 1U) % kFeedChunks);
        return false;
    }

    // Every lane of the wave calls this with the same `idleMask` (lanes that want
    // a ray).  Returns true with `index` set for the lanes that got one.
    __device__ __forceinline__ bool take(unsigned long long idleMask, bool idle, uint32_t& index) {
        const int lane = threadIdx.x & 63;
        while (exhausted < static_cast<uint32_t>(kFeedChunks)) {
            const uint32_t begin = min(chunk * chunkSize, n), end = min(begin + chunkSize, n);
            const uint32_t want = __popcll(idleMask);
            uint32_t base = 0;
            if (lane == 0)
                base = atomicAdd(&cursors[chunk * kFeedStride], want);
            base = __builtin_amdgcn_readfirstlane(base);
            if (base < end - begin) {
                const uint32_t mine = begin + base + __popcll(idleMask & ((1ULL << lane) - 1ULL));
                index = mine;
                return idle && mine < end;
            }
            exhausted = __builtin_amdgcn_readfirstlane(exhausted + 1U);
            chunk = __builtin_amdgcn_readfirstlane((chunk + 1U) % kFeedChunks);
        }
        return false;
    }
};

// A ray as it comes out of memory, before a lane's walk takes it over.
struct RayIn {
    float4 origin, direction;  // (origin, tmin), (direction, tmax)
    uint2 target;              // candidate (instance, primitive) a shadow ray starts with
};

// Persistent traversal loop of one lane.  load(index) reads ray `index`;
// start(index, ray, traversal) calls traversal.begin(...); store(index, result)
// consumes a finished ray.
//
// Per turn every lane with a ray does at most one wide node, one stack entry
// and one leaf (up to RT_TRACE_LEAF_STEPS triangles of it), so the lanes of a
// wave run the node code together and the leaf code together.
template <bool COUNT, typename Load, typename Start, typename Store>
__device__ __forceinline__ void tracePersistent(Traversal<COUNT>& walk, RayFeed& feed, Load&& load,
                                                Start&& start, Store&& store) {
    bool active = false;
    uint32_t rayIndex = 0;
#ifdef RT_TRACE_PROFILE
    bool sawDrained = false;
    unsigned long long drainTurns = 0, drainLanes = 0;
    if ((threadIdx.x & 63) == 0)
        atomicMin(&gTraceClock[0], wall_clock64());
#endif
    if (feed.surplus())
        return;
    for (;;) {
#ifdef RT_TRACE_PROFILE
        if (feed.drained()) {
            if (!sawDrained && (threadIdx.x & 63) == 0)
                atomicMin(&gTraceClock[1], wall_clock64());
            sawDrained = true;
            ++drainTurns;
            drainLanes += __popcll(__ballot(active));
        }
#endif
        const unsigned long long idleMask = __ballot(!active);
#if RT_TRACE_RIDING_REFILL
        // A refill rides along the turn: the atomic is sent here and read behind the node step,
        // the rays are loaded there and taken over at the end of the turn, so the two round trips
        // to memory overlap with the ones the walk makes anyway.
        const bool refilling = !feed.drained() && __popcll(idleMask) >= RT_TRACE_REFILL;
        uint32_t ticket = 0;
        if (refilling)
            ticket = feed.ask(__popcll(idleMask));
        if (__ballot(active) == 0ULL && !refilling) {
            if (feed.drained())
                break;
            continue;
        }
#else
        if (!feed.drained() && __popcll(idleMask) >= RT_TRACE_REFILL) {
            uint32_t index = 0;
            if (feed.take(idleMask, !active, index)) {
                start(index, load(index), walk);
                active = true;
                rayIndex = index;
            }
        }
        if (__ballot(active) == 0ULL) {
            if (feed.drained())
                break;
            continue;
        }
#endif
        if (active) {
#ifdef RT_TRACE_PROFILE
            float* profile = walk.profile;
            uint32_t raySteps = 0;
            RT_SITE(5);
#endif
            // A turn is flat: one wide node for the lanes at one, then one stack entry
            // for the lanes that ran out of work (kPop), then the leaf code.  A loop
            // at any of these places makes the other lanes of the wave wait for its
            // longest trip, and the values the compiler keeps across it cost registers
            // (three nodes per turn spilt 26-48 registers; this form spills none).
            if (!(walk.node & kLeafBit))
                walk.nodeStep();
#if RT_TRACE_RIDING_REFILL
        }
        bool taking = false;
        uint32_t incoming = 0;
        RayIn next{};
        if (refilling) {
            taking = feed.answer(ticket, idleMask, !active, incoming);
            if (taking)
                next = load(incoming);
        }
        if (active) {
#endif
#if RT_TRACE_POP_AFTER_NODE
            if (walk.node == kPop)
                walk.node = walk.pop();
#endif
            if ((walk.node & 0xf0000000U) == kLeafBit)
                walk.node = kPop;  // a leaf of no items: an empty slot that passed the slab test
            if ((walk.node & kLeafBit) && walk.node < kPop) {
                if (!walk.inBottom() && !(walk.node & kFoldedBit)) {
                    walk.enterInstance();
                }
                else {
                    const uint32_t count = min((walk.node >> 28) & 3U, static_cast<uint32_t>(RT_TRACE_LEAF_STEPS));
                    for (uint32_t k = 0; k < count && walk.node < kPop; ++k)
                        walk.triangleStep();
                }
            }
#if RT_TRACE_POP_AFTER_LEAF
            // a lane whose leaf is done takes its next stack entry at once, so that it starts the
            // next turn at a node instead of spending that turn's node step idle
            if (walk.node == kPop)
                walk.node = walk.pop();
#endif
        }
        if (active && walk.node == kDone) {
            store(rayIndex, walk.result());
            active = false;
#ifdef RT_TRACE_PROFILE
            walk.finishRay();
#endif
        }
#if RT_TRACE_RIDING_REFILL
        if (taking) {
            start(incoming, next, walk);
            active = true;
            rayIndex = incoming;
        }
#endif
    }
#ifdef RT_TRACE_PROFILE
    walk.flushProfile();
    if ((threadIdx.x & 63) == 0) {
        atomicMax(&gTraceClock[2], wall_clock64());
        atomicAdd(&gTraceClock[3], drainTurns);
        atomicAdd(&gTraceClock[4], drainLanes);
    }
#endif
}

// The item count of a scene as the ray kernels take it: the top bit says that folded triangles are among the items.
inline uint32_t itemCountWord(const rt_scene* scene) {
    return scene->itemCount | (scene->foldedTriangles ? 0x80000000U : 0U);
}

// Workgroups of a persistent traversal launch that fill `device` once: one per
// CU and wave slot the kernels' launch bounds leave room for.
inline uint32_t persistentBlocks(int device) {
    static std::atomic<uint32_t> cached[64];
    const int slot = device >= 0 && device < 64 ? device : 0;
    uint32_t blocks = cached[slot].load(std::memory_order_relaxed);
    if (blocks == 0U) {
        int units = 0;
        if (hipDeviceGetAttribute(&units, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || units <= 0)
            units = 256;
        blocks = static_cast<uint32_t>(units) * RT_TRACE_MIN_WAVES * 4U / (kTraceBlock / 64) * RT_TRACE_GRID_SCALE;
        cached[slot].store(blocks, std::memory_order_relaxed);
    }
    return blocks;
}

}  // namespace rt
