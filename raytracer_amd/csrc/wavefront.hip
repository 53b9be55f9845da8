// Wavefront path tracing and photon tracing for gfx950.
//
// Replaces the two OptiX ray-generation programs of the reference - the camera
// path integrator __raygen__render (src/device.cu:863-1075) and the photon
// tracer __raygen__photons (src/device.cu:768-861) - by queues of rays that
// move through separate kernels:
//
//   generate -> [ extend (closest hit) -> shade -> shadow ] x maxDepth -> accumulate
//
// A path keeps its state in arrays indexed by its pixel, so only the ray itself
// is moved when the queue is compacted.  Survivors get their slot in the next
// queue from a wave ballot, a workgroup prefix and one atomic per workgroup.
// Queue lengths stay on the device: every launch covers the whole capacity and
// surplus workgroups leave at once, so a sample runs without host round trips.
// Draw order of the random numbers and the order of the float sums per pixel
// are those of the reference's megakernel.
#include "photon_device.h"
#include "radix_sort.h"
#include "shading.h"
#include "traverse_core.h"

#include <cstring>
#include <type_traits>
#include <vector>

namespace rt {
namespace {

// Threads per workgroup of the stages between the ray launches.  A workgroup draws its slots in the next
// queue with one atomic on the queue's counter, and atomics on one word are served one after the other:
// the fewer workgroups a launch has, the shorter that line.
#ifndef RT_SHADE_BLOCK
#define RT_SHADE_BLOCK 256
#endif
constexpr int kShadeBlock = RT_SHADE_BLOCK;
constexpr int kShadeWaves = kShadeBlock / 64;
constexpr int kSlotWords = kShadeWaves + 1;  // shared words allocateSlots() needs; allocateSlotPair() twice as many

// Path flags packed beside lastOrigin: bits 0-1 lobe, 2 lastDelta,
// 3 primaryChain, 4 guidePending, bits 8.. (medium material + 1).
constexpr uint32_t kLastDelta = 4U, kPrimaryChain = 8U, kGuidePending = 16U;

struct Buffers {
    uint32_t capacity;
    float4* rayOrigin[2];
    float4* rayDirection[2];
    uint32_t* rayPath[2];   // path (pixel / emission) index of the ray in each slot
    uint4* hits;
    uint32_t* hitInstance;
    float4* shadowOrigin;
    float4* shadowDirection;
    float4* shadowContribution;  // xyz + bits: lobe (0-1), needs the light's own triangle (2)
    uint32_t* shadowPath;
    uint2* shadowTarget;         // instance, primitive of the sampled emissive triangle
    // the state a path carries from bounce to bounce travels with its ray, in the queue's slot order
    uint32_t* rng[2];
    float4* throughput[2];       // xyz + lastPdf (camera) / power xyz (photons)
    float4* lastOrigin[2];       // xyz + flags
    float4* sample[6];           // of this sample: [0] radiance (of the path's one lobe), [3] caustic, [4] albedo guide,
                                 // [5] normal guide with the lobe (0-2) as the bits of .w; [1], [2] unused
    float4* gather[4];           // photon queries a shade launch leaves for the gather launch behind it:
                                 // point + path, normal, throughput, surface colour x kernel weight
    uint32_t* heavyQueries;      // photon queries that read more photons than one lane should walk
    uint32_t heavyBudget;        // ... this many (kHeavyPhotons, or no limit when the hand-over is switched off)
    uint32_t* counts;            // [0], [1] queue lengths, [2] shadow queue length, [3] photon queries,
                                 // [4] heavy photon queries
    uint32_t* feed;              // ray-feed cursors of the trace launch (kFeedWords), zero between launches
    unsigned long long* stats;   // [0] extend rays, [1] shadow rays, [2] gathers, [3] photons visited by gathers,
                                 // [4] nodes, [5] triangles, [6] instances visited by counted extend launches
    const uint32_t* bucketBits;  // one bit per bucket of the photon grid, or null
    LightIndex lightIndex;       // thresholds and triangle table of the light list, or empty
    rt_photon* photonTemp;       // photon pass: one slot per emission of the batch
    uint32_t* photonFlags;
};

// The photon maps a pass gathers from.  The reference rebuilds its one map every fourth sample
// (renderer.cpp:177, 780-781); a pass over more samples than that needs the map of each group of samples it
// holds, so the gather kernels pick a query's map by the sample its path belongs to: sample k of the pass
// (path / perSample) reads map (phase + k) / samplesPerMap.  count == 0: the one map of the launch parameters.
struct PhotonMaps {
    rt_photon_map maps[RT_MAX_PHOTON_MAPS];
    uint32_t count, samplesPerMap, phase, perSample;

    __device__ void pick(const rt_launch_params& p, const uint32_t* bits, uint32_t path, const rt_photon*& sorted,
                         const uint32_t*& cellStart, const uint32_t*& bucketBits) const {
        if (count == 0U) {
            sorted = p.sortedPhotons;
            cellStart = p.photonCellStart;
            bucketBits = bits;
            return;
        }
        const uint32_t map = min((phase + path / perSample) / samplesPerMap, count - 1U);
        sorted = maps[map].sortedPhotons;
        cellStart = maps[map].cellStart;
        bucketBits = maps[map].bucketBits;
    }
};

// Slots for the lanes with `want` set: a wave ballot, a workgroup prefix and one
// atomic on *counter per workgroup.  Every thread of the workgroup calls this.
__device__ inline uint32_t allocateSlots(bool want, uint32_t* counter, uint32_t* shared /* kSlotWords */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long votes = __ballot(want);
    if (lane == 0)
        shared[wave] = __popcll(votes);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0;
        for (int w = 0; w < kShadeWaves; ++w) {
            const uint32_t c = shared[w];
            shared[w] = total;
            total += c;
        }
        shared[kShadeWaves] = total ? atomicAdd(counter, total) : 0U;
    }
    __syncthreads();
    const uint32_t slot = shared[kShadeWaves] + shared[wave] + __popcll(votes & ((1ULL << lane) - 1ULL));
    __syncthreads();
    return slot;
}

// Two queues at once (the next bounce's rays and the shadow rays of a shade
// launch): one pass through the workgroup's barriers instead of two.  `shared`
// holds 2 * kSlotWords words.
__device__ inline void allocateSlotPair(bool wantA, uint32_t* counterA, bool wantB, uint32_t* counterB,
                                        uint32_t* shared /* 2 * kSlotWords */, uint32_t& slotA, uint32_t& slotB) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long votesA = __ballot(wantA), votesB = __ballot(wantB);
    if (lane == 0) {
        shared[wave] = __popcll(votesA);
        shared[kSlotWords + wave] = __popcll(votesB);
    }
    __syncthreads();
    if (threadIdx.x < 2) {  // thread 0 totals queue A, thread 1 queue B
        uint32_t* counts = shared + kSlotWords * threadIdx.x;
        uint32_t total = 0;
        for (int w = 0; w < kShadeWaves; ++w) {
            const uint32_t c = counts[w];
            counts[w] = total;
            total += c;
        }
        counts[kShadeWaves] = total ? atomicAdd(threadIdx.x == 0 ? counterA : counterB, total) : 0U;
    }
    __syncthreads();
    const unsigned long long below = (1ULL << lane) - 1ULL;
    slotA = shared[kShadeWaves] + shared[wave] + __popcll(votesA & below);
    slotB = shared[kSlotWords + kShadeWaves] + shared[kSlotWords + wave] + __popcll(votesB & below);
}

// Three queues at once (a shade launch with a photon map: next rays, shadow rays, photon queries).  `shared` holds
// 3 * kSlotWords words.  The queries used to draw their slots where they arise, one atomic per wave on the queue's
// length: half a million of them in the first bounce of a sixteen-sample pass, served one after the other at some
// 11 ns apiece - 6 of the 12 ms of that launch.
__device__ inline void allocateSlotTriple(bool wantA, uint32_t* counterA, bool wantB, uint32_t* counterB, bool wantC,
                                          uint32_t* counterC, uint32_t* shared /* 3 * kSlotWords */, uint32_t& slotA,
                                          uint32_t& slotB, uint32_t& slotC) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long votesA = __ballot(wantA), votesB = __ballot(wantB), votesC = __ballot(wantC);
    if (lane == 0) {
        shared[wave] = __popcll(votesA);
        shared[kSlotWords + wave] = __popcll(votesB);
        shared[2 * kSlotWords + wave] = __popcll(votesC);
    }
    __syncthreads();
    if (threadIdx.x < 3) {  // one thread totals each queue
        uint32_t* counts = shared + kSlotWords * threadIdx.x;
        uint32_t total = 0;
        for (int w = 0; w < kShadeWaves; ++w) {
            const uint32_t c = counts[w];
            counts[w] = total;
            total += c;
        }
        uint32_t* counter = threadIdx.x == 0 ? counterA : (threadIdx.x == 1 ? counterB : counterC);
        counts[kShadeWaves] = total ? atomicAdd(counter, total) : 0U;
    }
    __syncthreads();
    const unsigned long long below = (1ULL << lane) - 1ULL;
    slotA = shared[kShadeWaves] + shared[wave] + __popcll(votesA & below);
    slotB = shared[kSlotWords + kShadeWaves] + shared[kSlotWords + wave] + __popcll(votesB & below);
    slotC = shared[2 * kSlotWords + kShadeWaves] + shared[2 * kSlotWords + wave] + __popcll(votesC & below);
}

// ---- camera paths -----------------------------------------------------------

// Which pixels a launch covers and how its paths are numbered.  The pixels are a run of
// bands of eight rows (every bandStride-th band of the image, so that several GPUs can
// take the bands of one frame in turn) followed by a row-major tail: the rows of a last,
// shorter band, or a whole range that does not start on a row.  Inside a full band the
// paths are numbered tile by tile (8 x 8 pixels), so the 64 lanes of a wave hold a compact
// patch of the image: their rays walk the same nodes and their photon queries the same
// cells.  Only the numbering of paths changes; a pixel's random streams depend on its own
// index.
struct PixelMap {
    uint32_t begin;       // first pixel of the first band
    uint32_t count;       // paths of the launch: perSample for each of its samples, sample after sample
    uint32_t perSample;   // pixels of the launch
    uint32_t samples;     // consecutive samples of these pixels that the launch renders (at least 1)
    uint32_t seedStride;  // the seeding sample index grows by this from one of them to the next
    uint32_t width;       // image width
    uint32_t bandPaths;   // paths in the tile-numbered bands (a multiple of 8 * width), then the tail
    uint32_t bandStride;  // pixels from one band of the launch to the next (8 * width when contiguous)
    uint32_t tailBegin;   // first pixel of the row-major tail
    uint32_t tiled;       // 0 when the width is no multiple of 8: the bands then keep row-major order too

    // pixel of path i of one sample, i < perSample
    __device__ uint32_t pixel(uint32_t i) const {
        if (i >= bandPaths)
            return tailBegin + (i - bandPaths);
        const uint32_t band = 8U * width;
        const uint32_t k = i / band, inBand = i - k * band, tile = inBand / 64U, within = inBand % 64U;
        if (!tiled)
            return begin + k * bandStride + inBand;
        return begin + k * bandStride + (within / 8U) * width + tile * 8U + within % 8U;
    }
};

// device.cu:882-906: one path per pixel of [pixelBegin, pixelBegin + count)
__global__ __launch_bounds__(kShadeBlock) void generateKernel(rt_launch_params p, PixelMap map, uint32_t seedSample,
                                                              Buffers b) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t count = map.count;
    if (i == 0) {
        b.counts[0] = count;
        b.counts[1] = 0;
        b.counts[2] = 0;
        b.counts[3] = 0;
        b.counts[4] = 0;
    }
    if (i >= count)
        return;
    const uint32_t batch = i / map.perSample;
    const uint32_t index = map.pixel(i - batch * map.perSample);
    const uint32_t px = index % p.width, py = index / p.width;
    uint32_t rng = index * 9781U + (seedSample + batch * map.seedStride) * 6271U + 0x68bc21ebU;
    const float jx = random01(rng);
    const float jy = random01(rng);
    const float sx = (static_cast<float>(px) + jx) / static_cast<float>(p.width);
    const float sy = (static_cast<float>(py) + jy) / static_cast<float>(p.height);
    const V3 pinhole = normalize(v3(p.cameraW) + (2.0f * sx - 1.0f) * v3(p.cameraU) + (2.0f * sy - 1.0f) * v3(p.cameraV));
    const V3 focalPoint = v3(p.eye) + pinhole * (p.focusDistance / fmaxf(dot(pinhole, v3(p.cameraW)), 1e-4f));
    // lensSample, device.cu:372-377 (drawn even with a closed aperture)
    const float radius = sqrtf(random01(rng));
    const float angle = 6.2831853f * random01(rng);
    const V3 origin = v3(p.eye) + (v3(p.lensU) * (radius * mathCos(angle) * p.aperture) +
                                   v3(p.lensV) * (radius * mathSin(angle) * p.aperture));
    const V3 direction = normalize(focalPoint - origin);
    b.rayOrigin[0][i] = make_float4(origin.x, origin.y, origin.z, 0.001f);
    b.rayDirection[0][i] = make_float4(direction.x, direction.y, direction.z, 1e16f);
    b.rayPath[0][i] = i;
    b.rng[0][i] = rng;
    b.throughput[0][i] = make_float4(1.0f, 1.0f, 1.0f, 0.0f);
    b.lastOrigin[0][i] = make_float4(origin.x, origin.y, origin.z, __uint_as_float(kLastDelta | kPrimaryChain | kGuidePending));
    const float4 zero = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    // A path adds to one of the three lobe sums only (the lobe is fixed at its first hit), so one buffer
    // holds its radiance and the lobe rides in the normal guide; the caustic sum exists with a photon map.
    b.sample[0][i] = zero;
    if (p.photonRadius > 0.0f)
        b.sample[3][i] = zero;
    b.sample[4][i] = zero;
    b.sample[5][i] = zero;
}

// The ray stage of a bounce, one persistent launch for both kinds of ray:
//   * extend: closest hit for the counts[which] rays of queue `which` (skipped
//     when which < 0);
//   * shadow: visible(), device.cu:548-559, for the counts[2] rays queued by the
//     previous shade launch (skipped when withShadow is 0): the light is seen
//     when nothing is hit before it (delta lights, environment) or when the
//     closest hit is the sampled emissive triangle itself; then the deferred
//     contribution is added.
// The two kinds are independent, and tracing them together leaves one tail of
// slow rays per bounce instead of two.
//
// "The closest hit is that triangle" is decided without a closest-hit walk: the
// shade launch has intersected the shadow ray with the sampled triangle (same
// arithmetic as the traversal, so the same t; lightTriangleDistance below) and
// queued (t, instance, primitive) with the ray; the scene is then searched for
// ANY hit that beats that candidate - the first one found ends the walk.  The
// outcome is the same bit for bit.
//
// COUNT adds the visited nodes / tested triangles / entered instances to stats[4..6].
template <bool COUNT>
__global__ __launch_bounds__(kTraceBlock, RT_TRACE_MIN_WAVES) void traceKernel(const WideNode* topNodes, const InstanceRecord* instances,
                                                           Buffers b, int which, int withShadow, uint2 limits) {
    __shared__ uint32_t stackRef[kLdsStack][kTraceBlock];
    __shared__ float stackT[kLdsStack][kTraceBlock];
    uint32_t spillRef[kPrivateStack];
    float spillT[kPrivateStack];
    Traversal<COUNT> walk(topNodes, limits.x, instances, limits.y, ldsView(&stackRef[0][threadIdx.x]),
                          ldsView(&stackT[0][threadIdx.x]), kTraceBlock, spillRef, spillT);
    const uint32_t extendCount = which >= 0 ? b.counts[which] : 0U;
    const uint32_t shadowCount = withShadow ? b.counts[2] : 0U;
    RayFeed feed(b.feed, extendCount + shadowCount);  // extend rays first: they are the longer ones
#ifdef RT_TRACE_PROFILE
    walk.profileBase = 16;
#endif
    const float4* origins = b.rayOrigin[which >= 0 ? which : 0];
    const float4* directions = b.rayDirection[which >= 0 ? which : 0];
    tracePersistent<COUNT>(
        walk, feed,
        [&](uint32_t index) {
            if (index < extendCount)
                return RayIn{origins[index], directions[index], make_uint2(kNoHit, kNoHit)};
            const uint32_t ray = index - extendCount;
            // kNoHit ids unless the ray is aimed at an emissive triangle
            return RayIn{b.shadowOrigin[ray], b.shadowDirection[ray], b.shadowTarget[ray]};
        },
        [&](uint32_t index, const RayIn& in, Traversal<COUNT>& w) {
            const float4 o4 = in.origin, d4 = in.direction;
            w.begin(Ray{o4.x, o4.y, o4.z, d4.x, d4.y, d4.z}, o4.w, d4.w, in.target.x, in.target.y, index >= extendCount);
        },
        [&](uint32_t index, const TraceResult& r) {
            if (index < extendCount) {
                b.hits[index] = make_uint4(__float_as_uint(r.t), __float_as_uint(r.u), __float_as_uint(r.v), r.primitive);
                b.hitInstance[index] = r.instance;
                return;
            }
            if (r.primitive != kNoHit)
                return;  // something lies in front of the light (or of the sampled triangle)
            const uint32_t ray = index - extendCount;
            const float4 c4 = b.shadowContribution[ray];
            const uint32_t path = b.shadowPath[ray];
            const float4 acc = b.sample[0][path];
            b.sample[0][path] = make_float4(acc.x + c4.x, acc.y + c4.y, acc.z + c4.z, 0.0f);
        });
    if (__ballot(walk.violations != 0U)) {
        if (walk.violations)
            atomicAdd(&b.stats[7], static_cast<unsigned long long>(walk.violations));
    }
    if (COUNT) {
        TraceCounters local = walk.counters;
#pragma unroll
        for (int delta = 32; delta >= 1; delta >>= 1) {
            local.nodes += __shfl_down(local.nodes, delta, 64);
            local.triangles += __shfl_down(local.triangles, delta, 64);
            local.instances += __shfl_down(local.instances, delta, 64);
        }
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&b.stats[4], local.nodes);
            atomicAdd(&b.stats[5], local.triangles);
            atomicAdd(&b.stats[6], local.instances);
        }
    }
}

// Where a shadow ray meets the emissive triangle it was aimed at, computed with
// the arithmetic of the traversal (object-space ray, object-space vertices), so
// that t equals what a closest-hit walk would report for that triangle bit for
// bit.  False when the ray does not reach the triangle inside (tmin, tmax].
__device__ inline bool lightTriangleDistance(const ShadeInstance* instances, const rt_hit_group_data* geometries,
                                             uint32_t instance, uint32_t primitive, V3 origin, V3 direction, float tmin,
                                             float tmax, float& distance) {
    const ShadeInstance& record = instances[instance];
    struct { const rt_gpu_vertex* vertices; const rt_uint3* indices; } geometry{record.vertices, record.indices};
    const rt_uint3 triangle = geometry.indices[primitive];
    const rt_float3 a = geometry.vertices[triangle.x].position;
    const rt_float3 bb = geometry.vertices[triangle.y].position;
    const rt_float3 c = geometry.vertices[triangle.z].position;
    const float* m = record.toObject;
    Ray local;
    local.ox = ((m[0] * origin.x + m[1] * origin.y) + m[2] * origin.z) + m[3];
    local.oy = ((m[4] * origin.x + m[5] * origin.y) + m[6] * origin.z) + m[7];
    local.oz = ((m[8] * origin.x + m[9] * origin.y) + m[10] * origin.z) + m[11];
    local.dx = (m[0] * direction.x + m[1] * direction.y) + m[2] * direction.z;
    local.dy = (m[4] * direction.x + m[5] * direction.y) + m[6] * direction.z;
    local.dz = (m[8] * direction.x + m[9] * direction.y) + m[10] * direction.z;
    const Shear shear = makeShear(local);
    float u, v;
    return intersectTriangle(local, shear, make_uint4(__float_as_uint(a.x), __float_as_uint(a.y), __float_as_uint(a.z), 0U),
                             make_uint4(__float_as_uint(bb.x), __float_as_uint(bb.y), __float_as_uint(bb.z), 0U),
                             make_uint4(__float_as_uint(c.x), __float_as_uint(c.y), __float_as_uint(c.z), 0U), tmin, tmax,
                             distance, u, v);
}

// What one bounce leaves behind besides the path state it updates in place.
struct ShadeOut {
    bool continues = false, wantsShadow = false;
    V3 origin{0, 0, 0}, direction{0, 0, 0};  // the next ray
    V3 shadowOrigin{0, 0, 0}, shadowDirection{0, 0, 0}, contribution{0, 0, 0};
    float shadowLimit = 0.0f;
    uint32_t shadowBits = 0, targetInstance = kNoHit, targetPrimitive = kNoHit;
    uint32_t gatherQueries = 0, gathered = 0;  // INLINE_GATHER only
    // the photon query of this bounce, for the launch's query queue
    bool wantsGather = false;
    V3 gatherPoint{0, 0, 0}, gatherNormal{0, 0, 0}, gatherThroughput{0, 0, 0}, gatherWeight{0, 0, 0};
    // the path's state behind the bounce (written next to its next ray when it continues)
    float4 throughputAfter{0, 0, 0, 0}, lastOriginAfter{0, 0, 0, 0};
    uint32_t rngAfter = 0;
};

// One bounce of device.cu:908-1041 for the path whose ray (origin, direction) met the scene at `h`.
// INLINE_GATHER: the caustic gather is done here instead of being queued.
template <bool INLINE_GATHER>
__device__ inline void shadeOne(const rt_launch_params& p, const ShadeInstance* instances,
                                const rt_hit_group_data* geometries, const Buffers& b, int which, uint32_t slot,
                                uint32_t path, V3 origin, V3 direction, uint4 h, uint32_t hitInstance, uint32_t depth,
                                ShadeOut& out) {
    bool& continues = out.continues;
    bool& wantsShadow = out.wantsShadow;
    V3& shadowOrigin = out.shadowOrigin;
    V3& shadowDirection = out.shadowDirection;
    V3& contribution = out.contribution;
    float& shadowLimit = out.shadowLimit;
    uint32_t& shadowBits = out.shadowBits;
    uint32_t& targetInstance = out.targetInstance;
    uint32_t& targetPrimitive = out.targetPrimitive;
    const float4 tp = b.throughput[which][slot];
    V3 throughput{tp.x, tp.y, tp.z};
    float lastPdf = tp.w;
    const float4 lo = b.lastOrigin[which][slot];
    const V3 lastOrigin{lo.x, lo.y, lo.z};
    uint32_t flags = __float_as_uint(lo.w);
    uint32_t lobe = flags & 3U;
    bool lastDelta = flags & kLastDelta, primaryChain = flags & kPrimaryChain, guidePending = flags & kGuidePending;
    int medium = static_cast<int>(flags >> 8) - 1;
    uint32_t rng = b.rng[which][slot];

    if (h.w == kNoHit) {
        // device.cu:910-915
        const float pdf = lastDelta ? 0.0f : environmentDensity(p, direction);
        const float weight = lastDelta ? 1.0f : powerHeuristic(lastPdf, pdf);
        const float4 r = b.sample[0][path];
        const V3 sum = rgb(r) + throughput * environmentRadiance(p, direction) * weight;
        b.sample[0][path] = make_float4(sum.x, sum.y, sum.z, 0.0f);
    } else {
        rt_hit hit;
        reconstructHit(instances, geometries, p.materials, origin, direction, __uint_as_float(h.x), __uint_as_float(h.y),
                       __uint_as_float(h.z), h.w, hitInstance, hit);
        const rt_gpu_material m = applyTextures(p.textures, p.materials[hit.material], hit);
        const V3 normal = shadingNormal(p.textures, hit, m);
        hit.normal = r3(normal);
        const V3 view = -direction;
        if (depth == 0U) {
            // device.cu:920-931
            lobe = m.transmission > 0.5f ? 2U : m.metallic > 0.5f ? 1U : 0U;
            b.sample[5][path] = make_float4(dot(normal, v3(p.lensU)), dot(normal, v3(p.lensV)),
                                            dot(normal, v3(p.cameraW)), __uint_as_float(lobe));
        }
        if (guidePending) {
            // device.cu:932-940
            const bool clearGlass = m.transmission > 0.95f && m.roughness < 0.1f;
            if (!clearGlass) {
                if (m.transmission < 0.05f)
                    b.sample[4][path] = make_float4(m.baseColor.x, m.baseColor.y, m.baseColor.z, 0.0f);
                guidePending = false;
            }
        }
        if (medium >= 0)
            throughput = throughput * mediumAbsorption(p.materials[medium], hit.distance);
        // device.cu:943-947
        // A surface that emits nothing adds a zero to the path's sum, whatever its weight: the sum is neither
        // read nor written for it and the light of its triangle is not looked up.  (A zero times a throughput or
        // a strength that is not finite would not be a zero, so those take the long way.)
        const bool emits = m.emissive.x != 0.0f || m.emissive.y != 0.0f || m.emissive.z != 0.0f ||
                           !isfinite((throughput.x + throughput.y + throughput.z) * m.emissiveStrength);
        V3 radiance{0.0f, 0.0f, 0.0f};
        if (emits) {
            const float lightPdf = lastDelta ? 0.0f : emissiveDensity(p, b.lightIndex, lastOrigin, hit);
            const float emissionWeight = lastDelta ? 1.0f : powerHeuristic(lastPdf, lightPdf);
            radiance = rgb(b.sample[0][path]);
            radiance = radiance + throughput * v3(m.emissive) * (m.emissiveStrength * emissionWeight);
        }

        // next event estimation, device.cu:580-605; the visibility query is
        // deferred to the shadow stage, which adds `contribution` when the
        // light is seen
        const float diffuseWeight = (1.0f - m.metallic) * (1.0f - m.transmission);
        if (diffuseWeight > 0.0f) {
            const LightDraw light = drawLight(p, b.lightIndex, v3(hit.position), rng);
            const float cosine = dot(normal, light.direction);
            if (light.valid && cosine > 0.0f) {
                const V3 reflectance = schlickFresnel(normalReflectance(m), fmaxf(dot(view, normal), 0.0f));
                const V3 diffuse{1.0f - reflectance.x, 1.0f - reflectance.y, 1.0f - reflectance.z};
                const V3 bsdf = rgb(m.baseColor) * diffuse * (diffuseWeight / 3.14159265f);
                const float bsdfPdf = (1.0f - m.transmission) * (1.0f - specularChoice(m, view, normal)) * cosine /
                                      3.14159265f;
                const float weight = light.delta ? 1.0f : powerHeuristic(light.pdf, bsdfPdf);
                contribution = throughput * (bsdf * light.radiance * (cosine * weight / light.pdf));
                wantsShadow = true;
                shadowOrigin = v3(hit.position) + normal * 0.001f;  // device.cu:550-552
                shadowDirection = light.direction;
                shadowLimit = light.delta ? light.distance - 0.002f : light.distance + 0.002f;
                shadowBits = lobe;
                if (!light.environment && !light.delta) {
                    // the light's own triangle bounds the search (see traceKernel); a ray
                    // that does not reach it at all cannot see the light
                    targetInstance = light.instance;
                    targetPrimitive = light.primitive;
                    float distance = 0.0f;
                    wantsShadow = lightTriangleDistance(instances, geometries, targetInstance, targetPrimitive,
                                                        shadowOrigin, shadowDirection, 0.001f, shadowLimit, distance);
                    shadowLimit = distance;
                }
            }
        }
        if (emits)
            b.sample[0][path] = make_float4(radiance.x, radiance.y, radiance.z, 0.0f);

        // caustic gather, device.cu:722-766, 949-950
    if (primaryChain && p.photonRadius > 0.0f && diffuseWeight > 0.0f) {
        const float radiusSquared = p.photonRadius * p.photonRadius;
        const float kernel = 2.0f / (3.14159265f * radiusSquared * 3.14159265f);
        const V3 weight = rgb(m.baseColor) * (diffuseWeight * kernel);
        if (INLINE_GATHER) {
            // the three float operations of causticGatherKernel, on the spot
            float fx, fy, fz;
            uint32_t gathered = 0;
            gatherFlux(p.sortedPhotons, p.photonCellStart, p.photonBucketCount, p.photonRadius, hit.position.x,
                       hit.position.y, hit.position.z, normal.x, normal.y, normal.z, fx, fy, fz, &gathered,
                       b.bucketBits);
            const V3 caustic = weight * V3{fx, fy, fz};
            const V3 sum = rgb(b.sample[3][path]) + throughput * caustic;
            b.sample[3][path] = make_float4(sum.x, sum.y, sum.z, 0.0f);
            out.gatherQueries = 1;
            out.gathered = gathered;
        } else {
            // the query is left for causticGatherKernel, which walks the photon map with dense
            // waves instead of the few lanes that want it here; the caller gives it its place in the queue
            out.wantsGather = true;
            out.gatherPoint = v3(hit.position);
            out.gatherNormal = normal;
            out.gatherThroughput = throughput;
            out.gatherWeight = weight;
        }
    }

    // scattering, device.cu:952-1032
        continues = true;
        const bool transmissive = m.transmission > 0.0f && random01(rng) < m.transmission;
        if (transmissive) {
            lastDelta = true;
            continues = dielectricEvent(m, hit, rng, origin, direction, throughput, medium);
        } else {
            const V3 color = rgb(m.baseColor);
            const V3 f0 = normalReflectance(m);
            const float viewCosine = fmaxf(dot(view, normal), 0.0f);
            const float probability = specularChoice(m, view, normal);
            if (random01(rng) < probability) {
                lastDelta = true;
                const float alpha = alphaOf(m.roughness);
                const V3 microNormal = sampleVisibleNormal(view, normal, alpha, rng);
                direction = reflectAbout(direction, microNormal);
                const float lightCosine = dot(direction, normal);
                if (lightCosine <= 0.0f)
                    continues = false;
                else
                    throughput = throughput * (schlickFresnel(f0, fmaxf(dot(view, microNormal), 0.0f)) *
                                               (maskingRatio(viewCosine, lightCosine, alpha) / probability));
            } else {
                lastDelta = false;
                primaryChain = false;
                direction = sampleCosine(normal, rng);
                const V3 reflectance = schlickFresnel(f0, viewCosine);
                const V3 diffuse{1.0f - reflectance.x, 1.0f - reflectance.y, 1.0f - reflectance.z};
                throughput = throughput * (color * diffuse * ((1.0f - m.metallic) / (1.0f - probability)));
                lastPdf = (1.0f - m.transmission) * (1.0f - probability) * fmaxf(dot(normal, direction), 0.0f) /
                          3.14159265f;
            }
            origin = v3(hit.position) + normal * 0.001f;
        }
        // Russian roulette, device.cu:1035-1040
        if (continues && depth > 2U) {
            const float survival = fminf(fmaxf(maximum(throughput), 0.05f), 0.95f);
            if (random01(rng) > survival)
                continues = false;
            else
                throughput = throughput / survival;
        }
        if (depth + 1U >= p.maxDepth)
            continues = false;
        if (continues) {
            out.rngAfter = rng;
            out.throughputAfter = make_float4(throughput.x, throughput.y, throughput.z, lastPdf);
            flags = lobe | (lastDelta ? kLastDelta : 0U) | (primaryChain ? kPrimaryChain : 0U) |
                    (guidePending ? kGuidePending : 0U) | static_cast<uint32_t>(medium + 1) << 8;
            out.lastOriginAfter = make_float4(hit.position.x, hit.position.y, hit.position.z, __uint_as_float(flags));
        }
    }
    out.origin = origin;
    out.direction = direction;
}

// One bounce for every path of queue `which`.
__global__ __launch_bounds__(kShadeBlock) void shadeKernel(rt_launch_params p, const ShadeInstance* instances,
                                                           const rt_hit_group_data* geometries, Buffers b, int which,
                                                           uint32_t depth) {
    __shared__ uint32_t slots[3 * kSlotWords];
    const uint32_t n = b.counts[which];
    if (blockIdx.x * kShadeBlock >= n)
        return;  // whole workgroup idle
    const uint32_t slot = blockIdx.x * kShadeBlock + threadIdx.x;
    ShadeOut out;
    uint32_t path = 0;
    if (slot < n) {
        path = b.rayPath[which][slot];
        const float4 o4 = b.rayOrigin[which][slot];
        const float4 d4 = b.rayDirection[which][slot];
        shadeOne<false>(p, instances, geometries, b, which, slot, path, V3{o4.x, o4.y, o4.z}, V3{d4.x, d4.y, d4.z},
                        b.hits[slot], b.hitInstance[slot], depth, out);
    }
    const bool continues = out.continues, wantsShadow = out.wantsShadow;
    const V3 origin = out.origin, direction = out.direction;
    const V3 shadowOrigin = out.shadowOrigin, shadowDirection = out.shadowDirection, contribution = out.contribution;
    const float shadowLimit = out.shadowLimit;
    const uint32_t shadowBits = out.shadowBits, targetInstance = out.targetInstance, targetPrimitive = out.targetPrimitive;

    const int next = which ^ 1;
    uint32_t nextSlot, shadowSlot, query = 0;
    if (p.photonRadius > 0.0f)  // the same for every thread: only then can a bounce leave a photon query
        allocateSlotTriple(continues, &b.counts[next], wantsShadow, &b.counts[2], out.wantsGather, &b.counts[3], slots, nextSlot,
                           shadowSlot, query);
    else
        allocateSlotPair(continues, &b.counts[next], wantsShadow, &b.counts[2], slots, nextSlot, shadowSlot);
    if (out.wantsGather) {
        b.gather[0][query] = make_float4(out.gatherPoint.x, out.gatherPoint.y, out.gatherPoint.z, __uint_as_float(path));
        b.gather[1][query] = make_float4(out.gatherNormal.x, out.gatherNormal.y, out.gatherNormal.z, 0.0f);
        b.gather[2][query] = make_float4(out.gatherThroughput.x, out.gatherThroughput.y, out.gatherThroughput.z, 0.0f);
        b.gather[3][query] = make_float4(out.gatherWeight.x, out.gatherWeight.y, out.gatherWeight.z, 0.0f);
    }
    if (continues) {
        b.rayOrigin[next][nextSlot] = make_float4(origin.x, origin.y, origin.z, 0.001f);
        b.rayDirection[next][nextSlot] = make_float4(direction.x, direction.y, direction.z, 1e16f);
        b.rayPath[next][nextSlot] = path;
        b.rng[next][nextSlot] = out.rngAfter;
        b.throughput[next][nextSlot] = out.throughputAfter;
        b.lastOrigin[next][nextSlot] = out.lastOriginAfter;
    }
    if (wantsShadow) {
        b.shadowOrigin[shadowSlot] = make_float4(shadowOrigin.x, shadowOrigin.y, shadowOrigin.z, 0.001f);
        b.shadowDirection[shadowSlot] = make_float4(shadowDirection.x, shadowDirection.y, shadowDirection.z, shadowLimit);
        b.shadowContribution[shadowSlot] =
            make_float4(contribution.x, contribution.y, contribution.z, __uint_as_float(shadowBits));
        b.shadowPath[shadowSlot] = path;
        b.shadowTarget[shadowSlot] = make_uint2(targetInstance, targetPrimitive);
    }
}

// A query that would read more photons than this is taken out of the lane that
// drew it: alone in its wave it is a chain of dependent loads that the whole
// launch ends up waiting for.
#ifndef RT_GATHER_HEAVY
#define RT_GATHER_HEAVY 256
#endif
constexpr uint32_t kHeavyPhotons = RT_GATHER_HEAVY;
constexpr unsigned kHeavyGatherBlocks = 512;

// The photon queries of one shade launch (device.cu:722-766, 949-950): each adds its density
// estimate to the caustic sum of its path.  A path has at most one query per bounce, so the
// order of the queue does not enter any sum.  Queries over the photon budget go on to
// causticGatherHeavyKernel.
__global__ __launch_bounds__(kShadeBlock, RT_GATHER_WAVES) void causticGatherKernel(rt_launch_params p, Buffers b,
                                                                                    PhotonMaps maps) {
    const uint32_t n = b.counts[3];
    if (blockIdx.x * kShadeBlock >= n)
        return;
    const uint32_t query = blockIdx.x * kShadeBlock + threadIdx.x;
    uint32_t gathered = 0;
    bool finished = false;
    if (query < n) {
        const float4 point = b.gather[0][query];
        const float4 normal = b.gather[1][query];
        float fx, fy, fz;
        const uint32_t path = __float_as_uint(point.w);
        const rt_photon* sorted;
        const uint32_t *cellStart, *bucketBits;
        maps.pick(p, b.bucketBits, path, sorted, cellStart, bucketBits);
        finished = gatherFlux(sorted, cellStart, p.photonBucketCount, p.photonRadius, point.x, point.y,
                              point.z, normal.x, normal.y, normal.z, fx, fy, fz, &gathered, bucketBits, b.heavyBudget);
        if (finished) {
            const V3 caustic = rgb(b.gather[3][query]) * V3{fx, fy, fz};
            const V3 sum = rgb(b.sample[3][path]) + rgb(b.gather[2][query]) * caustic;
            b.sample[3][path] = make_float4(sum.x, sum.y, sum.z, 0.0f);
        } else {
            const unsigned long long votes = __ballot(1);
            const int lane = threadIdx.x & 63;
            uint32_t base = 0;
            if (lane == __ffsll(static_cast<long long>(votes)) - 1)
                base = atomicAdd(&b.counts[4], static_cast<uint32_t>(__popcll(votes)));
            base = __builtin_amdgcn_readfirstlane(base);
            b.heavyQueries[base + __popcll(votes & ((1ULL << lane) - 1ULL))] = query;
        }
    }
    // statistics: one atomic pair per workgroup (every workgroup of a launch adds to the same two words)
    __shared__ unsigned long long waveQueries[kShadeWaves], wavePhotons[kShadeWaves];
    unsigned long long q = finished ? 1ULL : 0ULL, g = finished ? gathered : 0U;
#pragma unroll
    for (int delta = 32; delta >= 1; delta >>= 1) {
        q += __shfl_down(q, delta, 64);
        g += __shfl_down(g, delta, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        waveQueries[threadIdx.x >> 6] = q;
        wavePhotons[threadIdx.x >> 6] = g;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        q = 0;
        g = 0;
        for (int w = 0; w < kShadeWaves; ++w) {
            q += waveQueries[w];
            g += wavePhotons[w];
        }
        if (q) {
            atomicAdd(&b.stats[2], q);
            atomicAdd(&b.stats[3], g);
        }
    }
}

// The queries causticGatherKernel passed on, one per wave (gatherFluxWave); the waves of the
// launch take them in turn.
__global__ __launch_bounds__(kShadeBlock) void causticGatherHeavyKernel(rt_launch_params p, Buffers b, PhotonMaps maps) {
    const uint32_t n = b.counts[4];
    const uint32_t waves = gridDim.x * (kShadeBlock / 64);
    const uint32_t lane = threadIdx.x & 63U;
    unsigned long long queries = 0, photons = 0;  // of this wave, for the statistics (lane 0 counts)
    for (uint32_t item = blockIdx.x * (kShadeBlock / 64) + (threadIdx.x >> 6); item < n; item += waves) {
        const uint32_t query = b.heavyQueries[item];
        const float4 point = b.gather[0][query];
        const float4 normal = b.gather[1][query];
        float fx, fy, fz;
        uint32_t gathered = 0;
        const uint32_t path = __float_as_uint(point.w);
        const rt_photon* sorted;
        const uint32_t *cellStart, *bucketBits;
        maps.pick(p, b.bucketBits, path, sorted, cellStart, bucketBits);
        gatherFluxWave(sorted, cellStart, p.photonBucketCount, p.photonRadius, point.x, point.y, point.z,
                       normal.x, normal.y, normal.z, fx, fy, fz, gathered, bucketBits);
        if (lane == 0U) {
            const V3 caustic = rgb(b.gather[3][query]) * V3{fx, fy, fz};
            const V3 sum = rgb(b.sample[3][path]) + rgb(b.gather[2][query]) * caustic;
            b.sample[3][path] = make_float4(sum.x, sum.y, sum.z, 0.0f);
            ++queries;
            photons += gathered;
        }
    }
    if (lane == 0U && queries) {
        atomicAdd(&b.stats[2], queries);
        atomicAdd(&b.stats[3], photons);
    }
}

// Bookkeeping between bounces: ray statistics, then the consumed queues are emptied
// and the ray-feed cursors rewound.
__global__ void advanceKernel(Buffers b, int which) {
    for (int k = threadIdx.x; k < kFeedWords; k += blockDim.x)
        b.feed[k] = 0;
    if (threadIdx.x == 0 && blockIdx.x == 0 && which >= 0) {
        b.stats[0] += b.counts[which];
        b.stats[1] += b.counts[2];
        b.counts[which] = 0;
        b.counts[2] = 0;
        b.counts[3] = 0;
        b.counts[4] = 0;
    }
}

// device.cu:1043-1074: running means of the seven buffers and the tone-mapped byte image.
__global__ __launch_bounds__(kShadeBlock) void accumulateKernel(rt_launch_params p, PixelMap map, Buffers b) {
    const uint32_t pixelPath = blockIdx.x * blockDim.x + threadIdx.x;
    if (pixelPath >= map.perSample)
        return;
    const uint32_t index = map.pixel(pixelPath);
    // The running means of the pixel stay in registers over the samples of the pass: read once, written once.
    rt_float4* const buffers[7] = {p.accumulation, p.diffuse, p.reflection, p.refraction, p.caustics, p.albedoGuide,
                                   p.normalGuide};
    V3 mean[7];
#pragma unroll
    for (int k = 0; k < 7; ++k)
        mean[k] = rgb(buffers[k][index]);
    // the samples of a launch enter a pixel's means one after the other, in sample order
    for (uint32_t batch = 0; batch < map.samples; ++batch) {
        const uint32_t i = batch * map.perSample + pixelPath;
        const float weight = 1.0f / static_cast<float>(p.sample + batch + 1U);
        // the path's radiance belongs to the lobe of its first hit (lobe 0 when it met nothing); the other
        // two lobes got nothing from it
        const float4 guide = b.sample[5][i];
        const uint32_t lobe = __float_as_uint(guide.w) & 3U;
        const V3 radiance = rgb(b.sample[0][i]), nothing{0.0f, 0.0f, 0.0f};
        const V3 r0 = lobe == 0U ? radiance : nothing, r1 = lobe == 1U ? radiance : nothing,
                 r2 = lobe == 2U ? radiance : nothing;
        const V3 caustic = p.photonRadius > 0.0f ? rgb(b.sample[3][i]) : nothing;
        const V3 albedo = rgb(b.sample[4][i]), normal = rgb(guide);
        const V3 value[7] = {r0 + r1 + r2 + caustic, r0, r1, r2, caustic, albedo, normal};
#pragma unroll
        for (int k = 0; k < 7; ++k)
            mean[k] = mean[k] + (value[k] - mean[k]) * weight;
    }
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        rt_float4 out;
        out.x = mean[k].x;
        out.y = mean[k].y;
        out.z = mean[k].z;
        out.w = 1.0f;
        buffers[k][index] = out;
    }
    const V3 accumulated = mean[0];
    if (p.output) {
        const V3 mapped = acesTonemap(accumulated * mathExp2(p.exposure));
        rt_uchar4 out;
        out.x = static_cast<unsigned char>(mapped.x * 255.0f + 0.5f);
        out.y = static_cast<unsigned char>(mapped.y * 255.0f + 0.5f);
        out.z = static_cast<unsigned char>(mapped.z * 255.0f + 0.5f);
        out.w = 255;
        p.output[index] = out;
    }
}

// ---- photon pass --------------------------------------------------------------

// emitPhoton, device.cu:652-712, for emissions [first, first + count)
// Path i of a launch over several passes is emission first + i % perPass of pass p.photonPass + i / perPass
// (count = perPass x passes): the passes of consecutive photon maps share one launch of every stage.
__global__ __launch_bounds__(kShadeBlock) void photonGenerateKernel(rt_launch_params p, uint32_t first, uint32_t count,
                                                                    uint32_t perPass, Buffers b) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    bool valid = false;
    V3 origin{0, 0, 0}, direction{0, 0, 0}, power{0, 0, 0};
    uint32_t rng = 0;
    if (i < count) {
        b.photonFlags[i] = 0;
        const uint32_t pass = i / perPass;
        rng = seeded(first + (i - pass * perPass), p.photonPass + pass);
        float choice = 0.0f;
        const rt_gpu_light* selected = pickLight(p, b.lightIndex, rng, choice);
        if (selected) {
            const rt_gpu_light light = *selected;
            const float normalization = 1.0f / (static_cast<float>(p.photonEmissions) * choice);
            const V3 center = v3(p.sceneCenter);
            const float sceneRadius = p.sceneRadius;
            if (light.type == 4U) {
                const EnvironmentDraw source = drawEnvironment(p, rng);
                direction = -source.direction;
                origin = center - direction * sceneRadius + sampleDisk(direction, sceneRadius, rng);
                power = source.radiance * (3.14159265f * sceneRadius * sceneRadius * normalization / source.pdf);
            } else if (light.type == 3U) {
                const float root = sqrtf(random01(rng));
                const float u = 1.0f - root;
                const float v = random01(rng) * root;
                origin = v3(light.a) * u + v3(light.b) * v + v3(light.c) * (1.0f - u - v);
                const V3 side = random01(rng) < 0.5f ? v3(light.normal) : -v3(light.normal);
                direction = sampleCosine(side, rng);
                origin = origin + direction * 0.001f;
                power = v3(light.emission) * (6.2831853f * light.area * normalization);
            } else if (light.type == 0U) {
                direction = normalize(v3(light.b));
                origin = center - direction * sceneRadius + sampleDisk(direction, sceneRadius, rng);
                power = v3(light.emission) * (3.14159265f * sceneRadius * sceneRadius * normalization);
            } else {
                origin = v3(light.a);
                if (light.type == 1U) {
                    const float outer = mathCos(light.outerCone);
                    const float cosine = 1.0f - random01(rng) * (1.0f - outer);
                    direction = sampleAround(normalize(v3(light.b)), cosine, rng);
                    const float inner = mathCos(light.innerCone);
                    const float falloff = saturate((cosine - outer) / fmaxf(inner - outer, 1e-5f));
                    power = v3(light.emission) * (6.2831853f * (1.0f - outer) * falloff * falloff * normalization);
                } else {
                    const float cosine = 1.0f - 2.0f * random01(rng);
                    direction = sampleAround(V3{0.0f, 1.0f, 0.0f}, cosine, rng);
                    power = v3(light.emission) * (12.566371f * normalization);
                }
                origin = origin + direction * 0.001f;
            }
            valid = maximum(power) > 0.0f;
        }
    }
    // Every emission keeps the slot of its own index: nearly all of them carry power, and a slot drawn from the
    // queue's counter would cost an atomic per workgroup on one word - 260,000 of them in a launch over four passes
    // of 2^24 emissions, served one after the other (3 of the kernel's 4.3 ms).  An emission without power leaves a
    // ray with an empty interval, which meets nothing and ends at its first bounce.
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        b.counts[0] = count;
        b.counts[1] = 0;
        b.counts[2] = 0;
    }
    if (i < count) {
        b.rayOrigin[0][i] = make_float4(origin.x, origin.y, origin.z, 0.001f);
        b.rayDirection[0][i] = make_float4(direction.x, direction.y, direction.z, valid ? 1e16f : 0.0f);
        b.rayPath[0][i] = i;
        b.rng[0][i] = rng;
        b.throughput[0][i] = make_float4(power.x, power.y, power.z, 0.0f);
        b.lastOrigin[0][i] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(0U));  // flags: bit 0 specular path, medium
    }
}

// One bounce of device.cu:782-860 for every photon of queue `which`.
__global__ __launch_bounds__(kShadeBlock) void photonShadeKernel(rt_launch_params p, const ShadeInstance* instances,
                                                                 const rt_hit_group_data* geometries, Buffers b,
                                                                 int which, uint32_t depth) {
    __shared__ uint32_t slots[kSlotWords];
    const uint32_t n = b.counts[which];
    if (blockIdx.x * kShadeBlock >= n)
        return;
    const uint32_t slot = blockIdx.x * kShadeBlock + threadIdx.x;
    bool continues = false;
    V3 origin{0, 0, 0}, direction{0, 0, 0};
    uint32_t path = 0, rngAfter = 0, flagsAfter = 0;
    float4 powerAfter = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (slot < n) {
        path = b.rayPath[which][slot];
        const uint4 h = b.hits[slot];
        if (h.w != kNoHit) {
            uint32_t flags = __float_as_uint(b.lastOrigin[which][slot].w);
            bool specularPath = flags & 1U;
            // A photon that has met nothing specular yet ends, without a trace, on a surface that neither
            // transmits nor mirrors (device.cu:826-858 fall through): most photons of a pass, at their first
            // hit.  The maps of a material only scale its metallic and transmission factors down (texels
            // are at most one), so the factors alone say so, before the triangle's vertices are read.
            const uint32_t hitInstance = b.hitInstance[slot];
            const rt_gpu_material& factors = p.materials[instances[hitInstance].material];
            const bool ends = !specularPath && !(factors.transmission > 0.0f) && !(factors.metallic > 0.5f);
            if (!ends) {
            const float4 o4 = b.rayOrigin[which][slot];
            const float4 d4 = b.rayDirection[which][slot];
            origin = V3{o4.x, o4.y, o4.z};
            direction = V3{d4.x, d4.y, d4.z};
            const float4 pw = b.throughput[which][slot];
            V3 power{pw.x, pw.y, pw.z};
            int medium = static_cast<int>(flags >> 8) - 1;
            uint32_t rng = b.rng[which][slot];
            rt_hit hit;
            reconstructHit(instances, geometries, p.materials, origin, direction, __uint_as_float(h.x), __uint_as_float(h.y),
                           __uint_as_float(h.z), h.w, hitInstance, hit);
            const rt_gpu_material m = applyTextures(p.textures, p.materials[hit.material], hit);
            const V3 normal = shadingNormal(p.textures, hit, m);
            hit.normal = r3(normal);
            if (medium >= 0)
                power = power * mediumAbsorption(p.materials[medium], hit.distance);
            const float diffuseWeight = (1.0f - m.metallic) * (1.0f - m.transmission);
            if (specularPath && diffuseWeight > 0.0f) {
                // storePhoton, device.cu:714-720; its slot is decided after the batch, in emission order
                rt_photon photon;
                photon.position = hit.position;
                photon.power = r3(power);
                photon.normal = hit.normal;
                b.photonTemp[path] = photon;
                b.photonFlags[path] = 1;
            } else if (m.transmission > 0.0f && random01(rng) < m.transmission) {
                continues = dielectricEvent(m, hit, rng, origin, direction, power, medium);
                specularPath = true;
            } else if (m.metallic > 0.5f && m.roughness < 0.3f) {
                // device.cu:843-858
                const V3 view = -direction;
                const float alpha = alphaOf(m.roughness);
                const V3 microNormal = sampleVisibleNormal(view, normal, alpha, rng);
                const V3 reflected = reflectAbout(direction, microNormal);
                const float lightCosine = dot(reflected, normal);
                if (lightCosine > 0.0f) {
                    power = power * (schlickFresnel(normalReflectance(m), fmaxf(dot(view, microNormal), 0.0f)) *
                                     maskingRatio(fabsf(dot(view, normal)), lightCosine, alpha));
                    direction = reflected;
                    origin = v3(hit.position) + normal * 0.001f;
                    specularPath = true;
                    continues = true;
                }
            }
            if (depth + 1U >= 10U)
                continues = false;
            if (continues) {
                rngAfter = rng;
                powerAfter = make_float4(power.x, power.y, power.z, 0.0f);
                flagsAfter = (specularPath ? 1U : 0U) | static_cast<uint32_t>(medium + 1) << 8;
            }
            }
        }
    }
    const int next = which ^ 1;
    const uint32_t nextSlot = allocateSlots(continues, &b.counts[next], slots);
    if (continues) {
        b.rayOrigin[next][nextSlot] = make_float4(origin.x, origin.y, origin.z, 0.001f);
        b.rayDirection[next][nextSlot] = make_float4(direction.x, direction.y, direction.z, 1e16f);
        b.rayPath[next][nextSlot] = path;
        b.rng[next][nextSlot] = rngAfter;
        b.throughput[next][nextSlot] = powerAfter;
        b.lastOrigin[next][nextSlot] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(flagsAfter));
    }
}

// Appends the batch's stored photons behind *photonCount in emission order
// (offsets = exclusive scan of the flags; offsets[count] = photons of the batch).
// Where the photons of the passes of one launch go: a list, its counter and (optionally) its key array per pass.
struct PhotonLists {
    rt_photon* photons[RT_MAX_PHOTON_PASSES];
    uint32_t* counter[RT_MAX_PHOTON_PASSES];
    uint32_t* keys[RT_MAX_PHOTON_PASSES];
};

// Appends the batch's stored photons behind *counter of their pass in emission order (offsets = exclusive scan of
// the flags over all paths, pass after pass; offsets[count] = photons of the batch).  With keys the photon's bucket
// key goes to keys[slot] (photon_hash.cuh:10-24 at cell size p.photonRadius): the position is in registers here,
// where the grid build would read the whole record again for it.
__global__ __launch_bounds__(256) void photonCompactKernel(rt_launch_params p, uint32_t count, uint32_t perPass,
                                                           const uint32_t* offsets, Buffers b, PhotonLists lists) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count)
        return;
    if (offsets[i + 1] != offsets[i]) {
        const uint32_t pass = i / perPass;
        const uint32_t slot = *lists.counter[pass] + (offsets[i] - offsets[pass * perPass]);
        if (slot < p.photonCapacity) {
            const rt_photon photon = b.photonTemp[i];
            lists.photons[pass][slot] = photon;
            if (lists.keys[pass])
                lists.keys[pass][slot] = bucketOfPosition(photon.position.x, photon.position.y, photon.position.z,
                                                          p.photonRadius, p.photonBucketCount);
        }
    }
}

__global__ void photonCountKernel(uint32_t perPass, uint32_t passes, const uint32_t* offsets, PhotonLists lists) {
    const uint32_t pass = threadIdx.x;
    if (blockIdx.x == 0 && pass < passes)
        *lists.counter[pass] += offsets[(pass + 1U) * perPass] - offsets[pass * perPass];
}

// A kernel with the private segment of the ray kernels and nothing to do (rounds == 0): launched once on a stream,
// it makes the stream's hardware queue set up its scratch memory ahead of the first ray launch.  1 KiB per lane covers
// the 928 to 960 bytes the ray kernels take for the deep end of their stacks.
__global__ __launch_bounds__(kTraceBlock) void scratchSetupKernel(uint32_t* sink, uint32_t rounds) {
    uint32_t spill[256];
    if (rounds == 0U)
        return;
    for (uint32_t i = 0; i < 256U; ++i)
        spill[(i * 7U + threadIdx.x) & 255U] = i;
    uint32_t sum = 0;
    for (uint32_t i = 0; i < rounds; ++i)
        sum += spill[(sum + i) & 255U];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = sum;
}

}  // namespace
}  // namespace rt

struct StageTiming {
    int stage;
    hipEvent_t start, stop;
};

struct rt_wavefront {
    int device = 0;
    uint32_t flags = 0;               // RT_WAVEFRONT_TIME_STAGES | RT_WAVEFRONT_COUNT_TRAVERSAL
    uint32_t shareNumerator = 1, shareDenominator = 1;  // part of the resident wave slots a ray launch takes
    uint32_t batchSamples = 1, batchSeedStride = 1;     // consecutive samples a render call takes in one pass
    hipEvent_t accumulateAfter = nullptr;   // the accumulation of a sample waits for this event (or null)
    hipEvent_t accumulateSignal = nullptr;  // ... and records this one behind it (or null)
    std::vector<StageTiming> timings; // pending event pairs (flag TIME_STAGES)
    std::vector<hipEvent_t> eventPool;
    rt::Buffers buffers{};
    std::vector<void*> allocations;
    uint32_t* scanCounts = nullptr;    // capacity + 1
    uint32_t* photonKeys = nullptr;    // bucket keys of stored photons go here too (rt_wavefront_set_photon_keys)
    uint32_t lightIndexTag = 0;        // names the light array the index was built from
    rt::PhotonMaps photonMaps{};       // the maps of the next pass (rt_wavefront_set_photon_maps); count 0: the parameters' map
    uint32_t uses = RT_WAVEFRONT_FOR_CAMERA | RT_WAVEFRONT_FOR_PHOTONS;
    uint32_t* scanPartials = nullptr;
};

using namespace rt;

namespace {

// Every device array of a wavefront state in one place: `visit(pointer, count, uses)` is called once per
// array with the kinds of pass that need it, so the allocation and the size query cannot drift apart.
constexpr uint32_t kAllUses = RT_WAVEFRONT_FOR_CAMERA | RT_WAVEFRONT_FOR_CAMERA_NO_CAUSTICS | RT_WAVEFRONT_FOR_PHOTONS;

template <typename Visit>
void forEachArray(rt_wavefront& w, size_t maxPaths, Visit&& visit) {
    // `gathers`: camera passes that read a photon map; `camera`: those and the ones that never do
    constexpr uint32_t gathers = RT_WAVEFRONT_FOR_CAMERA, camera = gathers | RT_WAVEFRONT_FOR_CAMERA_NO_CAUSTICS;
    constexpr uint32_t photons = RT_WAVEFRONT_FOR_PHOTONS, both = camera | photons;
    Buffers& b = w.buffers;
    for (int k = 0; k < 2; ++k) {
        visit(b.rayOrigin[k], maxPaths, both);
        visit(b.rayDirection[k], maxPaths, both);
        visit(b.rayPath[k], maxPaths, both);
    }
    visit(b.hits, maxPaths, both);
    visit(b.hitInstance, maxPaths, both);
    visit(b.shadowOrigin, maxPaths, camera);
    visit(b.shadowDirection, maxPaths, camera);
    visit(b.shadowContribution, maxPaths, camera);
    visit(b.shadowPath, maxPaths, camera);
    visit(b.shadowTarget, maxPaths, camera);
    for (int k = 0; k < 2; ++k) {
        visit(b.rng[k], maxPaths, both);
        visit(b.throughput[k], maxPaths, both);
        visit(b.lastOrigin[k], maxPaths, both);
    }
    for (int k = 0; k < 6; ++k)
        if (k != 1 && k != 2)
            visit(b.sample[k], maxPaths, k == 3 ? gathers : camera);  // 3: the caustic sum
    for (int k = 0; k < 4; ++k)
        visit(b.gather[k], maxPaths, gathers);
    visit(b.heavyQueries, maxPaths, gathers);
    visit(b.counts, size_t{8}, both);
    visit(b.feed, static_cast<size_t>(kFeedWords), both);
    visit(b.stats, size_t{8}, both);
    visit(b.photonTemp, maxPaths, photons);
    visit(b.photonFlags, maxPaths + 1, photons);
    visit(w.scanCounts, maxPaths + 1, photons);
    visit(w.scanPartials, scanScratchEntries(maxPaths + 1), photons);
}

}  // namespace

extern "C" {

void rt_wavefront_destroy(rt_wavefront* w) {
    if (!w)
        return;
    for (void* p : w->allocations)
        (void)hipFree(p);
    for (auto& t : w->timings) {
        (void)hipEventDestroy(t.start);
        (void)hipEventDestroy(t.stop);
    }
    for (auto e : w->eventPool)
        (void)hipEventDestroy(e);
    delete w;
}

rt_status rt_wavefront_state_bytes(uint32_t maxPaths, uint32_t uses, size_t* bytes) {
    if (!bytes)
        return failInvalid("bytes is null");
    if (uses == 0 || (uses & ~kAllUses))
        return failInvalid("uses: a sum of RT_WAVEFRONT_FOR_CAMERA, _CAMERA_NO_CAUSTICS and _PHOTONS");
    rt_wavefront scratch;
    size_t total = 0;
    forEachArray(scratch, maxPaths, [&](auto*& pointer, size_t count, uint32_t neededBy) {
        using T = std::remove_reference_t<decltype(*pointer)>;
        if (neededBy & uses)
            total += alignUp(sizeof(T) * count, 256);
    });
    *bytes = total;
    return RT_SUCCESS;
}

rt_status rt_wavefront_create_for(uint32_t maxPaths, uint32_t uses, rt_wavefront** out) {
    if (!out)
        return failInvalid("out is null");
    *out = nullptr;
    if (maxPaths == 0)
        return failInvalid("maxPaths is zero");
    if (uses == 0 || (uses & ~kAllUses))
        return failInvalid("uses: a sum of RT_WAVEFRONT_FOR_CAMERA, _CAMERA_NO_CAUSTICS and _PHOTONS");
    auto* w = new rt_wavefront();
    w->uses = uses;
    RT_HIP(hipGetDevice(&w->device));
    rt_status status = RT_SUCCESS;
    forEachArray(*w, maxPaths, [&](auto*& pointer, size_t count, uint32_t neededBy) {
        using T = std::remove_reference_t<decltype(*pointer)>;
        void* raw = nullptr;
        if (status != RT_SUCCESS || !(neededBy & uses))
            return;
        const hipError_t e = hipMalloc(&raw, sizeof(T) * count);
        if (e != hipSuccess) {
            (void)hipGetLastError();  // the failure is reported through the status; the next call starts clean
            status = fail(e, "hipMalloc (wavefront buffers)");
            return;
        }
        w->allocations.push_back(raw);
        pointer = static_cast<T*>(raw);
        // development aid: RT_WAVEFRONT_POISON=<byte> fills every array, so that a read of something never
        // written shows up as a changed image instead of depending on what the allocator handed out
        if (const char* poison = std::getenv("RT_WAVEFRONT_POISON"))
            (void)hipMemset(raw, std::atoi(poison) & 0xff, sizeof(T) * count);
    });
    Buffers& b = w->buffers;
    b.capacity = maxPaths;
    b.heavyBudget = kHeavyPhotons;
    if (status == RT_SUCCESS) {
        if (hipMemset(b.counts, 0, sizeof(uint32_t) * 8) != hipSuccess ||
            hipMemset(b.stats, 0, sizeof(unsigned long long) * 8) != hipSuccess)
            status = failInvalid("hipMemset (wavefront counters)");
    }
    if (status != RT_SUCCESS) {
        rt_wavefront_destroy(w);
        return status;
    }
    *out = w;
    return RT_SUCCESS;
}

rt_status rt_wavefront_create(uint32_t maxPaths, rt_wavefront** out) {
    return rt_wavefront_create_for(maxPaths, RT_WAVEFRONT_FOR_CAMERA | RT_WAVEFRONT_FOR_PHOTONS, out);
}

rt_status rt_wavefront_stats(rt_wavefront* w, uint64_t* out8, int reset, void* stream) {
    if (!w || !out8)
        return failInvalid("null pointer");
    auto s = static_cast<hipStream_t>(stream);
    RT_HIP(hipMemcpyAsync(out8, w->buffers.stats, sizeof(uint64_t) * 8, hipMemcpyDeviceToHost, s));
    RT_HIP(hipStreamSynchronize(s));
    if (reset)
        RT_HIP(hipMemsetAsync(w->buffers.stats, 0, sizeof(uint64_t) * 8, s));
    return RT_SUCCESS;
}

rt_status rt_wavefront_configure(rt_wavefront* w, uint32_t flags) {
    if (!w)
        return failInvalid("null pointer");
    w->flags = flags;
    w->buffers.heavyBudget = flags & RT_WAVEFRONT_GATHER_PER_LANE ? 0xffffffffU : kHeavyPhotons;
    return RT_SUCCESS;
}

rt_status rt_wavefront_set_ray_share(rt_wavefront* w, uint32_t numerator, uint32_t denominator) {
    if (!w)
        return failInvalid("null pointer");
    if (numerator == 0 || denominator == 0 || numerator > denominator)
        return failInvalid("the share of a ray launch is a fraction in (0, 1]");
    w->shareNumerator = numerator;
    w->shareDenominator = denominator;
    return RT_SUCCESS;
}

rt_status rt_wavefront_prepare_stream(rt_wavefront* w, void* stream) {
    if (!w)
        return failInvalid("null pointer");
    // the grid of a full ray launch: the scratch memory of a queue is sized by the launch that asks for it
    hipLaunchKernelGGL(scratchSetupKernel, dim3(persistentBlocks(w->device)), dim3(kTraceBlock), 0,
                       static_cast<hipStream_t>(stream), static_cast<uint32_t*>(nullptr), 0U);
    RT_HIP(hipGetLastError());
    return RT_SUCCESS;
}

rt_status rt_wavefront_set_sample_batch(rt_wavefront* w, uint32_t samples, uint32_t seedStride) {
    if (!w)
        return failInvalid("null pointer");
    if (samples == 0 || samples > 64)
        return failInvalid("a batch holds 1 to 64 samples");
    w->batchSamples = samples;
    w->batchSeedStride = seedStride;
    return RT_SUCCESS;
}

rt_status rt_wavefront_set_accumulate_events(rt_wavefront* w, void* waitFor, void* record) {
    if (!w)
        return failInvalid("null pointer");
    w->accumulateAfter = static_cast<hipEvent_t>(waitFor);
    w->accumulateSignal = static_cast<hipEvent_t>(record);
    return RT_SUCCESS;
}

rt_status rt_wavefront_set_photon_keys(rt_wavefront* w, uint32_t* keys) {
    if (!w)
        return failInvalid("null pointer");
    w->photonKeys = keys;
    return RT_SUCCESS;
}

rt_status rt_wavefront_set_photon_maps(rt_wavefront* w, const rt_photon_map* maps, uint32_t count, uint32_t samplesPerMap,
                                       uint32_t phase) {
    if (!w)
        return failInvalid("null pointer");
    if (count > RT_MAX_PHOTON_MAPS)
        return failInvalid("more photon maps than RT_MAX_PHOTON_MAPS");
    if (count && (!maps || samplesPerMap == 0 || phase >= samplesPerMap))
        return failInvalid("photon maps: null table, no samples per map, or a phase beyond them");
    for (uint32_t k = 0; k < count; ++k)
        if (!maps[k].sortedPhotons || !maps[k].cellStart)
            return failInvalid("photon maps: a map without sorted photons or cell starts");
    w->photonMaps = PhotonMaps{};
    for (uint32_t k = 0; k < count; ++k)
        w->photonMaps.maps[k] = maps[k];
    w->photonMaps.count = count;
    w->photonMaps.samplesPerMap = count ? samplesPerMap : 1U;
    w->photonMaps.phase = phase;
    return RT_SUCCESS;
}

rt_status rt_wavefront_set_bucket_bitmap(rt_wavefront* w, const uint32_t* bitmap) {
    if (!w)
        return failInvalid("null pointer");
    w->buffers.bucketBits = bitmap;
    return RT_SUCCESS;
}

rt_status rt_wavefront_set_light_index(rt_wavefront* w, const void* index, uint32_t lightCount) {
    if (!w)
        return failInvalid("null pointer");
    w->buffers.lightIndex = rt::LightIndex{};
    if (!index || lightCount == 0)
        return RT_SUCCESS;
    uint32_t header[4] = {};
    RT_HIP(hipMemcpy(header, index, sizeof(header), hipMemcpyDeviceToHost));
    if (header[0] != lightCount || header[1] != rt::lightTableCapacity(lightCount))
        return failInvalid("the light index was built for another light count");
    w->buffers.lightIndex = rt::lightIndexView(index, lightCount, header[2] != 0U);
    w->lightIndexTag = header[3];
    return RT_SUCCESS;
}

rt_status rt_wavefront_timings(rt_wavefront* w, float* milliseconds5, uint32_t* launches5, void* stream) {
    if (!w || !milliseconds5 || !launches5)
        return failInvalid("null pointer");
    RT_HIP(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    for (int k = 0; k < 5; ++k) {
        milliseconds5[k] = 0.0f;
        launches5[k] = 0;
    }
    for (auto& t : w->timings) {
        float ms = 0.0f;
        RT_HIP(hipEventElapsedTime(&ms, t.start, t.stop));
        milliseconds5[t.stage] += ms;
        ++launches5[t.stage];
        w->eventPool.push_back(t.start);
        w->eventPool.push_back(t.stop);
    }
    w->timings.clear();
    return RT_SUCCESS;
}

namespace {

// Brackets one launch with events when stage timing is on.
struct StageTimer {
    rt_wavefront* w;
    hipStream_t stream;
    int stage;
    hipEvent_t stop = nullptr;
    StageTimer(rt_wavefront* wavefront, hipStream_t s, int stageIndex) : w(wavefront), stream(s), stage(stageIndex) {
        if (!(w->flags & RT_WAVEFRONT_TIME_STAGES))
            return;
        hipEvent_t start = take();
        stop = take();
        if (start && stop) {
            (void)hipEventRecord(start, stream);
            w->timings.push_back({stage, start, stop});
        } else {
            stop = nullptr;
        }
    }
    ~StageTimer() {
        if (stop)
            (void)hipEventRecord(stop, stream);
    }
    hipEvent_t take() {
        if (!w->eventPool.empty()) {
            hipEvent_t e = w->eventPool.back();
            w->eventPool.pop_back();
            return e;
        }
        hipEvent_t e = nullptr;
        return hipEventCreate(&e) == hipSuccess ? e : nullptr;
    }
};

}  // namespace

// The state's buffers for a launch with these parameters: the light index only describes the light array it
// was built from; for parameters that carry another array (or another count) the launch walks the list, which
// gives the reference's choice whatever the list.
static Buffers buffersFor(const rt_wavefront* w, const rt_launch_params* params) {
    Buffers b = w->buffers;
    if (b.lightIndex.count != params->lightCount || w->lightIndexTag != lightArrayTag(params->lights))
        b.lightIndex = LightIndex{};
    return b;
}

// Array sizes the ray stage checks references against: top-level nodes, instances.
static uint2 limitsOf(const rt_scene* scene) {
    return make_uint2(static_cast<uint32_t>(scene->nodeCount), itemCountWord(scene));
}

static rt_status checkLaunch(const rt_wavefront* w, const rt_launch_params* params, const rt_scene*& scene) {
    if (!w || !params)
        return failInvalid("null pointer");
    scene = reinterpret_cast<const rt_scene*>(params->scene);
    if (!scene)
        return failInvalid("launch parameters carry no scene handle");
    int device = 0;
    RT_HIP(hipGetDevice(&device));
    if (device != w->device || device != scene->device)
        return failInvalid("wavefront state or scene lives on another device");
    if (!scene->hitGroups || !scene->shadeInstances)
        return failInvalid("rt_scene_set_geometry has not been called");
    if (!params->materials)
        return failInvalid("launch parameters carry no materials");
    if (!fitsTraversalStack(scene->depth, scene->blasDepth)) {  // rt_scene_build refuses such a scene already
        setLastError("tree too deep for the traversal stack");
        return RT_ERROR_OUT_OF_RANGE;
    }
    return RT_SUCCESS;
}

#ifdef RT_TRACE_PROFILE
// Development aid: copies out (and optionally clears) the per-site issue / lane sums of the wavefront kernels.
rt_status rt_trace_profile(float* out64, int reset) {
    RT_HIP(hipDeviceSynchronize());
    RT_HIP(hipMemcpyFromSymbol(out64, HIP_SYMBOL(gTraceProfile), sizeof(float) * 64));
    if (reset) {
        const float zero[64] = {};
        RT_HIP(hipMemcpyToSymbol(HIP_SYMBOL(gTraceProfile), zero, sizeof(zero)));
    }
    return RT_SUCCESS;
}
rt_status rt_trace_histogram(uint32_t* out64, int reset) {
    RT_HIP(hipDeviceSynchronize());
    RT_HIP(hipMemcpyFromSymbol(out64, HIP_SYMBOL(gTraceHistogram), sizeof(uint32_t) * 64));
    if (reset) {
        const uint32_t zero[64] = {};
        RT_HIP(hipMemcpyToSymbol(HIP_SYMBOL(gTraceHistogram), zero, sizeof(zero)));
    }
    return RT_SUCCESS;
}
#endif

rt_status rt_render_sample(rt_wavefront* w, const rt_launch_params* params, uint32_t pixelBegin, uint32_t pixelEnd,
                           void* stream) {
    if (!params)
        return failInvalid("null pointer");
    return rt_render_sample_seeded(w, params, pixelBegin, pixelEnd, params->sample, stream);
}

// The samples of the wavefront's batch for the pixels of `pixels` (perSample set, at least 1 path): fills in the
// batch and checks it against the capacity.
static rt_status renderPaths(rt_wavefront* w, const rt_launch_params* params, const rt_scene* scene, const PixelMap& pixels,
                             uint32_t seedSample, hipStream_t s) {
    PixelMap map = pixels;
    map.samples = w->batchSamples;
    map.seedStride = w->batchSeedStride;
    if (static_cast<uint64_t>(map.perSample) * map.samples > w->buffers.capacity)
        return failInvalid("pixels x samples of the batch larger than the wavefront capacity");
    map.count = map.perSample * map.samples;
    const uint32_t count = map.count;
    const Buffers b = buffersFor(w, params);
    if (!(w->uses & (RT_WAVEFRONT_FOR_CAMERA | RT_WAVEFRONT_FOR_CAMERA_NO_CAUSTICS)))
        return failInvalid("this wavefront state was created for photon passes only");
    if (!(w->uses & RT_WAVEFRONT_FOR_CAMERA) && params->photonRadius > 0.0f)
        return failInvalid("this wavefront state was created without the photon-query arrays: photonRadius must be 0");
    PhotonMaps photonMaps = w->photonMaps;
    photonMaps.perSample = map.perSample;
    const unsigned blocks = divUp(count, kShadeBlock);
    const unsigned resident = std::max<unsigned>(
        1U, static_cast<unsigned>(static_cast<uint64_t>(persistentBlocks(w->device)) * w->shareNumerator / w->shareDenominator));
    const unsigned traceBlocks = std::min<unsigned>(divUp(count, kTraceBlock), resident);
    const bool counted = w->flags & RT_WAVEFRONT_COUNT_TRAVERSAL;
    RT_HIP(hipMemsetAsync(b.feed, 0, sizeof(uint32_t) * kFeedWords, s));
    {
        StageTimer timer(w, s, 0);
        hipLaunchKernelGGL(generateKernel, dim3(blocks), dim3(kShadeBlock), 0, s, *params, map, seedSample, b);
    }
    const uint2 limits = limitsOf(scene);
    // stage 1 is the ray stage (traceKernel), stage 3 the bookkeeping between bounces
    auto trace = [&](int which, int withShadow) {
        StageTimer timer(w, s, 1);
        if (counted)
            hipLaunchKernelGGL(traceKernel<true>, dim3(traceBlocks), dim3(kTraceBlock), 0, s, scene->nodes, scene->instances,
                               b, which, withShadow, limits);
        else
            hipLaunchKernelGGL(traceKernel<false>, dim3(traceBlocks), dim3(kTraceBlock), 0, s, scene->nodes, scene->instances,
                               b, which, withShadow, limits);
    };
    if (params->maxDepth > 0) {
        trace(0, 0);
        StageTimer timer(w, s, 3);
        hipLaunchKernelGGL(advanceKernel, dim3(1), dim3(256), 0, s, b, -1);
    }
    for (uint32_t depth = 0; depth < params->maxDepth; ++depth) {
        const int which = depth & 1U;
        {
            StageTimer timer(w, s, 2);
            hipLaunchKernelGGL(shadeKernel, dim3(blocks), dim3(kShadeBlock), 0, s, *params, scene->shadeInstances,
                               scene->hitGroups, b, which, depth);
            if (params->photonRadius > 0.0f) {
                hipLaunchKernelGGL(causticGatherKernel, dim3(blocks), dim3(kShadeBlock), 0, s, *params, b, photonMaps);
                if (b.heavyBudget != 0xffffffffU)
                    hipLaunchKernelGGL(causticGatherHeavyKernel, dim3(std::min(blocks, kHeavyGatherBlocks)),
                                       dim3(kShadeBlock), 0, s, *params, b, photonMaps);
            }
        }
        // the shadow rays of this bounce travel with the extend rays of the next
        trace(depth + 1 < params->maxDepth ? 1 - which : -1, 1);
        StageTimer timer(w, s, 3);
        hipLaunchKernelGGL(advanceKernel, dim3(1), dim3(256), 0, s, b, which);
    }
    if (w->accumulateAfter)
        RT_HIP(hipStreamWaitEvent(s, w->accumulateAfter, 0));
    {
        StageTimer timer(w, s, 4);
        hipLaunchKernelGGL(accumulateKernel, dim3(divUp(map.perSample, kShadeBlock)), dim3(kShadeBlock), 0, s, *params, map, b);
    }
    RT_HIP(hipGetLastError());
    if (w->accumulateSignal)
        RT_HIP(hipEventRecord(w->accumulateSignal, s));
    return RT_SUCCESS;
}

// What both sample entry points check before anything is launched.
static rt_status checkSample(rt_wavefront* w, const rt_launch_params* params, const rt_scene*& scene) {
    RT_TRY(checkLaunch(w, params, scene));
    if (params->width == 0 || params->height == 0 ||
        static_cast<uint64_t>(params->width) * params->height > 0xffffffffULL)
        return failInvalid("image size");
    // a positive photon radius switches the caustic gather on; it reads these
    if (params->photonRadius > 0.0f &&
        (!params->sortedPhotons || !params->photonCellStart || params->photonBucketCount == 0 ||
         (params->photonBucketCount & (params->photonBucketCount - 1U)) != 0))
        return failInvalid("photon radius set without a photon grid (sorted photons, cell starts, power-of-two bucket count)");
    return RT_SUCCESS;
}

rt_status rt_render_sample_seeded(rt_wavefront* w, const rt_launch_params* params, uint32_t pixelBegin,
                                  uint32_t pixelEnd, uint32_t seedSample, void* stream) {
    const rt_scene* scene = nullptr;
    RT_TRY(checkSample(w, params, scene));
    if (pixelEnd < pixelBegin || pixelEnd > params->width * params->height)
        return failInvalid("pixel range");
    const uint32_t count = pixelEnd - pixelBegin;
    if (count == 0)
        return RT_SUCCESS;
    if (count > w->buffers.capacity)
        return failInvalid("pixel range larger than the wavefront capacity");
    const uint32_t width = params->width, band = 8U * width;
    PixelMap map{};
    map.begin = pixelBegin;
    map.perSample = count;
    map.width = width;
    // full bands first when the range starts on a row and rows split into 8-pixel tiles
    map.bandPaths = (pixelBegin % width == 0U && width % 8U == 0U) ? count / band * band : 0U;
    map.bandStride = band;
    map.tailBegin = pixelBegin + map.bandPaths;
    map.tiled = 1;
    return renderPaths(w, params, scene, map, seedSample, static_cast<hipStream_t>(stream));
}

rt_status rt_render_sample_bands(rt_wavefront* w, const rt_launch_params* params, uint32_t firstBand,
                                 uint32_t bandStride, uint32_t bandCount, uint32_t seedSample, void* stream) {
    const rt_scene* scene = nullptr;
    RT_TRY(checkSample(w, params, scene));
    if (bandStride == 0)
        return failInvalid("band stride is zero");
    const uint32_t width = params->width, height = params->height;
    const uint32_t bands = (height + 7U) / 8U;
    if (bandCount == 0 || firstBand >= bands)
        return RT_SUCCESS;
    // the launch's bands: firstBand + k * bandStride below `bands`
    const uint32_t covered = std::min<uint64_t>(bandCount, (static_cast<uint64_t>(bands - 1U - firstBand) / bandStride) + 1U);
    const uint32_t lastBand = firstBand + (covered - 1U) * bandStride;
    const uint32_t lastRows = std::min(8U, height - lastBand * 8U);
    const uint32_t fullBands = lastRows == 8U ? covered : covered - 1U;
    const uint64_t count = static_cast<uint64_t>(fullBands) * 8U * width + (lastRows == 8U ? 0U : lastRows * width);
    if (count > w->buffers.capacity)
        return failInvalid("bands larger than the wavefront capacity");
    const uint32_t band = 8U * width;
    PixelMap map{};
    map.begin = firstBand * band;
    map.perSample = static_cast<uint32_t>(count);
    map.width = width;
    map.bandStride = bandStride * band;
    map.tailBegin = lastBand * band;  // only used when the last band is short
    map.bandPaths = fullBands * band;
    map.tiled = width % 8U == 0U ? 1U : 0U;
    return renderPaths(w, params, scene, map, seedSample, static_cast<hipStream_t>(stream));
}

rt_status rt_trace_photon_passes(rt_wavefront* w, const rt_launch_params* params, uint32_t firstEmission,
                                 uint32_t emissionCount, uint32_t passes, rt_photon* const* lists, uint32_t* const* counters,
                                 uint32_t* const* keys, void* stream) {
    const rt_scene* scene = nullptr;
    RT_TRY(checkLaunch(w, params, scene));
    if (passes == 0 || passes > RT_MAX_PHOTON_PASSES || !lists || !counters)
        return failInvalid("photon passes: between 1 and RT_MAX_PHOTON_PASSES, with a list and a counter each");
    if (!(w->uses & RT_WAVEFRONT_FOR_PHOTONS))
        return failInvalid("this wavefront state was created for camera passes only");
    PhotonLists all{};
    bool anyKeys = false;
    for (uint32_t k = 0; k < passes; ++k) {
        if (!lists[k] || !counters[k])
            return failInvalid("launch parameters carry no photon buffers");
        all.photons[k] = lists[k];
        all.counter[k] = counters[k];
        all.keys[k] = keys ? keys[k] : nullptr;
        anyKeys = anyKeys || all.keys[k];
    }
    if (anyKeys && (!(params->photonRadius > 0.0f) || params->photonBucketCount == 0 ||
                    (params->photonBucketCount & (params->photonBucketCount - 1U)) != 0))
        return failInvalid("photon keys asked for without a cell size (photonRadius) and a power-of-two bucket count");
    auto s = static_cast<hipStream_t>(stream);
    const Buffers b = buffersFor(w, params);
    // Whole passes side by side as long as the state holds them; a pass that does not fit by itself goes in pieces.
    const uint32_t together = emissionCount ? std::min<uint64_t>(passes, b.capacity / emissionCount) : passes;
    uint32_t pass = 0;
    while (pass < passes) {
        const uint32_t group = together >= 1U ? std::min(together, passes - pass) : 1U;
        rt_launch_params launch = *params;
        launch.photonPass = params->photonPass + pass;
        PhotonLists mine{};
        for (uint32_t k = 0; k < group; ++k) {
            mine.photons[k] = all.photons[pass + k];
            mine.counter[k] = all.counter[pass + k];
            mine.keys[k] = all.keys[pass + k];
        }
        uint32_t done = 0;
        while (done < emissionCount) {
            const uint32_t perPass = together >= 1U ? emissionCount : std::min(b.capacity, emissionCount - done);
            const uint32_t count = perPass * group;
            const unsigned blocks = divUp(count, kShadeBlock);
            const unsigned traceBlocks = std::min<unsigned>(divUp(count, kTraceBlock), persistentBlocks(w->device));
            const uint2 limits = limitsOf(scene);
            RT_HIP(hipMemsetAsync(b.counts, 0, sizeof(uint32_t) * 8, s));
            RT_HIP(hipMemsetAsync(b.feed, 0, sizeof(uint32_t) * kFeedWords, s));
            hipLaunchKernelGGL(photonGenerateKernel, dim3(blocks), dim3(kShadeBlock), 0, s, launch, firstEmission + done,
                               count, perPass, b);
            for (uint32_t depth = 0; depth < 10U; ++depth) {
                const int which = depth & 1U;
                hipLaunchKernelGGL(traceKernel<false>, dim3(traceBlocks), dim3(kTraceBlock), 0, s, scene->nodes,
                                   scene->instances, b, which, 0, limits);
                hipLaunchKernelGGL(photonShadeKernel, dim3(blocks), dim3(kShadeBlock), 0, s, launch, scene->shadeInstances,
                                   scene->hitGroups, b, which, depth);
                hipLaunchKernelGGL(advanceKernel, dim3(1), dim3(256), 0, s, b, which);
            }
            // emission-ordered append: offsets = exclusive scan of the flags (with a
            // trailing zero so that offsets[count] is the batch total)
            RT_HIP(hipMemsetAsync(b.photonFlags + count, 0, sizeof(uint32_t), s));
            RT_TRY(exclusiveScanU32(b.photonFlags, w->scanCounts, static_cast<size_t>(count) + 1, w->scanPartials, s));
            hipLaunchKernelGGL(photonCompactKernel, dim3(divUp(count, 256)), dim3(256), 0, s, launch, count, perPass,
                               w->scanCounts, b, mine);
            hipLaunchKernelGGL(photonCountKernel, dim3(1), dim3(64), 0, s, perPass, group, w->scanCounts, mine);
            RT_HIP(hipGetLastError());
            done += perPass;
        }
        pass += group;
    }
    return RT_SUCCESS;
}

rt_status rt_trace_photons(rt_wavefront* w, const rt_launch_params* params, uint32_t firstEmission,
                           uint32_t emissionCount, void* stream) {
    if (!w || !params)
        return failInvalid("null pointer");
    if (!params->photons || !params->photonCount)
        return failInvalid("launch parameters carry no photon buffers");
    rt_photon* list = params->photons;
    uint32_t* counter = params->photonCount;
    uint32_t* keys = w->photonKeys;
    return rt_trace_photon_passes(w, params, firstEmission, emissionCount, 1, &list, &counter, &keys, stream);
}

}  // extern "C"
