// Ray-query kernels: closest hit and any hit over a batch of rays.
//
// Replaces the optixTrace call of the reference (device.cu:388-397) and the
// occlusion use of it in visible() (device.cu:548-559).  Persistent waves, one
// ray per lane at a time; the traversal itself lives in traverse_core.h.
#include "radix_sort.h"
#include "traverse_core.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace rt {
namespace {

template <bool ANY, bool COUNT>
__global__ __launch_bounds__(kTraceBlock, RT_TRACE_MIN_WAVES) void traceKernel(const WideNode* topNodes, uint32_t topCount, const InstanceRecord* instances,
                                                           uint32_t instanceCount, const float4* originTmin, const float4* directionTmax,
                                                           uint32_t n, uint4* hits, uint32_t* instanceOut,
                                                           uint8_t* occluded, unsigned long long* counters,
                                                           uint32_t* cursors, const uint32_t* order) {
    __shared__ uint32_t stackRef[kLdsStack][kTraceBlock];
    __shared__ float stackT[kLdsStack][kTraceBlock];

    uint32_t spillRef[kPrivateStack];
    float spillT[kPrivateStack];
    Traversal<COUNT> walk(topNodes, topCount, instances, instanceCount, ldsView(&stackRef[0][threadIdx.x]),
                          ldsView(&stackT[0][threadIdx.x]), kTraceBlock, spillRef, spillT);
    RayFeed feed(cursors, n);
    unsigned long long rays = 0;
    tracePersistent<COUNT>(
        walk, feed,
        // with `order` the k-th ray handed out is ray order[k] of the caller's arrays (see orderRays)
        [&](uint32_t k) {
            const uint32_t ray = order ? order[k] : k;
            return RayIn{originTmin[ray], directionTmax[ray], make_uint2(kNoHit, kNoHit)};
        },
        [&](uint32_t, const RayIn& in, Traversal<COUNT>& w) {
            const float4 o4 = in.origin, d4 = in.direction;
            w.begin(Ray{o4.x, o4.y, o4.z, d4.x, d4.y, d4.z}, o4.w, d4.w, kNoHit, kNoHit, ANY);
            ++rays;
        },
        [&](uint32_t k, const TraceResult& r) {
            const uint32_t ray = order ? order[k] : k;
            const bool found = r.primitive != kNoHit;
            if (ANY) {
                occluded[ray] = found ? 1 : 0;
            } else {
                hits[ray] = make_uint4(__float_as_uint(found ? r.t : -1.0f), __float_as_uint(found ? r.u : 0.0f),
                                       __float_as_uint(found ? r.v : 0.0f), r.primitive);
                instanceOut[ray] = r.instance;
            }
        });

    if (COUNT) {
        // wave sums first, then one atomic per wave and counter
        TraceCounters local = walk.counters;
#pragma unroll
        for (int delta = 32; delta >= 1; delta >>= 1) {
            rays += __shfl_down(rays, delta, 64);
            local.nodes += __shfl_down(local.nodes, delta, 64);
            local.triangles += __shfl_down(local.triangles, delta, 64);
            local.instances += __shfl_down(local.instances, delta, 64);
        }
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&counters[0], rays);
            atomicAdd(&counters[1], local.nodes);
            atomicAdd(&counters[2], local.triangles);
            atomicAdd(&counters[3], local.instances);
        }
    }
}

// ---- rays in the order of their origins ------------------------------------------------------------------------
//
// A large batch of rays over a scene that does not fit the caches is traced faster when neighbouring lanes and
// waves start in the same part of the scene, whatever their directions: the nodes and triangles around the origin
// are then fetched once for many rays.  Measured on MI355X, 2^24 uniformly random rays, given in that order
// against physically sorted by the Morton code of a 32 x 32 x 32 grid of origin cells: 1 M triangle soup 2310 ->
// 2500 Mrays/s, 10 M 1640 -> 2130, 100 M 1320 -> 1820 (finer cells and the direction's octant add nothing); on the
// cache-resident 871 k triangle frame nothing is won.  So the batched queries order such a batch themselves:
// a key per ray, a stable two-pass radix sort of (key, index), and the kernel takes its k-th ray through the sorted
// index.  The caller's arrays stay as they are and every ray's result is what it would have been.  With the
// sort (0.3 ms for 2^24 rays) and the scattered reads of the rays and writes of the results in the launch's time:
// 10 M 1650 -> 1865 Mrays/s, 100 M 1320 -> 1600.

constexpr int kOrderBitsPerAxis = 5;
constexpr uint32_t kOrderMinimumRays = 1U << 18;
constexpr uint64_t kOrderMinimumSceneBytes = 128ULL << 20;  // the 67 MB of a 1 M triangle soup break even (2343 against 2357 Mrays/s)

__global__ __launch_bounds__(256) void rayKeysKernel(const float4* originTmin, uint32_t n, float3 lo, float3 scale,
                                                     uint32_t* keys) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const float4 o = originTmin[i];
    const float top = static_cast<float>((1 << kOrderBitsPerAxis) - 1);
    // fmaxf / fminf drop a NaN: such a ray lands in cell 0
    const uint32_t x = static_cast<uint32_t>(fminf(fmaxf((o.x - lo.x) * scale.x, 0.0f), top));
    const uint32_t y = static_cast<uint32_t>(fminf(fmaxf((o.y - lo.y) * scale.y, 0.0f), top));
    const uint32_t z = static_cast<uint32_t>(fminf(fmaxf((o.z - lo.z) * scale.z, 0.0f), top));
    uint32_t key = 0;
#pragma unroll
    for (int b = 0; b < kOrderBitsPerAxis; ++b)
        key |= ((x >> b) & 1U) << (3 * b) | ((y >> b) & 1U) << (3 * b + 1) | ((z >> b) & 1U) << (3 * b + 2);
    keys[i] = key;
}

// Whether a batch of n rays is ordered first: RT_TRACE_ORDER=0 never, =1 always (tests), else by the sizes above.
bool orderChoice(const rt_scene* scene, uint32_t n) {
    if (n < 2U)
        return false;
    if (const char* v = std::getenv("RT_TRACE_ORDER")) {
        if (std::strcmp(v, "0") == 0)
            return false;
        if (std::strcmp(v, "1") == 0)
            return true;
    }
    const uint64_t sceneBytes = scene->blasNodes * sizeof(WideNode) + scene->triangles * sizeof(PackedTriangle);
    return n >= kOrderMinimumRays && sceneBytes >= kOrderMinimumSceneBytes;
}

// Called with scene->feedMutex held.  Leaves the sorted index in *order (inside the scene's workspace, which the
// caller marks busy behind its kernel by recording scene->orderEvent).
rt_status orderRays(const rt_scene* scene, const float4* originTmin, uint32_t n, hipStream_t stream, const uint32_t** order) {
    const size_t entries = (static_cast<size_t>(n) + 63U) & ~static_cast<size_t>(63U);
    const size_t tableEntries = sortTableEntries(n), partialEntries = scanScratchEntries(tableEntries);
    const size_t needed = sizeof(uint32_t) * (4U * entries + ((tableEntries + 63U) & ~static_cast<size_t>(63U)) + partialEntries);
    if (!scene->orderEvent)
        RT_HIP(hipEventCreateWithFlags(&scene->orderEvent, hipEventDisableTiming));
    if (scene->orderBytes < needed) {
        if (scene->orderData) {
            RT_HIP(hipDeviceSynchronize());  // nothing may still read the block that is freed
            RT_HIP(hipFree(scene->orderData));
            scene->orderData = nullptr;
            scene->orderBytes = 0;
        }
        RT_HIP(hipMalloc(&scene->orderData, needed));
        scene->orderBytes = needed;
    } else if (scene->orderEventRecorded) {
        RT_HIP(hipStreamWaitEvent(stream, scene->orderEvent, 0));
    }
    uint32_t* base = static_cast<uint32_t*>(scene->orderData);
    uint32_t* const keys[2] = {base, base + entries};
    uint32_t* const idx[2] = {base + 2U * entries, base + 3U * entries};
    uint32_t* table = base + 4U * entries;
    uint32_t* partials = table + ((tableEntries + 63U) & ~static_cast<size_t>(63U));
    const float cells = static_cast<float>(1 << kOrderBitsPerAxis);
    float3 lo = make_float3(scene->bounds.lo[0], scene->bounds.lo[1], scene->bounds.lo[2]);
    float3 scale;
    const float extent[3] = {scene->bounds.hi[0] - scene->bounds.lo[0], scene->bounds.hi[1] - scene->bounds.lo[1],
                             scene->bounds.hi[2] - scene->bounds.lo[2]};
    scale.x = extent[0] > 0.0f ? cells / extent[0] : 0.0f;
    scale.y = extent[1] > 0.0f ? cells / extent[1] : 0.0f;
    scale.z = extent[2] > 0.0f ? cells / extent[2] : 0.0f;
    hipLaunchKernelGGL(rayKeysKernel, dim3(divUp(n, 256)), dim3(256), 0, stream, originTmin, n, lo, scale, keys[0]);
    RT_HIP(hipGetLastError());
    int result = 0;
    RT_TRY(radixSortPairs<uint32_t>(keys, idx, n, 3 * kOrderBitsPerAxis, table, partials, stream, &result));
    *order = idx[result];
    return RT_SUCCESS;
}

template <bool ANY, bool COUNT>
rt_status launchTrace(const rt_scene* scene, const rt_float4* originTmin, const rt_float4* directionTmax, uint32_t n,
                      rt_ray_hit* hits, uint32_t* instance, uint8_t* occluded, uint64_t* counters4, void* stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    // one slot of the scene's ring of cursor sets, private to this launch (see rt_scene::feedEvents)
    std::lock_guard<std::mutex> lock(scene->feedMutex);
    const uint32_t slot = scene->feedNext++ % kFeedSlots;
    if (scene->feedEventRecorded[slot])
        RT_HIP(hipStreamWaitEvent(s, scene->feedEvents[slot], 0));
    uint32_t* cursors = scene->feedCursors + slot * kFeedWords;
    RT_HIP(hipMemsetAsync(cursors, 0, kFeedWords * sizeof(uint32_t), s));
    const uint32_t* order = nullptr;
    if (orderChoice(scene, n))
        RT_TRY(orderRays(scene, reinterpret_cast<const float4*>(originTmin), n, s, &order));
    const uint32_t blocks = std::min<uint32_t>(divUp(n, kTraceBlock), persistentBlocks(scene->device));
    hipLaunchKernelGGL((traceKernel<ANY, COUNT>), dim3(blocks), dim3(kTraceBlock), 0, s, scene->nodes,
                       static_cast<uint32_t>(scene->nodeCount), scene->instances, itemCountWord(scene),
                       reinterpret_cast<const float4*>(originTmin), reinterpret_cast<const float4*>(directionTmax), n,
                       reinterpret_cast<uint4*>(hits), instance, occluded,
                       reinterpret_cast<unsigned long long*>(counters4), cursors, order);
    RT_HIP(hipGetLastError());
    RT_HIP(hipEventRecord(scene->feedEvents[slot], s));
    scene->feedEventRecorded[slot] = true;
    if (order) {
        RT_HIP(hipEventRecord(scene->orderEvent, s));
        scene->orderEventRecorded = true;
    }
    return RT_SUCCESS;
}

rt_status checkScene(const rt_scene* scene) {
    if (!scene)
        return failInvalid("scene is null");
    int device = 0;
    RT_HIP(hipGetDevice(&device));
    if (device != scene->device)
        return failInvalid("scene was built on another device");
    if (!fitsTraversalStack(scene->depth, scene->blasDepth)) {  // rt_scene_build refuses such a scene already
        setLastError("tree too deep for the traversal stack");
        return RT_ERROR_OUT_OF_RANGE;
    }
    return RT_SUCCESS;
}

}  // namespace
}  // namespace rt

using namespace rt;

extern "C" {

#ifdef RT_TRACE_PROFILE
// Development aid: launch timeline of the batched queries (see gTraceClock); reset arms the min / max slots.
rt_status rt_trace_clock(unsigned long long* out8, int reset) {
    RT_HIP(hipDeviceSynchronize());
    RT_HIP(hipMemcpyFromSymbol(out8, HIP_SYMBOL(gTraceClock), sizeof(unsigned long long) * 8));
    if (reset) {
        const unsigned long long init[8] = {~0ULL, ~0ULL, 0, 0, 0, 0, 0, 0};
        RT_HIP(hipMemcpyToSymbol(HIP_SYMBOL(gTraceClock), init, sizeof(init)));
    }
    return RT_SUCCESS;
}
#endif

rt_status rt_trace_closest(const rt_scene* scene, const rt_float4* originTmin, const rt_float4* directionTmax,
                           uint32_t n, rt_ray_hit* hits, uint32_t* instance, void* stream) {
    RT_TRY(checkScene(scene));
    if (n == 0)
        return RT_SUCCESS;
    if (!originTmin || !directionTmax || !hits || !instance)
        return failInvalid("null pointer");
    return launchTrace<false, false>(scene, originTmin, directionTmax, n, hits, instance, nullptr, nullptr, stream);
}

rt_status rt_trace_closest_counted(const rt_scene* scene, const rt_float4* originTmin,
                                   const rt_float4* directionTmax, uint32_t n, rt_ray_hit* hits, uint32_t* instance,
                                   uint64_t* counters4, void* stream) {
    RT_TRY(checkScene(scene));
    if (n == 0)
        return RT_SUCCESS;
    if (!originTmin || !directionTmax || !hits || !instance || !counters4)
        return failInvalid("null pointer");
    return launchTrace<false, true>(scene, originTmin, directionTmax, n, hits, instance, nullptr, counters4, stream);
}

rt_status rt_trace_any(const rt_scene* scene, const rt_float4* originTmin, const rt_float4* directionTmax, uint32_t n,
                       uint8_t* occluded, void* stream) {
    RT_TRY(checkScene(scene));
    if (n == 0)
        return RT_SUCCESS;
    if (!originTmin || !directionTmax || !occluded)
        return failInvalid("null pointer");
    return launchTrace<true, false>(scene, originTmin, directionTmax, n, nullptr, nullptr, occluded, nullptr, stream);
}

}  // extern "C"
