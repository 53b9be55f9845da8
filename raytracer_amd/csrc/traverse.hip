// Ray-query kernels: closest hit and any hit over a batch of rays.
//
// Replaces the optixTrace call of the reference (device.cu:388-397) and the
// occlusion use of it in visible() (device.cu:548-559).  Persistent waves, one
// ray per lane at a time; the traversal itself lives in traverse_core.h.
#include "traverse_core.h"

#include <algorithm>

namespace rt {
namespace {

template <bool ANY, bool COUNT>
__global__ __launch_bounds__(kTraceBlock, RT_TRACE_MIN_WAVES) void traceKernel(const WideNode* topNodes, uint32_t topCount, const InstanceRecord* instances,
                                                           uint32_t instanceCount, const float4* originTmin, const float4* directionTmax,
                                                           uint32_t n, uint4* hits, uint32_t* instanceOut,
                                                           uint8_t* occluded, unsigned long long* counters,
                                                           uint32_t* cursors) {
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
        [&](uint32_t ray) { return RayIn{originTmin[ray], directionTmax[ray], make_uint2(kNoHit, kNoHit)}; },
        [&](uint32_t, const RayIn& in, Traversal<COUNT>& w) {
            const float4 o4 = in.origin, d4 = in.direction;
            w.begin(Ray{o4.x, o4.y, o4.z, d4.x, d4.y, d4.z}, o4.w, d4.w, kNoHit, kNoHit, ANY);
            ++rays;
        },
        [&](uint32_t ray, const TraceResult& r) {
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
    const uint32_t blocks = std::min<uint32_t>(divUp(n, kTraceBlock), persistentBlocks(scene->device));
    hipLaunchKernelGGL((traceKernel<ANY, COUNT>), dim3(blocks), dim3(kTraceBlock), 0, s, scene->nodes,
                       static_cast<uint32_t>(scene->nodeCount), scene->instances, itemCountWord(scene),
                       reinterpret_cast<const float4*>(originTmin), reinterpret_cast<const float4*>(directionTmax), n,
                       reinterpret_cast<uint4*>(hits), instance, occluded,
                       reinterpret_cast<unsigned long long*>(counters4), cursors);
    RT_HIP(hipGetLastError());
    RT_HIP(hipEventRecord(scene->feedEvents[slot], s));
    scene->feedEventRecorded[slot] = true;
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
