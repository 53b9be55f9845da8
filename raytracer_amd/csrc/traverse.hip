// Ray-query kernels: closest hit and any hit over a batch of rays.
//
// Replaces the optixTrace call of the reference (device.cu:388-397) and the
// occlusion use of it in visible() (device.cu:548-559).  One ray per lane; the
// traversal itself lives in traverse_core.h.
#include "traverse_core.h"

namespace rt {
namespace {

template <bool ANY, bool COUNT>
__global__ __launch_bounds__(kTraceBlock, RT_TRACE_MIN_WAVES) void traceKernel(const WideNode* topNodes, const InstanceRecord* instances,
                                                           const float4* originTmin, const float4* directionTmax,
                                                           uint32_t n, uint4* hits, uint32_t* instanceOut,
                                                           uint8_t* occluded, unsigned long long* counters) {
    __shared__ uint32_t stackRef[kLdsStack][kTraceBlock];
    __shared__ float stackT[kLdsStack][kTraceBlock];

    const uint32_t ray = xcdRemap(blockIdx.x, gridDim.x) * kTraceBlock + threadIdx.x;
    const bool active = ray < n;
    TraceCounters local;

    if (active) {
        const float4 o4 = originTmin[ray];
        const float4 d4 = directionTmax[ray];
        const Ray world{o4.x, o4.y, o4.z, d4.x, d4.y, d4.z};
        const TraceResult r = traceRay<COUNT>(topNodes, instances, world, o4.w, d4.w, kNoHit, kNoHit, ANY,
                                              ldsView(&stackRef[0][threadIdx.x]), ldsView(&stackT[0][threadIdx.x]), kTraceBlock, local);
        const bool found = r.primitive != kNoHit;
        if (ANY) {
            occluded[ray] = found ? 1 : 0;
        } else {
            hits[ray] = make_uint4(__float_as_uint(found ? r.t : -1.0f), __float_as_uint(found ? r.u : 0.0f),
                                   __float_as_uint(found ? r.v : 0.0f), r.primitive);
            instanceOut[ray] = r.instance;
        }
    }

    if (COUNT) {
        // wave sums first, then one atomic per wave and counter
        unsigned long long rays = active ? 1ULL : 0ULL;
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

rt_status checkScene(const rt_scene* scene) {
    if (!scene)
        return failInvalid("scene is null");
    int device = 0;
    RT_HIP(hipGetDevice(&device));
    if (device != scene->device)
        return failInvalid("scene was built on another device");
    // every visited node leaves at most seven entries behind, plus one marker
    // per instance entered
    if ((kWide - 1) * (scene->depth + scene->blasDepth) + 2 > static_cast<uint32_t>(kMaxStack)) {
        setLastError("tree too deep for the traversal stack");
        return RT_ERROR_OUT_OF_RANGE;
    }
    return RT_SUCCESS;
}

}  // namespace
}  // namespace rt

using namespace rt;

extern "C" {

rt_status rt_trace_closest(const rt_scene* scene, const rt_float4* originTmin, const rt_float4* directionTmax,
                           uint32_t n, rt_ray_hit* hits, uint32_t* instance, void* stream) {
    RT_TRY(checkScene(scene));
    if (n == 0)
        return RT_SUCCESS;
    if (!originTmin || !directionTmax || !hits || !instance)
        return failInvalid("null pointer");
    hipLaunchKernelGGL((traceKernel<false, false>), dim3(divUp(n, kTraceBlock)), dim3(kTraceBlock), 0,
                       static_cast<hipStream_t>(stream), scene->nodes, scene->instances,
                       reinterpret_cast<const float4*>(originTmin), reinterpret_cast<const float4*>(directionTmax), n,
                       reinterpret_cast<uint4*>(hits), instance, nullptr, nullptr);
    RT_HIP(hipGetLastError());
    return RT_SUCCESS;
}

rt_status rt_trace_closest_counted(const rt_scene* scene, const rt_float4* originTmin,
                                   const rt_float4* directionTmax, uint32_t n, rt_ray_hit* hits, uint32_t* instance,
                                   uint64_t* counters4, void* stream) {
    RT_TRY(checkScene(scene));
    if (n == 0)
        return RT_SUCCESS;
    if (!originTmin || !directionTmax || !hits || !instance || !counters4)
        return failInvalid("null pointer");
    hipLaunchKernelGGL((traceKernel<false, true>), dim3(divUp(n, kTraceBlock)), dim3(kTraceBlock), 0,
                       static_cast<hipStream_t>(stream), scene->nodes, scene->instances,
                       reinterpret_cast<const float4*>(originTmin), reinterpret_cast<const float4*>(directionTmax), n,
                       reinterpret_cast<uint4*>(hits), instance, nullptr,
                       reinterpret_cast<unsigned long long*>(counters4));
    RT_HIP(hipGetLastError());
    return RT_SUCCESS;
}

rt_status rt_trace_any(const rt_scene* scene, const rt_float4* originTmin, const rt_float4* directionTmax, uint32_t n,
                       uint8_t* occluded, void* stream) {
    RT_TRY(checkScene(scene));
    if (n == 0)
        return RT_SUCCESS;
    if (!originTmin || !directionTmax || !occluded)
        return failInvalid("null pointer");
    hipLaunchKernelGGL((traceKernel<true, false>), dim3(divUp(n, kTraceBlock)), dim3(kTraceBlock), 0,
                       static_cast<hipStream_t>(stream), scene->nodes, scene->instances,
                       reinterpret_cast<const float4*>(originTmin), reinterpret_cast<const float4*>(directionTmax), n,
                       nullptr, nullptr, occluded, nullptr);
    RT_HIP(hipGetLastError());
    return RT_SUCCESS;
}

}  // extern "C"
