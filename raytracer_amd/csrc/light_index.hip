// Builds the light index (light_index.h) from a light table in device memory, and answers batches of
// selections and look-ups through it for callers that want the reference's light choice
// (device.cu:433-447, 607-621) outside a render.
#include "light_index.h"

#include "common.h"
#include "rt_types.h"

#include <cmath>

namespace rt {
namespace {

constexpr int kBlock = 256;

__global__ __launch_bounds__(kBlock) void lightIndexClearKernel(const rt_gpu_light* lights, uint32_t count, uint32_t capacity,
                                                                uint32_t withThresholds, uint32_t* words) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const size_t head = 4 + 2 * static_cast<size_t>(count);
    const size_t aligned = (head + 1) & ~static_cast<size_t>(1);
    if (i == 0) {
        words[0] = count;
        words[1] = capacity;
        words[2] = withThresholds;
        words[3] = lightArrayTag(lights);  // which array the index describes (rt_wavefront_set_light_index)
        if (aligned != head)
            words[head] = 0;
    }
    if (i < count)
        reinterpret_cast<float*>(words + 4 + count)[i] = lights[i].weight;
    if (i < capacity) {
        reinterpret_cast<unsigned long long*>(words + aligned)[i] = kNoLightKey;
        words[aligned + 2 * static_cast<size_t>(capacity) + i] = ~0U;
    }
}

__global__ __launch_bounds__(kBlock) void lightIndexInsertKernel(const rt_gpu_light* lights, uint32_t count, uint32_t capacity,
                                                                 uint32_t* words) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count || lights[i].type != 3U)
        return;
    const size_t head = 4 + 2 * static_cast<size_t>(count);
    const size_t aligned = (head + 1) & ~static_cast<size_t>(1);
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(words + aligned);
    uint32_t* values = words + aligned + 2 * static_cast<size_t>(capacity);
    const unsigned long long key = lightKey(lights[i].instance, lights[i].primitive);
    const uint32_t mask = capacity - 1U;
    uint32_t slot = lightSlot(key, mask);
    // the table has at least twice as many slots as lights, so a free slot is always reached
    for (uint32_t probes = 0; probes <= mask; ++probes) {
        const unsigned long long before = atomicCAS(&keys[slot], kNoLightKey, key);
        if (before == kNoLightKey || before == key) {
            atomicMin(&values[slot], i);  // several lights on one triangle: the walk finds the first
            return;
        }
        slot = (slot + 1U) & mask;
    }
}

// Largest float t with fl(t - w) <= y, for y >= 0 and w >= 0: the walk's step run backwards.
__device__ inline float stepBack(float y, float w) {
    // t stays positive where it is lowered (0 - w is never above y) and non-negative where it is
    // raised, so the neighbouring floats are the neighbouring bit patterns
    float t = y + w;
    while (t - w > y)
        t = __uint_as_float(__float_as_uint(t) - 1U);
    for (;;) {
        const float above = __uint_as_float(__float_as_uint(t) + 1U);
        if (!(above - w <= y))
            break;
        t = above;
    }
    return t;
}

__global__ __launch_bounds__(kBlock) void lightThresholdKernel(const float* weights, uint32_t count, float* thresholds) {
    // long chains first: light count - 1 has the most steps
    const uint32_t i = count - 1U - (blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= count)
        return;
    float t = 0.0f;
    for (uint32_t k = i + 1U; k-- > 0U;)
        t = stepBack(t, weights[k]);
    thresholds[i] = t;
}

__global__ __launch_bounds__(kBlock) void lightWeightCheckKernel(const rt_gpu_light* lights, uint32_t count, uint32_t* bad) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count && !(lights[i].weight >= 0.0f && lights[i].weight < INFINITY))
        atomicAdd(bad, 1U);
}

__global__ __launch_bounds__(kBlock) void lightQueryKernel(const rt_gpu_light* lights, uint32_t count, float lightWeight,
                                                           LightIndex index, const float* uniform, const uint32_t* instance,
                                                           const uint32_t* primitive, uint32_t n, uint32_t* chosen,
                                                           uint32_t* found) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    if (uniform && chosen) {
        const float target = uniform[i] * lightWeight;
        uint32_t pick;
        if (index.thresholds) {
            pick = lightByThreshold(index, target);
        } else {  // the walk itself
            float rest = target;
            pick = count - 1U;
            for (uint32_t k = 0; k < count; ++k) {
                rest -= lights[k].weight;
                if (rest <= 0.0f) {
                    pick = k;
                    break;
                }
            }
        }
        chosen[i] = pick;
    }
    if (instance && primitive && found) {
        uint32_t light = ~0U;
        if (index.keys) {
            light = lightOfTriangle(index, instance[i], primitive[i]);
        } else {
            for (uint32_t k = 0; k < count; ++k)
                if (lights[k].type == 3U && lights[k].instance == instance[i] && lights[k].primitive == primitive[i]) {
                    light = k;
                    break;
                }
        }
        found[i] = light;
    }
}

}  // namespace
}  // namespace rt

rt_status rt_light_index_bytes(uint32_t lightCount, size_t* bytes) {
    if (!bytes)
        return rt::failInvalid("null pointer");
    *bytes = rt::lightIndexWords(lightCount) * sizeof(uint32_t);
    return RT_SUCCESS;
}

rt_status rt_light_index_build(const rt_gpu_light* lights, uint32_t lightCount, void* index, size_t indexBytes, void* stream) {
    if (!index || (lightCount && !lights))
        return rt::failInvalid("null pointer");
    if (lightCount >= 0x40000000U)
        return rt::failInvalid("too many lights");
    if (reinterpret_cast<uintptr_t>(index) % 8 != 0)
        return rt::failInvalid("the light index must be aligned to 8 bytes");
    if (indexBytes < rt::lightIndexWords(lightCount) * sizeof(uint32_t))
        return rt::failInvalid("light index memory smaller than rt_light_index_bytes()");
    hipStream_t s = static_cast<hipStream_t>(stream);
    uint32_t* words = static_cast<uint32_t*>(index);
    const uint32_t capacity = rt::lightTableCapacity(lightCount);
    // a weight that is negative, infinite or not a number has no place in the walk's order: such a
    // table keeps the walk (word [2] stays 0)
    uint32_t bad = 0;
    if (lightCount) {
        RT_HIP(hipMemsetAsync(words + 3, 0, sizeof(uint32_t), s));
        hipLaunchKernelGGL(rt::lightWeightCheckKernel, dim3(rt::divUp(lightCount, rt::kBlock)), dim3(rt::kBlock), 0, s, lights,
                           lightCount, words + 3);
        RT_HIP(hipGetLastError());
        RT_HIP(hipMemcpyAsync(&bad, words + 3, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        RT_HIP(hipStreamSynchronize(s));
    }
    const uint32_t withThresholds = lightCount > 0 && lightCount <= rt::kMaxThresholdLights && bad == 0 ? 1U : 0U;
    hipLaunchKernelGGL(rt::lightIndexClearKernel, dim3(rt::divUp(capacity > lightCount ? capacity : lightCount, rt::kBlock)),
                       dim3(rt::kBlock), 0, s, lights, lightCount, capacity, withThresholds, words);
    RT_HIP(hipGetLastError());
    if (lightCount) {
        hipLaunchKernelGGL(rt::lightIndexInsertKernel, dim3(rt::divUp(lightCount, rt::kBlock)), dim3(rt::kBlock), 0, s, lights,
                           lightCount, capacity, words);
        RT_HIP(hipGetLastError());
    }
    if (withThresholds) {
        hipLaunchKernelGGL(rt::lightThresholdKernel, dim3(rt::divUp(lightCount, rt::kBlock)), dim3(rt::kBlock), 0, s,
                           reinterpret_cast<const float*>(words + 4 + lightCount), lightCount,
                           reinterpret_cast<float*>(words + 4));
        RT_HIP(hipGetLastError());
    }
    // set-up work: the index is complete when the call returns, whatever stream reads it next
    RT_HIP(hipStreamSynchronize(s));
    return RT_SUCCESS;
}

rt_status rt_light_query(const rt_gpu_light* lights, uint32_t lightCount, float lightWeight, const void* index,
                         const float* uniform, const uint32_t* instance, const uint32_t* primitive, uint32_t n,
                         uint32_t* chosen, uint32_t* found, void* stream) {
    if (!lights || lightCount == 0)
        return rt::failInvalid("no lights");
    if (n == 0)
        return RT_SUCCESS;
    hipStream_t s = static_cast<hipStream_t>(stream);
    rt::LightIndex view;
    if (index) {
        uint32_t header[4] = {};
        RT_HIP(hipMemcpyAsync(header, index, sizeof(header), hipMemcpyDeviceToHost, s));
        RT_HIP(hipStreamSynchronize(s));
        if (header[0] != lightCount || header[1] != rt::lightTableCapacity(lightCount))
            return rt::failInvalid("the light index was built for another light count");
        view = rt::lightIndexView(index, lightCount, header[2] != 0U);
    }
    hipLaunchKernelGGL(rt::lightQueryKernel, dim3(rt::divUp(n, rt::kBlock)), dim3(rt::kBlock), 0, s, lights, lightCount,
                       lightWeight, view, uniform, instance, primitive, n, chosen, found);
    RT_HIP(hipGetLastError());
    return RT_SUCCESS;
}
