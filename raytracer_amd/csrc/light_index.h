// Light selection without walking the light table.
//
// The reference picks a light by subtracting the weights from a random target one after the other
// until the remainder is no longer positive (device.cu:433-447) and finds the record of an emissive
// triangle it hit by comparing every light with the hit (device.cu:607-621): both are a walk over
// the whole table per shaded point.  The index below answers both without the walk and with the
// same result for every input:
//
//   * thresholds[i] is the largest float target for which the walk stops at light i or before it.
//     Every step of the walk, fl(t - w), is a non-decreasing function of t, so the targets that
//     have stopped by light i form an interval (-inf, thresholds[i]] and the thresholds never
//     decrease with i: the walk's light is the first i with target <= thresholds[i], the last
//     light when there is none.  thresholds[i] is found by running the walk backwards from a
//     remainder of zero, each step taking the largest float that the forward step maps onto or
//     below the value behind it.  That is i steps for light i, the square of the light count in
//     total, which a GPU does once per scene for up to kMaxThresholdLights lights.
//   * a hash table from (instance, primitive) to the lowest index of an emissive-triangle light
//     with that pair, which is the light the walk over the table finds first.
//
// Memory layout of an index for n lights (32-bit words; rt_light_index_bytes):
//   [0] n  [1] table capacity c (power of two >= 2 n)  [2] 1 when the thresholds are there  [3] 0
//   thresholds[n]  weights[n]  (pad to 8 bytes)  keys[c] (64 bit)  values[c]
#pragma once

#include <cstddef>
#include <cstdint>

#include <hip/hip_runtime.h>

namespace rt {

constexpr uint32_t kMaxThresholdLights = 1U << 20;
// Below this many lights the walk itself is as fast as the search.
constexpr uint32_t kLightWalkLimit = 8;
constexpr unsigned long long kNoLightKey = ~0ULL;

struct LightIndex {
    const float* thresholds = nullptr;            // null: walk the table
    const float* weights = nullptr;
    const unsigned long long* keys = nullptr;     // null: walk the table
    const uint32_t* values = nullptr;
    uint32_t count = 0;
    uint32_t mask = 0;                            // capacity - 1
};

// A word that names the light array an index was built from (its address folded to 32 bits): the launches
// compare it with the array of their parameters and walk the list when it is another one.
__host__ __device__ inline uint32_t lightArrayTag(const void* lights) {
    const unsigned long long address = reinterpret_cast<unsigned long long>(lights);
    return static_cast<uint32_t>(address ^ (address >> 32)) | 1U;
}

__host__ __device__ inline uint32_t lightTableCapacity(uint32_t lightCount) {
    uint32_t capacity = 2;
    while (capacity < 2U * lightCount && capacity < 0x80000000U)
        capacity <<= 1;
    return capacity;
}

__host__ __device__ inline size_t lightIndexWords(uint32_t lightCount) {
    const size_t head = 4 + 2 * static_cast<size_t>(lightCount);
    const size_t aligned = (head + 1) & ~static_cast<size_t>(1);
    return aligned + 3 * static_cast<size_t>(lightTableCapacity(lightCount));
}

// The parts of an index in memory laid out as above.
__host__ __device__ inline LightIndex lightIndexView(const void* memory, uint32_t lightCount, bool withThresholds) {
    LightIndex view;
    if (!memory || lightCount == 0)
        return view;
    const uint32_t* words = static_cast<const uint32_t*>(memory);
    const size_t head = 4 + 2 * static_cast<size_t>(lightCount);
    const size_t aligned = (head + 1) & ~static_cast<size_t>(1);
    const uint32_t capacity = lightTableCapacity(lightCount);
    view.thresholds = withThresholds ? reinterpret_cast<const float*>(words + 4) : nullptr;
    view.weights = reinterpret_cast<const float*>(words + 4 + lightCount);
    view.keys = reinterpret_cast<const unsigned long long*>(words + aligned);
    view.values = words + aligned + 2 * static_cast<size_t>(capacity);
    view.count = lightCount;
    view.mask = capacity - 1U;
    return view;
}

__host__ __device__ inline unsigned long long lightKey(uint32_t instance, uint32_t primitive) {
    return static_cast<unsigned long long>(instance) << 32 | primitive;
}

__host__ __device__ inline uint32_t lightSlot(unsigned long long key, uint32_t mask) {
    key ^= key >> 33;
    key *= 0xff51afd7ed558ccdULL;
    key ^= key >> 33;
    key *= 0xc4ceb9fe1a85ec53ULL;
    key ^= key >> 33;
    return static_cast<uint32_t>(key) & mask;
}

// The light the walk of device.cu:433-447 stops at for `target` (= a uniform number times the sum
// of the weights), by the thresholds.
__device__ inline uint32_t lightByThreshold(const LightIndex& index, float target) {
    uint32_t first = 0, last = index.count - 1U;  // the last light takes what no threshold covers
    while (first < last) {
        const uint32_t middle = (first + last) / 2U;
        if (target <= index.thresholds[middle])
            last = middle;
        else
            first = middle + 1U;
    }
    return first;
}

// Lowest index of an emissive-triangle light of (instance, primitive), or ~0u.
__device__ inline uint32_t lightOfTriangle(const LightIndex& index, uint32_t instance, uint32_t primitive) {
    const unsigned long long key = lightKey(instance, primitive);
    uint32_t slot = lightSlot(key, index.mask);
    for (uint32_t probes = 0; probes <= index.mask; ++probes) {
        const unsigned long long there = index.keys[slot];
        if (there == key)
            return index.values[slot];
        if (there == kNoLightKey)
            return ~0U;
        slot = (slot + 1U) & index.mask;
    }
    return ~0U;
}

}  // namespace rt
