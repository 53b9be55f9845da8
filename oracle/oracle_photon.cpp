// CPU oracle: photon hash, hash-grid build, radius gather, RNG.
// Test infrastructure only (see oracle.h).  Compile with -ffp-contract=off.
#include "oracle.h"

#include <cmath>
#include <cstring>
#include <thread>
#include <vector>

namespace {

// float -> int32 with the conversion semantics of the GPU the reference runs
// on (saturating, NaN -> 0); a plain C++ cast is undefined out of range.
int32_t toInt(float v) {
    if (v != v)
        return 0;
    if (v >= 2147483648.0f)
        return INT32_MAX;
    if (v <= -2147483648.0f)
        return INT32_MIN;
    return static_cast<int32_t>(v);
}

// photon_hash.cuh:10-15
void cellOf(const float* p, float cellSize, int32_t* cell) {
    cell[0] = toInt(std::floor(p[0] / cellSize));
    cell[1] = toInt(std::floor(p[1] / cellSize));
    cell[2] = toInt(std::floor(p[2] / cellSize));
}

// photon_hash.cuh:17-24
uint32_t bucketOf(const int32_t* cell, uint32_t bucketCount) {
    const uint32_t hash = static_cast<uint32_t>(cell[0]) * 73856093U ^
                          static_cast<uint32_t>(cell[1]) * 19349663U ^
                          static_cast<uint32_t>(cell[2]) * 83492791U;
    return hash & (bucketCount - 1U);
}

// device.cu:722-766, one query
void gatherOne(const rt_photon* sorted, const uint32_t* cellStart, uint32_t bucketCount, float radius,
               const float* hp, const float* hn, float* out) {
    int32_t center[3];
    cellOf(hp, radius, center);                       // device.cu:729
    const float radiusSquared = radius * radius;      // device.cu:730
    float fx = 0.0f, fy = 0.0f, fz = 0.0f;
    uint32_t visited[27];
    uint32_t visitedCount = 0;
    for (int z = -1; z <= 1; ++z)
        for (int y = -1; y <= 1; ++y)
            for (int x = -1; x <= 1; ++x) {
                // int32 wraparound on the add, as the device's two's complement add
                const int32_t cell[3] = {
                    static_cast<int32_t>(static_cast<uint32_t>(center[0]) + static_cast<uint32_t>(x)),
                    static_cast<int32_t>(static_cast<uint32_t>(center[1]) + static_cast<uint32_t>(y)),
                    static_cast<int32_t>(static_cast<uint32_t>(center[2]) + static_cast<uint32_t>(z))};
                const uint32_t bucket = bucketOf(cell, bucketCount);
                bool seen = false;
                for (uint32_t i = 0; i < visitedCount; ++i)
                    seen |= visited[i] == bucket;     // device.cu:742-746
                if (seen)
                    continue;
                visited[visitedCount++] = bucket;
                const uint32_t end = cellStart[bucket + 1U];
                for (uint32_t index = cellStart[bucket]; index < end; ++index) {
                    const rt_photon& photon = sorted[index];
                    const float ox = photon.position.x - hp[0];
                    const float oy = photon.position.y - hp[1];
                    const float oz = photon.position.z - hp[2];
                    const float distanceSquared = ox * ox + oy * oy + oz * oz;
                    const float cosine = photon.normal.x * hn[0] + photon.normal.y * hn[1] + photon.normal.z * hn[2];
                    if (distanceSquared >= radiusSquared || cosine < 0.7f)   // device.cu:754-756
                        continue;
                    const float weight = 1.0f - distanceSquared / radiusSquared;
                    fx = fx + photon.power.x * weight;                       // device.cu:757-758
                    fy = fy + photon.power.y * weight;
                    fz = fz + photon.power.z * weight;
                }
            }
    out[0] = fx;
    out[1] = fy;
    out[2] = fz;
}

}  // namespace

extern "C" {

void orc_photon_cell(const float* position3, float cellSize, int32_t* cell3) {
    cellOf(position3, cellSize, cell3);
}

uint32_t orc_photon_bucket(const int32_t* cell3, uint32_t bucketCount) {
    return bucketOf(cell3, bucketCount);
}

void orc_photon_keys(const float* positions, uint32_t n, float cellSize, uint32_t bucketCount, uint32_t* keys) {
    for (uint32_t i = 0; i < n; ++i) {
        int32_t cell[3];
        cellOf(positions + 3 * static_cast<size_t>(i), cellSize, cell);
        keys[i] = bucketOf(cell, bucketCount);
    }
}

void orc_build_photon_grid(const rt_photon* photons, uint32_t count, float cellSize, uint32_t bucketCount,
                           uint32_t* offsets, rt_photon* sorted) {
    // photon_grid.cu:53-61: zero the counters, count photons per bucket
    std::vector<uint32_t> counts(static_cast<size_t>(bucketCount) + 1U, 0U);
    std::vector<uint32_t> keys(count);
    for (uint32_t i = 0; i < count; ++i) {
        int32_t cell[3];
        cellOf(&photons[i].position.x, cellSize, cell);
        keys[i] = bucketOf(cell, bucketCount);
        ++counts[keys[i]];
    }
    // photon_grid.cu:66-67: exclusive sum over bucketCount + 1 entries
    uint32_t running = 0;
    for (size_t b = 0; b <= bucketCount; ++b) {
        offsets[b] = running;
        running += counts[b];
    }
    // photon_grid.cu:73-78: scatter behind a per-bucket cursor, here in input order
    std::fill(counts.begin(), counts.end(), 0U);
    for (uint32_t i = 0; i < count; ++i)
        sorted[offsets[keys[i]] + counts[keys[i]]++] = photons[i];
}

void orc_photon_gather(const rt_photon* sorted, const uint32_t* cellStart, uint32_t bucketCount, float radius,
                       const float* qpos, const float* qnormal, uint32_t n, float* flux, int threads) {
    auto work = [&](uint32_t begin, uint32_t end) {
        for (uint32_t q = begin; q < end; ++q)
            gatherOne(sorted, cellStart, bucketCount, radius, qpos + 3 * static_cast<size_t>(q),
                      qnormal + 3 * static_cast<size_t>(q), flux + 3 * static_cast<size_t>(q));
    };
    if (threads <= 1) {
        work(0, n);
        return;
    }
    std::vector<std::thread> pool;
    const uint32_t chunk = (n + threads - 1) / threads;
    for (int t = 0; t < threads; ++t) {
        const uint32_t begin = std::min<uint64_t>(n, static_cast<uint64_t>(t) * chunk);
        const uint32_t end = std::min<uint64_t>(n, static_cast<uint64_t>(begin) + chunk);
        if (begin < end)
            pool.emplace_back(work, begin, end);
    }
    for (auto& t : pool)
        t.join();
}

// device.cu:220-227
uint32_t orc_seeded(uint32_t a, uint32_t b) {
    uint32_t value = a * 0x9e3779b9U ^ b * 0x85ebca6bU;
    value ^= value >> 16U;
    value *= 0x7feb352dU;
    value ^= value >> 15U;
    value *= 0x846ca68bU;
    return value ^ (value >> 16U);
}

// device.cu:229-233 (the uint32 -> float conversion rounds to nearest even, so
// the result can be exactly 1.0f)
float orc_random(uint32_t* state) {
    *state = *state * 747796405U + 2891336453U;
    const uint32_t word = ((*state >> ((*state >> 28U) + 4U)) ^ *state) * 277803737U;
    return static_cast<float>((word >> 22U) ^ word) * 0x1p-32f;
}

}  // extern "C"
