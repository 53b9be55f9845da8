// Device-side photon hash and radius gather, shared by the grid kernels and the
// shading kernels.  Restates src/photon_hash.cuh:10-24 and the gather loop of
// src/device.cu:729-762 of the reference.
#pragma once

#include "common.h"

namespace rt {

// float -> int32, saturating, NaN -> 0 (the conversion the reference's
// static_cast<int>(floorf(..)) performs on the GPU; spelled out because an
// out-of-range cast is undefined for the compiler).
__device__ inline int toInt(float v) {
    if (v != v)
        return 0;
    if (v >= 2147483648.0f)
        return 2147483647;
    if (v <= -2147483648.0f)
        return -2147483647 - 1;
    return static_cast<int>(v);
}

// src/photon_hash.cuh:10-24.  The division is the correctly rounded IEEE one.
__device__ inline uint32_t bucketOfCell(int x, int y, int z, uint32_t bucketCount) {
    const uint32_t hash = static_cast<uint32_t>(x) * 73856093U ^ static_cast<uint32_t>(y) * 19349663U ^
                          static_cast<uint32_t>(z) * 83492791U;
    return hash & (bucketCount - 1U);
}

__device__ inline uint32_t bucketOfPosition(float px, float py, float pz, float cellSize, uint32_t bucketCount) {
    return bucketOfCell(toInt(floorf(px / cellSize)), toInt(floorf(py / cellSize)), toInt(floorf(pz / cellSize)),
                        bucketCount);
}

#ifndef RT_GATHER_CHUNK
#define RT_GATHER_CHUNK 8
#endif
constexpr uint32_t kGatherChunk = RT_GATHER_CHUNK;
// waves per SIMD the per-lane gather kernels are compiled for
#ifndef RT_GATHER_WAVES
#define RT_GATHER_WAVES 8
#endif

// Flux sum of one query in the reference's own order (cells z, y, x from -1 to
// 1, a bucket reached twice counted once, photons in stored order), so the
// float sums match a scalar evaluation bit for bit.  `visitedPhotons`, when not
// null, receives the number of photons read (cells out of the sphere's reach
// are not read, see below).
// `occupied`, when not null, holds one bit per bucket, set where the bucket has
// photons: with a map much sparser than its table most look-ups end at that bit
// (the bits of a 2^22-bucket table are 512 KB and stay in the L2 cache, the
// table's 16 MB do not), and an empty bucket adds nothing to any sum.
// `budget`: the walk gives up (returns false, sums and count unusable) before it
// reads a range that takes the photons read past this number; the caller hands
// such a query to gatherFluxWave.
__device__ inline bool gatherFlux(const rt_photon* sorted, const uint32_t* cellStart, uint32_t bucketCount,
                                  float radius, float hx, float hy, float hz, float nx, float ny, float nz, float& fx,
                                  float& fy, float& fz, uint32_t* visitedPhotons,
                                  const uint32_t* occupied = nullptr, uint32_t budget = 0xffffffffU);

// The cell of a query and the neighbour cells (bit c of the result, c = x + 3 y + 9 z
// with offsets 0..2) a gather does not read.
__device__ inline uint32_t gatherPlan(uint32_t bucketCount, float radius, float hx, float hy, float hz, int& cx, int& cy,
                                      int& cz, const uint32_t* occupied) {
    cx = toInt(floorf(hx / radius));
    cy = toInt(floorf(hy / radius));
    cz = toInt(floorf(hz / radius));
    const float radiusSquared = radius * radius;
    uint32_t visited[27];
#pragma unroll
    for (int c = 0; c < 27; ++c) {
        const int dx = c % 3 - 1, dy = (c / 3) % 3 - 1, dz = c / 9 - 1;
        visited[c] = bucketOfCell(static_cast<int>(static_cast<uint32_t>(cx) + static_cast<uint32_t>(dx)),
                                  static_cast<int>(static_cast<uint32_t>(cy) + static_cast<uint32_t>(dy)),
                                  static_cast<int>(static_cast<uint32_t>(cz) + static_cast<uint32_t>(dz)), bucketCount);
    }
    // Cells the sphere cannot reach are not read: every photon in them fails the
    // distance test, so leaving them out changes no sum (for photon positions that
    // are numbers: a NaN position passes `distance^2 >= r^2 -> skip` and would turn
    // the sum into NaN wherever its bucket is read).  gap[a][d] is a lower
    // bound of the distance along axis a to the neighbour cell at offset d - 1,
    // shortened by a tolerance that covers the rounding of the cell assignment
    // (floor of an IEEE quotient) and of the products here; with coordinates too
    // large for that reasoning nothing is skipped.
    float gap[3][3];
    bool mayCull = true;
    {
        const float h[3] = {hx, hy, hz};
        const int cell[3] = {cx, cy, cz};
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float base = static_cast<float>(cell[a]);
            const float tolerance = fmaxf(fabsf(h[a]), radius) * 0x1p-19f;
            const float below = (h[a] - base * radius) - tolerance;           // to the cell at -1
            const float above = ((base + 1.0f) * radius - h[a]) - tolerance;  // to the cell at +1
            gap[a][0] = below > 0.0f ? below * below : 0.0f;
            gap[a][1] = 0.0f;
            gap[a][2] = above > 0.0f ? above * above : 0.0f;
            mayCull = mayCull && fabsf(base) < 0x1p22f && radius > 0.0f;
        }
    }
    const float reach = radiusSquared * (1.0f + 0x1p-16f);
    // a bucket reached twice is read where the reference reads it: at its first
    // cell, even when that cell is out of reach and a later one is not
    uint32_t skip = 0, repeatedLater = 0;
#pragma unroll
    for (int c = 0; c < 27; ++c) {
#pragma unroll
        for (int p = 0; p < c; ++p)
            if (visited[p] == visited[c]) {
                skip |= 1U << c;
                repeatedLater |= 1U << p;
            }
        // keeps the 351 comparisons in program order: left to itself the
        // scheduler evaluates them as one wide tree and runs out of registers
        asm volatile("" : "+v"(skip), "+v"(repeatedLater));
    }
#pragma unroll
    for (int c = 0; c < 27; ++c) {
        const int dx = c % 3, dy = (c / 3) % 3, dz = c / 9;
        const bool outOfReach = mayCull && gap[0][dx] + gap[1][dy] + gap[2][dz] > reach;
        if (outOfReach && !(repeatedLater >> c & 1U))
            skip |= 1U << c;
    }
    // the bucket bits of the cells that are left, nine independent loads at a time
    // (a cell already out of the walk reads word 0)
    if (occupied) {
#pragma unroll
        for (int group = 0; group < 27; group += 9) {
            uint32_t words[9];
#pragma unroll
            for (int k = 0; k < 9; ++k)
                words[k] = occupied[skip >> (group + k) & 1U ? 0U : visited[group + k] >> 5];
#pragma unroll
            for (int k = 0; k < 9; ++k)
                if (!(words[k] >> (visited[group + k] & 31U) & 1U))
                    skip |= 1U << (group + k);
        }
    }
    return skip;
}

__device__ inline bool gatherFlux(const rt_photon* sorted, const uint32_t* cellStart, uint32_t bucketCount,
                                  float radius, float hx, float hy, float hz, float nx, float ny, float nz, float& fx,
                                  float& fy, float& fz, uint32_t* visitedPhotons, const uint32_t* occupied,
                                  uint32_t budget) {
    int cx, cy, cz;
    const uint32_t skip = gatherPlan(bucketCount, radius, hx, hy, hz, cx, cy, cz, occupied);
    const float radiusSquared = radius * radius;
    fx = 0.0f;
    fy = 0.0f;
    fz = 0.0f;
    uint32_t read = 0;
    // The walk itself is a plain loop (the bucket of cell c is hashed again
    // rather than kept): unrolled 27 times the compiler hoists all the range
    // loads and the kernel's register use quadruples.  Without bucket bits every
    // lane steps through the 27 cells together.  With them each lane goes from
    // one of its occupied cells to its next: in a sparse map a lane has one or two
    // of them, not the same ones as its neighbours, and a wave that stepped
    // through all 27 would wait out a chain of dependent loads at nearly every
    // one for the sake of a few lanes; lanes that share their cells (a dense map,
    // neighbouring queries) still move together.
    uint32_t todo = ~skip & 0x7ffffffU;
#pragma unroll 1
    for (int step = 0; step < 27; ++step) {
        int c = step;
        if (occupied) {
            if (todo == 0U)
                break;
            c = __ffs(static_cast<int>(todo)) - 1;
            todo &= todo - 1U;
        } else if (skip >> c & 1U) {
            continue;
        }
        const int dx = c % 3 - 1, dy = (c / 3) % 3 - 1, dz = c / 9 - 1;
        const uint32_t bucket =
            bucketOfCell(static_cast<int>(static_cast<uint32_t>(cx) + static_cast<uint32_t>(dx)),
                         static_cast<int>(static_cast<uint32_t>(cy) + static_cast<uint32_t>(dy)),
                         static_cast<int>(static_cast<uint32_t>(cz) + static_cast<uint32_t>(dz)), bucketCount);
        const uint32_t begin = cellStart[bucket];
        const uint32_t end = cellStart[bucket + 1U];
        read += end - begin;
        if (read > budget)
            return false;
        // kGatherChunk photons at a time: their positions are loaded together (the
        // index is clamped into the range, so no load waits for a comparison), which
        // keeps several loads in flight where a photon-by-photon loop has one; the
        // rest of a photon is read only when it is within the radius, which few of a
        // neighbour cell's are.  The sums still take the photons in index order.
#pragma unroll 1
        for (uint32_t index = begin; index < end; index += kGatherChunk) {
            float distanceSquared[kGatherChunk];
#pragma unroll
            for (uint32_t k = 0; k < kGatherChunk; ++k) {
                const uint32_t at = index + k < end ? index + k : end - 1U;
                const float* p = reinterpret_cast<const float*>(sorted + at);
                const float ox = p[0] - hx, oy = p[1] - hy, oz = p[2] - hz;
                distanceSquared[k] = ox * ox + oy * oy + oz * oz;
            }
#pragma unroll
            for (uint32_t k = 0; k < kGatherChunk; ++k) {
                if (index + k >= end || distanceSquared[k] >= radiusSquared)
                    continue;
                const float* p = reinterpret_cast<const float*>(sorted + index + k);
                const float cosine = p[6] * nx + p[7] * ny + p[8] * nz;
                if (cosine < 0.7f)
                    continue;
                const float weight = 1.0f - distanceSquared[k] / radiusSquared;
                fx = fx + p[3] * weight;
                fy = fy + p[4] * weight;
                fz = fz + p[5] * weight;
            }
        }
    }
    if (visitedPhotons)
        *visitedPhotons = read;
    return true;
}

// gatherFlux for one query by a whole wave (every lane passes the same query and
// gets the same result): the lanes test 64 photons of a range at a time, and
// the photons inside the radius are then added in index order, as a single lane
// would add them - a long range costs one coalesced load per 64 photons instead
// of a chain of dependent loads in one lane.
__device__ inline void gatherFluxWave(const rt_photon* sorted, const uint32_t* cellStart, uint32_t bucketCount,
                                      float radius, float hx, float hy, float hz, float nx, float ny, float nz,
                                      float& fx, float& fy, float& fz, uint32_t& visitedPhotons,
                                      const uint32_t* occupied) {
    int cx, cy, cz;
    const uint32_t skip = gatherPlan(bucketCount, radius, hx, hy, hz, cx, cy, cz, occupied);
    const float radiusSquared = radius * radius;
    const uint32_t lane = threadIdx.x & 63U;
    fx = 0.0f;
    fy = 0.0f;
    fz = 0.0f;
    uint32_t read = 0;
#pragma unroll 1
    for (int c = 0; c < 27; ++c) {
        if (skip >> c & 1U)
            continue;
        const int dx = c % 3 - 1, dy = (c / 3) % 3 - 1, dz = c / 9 - 1;
        const uint32_t bucket =
            bucketOfCell(static_cast<int>(static_cast<uint32_t>(cx) + static_cast<uint32_t>(dx)),
                         static_cast<int>(static_cast<uint32_t>(cy) + static_cast<uint32_t>(dy)),
                         static_cast<int>(static_cast<uint32_t>(cz) + static_cast<uint32_t>(dz)), bucketCount);
        const uint32_t begin = cellStart[bucket];
        const uint32_t end = cellStart[bucket + 1U];
        read += end - begin;
#pragma unroll 1
        for (uint32_t base = begin; base < end; base += 64U) {
            const uint32_t index = base + lane;
            bool inside = false;
            float px = 0.0f, py = 0.0f, pz = 0.0f;
            if (index < end) {
                const float* p = reinterpret_cast<const float*>(sorted + index);
                const float ox = p[0] - hx, oy = p[1] - hy, oz = p[2] - hz;
                const float distanceSquared = ox * ox + oy * oy + oz * oz;
                const float cosine = p[6] * nx + p[7] * ny + p[8] * nz;
                if (!(distanceSquared >= radiusSquared || cosine < 0.7f)) {
                    const float weight = 1.0f - distanceSquared / radiusSquared;
                    px = p[3] * weight;
                    py = p[4] * weight;
                    pz = p[5] * weight;
                    inside = true;
                }
            }
            unsigned long long votes = __ballot(inside);
            while (votes) {
                const int from = __ffsll(static_cast<long long>(votes)) - 1;
                votes &= votes - 1ULL;
                fx = fx + __shfl(px, from, 64);
                fy = fy + __shfl(py, from, 64);
                fz = fz + __shfl(pz, from, 64);
            }
        }
    }
    visitedPhotons = read;
}

}  // namespace rt
