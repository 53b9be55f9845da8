// Stable LSD radix sort of (key, uint32 payload) pairs and an exclusive prefix
// sum, written for 64-lane wavefronts: 8-bit digits, three kernels per pass
// (per-workgroup digit histogram, one scan of the digit-major table, ranked
// scatter).  Ranks inside a wave come from ballot-built peer masks, the tile is
// put in sorted order in LDS first so that lanes store runs of consecutive
// addresses, and no kernel waits on another workgroup.
#pragma once

#include "common.h"

namespace rt {

constexpr int kSortBlock = 512;                       // threads per workgroup
constexpr int kSortWaves = kSortBlock / 64;           // 8
constexpr int kSortItems = 16;                        // keys per thread
constexpr int kSortTile = kSortBlock * kSortItems;    // 8192 keys per workgroup
constexpr int kSortWaveTile = 64 * kSortItems;        // 1024 keys per wave
constexpr int kRadix = 256;

constexpr int kScanBlock = 256;
constexpr int kScanItems = 16;
constexpr int kScanTile = kScanBlock * kScanItems;    // 4096 values per workgroup
constexpr int kScanMaxPartials = 1024 * 64;           // one workgroup scans the partial sums

constexpr unsigned sortBlocks(uint32_t n) { return n ? divUp(n, kSortTile) : 0U; }
constexpr size_t sortTableEntries(uint32_t n) { return static_cast<size_t>(kRadix) * sortBlocks(n); }
constexpr size_t scanScratchEntries(size_t n) { return (n + kScanTile - 1) / kScanTile + 1; }

// Exclusive scan of `value` across the workgroup; `total` is the workgroup sum.
template <int BLOCK>
__device__ inline uint32_t blockExclusiveScan(uint32_t value, uint32_t* waveSums, uint32_t& total) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    uint32_t inclusive = value;
#pragma unroll
    for (int delta = 1; delta < 64; delta <<= 1) {
        const uint32_t other = __shfl_up(inclusive, delta, 64);
        if (lane >= delta)
            inclusive += other;
    }
    if (lane == 63)
        waveSums[wave] = inclusive;
    __syncthreads();
    uint32_t base = 0, sum = 0;
#pragma unroll
    for (int w = 0; w < BLOCK / 64; ++w) {
        const uint32_t s = waveSums[w];
        if (w < wave)
            base += s;
        sum += s;
    }
    total = sum;
    __syncthreads();
    return base + inclusive - value;
}

// ---- exclusive prefix sum over uint32 -------------------------------------

static __global__ __launch_bounds__(kScanBlock) void scanReduceKernel(const uint32_t* in, size_t n, uint32_t* partials) {
    __shared__ uint32_t waveSums[kScanBlock / 64];
    const size_t base = static_cast<size_t>(blockIdx.x) * kScanTile;
    uint32_t sum = 0;
#pragma unroll
    for (int i = 0; i < kScanItems; ++i) {
        const size_t index = base + static_cast<size_t>(i) * kScanBlock + threadIdx.x;
        if (index < n)
            sum += in[index];
    }
    uint32_t total;
    blockExclusiveScan<kScanBlock>(sum, waveSums, total);
    if (threadIdx.x == 0)
        partials[blockIdx.x] = total;
}

// One workgroup turns the per-tile sums into exclusive prefixes in place, 1024 consecutive sums at a time (each
// thread one of them, so that the loads and stores of a round are whole lines) with the total so far carried from
// round to round.  (Each thread walking a run of its own instead read a line per value: 0.6 ms for the 32,768 sums
// of a scan over 2^26 values, now 0.05.)
static __global__ __launch_bounds__(1024) void scanPartialsKernel(uint32_t* partials, uint32_t count) {
    __shared__ uint32_t waveSums[1024 / 64];
    uint32_t carried = 0;
    for (uint32_t base = 0; base < count; base += 1024U) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t value = i < count ? partials[i] : 0U;
        uint32_t total;
        const uint32_t before = blockExclusiveScan<1024>(value, waveSums, total);
        if (i < count)
            partials[i] = carried + before;
        carried += total;
    }
}

static __global__ __launch_bounds__(kScanBlock) void scanApplyKernel(const uint32_t* in, uint32_t* out, size_t n,
                                                              const uint32_t* partials) {
    __shared__ uint32_t waveSums[kScanBlock / 64];
    // each thread owns kScanItems consecutive values
    const size_t base = static_cast<size_t>(blockIdx.x) * kScanTile + static_cast<size_t>(threadIdx.x) * kScanItems;
    uint32_t values[kScanItems];
    uint32_t sum = 0;
#pragma unroll
    for (int i = 0; i < kScanItems; ++i) {
        values[i] = base + i < n ? in[base + i] : 0U;
        sum += values[i];
    }
    uint32_t total;
    uint32_t running = blockExclusiveScan<kScanBlock>(sum, waveSums, total) + partials[blockIdx.x];
#pragma unroll
    for (int i = 0; i < kScanItems; ++i) {
        if (base + i < n)
            out[base + i] = running;
        running += values[i];
    }
}

// out[i] = sum of in[0..i).  in == out is allowed.  `partials` holds
// scanScratchEntries(n) uint32.
inline rt_status exclusiveScanU32(const uint32_t* in, uint32_t* out, size_t n, uint32_t* partials,
                                  hipStream_t stream) {
    if (n == 0)
        return RT_SUCCESS;
    const size_t tiles = (n + kScanTile - 1) / kScanTile;
    if (tiles > kScanMaxPartials)
        return failInvalid("exclusive scan: input too large");
    hipLaunchKernelGGL(scanReduceKernel, dim3(static_cast<unsigned>(tiles)), dim3(kScanBlock), 0, stream, in, n,
                       partials);
    hipLaunchKernelGGL(scanPartialsKernel, dim3(1), dim3(1024), 0, stream, partials, static_cast<uint32_t>(tiles));
    hipLaunchKernelGGL(scanApplyKernel, dim3(static_cast<unsigned>(tiles)), dim3(kScanBlock), 0, stream, in, out, n,
                       partials);
    RT_HIP(hipGetLastError());
    return RT_SUCCESS;
}

// ---- radix sort -----------------------------------------------------------

// The number of items a sort kernel works on: the host's `n`, or - when the count
// only exists in device memory (a list whose length the GPU has just produced) -
// the word at `nDevice`, cut at `n`.  The launch is then sized for `n`, and the
// workgroups beyond the real count find nothing to do.
__device__ inline uint32_t itemCount(uint32_t n, const uint32_t* nDevice) {
    return nDevice ? min(*nDevice, n) : n;
}

template <typename K>
__global__ __launch_bounds__(kSortBlock) void radixHistKernel(const K* keys, uint32_t n, int shift, uint32_t* table,
                                                              uint32_t numBlocks, const uint32_t* nDevice) {
    __shared__ uint32_t hist[kRadix];
    n = itemCount(n, nDevice);
    if (threadIdx.x < kRadix)
        hist[threadIdx.x] = 0;
    __syncthreads();
    const size_t base = static_cast<size_t>(blockIdx.x) * kSortTile;
#pragma unroll
    for (int i = 0; i < kSortItems; ++i) {
        const size_t index = base + static_cast<size_t>(i) * kSortBlock + threadIdx.x;
        if (index < n)
            atomicAdd(&hist[static_cast<uint32_t>(keys[index] >> shift) & (kRadix - 1)], 1U);
    }
    __syncthreads();
    if (threadIdx.x < kRadix)
        table[static_cast<size_t>(threadIdx.x) * numBlocks + blockIdx.x] = hist[threadIdx.x];
}

// `table` holds the exclusive scan of the digit-major histogram table.  With
// FIRST the payload of key i is i itself and idxIn is not read.
template <typename K, bool FIRST>
__global__ __launch_bounds__(kSortBlock) void radixScatterKernel(const K* keysIn, const uint32_t* idxIn, K* keysOut,
                                                                 uint32_t* idxOut, uint32_t n, int shift,
                                                                 const uint32_t* table, uint32_t numBlocks,
                                                                 const uint32_t* nDevice) {
    __shared__ uint32_t waveHist[kSortWaves][kRadix];
    __shared__ uint32_t digitBase[kRadix];
    __shared__ uint32_t waveSums[kSortWaves];
    __shared__ K sKeys[kSortTile];
    __shared__ uint32_t sIdx[kSortTile];

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const unsigned long long lanesBelow = (1ULL << lane) - 1ULL;
    const size_t tileBase = static_cast<size_t>(blockIdx.x) * kSortTile;
    n = itemCount(n, nDevice);
    if (tileBase >= n)
        return;  // the whole workgroup: a launch sized for more items than the device-side count
    const uint32_t tileCount = static_cast<uint32_t>(min(static_cast<size_t>(kSortTile), n - tileBase));

    for (int i = threadIdx.x; i < kSortWaves * kRadix; i += kSortBlock)
        (&waveHist[0][0])[i] = 0;
    __syncthreads();

    // Phase 1: every wave walks its 1024 consecutive keys 64 at a time and
    // ranks each key among the equal digits the wave has seen so far.
    volatile uint32_t* myHist = waveHist[wave];
    K key[kSortItems];
    uint32_t payload[kSortItems];
    uint32_t rank[kSortItems];
#pragma unroll
    for (int it = 0; it < kSortItems; ++it) {
        const uint32_t local = wave * kSortWaveTile + it * 64 + lane;
        const bool valid = local < tileCount;
        key[it] = valid ? keysIn[tileBase + local] : static_cast<K>(0);
        payload[it] = FIRST ? static_cast<uint32_t>(tileBase + local) : (valid ? idxIn[tileBase + local] : 0U);
    }
#pragma unroll
    for (int it = 0; it < kSortItems; ++it) {
        const uint32_t local = wave * kSortWaveTile + it * 64 + lane;
        const bool valid = local < tileCount;
        const uint32_t digit = static_cast<uint32_t>(key[it] >> shift) & (kRadix - 1);
        unsigned long long peers = __ballot(valid);
#pragma unroll
        for (int bit = 0; bit < 8; ++bit) {
            const bool set = (digit >> bit) & 1U;
            const unsigned long long vote = __ballot(set);
            peers &= set ? vote : ~vote;
        }
        const uint32_t before = myHist[digit];
        const uint32_t below = __popcll(peers & lanesBelow);
        rank[it] = before + below;
        if (valid && below == 0)
            myHist[digit] = before + __popcll(peers);
    }
    __syncthreads();

    // Phase 2: per digit, prefix over the waves; then prefix over the digits.
    uint32_t digitTotal = 0;
    if (threadIdx.x < kRadix) {
        uint32_t running = 0;
#pragma unroll
        for (int w = 0; w < kSortWaves; ++w) {
            const uint32_t c = waveHist[w][threadIdx.x];
            waveHist[w][threadIdx.x] = running;
            running += c;
        }
        digitTotal = running;
    }
    uint32_t blockTotal;
    const uint32_t localStart = blockExclusiveScan<kSortBlock>(digitTotal, waveSums, blockTotal);
    if (threadIdx.x < kRadix) {
        digitBase[threadIdx.x] = table[static_cast<size_t>(threadIdx.x) * numBlocks + blockIdx.x] - localStart;
#pragma unroll
        for (int w = 0; w < kSortWaves; ++w)
            waveHist[w][threadIdx.x] += localStart;
    }
    __syncthreads();

    // Phase 3: place the tile in sorted order in LDS.
#pragma unroll
    for (int it = 0; it < kSortItems; ++it) {
        const uint32_t local = wave * kSortWaveTile + it * 64 + lane;
        if (local < tileCount) {
            const uint32_t digit = static_cast<uint32_t>(key[it] >> shift) & (kRadix - 1);
            const uint32_t slot = waveHist[wave][digit] + rank[it];
            sKeys[slot] = key[it];
            sIdx[slot] = payload[it];
        }
    }
    __syncthreads();

    // Phase 4: stream the sorted tile out; equal digits are consecutive lanes.
#pragma unroll
    for (int it = 0; it < kSortItems; ++it) {
        const uint32_t slot = it * kSortBlock + threadIdx.x;
        if (slot < tileCount) {
            const K k = sKeys[slot];
            const uint32_t digit = static_cast<uint32_t>(k >> shift) & (kRadix - 1);
            const uint32_t dst = digitBase[digit] + slot;
            keysOut[dst] = k;
            idxOut[dst] = sIdx[slot];
        }
    }
}

// Sorts n (key, payload) pairs by the low `bits` bits of the key, stable.  The
// initial payload of key i is i.  keys[0] holds the input; the sorted keys and
// payloads end in keys[result] / idx[result] (result = number of passes & 1).
// `table` holds sortTableEntries(n) uint32, `partials`
// scanScratchEntries(sortTableEntries(n)).  With `nDevice` the number of pairs is
// min(*nDevice, n), read by the kernels when they run; the host never learns it.
template <typename K>
inline rt_status radixSortPairs(K* const keys[2], uint32_t* const idx[2], uint32_t n, int bits, uint32_t* table,
                                uint32_t* partials, hipStream_t stream, int* result,
                                const uint32_t* nDevice = nullptr) {
    *result = 0;
    if (n == 0)
        return RT_SUCCESS;
    const unsigned blocks = sortBlocks(n);
    const int passes = (bits + 7) / 8;
    for (int pass = 0; pass < passes; ++pass) {
        const int in = pass & 1, out = in ^ 1;
        const int shift = pass * 8;
        hipLaunchKernelGGL(radixHistKernel<K>, dim3(blocks), dim3(kSortBlock), 0, stream, keys[in], n, shift, table,
                           blocks, nDevice);
        RT_TRY(exclusiveScanU32(table, table, sortTableEntries(n), partials, stream));
        if (pass == 0)
            hipLaunchKernelGGL((radixScatterKernel<K, true>), dim3(blocks), dim3(kSortBlock), 0, stream, keys[in],
                               idx[in], keys[out], idx[out], n, shift, table, blocks, nDevice);
        else
            hipLaunchKernelGGL((radixScatterKernel<K, false>), dim3(blocks), dim3(kSortBlock), 0, stream, keys[in],
                               idx[in], keys[out], idx[out], n, shift, table, blocks, nDevice);
        RT_HIP(hipGetLastError());
    }
    *result = passes & 1;
    return RT_SUCCESS;
}

}  // namespace rt
