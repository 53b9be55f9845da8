// Photon hash grid for gfx950: bucket keys, a stable radix sort of (key, index)
// pairs, bucket offsets from the key boundaries of the sorted list, a payload
// reorder staged through LDS, and the radius gather.
//
// Replaces src/photon_grid.cu:17-83 (count / scan / scatter with global
// atomics) and the gather loop of src/device.cu:722-766 of the reference.  The
// sort keeps photons of one bucket in input order, so the grid and every gather
// sum are reproducible.
#include "common.h"
#include "photon_device.h"
#include "radix_sort.h"

#include <mutex>

namespace rt {
namespace {

// positions: float3 at `stride` floats.
__global__ __launch_bounds__(256) void photonKeysKernel(const float* positions, uint32_t stride, uint32_t n,
                                                        float cellSize, uint32_t bucketCount, uint32_t* keys,
                                                        const uint32_t* nDevice) {
    n = itemCount(n, nDevice);
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n;
         i += static_cast<size_t>(gridDim.x) * blockDim.x) {
        const float* p = positions + i * stride;
        keys[i] = bucketOfPosition(p[0], p[1], p[2], cellSize, bucketCount);
    }
}

// offsets[b] = first sorted position whose key is >= b, for b in [0, bucketCount]:
// position i starts every bucket in (key[i-1], key[i]], with key[-1] = -1 and
// key[n] = bucketCount.  One streaming pass over the sorted keys; every entry
// of offsets is written exactly once.
//
// Which of the two offset kernels runs is a matter of the photon count (dense maps: this one; maps with far
// fewer photons than buckets: the search below).  With the count in device memory both are launched and
// each looks at the count itself: `when` > 0 runs this kernel only for n >= bucketCount / 8, `when` < 0 only
// for n below that, 0 always.
__device__ inline bool offsetsTurn(int when, uint32_t n, uint32_t bucketCount) {
    return when == 0 || (when > 0) == (n >= bucketCount / 8U);
}

__global__ __launch_bounds__(256) void photonOffsetsKernel(const uint32_t* sortedKeys, uint32_t n,
                                                           uint32_t bucketCount, uint32_t* offsets,
                                                           const uint32_t* nDevice, int when) {
    n = itemCount(n, nDevice);
    if (!offsetsTurn(when, n, bucketCount))
        return;
    const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i > n)
        return;
    const uint32_t first = i == 0 ? 0U : sortedKeys[i - 1U] + 1U;
    const uint32_t last = i == n ? bucketCount : sortedKeys[i];
    for (uint32_t bucket = first; bucket <= last; ++bucket)
        offsets[bucket] = static_cast<uint32_t>(i);
}

// The same table by binary search, one thread per bucket: for sparse maps (far
// fewer photons than buckets), where the boundary pass would leave long runs of
// buckets to single threads.
__global__ __launch_bounds__(256) void photonOffsetsSearchKernel(const uint32_t* sortedKeys, uint32_t n,
                                                                 uint32_t bucketCount, uint32_t* offsets,
                                                                 const uint32_t* nDevice, int when) {
    n = itemCount(n, nDevice);
    if (!offsetsTurn(when, n, bucketCount))
        return;
    const size_t b = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (b > bucketCount)
        return;
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        const uint32_t mid = lo + (hi - lo) / 2U;
        if (sortedKeys[mid] < b)
            lo = mid + 1U;
        else
            hi = mid;
    }
    offsets[b] = lo;
}

// sorted[i] = photons[order[i]].  A workgroup gathers 256 photons (36 bytes
// each, scattered) into LDS and streams the 9216 contiguous bytes out as
// 16-byte stores.
__global__ __launch_bounds__(256) void photonReorderKernel(const rt_photon* photons, const uint32_t* order,
                                                           uint32_t n, rt_photon* sorted, const uint32_t* nDevice) {
    __shared__ float4 stage[256 * 9 / 4];  // 9216 bytes
    float* flat = reinterpret_cast<float*>(stage);
    const size_t base = static_cast<size_t>(blockIdx.x) * 256;
    n = itemCount(n, nDevice);
    if (base >= n)
        return;  // the whole workgroup
    const size_t i = base + threadIdx.x;
    if (i < n) {
        const float* src = reinterpret_cast<const float*>(photons + order[i]);
        float* dst = flat + threadIdx.x * 9;
#pragma unroll
        for (int k = 0; k < 9; ++k)
            dst[k] = src[k];
    }
    __syncthreads();
    const uint32_t count = static_cast<uint32_t>(min(static_cast<size_t>(256), n - base));
    float* out = reinterpret_cast<float*>(sorted + base);
    if (count == 256) {
        // workgroup base offset 256 * 36 * blockIdx is a multiple of 16 bytes
        float4* out4 = reinterpret_cast<float4*>(out);
        for (uint32_t k = threadIdx.x; k < 256 * 9 / 4; k += 256)
            out4[k] = stage[k];
    } else {
        for (uint32_t k = threadIdx.x; k < count * 9; k += 256)
            out[k] = flat[k];
    }
}

// One bit per bucket, set where the bucket holds photons (bucket b is bit b & 31 of word b >> 5).
__global__ __launch_bounds__(256) void bucketBitmapKernel(const uint32_t* cellStart, uint32_t bucketCount,
                                                          uint32_t* bitmap) {
    const uint32_t bucket = blockIdx.x * 256U + threadIdx.x;
    const bool holds = bucket < bucketCount && cellStart[bucket + 1U] != cellStart[bucket];
    const unsigned long long votes = __ballot(holds);
    const uint32_t half = threadIdx.x & 32U;  // lanes 0 and 32 of a wave write one word each
    if ((threadIdx.x & 31U) == 0U && bucket < bucketCount)
        bitmap[bucket >> 5] = static_cast<uint32_t>(votes >> half);
}

// One query per lane.
__global__ __launch_bounds__(256, RT_GATHER_WAVES) void photonGatherKernel(const rt_photon* sorted, const uint32_t* cellStart,
                                                          uint32_t bucketCount, float radius, const float* qpos,
                                                          const float* qnormal, uint32_t n, float* flux,
                                                          unsigned long long* visitedTotal,
                                                          const uint32_t* occupied) {
    const size_t q = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    uint32_t read = 0;
    if (q < n) {
        float fx, fy, fz;
        gatherFlux(sorted, cellStart, bucketCount, radius, qpos[3 * q], qpos[3 * q + 1], qpos[3 * q + 2], qnormal[3 * q],
                   qnormal[3 * q + 1], qnormal[3 * q + 2], fx, fy, fz, visitedTotal ? &read : nullptr, occupied);
        flux[3 * q] = fx;
        flux[3 * q + 1] = fy;
        flux[3 * q + 2] = fz;
    }
    if (visitedTotal) {
        unsigned long long sum = read;
#pragma unroll
        for (int delta = 32; delta >= 1; delta >>= 1)
            sum += __shfl_down(sum, delta, 64);
        if ((threadIdx.x & 63) == 0)
            atomicAdd(visitedTotal, sum);
    }
}

struct GridScratch {
    uint32_t* keys[2];
    uint32_t* idx[2];
    uint32_t* table;
    uint32_t* partials;
};

constexpr size_t scratchBytes(uint32_t maxCount, uint32_t bucketCount) {
    (void)bucketCount;
    const size_t n = alignUp(static_cast<size_t>(maxCount) * sizeof(uint32_t), 256);
    const size_t table = alignUp(sortTableEntries(maxCount) * sizeof(uint32_t), 256);
    const size_t partials = alignUp(scanScratchEntries(sortTableEntries(maxCount)) * sizeof(uint32_t), 256);
    return 4 * n + table + partials + 256;
}

GridScratch carve(void* scratch, uint32_t maxCount) {
    auto base = reinterpret_cast<uintptr_t>(scratch);
    base = (base + 255) / 256 * 256;
    const size_t n = alignUp(static_cast<size_t>(maxCount) * sizeof(uint32_t), 256);
    const size_t table = alignUp(sortTableEntries(maxCount) * sizeof(uint32_t), 256);
    GridScratch s;
    s.keys[0] = reinterpret_cast<uint32_t*>(base);
    s.keys[1] = reinterpret_cast<uint32_t*>(base + n);
    s.idx[0] = reinterpret_cast<uint32_t*>(base + 2 * n);
    s.idx[1] = reinterpret_cast<uint32_t*>(base + 3 * n);
    s.table = reinterpret_cast<uint32_t*>(base + 4 * n);
    s.partials = reinterpret_cast<uint32_t*>(base + 4 * n + table);
    return s;
}

// Workspace the library owns, per device, grown on demand, for callers whose
// scratch buffer is smaller than the sort needs.
// One build uses it at a time: `busy` is recorded behind the last kernel of the build that holds the
// workspace, and the next build - on whatever stream - makes its stream wait for it first, so builds
// on different streams that fall back to this workspace run one after the other instead of sorting
// into each other's arrays.
struct Workspace {
    void* data = nullptr;
    size_t bytes = 0;
    hipEvent_t busy = nullptr;
    bool recorded = false;
};
std::mutex workspaceMutex;
Workspace workspaces[64];

// Called with workspaceMutex held.
rt_status ownedWorkspace(size_t bytes, hipStream_t stream, Workspace** out) {
    int device = 0;
    RT_HIP(hipGetDevice(&device));
    if (device < 0 || device >= 64)
        return failInvalid("device index");
    Workspace& w = workspaces[device];
    if (!w.busy)
        RT_HIP(hipEventCreateWithFlags(&w.busy, hipEventDisableTiming));
    if (w.bytes < bytes) {
        if (w.data) {
            RT_HIP(hipDeviceSynchronize());  // nothing may still be sorting in the block that is freed
            RT_HIP(hipFree(w.data));
            w.data = nullptr;
            w.bytes = 0;
        }
        RT_HIP(hipMalloc(&w.data, bytes));
        w.bytes = bytes;
    } else if (w.recorded) {
        RT_HIP(hipStreamWaitEvent(stream, w.busy, 0));
    }
    *out = &w;
    return RT_SUCCESS;
}

// Scratch of the entry point that allocates nothing: a file-scope device array sized
// for RT_PHOTON_GRID_STATIC_MAX_PHOTONS photons, one copy per GPU that loads the
// library.  Host code only ever uses the address hipGetSymbolAddress reports for the
// current device (the array's name is a host-side stand-in, not a device address).
constexpr size_t kStaticScratchBytes = scratchBytes(RT_PHOTON_GRID_STATIC_MAX_PHOTONS, 1U);
static __device__ __align__(256) unsigned char scratch[kStaticScratchBytes];

std::mutex staticScratchMutex;
void* staticScratchAddress[64];  // per device, filled by the first call on that device

rt_status staticScratch(void** out) {
    int device = 0;
    RT_HIP(hipGetDevice(&device));
    if (device < 0 || device >= 64)
        return failInvalid("device index");
    std::lock_guard<std::mutex> lock(staticScratchMutex);
    if (!staticScratchAddress[device]) {
        void* address = nullptr;
        RT_HIP(hipGetSymbolAddress(&address, HIP_SYMBOL(scratch)));
        if (!address)
            return failInvalid("the static photon-grid scratch has no device address");
        staticScratchAddress[device] = address;
    }
    *out = staticScratchAddress[device];
    return RT_SUCCESS;
}

int log2Exact(uint32_t v) {
    int bits = 0;
    while ((1U << bits) < v)
        ++bits;
    return bits;
}

// The keys of a build live at the front of its scratch (GridScratch::keys[0]); a caller that already
// knows a photon's bucket when it stores the photon writes them there itself (keysReady).
rt_status buildGrid(const rt_photon* photons, uint32_t count, float cellSize, uint32_t bucketCount, uint32_t* counts,
                    uint32_t* offsets, void* scan, size_t scanBytes, rt_photon* sorted, hipStream_t stream,
                    const uint32_t* countDevice = nullptr, bool keysReady = false) {
    (void)counts;  // clobbered upstream; nothing here needs it
    if (bucketCount == 0 || (bucketCount & (bucketCount - 1U)) != 0)
        return failInvalid("bucketCount must be a power of two");
    if (!offsets)
        return failInvalid("offsets is null");
    if (count == 0) {
        RT_HIP(hipMemsetAsync(offsets, 0, (static_cast<size_t>(bucketCount) + 1U) * sizeof(uint32_t), stream));
        return RT_SUCCESS;
    }
    if (!photons || !sorted)
        return failInvalid("photons / sorted is null");
    const size_t needed = scratchBytes(count, bucketCount);
    void* workspace = scan;
    Workspace* owned = nullptr;
    std::unique_lock<std::mutex> lock(workspaceMutex, std::defer_lock);
    if (!scan || scanBytes < needed) {
        if (keysReady)
            return failInvalid("keys were announced in a scratch buffer that is too small for the build");
        lock.lock();
        RT_TRY(ownedWorkspace(needed, stream, &owned));
        workspace = owned->data;
    }
    const GridScratch s = carve(workspace, count);

    if (!keysReady)
        hipLaunchKernelGGL(photonKeysKernel, dim3(stridedGrid(count, 256)), dim3(256), 0, stream,
                           reinterpret_cast<const float*>(photons), 9U, count, cellSize, bucketCount, s.keys[0], countDevice);
    RT_HIP(hipGetLastError());
    int result = 0;
    const int bits = log2Exact(bucketCount) > 0 ? log2Exact(bucketCount) : 1;
    RT_TRY(radixSortPairs<uint32_t>(s.keys, s.idx, count, bits, s.table, s.partials, stream, &result, countDevice));
    const bool dense = count >= bucketCount / 8U;
    if (countDevice || dense)
        hipLaunchKernelGGL(photonOffsetsKernel, dim3(divUp(static_cast<size_t>(count) + 1U, 256)), dim3(256), 0, stream,
                           s.keys[result], count, bucketCount, offsets, countDevice, countDevice ? 1 : 0);
    if (countDevice || !dense)
        hipLaunchKernelGGL(photonOffsetsSearchKernel, dim3(divUp(static_cast<size_t>(bucketCount) + 1U, 256)), dim3(256),
                           0, stream, s.keys[result], count, bucketCount, offsets, countDevice, countDevice ? -1 : 0);
    hipLaunchKernelGGL(photonReorderKernel, dim3(divUp(count, 256)), dim3(256), 0, stream, photons, s.idx[result],
                       count, sorted, countDevice);
    RT_HIP(hipGetLastError());
    if (owned) {
        RT_HIP(hipEventRecord(owned->busy, stream));
        owned->recorded = true;
    }
    return RT_SUCCESS;
}

}  // namespace
}  // namespace rt

extern "C" {

size_t rt_photon_grid_scratch_bytes(uint32_t maxCount, uint32_t bucketCount) {
    return rt::scratchBytes(maxCount, bucketCount);
}

size_t rt_photon_scan_storage(uint32_t bucketCount) {
    const uint64_t capacity = 2ULL * bucketCount;
    return rt::scratchBytes(static_cast<uint32_t>(capacity > 0xffffffffULL ? 0xffffffffULL : capacity), bucketCount);
}

rt_status rt_build_photon_grid_async(const rt_photon* photons, uint32_t count, float cellSize, uint32_t bucketCount,
                                     uint32_t* counts, uint32_t* offsets, void* scan, size_t scanBytes,
                                     rt_photon* sorted, void* stream) {
    return rt::buildGrid(photons, count, cellSize, bucketCount, counts, offsets, scan, scanBytes, sorted,
                         static_cast<hipStream_t>(stream));
}

rt_status rt_build_photon_grid(const rt_photon* photons, uint32_t count, float cellSize, uint32_t bucketCount,
                               uint32_t* counts, uint32_t* offsets, void* scan, size_t scanBytes, rt_photon* sorted) {
    RT_TRY(rt::buildGrid(photons, count, cellSize, bucketCount, counts, offsets, scan, scanBytes, sorted, nullptr));
    RT_HIP(hipDeviceSynchronize());
    return RT_SUCCESS;
}

rt_status rt_build_photon_grid_counted(const rt_photon* photons, const uint32_t* countOnDevice, uint32_t maxCount,
                                       float cellSize, uint32_t bucketCount, uint32_t* offsets, void* scan,
                                       size_t scanBytes, rt_photon* sorted, int keysReady, void* stream) {
    if (!countOnDevice)
        return rt::failInvalid("countOnDevice is null");
    return rt::buildGrid(photons, maxCount, cellSize, bucketCount, nullptr, offsets, scan, scanBytes, sorted,
                         static_cast<hipStream_t>(stream), countOnDevice, keysReady != 0);
}

uint32_t* rt_photon_grid_keys(void* scan, size_t scanBytes, uint32_t maxCount, uint32_t bucketCount) {
    if (!scan || maxCount == 0 || scanBytes < rt::scratchBytes(maxCount, bucketCount))
        return nullptr;
    return rt::carve(scan, maxCount).keys[0];
}

rt_status rt_build_photon_grid_static(const rt_photon* photons, uint32_t count, float cellSize, uint32_t bucketCount,
                                      uint32_t* offsets, rt_photon* sorted, void* stream) {
    if (count > RT_PHOTON_GRID_STATIC_MAX_PHOTONS) {
        rt::setLastError("rt_build_photon_grid_static: " + std::to_string(count) + " photons exceed the static scratch, "
                         "which holds RT_PHOTON_GRID_STATIC_MAX_PHOTONS = " +
                         std::to_string(RT_PHOTON_GRID_STATIC_MAX_PHOTONS) + "; use rt_build_photon_grid_async");
        return RT_ERROR_OUT_OF_RANGE;
    }
    void* workspace = nullptr;
    RT_TRY(rt::staticScratch(&workspace));
    // every word of the scratch the sort reads it has written first, so nothing of the last call's contents matters
    return rt::buildGrid(photons, count, cellSize, bucketCount, nullptr, offsets, workspace, rt::kStaticScratchBytes,
                         sorted, static_cast<hipStream_t>(stream));
}

rt_status rt_photon_keys(const float* positions, uint32_t n, float cellSize, uint32_t bucketCount, uint32_t* keys,
                         void* stream) {
    if (bucketCount == 0 || (bucketCount & (bucketCount - 1U)) != 0)
        return rt::failInvalid("bucketCount must be a power of two");
    if (n == 0)
        return RT_SUCCESS;
    if (!positions || !keys)
        return rt::failInvalid("positions / keys is null");
    hipLaunchKernelGGL(rt::photonKeysKernel, dim3(rt::stridedGrid(n, 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), positions, 3U, n, cellSize, bucketCount, keys,
                       static_cast<const uint32_t*>(nullptr));
    RT_HIP(hipGetLastError());
    return RT_SUCCESS;
}

rt_status rt_photon_gather(const rt_photon* sorted, const uint32_t* cellStart, uint32_t bucketCount, float radius,
                           const float* qpos, const float* qnormal, uint32_t n, float* flux, void* stream) {
    if (bucketCount == 0 || (bucketCount & (bucketCount - 1U)) != 0)
        return rt::failInvalid("bucketCount must be a power of two");
    if (n == 0)
        return RT_SUCCESS;
    if (!cellStart || !qpos || !qnormal || !flux)
        return rt::failInvalid("null pointer");
    hipLaunchKernelGGL(rt::photonGatherKernel, dim3(rt::divUp(n, 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), sorted, cellStart, bucketCount, radius, qpos, qnormal, n,
                       flux, static_cast<unsigned long long*>(nullptr), static_cast<const uint32_t*>(nullptr));
    RT_HIP(hipGetLastError());
    return RT_SUCCESS;
}

rt_status rt_photon_bucket_bitmap(const uint32_t* cellStart, uint32_t bucketCount, uint32_t* bitmap, void* stream) {
    if (bucketCount == 0 || (bucketCount & (bucketCount - 1U)) != 0)
        return rt::failInvalid("bucketCount must be a power of two");
    if (!cellStart || !bitmap)
        return rt::failInvalid("null pointer");
    hipLaunchKernelGGL(rt::bucketBitmapKernel, dim3(rt::divUp(bucketCount, 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), cellStart, bucketCount, bitmap);
    RT_HIP(hipGetLastError());
    return RT_SUCCESS;
}

rt_status rt_photon_gather_masked(const rt_photon* sorted, const uint32_t* cellStart, const uint32_t* bitmap,
                                  uint32_t bucketCount, float radius, const float* qpos, const float* qnormal,
                                  uint32_t n, float* flux, void* stream) {
    if (bucketCount == 0 || (bucketCount & (bucketCount - 1U)) != 0)
        return rt::failInvalid("bucketCount must be a power of two");
    if (n == 0)
        return RT_SUCCESS;
    if (!cellStart || !bitmap || !qpos || !qnormal || !flux)
        return rt::failInvalid("null pointer");
    hipLaunchKernelGGL(rt::photonGatherKernel, dim3(rt::divUp(n, 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), sorted, cellStart, bucketCount, radius, qpos, qnormal, n,
                       flux, static_cast<unsigned long long*>(nullptr), bitmap);
    RT_HIP(hipGetLastError());
    return RT_SUCCESS;
}

rt_status rt_photon_gather_counted(const rt_photon* sorted, const uint32_t* cellStart, uint32_t bucketCount,
                                   float radius, const float* qpos, const float* qnormal, uint32_t n, float* flux,
                                   uint64_t* visitedPhotons, void* stream) {
    if (bucketCount == 0 || (bucketCount & (bucketCount - 1U)) != 0)
        return rt::failInvalid("bucketCount must be a power of two");
    if (n == 0)
        return RT_SUCCESS;
    if (!cellStart || !qpos || !qnormal || !flux || !visitedPhotons)
        return rt::failInvalid("null pointer");
    hipLaunchKernelGGL(rt::photonGatherKernel, dim3(rt::divUp(n, 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), sorted, cellStart, bucketCount, radius, qpos, qnormal, n,
                       flux, reinterpret_cast<unsigned long long*>(visitedPhotons),
                       static_cast<const uint32_t*>(nullptr));
    RT_HIP(hipGetLastError());
    return RT_SUCCESS;
}

}  // extern "C"
