// Acceleration-structure build for gfx950.
//
// Bottom level (replaces the OptiX GAS build, renderer.cpp:343-396), all on the
// device: triangle bounds -> 63-bit Morton codes of the centroids -> stable radix
// sort -> binary radix tree (Karras 2012) -> bottom-up bounds (or, on request,
// a binary tree by locally-ordered clustering, section 3b) -> top-down
// collapse into 8-wide nodes, one level per launch, where the child with the
// largest surface area is opened first and subtrees of at most kMaxLeaf
// triangles become leaf children -> 8-bit quantised child boxes.  Node and
// triangle slots come from prefix sums over the level, so the layout is the
// same on every run and every GPU.
//
// Top level (replaces the IAS build, renderer.cpp:398-434): built on the host
// from the instances' transformed root boxes - instance counts are small - with
// the same node format.
#include "bvh.h"
#include "radix_sort.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <initializer_list>
#include <memory>
#include <system_error>
#include <thread>
#include <vector>

// Whether node slots left over by the collapse are spent on splitting leaves.
#ifndef RT_BLAS_SPLIT_LEAVES
#define RT_BLAS_SPLIT_LEAVES 1
#endif

namespace rt {
namespace {

// ---- device helpers -------------------------------------------------------

// Monotone map float -> uint32 so that integer atomics order floats.
__device__ inline uint32_t orderedBits(float v) {
    const uint32_t b = __float_as_uint(v);
    return (b & 0x80000000U) ? ~b : (b | 0x80000000U);
}
__host__ __device__ inline float fromOrderedBits(uint32_t b) {
    const uint32_t raw = (b & 0x80000000U) ? (b & 0x7fffffffU) : ~b;
    float v;
#if defined(__HIP_DEVICE_COMPILE__)
    v = __uint_as_float(raw);
#else
    std::memcpy(&v, &raw, sizeof(v));
#endif
    return v;
}

struct Box4 {  // 32 bytes: lo.xyz, pad, hi.xyz, pad
    float4 lo;
    float4 hi;
};

__device__ inline float halfArea(const Box4& b) {
    const float x = b.hi.x - b.lo.x, y = b.hi.y - b.lo.y, z = b.hi.z - b.lo.z;
    return x * y + y * z + z * x;
}

__device__ inline Box4 merge(const Box4& a, const Box4& b) {
    Box4 r;
    r.lo = make_float4(fminf(a.lo.x, b.lo.x), fminf(a.lo.y, b.lo.y), fminf(a.lo.z, b.lo.z), 0.0f);
    r.hi = make_float4(fmaxf(a.hi.x, b.hi.x), fmaxf(a.hi.y, b.hi.y), fmaxf(a.hi.z, b.hi.z), 0.0f);
    return r;
}

constexpr unsigned kBoundsBlocks = 8192;  // workgroups of the bounds pass: 32 per compute unit

__device__ inline float3 loadPosition(const uint8_t* vertices, uint32_t stride, uint32_t index) {
    const float* p = reinterpret_cast<const float*>(vertices + static_cast<size_t>(index) * stride);
    return make_float3(p[0], p[1], p[2]);
}

// ---- 0. input check ---------------------------------------------------------

// Sets *flag when an index does not name a vertex.  The build reads vertices
// through the indices, so a bad index has to be an error before it is a fault.
__global__ __launch_bounds__(256) void indexCheckKernel(const uint32_t* indices, size_t count, uint32_t vertexCount,
                                                        uint32_t* flag) {
    bool bad = false;
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < count;
         i += static_cast<size_t>(gridDim.x) * blockDim.x)
        bad = bad || indices[i] >= vertexCount;
    if (__ballot(bad) != 0ULL && (threadIdx.x & 63) == 0)
        atomicOr(flag, 1U);
}

// ---- 1. triangle bounds + centroid extent ---------------------------------

// centroidBounds: 6 ordered-bit words, lo.xyz then hi.xyz.
// Box of every triangle and the bounds of the box centres (ordered bits, min then max per axis).  A workgroup
// walks triangles a grid apart and reports its bounds once: atomics of many workgroups on the same six words are
// served one after the other, and one report per 256 triangles held a mesh of 10^8 triangles up for 10 ms.
__global__ __launch_bounds__(256) void triangleBoundsKernel(const uint8_t* vertices, uint32_t stride,
                                                            const uint32_t* indices, uint32_t n, Box4* boxes,
                                                            uint32_t* centroidBounds) {
    __shared__ uint32_t block[6];
    if (threadIdx.x < 3)
        block[threadIdx.x] = 0xffffffffU;
    else if (threadIdx.x < 6)
        block[threadIdx.x] = 0U;
    __syncthreads();
    uint32_t lo[3] = {0xffffffffU, 0xffffffffU, 0xffffffffU}, hi[3] = {0U, 0U, 0U};
    const size_t step = static_cast<size_t>(gridDim.x) * blockDim.x;
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += step) {
        const float3 a = loadPosition(vertices, stride, indices[3 * i]);
        const float3 b = loadPosition(vertices, stride, indices[3 * i + 1]);
        const float3 c = loadPosition(vertices, stride, indices[3 * i + 2]);
        Box4 box;
        box.lo = make_float4(fminf(fminf(a.x, b.x), c.x), fminf(fminf(a.y, b.y), c.y), fminf(fminf(a.z, b.z), c.z), 0.0f);
        box.hi = make_float4(fmaxf(fmaxf(a.x, b.x), c.x), fmaxf(fmaxf(a.y, b.y), c.y), fmaxf(fmaxf(a.z, b.z), c.z), 0.0f);
        boxes[i] = box;
        const uint32_t centre[3] = {orderedBits(0.5f * (box.lo.x + box.hi.x)), orderedBits(0.5f * (box.lo.y + box.hi.y)),
                                    orderedBits(0.5f * (box.lo.z + box.hi.z))};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            lo[k] = min(lo[k], centre[k]);
            hi[k] = max(hi[k], centre[k]);
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
#pragma unroll
        for (int delta = 32; delta >= 1; delta >>= 1) {
            lo[k] = min(lo[k], static_cast<uint32_t>(__shfl_xor(static_cast<int>(lo[k]), delta, 64)));
            hi[k] = max(hi[k], static_cast<uint32_t>(__shfl_xor(static_cast<int>(hi[k]), delta, 64)));
        }
        if ((threadIdx.x & 63U) == 0U) {
            atomicMin(&block[k], lo[k]);
            atomicMax(&block[3 + k], hi[k]);
        }
    }
    __syncthreads();
    if (threadIdx.x < 3)
        atomicMin(&centroidBounds[threadIdx.x], block[threadIdx.x]);
    else if (threadIdx.x < 6)
        atomicMax(&centroidBounds[threadIdx.x], block[threadIdx.x]);
}

// ---- 2. Morton codes -------------------------------------------------------

__device__ inline uint64_t spread21(uint64_t v) {
    v &= 0x1fffffULL;
    v = (v | v << 32) & 0x1f00000000ffffULL;
    v = (v | v << 16) & 0x1f0000ff0000ffULL;
    v = (v | v << 8) & 0x100f00f00f00f00fULL;
    v = (v | v << 4) & 0x10c30c30c30c30c3ULL;
    v = (v | v << 2) & 0x1249249249249249ULL;
    return v;
}

__global__ __launch_bounds__(256) void mortonKernel(const Box4* boxes, uint32_t n, const uint32_t* centroidBounds,
                                                    uint64_t* keys) {
    const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const Box4 box = boxes[i];
    float c[3] = {0.5f * (box.lo.x + box.hi.x), 0.5f * (box.lo.y + box.hi.y), 0.5f * (box.lo.z + box.hi.z)};
    uint64_t q[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float lo = fromOrderedBits(centroidBounds[a]);
        const float hi = fromOrderedBits(centroidBounds[3 + a]);
        const float extent = hi - lo;
        float u = extent > 0.0f ? (c[a] - lo) / extent : 0.0f;
        u = fminf(fmaxf(u, 0.0f), 1.0f);
        const uint32_t cell = static_cast<uint32_t>(u * 2097151.0f);
        q[a] = cell > 2097151U ? 2097151U : cell;
    }
    keys[i] = spread21(q[0]) << 2 | spread21(q[1]) << 1 | spread21(q[2]);
}

// ---- 3. binary radix tree ---------------------------------------------------
// Node ids: internal node i is id i (0 .. n-2), leaf k is id (n-1) + k.

__device__ inline int commonPrefix(const uint64_t* keys, int n, int i, int j) {
    if (j < 0 || j >= n)
        return -1;
    const uint64_t a = keys[i], b = keys[j];
    if (a == b)
        return 64 + __clz(static_cast<int>(static_cast<uint32_t>(i) ^ static_cast<uint32_t>(j)));
    return __clzll(static_cast<long long>(a ^ b));
}

__global__ __launch_bounds__(256) void radixTreeKernel(const uint64_t* keys, uint32_t n, uint32_t* left,
                                                       uint32_t* right, uint32_t* parent, uint32_t* first,
                                                       uint32_t* last) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int count = static_cast<int>(n);
    if (i >= count - 1)
        return;
    const int d = commonPrefix(keys, count, i, i + 1) - commonPrefix(keys, count, i, i - 1) >= 0 ? 1 : -1;
    const int deltaMin = commonPrefix(keys, count, i, i - d);
    int lmax = 2;
    while (commonPrefix(keys, count, i, i + lmax * d) > deltaMin)
        lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (commonPrefix(keys, count, i, i + (l + t) * d) > deltaMin)
            l += t;
    const int j = i + l * d;
    const int deltaNode = commonPrefix(keys, count, i, j);
    int s = 0;
    int t = l;
    do {
        t = (t + 1) / 2;
        if (commonPrefix(keys, count, i, i + (s + t) * d) > deltaNode)
            s += t;
    } while (t > 1);
    const int gamma = i + s * d + min(d, 0);
    const int lo = min(i, j), hi = max(i, j);
    const uint32_t l_id = lo == gamma ? (n - 1U) + gamma : static_cast<uint32_t>(gamma);
    const uint32_t r_id = hi == gamma + 1 ? (n - 1U) + gamma + 1 : static_cast<uint32_t>(gamma + 1);
    left[i] = l_id;
    right[i] = r_id;
    parent[l_id] = i;
    parent[r_id] = i;
    first[i] = lo;
    last[i] = hi;
    if (i == 0)
        parent[0] = kNoHit;
}

// bounds[(n-1) + k] = box of the k-th triangle in sorted order
__global__ __launch_bounds__(256) void leafBoundsKernel(const Box4* boxes, const uint32_t* order, uint32_t n,
                                                        Box4* bounds) {
    const size_t k = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (k < n)
        bounds[(n - 1U) + k] = boxes[order[k]];
}

// Bottom-up bounds, one launch per wave of ready nodes.  A node is ready when
// both children have their boxes; the second child to report at the parent's
// arrival counter puts the parent on the next launch's list.  Boxes written by
// one launch are read by the next, so the kernel boundary is the only ordering
// needed (no fences, whose agent-scope form writes back a whole L2 on this
// part: the single-launch version with fences took 5.4 ms for 871 k triangles).
// Which thread appends a node varies from run to run, the boxes do not.
//
// level 0 lists the parents both of whose children are leaves; counts[k] is the
// length of list k and is zeroed before the first launch.
// A slot of `list` for every thread of the workgroup with `wants` set: one atomic per workgroup on the list's
// length.  Atomics on one word queue up behind each other at some 11 ns apiece - one per lane was 18 of the
// 126 ms of a 100 M triangle build, one per wave still 1.8 of the 14 ms of a 10 M build - so the kernels below
// run workgroups of 1024 threads and each asks once per pass.  Every thread of the workgroup must call this
// (it holds barriers).  The order of a list does not enter the bounds.
constexpr int kRefitBlock = 1024;
__device__ inline uint32_t appendSlot(bool wants, uint32_t* length) {
    __shared__ uint32_t waveCount[kRefitBlock / 64];
    __shared__ uint32_t blockBase;
    const unsigned long long votes = __ballot(wants);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
    if (lane == 0)
        waveCount[wave] = static_cast<uint32_t>(__popcll(votes));
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0;
        for (int w = 0; w < waves; ++w)
            total += waveCount[w];
        blockBase = total ? atomicAdd(length, total) : 0U;
    }
    __syncthreads();
    uint32_t before = blockBase;
    for (int w = 0; w < wave; ++w)
        before += waveCount[w];
    const uint32_t slot = before + static_cast<uint32_t>(__popcll(votes & ((1ULL << lane) - 1ULL)));
    __syncthreads();  // the next call of this workgroup writes the counts again
    return slot;
}

// The start of the bounds pass, one thread per inner node: a leaf's box is there from the beginning, so a node's
// arrival counter starts at the number of its children that are leaves - a plain store where every leaf used to
// report with an atomic (ten million of them queue up for 1.8 ms) - and a node of two leaves goes on list 0.
__global__ __launch_bounds__(kRefitBlock) void refitSeedKernel(const uint32_t* left, const uint32_t* right, uint32_t n,
                                                       uint32_t* arrivals, uint32_t* list, uint32_t* counts) {
    const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    bool ready = false;
    if (i + 1U < n) {
        const uint32_t leaves = (left[i] >= n - 1U ? 1U : 0U) + (right[i] >= n - 1U ? 1U : 0U);
        arrivals[i] = leaves;
        ready = leaves == 2U;
    }
    const uint32_t slot = appendSlot(ready, &counts[0]);
    if (ready)
        list[slot] = static_cast<uint32_t>(i);
}

__global__ __launch_bounds__(kRefitBlock) void refitLevelKernel(const uint32_t* left, const uint32_t* right,
                                                        const uint32_t* parent, Box4* bounds, uint32_t* arrivals,
                                                        const uint32_t* listIn, uint32_t* listOut, uint32_t* counts,
                                                        uint32_t level) {
    const uint32_t ready = counts[level];
    // whole workgroups go round together: the append holds barriers
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    for (size_t base = static_cast<size_t>(blockIdx.x) * blockDim.x; base < ready; base += stride) {
        const size_t i = base + threadIdx.x;
        uint32_t up = kNoHit;
        bool next = false;
        if (i < ready) {
            const uint32_t node = listIn[i];
            bounds[node] = merge(bounds[left[node]], bounds[right[node]]);
            up = parent[node];
            next = up != kNoHit && atomicAdd(&arrivals[up], 1U) == 1U;
        }
        const uint32_t slot = appendSlot(next, &counts[level + 1U]);
        if (next)
            listOut[slot] = up;
    }
}

constexpr uint32_t kRefitMaxLevels = 256;  // far above any radix-tree height (64 key bits + 32 index bits)
constexpr uint32_t kRefitBatch = 16;       // launches between looks at the list length

// ---- 3b. agglomerative binary tree (parallel locally-ordered clustering) ----
//
// Alternative to the radix tree, after Meister and Bittner (2018): the clusters
// (first the triangles in Morton order) each look kPlocRadius places to either
// side for the neighbour whose union with them has the smallest surface area;
// pairs that chose each other merge into a new node, the array is compacted and
// the search repeats until one cluster is left.  The trees come out close to a
// surface-area-heuristic build.  Every choice is made with ties to the lower index and slots come from
// prefix sums, so the tree is the same on every run.
//
// Node ids as above (internal 0 .. n-2, leaf k = (n-1) + k); internal ids are
// handed out from the top down, so the last merge - the root - is id 0.

constexpr int kPlocRadius = 16;

__global__ __launch_bounds__(256) void plocNearestKernel(const uint32_t* clusters, uint32_t m, const Box4* bounds,
                                                         uint32_t* nearest) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m)
        return;
    const Box4 mine = bounds[clusters[i]];
    const uint32_t lo = i > static_cast<uint32_t>(kPlocRadius) ? i - kPlocRadius : 0U;
    const uint32_t hi = min(m - 1U, i + kPlocRadius);
    float best = INFINITY;
    uint32_t which = i;
    for (uint32_t j = lo; j <= hi; ++j) {
        if (j == i)
            continue;
        const float area = halfArea(merge(mine, bounds[clusters[j]]));
        if (area < best || which == i) {
            best = area;
            which = j;
        }
    }
    nearest[i] = which;
}

// keep[i]: cluster i has a slot in the next array (it survives, or it is the
// lower partner of a merging pair); merges[i]: it is the lower partner.  Both
// arrays end with a zero so that the exclusive scans yield the totals.
__global__ __launch_bounds__(256) void plocFlagKernel(const uint32_t* nearest, uint32_t m, uint32_t* keep,
                                                      uint32_t* merges) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > m)
        return;
    if (i == m) {
        keep[i] = 0;
        merges[i] = 0;
        return;
    }
    const uint32_t j = nearest[i];
    const bool mutual = j != i && nearest[j] == i;
    keep[i] = (!mutual || i < j) ? 1U : 0U;
    merges[i] = (mutual && i < j) ? 1U : 0U;
}

__device__ inline uint32_t leavesBelow(uint32_t id, uint32_t n, const uint32_t* counts) {
    return id >= n - 1U ? 1U : counts[id];
}

__global__ __launch_bounds__(256) void plocMergeKernel(const uint32_t* clusters, const uint32_t* nearest,
                                                       const uint32_t* keepOffsets, const uint32_t* mergeOffsets,
                                                       uint32_t m, uint32_t n, uint32_t idBase, Box4* bounds,
                                                       uint32_t* left, uint32_t* right, uint32_t* parent,
                                                       uint32_t* counts, uint32_t* clustersOut) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m)
        return;
    const uint32_t j = nearest[i];
    const bool mutual = j != i && nearest[j] == i;
    if (mutual && i > j)
        return;  // absorbed by its partner
    uint32_t id = clusters[i];
    if (mutual) {
        const uint32_t a = id, b = clusters[j];
        id = idBase + mergeOffsets[i];
        left[id] = a;
        right[id] = b;
        parent[a] = id;
        parent[b] = id;
        bounds[id] = merge(bounds[a], bounds[b]);
        counts[id] = leavesBelow(a, n, counts) + leavesBelow(b, n, counts);
    }
    clustersOut[keepOffsets[i]] = id;
}

// Position of the first leaf below `id` in a depth-first walk of the tree: the
// leaves of the left subtrees it lies to the right of, summed on the way up.
__device__ inline uint32_t firstLeafBelow(uint32_t id, uint32_t n, const uint32_t* left, const uint32_t* right,
                                          const uint32_t* parent, const uint32_t* counts) {
    uint32_t first = 0, cur = id;
    for (uint32_t p = parent[cur]; p != kNoHit; p = parent[cur]) {
        if (right[p] == cur)
            first += leavesBelow(left[p], n, counts);
        cur = p;
    }
    return first;
}

// Leaves are renumbered by their depth-first position, so that every subtree
// covers a contiguous range of the new triangle order (which the collapse
// relies on).  Pass 1: where each leaf goes.
__global__ __launch_bounds__(256) void plocLeafOrderKernel(uint32_t n, const uint32_t* left, const uint32_t* right,
                                                           const uint32_t* parent, const uint32_t* counts,
                                                           const uint32_t* order, const Box4* bounds,
                                                           uint32_t* leafPosition, uint32_t* orderOut, Box4* leafBoundsOut) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n)
        return;
    const uint32_t position = firstLeafBelow((n - 1U) + k, n, left, right, parent, counts);
    leafPosition[k] = position;
    orderOut[position] = order[k];
    leafBoundsOut[position] = bounds[(n - 1U) + k];
}

// Pass 2: leaf ranges of the internal nodes.
__global__ __launch_bounds__(256) void plocRangeKernel(uint32_t n, const uint32_t* left, const uint32_t* right,
                                                       const uint32_t* parent, const uint32_t* counts, uint32_t* first,
                                                       uint32_t* last) {
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= n - 1U)
        return;
    const uint32_t begin = firstLeafBelow(id, n, left, right, parent, counts);
    first[id] = begin;
    last[id] = begin + counts[id] - 1U;
}

// Pass 3, once every walk up has finished: child references in the new leaf numbering.
__global__ __launch_bounds__(256) void plocChildKernel(uint32_t n, uint32_t* left, uint32_t* right,
                                                       const uint32_t* leafPosition) {
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= n - 1U)
        return;
    const uint32_t l = left[id], r = right[id];
    if (l >= n - 1U)
        left[id] = (n - 1U) + leafPosition[l - (n - 1U)];
    if (r >= n - 1U)
        right[id] = (n - 1U) + leafPosition[r - (n - 1U)];
}

__global__ __launch_bounds__(256) void iotaKernel(uint32_t* out, uint32_t n, uint32_t base) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        out[i] = base + i;
}

// ---- 3c. surface-area refinement of the upper levels --------------------------
//
// The radix tree splits space where the Morton code says, which is blind to how the triangles
// are spread.  Its subtrees of at most `limit` triangles are therefore kept as they are
// ("clusters"), and the tree above them is rebuilt top-down with the binned surface-area
// heuristic (16 bins per axis, cost = box area x triangles on either side): the top of the tree
// is where a poor cut costs most, and with a few thousand clusters it is small enough to build
// on the host while the device holds the rest.  Leaves are then renumbered in depth-first order
// of the new tree (the kernels of section 3b), so that every subtree still covers a contiguous
// range of the triangle order, which the collapse relies on.

__device__ inline uint32_t trianglesBelow(uint32_t id, uint32_t n, const uint32_t* first, const uint32_t* last) {
    return id >= n - 1U ? 1U : last[id] - first[id] + 1U;
}

// marks[k] = 1 where the k-th sorted triangle is the first one of a cluster, roots[k] = that
// cluster's node; upper[i] = 1 for the internal nodes above the clusters.  Each array ends with a
// zero so that the exclusive scans yield the totals.
__global__ __launch_bounds__(256) void clusterMarkKernel(uint32_t n, uint32_t limit, const uint32_t* parent,
                                                         const uint32_t* first, const uint32_t* last, uint32_t* marks,
                                                         uint32_t* roots, uint32_t* upper, uint32_t* counts) {
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= 2U * n - 1U)
        return;
    const uint32_t below = trianglesBelow(id, n, first, last);
    if (id < n - 1U) {
        upper[id] = below > limit ? 1U : 0U;
        counts[id] = below;
    }
    if (id == n - 1U)
        upper[id] = 0U;  // trailing zero (upper has n entries)
    const uint32_t up = parent[id];
    if (below <= limit && up != kNoHit && trianglesBelow(up, n, first, last) > limit) {
        const uint32_t k = id >= n - 1U ? id - (n - 1U) : first[id];
        marks[k] = 1U;
        roots[k] = id;
    }
}

struct ClusterRecord {
    Box4 box;
    uint32_t node, triangles, pad[2];
};

__global__ __launch_bounds__(256) void clusterGatherKernel(uint32_t n, const uint32_t* marks, const uint32_t* slots,
                                                           const uint32_t* roots, const uint32_t* first,
                                                           const uint32_t* last, const Box4* bounds,
                                                           ClusterRecord* clusters) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n || slots[k + 1U] == slots[k])
        return;
    (void)marks;
    ClusterRecord record;
    record.node = roots[k];
    record.box = bounds[record.node];
    record.triangles = trianglesBelow(record.node, n, first, last);
    record.pad[0] = record.pad[1] = 0U;
    clusters[slots[k]] = record;
}

__global__ __launch_bounds__(256) void upperIdsKernel(uint32_t n, const uint32_t* slots, uint32_t* ids) {
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id < n - 1U && slots[id + 1U] != slots[id])
        ids[slots[id]] = id;
}

struct UpperNode {
    Box4 box;
    uint32_t id, left, right, triangles;
};

// Surface area x triangles of every radix-tree node above the clusters, in the order of `ids`: what the tree that
// is there costs, to hold against the cost of the rebuilt one.
__global__ __launch_bounds__(256) void upperCostKernel(const uint32_t* ids, uint32_t count, const uint32_t* first,
                                                       const uint32_t* last, const Box4* bounds, float* cost) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= count)
        return;
    const uint32_t id = ids[k];
    const Box4 b = bounds[id];
    const float x = b.hi.x - b.lo.x, y = b.hi.y - b.lo.y, z = b.hi.z - b.lo.z;
    cost[k] = (x * y + y * z + z * x) * static_cast<float>(last[id] - first[id] + 1U);
}

__global__ __launch_bounds__(256) void upperScatterKernel(const UpperNode* nodes, uint32_t count, uint32_t* left,
                                                          uint32_t* right, uint32_t* parent, Box4* bounds,
                                                          uint32_t* counts) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= count)
        return;
    const UpperNode node = nodes[k];
    left[node.id] = node.left;
    right[node.id] = node.right;
    parent[node.left] = node.id;
    parent[node.right] = node.id;
    bounds[node.id] = node.box;
    counts[node.id] = node.triangles;
    if (k == 0)
        parent[node.id] = kNoHit;  // the root
}

// ---- 4. collapse into wide nodes --------------------------------------------

struct Child {
    uint32_t id;     // radix-tree node id
    uint32_t first;  // first sorted triangle
    uint32_t count;  // triangles below
    float area;
};

__device__ inline Child makeChild(uint32_t id, uint32_t n, const uint32_t* first, const uint32_t* last,
                                  const Box4* bounds) {
    Child c;
    c.id = id;
    if (id >= n - 1U) {
        c.first = id - (n - 1U);
        c.count = 1;
    } else {
        c.first = first[id];
        c.count = last[id] - first[id] + 1U;
    }
    c.area = halfArea(bounds[id]);
    return c;
}

// Opens the children of radix-tree node `root` until eight are held or none can
// be opened; children stay in sorted-triangle order.
__device__ inline int gatherChildren(uint32_t root, uint32_t n, const uint32_t* left, const uint32_t* right,
                                     const uint32_t* first, const uint32_t* last, const Box4* bounds,
                                     Child* children) {
    int count;
    if (root >= n - 1U || last[root] - first[root] + 1U <= static_cast<uint32_t>(kMaxLeaf)) {
        // the whole tree is one leaf (only the root entry can get here)
        children[0] = makeChild(root, n, first, last, bounds);
        return 1;
    }
    children[0] = makeChild(left[root], n, first, last, bounds);
    children[1] = makeChild(right[root], n, first, last, bounds);
    count = 2;
    while (count < kWide) {
        int pick = -1;
        float best = -1.0f;
        for (int k = 0; k < count; ++k)
            if (children[k].count > static_cast<uint32_t>(kMaxLeaf) && children[k].area > best) {
                best = children[k].area;
                pick = k;
            }
        if (pick < 0)
            break;
        const uint32_t id = children[pick].id;
        for (int k = count; k > pick + 1; --k)
            children[k] = children[k - 1];
        children[pick] = makeChild(left[id], n, first, last, bounds);
        children[pick + 1] = makeChild(right[id], n, first, last, bounds);
        ++count;
    }
#if RT_BLAS_SPLIT_LEAVES
    // The traversal tests all eight slots of a node whether they are used or
    // not, so slots left over are spent on splitting the leaves: each part
    // gets a box of its own and a ray only meets the triangles whose box it hits.
    while (count < kWide) {
        int pick = -1;
        float best = -1.0f;
        for (int k = 0; k < count; ++k)
            if (children[k].count > 1U && children[k].count <= static_cast<uint32_t>(kMaxLeaf) &&
                children[k].area > best) {
                best = children[k].area;
                pick = k;
            }
        if (pick < 0)
            break;
        const uint32_t id = children[pick].id;
        for (int k = count; k > pick + 1; --k)
            children[k] = children[k - 1];
        children[pick] = makeChild(left[id], n, first, last, bounds);
        children[pick + 1] = makeChild(right[id], n, first, last, bounds);
        ++count;
    }
#endif
    return count;
}

// Pass A over one level: how many node and triangle slots each entry needs.
__global__ __launch_bounds__(128) void collapseCountKernel(const uint32_t* frontier, uint32_t entries, uint32_t n,
                                                           const uint32_t* left, const uint32_t* right,
                                                           const uint32_t* first, const uint32_t* last,
                                                           const Box4* bounds, uint32_t* nodeCounts,
                                                           uint32_t* triCounts, uint2* gathered) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e > entries)
        return;
    if (e == entries) {  // sentinel so the exclusive scan yields the totals
        nodeCounts[e] = 0;
        triCounts[e] = 0;
        return;
    }
    Child children[kWide];
    const int count = gatherChildren(frontier[e], n, left, right, first, last, bounds, children);
    uint32_t nodes = 0, tris = 0;
    for (int k = 0; k < count; ++k) {
        if (children[k].count > static_cast<uint32_t>(kMaxLeaf))
            ++nodes;
        else
            tris += children[k].count;
    }
    nodeCounts[e] = nodes;
    triCounts[e] = tris;
    // what was gathered, for pass B: (id, triangles of a leaf child or 0) and the first triangle, kNoHit ends the list
    static_assert(kMaxLeaf <= 7, "the triangle count of a leaf child travels in three bits");
    for (int k = 0; k < kWide; ++k) {
        uint2 record = make_uint2(kNoHit, 0U);
        if (k < count) {
            const uint32_t leaf = children[k].count <= static_cast<uint32_t>(kMaxLeaf) ? children[k].count : 0U;
            record = make_uint2(children[k].id | leaf << 29, children[k].first);
        }
        gathered[static_cast<size_t>(e) * kWide + k] = record;
    }
}

__device__ inline float scaleOf(uint8_t e) {
    return __uint_as_float(static_cast<uint32_t>(e) << 23);
}

// Grows the box by a few ulps of its largest coordinate, on every axis: the traversal's slab test then never
// rejects a triangle its own intersection test accepts.  The distance that test computes is a weighted sum of
// the corners' distances along the ray's main axis, so its rounding error is a few ulps of the triangle's size
// whatever the distance itself, and moves the hit point by as much along every axis; a margin taken from an
// axis' own coordinates would be nothing on the axis a flat triangle has no extent on (a floor at y = 0), and a
// query whose interval ends at a hit (tmax = t, or a second triangle at the same distance) could lose it.
__host__ __device__ inline float boxMargin(const float* lo, const float* hi) {
    float m = 0.0f;
    for (int a = 0; a < 3; ++a)
        m = fmaxf(m, fmaxf(fmaxf(fabsf(lo[a]), fabsf(hi[a])), hi[a] - lo[a]));
    return fmaxf(m * 0x1p-18f, 1e-30f);
}

__device__ inline void padBox(Box4& b) {
    float* lo = &b.lo.x;
    float* hi = &b.hi.x;
    const float pad = boxMargin(lo, hi);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        lo[a] -= pad;
        hi[a] += pad;
    }
}

// Pass B: writes the wide node of each entry, its leaf triangles and the next
// level's entries.  levelNodeBase = index of this level's first node;
// nextNodeBase = index the next level starts at; triBase likewise.
__global__ __launch_bounds__(128) void collapseEmitKernel(
    const uint32_t* frontier, uint32_t entries, uint32_t n, const uint32_t* left, const uint32_t* right,
    const uint32_t* first, const uint32_t* last, const Box4* bounds, const uint32_t* nodeOffsets,
    const uint32_t* triOffsets, uint32_t levelNodeBase, uint32_t nextNodeBase, uint32_t triBase,
    const uint32_t* order, WideNode* nodes, uint32_t* slotPrimitive, uint32_t* nextFrontier, Box4* rootBox,
    const uint2* gathered) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= entries)
        return;
    // the children pass A gathered for this entry
    Child children[kWide];
    int count = 0;
    for (int k = 0; k < kWide; ++k) {
        const uint2 record = gathered[static_cast<size_t>(e) * kWide + k];
        if (record.x == kNoHit)
            break;
        const uint32_t leaf = record.x >> 29;
        children[k].id = record.x & 0x1fffffffU;
        children[k].first = record.y;
        children[k].count = leaf ? leaf : static_cast<uint32_t>(kMaxLeaf) + 1U;  // pass B only asks: leaf of how many, or not
        children[k].area = 0.0f;
        count = k + 1;
    }

    Box4 boxes[kWide];
    Box4 all;
    for (int k = 0; k < count; ++k) {
        boxes[k] = bounds[children[k].id];
        padBox(boxes[k]);
        all = k == 0 ? boxes[0] : merge(all, boxes[k]);
    }
    if (rootBox && e == 0)
        *rootBox = all;

    WideNode node = {};
    const float lo[3] = {all.lo.x, all.lo.y, all.lo.z};
    const float hi[3] = {all.hi.x, all.hi.y, all.hi.z};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        node.origin[a] = lo[a];
        // smallest power of two with 255 steps covering the extent
        int ex;
        frexpf((hi[a] - lo[a]) / 255.0f, &ex);
        int biased = ex + 127;
        biased = biased < 1 ? 1 : (biased > 254 ? 254 : biased);
        while (biased < 254 && lo[a] + 255.0f * scaleOf(static_cast<uint8_t>(biased)) < hi[a])
            ++biased;
        node.exp[a] = static_cast<uint8_t>(biased);
    }
    const uint32_t childBase = nextNodeBase + nodeOffsets[e];
    const uint32_t primBase = triBase + triOffsets[e];
    node.childBase = childBase;
    node.primBase = primBase;
    uint32_t nodeSlot = 0, triSlot = 0;
    for (int k = 0; k < count; ++k) {
        const float* clo = &boxes[k].lo.x;
        const float* chi = &boxes[k].hi.x;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float scale = scaleOf(node.exp[a]);
            float ql = floorf((clo[a] - lo[a]) / scale);
            ql = fminf(fmaxf(ql, 0.0f), 255.0f);
            while (ql > 0.0f && lo[a] + ql * scale > clo[a])
                ql -= 1.0f;
            float qh = ceilf((chi[a] - lo[a]) / scale);
            qh = fminf(fmaxf(qh, 0.0f), 255.0f);
            while (qh < 255.0f && lo[a] + qh * scale < chi[a])
                qh += 1.0f;
            node.qlo[a][k] = static_cast<uint8_t>(ql);
            node.qhi[a][k] = static_cast<uint8_t>(qh);
        }
        if (children[k].count > static_cast<uint32_t>(kMaxLeaf)) {
            node.imask |= static_cast<uint8_t>(1U << k);
            node.meta[k] = static_cast<uint8_t>(nodeSlot);
            nextFrontier[nodeOffsets[e] + nodeSlot] = children[k].id;
            ++nodeSlot;
        } else {
            node.meta[k] = static_cast<uint8_t>(children[k].count << 5 | triSlot);
            // which triangle goes to which slot; packTrianglesKernel fetches the corners, one thread per slot
            for (uint32_t t = 0; t < children[k].count; ++t)
                slotPrimitive[primBase + triSlot + t] = order[children[k].first + t];
            triSlot += children[k].count;
        }
    }
    // empty slots: a box no ray enters (see WideNode)
    for (int k = count; k < kWide; ++k)
        for (int a = 0; a < 3; ++a) {
            node.qlo[a][k] = 255;
            node.qhi[a][k] = 0;
        }
    nodes[levelNodeBase + e] = node;
}

// The triangle records of the leaves, one thread per slot of the triangle array: the corners of the triangle
// the collapse assigned to the slot (a thread of the collapse has up to sixteen slots to fill; fetching their
// corners there, one dependent chain after the other, was a third of its time).
__global__ __launch_bounds__(256) void packTrianglesKernel(const uint32_t* slotPrimitive, uint32_t n, const uint8_t* vertices,
                                                           uint32_t stride, const uint32_t* indices,
                                                           PackedTriangle* triangles) {
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= n)
        return;
    const uint32_t prim = slotPrimitive[slot];
    const float3 a = loadPosition(vertices, stride, indices[3 * static_cast<size_t>(prim)]);
    const float3 b = loadPosition(vertices, stride, indices[3 * static_cast<size_t>(prim) + 1]);
    const float3 c = loadPosition(vertices, stride, indices[3 * static_cast<size_t>(prim) + 2]);
    PackedTriangle tri;
    tri.v0[0] = a.x; tri.v0[1] = a.y; tri.v0[2] = a.z; tri.primitive = prim;
    tri.v1[0] = b.x; tri.v1[1] = b.y; tri.v1[2] = b.z; tri.pad1 = 0;
    tri.v2[0] = c.x; tri.v2[1] = c.y; tri.v2[2] = c.z; tri.pad2 = 0;
    triangles[slot] = tri;
}

// ---- host side --------------------------------------------------------------

// Which binary tree the wide nodes are collapsed from: the Morton radix tree
// unless RT_BLAS_BUILDER=ploc asks for the clustering build.  (Measured on an
// MI355X the two trace within a few percent of each other on a dense mesh and
// the radix tree is ahead on triangle soups, so it is the default.)
enum class Builder { Clustering, RadixTree };
static Builder builderChoice() {
    const char* v = std::getenv("RT_BLAS_BUILDER");
    return v && std::strcmp(v, "ploc") == 0 ? Builder::Clustering : Builder::RadixTree;
}

struct DeviceBuffer {
    void* data = nullptr;
    bool borrowed = false;  // a part of another buffer (allocateTogether): not this one's to free
    ~DeviceBuffer() {
        if (data && !borrowed)
            (void)hipFree(data);
    }
    rt_status allocate(size_t bytes) {
        RT_HIP(hipMalloc(&data, bytes ? bytes : 1));
        return RT_SUCCESS;
    }
    template <typename T>
    T* as() const {
        return static_cast<T*>(data);
    }
};

// One allocation for several buffers (each part aligned to 256 bytes): a build asks for some twenty scratch
// arrays, and twenty trips to the allocator are a tenth of a ten-million-triangle build.
struct BufferPart {
    DeviceBuffer* buffer;
    size_t bytes;
};
static rt_status allocateTogether(DeviceBuffer& whole, std::initializer_list<BufferPart> parts) {
    size_t total = 0;
    for (const BufferPart& part : parts)
        total += (part.bytes + 255U) & ~static_cast<size_t>(255U);
    RT_TRY(whole.allocate(total));
    size_t offset = 0;
    for (const BufferPart& part : parts) {
        part.buffer->data = static_cast<uint8_t*>(whole.data) + offset;
        part.buffer->borrowed = true;
        offset += (part.bytes + 255U) & ~static_cast<size_t>(255U);
    }
    return RT_SUCCESS;
}


// ---- surface-area rebuild of the tree above the clusters (host) ----------------

// Whether the upper levels are rebuilt (RT_BLAS_REFINE=0 keeps the plain radix tree, for comparison).
static bool refineChoice() {
    const char* v = std::getenv("RT_BLAS_REFINE");
    return !(v && std::strcmp(v, "0") == 0);
}

struct HostBox {
    float lo[3], hi[3];
    void grow(const HostBox& o) {
        for (int a = 0; a < 3; ++a) {
            lo[a] = std::min(lo[a], o.lo[a]);
            hi[a] = std::max(hi[a], o.hi[a]);
        }
    }
    float area() const {
        const float x = hi[0] - lo[0], y = hi[1] - lo[1], z = hi[2] - lo[2];
        return x * y + y * z + z * x;
    }
    static HostBox empty() { return {{INFINITY, INFINITY, INFINITY}, {-INFINITY, -INFINITY, -INFINITY}}; }
};

// Binary tree over the clusters by top-down binned surface-area splits.  Nodes come out in
// pre-order (the root first), which is also the order the old upper nodes' ids are handed to
// them in; a child reference is either an index into `nodes` or, with kClusterBit, a cluster.
//
// A subtree over m clusters has m - 1 nodes, so the pre-order position of every node follows
// from the sizes of the ranges alone: the node of a range sits at `base`, its left subtree
// behind it, its right subtree behind that.  Subtrees therefore write disjoint parts of
// `nodes` and `order`, and the first levels hand their halves to threads of their own; the
// tree is the same whatever the number of threads.
struct UpperBuilder {
    static constexpr int kBins = 16;
    static constexpr uint32_t kClusterBit = 0x80000000U;
    static constexpr uint32_t kForkLevels = 3, kForkMinimum = 512;  // up to eight threads, none for small ranges
    struct Node {
        HostBox box;
        uint32_t left, right, triangles;
    };
    const std::vector<ClusterRecord>& clusters;
    std::vector<HostBox> boxes;
    std::vector<float> centre[3];
    std::vector<uint32_t> order, scratch;
    std::vector<Node> nodes;
    // A cut that leaves one side with less than this share of a range's clusters is replaced by the median
    // cut: a chain of lopsided cuts is what makes a tree deep, and the traversal stack is finite.
    static constexpr uint32_t kLopsided = 16;

    explicit UpperBuilder(const std::vector<ClusterRecord>& c)
        : clusters(c), boxes(c.size()), order(c.size()), scratch(c.size()), nodes(c.empty() ? 0 : c.size() - 1) {
        for (auto& axis : centre)
            axis.resize(c.size());
        for (size_t i = 0; i < c.size(); ++i) {
            boxes[i] = {{c[i].box.lo.x, c[i].box.lo.y, c[i].box.lo.z}, {c[i].box.hi.x, c[i].box.hi.y, c[i].box.hi.z}};
            for (int a = 0; a < 3; ++a)
                centre[a][i] = 0.5f * (boxes[i].lo[a] + boxes[i].hi[a]);
            order[i] = static_cast<uint32_t>(i);
        }
    }

    uint32_t build(uint32_t begin, uint32_t end) { return build(begin, end, 0U, 0U); }

    // Builds the subtree over order[begin, end), whose node goes to nodes[base], and returns its reference.
    uint32_t build(uint32_t begin, uint32_t end, uint32_t base, uint32_t level) {
        if (end - begin == 1U)
            return kClusterBit | order[begin];
        HostBox all = HostBox::empty();
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        uint32_t triangles = 0;
        for (uint32_t i = begin; i < end; ++i) {
            const uint32_t c = order[i];
            all.grow(boxes[c]);
            for (int a = 0; a < 3; ++a) {
                lo[a] = std::min(lo[a], centre[a][c]);
                hi[a] = std::max(hi[a], centre[a][c]);
            }
            triangles += clusters[c].triangles;
        }
        // the three axes are binned in one pass over the range
        HostBox binBox[3][kBins];
        uint32_t binCount[3][kBins] = {};
        float scale[3];
        bool usable[3];
        for (int a = 0; a < 3; ++a) {
            usable[a] = hi[a] - lo[a] > 0.0f;
            scale[a] = usable[a] ? static_cast<float>(kBins) / (hi[a] - lo[a]) : 0.0f;
            for (auto& b : binBox[a])
                b = HostBox::empty();
        }
        for (uint32_t i = begin; i < end; ++i) {
            const uint32_t c = order[i];
            for (int a = 0; a < 3; ++a)
                if (usable[a]) {
                    const int bin = binOf(centre[a][c], lo[a], scale[a]);
                    binBox[a][bin].grow(boxes[c]);
                    binCount[a][bin] += clusters[c].triangles;
                }
        }
        int bestAxis = -1, bestBin = 0;
        float bestCost = INFINITY;
        for (int axis = 0; axis < 3; ++axis) {
            if (!usable[axis])
                continue;
            float rightArea[kBins];
            uint32_t rightCount[kBins];
            HostBox run = HostBox::empty();
            uint32_t count = 0;
            for (int k = kBins - 1; k >= 1; --k) {
                run.grow(binBox[axis][k]);
                count += binCount[axis][k];
                rightArea[k] = run.area();
                rightCount[k] = count;
            }
            run = HostBox::empty();
            count = 0;
            for (int k = 1; k < kBins; ++k) {  // cut between bin k - 1 and bin k
                run.grow(binBox[axis][k - 1]);
                count += binCount[axis][k - 1];
                if (count == 0U || rightCount[k] == 0U)
                    continue;
                const float cost = run.area() * static_cast<float>(count) + rightArea[k] * static_cast<float>(rightCount[k]);
                if (cost < bestCost) {
                    bestCost = cost;
                    bestAxis = axis;
                    bestBin = k;
                }
            }
        }
        uint32_t middle = begin + (end - begin) / 2U;
        if (bestAxis >= 0) {
            // stable partition through the scratch array (this range's part of it)
            uint32_t left = begin, right = 0;
            for (uint32_t i = begin; i < end; ++i) {
                const uint32_t c = order[i];
                if (binOf(centre[bestAxis][c], lo[bestAxis], scale[bestAxis]) < bestBin)
                    order[left++] = c;
                else
                    scratch[begin + right++] = c;
            }
            const uint32_t size = end - begin, smaller = std::min(left - begin, right);
            if (smaller != 0U && static_cast<uint64_t>(smaller) * kLopsided >= size) {
                std::copy(scratch.begin() + begin, scratch.begin() + begin + right, order.begin() + left);
                middle = left;
            } else {
                // the median of the range in the order it has now (left part, then right part)
                std::copy(scratch.begin() + begin, scratch.begin() + begin + right, order.begin() + left);
            }
        }
        nodes[base].box = all;
        nodes[base].triangles = triangles;
        const uint32_t leftBase = base + 1U, rightBase = base + (middle - begin);
        uint32_t l, r;
        bool forked = false;
        if (level < kForkLevels && end - begin >= kForkMinimum) {
            try {
                std::thread other([&] { l = build(begin, middle, leftBase, level + 1U); });
                forked = true;
                r = build(middle, end, rightBase, level + 1U);
                other.join();
            } catch (const std::system_error&) {
                // no thread to be had: this one does both halves
            }
        }
        if (!forked) {
            l = build(begin, middle, leftBase, level + 1U);
            r = build(middle, end, rightBase, level + 1U);
        }
        nodes[base].left = l;
        nodes[base].right = r;
        return base;
    }

    static int binOf(float c, float lo, float scale) {
        const int bin = static_cast<int>((c - lo) * scale);
        return bin < 0 ? 0 : (bin >= kBins ? kBins - 1 : bin);
    }
};

// Rebuilds the tree above the radix tree's subtrees of at most `limit` triangles by surface area, in two steps
// so that the host's share runs beside the GPU: start() cuts the clusters out on the device, brings their boxes
// to the host and hands them to a thread that builds the new upper tree; the caller meanwhile collapses the
// radix tree as it is.  finish() waits for the thread and puts the rebuilt tree in place only where it is the
// cheaper one by the measure it was built for (surface area x triangles over its nodes, summed in a fixed order
// so that every run decides alike) - the caller then collapses again.  On triangles spread evenly the radix
// tree's median cuts are the surface-area cuts already: nothing is replaced, and the probe has cost the build
// its device part only.  RT_BLAS_REFINE=always skips the comparison.
struct UpperRefinement {
    uint32_t n;
    hipStream_t stream;
    DeviceBuffer &left, &right, &parent, &first, &last, &bounds, &counts, &partials;
    DeviceBuffer marks, roots, upper, upperIds, clusterRecords, upperCost;
    std::vector<ClusterRecord> clusters;
    std::vector<uint32_t> ids;
    std::vector<float> oldCost;
    uint32_t clusterCount = 0, upperCount = 0;
    std::unique_ptr<UpperBuilder> builder;
    std::thread worker;
    bool started = false;
    // RT_BLAS_TIMING=1 prints where the time of the steps goes (host clock, the stream waited for at each mark)
    const bool timing = std::getenv("RT_BLAS_TIMING") != nullptr;
    std::chrono::steady_clock::time_point clock = std::chrono::steady_clock::now();

    UpperRefinement(uint32_t count, hipStream_t s, DeviceBuffer& l, DeviceBuffer& r, DeviceBuffer& p, DeviceBuffer& f,
                    DeviceBuffer& la, DeviceBuffer& b, DeviceBuffer& c, DeviceBuffer& scan)
        : n(count), stream(s), left(l), right(r), parent(p), first(f), last(la), bounds(b), counts(c), partials(scan) {}
    ~UpperRefinement() {
        if (worker.joinable())
            worker.join();
    }

    void mark(const char* what) {
        if (!timing)
            return;
        (void)hipStreamSynchronize(stream);
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "  refine %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - clock).count());
        clock = now;
    }

    static bool always() {
        const char* choice = std::getenv("RT_BLAS_REFINE");
        return choice && std::strcmp(choice, "always") == 0;
    }

    rt_status start() {
        clock = std::chrono::steady_clock::now();
        // about this many clusters (RT_BLAS_CLUSTERS overrides): the host's share of the build grows with it
        uint32_t targetClusters = 8192;
        if (const char* v = std::getenv("RT_BLAS_CLUSTERS"))
            targetClusters = static_cast<uint32_t>(std::max(2L, std::atol(v)));
        const uint32_t limit = std::max(4U, n / targetClusters);
        const unsigned idBlocks = divUp(2ULL * n - 1U, 256), blocks = divUp(n, 256);
        RT_TRY(marks.allocate(sizeof(uint32_t) * (static_cast<size_t>(n) + 1)));
        RT_TRY(roots.allocate(sizeof(uint32_t) * n));
        RT_TRY(upper.allocate(sizeof(uint32_t) * n));
        RT_HIP(hipMemsetAsync(marks.data, 0, sizeof(uint32_t) * (static_cast<size_t>(n) + 1), stream));
        hipLaunchKernelGGL(clusterMarkKernel, dim3(idBlocks), dim3(256), 0, stream, n, limit, parent.as<uint32_t>(),
                           first.as<uint32_t>(), last.as<uint32_t>(), marks.as<uint32_t>(), roots.as<uint32_t>(),
                           upper.as<uint32_t>(), counts.as<uint32_t>());
        RT_TRY(exclusiveScanU32(marks.as<uint32_t>(), marks.as<uint32_t>(), static_cast<size_t>(n) + 1, partials.as<uint32_t>(),
                                stream));
        RT_TRY(exclusiveScanU32(upper.as<uint32_t>(), upper.as<uint32_t>(), n, partials.as<uint32_t>(), stream));
        RT_HIP(hipMemcpyAsync(&clusterCount, marks.as<uint32_t>() + n, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        RT_HIP(hipMemcpyAsync(&upperCount, upper.as<uint32_t>() + (n - 1U), sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        RT_HIP(hipStreamSynchronize(stream));
        mark("cluster cut + counts");
        if (clusterCount < 2U)
            return RT_SUCCESS;  // the whole tree is one cluster
        if (upperCount + 1U != clusterCount)
            return failInvalid("cluster cut is not a tree");
        RT_TRY(clusterRecords.allocate(sizeof(ClusterRecord) * clusterCount));
        RT_TRY(upperIds.allocate(sizeof(uint32_t) * upperCount));
        hipLaunchKernelGGL(clusterGatherKernel, dim3(blocks), dim3(256), 0, stream, n, marks.as<uint32_t>(), marks.as<uint32_t>(),
                           roots.as<uint32_t>(), first.as<uint32_t>(), last.as<uint32_t>(), bounds.as<Box4>(),
                           clusterRecords.as<ClusterRecord>());
        hipLaunchKernelGGL(upperIdsKernel, dim3(blocks), dim3(256), 0, stream, n, upper.as<uint32_t>(), upperIds.as<uint32_t>());
        RT_TRY(upperCost.allocate(sizeof(float) * upperCount));
        hipLaunchKernelGGL(upperCostKernel, dim3(divUp(upperCount, 256)), dim3(256), 0, stream, upperIds.as<uint32_t>(), upperCount,
                           first.as<uint32_t>(), last.as<uint32_t>(), bounds.as<Box4>(), upperCost.as<float>());
        clusters.resize(clusterCount);
        ids.resize(upperCount);
        oldCost.resize(upperCount);
        RT_HIP(hipMemcpyAsync(oldCost.data(), upperCost.data, sizeof(float) * upperCount, hipMemcpyDeviceToHost, stream));
        RT_HIP(hipMemcpyAsync(clusters.data(), clusterRecords.data, sizeof(ClusterRecord) * clusterCount, hipMemcpyDeviceToHost,
                              stream));
        RT_HIP(hipMemcpyAsync(ids.data(), upperIds.data, sizeof(uint32_t) * upperCount, hipMemcpyDeviceToHost, stream));
        RT_HIP(hipStreamSynchronize(stream));
        if (ids.empty() || ids[0] != 0U)
            return failInvalid("the root is not among the upper nodes");
        mark("clusters to the host");
        builder = std::make_unique<UpperBuilder>(clusters);
        try {
            worker = std::thread([this] { builder->build(0, clusterCount); });
        } catch (const std::system_error&) {
            builder->build(0, clusterCount);  // no thread to be had: the rebuild runs here, ahead of the collapse
        }
        started = true;
        return RT_SUCCESS;
    }

    // `replaced` comes back true when the rebuilt upper tree was put in place; `order` then points at the new
    // triangle order (in spareOrder).
    rt_status finish(const uint32_t*& order, uint32_t* spareOrder, bool& replaced) {
        replaced = false;
        if (!started)
            return RT_SUCCESS;
        clock = std::chrono::steady_clock::now();
        worker.join();
        if (builder->nodes.size() != upperCount)
            return failInvalid("upper tree has the wrong size");
        mark("waiting for the surface-area build");
        double before = 0.0, after = 0.0;
        for (float c : oldCost)
            before += c;
        for (const UpperBuilder::Node& node : builder->nodes)
            after += static_cast<double>(node.box.area()) * node.triangles;
        if (timing)
            std::fprintf(stderr, "  refine cost of the upper tree: radix %.6g, rebuilt %.6g (%.4f)\n", before, after,
                         before > 0.0 ? after / before : 0.0);
        if (!always() && !(after < before * 0.98))
            return RT_SUCCESS;
        auto reference = [&](uint32_t r) {
            return (r & UpperBuilder::kClusterBit) ? clusters[r & ~UpperBuilder::kClusterBit].node : ids[r];
        };
        std::vector<UpperNode> scatter(upperCount);
        for (uint32_t k = 0; k < upperCount; ++k) {
            const UpperBuilder::Node& node = builder->nodes[k];
            UpperNode& out = scatter[k];
            out.box.lo = make_float4(node.box.lo[0], node.box.lo[1], node.box.lo[2], 0.0f);
            out.box.hi = make_float4(node.box.hi[0], node.box.hi[1], node.box.hi[2], 0.0f);
            out.id = ids[k];  // pre-order: node 0, the root, keeps id 0
            out.left = reference(node.left);
            out.right = reference(node.right);
            out.triangles = node.triangles;
        }
        const unsigned blocks = divUp(n, 256);
        DeviceBuffer upperNodes, leafPosition, leafBounds;
        RT_TRY(upperNodes.allocate(sizeof(UpperNode) * upperCount));
        RT_HIP(hipMemcpyAsync(upperNodes.data, scatter.data(), sizeof(UpperNode) * upperCount, hipMemcpyHostToDevice, stream));
        hipLaunchKernelGGL(upperScatterKernel, dim3(divUp(upperCount, 256)), dim3(256), 0, stream, upperNodes.as<UpperNode>(),
                           upperCount, left.as<uint32_t>(), right.as<uint32_t>(), parent.as<uint32_t>(), bounds.as<Box4>(),
                           counts.as<uint32_t>());
        // leaves in depth-first order of the new tree
        RT_TRY(leafPosition.allocate(sizeof(uint32_t) * n));
        RT_TRY(leafBounds.allocate(sizeof(Box4) * n));
        hipLaunchKernelGGL(plocLeafOrderKernel, dim3(blocks), dim3(256), 0, stream, n, left.as<uint32_t>(), right.as<uint32_t>(),
                           parent.as<uint32_t>(), counts.as<uint32_t>(), order, bounds.as<Box4>(), leafPosition.as<uint32_t>(),
                           spareOrder, leafBounds.as<Box4>());
        hipLaunchKernelGGL(plocRangeKernel, dim3(divUp(n - 1, 256)), dim3(256), 0, stream, n, left.as<uint32_t>(),
                           right.as<uint32_t>(), parent.as<uint32_t>(), counts.as<uint32_t>(), first.as<uint32_t>(),
                           last.as<uint32_t>());
        hipLaunchKernelGGL(plocChildKernel, dim3(divUp(n - 1, 256)), dim3(256), 0, stream, n, left.as<uint32_t>(),
                           right.as<uint32_t>(), leafPosition.as<uint32_t>());
        RT_HIP(hipMemcpyAsync(bounds.as<Box4>() + (n - 1U), leafBounds.data, sizeof(Box4) * n, hipMemcpyDeviceToDevice, stream));
        RT_HIP(hipGetLastError());
        RT_HIP(hipStreamSynchronize(stream));  // the scratch buffers of this function go out of scope
        mark("scatter + leaf order");
        order = spareOrder;
        replaced = true;
        return RT_SUCCESS;
    }
};

rt_status buildBlas(const void* vertices, uint32_t vertexCount, uint32_t stride, const uint32_t* indices, uint32_t n,
                    hipStream_t stream, rt_blas* blas, bool allowRefine = true) {
    // the two events of the build's own time; released on every way out
    struct Events {
        hipEvent_t start = nullptr, stop = nullptr;
        ~Events() {
            if (start)
                (void)hipEventDestroy(start);
            if (stop)
                (void)hipEventDestroy(stop);
        }
    } events;
    RT_HIP(hipEventCreate(&events.start));
    RT_HIP(hipEventCreate(&events.stop));
    const hipEvent_t start = events.start, stop = events.stop;
    RT_HIP(hipEventRecord(start, stream));

    // RT_BLAS_TIMING=1 prints where the time of a build goes (host clock, the stream waited for at each mark)
    const bool timing = std::getenv("RT_BLAS_TIMING") != nullptr;
    auto phaseClock = std::chrono::steady_clock::now();
    auto phase = [&](const char* what) {
        if (!timing)
            return;
        (void)hipStreamSynchronize(stream);
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "  build  %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - phaseClock).count());
        phaseClock = now;
    };
    const uint32_t ids = 2U * n - 1U;
    const size_t maxNodes = static_cast<size_t>(0.35 * n) + 16;
    DeviceBuffer scratch, boxes, cbounds, keysA, keysB, idxA, idxB, table, partials, left, right, parent, first, last, bounds,
        arrivals, frontierA, frontierB, nodeCounts, triCounts, tempNodes, rootBox, gathered;
    const size_t scanEntries = std::max({scanScratchEntries(sortTableEntries(n)), scanScratchEntries(maxNodes + 1),
                                         scanScratchEntries(static_cast<size_t>(n) + 1)});
    RT_TRY(allocateTogether(scratch, {{&boxes, sizeof(Box4) * n},
                                      {&cbounds, sizeof(uint32_t) * 6},
                                      {&keysA, sizeof(uint64_t) * n},
                                      {&keysB, sizeof(uint64_t) * n},
                                      {&idxA, sizeof(uint32_t) * n},
                                      {&idxB, sizeof(uint32_t) * n},
                                      {&table, sizeof(uint32_t) * sortTableEntries(n)},
                                      {&partials, sizeof(uint32_t) * scanEntries},
                                      {&left, sizeof(uint32_t) * n},
                                      {&right, sizeof(uint32_t) * n},
                                      {&parent, sizeof(uint32_t) * ids},
                                      {&first, sizeof(uint32_t) * n},
                                      {&last, sizeof(uint32_t) * n},
                                      {&bounds, sizeof(Box4) * ids},
                                      {&arrivals, sizeof(uint32_t) * n},
                                      {&frontierA, sizeof(uint32_t) * maxNodes},
                                      {&frontierB, sizeof(uint32_t) * maxNodes},
                                      {&nodeCounts, sizeof(uint32_t) * (maxNodes + 1)},
                                      {&triCounts, sizeof(uint32_t) * (maxNodes + 1)},
                                      {&tempNodes, sizeof(WideNode) * maxNodes},
                                      {&rootBox, sizeof(Box4)},
                                      {&gathered, sizeof(uint2) * kWide * maxNodes}}));
    RT_HIP(hipMalloc(reinterpret_cast<void**>(&blas->triangles), sizeof(PackedTriangle) * n));

    phase("allocations");
    {
        // cbounds doubles as the flag of the index check before it takes the centroid bounds
        RT_HIP(hipMemsetAsync(cbounds.data, 0, sizeof(uint32_t), stream));
        const size_t count = 3 * static_cast<size_t>(n);
        const unsigned grid = static_cast<unsigned>(std::min<size_t>(divUp(count, 256), 65536));
        hipLaunchKernelGGL(indexCheckKernel, dim3(grid), dim3(256), 0, stream, indices, count, vertexCount,
                           cbounds.as<uint32_t>());
        uint32_t bad = 0;
        RT_HIP(hipMemcpyAsync(&bad, cbounds.data, sizeof(bad), hipMemcpyDeviceToHost, stream));
        RT_HIP(hipStreamSynchronize(stream));
        if (bad) {
            (void)hipFree(blas->triangles);
            blas->triangles = nullptr;
            return failInvalid("a triangle index is not below the vertex count");
        }
    }
    const uint32_t initBounds[6] = {0xffffffffU, 0xffffffffU, 0xffffffffU, 0U, 0U, 0U};
    RT_HIP(hipMemcpyAsync(cbounds.data, initBounds, sizeof(initBounds), hipMemcpyHostToDevice, stream));
    const auto* vbytes = static_cast<const uint8_t*>(vertices);
    const unsigned blocks = divUp(n, 256);
    hipLaunchKernelGGL(triangleBoundsKernel, dim3(std::min(blocks, kBoundsBlocks)), dim3(256), 0, stream, vbytes, stride,
                       indices, n, boxes.as<Box4>(), cbounds.as<uint32_t>());
    hipLaunchKernelGGL(mortonKernel, dim3(blocks), dim3(256), 0, stream, boxes.as<Box4>(), n, cbounds.as<uint32_t>(),
                       keysA.as<uint64_t>());
    RT_HIP(hipGetLastError());
    uint64_t* const keys[2] = {keysA.as<uint64_t>(), keysB.as<uint64_t>()};
    uint32_t* const idx[2] = {idxA.as<uint32_t>(), idxB.as<uint32_t>()};
    int result = 0;
    RT_TRY(radixSortPairs<uint64_t>(keys, idx, n, 63, table.as<uint32_t>(), partials.as<uint32_t>(), stream, &result));
    const uint64_t* sortedKeys = keys[result];
    const uint32_t* order = idx[result];
    phase("bounds, codes, sort");

    hipLaunchKernelGGL(leafBoundsKernel, dim3(blocks), dim3(256), 0, stream, boxes.as<Box4>(), order, n,
                       bounds.as<Box4>());
    bool probing = false;
    if (n >= 2 && builderChoice() == Builder::Clustering) {
        DeviceBuffer clustersA, clustersB, nearest, keep, merges, leafPosition, leafBounds;
        RT_TRY(clustersA.allocate(sizeof(uint32_t) * n));
        RT_TRY(clustersB.allocate(sizeof(uint32_t) * n));
        RT_TRY(nearest.allocate(sizeof(uint32_t) * n));
        RT_TRY(keep.allocate(sizeof(uint32_t) * (static_cast<size_t>(n) + 1)));
        RT_TRY(merges.allocate(sizeof(uint32_t) * (static_cast<size_t>(n) + 1)));
        RT_TRY(leafPosition.allocate(sizeof(uint32_t) * n));
        RT_TRY(leafBounds.allocate(sizeof(Box4) * n));
        uint32_t* counts = arrivals.as<uint32_t>();
        uint32_t* current = clustersA.as<uint32_t>();
        uint32_t* next = clustersB.as<uint32_t>();
        hipLaunchKernelGGL(iotaKernel, dim3(blocks), dim3(256), 0, stream, current, n, n - 1U);
        uint32_t m = n, created = 0;
        while (m > 1) {
            const unsigned grid = divUp(m + 1, 256);
            hipLaunchKernelGGL(plocNearestKernel, dim3(grid), dim3(256), 0, stream, current, m, bounds.as<Box4>(),
                               nearest.as<uint32_t>());
            hipLaunchKernelGGL(plocFlagKernel, dim3(grid), dim3(256), 0, stream, nearest.as<uint32_t>(), m,
                               keep.as<uint32_t>(), merges.as<uint32_t>());
            RT_TRY(exclusiveScanU32(keep.as<uint32_t>(), keep.as<uint32_t>(), static_cast<size_t>(m) + 1,
                                    partials.as<uint32_t>(), stream));
            RT_TRY(exclusiveScanU32(merges.as<uint32_t>(), merges.as<uint32_t>(), static_cast<size_t>(m) + 1,
                                    partials.as<uint32_t>(), stream));
            uint32_t totals[2] = {0, 0};
            RT_HIP(hipMemcpyAsync(&totals[0], keep.as<uint32_t>() + m, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
            RT_HIP(hipMemcpyAsync(&totals[1], merges.as<uint32_t>() + m, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
            RT_HIP(hipStreamSynchronize(stream));
            if (totals[1] == 0 || totals[0] + totals[1] != m || created + totals[1] > n - 1U)
                return failInvalid("clustering made no progress");
            const uint32_t idBase = (n - 1U) - created - totals[1];
            hipLaunchKernelGGL(plocMergeKernel, dim3(grid), dim3(256), 0, stream, current, nearest.as<uint32_t>(),
                               keep.as<uint32_t>(), merges.as<uint32_t>(), m, n, idBase, bounds.as<Box4>(),
                               left.as<uint32_t>(), right.as<uint32_t>(), parent.as<uint32_t>(), counts, next);
            created += totals[1];
            m = totals[0];
            std::swap(current, next);
        }
        if (created != n - 1U)
            return failInvalid("clustering lost nodes");
        const uint32_t none = kNoHit;
        RT_HIP(hipMemcpyAsync(parent.data, &none, sizeof(none), hipMemcpyHostToDevice, stream));
        uint32_t* reordered = idx[1 - result];
        hipLaunchKernelGGL(plocLeafOrderKernel, dim3(blocks), dim3(256), 0, stream, n, left.as<uint32_t>(),
                           right.as<uint32_t>(), parent.as<uint32_t>(), counts, order, bounds.as<Box4>(),
                           leafPosition.as<uint32_t>(), reordered, leafBounds.as<Box4>());
        hipLaunchKernelGGL(plocRangeKernel, dim3(divUp(n - 1, 256)), dim3(256), 0, stream, n, left.as<uint32_t>(),
                           right.as<uint32_t>(), parent.as<uint32_t>(), counts, first.as<uint32_t>(),
                           last.as<uint32_t>());
        hipLaunchKernelGGL(plocChildKernel, dim3(divUp(n - 1, 256)), dim3(256), 0, stream, n, left.as<uint32_t>(),
                           right.as<uint32_t>(), leafPosition.as<uint32_t>());
        RT_HIP(hipMemcpyAsync(bounds.as<Box4>() + (n - 1U), leafBounds.data, sizeof(Box4) * n, hipMemcpyDeviceToDevice,
                              stream));
        RT_HIP(hipStreamSynchronize(stream));  // the scratch buffers of this block go out of scope
        order = reordered;
    } else if (n >= 2) {
        hipLaunchKernelGGL(radixTreeKernel, dim3(divUp(n - 1, 256)), dim3(256), 0, stream, sortedKeys, n,
                           left.as<uint32_t>(), right.as<uint32_t>(), parent.as<uint32_t>(), first.as<uint32_t>(),
                           last.as<uint32_t>());
        DeviceBuffer readyA, readyB, readyCounts;
        RT_TRY(readyA.allocate(sizeof(uint32_t) * n));
        RT_TRY(readyB.allocate(sizeof(uint32_t) * n));
        RT_TRY(readyCounts.allocate(sizeof(uint32_t) * (kRefitMaxLevels + 1U)));
        RT_HIP(hipMemsetAsync(readyCounts.data, 0, sizeof(uint32_t) * (kRefitMaxLevels + 1U), stream));
        hipLaunchKernelGGL(refitSeedKernel, dim3(divUp(n, kRefitBlock)), dim3(kRefitBlock), 0, stream, left.as<uint32_t>(), right.as<uint32_t>(), n,
                           arrivals.as<uint32_t>(), readyA.as<uint32_t>(), readyCounts.as<uint32_t>());
        uint32_t* listIn = readyA.as<uint32_t>();
        uint32_t* listOut = readyB.as<uint32_t>();
        // a level holds at most half the nodes of the one below; the grid-stride loop covers any list
        const unsigned levelBlocks = std::max(1U, std::min<unsigned>(divUp(n, kRefitBlock), 1024U));
        bool finished = false;
        for (uint32_t level = 0; level < kRefitMaxLevels && !finished; level += kRefitBatch) {
            for (uint32_t k = level; k < level + kRefitBatch; ++k) {
                hipLaunchKernelGGL(refitLevelKernel, dim3(levelBlocks), dim3(kRefitBlock), 0, stream, left.as<uint32_t>(),
                                   right.as<uint32_t>(), parent.as<uint32_t>(), bounds.as<Box4>(),
                                   arrivals.as<uint32_t>(), listIn, listOut, readyCounts.as<uint32_t>(), k);
                std::swap(listIn, listOut);
            }
            uint32_t remaining = 0;
            RT_HIP(hipMemcpyAsync(&remaining, readyCounts.as<uint32_t>() + level + kRefitBatch, sizeof(uint32_t),
                                  hipMemcpyDeviceToHost, stream));
            RT_HIP(hipStreamSynchronize(stream));
            finished = remaining == 0;
        }
        if (!finished)
            return failInvalid("bounds pass did not finish");
        probing = allowRefine && refineChoice() && n >= 64U;
    }
    phase("binary tree and its boxes");
    RT_HIP(hipGetLastError());
    // The surface-area rebuild of the upper levels runs beside the collapse of the tree as it is (UpperRefinement);
    // if the rebuilt levels are taken, the collapse runs again over them.
    UpperRefinement refinement(n, stream, left, right, parent, first, last, bounds, arrivals, partials);
    if (probing) {
        RT_TRY(refinement.start());
        if (UpperRefinement::always()) {
            bool replaced = false;
            RT_TRY(refinement.finish(order, idx[1 - result], replaced));
            probing = false;
        }
    }

    // top-down collapse, one level per iteration
    uint32_t levelNodeBase = 0, depth = 0;
    auto collapse = [&]() -> rt_status {
    // slot -> triangle, in whichever of the sort's two index arrays does not hold the triangle order
    uint32_t* slotPrimitive = order == idx[0] ? idx[1] : idx[0];
    uint32_t* frontier = frontierA.as<uint32_t>();
    uint32_t* nextFrontier = frontierB.as<uint32_t>();
    const uint32_t rootId = 0;  // internal node 0, or leaf 0 when n == 1 (id n-1 = 0)
    RT_HIP(hipMemcpyAsync(frontier, &rootId, sizeof(rootId), hipMemcpyHostToDevice, stream));
    uint32_t entries = 1, triBase = 0;
    levelNodeBase = 0;
    depth = 0;
    while (entries > 0) {
        if (static_cast<size_t>(levelNodeBase) + entries > maxNodes)
            return failInvalid("wide-node estimate exceeded");
        hipLaunchKernelGGL(collapseCountKernel, dim3(divUp(entries + 1, 128)), dim3(128), 0, stream, frontier, entries,
                           n, left.as<uint32_t>(), right.as<uint32_t>(), first.as<uint32_t>(), last.as<uint32_t>(),
                           bounds.as<Box4>(), nodeCounts.as<uint32_t>(), triCounts.as<uint32_t>(), gathered.as<uint2>());
        RT_TRY(exclusiveScanU32(nodeCounts.as<uint32_t>(), nodeCounts.as<uint32_t>(), entries + 1,
                                partials.as<uint32_t>(), stream));
        RT_TRY(exclusiveScanU32(triCounts.as<uint32_t>(), triCounts.as<uint32_t>(), entries + 1,
                                partials.as<uint32_t>(), stream));
        uint32_t totals[2] = {0, 0};
        RT_HIP(hipMemcpyAsync(&totals[0], nodeCounts.as<uint32_t>() + entries, sizeof(uint32_t),
                              hipMemcpyDeviceToHost, stream));
        RT_HIP(hipMemcpyAsync(&totals[1], triCounts.as<uint32_t>() + entries, sizeof(uint32_t),
                              hipMemcpyDeviceToHost, stream));
        RT_HIP(hipStreamSynchronize(stream));
        const uint32_t nextNodeBase = levelNodeBase + entries;
        if (static_cast<size_t>(nextNodeBase) + totals[0] > maxNodes || static_cast<uint64_t>(triBase) + totals[1] > n)
            return failInvalid("wide-node estimate exceeded");
        hipLaunchKernelGGL(collapseEmitKernel, dim3(divUp(entries, 128)), dim3(128), 0, stream, frontier, entries, n,
                           left.as<uint32_t>(), right.as<uint32_t>(), first.as<uint32_t>(), last.as<uint32_t>(),
                           bounds.as<Box4>(), nodeCounts.as<uint32_t>(), triCounts.as<uint32_t>(), levelNodeBase,
                           nextNodeBase, triBase, order, tempNodes.as<WideNode>(), slotPrimitive, nextFrontier,
                           depth == 0 ? rootBox.as<Box4>() : nullptr, gathered.as<uint2>());
        RT_HIP(hipGetLastError());
        levelNodeBase = nextNodeBase;
        triBase += totals[1];
        entries = totals[0];
        std::swap(frontier, nextFrontier);
        ++depth;
    }
    if (triBase != n)
        return failInvalid("collapse lost triangles");
    hipLaunchKernelGGL(packTrianglesKernel, dim3(blocks), dim3(256), 0, stream, slotPrimitive, n, vbytes, stride, indices,
                       blas->triangles);
    RT_HIP(hipGetLastError());
    return RT_SUCCESS;
    };
    RT_TRY(collapse());
    phase("collapse");
    if (probing) {
        bool replaced = false;
        RT_TRY(refinement.finish(order, idx[1 - result], replaced));
        if (replaced)
            RT_TRY(collapse());
    }

    blas->nodeCount = levelNodeBase;
    blas->triangleCount = n;
    blas->depth = depth;
    RT_HIP(hipMalloc(reinterpret_cast<void**>(&blas->nodes), sizeof(WideNode) * blas->nodeCount));
    RT_HIP(hipMemcpyAsync(blas->nodes, tempNodes.data, sizeof(WideNode) * blas->nodeCount, hipMemcpyDeviceToDevice,
                          stream));
    Box4 root;
    RT_HIP(hipMemcpyAsync(&root, rootBox.data, sizeof(root), hipMemcpyDeviceToHost, stream));
    RT_HIP(hipEventRecord(stop, stream));
    RT_HIP(hipStreamSynchronize(stream));
    phase("second collapse, node copy");
    RT_HIP(hipEventElapsedTime(&blas->buildMilliseconds, start, stop));
    blas->bounds = {{root.lo.x, root.lo.y, root.lo.z}, {root.hi.x, root.hi.y, root.hi.z}};
    return RT_SUCCESS;
}

// ---- top level (host) ---------------------------------------------------------

// Inverse of the affine 3x4 in binary64, each entry rounded once to binary32.
void invertAffine(const float* m, float* out) {
    const double a = m[0], b = m[1], c = m[2], d = m[4], e = m[5], f = m[6], g = m[8], h = m[9], i = m[10];
    const double c00 = e * i - f * h, c01 = c * h - b * i, c02 = b * f - c * e;
    const double c10 = f * g - d * i, c11 = a * i - c * g, c12 = c * d - a * f;
    const double c20 = d * h - e * g, c21 = b * g - a * h, c22 = a * e - b * d;
    const double det = a * c00 + b * c10 + c * c20;
    const double r = 1.0 / det;
    const double inv[9] = {c00 * r, c01 * r, c02 * r, c10 * r, c11 * r, c12 * r, c20 * r, c21 * r, c22 * r};
    const double tx = m[3], ty = m[7], tz = m[11];
    for (int row = 0; row < 3; ++row) {
        out[row * 4 + 0] = static_cast<float>(inv[row * 3 + 0]);
        out[row * 4 + 1] = static_cast<float>(inv[row * 3 + 1]);
        out[row * 4 + 2] = static_cast<float>(inv[row * 3 + 2]);
        out[row * 4 + 3] =
            static_cast<float>(-(inv[row * 3 + 0] * tx + inv[row * 3 + 1] * ty + inv[row * 3 + 2] * tz));
    }
}

Aabb emptyBox() {
    return {{INFINITY, INFINITY, INFINITY}, {-INFINITY, -INFINITY, -INFINITY}};
}

void growBox(Aabb& b, const Aabb& o) {
    for (int a = 0; a < 3; ++a) {
        b.lo[a] = std::min(b.lo[a], o.lo[a]);
        b.hi[a] = std::max(b.hi[a], o.hi[a]);
    }
}

// World-space box of an instance: the eight corners of the bottom-level root
// box through the object->world transform, then a margin for the rounding of
// the inverse transform the rays go through.
Aabb instanceBox(const float* m, const Aabb& object) {
    Aabb box = emptyBox();
    for (int corner = 0; corner < 8; ++corner) {
        const double x = corner & 1 ? object.hi[0] : object.lo[0];
        const double y = corner & 2 ? object.hi[1] : object.lo[1];
        const double z = corner & 4 ? object.hi[2] : object.lo[2];
        for (int a = 0; a < 3; ++a) {
            const double v = m[a * 4] * x + m[a * 4 + 1] * y + m[a * 4 + 2] * z + m[a * 4 + 3];
            box.lo[a] = std::min(box.lo[a], static_cast<float>(std::nextafter(static_cast<float>(v), -INFINITY)));
            box.hi[a] = std::max(box.hi[a], static_cast<float>(std::nextafter(static_cast<float>(v), INFINITY)));
        }
    }
    // from the largest coordinate, on every axis (see boxMargin: an instance may be as flat as a triangle)
    float m2 = 0.0f;
    for (int a = 0; a < 3; ++a)
        m2 = std::max(m2, std::max(std::max(std::fabs(box.lo[a]), std::fabs(box.hi[a])), box.hi[a] - box.lo[a]));
    const float pad = std::max(m2 * 0x1p-16f, 1e-30f);
    for (int a = 0; a < 3; ++a) {
        box.lo[a] -= pad;
        box.hi[a] += pad;
    }
    return box;
}

float scaleOfHost(uint8_t e) {
    const uint32_t bits = static_cast<uint32_t>(e) << 23;
    float v;
    std::memcpy(&v, &bits, sizeof(v));
    return v;
}

// Tree over instance boxes written straight into wide nodes: every node takes up
// to eight children by repeatedly splitting the group with the largest box (see
// split()); a group of one instance is a leaf child.
struct TopBuilder {
    const std::vector<Aabb>& boxes;   // one per item: an instance, or a folded triangle (see kFoldedBit)
    const std::vector<uint8_t>& folded;  // per item: whether it is a folded triangle
    std::vector<uint32_t> order;      // item permutation the splits work on
    std::vector<uint32_t> items;      // item index per leaf item slot, contiguous per node
    std::vector<WideNode> nodes;
    uint32_t depth = 0;
    // Levels the tree may have: what the traversal stack leaves after the deepest bottom-level tree.  Cuts by
    // surface area are as lopsided as the boxes ask for (a chain of nested instances peels off a few per level),
    // so a node whose groups would not fit the levels left below it is cut by count instead: eight even parts
    // per node always fit count <= 8^levels.
    uint32_t budget = 0xffffffffU;
    bool pairs = true;  // leaf children of two folded triangles (RT_SCENE_FOLD=single: one each)

    static uint64_t capacityOf(uint32_t levels) { return levels >= 21U ? ~0ULL : 1ULL << (3U * levels); }

    struct Group {
        uint32_t first, count;
        Aabb box;
    };

    TopBuilder(const std::vector<Aabb>& b, const std::vector<uint8_t>& f) : boxes(b), folded(f), order(b.size()) {
        for (uint32_t i = 0; i < order.size(); ++i)
            order[i] = i;
    }

    Aabb boxOf(uint32_t first, uint32_t count) const {
        Aabb box = emptyBox();
        for (uint32_t i = first; i < first + count; ++i)
            growBox(box, boxes[order[i]]);
        return box;
    }

    static float area(const Aabb& b) {
        const float x = b.hi[0] - b.lo[0], y = b.hi[1] - b.lo[1], z = b.hi[2] - b.lo[2];
        return x * y + y * z + z * x;
    }

    // Cuts a group in two.  Up to kSweepLimit instances the cut is the one of least
    // surface-area cost, area(left) * |left| + area(right) * |right|, over all three
    // axes and every position in centroid order (a full sweep: instances of very
    // different sizes, a room around a field of small objects, are told apart where
    // a median cut would put them in overlapping halves); larger groups are cut at
    // the median of the axis their centroids spread most along.
    static constexpr uint32_t kSweepLimit = 4096;

    void sortAlong(const Group& g, int axis) {
        std::sort(order.begin() + g.first, order.begin() + g.first + g.count, [&](uint32_t x, uint32_t y) {
            const float cx = boxes[x].lo[axis] + boxes[x].hi[axis];
            const float cy = boxes[y].lo[axis] + boxes[y].hi[axis];
            return cx < cy || (cx == cy && x < y);
        });
    }

    void split(const Group& g, Group& l, Group& r, bool byCount = false) {
        uint32_t mid = g.first + g.count / 2;
        if (g.count <= kSweepLimit && !byCount) {
            int bestAxis = -1;
            uint32_t bestLeft = 0;
            float bestCost = INFINITY;
            std::vector<float> rightArea(g.count);
            for (int axis = 0; axis < 3; ++axis) {
                sortAlong(g, axis);
                Aabb box = emptyBox();
                for (uint32_t i = g.count; i-- > 1;) {
                    growBox(box, boxes[order[g.first + i]]);
                    rightArea[i] = area(box);
                }
                box = emptyBox();
                for (uint32_t i = 1; i < g.count; ++i) {
                    growBox(box, boxes[order[g.first + i - 1]]);
                    const float cost = area(box) * static_cast<float>(i) + rightArea[i] * static_cast<float>(g.count - i);
                    if (cost < bestCost) {
                        bestCost = cost;
                        bestAxis = axis;
                        bestLeft = i;
                    }
                }
            }
            if (bestAxis >= 0) {
                if (bestAxis != 2)
                    sortAlong(g, bestAxis);
                mid = g.first + bestLeft;
            }
        } else {
            Aabb centers = emptyBox();
            for (uint32_t i = g.first; i < g.first + g.count; ++i) {
                const Aabb& b = boxes[order[i]];
                for (int a = 0; a < 3; ++a) {
                    const float c = 0.5f * (b.lo[a] + b.hi[a]);
                    centers.lo[a] = std::min(centers.lo[a], c);
                    centers.hi[a] = std::max(centers.hi[a], c);
                }
            }
            int axis = 0;
            for (int a = 1; a < 3; ++a)
                if (centers.hi[a] - centers.lo[a] > centers.hi[axis] - centers.lo[axis])
                    axis = a;
            std::nth_element(order.begin() + g.first, order.begin() + mid, order.begin() + g.first + g.count,
                             [&](uint32_t x, uint32_t y) {
                                 const float cx = boxes[x].lo[axis] + boxes[x].hi[axis];
                                 const float cy = boxes[y].lo[axis] + boxes[y].hi[axis];
                                 return cx < cy || (cx == cy && x < y);
                             });
        }
        l = {g.first, mid - g.first, boxOf(g.first, mid - g.first)};
        r = {mid, g.first + g.count - mid, boxOf(mid, g.first + g.count - mid)};
    }

    // Builds the node for `group` at index `self`; children nodes are appended.
    void build(uint32_t self, const Group& group, uint32_t level) {
        depth = std::max(depth, level + 1);
        std::vector<Group> groups{group};
        // a group of two folded triangles is a leaf child of two items, as in a bottom-level tree: it takes the
        // place of a node that would hold nothing else
        auto isLeaf = [&](const Group& g) {
            return g.count == 1 || (pairs && g.count == 2 && folded[order[g.first]] && folded[order[g.first + 1]]);
        };
        const uint32_t below = budget > level + 1U ? budget - level - 1U : 0U;  // levels left for the children
        for (int byCount = 0; byCount < 2 && group.count > 1; ++byCount) {
            groups.assign(1, group);
            while (groups.size() < static_cast<size_t>(kWide)) {
                int pick = -1;
                float best = -1.0f;
                for (size_t k = 0; k < groups.size(); ++k) {
                    const float weight = byCount ? static_cast<float>(groups[k].count) : area(groups[k].box);
                    if (groups[k].count > 1 && weight > best) {
                        best = weight;
                        pick = static_cast<int>(k);
                    }
                }
                if (pick < 0)
                    break;
                Group l, r;
                split(groups[pick], l, r, byCount != 0);
                groups[pick] = l;
                groups.insert(groups.begin() + pick + 1, r);
            }
            bool fits = true;
            for (const Group& g : groups)
                fits = fits && (isLeaf(g) || g.count <= capacityOf(below));
            if (fits)
                break;
        }
        Aabb all = emptyBox();
        for (const auto& g : groups)
            growBox(all, g.box);
        WideNode node;
        std::memset(&node, 0, sizeof(node));
        for (int a = 0; a < 3; ++a) {
            node.origin[a] = all.lo[a];
            int ex;
            std::frexp((all.hi[a] - all.lo[a]) / 255.0f, &ex);
            int biased = std::min(std::max(ex + 127, 1), 254);
            while (biased < 254 && all.lo[a] + 255.0f * scaleOfHost(static_cast<uint8_t>(biased)) < all.hi[a])
                ++biased;
            node.exp[a] = static_cast<uint8_t>(biased);
        }
        uint32_t internal = 0;
        for (const auto& g : groups)
            internal += !isLeaf(g);
        const uint32_t childBase = static_cast<uint32_t>(nodes.size());
        nodes.resize(nodes.size() + internal);
        node.childBase = childBase;
        // the leaf items of one node take consecutive slots of `items`
        node.primBase = static_cast<uint32_t>(items.size());
        uint32_t slot = 0, itemSlot = 0;
        std::vector<std::pair<uint32_t, Group>> pending;
        for (size_t k = 0; k < groups.size(); ++k) {
            const Group& g = groups[k];
            for (int a = 0; a < 3; ++a) {
                const float scale = scaleOfHost(node.exp[a]);
                float ql = std::floor((g.box.lo[a] - all.lo[a]) / scale);
                ql = std::min(std::max(ql, 0.0f), 255.0f);
                while (ql > 0.0f && all.lo[a] + ql * scale > g.box.lo[a])
                    ql -= 1.0f;
                float qh = std::ceil((g.box.hi[a] - all.lo[a]) / scale);
                qh = std::min(std::max(qh, 0.0f), 255.0f);
                while (qh < 255.0f && all.lo[a] + qh * scale < g.box.hi[a])
                    qh += 1.0f;
                node.qlo[a][k] = static_cast<uint8_t>(ql);
                node.qhi[a][k] = static_cast<uint8_t>(qh);
            }
            if (!isLeaf(g)) {
                node.imask |= static_cast<uint8_t>(1U << k);
                node.meta[k] = static_cast<uint8_t>(slot);
                pending.emplace_back(childBase + slot, g);
                ++slot;
            } else {
                // the count field of folded triangles also carries kFoldedBit's place (bit 2 of the field)
                const uint32_t field = folded[order[g.first]] ? (kFoldedBit >> 28 | g.count) : 1U;
                node.meta[k] = static_cast<uint8_t>(field << 5 | itemSlot);
                for (uint32_t i = 0; i < g.count; ++i)
                    items.push_back(order[g.first + i]);
                itemSlot += g.count;
            }
        }
        for (size_t k = groups.size(); k < static_cast<size_t>(kWide); ++k)  // empty slots: a box no ray enters
            for (int a = 0; a < 3; ++a) {
                node.qlo[a][k] = 255;
                node.qhi[a][k] = 0;
            }
        nodes[self] = node;
        for (auto& p : pending)
            build(p.first, p.second, level + 1);
    }
};

// Whether an instance's triangles go into the top-level tree themselves (see kFoldedBit): its world->object
// record is the identity bit for bit and its mesh is small.  RT_SCENE_FOLD=0 folds nothing.
constexpr uint32_t kFoldTrianglesPerInstance = 64;
constexpr uint32_t kFoldTrianglesPerScene = 4096;

static bool foldChoice() {
    const char* v = std::getenv("RT_SCENE_FOLD");
    return !(v && std::strcmp(v, "0") == 0);
}

// Bitwise comparison: the zeros of the rotation part must be positive, those of the translation have the sign
// `translation` names (the inverse of the identity comes out of invertAffine with translations of -0, which the
// transform of a point adds last and which therefore change no value, not even a zero's sign).
static bool isIdentity(const float* m, float translation) {
    const float identity[12] = {1.0f, 0.0f, 0.0f, translation, 0.0f, 1.0f, 0.0f, translation, 0.0f, 0.0f, 1.0f, translation};
    return std::memcmp(m, identity, sizeof(identity)) == 0;
}

// Box of one triangle with the margin the bottom-level builder gives its leaf boxes (padBox).
static Aabb triangleBox(const PackedTriangle& t) {
    Aabb box = emptyBox();
    const float* corners[3] = {t.v0, t.v1, t.v2};
    for (const float* c : corners)
        for (int a = 0; a < 3; ++a) {
            box.lo[a] = std::min(box.lo[a], c[a]);
            box.hi[a] = std::max(box.hi[a], c[a]);
        }
    const float pad = boxMargin(box.lo, box.hi);
    for (int a = 0; a < 3; ++a) {
        box.lo[a] -= pad;
        box.hi[a] += pad;
    }
    return box;
}

rt_status buildScene(const rt_instance_desc* instances, uint32_t count, hipStream_t stream, rt_scene* scene) {
    // items of the top-level tree: the instances that are entered, and the triangles of those that are folded
    std::vector<Aabb> boxes;
    std::vector<uint8_t> folded;
    std::vector<InstanceRecord> records;
    boxes.reserve(count);
    folded.reserve(count);
    records.reserve(count);
    std::vector<InstanceRecord> own(count);  // every instance's record, folded or not (the shading table reads it)
    scene->inverses.resize(static_cast<size_t>(count) * 12);
    const bool fold = foldChoice();
    std::vector<PackedTriangle> mesh;
    for (uint32_t i = 0; i < count; ++i) {
        const rt_instance_desc& d = instances[i];
        if (!d.blas)
            return failInvalid("instance without a bottom-level structure");
        if (d.blas->device != scene->device)
            return failInvalid("bottom-level structure lives on another device");
        InstanceRecord& r = own[i];
        std::memset(&r, 0, sizeof(r));
        invertAffine(d.transform, r.toObject);
        std::memcpy(&scene->inverses[static_cast<size_t>(i) * 12], r.toObject, sizeof(float) * 12);
        r.nodes = d.blas->nodes;
        r.triangles = d.blas->triangles;
        r.instanceId = d.instanceId;
        r.geometry = d.geometry;
        r.nodeCount = static_cast<uint32_t>(d.blas->nodeCount);
        r.triangleCount = static_cast<uint32_t>(d.blas->triangleCount);
        scene->blasNodes += d.blas->nodeCount;
        scene->triangles += d.blas->triangleCount;
        if (fold && d.blas->triangleCount <= kFoldTrianglesPerInstance &&
            scene->foldedTriangles + d.blas->triangleCount <= kFoldTrianglesPerScene && isIdentity(r.toObject, -0.0f) &&
            isIdentity(d.transform, 0.0f)) {
            mesh.resize(d.blas->triangleCount);
            RT_HIP(hipMemcpy(mesh.data(), d.blas->triangles, sizeof(PackedTriangle) * mesh.size(), hipMemcpyDeviceToHost));
            for (PackedTriangle t : mesh) {
                t.pad1 = d.instanceId;
                t.pad2 = 0;
                InstanceRecord item;
                std::memset(&item, 0, sizeof(item));
                static_assert(sizeof(PackedTriangle) <= sizeof(InstanceRecord), "a folded triangle fits an item record");
                std::memcpy(&item, &t, sizeof(t));
                boxes.push_back(triangleBox(t));
                folded.push_back(1);
                records.push_back(item);
            }
            scene->foldedTriangles += static_cast<uint32_t>(mesh.size());
            ++scene->foldedInstances;
            continue;
        }
        boxes.push_back(instanceBox(d.transform, d.blas->bounds));
        folded.push_back(0);
        records.push_back(r);
        scene->blasDepth = std::max(scene->blasDepth, d.blas->depth);
    }
    const uint32_t itemCount = static_cast<uint32_t>(records.size());
    TopBuilder builder(boxes, folded);
    builder.nodes.resize(1);
    // top + bottom levels the traversal stack can follow (fitsTraversalStack): the top tree takes what the deepest
    // bottom-level tree leaves
    uint32_t levels = 0;
    while (fitsTraversalStack(levels + 1U, scene->blasDepth))
        ++levels;
    if (levels == 0U || itemCount > TopBuilder::capacityOf(levels)) {
        setLastError("tree too deep for the traversal stack: " + std::to_string(count) + " instances over bottom-level trees of " +
                     std::to_string(scene->blasDepth) + " levels");
        return RT_ERROR_OUT_OF_RANGE;
    }
    builder.budget = levels;
    if (const char* v = std::getenv("RT_SCENE_FOLD"))
        builder.pairs = std::strcmp(v, "single") != 0;
    scene->bounds = builder.boxOf(0, itemCount);
    builder.build(0, {0, itemCount, scene->bounds}, 0);
    std::vector<InstanceRecord> ordered(itemCount);
    if (builder.items.size() != itemCount)
        return failInvalid("top-level build lost instances");
    for (uint32_t k = 0; k < itemCount; ++k)
        ordered[k] = records[builder.items[k]];
    scene->nodeCount = builder.nodes.size();
    scene->instanceCount = count;
    scene->itemCount = itemCount;
    scene->depth = builder.depth;
    if (!fitsTraversalStack(scene->depth, scene->blasDepth)) {
        setLastError("tree too deep for the traversal stack: " + std::to_string(scene->depth) + " top levels + " +
                     std::to_string(scene->blasDepth) + " bottom levels");
        return RT_ERROR_OUT_OF_RANGE;
    }
    RT_HIP(hipMalloc(reinterpret_cast<void**>(&scene->nodes), sizeof(WideNode) * builder.nodes.size()));
    RT_HIP(hipMalloc(reinterpret_cast<void**>(&scene->instances), sizeof(InstanceRecord) * itemCount));
    RT_HIP(hipMalloc(reinterpret_cast<void**>(&scene->feedCursors), sizeof(uint32_t) * kFeedWords * kFeedSlots));
    for (hipEvent_t& event : scene->feedEvents)
        RT_HIP(hipEventCreateWithFlags(&event, hipEventDisableTiming));
    RT_HIP(hipMemcpyAsync(scene->nodes, builder.nodes.data(), sizeof(WideNode) * builder.nodes.size(),
                          hipMemcpyHostToDevice, stream));
    RT_HIP(hipMemcpyAsync(scene->instances, ordered.data(), sizeof(InstanceRecord) * itemCount, hipMemcpyHostToDevice,
                          stream));
    // shading-side table indexed by instance id
    uint32_t maxId = 0;
    for (uint32_t i = 0; i < count; ++i)
        maxId = std::max(maxId, instances[i].instanceId);
    if (maxId > (1U << 26))
        return failInvalid("instance ids must be small enough to index a table");
    std::vector<ShadeInstance> shade(static_cast<size_t>(maxId) + 1);
    std::memset(shade.data(), 0, shade.size() * sizeof(ShadeInstance));
    for (uint32_t i = 0; i < count; ++i) {
        ShadeInstance& si = shade[instances[i].instanceId];
        std::memcpy(si.toObject, own[i].toObject, sizeof(si.toObject));
        std::memcpy(si.toWorld, instances[i].transform, sizeof(si.toWorld));
        si.geometry = instances[i].geometry;
    }
    scene->shadeInstanceCount = maxId + 1;
    RT_HIP(hipMalloc(reinterpret_cast<void**>(&scene->shadeInstances), sizeof(ShadeInstance) * shade.size()));
    RT_HIP(hipMemcpyAsync(scene->shadeInstances, shade.data(), sizeof(ShadeInstance) * shade.size(),
                          hipMemcpyHostToDevice, stream));
    RT_HIP(hipStreamSynchronize(stream));
    scene->shadeHost = std::move(shade);
    return RT_SUCCESS;
}

}  // namespace
}  // namespace rt

using namespace rt;

extern "C" {

rt_status rt_blas_build(const void* vertices, uint32_t vertexCount, uint32_t vertexStride, const uint32_t* indices,
                        uint32_t triangleCount, void* stream, rt_blas** out) {
    if (!out)
        return failInvalid("out is null");
    *out = nullptr;
    if (!vertices || !indices || triangleCount == 0 || vertexCount == 0)
        return failInvalid("empty geometry");
    if (vertexStride < 12 || vertexStride % 4 != 0)
        return failInvalid("vertex stride must be a multiple of 4 and at least 12");
    if (triangleCount > (1U << 28) - 2U)
        return failInvalid("too many triangles");
    int device = 0;
    RT_HIP(hipGetDevice(&device));
    auto* blas = new rt_blas();
    blas->device = device;
    rt_status status = buildBlas(vertices, vertexCount, vertexStride, indices, triangleCount,
                                 static_cast<hipStream_t>(stream), blas);
    // The surface-area rebuild of the upper levels follows the boxes, not the count, and can come out deeper than
    // the radix tree it replaced.  A tree that leaves the top level fewer than four of the levels the traversal
    // stack can follow is built again without the rebuild, and the shallower of the two is kept.
    if (status == RT_SUCCESS && refineChoice() && !fitsTraversalStack(4U, blas->depth)) {
        auto* plain = new rt_blas();
        plain->device = blas->device;
        status = buildBlas(vertices, vertexCount, vertexStride, indices, triangleCount, static_cast<hipStream_t>(stream),
                           plain, false);
        if (status == RT_SUCCESS && plain->depth < blas->depth)
            std::swap(blas, plain);
        rt_blas_destroy(plain);
    }
    if (status != RT_SUCCESS) {
        rt_blas_destroy(blas);
        return status;
    }
    *out = blas;
    return RT_SUCCESS;
}

void rt_blas_destroy(rt_blas* blas) {
    if (!blas)
        return;
    if (blas->nodes)
        (void)hipFree(blas->nodes);
    if (blas->triangles)
        (void)hipFree(blas->triangles);
    delete blas;
}

rt_status rt_scene_build(const rt_instance_desc* instances, uint32_t instanceCount, void* stream, rt_scene** out) {
    if (!out)
        return failInvalid("out is null");
    *out = nullptr;
    if (!instances || instanceCount == 0)
        return failInvalid("the scene has no geometry");  // renderer.cpp:217-218
    auto* scene = new rt_scene();
    RT_HIP(hipGetDevice(&scene->device));
    const rt_status status = buildScene(instances, instanceCount, static_cast<hipStream_t>(stream), scene);
    if (status != RT_SUCCESS) {
        rt_scene_destroy(scene);
        return status;
    }
    *out = scene;
    return RT_SUCCESS;
}

void rt_scene_destroy(rt_scene* scene) {
    if (!scene)
        return;
    if (scene->nodes)
        (void)hipFree(scene->nodes);
    if (scene->instances)
        (void)hipFree(scene->instances);
    if (scene->feedCursors)
        (void)hipFree(scene->feedCursors);
    for (hipEvent_t event : scene->feedEvents)
        if (event)
            (void)hipEventDestroy(event);
    if (scene->shadeInstances)
        (void)hipFree(scene->shadeInstances);
    if (scene->hitGroups)
        (void)hipFree(scene->hitGroups);
    if (scene->orderData)
        (void)hipFree(scene->orderData);
    if (scene->orderEvent)
        (void)hipEventDestroy(scene->orderEvent);
    delete scene;
}

rt_status rt_scene_set_geometry(rt_scene* scene, const rt_hit_group_data* records, uint32_t count) {
    if (!scene || !records || count == 0)
        return failInvalid("null pointer");
    // nothing of the scene changes before the new records are known to be good and in place
    for (const ShadeInstance& si : scene->shadeHost)
        if (si.geometry >= count) {
            setLastError("an instance refers to a geometry beyond the records given");
            return RT_ERROR_OUT_OF_RANGE;
        }
    rt_hit_group_data* fresh = nullptr;
    RT_HIP(hipMalloc(reinterpret_cast<void**>(&fresh), sizeof(rt_hit_group_data) * count));
    const hipError_t copied = hipMemcpy(fresh, records, sizeof(rt_hit_group_data) * count, hipMemcpyHostToDevice);
    if (copied != hipSuccess) {
        (void)hipFree(fresh);
        return fail(copied, "hipMemcpy (hit-group records)");
    }
    // launches that still read the old records or the shading table finish first
    RT_HIP(hipDeviceSynchronize());
    rt_hit_group_data* old = scene->hitGroups;
    scene->hitGroups = fresh;
    scene->hitGroupCount = count;
    // every instance carries its geometry's record
    for (ShadeInstance& si : scene->shadeHost) {
        si.material = records[si.geometry].material;
        si.vertices = records[si.geometry].vertices;
        si.indices = records[si.geometry].indices;
    }
    RT_HIP(hipMemcpy(scene->shadeInstances, scene->shadeHost.data(), sizeof(ShadeInstance) * scene->shadeHost.size(),
                     hipMemcpyHostToDevice));
    if (old)
        (void)hipFree(old);
    return RT_SUCCESS;
}

uint64_t rt_scene_handle(const rt_scene* scene) {
    return reinterpret_cast<uint64_t>(scene);
}

rt_status rt_scene_instance_inverse(const rt_scene* scene, uint32_t index, float* out12) {
    if (!scene || !out12 || index >= scene->instanceCount)
        return failInvalid("instance index");
    std::memcpy(out12, &scene->inverses[static_cast<size_t>(index) * 12], sizeof(float) * 12);
    return RT_SUCCESS;
}

rt_status rt_scene_folded(const rt_scene* scene, uint32_t* instances, uint32_t* triangles) {
    if (!scene || !instances || !triangles)
        return failInvalid("null argument");
    *instances = scene->foldedInstances;
    *triangles = scene->foldedTriangles;
    return RT_SUCCESS;
}

rt_status rt_scene_stats(const rt_scene* scene, rt_bvh_stats* out) {
    if (!scene || !out)
        return failInvalid("null pointer");
    out->nodeCount = scene->nodeCount + scene->blasNodes;
    out->nodeBytes = out->nodeCount * sizeof(WideNode);
    out->triangleCount = scene->triangles;
    out->triangleBytes = scene->triangles * sizeof(PackedTriangle);
    out->instanceCount = scene->instanceCount;
    out->maxDepth = scene->depth + scene->blasDepth;
    out->buildMilliseconds = scene->buildMilliseconds;
    return RT_SUCCESS;
}

rt_status rt_blas_stats(const rt_blas* blas, rt_bvh_stats* out) {
    if (!blas || !out)
        return failInvalid("null pointer");
    out->nodeCount = blas->nodeCount;
    out->nodeBytes = blas->nodeCount * sizeof(WideNode);
    out->triangleCount = blas->triangleCount;
    out->triangleBytes = blas->triangleCount * sizeof(PackedTriangle);
    out->instanceCount = 0;
    out->maxDepth = blas->depth;
    out->buildMilliseconds = blas->buildMilliseconds;
    return RT_SUCCESS;
}

}  // extern "C"
