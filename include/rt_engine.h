/*
 * C ABI of the MI355X ray-intersection + photon-gather engine
 * (librt_engine.so).  Plain pointers and sizes only.  Every pointer marked
 * "device" is a device address valid on the current HIP device; streams are
 * hipStream_t passed as void* (NULL = the default stream).
 *
 * Each entry point names the reference interface it replaces
 * (paths relative to the reference's repository root).
 *
 * Status codes: 0 (RT_SUCCESS) on success; a positive value is the hipError_t
 * of the first failing HIP call (the reference returns the first failing
 * cudaError_t, src/photon_grid.cu:53-82); negative values are RT_ERROR_*.
 */
#ifndef RT_ENGINE_H
#define RT_ENGINE_H

#include <stddef.h>
#include <stdint.h>

#include "rt_types.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef int rt_status;
#define RT_SUCCESS 0
#define RT_ERROR_INVALID_VALUE (-1)
#define RT_ERROR_OUT_OF_RANGE (-2)

/* Message of the most recent failure on the calling host thread. */
const char* rt_last_error(void);
/* Printable name of a status code (cudaGetErrorString in src/renderer.cpp:30-33). */
const char* rt_status_string(rt_status status);

/* ------------------------------------------------------------------------
 * Photon hash grid — replaces src/photon_grid.hpp:12-20 (src/photon_grid.cu)
 * ------------------------------------------------------------------------ */

/* Scratch bytes needed to sort up to maxCount photons into bucketCount buckets. */
size_t rt_photon_grid_scratch_bytes(uint32_t maxCount, uint32_t bucketCount);

/* photonScanStorage(bucketCount), src/photon_grid.hpp:12.  The reference sizes
 * its photon buffer at twice the bucket count (src/renderer.cpp:533-534), so
 * this returns rt_photon_grid_scratch_bytes(2 * bucketCount, bucketCount). */
size_t rt_photon_scan_storage(uint32_t bucketCount);

/* buildPhotonGrid(...), src/photon_grid.hpp:17-20.  All pointers are device
 * pointers allocated by the caller: counts and offsets hold bucketCount + 1
 * uint32, sorted holds count photons.  On return offsets[b]..offsets[b+1]
 * delimits bucket b inside sorted and counts has been clobbered.  Photons of
 * one bucket keep their input order (the reference leaves the order to an
 * atomic race, src/photon_grid.cu:36).  bucketCount must be a power of two.
 * If scanBytes is smaller than rt_photon_grid_scratch_bytes(count,
 * bucketCount) the library uses a cached workspace of its own, one per device:
 * builds that fall back to it are ordered one behind the other by an event,
 * whatever streams they are on, and a build that needs a larger workspace
 * than the cached one synchronises the device while it is replaced.  Builds
 * that bring their own scratch run concurrently on different streams.
 * count == 0 is legal.  Runs on the default stream and synchronises the device before
 * returning (src/photon_grid.cu:82). */
rt_status rt_build_photon_grid(const rt_photon* photons, uint32_t count, float cellSize, uint32_t bucketCount,
                               uint32_t* counts, uint32_t* offsets, void* scan, size_t scanBytes, rt_photon* sorted);

/* Same work enqueued on `stream` without the final synchronisation. */
rt_status rt_build_photon_grid_async(const rt_photon* photons, uint32_t count, float cellSize, uint32_t bucketCount,
                                     uint32_t* counts, uint32_t* offsets, void* scan, size_t scanBytes,
                                     rt_photon* sorted, void* stream);

/* rt_build_photon_grid_async for a photon list whose length only exists in
 * device memory: the count is min(*countOnDevice, maxCount), read by the
 * kernels when they run, so the host neither waits for the photon tracer nor
 * reads the counter back before it enqueues the build (the reference does both,
 * src/renderer.cpp:753-769, and clamps the count to the capacity on the host,
 * :769; here the clamp is the min above).  Launches are sized for maxCount;
 * scan must hold rt_photon_grid_scratch_bytes(maxCount, bucketCount) or the
 * library's own workspace is used.  Same offsets and sorted list as a build
 * with the count on the host.  keysReady != 0: the caller has already written
 * the bucket key of photon i to rt_photon_grid_keys(scan, ...)[i] for every
 * i < count (rt_wavefront_set_photon_keys makes the photon tracer do so while
 * it stores the photon, which saves the pass that reads every record again
 * for its position); scan must then be the caller's. */
rt_status rt_build_photon_grid_counted(const rt_photon* photons, const uint32_t* countOnDevice, uint32_t maxCount,
                                       float cellSize, uint32_t bucketCount, uint32_t* offsets, void* scan,
                                       size_t scanBytes, rt_photon* sorted, int keysReady, void* stream);
/* Where a build over `scan` (sized for maxCount photons) expects the bucket keys of
 * its photons: maxCount uint32 inside the scratch, or null when the scratch is too small. */
uint32_t* rt_photon_grid_keys(void* scan, size_t scanBytes, uint32_t maxCount, uint32_t bucketCount);

/* rt_build_photon_grid_async for callers that cannot allocate device memory
 * (code under HIP graph capture, serving loops, harnesses that forbid
 * hipMalloc).  No counterpart in the reference.  The sort's scratch is a
 * static device array inside the library (about 16.2 MiB per GPU, held until
 * the process ends) sized for RT_PHOTON_GRID_STATIC_MAX_PHOTONS photons; a
 * larger count returns RT_ERROR_OUT_OF_RANGE.  The call allocates and frees
 * nothing.  Same results as rt_build_photon_grid_async (offsets: bucketCount
 * + 1 uint32, sorted: count photons, both caller-owned device memory; the
 * reference's `counts` array, which nothing reads, is not asked for).  The
 * first call on a device looks the array's address up: make it outside a
 * graph capture (count = 0 does).  Every call on a GPU shares that GPU's
 * array: calls that overlap in time on one GPU (two streams, two host
 * threads) race and must be ordered by the caller. */
#define RT_PHOTON_GRID_STATIC_MAX_PHOTONS (1u << 20)
rt_status rt_build_photon_grid_static(const rt_photon* photons, uint32_t count, float cellSize, uint32_t bucketCount,
                                      uint32_t* offsets, rt_photon* sorted, void* stream);

/* photonBucket(photonCell(p, cellSize), bucketCount) for n packed float3
 * device positions (src/photon_hash.cuh:10-24) -> keys (device, n uint32). */
rt_status rt_photon_keys(const float* positions, uint32_t n, float cellSize, uint32_t bucketCount, uint32_t* keys,
                         void* stream);

/* Radius gather, the loop of causticLighting() in src/device.cu:729-762.  For
 * each of n queries (packed float3 device arrays qpos, qnormal) writes the
 * flux sum over photons with d^2 < r^2 and dot(n_photon, n_query) >= 0.7,
 * weight 1 - d^2/r^2, visiting the 27 neighbour cells in the reference's
 * z, y, x order with repeated buckets skipped.  The caller applies
 * baseColor * diffuseWeight * 2 / (pi r^2 pi) (src/device.cu:763-765). */
rt_status rt_photon_gather(const rt_photon* sorted, const uint32_t* cellStart, uint32_t bucketCount, float radius,
                           const float* qpos, const float* qnormal, uint32_t n, float* flux, void* stream);

/* Same, and adds the number of photons read (the sum over queries of the sizes
 * of the distinct buckets visited) to *visitedPhotons (device). */
rt_status rt_photon_gather_counted(const rt_photon* sorted, const uint32_t* cellStart, uint32_t bucketCount,
                                   float radius, const float* qpos, const float* qnormal, uint32_t n, float* flux,
                                   uint64_t* visitedPhotons, void* stream);

/* One bit per bucket of a built grid, set where the bucket holds photons:
 * bucket b is bit (b & 31) of word (b >> 5); `bitmap` has
 * max(1, bucketCount / 32) words.  No counterpart upstream (device.cu:729-762
 * reads the range of every neighbour cell): a photon map much sparser than
 * its table leaves most buckets empty, the bits of a 2^22-bucket table stay
 * in the L2 cache where the table itself does not, and an empty bucket adds
 * nothing to a sum, so a gather that asks the bit first returns the same
 * floats. */
rt_status rt_photon_bucket_bitmap(const uint32_t* cellStart, uint32_t bucketCount, uint32_t* bitmap, void* stream);

/* rt_photon_gather that skips the buckets whose bit is clear. */
rt_status rt_photon_gather_masked(const rt_photon* sorted, const uint32_t* cellStart, const uint32_t* bitmap,
                                  uint32_t bucketCount, float radius, const float* qpos, const float* qnormal,
                                  uint32_t n, float* flux, void* stream);

/* ------------------------------------------------------------------------
 * Acceleration structures — replace the OptiX GAS / IAS builds of
 * src/renderer.cpp:343-396 and :398-434
 * ------------------------------------------------------------------------ */

typedef struct rt_blas rt_blas;   /* per-geometry BVH (GAS equivalent) */
typedef struct rt_scene rt_scene; /* top level over instances (IAS equivalent) */

/* optixAccelBuild over one triangle geometry (src/renderer.cpp:366-393):
 * float3 positions at `vertexStride` bytes (64 for rt_gpu_vertex) in a device
 * buffer, uint32 index triplets in a device buffer.  The buffers are read
 * during the build only.  Synchronises `stream` before returning. */
rt_status rt_blas_build(const void* vertices, uint32_t vertexCount, uint32_t vertexStride, const uint32_t* indices,
                        uint32_t triangleCount, void* stream, rt_blas** out);
void rt_blas_destroy(rt_blas* blas);

/* OptixInstance as filled in src/renderer.cpp:400-412 (host memory). */
typedef struct rt_instance_desc {
    float transform[12];   /* object->world, row-major 3x4 */
    uint32_t instanceId;   /* reported in hits */
    uint32_t geometry;     /* sbtOffset upstream: index of the geometry record */
    const rt_blas* blas;   /* traversableHandle upstream */
} rt_instance_desc;

/* optixAccelBuild over instances (src/renderer.cpp:414-432).  `instances` is
 * host memory.  The BLASes must outlive the scene.  Returns
 * RT_ERROR_OUT_OF_RANGE when the tree over the instances plus the deepest
 * bottom-level tree has more levels (18) than the traversal stack of the ray
 * kernels can follow: such a scene is refused here rather than traced with
 * dropped stack entries. */
rt_status rt_scene_build(const rt_instance_desc* instances, uint32_t instanceCount, void* stream, rt_scene** out);
void rt_scene_destroy(rt_scene* scene);
/* rt_scene_build puts the triangles of an instance into the tree over the instances themselves when the
 * instance's transform is the identity bit for bit and its mesh has at most 64 triangles (at most 4096 such
 * triangles per scene): a ray then tests them where it would have entered the instance.  The intersection
 * arithmetic and every reported hit (t, u, v, primitive, instance) are those of the instance; RT_SCENE_FOLD=0
 * in the environment folds nothing.  No counterpart in the reference (OptiX decides such things inside its
 * acceleration-structure build).  Reports how many instances and triangles of `scene` were folded. */
rt_status rt_scene_folded(const rt_scene* scene, uint32_t* instances, uint32_t* triangles);
/* The hit-group records of the shader binding table (src/renderer.cpp:649-660):
 * one rt_hit_group_data per geometry, indexed by rt_instance_desc.geometry.
 * `records` is host memory holding device pointers; the shading kernels read
 * vertices, indices and the material index through it.  The records must cover
 * every geometry number the scene's instances carry (RT_ERROR_OUT_OF_RANGE
 * otherwise).  Call again after changing a record. */
rt_status rt_scene_set_geometry(rt_scene* scene, const rt_hit_group_data* records, uint32_t count);
/* The 8-byte value stored in rt_launch_params.scene (src/gpu_shared.hpp:117). */
uint64_t rt_scene_handle(const rt_scene* scene);
/* world->object 3x4 (row-major) the engine uses for instance `index`. */
rt_status rt_scene_instance_inverse(const rt_scene* scene, uint32_t index, float* out12);

typedef struct rt_bvh_stats {
    uint64_t nodeCount;       /* wide nodes, all levels */
    uint64_t nodeBytes;
    uint64_t triangleCount;
    uint64_t triangleBytes;
    uint32_t instanceCount;
    uint32_t maxDepth;
    float buildMilliseconds;  /* device time of the last build */
} rt_bvh_stats;
rt_status rt_blas_stats(const rt_blas* blas, rt_bvh_stats* out);
rt_status rt_scene_stats(const rt_scene* scene, rt_bvh_stats* out);

/* ------------------------------------------------------------------------
 * Ray queries — replace trace() (src/device.cu:388-397) and the occlusion use
 * of it in visible() (src/device.cu:548-559)
 * ------------------------------------------------------------------------ */

/* Result of one closest-hit query: t, the barycentric weights of vertex 1 and
 * vertex 2 (optixGetTriangleBarycentrics, src/device.cu:1085), the
 * geometry-local primitive index (0xffffffff and t = -1 on a miss). */
typedef struct rt_ray_hit {
    float t, u, v;
    uint32_t primitive;
} rt_ray_hit;

/* The three queries below are enqueued on `stream` and return without waiting.
 * Any number of them may be in flight on any number of streams and host
 * threads: each launch draws its rays through a set of cursors of its own (the
 * scene keeps a ring of sets; a launch that is dealt a set still in use makes
 * its stream wait for the launch that holds it).  The scene and its BLASes must
 * stay alive until the launches have finished.
 *
 * n rays as two device arrays of float4: (origin, tmin) and (direction, tmax).
 * A hit has tmin < t <= tmax; the closest hit is the smallest t, exact ties
 * going to the lowest (instance, primitive).  A tmin below zero or NaN counts
 * as zero (the reference's traversal rejects such a ray as invalid).  instance[i] = instanceId or
 * 0xffffffff. */
rt_status rt_trace_closest(const rt_scene* scene, const rt_float4* originTmin, const rt_float4* directionTmax,
                           uint32_t n, rt_ray_hit* hits, uint32_t* instance, void* stream);

/* occluded[i] = 1 iff some triangle has tmin < t <= tmax. */
rt_status rt_trace_any(const rt_scene* scene, const rt_float4* originTmin, const rt_float4* directionTmax, uint32_t n,
                       uint8_t* occluded, void* stream);

/* Per-launch traversal counters accumulated by the *_counted variants:
 * [0] rays, [1] wide nodes visited, [2] triangles tested, [3] instances entered. */
rt_status rt_trace_closest_counted(const rt_scene* scene, const rt_float4* originTmin,
                                   const rt_float4* directionTmax, uint32_t n, rt_ray_hit* hits, uint32_t* instance,
                                   uint64_t* counters4, void* stream);

/* ------------------------------------------------------------------------
 * Wavefront pipeline — replaces the two optixLaunch calls of
 * src/renderer.cpp:748-750 (photon pass) and :790-791 (camera sample), i.e. the
 * ray-generation programs __raygen__photons and __raygen__render of
 * src/device.cu:768-861 and :863-1075
 * ------------------------------------------------------------------------ */

typedef struct rt_wavefront rt_wavefront; /* ray queues and per-path state for up to maxPaths paths */
rt_status rt_wavefront_create(uint32_t maxPaths, rt_wavefront** out);
void rt_wavefront_destroy(rt_wavefront* wavefront);
/* A state for one kind of pass only leaves out the arrays the other kinds need
 * (a camera pass has no use for the per-emission photon slots, a photon pass
 * none for shadow rays, sample sums and photon queries, a camera pass that
 * never reads a photon map none for the query queues and the caustic sum:
 * 84 of its 356 bytes per path); rt_wavefront_create is all of them.  rt_wavefront_state_bytes: the device memory such a state takes,
 * from the same list of arrays the allocation walks - what a caller sizes a
 * pass against free memory with.  No counterpart upstream (OptiX keeps the
 * state of a launch in its own stack and registers). */
#define RT_WAVEFRONT_FOR_CAMERA 1u  /* rt_render_sample*, with or without a photon map */
#define RT_WAVEFRONT_FOR_PHOTONS 2u /* rt_trace_photons */
#define RT_WAVEFRONT_FOR_CAMERA_NO_CAUSTICS 4u /* rt_render_sample* with params->photonRadius == 0 only */
rt_status rt_wavefront_create_for(uint32_t maxPaths, uint32_t uses, rt_wavefront** out);
rt_status rt_wavefront_state_bytes(uint32_t maxPaths, uint32_t uses, size_t* bytes);

/* One camera sample (params->sample) for the pixels [pixelBegin, pixelEnd) in
 * row-major order, row 0 at the bottom: generate, then maxDepth rounds of extend
 * (closest hit) / shade / shadow, then the running-mean accumulation into the
 * seven float4 buffers and the tone-mapped byte image of `params`
 * (src/device.cu:1043-1074).  `params` is host memory; every pointer in it is a
 * device pointer and params->scene is rt_scene_handle() of a scene that has its
 * geometry records set.  pixelEnd - pixelBegin must not exceed maxPaths.
 * Enqueued on `stream`; returns without waiting (src/renderer.cpp:790-793). */
rt_status rt_render_sample(rt_wavefront* wavefront, const rt_launch_params* params, uint32_t pixelBegin,
                           uint32_t pixelEnd, void* stream);

/* Same, with the sample index that seeds the random streams (src/device.cu:882)
 * given apart from params->sample, which then only sets the running-mean weight
 * 1 / (sample + 1): lets several GPUs each average their own share of the
 * sample indices of one image. */
rt_status rt_render_sample_seeded(rt_wavefront* wavefront, const rt_launch_params* params, uint32_t pixelBegin,
                                  uint32_t pixelEnd, uint32_t seedSample, void* stream);

/* One camera sample for every bandStride-th band of eight pixel rows, starting
 * with band firstBand (rows 8 * firstBand ..), bandCount bands at most; bands
 * beyond the image are left out and the last band of an image whose height is
 * no multiple of 8 is shorter.  With bandStride = 1 this is a contiguous block
 * of rows; with bandStride = N and firstBand = r it is the share of GPU r of N
 * when the bands of one frame are dealt out in turn (no counterpart upstream,
 * which renders a frame with one launch, src/renderer.cpp:790-791).  A pixel's
 * random streams depend on its own index and the sample only
 * (src/device.cu:882), so the buffers equal those of rt_render_sample over the
 * same pixels bit for bit.  The pixels covered must not exceed maxPaths. */
rt_status rt_render_sample_bands(rt_wavefront* wavefront, const rt_launch_params* params, uint32_t firstBand,
                                 uint32_t bandStride, uint32_t bandCount, uint32_t seedSample, void* stream);

/* The photon pass for emission indices [firstEmission, firstEmission +
 * emissionCount) of pass params->photonPass: photons are appended to
 * params->photons behind *params->photonCount in emission order (the reference
 * appends in atomic order, src/device.cu:716), and *params->photonCount grows by
 * the number of photons the tracer wanted to store, which can exceed
 * params->photonCapacity as upstream (src/renderer.cpp:755-769).  The caller
 * zeroes *params->photonCount before the first range of a pass
 * (src/renderer.cpp:743).  Enqueued on `stream`. */
rt_status rt_trace_photons(rt_wavefront* wavefront, const rt_launch_params* params, uint32_t firstEmission,
                           uint32_t emissionCount, void* stream);

/* rt_trace_photons for `passes` consecutive passes (params->photonPass, + 1, ...)
 * of the same emission range in one go: the passes of the photon maps a long
 * camera pass will gather from share every stage launch, whose cost at the late
 * bounces is the same few slow photons however many passes ride along.  Pass k
 * appends to lists[k] behind *counters[k] (device words the caller has zeroed)
 * and, with keys and keys[k], writes bucket keys to keys[k]; all are host
 * arrays of `passes` device pointers.  params->photons / photonCount are not
 * used.  Photon i of pass k draws the streams seeded (i, photonPass + k)
 * (src/device.cu:772), so every list equals that of a pass traced alone.  As many
 * passes as the state holds (maxPaths / emissionCount) run side by side. */
#define RT_MAX_PHOTON_PASSES 16
rt_status rt_trace_photon_passes(rt_wavefront* wavefront, const rt_launch_params* params, uint32_t firstEmission,
                                 uint32_t emissionCount, uint32_t passes, rt_photon* const* lists,
                                 uint32_t* const* counters, uint32_t* const* keys, void* stream);

/* Later rt_trace_photons calls on this state also write the bucket key of every
 * photon they store - photonBucket(photonCell(position, params->photonRadius),
 * params->photonBucketCount), src/photon_hash.cuh:10-24 - to keys[slot], slot
 * being the photon's index in params->photons; keys holds params->photonCapacity
 * uint32 (device).  Null (the default) switches it off.  Meant for
 * rt_build_photon_grid_counted(keysReady = 1) with cellSize = params->photonRadius. */
rt_status rt_wavefront_set_photon_keys(rt_wavefront* wavefront, uint32_t* keys);

/* Copies the running counters to out8 (host): [0] closest-hit (extend) rays,
 * [1] shadow rays, [2] caustic gather queries, [3] photons read by them,
 * [4] wide nodes visited, [5] triangles tested, [6] instances entered by the
 * ray launches (both kinds of ray) made while RT_WAVEFRONT_COUNT_TRAVERSAL was
 * set, [7] node / triangle / instance references the ray stage refused because
 * they pointed outside their array (0 unless the acceleration structure is
 * damaged; such a ray ends as a miss instead of faulting).
 * Synchronises `stream`; reset != 0 zeroes the counters afterwards. */
rt_status rt_wavefront_stats(rt_wavefront* wavefront, uint64_t* out8, int reset, void* stream);

/* Measurement switches (all off by default). */
#define RT_WAVEFRONT_TIME_STAGES 1u      /* bracket every stage launch of rt_render_sample with HIP events */
#define RT_WAVEFRONT_COUNT_TRAVERSAL 2u  /* ray launches count visited nodes / tested triangles */
#define RT_WAVEFRONT_GATHER_PER_LANE 4u   /* photon queries that read many photons stay in the lane that drew them
                                          * instead of going to a wave each: the same floats, for comparison */
rt_status rt_wavefront_configure(rt_wavefront* wavefront, uint32_t flags);
/* Share of the GPU's wave slots a ray launch of this state occupies, as
 * numerator / denominator of the resident set (default 1/1).  The ray stage
 * runs in persistent waves that stay until the launch has drained, so two
 * states rendering parts of a frame on two streams only run side by side when
 * each leaves room for the other; rt::Renderer::render() sets 1/2 when it
 * splits a frame (no counterpart upstream: the reference renders a frame
 * with one launch, renderer.cpp:779-796). */
rt_status rt_wavefront_set_ray_share(rt_wavefront* wavefront, uint32_t numerator, uint32_t denominator);
/* Enqueues an empty launch with the private segment and the grid of this
 * state's ray launches on `stream`.  The ray kernels keep
 * the deep end of their traversal stacks in private memory, and the first
 * launch of such a kernel on a hardware queue waits for that queue's scratch
 * memory to be set up (about 130 us on an MI355X, once per queue and process):
 * a caller that creates its streams ahead of time pays for it there instead of
 * in the middle of its first pass on each of them.  No counterpart upstream. */
rt_status rt_wavefront_prepare_stream(rt_wavefront* wavefront, void* stream);
/* Later rt_render_sample* calls on this state render `samples` consecutive
 * samples of their pixels in one pass: sample params->sample + k is seeded
 * with seedSample + k * seedStride, k < samples, and the samples enter the
 * running means in that order, so the buffers end up as after `samples` calls
 * of one sample each (src/device.cu:1043-1046).  The state must hold pixels x
 * samples paths.  One pass over four samples of a small pixel range (one
 * GPU's share of a frame that several GPUs cut up) costs little more than a
 * pass over one, because a pass is as long as its slowest rays.  The default
 * is (1, 1).  No counterpart upstream. */
rt_status rt_wavefront_set_sample_batch(rt_wavefront* wavefront, uint32_t samples, uint32_t seedStride);
/* Orders the accumulation step of later rt_render_sample* calls on this state
 * against other streams: before the running means are updated the stream
 * waits for the HIP event `waitFor`, and behind the update it records the HIP
 * event `record` (hipEvent_t passed as void*; either may be null, which is the
 * default).  The running mean of a pixel must take its samples in order
 * (src/device.cu:1043-1046), so a caller that keeps several samples of one
 * region in flight on several streams - which hides the slow last rays of one
 * sample's launches behind the next sample's work - chains the accumulation
 * steps of that region with these events; everything before the accumulation
 * of a sample is independent of other samples.  No counterpart upstream. */
rt_status rt_wavefront_set_accumulate_events(rt_wavefront* wavefront, void* waitFor, void* record);
/* The photon maps later rt_render_sample* calls on this state gather from, for
 * passes of more samples than one map serves.  The reference traces a new map
 * before every fourth sample (src/renderer.cpp:177, 780-781), so samples 4e ..
 * 4e + 3 read map e; a pass over samples s0 .. s0 + n - 1 (rt_wavefront_set_
 * sample_batch) is given the maps of the groups it covers, in order, with
 * samplesPerMap = 4 and phase = s0 % 4, and sample s0 + k gathers from map
 * (phase + k) / samplesPerMap - the image is that of one pass per sample with
 * the map rebuilt in between.  All maps share params->photonBucketCount and
 * params->photonRadius; bucketBits may be null (every range is read).  count =
 * 0 (the default): the one map of the launch parameters and rt_wavefront_set_
 * bucket_bitmap.  The table is copied; the maps must stay as they are until
 * the passes that read them have finished. */
#define RT_MAX_PHOTON_MAPS 16
typedef struct rt_photon_map {
    const rt_photon* sortedPhotons; /* rt_launch_params.sortedPhotons of that map */
    const uint32_t* cellStart;      /* rt_launch_params.photonCellStart */
    const uint32_t* bucketBits;     /* rt_photon_bucket_bitmap of it, or null */
} rt_photon_map;
rt_status rt_wavefront_set_photon_maps(rt_wavefront* wavefront, const rt_photon_map* maps, uint32_t count,
                                       uint32_t samplesPerMap, uint32_t phase);

/* Bucket bits (rt_photon_bucket_bitmap) of the photon grid the launch
 * parameters of later rt_render_sample calls point at, or null (the default)
 * to read every range.  The launch parameters keep the reference's layout
 * (gpu_shared.hpp:95-145), which has no room for the pointer.  The caller
 * keeps the bits in step with the grid. */
rt_status rt_wavefront_set_bucket_bitmap(rt_wavefront* wavefront, const uint32_t* bitmap);
/* ---- light index ------------------------------------------------------------
 * The reference picks a light by subtracting the weights of the light list
 * from a random target until the remainder is no longer positive
 * (src/device.cu:433-447) and finds the light of an emissive triangle a path
 * hit by comparing every light with the hit (src/device.cu:607-621): a walk
 * over the whole list per shaded point.  The index holds, per light, the
 * largest float target for which that walk has stopped at the light or before
 * it - found by running the walk's rounded subtractions backwards, so a binary
 * search over these thresholds returns the walk's light for every target, bit
 * for bit - and a hash table from (instance, primitive) to the first light of
 * that triangle.  Optional: without it the kernels walk the list.  Lists of
 * more than 2^20 lights, or with a weight that is negative or not finite, get
 * the table only (the thresholds take lightCount^2 / 2 steps to build: 35 ms
 * for 2^18 lights on an MI355X).
 * rt_light_index_bytes: size of the caller-owned device memory an index for
 * lightCount lights takes (8-byte aligned). */
rt_status rt_light_index_bytes(uint32_t lightCount, size_t* bytes);
/* Builds the index of `lights` (device, the array rt_launch_params.lights
 * points at) into `index`.  Set-up work: the stream is waited for, the index is
 * complete when the call returns. */
rt_status rt_light_index_build(const rt_gpu_light* lights, uint32_t lightCount, void* index, size_t indexBytes,
                               void* stream);
/* The index of the light list the launch parameters of later rt_render_sample*
 * and rt_trace_photons calls on this state point at, or null (the default) to
 * walk the list.  The launch parameters keep the reference's layout
 * (gpu_shared.hpp:95-145), which has no room for the pointer. */
rt_status rt_wavefront_set_light_index(rt_wavefront* wavefront, const void* index, uint32_t lightCount);
/* The two questions of the reference's light code for n inputs at once (all
 * arrays device): chosen[i] = the light src/device.cu:433-447 selects for the
 * uniform number uniform[i] (target = uniform[i] * lightWeight); found[i] = the
 * light src/device.cu:607-621 finds for the triangle (instance[i],
 * primitive[i]), or 0xffffffff.  Through `index`, or by the walks themselves
 * when `index` is null.  Either pair of arrays may be null. */
rt_status rt_light_query(const rt_gpu_light* lights, uint32_t lightCount, float lightWeight, const void* index,
                         const float* uniform, const uint32_t* instance, const uint32_t* primitive, uint32_t n,
                         uint32_t* chosen, uint32_t* found, void* stream);

/* The elementary functions of the shading kernels (include/rt_math.h: sequences of
 * correctly rounded binary64 operations standing in for the sinf / cosf / powf /
 * atan2f / acosf / exp2f calls of src/device.cu) evaluated on the device for n
 * inputs: which = 0 sine(x), 1 cosine(x), 2 power(x, y), 3 arcTangent2(x, y) with x
 * the ordinate, 4 arcCosine(x), 5 exp2(x).  All arrays device; y only for 2 and 3.
 * A host build of the same header gives the same bits, which is what the tests
 * use this entry for. */
rt_status rt_shared_math(int which, const float* x, const float* y, uint32_t n, float* out, void* stream);

/* Whether kernels put on these HIP streams (at most 16) really run side by
 * side: a process has a limited number of hardware queues, streams beyond
 * that number share one, and work on two streams of one queue runs in turn.
 * Starts one 200-microsecond single-wave kernel per stream and times the lot
 * (synchronises the device).  rt::Renderer uses it to pick streams for the
 * parts of a split frame (no counterpart upstream). */
rt_status rt_streams_run_concurrently(void* const* streams, uint32_t count, int* concurrent);
/* Sums the event-timed launches since the last call per stage: [0] generate,
 * [1] trace (the shadow rays of a bounce and the closest-hit rays of the next
 * share one launch), [2] shade, [3] queue bookkeeping between bounces,
 * [4] accumulate.  Synchronises `stream`. */
rt_status rt_wavefront_timings(rt_wavefront* wavefront, float* milliseconds5, uint32_t* launches5, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RT_ENGINE_H */
