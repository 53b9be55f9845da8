/*
 * CPU oracle for the ray-intersection + photon-gather path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (raytracer_amd/, include/)
 * includes, links or calls this library; only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg load it, and only as the checker / as the
 * timed CPU baseline.
 *
 * It is a scalar, IEEE-754 (no contraction, no fast-math) restatement of the
 * reference's algorithms, each function citing the reference file:line it
 * follows.  Pinning status:
 *   - photon hash: pinned by the four known-answer vectors produced from the
 *     reference's own src/photon_hash.cuh (tests/golden/photon_hash_kat.json).
 *   - everything else: PARITY UNPINNED.  The reference ships no tests, golden
 *     vectors or fixtures, has no CPU path, and its traversal lives in the
 *     closed OptiX 9 driver, so none of it can be built or run here.
 */
#ifndef ORACLE_H
#define ORACLE_H

#include <stddef.h>
#include <stdint.h>

#include "rt_types.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- photon hash (photon_hash.cuh:10-24) ------------------------------- */
void orc_photon_cell(const float* position3, float cellSize, int32_t* cell3);
uint32_t orc_photon_bucket(const int32_t* cell3, uint32_t bucketCount);
/* keys[i] = bucket(cell(positions[i])) for n packed float3 positions */
// include/rt_math.h over arrays: which = 0 sine, 1 cosine, 2 power(x, y), 3 arcTangent2(x, y), 4 arcCosine, 5 exp2.
void orc_shared_math(int which, const float* x, const float* y, uint32_t n, float* out);
void orc_photon_keys(const float* positions, uint32_t n, float cellSize, uint32_t bucketCount, uint32_t* keys);

/* ---- photon grid (photon_grid.cu:17-83) --------------------------------
 * Counting sort of photons into hash buckets.  offsets has bucketCount+1
 * entries; offsets[b]..offsets[b+1] delimits bucket b inside sorted.  Order
 * inside a bucket is ascending input index (the reference leaves it to the
 * atomic race; this is the canonical order).  count == 0 is legal. */
void orc_build_photon_grid(const rt_photon* photons, uint32_t count, float cellSize, uint32_t bucketCount,
                           uint32_t* offsets, rt_photon* sorted);

/* ---- photon gather (device.cu:722-766) ---------------------------------
 * flux[q] = sum over the de-duplicated 27 neighbour buckets of
 * power * (1 - d^2/r^2) for photons with d^2 < r^2 and dot(n_p, n_q) >= 0.7,
 * accumulated in the reference's z,y,x / bucket order.  The factor
 * baseColor * diffuseWeight * 2/(pi r^2 pi) is applied by the caller.
 * threads > 1 splits the query range over std::thread workers. */
void orc_photon_gather(const rt_photon* sorted, const uint32_t* cellStart, uint32_t bucketCount, float radius,
                       const float* qpos, const float* qnormal, uint32_t n, float* flux, int threads);

/* ---- RNG (device.cu:220-233) ------------------------------------------- */
uint32_t orc_seeded(uint32_t a, uint32_t b);
float orc_random(uint32_t* state);

/* ---- scene + ray queries (device.cu:388-397, 548-559, 1077-1145;
 *      renderer.cpp:343-434) --------------------------------------------- */
typedef struct orc_scene orc_scene;
orc_scene* orc_scene_create(void);
void orc_scene_destroy(orc_scene*);
/* vertices are rt_gpu_vertex (64-byte stride); returns the geometry index */
int orc_scene_add_geometry(orc_scene*, const rt_gpu_vertex* vertices, uint32_t vertexCount,
                           const uint32_t* indices, uint32_t triangleCount, uint32_t material);
/* transform16: column-major 4x4 as in scene.hpp:31-33; returns the instance index */
int orc_scene_add_instance(orc_scene*, uint32_t geometry, const float* transform16);
/* computes world->object transforms and the CPU BVHs */
void orc_scene_commit(orc_scene*);
/* world->object 3x4 row-major of instance i, as used by the ray queries */
void orc_scene_inverse(const orc_scene*, uint32_t instance, float* out12);

/* Closest hit of n rays; org_tmin / dir_tmax are n float4 (ox,oy,oz,tmin) /
 * (dx,dy,dz,tmax).  A hit has tmin < t <= tmax; the closest is min t with
 * exact ties resolved to the lowest (instance, primitive).  Outputs: tuv (n
 * float3: t, u = weight of vertex 1, v = weight of vertex 2), prim, inst
 * (0xffffffff on a miss, t = -1).  use_bvh = 0 walks every triangle of every
 * instance; 1 uses the oracle's own binned-SAH BVH (same result). */
void orc_trace_closest(const orc_scene*, const float* org_tmin, const float* dir_tmax, uint32_t n, int use_bvh,
                       int threads, float* tuv, uint32_t* prim, uint32_t* inst);
/* occluded[i] = 1 iff any triangle has tmin < t <= tmax */
void orc_trace_any(const orc_scene*, const float* org_tmin, const float* dir_tmax, uint32_t n, int use_bvh,
                   int threads, uint8_t* occluded);
/* Attribute reconstruction at the hit, device.cu:1081-1145 (miss: found = 0,
 * everything else zero as in device.cu:390, 1077-1079). */
void orc_shade_hits(const orc_scene*, const float* org_tmin, const float* dir_tmax, const float* tuv,
                    const uint32_t* prim, const uint32_t* inst, uint32_t n, rt_hit* hits);

/* ---- integrator (device.cu:863-1075) and photon tracer (device.cu:768-861) --
 * Every pointer inside `params` is a HOST pointer here; params->scene is
 * ignored.  orc_render_sample runs sample params->sample for the pixels
 * [pixelBegin, pixelEnd) and updates the seven accumulation buffers and the
 * byte image exactly as device.cu:1043-1074 does. */
void orc_render_sample(const orc_scene* scene, const rt_launch_params* params, uint32_t pixelBegin, uint32_t pixelEnd,
                       int threads);
/* Emission indices [first, first + count) of pass params->photonPass; stored
 * photons go to `out` in emission order (at most `capacity`); returns how many
 * the tracer wanted to store. */
uint32_t orc_trace_photons(const orc_scene* scene, const rt_launch_params* params, uint32_t first, uint32_t count,
                           rt_photon* out, uint32_t capacity, int threads);
/* Same, and emission[k] = the emission index that stored photon k. */
uint32_t orc_trace_photons_indexed(const orc_scene* scene, const rt_launch_params* params, uint32_t first,
                                   uint32_t count, rt_photon* out, uint32_t* emission, uint32_t capacity, int threads);

/* The two walks of the reference's light code over `lights` for n inputs:
 * chosen[i] = the light device.cu:433-447 selects for the uniform number
 * uniform[i]; found[i] = the light device.cu:607-621 finds for the triangle
 * (instance[i], primitive[i]), 0xffffffff when there is none.  Either pair of
 * arrays may be null. */
void orc_light_walks(const rt_gpu_light* lights, uint32_t lightCount, float lightWeight, const float* uniform,
                     const uint32_t* instance, const uint32_t* primitive, uint32_t n, uint32_t* chosen, uint32_t* found,
                     int threads);

/* Rays traced by orc_render_sample / orc_trace_photons since the last reset
 * (for timing the oracle as the CPU baseline). */
uint64_t orc_ray_counter(int reset);

#ifdef __cplusplus
}
#endif
#endif
