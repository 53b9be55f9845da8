/*
 * Plain-old-data contract between the host runtime and the HIP device code.
 *
 * Every struct below has the same size, alignment and member offsets as the
 * struct of the same role in the reference's src/gpu_shared.hpp:8-145 (where
 * the members are CUDA vector types: float2 -> align 8, float4 -> align 16,
 * float3 -> align 4, uchar4 -> align 4).  The vector types are repo-owned so
 * that the header compiles with gcc, g++ and hipcc alike; LaunchParams.scene
 * (an OptixTraversableHandle upstream) is an opaque 64-bit scene handle here.
 *
 * The static assertions at the bottom pin the layout.
 */
#ifndef RT_TYPES_H
#define RT_TYPES_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
#define RT_ALIGNAS(n) alignas(n)
#define RT_STATIC_ASSERT(c, m) static_assert(c, m)
extern "C" {
#else
#define RT_ALIGNAS(n) _Alignas(n)
#define RT_STATIC_ASSERT(c, m) _Static_assert(c, m)
#endif

typedef struct rt_float2 { RT_ALIGNAS(8) float x; float y; } rt_float2;
typedef struct rt_float3 { float x, y, z; } rt_float3;
typedef struct rt_float4 { RT_ALIGNAS(16) float x; float y, z, w; } rt_float4;
typedef struct rt_uchar4 { RT_ALIGNAS(4) unsigned char x; unsigned char y, z, w; } rt_uchar4;
typedef struct rt_uint3 { uint32_t x, y, z; } rt_uint3;
typedef struct rt_int3 { int32_t x, y, z; } rt_int3;

/* gpu_shared.hpp:8-14.  BVH input positions are read at this 64-byte stride
 * (renderer.cpp:371-372). */
typedef struct rt_gpu_vertex {
    rt_float3 position;
    rt_float3 normal;
    rt_float4 tangent;
    rt_float2 uv;
    rt_float2 uv1;
} rt_gpu_vertex;

/* gpu_shared.hpp:16-23 */
typedef struct rt_gpu_texture_ref {
    int32_t texture;
    uint32_t texCoord;
    rt_float2 offset;
    rt_float2 scale;
    float rotation;
    float strength;
} rt_gpu_texture_ref;

/* gpu_shared.hpp:25-43 */
typedef struct rt_gpu_material {
    rt_float4 baseColor;
    rt_float3 emissive;
    rt_float3 attenuationColor;
    float metallic;
    float roughness;
    float transmission;
    float ior;
    float thickness;
    float attenuationDistance;
    float emissiveStrength;
    float dispersion;
    rt_gpu_texture_ref baseColorTexture;
    rt_gpu_texture_ref metallicRoughnessTexture;
    rt_gpu_texture_ref normalTexture;
    rt_gpu_texture_ref emissiveTexture;
    rt_gpu_texture_ref transmissionTexture;
    rt_gpu_texture_ref thicknessTexture;
} rt_gpu_material;

/* gpu_shared.hpp:45-51 */
typedef struct rt_gpu_texture {
    const rt_uchar4* pixels;
    uint32_t width;
    uint32_t height;
    uint32_t wrapU;
    uint32_t wrapV;
} rt_gpu_texture;

/* gpu_shared.hpp:53-65.  `primitive` is the geometry-local triangle index,
 * `instance` indexes Scene::instances. */
typedef struct rt_hit {
    rt_float3 position;
    rt_float3 normal;
    rt_float4 tangent;
    rt_float2 uv;
    rt_float2 uv1;
    float distance;
    uint32_t material;
    uint32_t instance;
    uint32_t primitive;
    uint8_t frontFace; /* bool upstream */
    uint8_t found;     /* bool upstream */
} rt_hit;

/* gpu_shared.hpp:67-71.  One record per geometry, selected through
 * instance.geometry (renderer.cpp:408, 649-659). */
typedef struct rt_hit_group_data {
    const rt_gpu_vertex* vertices;
    const rt_uint3* indices;
    uint32_t material;
} rt_hit_group_data;

/* gpu_shared.hpp:73-87.  type: 0 directional, 1 spot, 2 point, 3 emissive
 * triangle, 4 environment (renderer.cpp:470, 489, 495). */
typedef struct rt_gpu_light {
    rt_float3 a;
    rt_float3 b;
    rt_float3 c;
    rt_float3 normal;
    rt_float3 emission;
    float area;
    float weight;
    float range;
    float innerCone;
    float outerCone;
    uint32_t instance;
    uint32_t primitive;
    uint32_t type;
} rt_gpu_light;

/* gpu_shared.hpp:89-93 */
typedef struct rt_photon {
    rt_float3 position;
    rt_float3 power;
    rt_float3 normal;
} rt_photon;

/* gpu_shared.hpp:95-145 */
typedef struct rt_launch_params {
    rt_uchar4* output;
    rt_float4* accumulation;
    rt_float4* diffuse;
    rt_float4* reflection;
    rt_float4* refraction;
    rt_float4* caustics;
    rt_float4* albedoGuide;
    rt_float4* normalGuide;
    const rt_float4* display;
    const rt_float4* display2;
    const rt_float4* display3;
    rt_float4* composite;
    const rt_gpu_material* materials;
    const rt_gpu_texture* textures;
    const rt_gpu_light* lights;
    const rt_float4* environment;
    const float* environmentCdf;
    rt_photon* photons;
    uint32_t* photonCount;
    const rt_photon* sortedPhotons;
    const uint32_t* photonCellStart;
    uint64_t scene; /* opaque scene handle (rt_scene_handle) */
    rt_float3 eye;
    rt_float3 cameraU;
    rt_float3 cameraV;
    rt_float3 cameraW;
    rt_float3 lensU;
    rt_float3 lensV;
    float aperture;
    float focusDistance;
    float lightWeight;
    float environmentWeight;
    float environmentRotation;
    float environmentStrength;
    float exposure;
    rt_float3 sceneCenter;
    float sceneRadius;
    float photonRadius;
    uint32_t lightCount;
    uint32_t environmentWidth;
    uint32_t environmentHeight;
    uint32_t width;
    uint32_t height;
    uint32_t sample;
    uint32_t maxDepth;
    uint32_t photonBucketCount;
    uint32_t photonCapacity;
    uint32_t photonEmissions;
    uint32_t photonPass;
} rt_launch_params;

#ifdef __cplusplus
}
#endif

RT_STATIC_ASSERT(sizeof(rt_float2) == 8 && sizeof(rt_float3) == 12 && sizeof(rt_float4) == 16, "vector sizes");
RT_STATIC_ASSERT(sizeof(rt_gpu_vertex) == 64, "GpuVertex is 64 bytes");
RT_STATIC_ASSERT(offsetof(rt_gpu_vertex, normal) == 12 && offsetof(rt_gpu_vertex, tangent) == 32 &&
                 offsetof(rt_gpu_vertex, uv) == 48 && offsetof(rt_gpu_vertex, uv1) == 56, "GpuVertex offsets");
RT_STATIC_ASSERT(sizeof(rt_gpu_texture_ref) == 32, "GpuTextureRef is 32 bytes");
RT_STATIC_ASSERT(sizeof(rt_gpu_material) == 272, "GpuMaterial is 272 bytes");
RT_STATIC_ASSERT(offsetof(rt_gpu_material, emissive) == 16 && offsetof(rt_gpu_material, metallic) == 40 &&
                 offsetof(rt_gpu_material, baseColorTexture) == 72, "GpuMaterial offsets");
RT_STATIC_ASSERT(sizeof(rt_gpu_texture) == 24, "GpuTexture is 24 bytes");
RT_STATIC_ASSERT(sizeof(rt_hit) == 96, "Hit is 96 bytes");
RT_STATIC_ASSERT(offsetof(rt_hit, normal) == 12 && offsetof(rt_hit, tangent) == 32 && offsetof(rt_hit, uv) == 48 &&
                 offsetof(rt_hit, uv1) == 56 && offsetof(rt_hit, distance) == 64 && offsetof(rt_hit, material) == 68 &&
                 offsetof(rt_hit, instance) == 72 && offsetof(rt_hit, primitive) == 76 &&
                 offsetof(rt_hit, frontFace) == 80 && offsetof(rt_hit, found) == 81, "Hit offsets");
RT_STATIC_ASSERT(sizeof(rt_hit_group_data) == 24, "HitGroupData is 24 bytes");
RT_STATIC_ASSERT(sizeof(rt_gpu_light) == 92, "GpuLight is 92 bytes");
RT_STATIC_ASSERT(offsetof(rt_gpu_light, normal) == 36 && offsetof(rt_gpu_light, emission) == 48 &&
                 offsetof(rt_gpu_light, area) == 60 && offsetof(rt_gpu_light, weight) == 64 &&
                 offsetof(rt_gpu_light, range) == 68 && offsetof(rt_gpu_light, instance) == 80 &&
                 offsetof(rt_gpu_light, type) == 88, "GpuLight offsets");
RT_STATIC_ASSERT(sizeof(rt_photon) == 36, "Photon is 36 bytes");
RT_STATIC_ASSERT(sizeof(rt_launch_params) == 344, "LaunchParams is 344 bytes");
RT_STATIC_ASSERT(offsetof(rt_launch_params, scene) == 168 && offsetof(rt_launch_params, eye) == 176 &&
                 offsetof(rt_launch_params, sceneCenter) == 276 && offsetof(rt_launch_params, photonRadius) == 292 &&
                 offsetof(rt_launch_params, lightCount) == 296 && offsetof(rt_launch_params, photonPass) == 336,
                 "LaunchParams offsets");

#endif /* RT_TYPES_H */
