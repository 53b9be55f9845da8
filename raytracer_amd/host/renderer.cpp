// rt::Renderer on top of the engine's C ABI (include/rt_engine.h).
//
// The class does what the reference's src/renderer.cpp does between a loaded rt::Scene and
// the launches - tables for the device, acceleration structures, light list, photon map
// constants, camera frame, the sample and photon-pass loops, read-backs - organised for this
// engine rather than for OptiX:
//
//   * device memory comes from three pools (scene data, frame buffers, photon map), each one
//     hipMalloc; the scene pool is filled on the host and uploaded with a single copy, so a
//     scene of thousands of meshes and textures costs one allocation and one transfer
//     instead of one of each per table;
//   * the host-side arithmetic that ends up in device tables (light weights, environment
//     distribution, scene bounds, camera frame, caustic radius) sits in free functions that
//     evaluate exactly the expressions of the reference, operation by operation, because the
//     integrator's random decisions depend on the resulting floats; each cites its lines;
//   * a frame is rendered as bands of eight pixel rows: cut into parts that run side by side
//     on streams of their own, and, when several GPUs share a frame, dealt out in turn.
#include "renderer.hpp"

#include "photon_grid.hpp"
#include "rt_engine.h"

#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <stdexcept>
#include <string>

namespace rt {
namespace {

// ---- failures ------------------------------------------------------------------------------------

[[noreturn]] void raise(const std::string& message) { throw std::runtime_error(message); }

// The reference reports the runtime's own error string (renderer.cpp:30-33).
void hipOk(hipError_t status) {
    if (status != hipSuccess)
        raise(hipGetErrorString(status));
}

void engineOk(rt_status status, const char* call) {
    if (status != RT_SUCCESS)
        raise(std::string(call) + ": " + rt_last_error());
}

// ---- device memory -------------------------------------------------------------------------------

constexpr std::size_t kBlockAlignment = 256;

// Hands out aligned offsets inside a pool that does not exist yet.
class Layout {
public:
    std::size_t take(std::size_t bytes) {
        const std::size_t offset = (end_ + kBlockAlignment - 1) / kBlockAlignment * kBlockAlignment;
        end_ = offset + bytes;
        return offset;
    }
    std::size_t bytes() const { return end_; }

private:
    std::size_t end_ = 0;
};

// One device allocation that tables are carved out of.
class DevicePool {
public:
    DevicePool() = default;
    DevicePool(const DevicePool&) = delete;
    DevicePool& operator=(const DevicePool&) = delete;
    ~DevicePool() { release(); }

    void allocate(std::size_t bytes) {
        release();
        if (bytes == 0)
            return;
        void* memory = nullptr;
        hipOk(hipMalloc(&memory, bytes));
        base_ = static_cast<unsigned char*>(memory);
        bytes_ = bytes;
    }
    void release() {
        if (base_)
            (void)hipFree(base_);
        base_ = nullptr;
        bytes_ = 0;
    }
    template <typename T>
    T* at(std::size_t offset) const {
        return reinterpret_cast<T*>(base_ + offset);
    }
    std::size_t bytes() const { return bytes_; }

private:
    unsigned char* base_ = nullptr;
    std::size_t bytes_ = 0;
};

// Host image of a pool: blocks are appended at the offsets a Layout hands out, pointers
// between blocks are written once the pool's address is known, then everything goes up in one copy.
class Staging {
public:
    std::size_t append(const void* data, std::size_t bytes) {
        const std::size_t offset = layout_.take(bytes);
        image_.resize(layout_.bytes());
        if (bytes)
            std::memcpy(image_.data() + offset, data, bytes);
        return offset;
    }
    template <typename T>
    std::size_t reserve(std::size_t count) {
        const std::size_t offset = layout_.take(count * sizeof(T));
        image_.resize(layout_.bytes());
        return offset;
    }
    template <typename T>
    T* at(std::size_t offset) {
        return reinterpret_cast<T*>(image_.data() + offset);
    }
    std::size_t bytes() const { return image_.size(); }

private:
    Layout layout_;
    std::vector<unsigned char> image_;
};

// ---- float3 arithmetic in the reference's evaluation order ---------------------------------------

struct F3 {
    float x, y, z;
};

inline rt_float3 packed(F3 v) { return rt_float3{v.x, v.y, v.z}; }
inline F3 operator-(F3 a, F3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline F3 scaled(F3 v, float k) { return {v.x * k, v.y * k, v.z * k}; }
inline float norm(F3 v) { return std::sqrt(v.x * v.x + v.y * v.y + v.z * v.z); }
inline F3 cross(F3 a, F3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
// reciprocal length first, then three products (renderer.cpp:103-106)
inline F3 unit(float x, float y, float z) { return scaled(F3{x, y, z}, 1.0f / norm(F3{x, y, z})); }
// Rec. 709 weights (renderer.cpp:166-168)
inline float relativeLuminance(F3 c) { return 0.2126f * c.x + 0.7152f * c.y + 0.0722f * c.z; }
// column-major 4x4 times (p, 1), sums left to right (renderer.cpp:145-151)
inline F3 toWorld(const Mat4& transform, const Vec3& p) {
    const auto& m = transform.values;
    return {m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12], m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
            m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14]};
}

// ---- tables derived from the scene (host arithmetic the device depends on) -----------------------

// renderer.cpp:108-143: the device's material record.
rt_gpu_material deviceMaterial(const Material& from) {
    rt_gpu_material to;
    std::memset(&to, 0, sizeof(to));
    to.baseColor = rt_float4{from.baseColor.x, from.baseColor.y, from.baseColor.z, from.baseColor.w};
    to.emissive = rt_float3{from.emissive.x, from.emissive.y, from.emissive.z};
    to.attenuationColor = rt_float3{from.attenuationColor.x, from.attenuationColor.y, from.attenuationColor.z};
    to.metallic = from.metallic;
    to.roughness = from.roughness;
    to.transmission = from.transmission;
    to.ior = from.ior;
    to.thickness = from.thickness;
    to.attenuationDistance = from.attenuationDistance;
    to.emissiveStrength = from.emissiveStrength;
    to.dispersion = from.dispersion;
    const std::pair<const TextureRef*, rt_gpu_texture_ref*> slots[] = {
        {&from.baseColorTexture, &to.baseColorTexture},       {&from.metallicRoughnessTexture, &to.metallicRoughnessTexture},
        {&from.normalTexture, &to.normalTexture},             {&from.emissiveTexture, &to.emissiveTexture},
        {&from.transmissionTexture, &to.transmissionTexture}, {&from.thicknessTexture, &to.thicknessTexture}};
    for (const auto& [source, target] : slots) {
        if (source->texture >= 0 && source->texCoord > 1)
            raise("Only glTF TEXCOORD_0 and TEXCOORD_1 are supported");  // renderer.cpp:109-110
        target->texture = source->texture;
        target->texCoord = source->texCoord;
        target->offset = rt_float2{source->offset.x, source->offset.y};
        target->scale = rt_float2{source->scale.x, source->scale.y};
        target->rotation = source->rotation;
        target->strength = source->strength;
    }
    return to;
}

// renderer.cpp:301-341: the latitude/longitude map as the device reads it, the running sum of
// luminance x sin(latitude) normalised to a distribution, and the map's power as a light.
struct EnvironmentTable {
    std::vector<rt_float4> texels;
    std::vector<float> distribution;
    float power = 0.0f;
};

EnvironmentTable environmentTable(const Environment& sky) {
    EnvironmentTable table;
    const std::size_t count = sky.pixels.size();
    if (count == 0)
        return table;
    table.texels.resize(count);
    table.distribution.resize(count);
    double sum = 0.0;
    std::size_t at = 0;
    for (std::uint32_t row = 0; row < sky.height; ++row) {
        const float sine = std::sin(3.14159265f * (static_cast<float>(row) + 0.5f) / sky.height);
        for (std::uint32_t column = 0; column < sky.width; ++column, ++at) {
            const auto& texel = sky.pixels[at];
            table.texels[at] = rt_float4{texel.x, texel.y, texel.z, texel.w};
            sum += relativeLuminance(F3{texel.x, texel.y, texel.z}) * sine;
            table.distribution[at] = static_cast<float>(sum);
        }
    }
    if (sum <= 0.0)
        return EnvironmentTable{};  // a black map is no light (renderer.cpp:326-327)
    for (float& value : table.distribution)
        value = static_cast<float>(value / sum);
    table.power = static_cast<float>(sum * 19.7392088 / static_cast<double>(sky.width * sky.height) * sky.strength);
    return table;
}

// renderer.cpp:436-507: emissive triangles in world space (weight = area x luminance), then the
// punctual lights (weight = luminance x solid angle), then the environment; `total` is the sum of
// the weights in list order.
struct LightTable {
    std::vector<rt_gpu_light> lights;
    float total = 0.0f;
};

LightTable lightTable(const Scene& scene, float environmentPower) {
    LightTable table;
    auto blank = [] {
        rt_gpu_light light;
        std::memset(&light, 0, sizeof(light));
        light.instance = light.primitive = 0xffffffffU;
        return light;
    };
    for (std::size_t i = 0; i < scene.instances.size(); ++i) {
        const Instance& instance = scene.instances[i];
        const Geometry& mesh = scene.geometries[instance.geometry];
        const Material& surface = scene.materials[mesh.material];
        const F3 emission = scaled(F3{surface.emissive.x, surface.emissive.y, surface.emissive.z}, surface.emissiveStrength);
        if (relativeLuminance(emission) <= 0.0f)
            continue;
        for (std::size_t t = 0; t + 2 < mesh.indices.size(); t += 3) {
            const F3 a = toWorld(instance.transform, mesh.vertices[mesh.indices[t]].position);
            const F3 b = toWorld(instance.transform, mesh.vertices[mesh.indices[t + 1]].position);
            const F3 c = toWorld(instance.transform, mesh.vertices[mesh.indices[t + 2]].position);
            const F3 areaVector = cross(b - a, c - a);
            const float twiceArea = norm(areaVector);
            if (twiceArea <= 1e-8f)
                continue;  // renderer.cpp:458-459
            rt_gpu_light light = blank();
            light.a = packed(a);
            light.b = packed(b);
            light.c = packed(c);
            light.normal = rt_float3{areaVector.x / twiceArea, areaVector.y / twiceArea, areaVector.z / twiceArea};
            light.emission = packed(emission);
            light.area = 0.5f * twiceArea;
            light.weight = light.area * relativeLuminance(emission);
            light.instance = static_cast<std::uint32_t>(i);
            light.primitive = static_cast<std::uint32_t>(t / 3);
            light.type = 3U;
            table.lights.push_back(light);
        }
    }
    for (const Light& source : scene.lights) {
        const auto& m = source.transform.values;
        const F3 emission = scaled(F3{source.color.x, source.color.y, source.color.z}, source.intensity);
        float solidAngle = 1.0f;  // directional
        if (source.type == 1U)
            solidAngle = 6.2831853f * (1.0f - std::cos(source.outerCone));
        else if (source.type == 2U)
            solidAngle = 12.566371f;
        rt_gpu_light light = blank();
        light.a = rt_float3{m[12], m[13], m[14]};
        light.b = packed(unit(-m[8], -m[9], -m[10]));  // the light looks down its -Z
        light.emission = packed(emission);
        light.weight = relativeLuminance(emission) * solidAngle;
        light.range = source.range;
        light.innerCone = source.innerCone;
        light.outerCone = source.outerCone;
        light.type = source.type;
        table.lights.push_back(light);
    }
    if (environmentPower > 0.0f) {
        rt_gpu_light light = blank();
        light.weight = environmentPower;
        light.type = 4U;
        table.lights.push_back(light);
    }
    for (const rt_gpu_light& light : table.lights)
        table.total += light.weight;
    return table;
}

// renderer.cpp:509-528: centre of the box around every transformed vertex and 0.525 of its diagonal.
struct SceneSphere {
    F3 centre{0.0f, 0.0f, 0.0f};
    float radius = 0.0f;
};

SceneSphere sceneSphere(const Scene& scene) {
    const float big = std::numeric_limits<float>::max();
    F3 lo{big, big, big}, hi{-big, -big, -big};
    for (const Instance& instance : scene.instances)
        for (const Vertex& vertex : scene.geometries[instance.geometry].vertices) {
            const F3 p = toWorld(instance.transform, vertex.position);
            lo = {std::min(lo.x, p.x), std::min(lo.y, p.y), std::min(lo.z, p.z)};
            hi = {std::max(hi.x, p.x), std::max(hi.y, p.y), std::max(hi.z, p.z)};
        }
    SceneSphere sphere;
    sphere.centre = {(lo.x + hi.x) * 0.5f, (lo.y + hi.y) * 0.5f, (lo.z + hi.z) * 0.5f};
    sphere.radius = norm(hi - lo) * 0.525f;
    return sphere;
}

// renderer.cpp:672-700: image-plane frame of a camera that looks down its -Z, and the photon gather
// radius: one pixel's footprint at the focus depth (clamped between the near plane and the scene's
// diameter), never below half a millimetre.
struct CameraFrame {
    F3 eye, u, v, w, lensU, lensV;
    float gatherRadius;
};

CameraFrame cameraFrame(const Camera& camera, std::uint32_t width, std::uint32_t height, float sceneRadius) {
    const auto& m = camera.transform.values;
    const F3 right = unit(m[0], m[1], m[2]);
    const F3 up = unit(m[4], m[5], m[6]);
    const F3 back = unit(m[8], m[9], m[10]);
    const float aspect = camera.aspectRatio > 0.0f ? camera.aspectRatio : static_cast<float>(width) / static_cast<float>(height);
    const float halfHeight = std::tan(camera.verticalFov * 0.5f);
    CameraFrame frame;
    frame.eye = {m[12], m[13], m[14]};
    frame.u = {right.x * halfHeight * aspect, right.y * halfHeight * aspect, right.z * halfHeight * aspect};
    frame.v = {up.x * halfHeight, up.y * halfHeight, up.z * halfHeight};
    frame.w = {-back.x, -back.y, -back.z};
    frame.lensU = right;
    frame.lensV = up;
    constexpr float radiusInPixels = 1.0f;  // renderer.cpp:173
    const float depth = std::clamp(camera.focusDistance, camera.nearPlane, sceneRadius * 2.0f);
    frame.gatherRadius = std::max(2.0f * radiusInPixels * depth * halfHeight / static_cast<float>(height), 0.0005f);
    return frame;
}

// Every index one table of the scene holds into another (instance -> geometry -> material ->
// texture -> image, triangle -> vertex) is checked before anything is uploaded: the device code
// follows them without looking, and a bad one is better an exception than a read outside a buffer.
// (The reference relies on its loader for this.)
void checkReferences(const Scene& scene) {
    auto broken = [](const char* what, const std::string& name) {
        raise(name.empty() ? std::string(what) : std::string(what) + " (" + name + ")");
    };
    for (const Instance& instance : scene.instances)
        if (instance.geometry >= scene.geometries.size())
            broken("An instance refers to a geometry the scene does not have", instance.name);
    for (const Geometry& mesh : scene.geometries) {
        if (mesh.material >= scene.materials.size())
            broken("A geometry refers to a material the scene does not have", mesh.name);
        if (mesh.indices.size() % 3 != 0)
            broken("The index count of a geometry is not a multiple of three", mesh.name);
        const std::size_t vertexCount = mesh.vertices.size();
        if (std::any_of(mesh.indices.begin(), mesh.indices.end(), [&](std::uint32_t index) { return index >= vertexCount; }))
            broken("A triangle refers to a vertex its geometry does not have", mesh.name);
    }
    for (const Material& material : scene.materials)
        for (const TextureRef* slot : {&material.baseColorTexture, &material.metallicRoughnessTexture, &material.normalTexture,
                                       &material.emissiveTexture, &material.transmissionTexture, &material.thicknessTexture})
            if (slot->texture >= 0 && static_cast<std::size_t>(slot->texture) >= scene.textures.size())
                broken("A material refers to a texture the scene does not have", material.name);
    for (const Texture& texture : scene.textures)
        if (texture.image < 0 || static_cast<std::size_t>(texture.image) >= scene.images.size())
            broken("A texture refers to an image the scene does not have", texture.name);
}

constexpr std::uint32_t kPhotonRebuildInterval = 4U;  // renderer.cpp:177: a new photon map every fourth sample
constexpr std::uint32_t kBandRows = 8U;               // rows per band of a frame (the engine's tile height)

}  // namespace

struct Renderer::Impl {
    // ---- scene data resident on the device --------------------------------------------------------
    struct Mesh {
        std::size_t vertices = 0, indices = 0;  // offsets in the scene pool
        std::uint32_t vertexCount = 0, triangleCount = 0, material = 0;
        rt_blas* blas = nullptr;
    };
    DevicePool scenePool;
    std::size_t lightIndexAt = 0;
    void* lightIndex = nullptr;  // in the scene pool: thresholds and triangle table of the light list
    std::vector<Mesh> meshes;
    rt_scene* tree = nullptr;

    // ---- frame buffers ------------------------------------------------------------------------------
    DevicePool framePool;
    std::array<std::size_t, 7> aovOffset{};  // accumulation, diffuse, reflection, refraction, caustics, albedo, normal
    std::size_t aovBytes = 0, outputOffset = 0, outputBytes = 0;

    // ---- photon map -----------------------------------------------------------------------------------
    // The reference keeps one map and rebuilds it before every fourth sample (renderer.cpp:177, 780-781).  Here the
    // maps of consecutive groups of four samples live side by side in a ring, so that a pass can hold the samples
    // of several groups (each gathering from its own group's map) and the next pass's maps can be traced while
    // this pass still renders.  A map is the sorted photon list, its bucket table and its bucket bits; the
    // unsorted list, the counter and the sort's scratch serve one photon pass at a time, on the photon stream.
    DevicePool photonPool;
    struct PhotonBlocks {
        std::vector<std::size_t> photons;  // one unsorted list per pass that is traced together with others
        std::size_t cellCounts = 0, counter = 0, scan = 0;  // counter: one word per list
        std::size_t scanBytes = 0;
    } photonAt;
    static constexpr std::uint32_t kNoEpoch = 0xffffffffU;
    struct PhotonMap {
        std::size_t sorted = 0, cellStart = 0, bucketBits = 0;  // offsets in the photon pool
        std::uint32_t epoch = kNoEpoch;                         // the group of four samples whose photons it holds
        hipEvent_t built = nullptr;                             // recorded on the photon stream behind its build
        std::vector<hipEvent_t> read;                           // per part: behind the part's last pass that gathers from it
        std::vector<char> readRecorded;
    };
    std::vector<PhotonMap> maps;
    // the maps of the passes that are being built step by step (manual passes), one per list in use
    std::vector<std::uint32_t> currentMaps{0};
    hipStream_t photonStream = nullptr;
    std::uint32_t* storedOnHost = nullptr;  // pinned words the stored counts of manual passes are copied to
    rt_wavefront* photonWavefront = nullptr;  // created by the first photon pass
    std::uint32_t photonWavefrontCapacity = 0;
    std::uint64_t photonWavefrontAsked = 0;  // paths the state was last sized for, before the memory budget cut them
    float gatherRadius = 0.0f;
    bool caustics = false;
    bool partsGather = false;  // the camera states of `parts` have the arrays a gather needs
    bool overflowReported = false;
    // The stored count of a pass stays on the device; a copy of it travels to pinned host memory behind the
    // pass and is looked at when it has arrived (reportOverflow), for the reference's warning.
    struct StoredCount {
        std::uint32_t* value = nullptr;  // pinned host memory
        hipEvent_t arrived = nullptr;
        bool pending = false;
    };
    std::vector<StoredCount> storedCounts;
    std::uint32_t* photonKeys = nullptr;  // inside the sort's scratch: the bucket keys the tracer writes
    bool manualPhotonPasses = false;

    // ---- launch state ---------------------------------------------------------------------------------
    rt_launch_params launch;
    std::uint32_t width = 0, height = 0, sample = 0;
    std::uint32_t sampleFirst = 0, sampleStride = 1;
    // what render() covers: every bandStride-th band from firstBand on (the whole frame by default), or
    // the pixel range [rangeBegin, rangeEnd) when rangeSet
    std::uint32_t firstBand = 0, bandStride = 1;
    bool rangeSet = false;
    std::uint32_t rangeBegin = 0, rangeEnd = 0;
    // A frame is cut into `partCount` regions that run side by side, and `depth` samples of a region
    // may be in flight at once: parts[(sample % depth) * partCount + region].  The accumulation steps
    // of one region are chained by events, because a pixel's running mean takes its samples in order.
    struct Part {
        rt_wavefront* wavefront = nullptr;
        hipStream_t stream = nullptr;
        hipEvent_t accumulated = nullptr;  // recorded behind the part's accumulation step (depth > 1)
        bool recorded = false;
    };
    std::vector<Part> parts;
    std::uint32_t partCount = 0, depth = 1;
    std::uint32_t depthWanted = 0;  // passes in flight as asked for; 0: two, or four with a photon map (see createParts)
    // render() calls are collected into passes of up to `batch` samples (setSampleBatch; 1: every call is a
    // pass of its own): `pending` calls wait for their pass, which goes out when it is full, before a photon
    // pass, and whenever the image, the counters or the GPU are asked for.
    std::uint32_t batch = 1, pending = 0, passes = 0;
    std::uint32_t batchWanted = 0;  // what setSampleBatch asked for; 0: as many as kStateBudget allows
    void* pendingTarget = nullptr;
    // The parts are created when the first pass goes out, for the pixels render() covers by then, and again
    // when that grows or shrinks to less than half: a GPU that takes one share of many has state for its share.
    std::uint64_t partCapacity = 0;                     // pixels one part has room for
    std::uint32_t shareNumerator = 0, shareDenominator = 0;  // setRayShare; 0: the policy of createParts
    // wavefront state a renderer takes by itself: pixels x samples per pass x passes in flight x the bytes the
    // engine states per path, at most this much and at most half of the memory that is free when the state is made
    static constexpr std::uint64_t kStateBudget = 24ULL << 30;
    // a pass with a photon map holds at most this many groups of four samples
    static constexpr std::uint32_t kMapsPerPass = 16;  // RT_MAX_PHOTON_MAPS: what the gather kernels take a table of
    std::uint32_t measurementFlags = 0;
    // one region, four samples of it in flight: measured best on every workload (a 1080p sample's
    // launches leave the GPU half idle in their tails; four of them side by side fill it)
    static constexpr std::uint32_t kDefaultParts = 1;

    ~Impl() {
        (void)hipDeviceSynchronize();
        for (StoredCount& count : storedCounts)
            if (count.pending)
                warnIfOverfull(*count.value);  // everything has arrived behind the synchronisation
        destroyParts();
        destroyMaps();
        if (photonStream)
            (void)hipStreamDestroy(photonStream);
        if (storedOnHost)
            (void)hipHostFree(storedOnHost);
        if (photonWavefront)
            rt_wavefront_destroy(photonWavefront);
        for (StoredCount& count : storedCounts) {
            if (count.value)
                (void)hipHostFree(count.value);
            if (count.arrived)
                (void)hipEventDestroy(count.arrived);
        }
        if (tree)
            rt_scene_destroy(tree);
        for (Mesh& mesh : meshes)
            if (mesh.blas)
                rt_blas_destroy(mesh.blas);
    }

    // renderer.cpp:215-254
    void initialize(const Scene& scene, std::uint32_t renderWidth, std::uint32_t renderHeight, Profile profile) {
        if (scene.instances.empty())
            raise("The scene has no geometry");  // renderer.cpp:217-218
        if (scene.cameras.empty())
            raise("The scene has no perspective camera. Blender's File > Export > glTF leaves Cameras unchecked by "
                  "default; use the Raytracer panel's Export GLB instead");  // renderer.cpp:219-223
        if (renderWidth == 0 || renderHeight == 0)
            raise("The image has no pixels");
        if (scene.materials.empty())
            raise("The scene has no materials");
        checkReferences(scene);
        width = renderWidth;
        height = renderHeight;
        std::memset(&launch, 0, sizeof(launch));
        launch.width = width;
        launch.height = height;
        hipOk(hipFree(nullptr));  // brings the runtime up (renderer.cpp:228)

        uploadScene(scene);
        buildTrees(scene);
        const SceneSphere sphere = sceneSphere(scene);
        launch.sceneCenter = packed(sphere.centre);
        launch.sceneRadius = sphere.radius;
        setPhotonSizes(1U << 24U, 1U << 23U, 1U << 22U);  // renderer.cpp:533-535
        allocateFrame();
        partCount = kDefaultParts;
        depthWanted = 0;
        applyProfile(profile);
        applyCamera(scene.cameras.front());
        restart();
    }

    // Materials, textures and their pixels, the environment with its distribution, the light list and
    // every mesh's vertices and indices: staged on the host at their offsets in the scene pool and
    // uploaded with one copy (the reference: renderer.cpp:256-299, 301-341, 349-361, 436-507, one
    // allocation and one copy per table).
    void uploadScene(const Scene& scene) {
        Staging staging;

        std::vector<rt_gpu_material> materials;
        materials.reserve(scene.materials.size());
        for (const Material& material : scene.materials)
            materials.push_back(deviceMaterial(material));
        const std::size_t materialsAt = staging.append(materials.data(), materials.size() * sizeof(rt_gpu_material));

        std::vector<std::size_t> imageAt;
        for (const Image& image : scene.images) {
            if (image.rgba.empty() || image.rgba.size() != static_cast<std::size_t>(image.width) * image.height * 4)
                raise("Unable to decode texture " + image.name + ": this build takes decoded RGBA8 pixels only");  // :275-277
            imageAt.push_back(staging.append(image.rgba.data(), image.rgba.size()));
        }
        const std::size_t texturesAt = staging.reserve<rt_gpu_texture>(scene.textures.size());

        const EnvironmentTable sky = environmentTable(scene.environment);
        const std::size_t skyAt = staging.append(sky.texels.data(), sky.texels.size() * sizeof(rt_float4));
        const std::size_t skyCdfAt = staging.append(sky.distribution.data(), sky.distribution.size() * sizeof(float));

        const LightTable lights = lightTable(scene, sky.power);
        const std::size_t lightsAt = staging.append(lights.lights.data(), lights.lights.size() * sizeof(rt_gpu_light));
        // room for the index that replaces the walks over the light list, built on the device below
        std::size_t lightIndexBytes = 0;
        engineOk(rt_light_index_bytes(static_cast<std::uint32_t>(lights.lights.size()), &lightIndexBytes), "rt_light_index_bytes");
        lightIndexAt = lights.lights.empty() ? 0 : staging.reserve<unsigned char>(lightIndexBytes);

        meshes.resize(scene.geometries.size());
        for (std::size_t g = 0; g < scene.geometries.size(); ++g) {
            const Geometry& source = scene.geometries[g];
            Mesh& mesh = meshes[g];
            mesh.vertexCount = static_cast<std::uint32_t>(source.vertices.size());
            mesh.triangleCount = static_cast<std::uint32_t>(source.indices.size() / 3);
            mesh.material = source.material;
            // the 56-byte host vertex widened to the 64-byte device vertex in place (renderer.cpp:349-358)
            mesh.vertices = staging.reserve<rt_gpu_vertex>(source.vertices.size());
            rt_gpu_vertex* out = staging.at<rt_gpu_vertex>(mesh.vertices);
            for (const Vertex& v : source.vertices) {
                std::memset(out, 0, sizeof(*out));
                out->position = rt_float3{v.position.x, v.position.y, v.position.z};
                out->normal = rt_float3{v.normal.x, v.normal.y, v.normal.z};
                out->tangent = rt_float4{v.tangent.x, v.tangent.y, v.tangent.z, v.tangent.w};
                out->uv = rt_float2{v.uv.x, v.uv.y};
                out->uv1 = rt_float2{v.uv1.x, v.uv1.y};
                ++out;
            }
            mesh.indices = staging.append(source.indices.data(), source.indices.size() * sizeof(std::uint32_t));
        }

        // the pool's address is needed for the pixel pointers of the texture records
        scenePool.allocate(staging.bytes());
        rt_gpu_texture* records = staging.at<rt_gpu_texture>(texturesAt);
        for (std::size_t t = 0; t < scene.textures.size(); ++t) {
            const Texture& texture = scene.textures[t];
            const Image& image = scene.images[static_cast<std::size_t>(texture.image)];
            records[t].pixels = scenePool.at<const rt_uchar4>(imageAt[static_cast<std::size_t>(texture.image)]);
            records[t].width = image.width;
            records[t].height = image.height;
            records[t].wrapU = texture.wrapU;
            records[t].wrapV = texture.wrapV;
        }
        if (staging.bytes())
            hipOk(hipMemcpy(scenePool.at<void>(0), staging.at<void>(0), staging.bytes(), hipMemcpyHostToDevice));

        launch.materials = scenePool.at<const rt_gpu_material>(materialsAt);
        launch.textures = scene.textures.empty() ? nullptr : scenePool.at<const rt_gpu_texture>(texturesAt);
        launch.environmentRotation = scene.environment.rotation;
        launch.environmentStrength = scene.environment.strength;
        launch.exposure = scene.environment.exposure;
        if (sky.power > 0.0f || !sky.texels.empty()) {
            launch.environment = scenePool.at<const rt_float4>(skyAt);
            launch.environmentCdf = scenePool.at<const float>(skyCdfAt);
            launch.environmentWidth = scene.environment.width;
            launch.environmentHeight = scene.environment.height;
            launch.environmentWeight = sky.power;
        }
        launch.lightCount = static_cast<std::uint32_t>(lights.lights.size());
        launch.lightWeight = lights.total;
        launch.lights = lights.lights.empty() ? nullptr : scenePool.at<const rt_gpu_light>(lightsAt);
        lightIndex = nullptr;
        // RT_LIGHT_INDEX=0 keeps the walks over the light list (for comparison; the image is the same)
        const char* indexChoice = std::getenv("RT_LIGHT_INDEX");
        if (!lights.lights.empty() && !(indexChoice && indexChoice[0] == '0')) {
            lightIndex = scenePool.at<void>(lightIndexAt);
            engineOk(rt_light_index_build(launch.lights, launch.lightCount, scenePool.at<void>(lightIndexAt), lightIndexBytes,
                                          nullptr),
                     "rt_light_index_build");
        }
    }

    // One bottom-level tree per geometry, the tree over the instances, and the per-geometry records
    // the shading kernels find vertices, indices and material through (renderer.cpp:343-434, 633-670).
    void buildTrees(const Scene& scene) {
        for (Mesh& mesh : meshes)
            engineOk(rt_blas_build(scenePool.at<void>(mesh.vertices), mesh.vertexCount, sizeof(rt_gpu_vertex),
                                   scenePool.at<const std::uint32_t>(mesh.indices), mesh.triangleCount, nullptr, &mesh.blas),
                     "rt_blas_build");
        std::vector<rt_instance_desc> placements(scene.instances.size());
        for (std::size_t i = 0; i < placements.size(); ++i) {
            const Instance& instance = scene.instances[i];
            // rows of the column-major 4x4 (renderer.cpp:403-406)
            for (int row = 0; row < 3; ++row)
                for (int column = 0; column < 4; ++column)
                    placements[i].transform[row * 4 + column] = instance.transform.values[column * 4 + row];
            placements[i].instanceId = static_cast<std::uint32_t>(i);
            placements[i].geometry = instance.geometry;
            placements[i].blas = meshes[instance.geometry].blas;
        }
        engineOk(rt_scene_build(placements.data(), static_cast<std::uint32_t>(placements.size()), nullptr, &tree),
                 "rt_scene_build");
        std::vector<rt_hit_group_data> records(meshes.size());
        for (std::size_t g = 0; g < meshes.size(); ++g) {
            records[g].vertices = scenePool.at<const rt_gpu_vertex>(meshes[g].vertices);
            records[g].indices = scenePool.at<const rt_uint3>(meshes[g].indices);
            records[g].material = meshes[g].material;
        }
        engineOk(rt_scene_set_geometry(tree, records.data(), static_cast<std::uint32_t>(records.size())),
                 "rt_scene_set_geometry");
        launch.scene = rt_scene_handle(tree);
    }

    // Seven float4 buffers and the byte image (renderer.cpp:243-250) in one allocation.
    void allocateFrame() {
        const std::size_t pixels = static_cast<std::size_t>(width) * height;
        Layout layout;
        aovBytes = pixels * sizeof(rt_float4);
        for (std::size_t& offset : aovOffset)
            offset = layout.take(aovBytes);
        outputBytes = pixels * sizeof(rt_uchar4);
        outputOffset = layout.take(outputBytes);
        framePool.allocate(layout.bytes());
        rt_float4** targets[7] = {&launch.accumulation, &launch.diffuse,     &launch.reflection, &launch.refraction,
                                  &launch.caustics,     &launch.albedoGuide, &launch.normalGuide};
        for (int k = 0; k < 7; ++k)
            *targets[k] = framePool.at<rt_float4>(aovOffset[static_cast<std::size_t>(k)]);
    }

    // Photon list, the counter, the sort's scratch and, per map of the ring, the sorted copy, the bucket table
    // and the bucket bits (renderer.cpp:537-550) in one allocation.
    void setPhotonSizes(std::uint32_t emissions, std::uint32_t capacity, std::uint32_t buckets) {
        launch.photonEmissions = emissions;
        launch.photonCapacity = capacity;
        launch.photonBucketCount = buckets;
        allocateMaps(std::max<std::size_t>(1, maps.size()));
    }

    void destroyMaps() {
        for (PhotonMap& map : maps) {
            if (map.built)
                (void)hipEventDestroy(map.built);
            for (hipEvent_t event : map.read)
                if (event)
                    (void)hipEventDestroy(event);
        }
        maps.clear();
    }

    void allocateMaps(std::size_t count) {
        hipOk(hipDeviceSynchronize());
        destroyMaps();
        if (!photonStream)
            hipOk(hipStreamCreate(&photonStream));
        if (!storedOnHost)
            hipOk(hipHostMalloc(reinterpret_cast<void**>(&storedOnHost), sizeof(std::uint32_t) * RT_MAX_PHOTON_PASSES,
                                hipHostMallocDefault));
        const std::uint32_t capacity = launch.photonCapacity, buckets = launch.photonBucketCount;
        Layout layout;
        const std::size_t listBytes = static_cast<std::size_t>(capacity) * sizeof(rt_photon);
        const std::size_t tableBytes = (static_cast<std::size_t>(buckets) + 1U) * sizeof(std::uint32_t);
        const std::size_t bitBytes = std::max<std::size_t>(1, buckets / 32U) * sizeof(std::uint32_t);
        photonAt.photons.assign(listsWanted(), 0);
        for (std::size_t& list : photonAt.photons)
            list = layout.take(listBytes);
        photonAt.cellCounts = layout.take(tableBytes);
        photonAt.counter = layout.take(sizeof(std::uint32_t) * RT_MAX_PHOTON_PASSES);
        // room for the sort of a full photon list (the reference sizes its scan for the buckets alone)
        photonAt.scanBytes = std::max(photonScanStorage(buckets), rt_photon_grid_scratch_bytes(capacity, buckets));
        photonAt.scan = layout.take(photonAt.scanBytes);
        maps.resize(count);
        for (PhotonMap& map : maps) {
            map.sorted = layout.take(listBytes);
            map.cellStart = layout.take(tableBytes);
            map.bucketBits = layout.take(bitBytes);
        }
        photonPool.allocate(layout.bytes());
        for (PhotonMap& map : maps) {
            // until its first pass every bucket of a map is empty
            hipOk(hipMemset(photonPool.at<void>(map.cellStart), 0, tableBytes));
            hipOk(hipMemset(photonPool.at<void>(map.bucketBits), 0, bitBytes));
            hipOk(hipEventCreateWithFlags(&map.built, hipEventDisableTiming));
            map.read.assign(parts.size(), nullptr);
            map.readRecorded.assign(parts.size(), 0);
            for (hipEvent_t& event : map.read)
                hipOk(hipEventCreateWithFlags(&event, hipEventDisableTiming));
        }
        currentMaps.assign(1, 0U);
        launch.photons = photonPool.at<rt_photon>(photonAt.photons[0]);
        launch.photonCount = photonPool.at<std::uint32_t>(photonAt.counter);
        launch.sortedPhotons = photonPool.at<const rt_photon>(maps[0].sorted);
        launch.photonCellStart = photonPool.at<const std::uint32_t>(maps[0].cellStart);
        photonKeys = rt_photon_grid_keys(photonPool.at<void>(photonAt.scan), photonAt.scanBytes, capacity, buckets);
    }

    // Maps the ring must hold: those of the passes in flight.  A pass of n samples that starts anywhere in a
    // group of four touches (n + 6) / 4 groups at most, n / 4 when passes start on a group's first sample.
    std::size_t mapsWanted() const {
        if (!caustics)
            return 1;
        const std::uint32_t perPass = batch % kPhotonRebuildInterval == 0U ? batch / kPhotonRebuildInterval
                                                                          : (batch + 6U) / kPhotonRebuildInterval;
        return static_cast<std::size_t>(std::max(perPass, 1U)) * depth;
    }

    // Bytes of the photon pool for passes of `samples` samples (whole groups of four): see allocateMaps.
    std::uint64_t photonPoolBytes(std::uint32_t samples) const {
        const std::uint64_t listBytes = static_cast<std::uint64_t>(launch.photonCapacity) * sizeof(rt_photon);
        const std::uint64_t tableBytes = (static_cast<std::uint64_t>(launch.photonBucketCount) + 1U) * sizeof(std::uint32_t);
        const std::uint64_t perPass = std::max<std::uint32_t>(samples / kPhotonRebuildInterval, 1U);
        const std::uint64_t lists = std::min<std::uint64_t>(perPass, RT_MAX_PHOTON_PASSES);
        return perPass * depth * (listBytes + tableBytes + launch.photonBucketCount / 8U) + lists * listBytes;
    }

    // Photon passes traced together: those of the maps one camera pass gathers from (when its samples are whole
    // groups of four), at most RT_MAX_PHOTON_PASSES.  Each has an unsorted list of its own.
    std::size_t listsWanted() const {
        if (!caustics || batch % kPhotonRebuildInterval != 0U)
            return 1;
        return std::min<std::size_t>(std::max<std::uint32_t>(batch / kPhotonRebuildInterval, 1U), RT_MAX_PHOTON_PASSES);
    }

    void shareLightIndex() {
        for (Part& part : parts)
            if (part.wavefront)
                engineOk(rt_wavefront_set_light_index(part.wavefront, lightIndex, launch.lightCount),
                         "rt_wavefront_set_light_index");
    }

    void applyProfile(Profile profile) {  // renderer.cpp:226, 251, 735-737
        launch.maxDepth = profile == Profile::Final ? 32U : 5U;
        caustics = profile == Profile::Final;
    }

    void applyCamera(const Camera& camera) {  // renderer.cpp:672-714
        const CameraFrame frame = cameraFrame(camera, width, height, launch.sceneRadius);
        launch.eye = packed(frame.eye);
        launch.cameraU = packed(frame.u);
        launch.cameraV = packed(frame.v);
        launch.cameraW = packed(frame.w);
        launch.lensU = packed(frame.lensU);
        launch.lensV = packed(frame.lensV);
        launch.aperture = camera.aperture;
        launch.focusDistance = camera.focusDistance;
        gatherRadius = frame.gatherRadius;
    }

    void restart() {  // renderer.cpp:716-728
        hipOk(hipDeviceSynchronize());  // samples may still be in flight on the parts' streams
        reportOverflow(true);
        for (Part& part : parts)
            part.recorded = false;
        for (PhotonMap& map : maps) {
            map.epoch = kNoEpoch;
            std::fill(map.readRecorded.begin(), map.readRecorded.end(), 0);
        }
        for (std::size_t offset : aovOffset)
            hipOk(hipMemset(framePool.at<void>(offset), 0, aovBytes));
        launch.photonPass = 0;
        launch.photonRadius = caustics ? gatherRadius : 0.0f;
        sample = 0;
        pending = 0;  // calls that were still waiting belonged to the sequence that ends here
    }

    // ---- photon pass (renderer.cpp:741-777), in the three steps a multi-GPU driver interleaves with
    // its exchange.  Everything of a pass runs on the photon stream.

    // The group of four samples the next render() call belongs to, and whether its map is missing.
    std::uint32_t nextEpoch() const { return (sample + pending) / kPhotonRebuildInterval; }
    bool photonPassDue() const {  // renderer.cpp:780
        if (!(launch.photonRadius > 0.0f) || (sample + pending) % kPhotonRebuildInterval != 0U)
            return false;
        return maps.empty() || parts.empty() || maps[nextEpoch() % maps.size()].epoch != nextEpoch();
    }

    // The ring slot for the map of `epoch`, free to be overwritten: calls that wait for their pass and still read
    // the map the slot holds go out first, and the photon stream waits for the passes that gather from it.
    std::uint32_t claimMap(std::uint32_t epoch) {
        ensureParts();
        PhotonMap* map = &maps[epoch % maps.size()];
        if (pending > 0 && map->epoch != kNoEpoch && map->epoch >= sample / kPhotonRebuildInterval) {
            flush();
            map = &maps[epoch % maps.size()];
        }
        for (std::size_t k = 0; k < map->read.size(); ++k)
            if (map->readRecorded[k]) {
                hipOk(hipStreamWaitEvent(photonStream, map->read[k], 0));
                map->readRecorded[k] = 0;
            }
        map->epoch = kNoEpoch;
        return static_cast<std::uint32_t>(map - maps.data());
    }

    std::uint32_t* counterOf(std::size_t list) const { return photonPool.at<std::uint32_t>(photonAt.counter) + list; }
    rt_photon* listOf(std::size_t list) const { return photonPool.at<rt_photon>(photonAt.photons[list]); }

    // Starts the passes of the next `wanted` photon maps (those of the groups of four samples from the next
    // render() call on); fewer when the lists, the ring or the group's alignment do not allow more, at least one.
    std::uint32_t beginPhotonPasses(std::uint32_t wanted) {  // :742-745
        reportOverflow(false);
        launch.photonRadius = gatherRadius;
        ensureParts();
        const std::uint32_t epoch = nextEpoch();
        std::uint32_t count = std::min<std::uint32_t>(std::max(wanted, 1U), static_cast<std::uint32_t>(photonAt.photons.size()));
        if (epoch % count != 0U || count > maps.size())
            count = 1;  // groups start on a multiple of their size, so that the passes after them do as well
        currentMaps.resize(count);
        for (std::uint32_t k = 0; k < count; ++k)
            currentMaps[k] = claimMap(epoch + k);
        launch.photonPass = epoch;
        hipOk(hipMemsetAsync(counterOf(0), 0, sizeof(std::uint32_t) * RT_MAX_PHOTON_PASSES, photonStream));
        return count;
    }
    void beginPhotonPass() { beginPhotonPasses(1); }

    // The photon passes have wavefront state of their own, sized for the emissions of the passes that run
    // together (at most what a quarter of the free memory and 8 GiB hold, at least one pass of up to 2^24
    // emissions, which goes in pieces when it is larger): every batch ends with the drain of its ray launches,
    // so few large batches beat many small ones.
    void ensurePhotonWavefront(std::uint32_t emissions, std::uint32_t passes) {
        const std::uint64_t floorPaths = std::min<std::uint32_t>(std::max<std::uint32_t>(emissions, 1U), 1U << 24);
        std::uint64_t wanted = std::max<std::uint64_t>(static_cast<std::uint64_t>(std::max(emissions, 1U)) * passes, floorPaths);
        // (a state that was cut to the memory budget is not made again for the request it was cut from)
        if (photonWavefront && (photonWavefrontCapacity >= wanted || photonWavefrontAsked >= wanted))
            return;
        photonWavefrontAsked = wanted;
        hipOk(hipDeviceSynchronize());
        if (photonWavefront)
            rt_wavefront_destroy(photonWavefront);
        photonWavefront = nullptr;
        photonWavefrontCapacity = 0;
        std::size_t freeBytes = 0, totalBytes = 0;
        hipOk(hipMemGetInfo(&freeBytes, &totalBytes));
        const std::uint64_t budget = std::min<std::uint64_t>(8ULL << 30, freeBytes / 4U);
        while (wanted > floorPaths) {  // whole passes
            std::size_t bytes = 0;
            engineOk(rt_wavefront_state_bytes(static_cast<std::uint32_t>(std::min<std::uint64_t>(wanted, 0xffffffffULL)),
                                              RT_WAVEFRONT_FOR_PHOTONS, &bytes),
                     "rt_wavefront_state_bytes");
            if (wanted <= 0xffffffffULL && bytes <= budget)
                break;
            wanted -= std::max(emissions, 1U);
        }
        wanted = std::max(wanted, floorPaths);
        engineOk(rt_wavefront_create_for(static_cast<std::uint32_t>(wanted), RT_WAVEFRONT_FOR_PHOTONS, &photonWavefront),
                 "rt_wavefront_create_for");
        engineOk(rt_wavefront_set_light_index(photonWavefront, lightIndex, launch.lightCount), "rt_wavefront_set_light_index");
        engineOk(rt_wavefront_prepare_stream(photonWavefront, photonStream), "rt_wavefront_prepare_stream");
        photonWavefrontCapacity = static_cast<std::uint32_t>(wanted);
    }

    // The emissions [first, first + count) of every pass that has been begun, on the photon stream.
    void tracePasses(std::uint32_t first, std::uint32_t count, bool intoStaging, bool withKeys) {
        const std::uint32_t passes = static_cast<std::uint32_t>(currentMaps.size());
        ensurePhotonWavefront(count, passes);
        rt_photon* lists[RT_MAX_PHOTON_PASSES] = {};
        std::uint32_t* counters[RT_MAX_PHOTON_PASSES] = {};
        std::uint32_t* keys[RT_MAX_PHOTON_PASSES] = {};
        for (std::uint32_t k = 0; k < passes; ++k) {
            lists[k] = intoStaging ? photonPool.at<rt_photon>(maps[currentMaps[k]].sorted) : listOf(k);
            counters[k] = counterOf(k);
            keys[k] = withKeys && passes == 1U ? photonKeys : nullptr;  // the sort's scratch holds the keys of one list
        }
        engineOk(rt_trace_photon_passes(photonWavefront, &launch, first, count, passes, lists, counters, keys, photonStream),
                 "rt_trace_photon_passes");
    }

    // One range of the begun passes for a driver that merges the lists of several GPUs: waits for the tracer (not
    // for the samples that are still rendering on other streams) and reports what this GPU stored per pass (:746-769).
    void tracePhotonRangePasses(std::uint32_t first, std::uint32_t count, bool intoStaging, std::uint32_t* stored) {
        tracePasses(first, count, intoStaging, false);
        const std::size_t passes = currentMaps.size();
        hipOk(hipMemcpyAsync(storedOnHost, counterOf(0), sizeof(std::uint32_t) * passes, hipMemcpyDeviceToHost, photonStream));
        hipOk(hipStreamSynchronize(photonStream));
        for (std::size_t k = 0; k < passes; ++k) {
            warnIfOverfull(storedOnHost[k]);
            stored[k] = std::min(storedOnHost[k], launch.photonCapacity);
        }
    }
    std::uint32_t tracePhotonRange(std::uint32_t first, std::uint32_t count, bool intoStaging) {
        std::uint32_t stored[RT_MAX_PHOTON_PASSES] = {};
        tracePhotonRangePasses(first, count, intoStaging, stored);
        return stored[0];
    }

    void warnIfOverfull(std::uint32_t stored) {
        if (stored > launch.photonCapacity && !overflowReported) {
            overflowReported = true;  // once, like renderer.cpp:762-768
            std::fprintf(stderr, "warning: photon buffer full (%u of %u stored); the caustic will be darker than it should be\n",
                         stored, launch.photonCapacity);
        }
    }

    void finishPhotonPasses(const std::uint32_t* totals) {  // :770-776
        for (std::size_t k = 0; k < currentMaps.size(); ++k) {
            const std::uint32_t stored = std::min(totals[k], launch.photonCapacity);
            PhotonMap& map = maps[currentMaps[k]];
            engineOk(rt_build_photon_grid_async(listOf(k), stored, launch.photonRadius, launch.photonBucketCount,
                                                photonPool.at<std::uint32_t>(photonAt.cellCounts),
                                                photonPool.at<std::uint32_t>(map.cellStart), photonPool.at<void>(photonAt.scan),
                                                photonAt.scanBytes, photonPool.at<rt_photon>(map.sorted), photonStream),
                     "buildPhotonGrid");
            sealMap(map, launch.photonPass + static_cast<std::uint32_t>(k));
        }
    }
    void finishPhotonPass(std::uint32_t stored) { finishPhotonPasses(&stored); }

    // Bucket bits of a freshly built map, the event its readers wait for, and the group it now serves.
    void sealMap(PhotonMap& map, std::uint32_t epoch) {
        engineOk(rt_photon_bucket_bitmap(photonPool.at<const std::uint32_t>(map.cellStart), launch.photonBucketCount,
                                         photonPool.at<std::uint32_t>(map.bucketBits), photonStream),
                 "rt_photon_bucket_bitmap");
        hipOk(hipEventRecord(map.built, photonStream));
        map.epoch = epoch;
    }

    // The passes of renderer.cpp:741-777 without a wait: the reference synchronises behind the tracer and reads the
    // stored count back (:753-758) to clamp it (:769) and hand it to the grid build (:770-775); here the count stays
    // in device memory, the build reads and clamps it there, the tracer writes every photon's bucket key beside it
    // (a single pass) and the warning of :762-768 is printed when a copy of the count has reached the host by
    // itself.  The passes of all the maps the next camera pass gathers from are traced together.  The host runs on
    // to the samples that read the new maps.
    void photonPassInFlight() {
        const std::uint32_t passes = beginPhotonPasses(static_cast<std::uint32_t>(photonAt.photons.size()));
        tracePasses(0, launch.photonEmissions, false, true);
        const std::uint32_t most = std::min(launch.photonCapacity, launch.photonEmissions);
        for (std::uint32_t k = 0; k < passes; ++k) {
            PhotonMap& map = maps[currentMaps[k]];
            watchStoredCount(counterOf(k));
            engineOk(rt_build_photon_grid_counted(listOf(k), counterOf(k), most, launch.photonRadius, launch.photonBucketCount,
                                                  photonPool.at<std::uint32_t>(map.cellStart), photonPool.at<void>(photonAt.scan),
                                                  photonAt.scanBytes, photonPool.at<rt_photon>(map.sorted),
                                                  passes == 1U && photonKeys ? 1 : 0, photonStream),
                     "rt_build_photon_grid_counted");
            sealMap(map, launch.photonPass + k);
        }
    }

    void watchStoredCount(const std::uint32_t* counter) {
        StoredCount* slot = nullptr;
        for (StoredCount& candidate : storedCounts)
            if (!candidate.pending)
                slot = &candidate;
        if (!slot) {
            if (storedCounts.size() >= 16) {  // the host is many passes ahead: look at the oldest first
                reportOverflow(true);
                return watchStoredCount(counter);
            }
            storedCounts.emplace_back();
            slot = &storedCounts.back();
            hipOk(hipHostMalloc(reinterpret_cast<void**>(&slot->value), sizeof(std::uint32_t), hipHostMallocDefault));
            hipOk(hipEventCreateWithFlags(&slot->arrived, hipEventDisableTiming));
        }
        hipOk(hipMemcpyAsync(slot->value, counter, sizeof(std::uint32_t), hipMemcpyDeviceToHost, photonStream));
        hipOk(hipEventRecord(slot->arrived, photonStream));
        slot->pending = true;
    }

    // Looks at the stored counts that have reached the host (all of them after `wait`).
    void reportOverflow(bool wait) {
        for (StoredCount& count : storedCounts) {
            if (!count.pending)
                continue;
            if (wait)
                hipOk(hipEventSynchronize(count.arrived));
            else if (hipEventQuery(count.arrived) != hipSuccess)
                continue;
            count.pending = false;
            warnIfOverfull(*count.value);
        }
    }

    // ---- one sample (renderer.cpp:779-796) --------------------------------------------------------------
    void render(void* target) {
        if (photonPassDue()) {
            if (manualPhotonPasses)
                raise("a photon pass is due: run beginPhotonPass/tracePhotonRange/finishPhotonPass");
            photonPassInFlight();
        }
        if (parts.empty())
            ensureParts();  // sets the number of samples a pass takes
        pendingTarget = target;
        ++pending;
        // a caller's own target is written by the pass this call belongs to, so that pass goes out now
        if (pending >= batch || target)
            flush();
    }

    // Sends the waiting render() calls off as one pass over their samples.
    void flush() {
        if (pending == 0)
            return;
        ensureParts();
        void* const target = pendingTarget;
        launch.display = nullptr;
        launch.display2 = nullptr;
        launch.display3 = nullptr;
        launch.composite = nullptr;
        launch.output = target ? static_cast<rt_uchar4*>(target) : framePool.at<rt_uchar4>(outputOffset);
        launch.sample = sample;
        const std::uint32_t seed = sampleFirst + sample * sampleStride;
        // The parts run on their own streams, so the slow rays that end one part's ray launch overlap
        // with the other parts' work instead of idling the GPU.  The streams are ordinary (blocking)
        // ones: work on the default stream before and after still orders against them.
        const std::uint32_t count = partCount;
        const std::uint32_t set = passes % depth, before = (passes + depth - 1U) % depth;
        // the maps of the groups of four samples this pass covers, in order
        std::vector<rt_photon_map> gathered;
        std::vector<PhotonMap*> gatheredFrom;
        if (launch.photonRadius > 0.0f) {
            const std::uint32_t firstEpoch = sample / kPhotonRebuildInterval, lastEpoch = (sample + pending - 1U) / kPhotonRebuildInterval;
            for (std::uint32_t epoch = firstEpoch; epoch <= lastEpoch; ++epoch) {
                PhotonMap& map = maps[epoch % maps.size()];
                if (map.epoch != epoch)
                    raise("no photon map for the samples of this pass");
                gathered.push_back(rt_photon_map{photonPool.at<const rt_photon>(map.sorted),
                                                 photonPool.at<const std::uint32_t>(map.cellStart),
                                                 photonPool.at<const std::uint32_t>(map.bucketBits)});
                gatheredFrom.push_back(&map);
            }
            launch.sortedPhotons = gathered.front().sortedPhotons;
            launch.photonCellStart = gathered.front().cellStart;
        }
        auto partOf = [&](std::uint32_t k) -> Part& {
            Part& part = parts[set * count + k];
            engineOk(rt_wavefront_set_sample_batch(part.wavefront, pending, sampleStride), "rt_wavefront_set_sample_batch");
            engineOk(rt_wavefront_set_photon_maps(part.wavefront, gathered.data(), static_cast<std::uint32_t>(gathered.size()),
                                                  kPhotonRebuildInterval, sample % kPhotonRebuildInterval),
                     "rt_wavefront_set_photon_maps");
            for (PhotonMap* map : gatheredFrom)
                hipOk(hipStreamWaitEvent(part.stream, map->built, 0));
            if (depth > 1U) {
                // this region's previous sample was accumulated by the part of the set before this one
                Part& previous = parts[before * count + k];
                engineOk(rt_wavefront_set_accumulate_events(part.wavefront, previous.recorded ? previous.accumulated : nullptr,
                                                            part.accumulated),
                         "rt_wavefront_set_accumulate_events");
                part.recorded = true;
            }
            return part;
        };
        if (rangeSet) {
            const std::uint64_t pixels = rangeEnd - rangeBegin;
            for (std::uint32_t k = 0; k < count; ++k) {
                const auto begin = rangeBegin + static_cast<std::uint32_t>(pixels * k / count);
                const auto end = rangeBegin + static_cast<std::uint32_t>(pixels * (k + 1) / count);
                Part& part = partOf(k);
                engineOk(rt_render_sample_seeded(part.wavefront, &launch, begin, end, seed, part.stream), "rt_render_sample");
            }
        } else {
            const std::uint32_t bands = (height + kBandRows - 1U) / kBandRows;
            const std::uint64_t mine = firstBand < bands ? (bands - 1U - firstBand) / bandStride + 1U : 0U;
            for (std::uint32_t k = 0; k < count; ++k) {
                const auto from = static_cast<std::uint32_t>(mine * k / count), to = static_cast<std::uint32_t>(mine * (k + 1) / count);
                Part& part = partOf(k);
                if (to > from)
                    engineOk(rt_render_sample_bands(part.wavefront, &launch, firstBand + from * bandStride, bandStride,
                                                    to - from, seed, part.stream),
                             "rt_render_sample_bands");
                else
                    part.recorded = false;  // nothing was enqueued, so nothing records the event
            }
        }
        // the photon pass that overwrites one of these maps waits for the parts that gather from it
        for (PhotonMap* map : gatheredFrom)
            for (std::uint32_t k = 0; k < count; ++k) {
                const std::size_t index = static_cast<std::size_t>(set) * count + k;
                hipOk(hipEventRecord(map->read[index], parts[index].stream));
                map->readRecorded[index] = 1;
            }
        sample += pending;
        pending = 0;
        pendingTarget = nullptr;
        ++passes;
    }

    // ---- the concurrent parts of a frame ------------------------------------------------------------------
    void destroyParts() {
        for (Part& part : parts) {
            if (part.wavefront)
                rt_wavefront_destroy(part.wavefront);
            if (part.stream)
                (void)hipStreamDestroy(part.stream);
            if (part.accumulated)
                (void)hipEventDestroy(part.accumulated);
        }
        parts.clear();
    }

    // Pixels one part must have room for: an even share of what render() covers.
    std::uint64_t neededCapacity() const {
        const std::uint64_t count = partCount;
        if (rangeSet)
            return std::max<std::uint64_t>(1, (rangeEnd - rangeBegin + count - 1U) / count);
        const std::uint64_t bands = (height + kBandRows - 1U) / kBandRows;
        const std::uint64_t mine = firstBand < bands ? (bands - 1U - firstBand) / bandStride + 1U : 0U;
        return std::max<std::uint64_t>(1, (mine + count - 1U) / count * kBandRows * width);
    }

    void ensureParts() {
        const std::uint64_t needed = neededCapacity();
        if (parts.empty() || needed > partCapacity || needed * 2U < partCapacity || (gatherWanted() && !partsGather))
            createParts(needed);
    }

    // Camera states carry the photon-query queues and the caustic sum only where a pass can gather: with the
    // final profile, or after a driver started a photon pass by hand in another.
    bool gatherWanted() const { return caustics || launch.photonRadius > 0.0f; }
    std::uint32_t cameraUses() const { return partsGather ? RT_WAVEFRONT_FOR_CAMERA : RT_WAVEFRONT_FOR_CAMERA_NO_CAUSTICS; }

    void dropParts() {
        hipOk(hipDeviceSynchronize());
        destroyParts();
        partCapacity = 0;
    }

    void createParts(std::uint64_t capacity) {
        hipOk(hipDeviceSynchronize());
        destroyParts();
        const std::uint32_t count = partCount;
        partCapacity = capacity;
        partsGather = gatherWanted();
        // Passes in flight: what was asked for, else two - with passes of several samples a second pass is
        // enough to cover the end of the first, and the state it saves goes into larger passes (1080p: 16 x 2
        // 1.95 ms per sample, 8 x 4 1.99) - or one, see below.
        depth = depthWanted ? depthWanted : 2U;
        // Samples per pass: what was asked for, else the most of 64, 32, ..., 1 whose state fits the budget - the
        // lesser of kStateBudget and half of the device memory that is free now, priced with the engine's own
        // figure for a state of that many paths.  With a photon map a pass holds at most kMapsPerPass groups of
        // four samples, each gathering from its group's map.
        batch = caustics ? std::min(batchWanted, kMapsPerPass * kPhotonRebuildInterval) : batchWanted;
        if (batch == 0) {
            std::size_t freeBytes = 0, totalBytes = 0;
            hipOk(hipMemGetInfo(&freeBytes, &totalBytes));
            const std::uint64_t budget = std::min<std::uint64_t>(kStateBudget, freeBytes / 2U);
            batch = caustics ? kMapsPerPass * kPhotonRebuildInterval : 64U;
            while (batch > 1 && stateBytes(capacity * batch) * count * depth > budget)
                batch /= 2;
            // the photon maps of the passes in flight and the lists of the passes traced together have a budget of
            // their own: a pass of 64 samples gathers from sixteen maps, two such passes from thirty-two (10 GiB at
            // the reference's sizes)
            const std::uint64_t photonBudget = std::min<std::uint64_t>(16ULL << 30, freeBytes / 4U);
            while (caustics && batch > kPhotonRebuildInterval && photonPoolBytes(batch) > photonBudget)
                batch /= 2;
            // Without a photon map a larger pass beats a second pass in flight (1080p: 32 x 1 1.65 ms per sample, 16 x 2
            // 1.69, 16 x 1 1.70): where two passes of the full 64 samples do not fit, the memory of the second goes
            // into one pass of twice the samples.  With a photon map the second pass stays: the photon passes of the
            // next camera pass run beside the current one (32 x 1 2.19 ms per sample, 16 x 2 2.01).
            if (!depthWanted && !caustics && batch < 64U && stateBytes(capacity * batch * 2U) * count <= budget) {
                batch *= 2U;
                depth = 1U;
            }
        }
        for (;;) {
            try {
                makeParts(capacity, count);
                break;
            } catch (const std::exception&) {
                // a state that does not fit after all (memory taken by others in the meantime): half the pass
                destroyParts();
                (void)hipGetLastError();
                if (batch == 1 || batchWanted != 0)
                    throw;
                batch /= 2;
            }
        }
        if (maps.size() != mapsWanted() || photonAt.photons.size() != listsWanted() ||
            (!maps.empty() && maps.front().read.size() != parts.size()))
            allocateMaps(mapsWanted());
        shareLightIndex();
    }

    // Device bytes of a camera state of `paths` paths, as the engine states them.
    std::uint64_t stateBytes(std::uint64_t paths) const {
        std::size_t bytes = 0;
        engineOk(rt_wavefront_state_bytes(static_cast<std::uint32_t>(std::min<std::uint64_t>(paths, 0xffffffffULL)),
                                          cameraUses(), &bytes),
                 "rt_wavefront_state_bytes");
        return bytes;
    }

    void makeParts(std::uint64_t capacity, std::uint32_t count) {
        parts.resize(static_cast<std::size_t>(count) * depth);
        if (parts.size() > 1)
            pickStreams();
        for (Part& part : parts) {
            if (capacity * batch > 0xffffffffULL)
                raise("sample batch: pixels x samples of a pass do not fit 32 bits");
            engineOk(rt_wavefront_create_for(static_cast<std::uint32_t>(capacity * batch), cameraUses(), &part.wavefront),
                     "rt_wavefront_create_for");
            engineOk(rt_wavefront_configure(part.wavefront, measurementFlags), "rt_wavefront_configure");
            // a single part stays on the default stream; parts that run side by side take half the wave
            // slots per ray launch: two of them then trace side by side, each filling the slots the
            // other's last rays leave idle.  Passes of four samples and more have enough rays to use every
            // slot themselves (measured: 2.39 against 2.42 ms per 1080p sample at eight per pass).
            if (shareNumerator)
                engineOk(rt_wavefront_set_ray_share(part.wavefront, shareNumerator, shareDenominator), "rt_wavefront_set_ray_share");
            else if (parts.size() > 1 && batch < 4U)
                engineOk(rt_wavefront_set_ray_share(part.wavefront, 1, 2), "rt_wavefront_set_ray_share");
            if (depth > 1)
                hipOk(hipEventCreateWithFlags(&part.accumulated, hipEventDisableTiming));
            // the stream's hardware queue sets up scratch memory for the ray kernels now, not in its first pass
            engineOk(rt_wavefront_prepare_stream(part.wavefront, part.stream), "rt_wavefront_prepare_stream");
        }
    }

    // One stream per part, chosen so that they run side by side.  A process has a few hardware queues
    // (four by default) which its streams are dealt onto in turn; when other code of the process (a
    // collective library, the caller's own streams) has taken some, two of a fresh set of streams can
    // land on one queue and their parts of the frame then run one after the other (measured: 5.3
    // instead of 3.7 ms per step).  So a set is probed, and a set that shares a queue is parked - which
    // moves the next set one place on - until one is clean or a handful of sets has been tried; the
    // last set is kept in that case.
    void pickStreams() {
        std::vector<hipStream_t> parked;
        for (int attempt = 0;; ++attempt) {
            std::vector<void*> set;
            for (Part& part : parts) {
                hipOk(hipStreamCreate(&part.stream));
                set.push_back(part.stream);
            }
            int concurrent = 0;
            engineOk(rt_streams_run_concurrently(set.data(), static_cast<std::uint32_t>(set.size()), &concurrent),
                     "rt_streams_run_concurrently");
            if (concurrent || attempt == 5)
                break;
            // keep one stream of the failed set alive so that the next set starts one queue further
            parked.push_back(parts[0].stream);
            for (std::size_t k = 1; k < parts.size(); ++k)
                (void)hipStreamDestroy(parts[k].stream);
            for (Part& part : parts)
                part.stream = nullptr;
        }
        for (hipStream_t stream : parked)
            (void)hipStreamDestroy(stream);
    }
};

Renderer::Renderer(const Scene& scene, std::uint32_t width, std::uint32_t height, Profile profile)
    : impl_(std::make_unique<Impl>()) {
    impl_->initialize(scene, width, height, profile);
}

Renderer::~Renderer() = default;

void Renderer::render(void* output) { impl_->render(output); }

void Renderer::denoise() { raise("denoise: the OptiX denoiser has no counterpart in this engine"); }

void Renderer::setCamera(const Camera& camera) {  // renderer.cpp:730-733
    impl_->applyCamera(camera);
    impl_->restart();
}

void Renderer::setProfile(Profile profile) {  // renderer.cpp:735-739
    impl_->applyProfile(profile);
    impl_->dropParts();  // the samples a pass takes depend on the profile
    impl_->restart();
}

void Renderer::copyOutput(void* output) const {  // renderer.cpp:888-891: device to device
    impl_->flush();
    hipOk(hipMemcpy(output, impl_->framePool.at<void>(impl_->outputOffset), impl_->outputBytes, hipMemcpyDeviceToDevice));
}

std::vector<std::uint8_t> Renderer::pixels() const {  // renderer.cpp:881-886
    impl_->flush();
    std::vector<std::uint8_t> image(impl_->outputBytes);
    hipOk(hipMemcpy(image.data(), impl_->framePool.at<void>(impl_->outputOffset), image.size(), hipMemcpyDeviceToHost));
    impl_->reportOverflow(false);
    return image;
}

std::vector<float> Renderer::linearPixels() const {  // renderer.cpp:893-899
    impl_->flush();
    std::vector<float> image(impl_->aovBytes / sizeof(float));
    hipOk(hipMemcpy(image.data(), impl_->framePool.at<void>(impl_->aovOffset[0]), impl_->aovBytes, hipMemcpyDeviceToHost));
    impl_->reportOverflow(false);
    return image;
}

std::uint32_t Renderer::samples() const { return impl_->sample + impl_->pending; }

void Renderer::setPhotonBudget(std::uint32_t emissions, std::uint32_t capacity, std::uint32_t buckets) {
    if (emissions == 0 || capacity == 0 || buckets == 0 || (buckets & (buckets - 1U)) != 0)
        raise("photon budget: counts must be positive and the bucket count a power of two");
    hipOk(hipDeviceSynchronize());
    impl_->setPhotonSizes(emissions, capacity, buckets);
    impl_->restart();
}

void Renderer::setSampleSequence(std::uint32_t first, std::uint32_t stride) {
    impl_->sampleFirst = first;
    impl_->sampleStride = stride;
    impl_->restart();
}

void Renderer::setPixelRange(std::uint32_t begin, std::uint32_t end) {
    if (end < begin || end > impl_->width * impl_->height)
        raise("pixel range outside the image");
    impl_->flush();  // calls that wait were made for the pixels covered so far
    impl_->rangeSet = true;
    impl_->rangeBegin = begin;
    impl_->rangeEnd = end;
}

void Renderer::setRowInterleave(std::uint32_t first, std::uint32_t stride) {
    if (stride == 0 || first >= stride)
        raise("row interleave: the stride must be positive and the first band below it");
    impl_->flush();
    impl_->rangeSet = false;
    impl_->firstBand = first;
    impl_->bandStride = stride;
}

void Renderer::setManualPhotonPasses(bool manual) { impl_->manualPhotonPasses = manual; }
bool Renderer::photonPassDue() const { return impl_->photonPassDue(); }
void Renderer::beginPhotonPass() { impl_->beginPhotonPass(); }
std::uint32_t Renderer::tracePhotonRange(std::uint32_t firstEmission, std::uint32_t emissionCount, bool intoStaging) {
    return impl_->tracePhotonRange(firstEmission, emissionCount, intoStaging);
}
void* Renderer::photonBuffer(std::uint32_t pass) const {
    if (pass >= impl_->photonAt.photons.size())
        raise("photon buffer: no such pass");
    return impl_->listOf(pass);
}
void* Renderer::photonStagingBuffer(std::uint32_t pass) const {
    if (pass >= impl_->currentMaps.size())
        raise("photon staging buffer: no such pass");
    return impl_->photonPool.at<void>(impl_->maps[impl_->currentMaps[pass]].sorted);
}
std::uint32_t Renderer::beginPhotonPasses(std::uint32_t wanted) { return impl_->beginPhotonPasses(wanted); }
std::uint32_t Renderer::photonPassesTogether() const {
    impl_->ensureParts();
    return static_cast<std::uint32_t>(impl_->photonAt.photons.size());
}
void Renderer::tracePhotonRangePasses(std::uint32_t firstEmission, std::uint32_t emissionCount, bool intoStaging,
                                      std::uint32_t* stored) {
    impl_->tracePhotonRangePasses(firstEmission, emissionCount, intoStaging, stored);
}
void Renderer::finishPhotonPasses(const std::uint32_t* totals) { impl_->finishPhotonPasses(totals); }
std::uint32_t Renderer::photonCapacity() const { return impl_->launch.photonCapacity; }
std::uint32_t Renderer::photonEmissions() const { return impl_->launch.photonEmissions; }
void Renderer::finishPhotonPass(std::uint32_t totalPhotons) { impl_->finishPhotonPass(totalPhotons); }

void* Renderer::accumulationBuffer(int index) const {
    if (index < 0 || index > 6)
        raise("accumulation buffer index");
    impl_->flush();  // what reads the buffer behind this call finds every sample on its way
    return impl_->framePool.at<void>(impl_->aovOffset[static_cast<std::size_t>(index)]);
}

void* Renderer::outputBuffer() const {
    impl_->flush();
    return impl_->framePool.at<void>(impl_->outputOffset);
}

void Renderer::statistics(std::uint64_t out[8]) {
    impl_->flush();
    std::fill(out, out + 8, std::uint64_t{0});
    auto add = [&](rt_wavefront* state, hipStream_t stream) {
        std::uint64_t part[8] = {};
        engineOk(rt_wavefront_stats(state, part, 1, stream), "rt_wavefront_stats");
        for (int k = 0; k < 8; ++k)
            out[k] += part[k];
    };
    for (auto& part : impl_->parts)
        add(part.wavefront, part.stream);
    if (impl_->photonWavefront)
        add(impl_->photonWavefront, impl_->photonStream);
}

void Renderer::setMeasurement(bool timeStages, bool countTraversal, bool gatherPerLane) {
    impl_->measurementFlags = (timeStages ? RT_WAVEFRONT_TIME_STAGES : 0U) | (countTraversal ? RT_WAVEFRONT_COUNT_TRAVERSAL : 0U) |
                              (gatherPerLane ? RT_WAVEFRONT_GATHER_PER_LANE : 0U);
    for (auto& part : impl_->parts)
        engineOk(rt_wavefront_configure(part.wavefront, impl_->measurementFlags), "rt_wavefront_configure");
}

void Renderer::stageTimings(float milliseconds[5], std::uint32_t launches[5]) {
    impl_->flush();
    std::fill(milliseconds, milliseconds + 5, 0.0f);
    std::fill(launches, launches + 5, 0U);
    for (auto& part : impl_->parts) {
        float ms[5] = {};
        std::uint32_t n[5] = {};
        engineOk(rt_wavefront_timings(part.wavefront, ms, n, part.stream), "rt_wavefront_timings");
        for (int k = 0; k < 5; ++k) {
            milliseconds[k] += ms[k];
            launches[k] += n[k];
        }
    }
}

void Renderer::setConcurrentTiles(std::uint32_t count, std::uint32_t samplesInFlight) {
    impl_->dropParts();
    impl_->partCount = std::clamp<std::uint32_t>(count, 1U, 8U);
    impl_->depthWanted = std::min<std::uint32_t>(samplesInFlight, 8U);
    impl_->shareNumerator = impl_->shareDenominator = 0;
    impl_->restart();
}

void Renderer::setSampleBatch(std::uint32_t samples) {
    if (samples > 64)
        raise("sample batch: at most 64 samples");
    impl_->dropParts();
    impl_->batchWanted = samples;
    impl_->restart();
}

std::uint32_t Renderer::sampleBatch() const {
    impl_->flush();
    impl_->ensureParts();
    return impl_->batch;
}

void Renderer::setRayShare(std::uint32_t numerator, std::uint32_t denominator) {
    if (numerator == 0 || denominator == 0 || numerator > denominator)
        raise("the share of a ray launch is a fraction in (0, 1]");
    impl_->shareNumerator = numerator;
    impl_->shareDenominator = denominator;
    for (auto& part : impl_->parts)
        engineOk(rt_wavefront_set_ray_share(part.wavefront, numerator, denominator), "rt_wavefront_set_ray_share");
}

std::uint32_t Renderer::concurrentTiles() const { return impl_->partCount; }
std::uint32_t Renderer::samplesInFlight() const {
    impl_->flush();
    impl_->ensureParts();
    return impl_->depth;
}
void Renderer::synchronize() const {
    impl_->flush();
    hipOk(hipDeviceSynchronize());
    impl_->reportOverflow(true);
}
const rt_launch_params& Renderer::launchParams() const { return impl_->launch; }
float Renderer::causticRadius() const { return impl_->gatherRadius; }

}  // namespace rt
