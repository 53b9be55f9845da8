// Host runtime of rt::Renderer on top of the engine's C ABI (include/rt_engine.h).
//
// Follows the reference's src/renderer.cpp step by step - uploads, per-geometry
// and instance acceleration builds, light table, photon buffers and constants,
// camera set-up, reset, photon pass, sample launch, read-backs - with the OptiX
// and CUDA calls replaced by the engine's entry points and the HIP runtime.
#include "renderer.hpp"

#include "photon_grid.hpp"
#include "rt_engine.h"

#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <stdexcept>
#include <string>
#include <utility>

namespace rt {
namespace {

void hipCheck(hipError_t result) {
    if (result != hipSuccess)
        throw std::runtime_error(hipGetErrorString(result));  // renderer.cpp:30-33
}

void engineCheck(rt_status status, const char* operation) {
    if (status != RT_SUCCESS)
        throw std::runtime_error(std::string(operation) + ": " + rt_last_error());  // renderer.cpp:35-39
}

// Move-only device allocation, renderer.cpp:41-93.
class Buffer {
public:
    Buffer() = default;
    explicit Buffer(std::size_t bytes) { resize(bytes); }
    ~Buffer() { reset(); }
    Buffer(Buffer&& other) noexcept : data_(std::exchange(other.data_, nullptr)), size_(std::exchange(other.size_, 0)) {}
    Buffer& operator=(Buffer&& other) noexcept {
        if (this != &other) {
            reset();
            data_ = std::exchange(other.data_, nullptr);
            size_ = std::exchange(other.size_, 0);
        }
        return *this;
    }
    Buffer(const Buffer&) = delete;
    Buffer& operator=(const Buffer&) = delete;

    void resize(std::size_t bytes) {
        if (bytes == size_)
            return;
        reset();
        if (bytes)
            hipCheck(hipMalloc(&data_, bytes));
        size_ = bytes;
    }
    void upload(const void* source, std::size_t bytes) {
        resize(bytes);
        if (bytes)
            hipCheck(hipMemcpy(data_, source, bytes, hipMemcpyHostToDevice));
    }
    void* data() const { return data_; }
    std::size_t size() const { return size_; }

private:
    void reset() {
        if (data_)
            (void)hipFree(data_);
        data_ = nullptr;
        size_ = 0;
    }
    void* data_ = nullptr;
    std::size_t size_ = 0;
};

rt_float3 f3(float x, float y, float z) { return rt_float3{x, y, z}; }

rt_float3 normalized(float x, float y, float z) {  // renderer.cpp:103-106
    const float inverse = 1.0f / std::sqrt(x * x + y * y + z * z);
    return f3(x * inverse, y * inverse, z * inverse);
}

rt_gpu_texture_ref gpuTextureRef(const TextureRef& source) {  // renderer.cpp:108-119
    if (source.texture >= 0 && source.texCoord > 1)
        throw std::runtime_error("Only glTF TEXCOORD_0 and TEXCOORD_1 are supported");
    rt_gpu_texture_ref ref;
    ref.texture = source.texture;
    ref.texCoord = source.texCoord;
    ref.offset.x = source.offset.x;
    ref.offset.y = source.offset.y;
    ref.scale.x = source.scale.x;
    ref.scale.y = source.scale.y;
    ref.rotation = source.rotation;
    ref.strength = source.strength;
    return ref;
}

rt_gpu_material gpuMaterial(const Material& source) {  // renderer.cpp:121-143
    rt_gpu_material m;
    std::memset(&m, 0, sizeof(m));
    m.baseColor.x = source.baseColor.x;
    m.baseColor.y = source.baseColor.y;
    m.baseColor.z = source.baseColor.z;
    m.baseColor.w = source.baseColor.w;
    m.emissive = f3(source.emissive.x, source.emissive.y, source.emissive.z);
    m.attenuationColor = f3(source.attenuationColor.x, source.attenuationColor.y, source.attenuationColor.z);
    m.metallic = source.metallic;
    m.roughness = source.roughness;
    m.transmission = source.transmission;
    m.ior = source.ior;
    m.thickness = source.thickness;
    m.attenuationDistance = source.attenuationDistance;
    m.emissiveStrength = source.emissiveStrength;
    m.dispersion = source.dispersion;
    m.baseColorTexture = gpuTextureRef(source.baseColorTexture);
    m.metallicRoughnessTexture = gpuTextureRef(source.metallicRoughnessTexture);
    m.normalTexture = gpuTextureRef(source.normalTexture);
    m.emissiveTexture = gpuTextureRef(source.emissiveTexture);
    m.transmissionTexture = gpuTextureRef(source.transmissionTexture);
    m.thicknessTexture = gpuTextureRef(source.thicknessTexture);
    return m;
}

rt_float3 point(const Mat4& matrix, const Vec3& value) {  // renderer.cpp:145-151
    const auto& m = matrix.values;
    return f3(m[0] * value.x + m[4] * value.y + m[8] * value.z + m[12],
              m[1] * value.x + m[5] * value.y + m[9] * value.z + m[13],
              m[2] * value.x + m[6] * value.y + m[10] * value.z + m[14]);
}

rt_float3 difference(rt_float3 a, rt_float3 b) { return f3(a.x - b.x, a.y - b.y, a.z - b.z); }

rt_float3 crossProduct(rt_float3 a, rt_float3 b) {
    return f3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}

float length(rt_float3 v) { return std::sqrt(v.x * v.x + v.y * v.y + v.z * v.z); }

float luminance(rt_float3 v) { return 0.2126f * v.x + 0.7152f * v.y + 0.0722f * v.z; }  // renderer.cpp:166-168

constexpr float CausticRadiusPixels = 1.0f;          // renderer.cpp:173
constexpr std::uint32_t PhotonRebuildInterval = 4U;  // renderer.cpp:177

}  // namespace

struct Renderer::Impl {
    struct GeometryState {
        Buffer vertices;
        Buffer indices;
        rt_blas* blas = nullptr;
        std::uint32_t material = 0;
    };
    struct ImageState {
        Buffer pixels;
        std::uint32_t width = 0;
        std::uint32_t height = 0;
    };

    ~Impl() {
        (void)hipDeviceSynchronize();
        destroyTiles();
        if (photonWavefront)
            rt_wavefront_destroy(photonWavefront);
        if (sceneHandle)
            rt_scene_destroy(sceneHandle);
        for (auto& g : geometries)
            if (g.blas)
                rt_blas_destroy(g.blas);
    }

    void initialize(const Scene& scene, std::uint32_t renderWidth, std::uint32_t renderHeight, Profile profile) {
        if (scene.instances.empty())
            throw std::runtime_error("The scene has no geometry");  // renderer.cpp:217-218
        if (scene.cameras.empty())
            throw std::runtime_error(
                "The scene has no perspective camera. Blender's File > Export > glTF "
                "leaves Cameras unchecked by default; use the Raytracer panel's "
                "Export GLB instead");  // renderer.cpp:219-223
        if (renderWidth == 0 || renderHeight == 0)
            throw std::runtime_error("The image has no pixels");
        checkReferences(scene);
        width = renderWidth;
        height = renderHeight;
        std::memset(&launch, 0, sizeof(launch));
        launch.maxDepth = profile == Profile::Final ? 32U : 5U;  // renderer.cpp:226
        hipCheck(hipFree(nullptr));

        uploadTextures(scene);
        uploadMaterials(scene);
        uploadEnvironment(scene);
        buildGeometry(scene);
        buildInstances(scene);
        buildLights(scene);
        configureCaustics(scene);
        createBindingTable();
        const std::size_t pixels = static_cast<std::size_t>(width) * height;
        for (auto& buffer : aov)
            buffer.resize(pixels * sizeof(rt_float4));
        output.resize(pixels * sizeof(rt_uchar4));
        createTiles(DefaultTiles);
        pixelBegin = 0;
        pixelEnd = static_cast<std::uint32_t>(pixels);
        causticsEnabled = profile == Profile::Final;  // renderer.cpp:251
        configureCamera(scene.cameras.front());
        reset();
    }

    // Every index one table of the scene holds into another (instance -> geometry ->
    // material -> texture -> image, triangle -> vertex) is checked here: the device
    // code follows them without looking, and a bad one is better an exception than
    // a read outside a buffer.  (The reference relies on its loader for this.)
    static void checkReferences(const Scene& scene) {
        auto fail = [](const std::string& what, const std::string& name) {
            throw std::runtime_error(what + (name.empty() ? std::string() : " (" + name + ")"));
        };
        for (const auto& instance : scene.instances)
            if (instance.geometry >= scene.geometries.size())
                fail("An instance refers to a geometry the scene does not have", instance.name);
        for (const auto& geometry : scene.geometries) {
            if (geometry.material >= scene.materials.size())
                fail("A geometry refers to a material the scene does not have", geometry.name);
            if (geometry.indices.size() % 3 != 0)
                fail("The index count of a geometry is not a multiple of three", geometry.name);
            for (const std::uint32_t index : geometry.indices)
                if (index >= geometry.vertices.size())
                    fail("A triangle refers to a vertex its geometry does not have", geometry.name);
        }
        for (const auto& material : scene.materials)
            for (const TextureRef* ref : {&material.baseColorTexture, &material.metallicRoughnessTexture,
                                          &material.normalTexture, &material.emissiveTexture,
                                          &material.transmissionTexture, &material.thicknessTexture})
                if (ref->texture >= 0 && static_cast<std::size_t>(ref->texture) >= scene.textures.size())
                    fail("A material refers to a texture the scene does not have", material.name);
        for (const auto& texture : scene.textures)
            if (texture.image < 0 || static_cast<std::size_t>(texture.image) >= scene.images.size())
                fail("A texture refers to an image the scene does not have", texture.name);
    }

    void uploadMaterials(const Scene& scene) {  // renderer.cpp:256-262
        std::vector<rt_gpu_material> gpuMaterials;
        gpuMaterials.reserve(scene.materials.size());
        for (const auto& material : scene.materials)
            gpuMaterials.push_back(gpuMaterial(material));
        if (gpuMaterials.empty())
            throw std::runtime_error("The scene has no materials");
        materials.upload(gpuMaterials.data(), gpuMaterials.size() * sizeof(rt_gpu_material));
    }

    void uploadTextures(const Scene& scene) {  // renderer.cpp:264-299
        images.reserve(scene.images.size());
        for (const auto& source : scene.images) {
            if (source.rgba.empty() || source.rgba.size() != static_cast<std::size_t>(source.width) * source.height * 4)
                throw std::runtime_error("Unable to decode texture " + source.name +
                                         ": this build takes decoded RGBA8 pixels only");  // renderer.cpp:275-277
            auto& image = images.emplace_back();
            image.width = source.width;
            image.height = source.height;
            image.pixels.upload(source.rgba.data(), source.rgba.size());
        }
        std::vector<rt_gpu_texture> records;
        records.reserve(scene.textures.size());
        for (const auto& source : scene.textures) {
            const auto& image = images.at(static_cast<std::size_t>(source.image));
            rt_gpu_texture record;
            record.pixels = static_cast<const rt_uchar4*>(image.pixels.data());
            record.width = image.width;
            record.height = image.height;
            record.wrapU = source.wrapU;
            record.wrapV = source.wrapV;
            records.push_back(record);
        }
        if (!records.empty()) {
            textures.upload(records.data(), records.size() * sizeof(rt_gpu_texture));
            launch.textures = static_cast<const rt_gpu_texture*>(textures.data());
        }
    }

    void uploadEnvironment(const Scene& scene) {  // renderer.cpp:301-341
        const auto& source = scene.environment;
        launch.environmentRotation = source.rotation;
        launch.environmentStrength = source.strength;
        launch.exposure = source.exposure;
        if (source.pixels.empty())
            return;
        std::vector<rt_float4> pixels;
        std::vector<float> cdf;
        pixels.reserve(source.pixels.size());
        cdf.reserve(source.pixels.size());
        double total = 0.0;
        for (std::uint32_t y = 0; y < source.height; ++y) {
            const float sine = std::sin(3.14159265f * (static_cast<float>(y) + 0.5f) / source.height);
            for (std::uint32_t x = 0; x < source.width; ++x) {
                const auto& pixel = source.pixels[y * source.width + x];
                rt_float4 value;
                value.x = pixel.x;
                value.y = pixel.y;
                value.z = pixel.z;
                value.w = pixel.w;
                pixels.push_back(value);
                total += luminance(f3(pixel.x, pixel.y, pixel.z)) * sine;
                cdf.push_back(static_cast<float>(total));
            }
        }
        if (total <= 0.0)
            return;
        for (auto& value : cdf)
            value = static_cast<float>(value / total);
        environment.upload(pixels.data(), pixels.size() * sizeof(rt_float4));
        environmentCdf.upload(cdf.data(), cdf.size() * sizeof(float));
        launch.environment = static_cast<const rt_float4*>(environment.data());
        launch.environmentCdf = static_cast<const float*>(environmentCdf.data());
        launch.environmentWidth = source.width;
        launch.environmentHeight = source.height;
        environmentPower = static_cast<float>(total * 19.7392088 / static_cast<double>(source.width * source.height) *
                                              source.strength);
        launch.environmentWeight = environmentPower;
    }

    void buildGeometry(const Scene& scene) {  // renderer.cpp:343-396
        geometries.reserve(scene.geometries.size());
        for (const auto& geometry : scene.geometries) {
            auto& state = geometries.emplace_back();
            std::vector<rt_gpu_vertex> vertices;
            vertices.reserve(geometry.vertices.size());
            for (const auto& vertex : geometry.vertices) {
                rt_gpu_vertex v;
                std::memset(&v, 0, sizeof(v));
                v.position = f3(vertex.position.x, vertex.position.y, vertex.position.z);
                v.normal = f3(vertex.normal.x, vertex.normal.y, vertex.normal.z);
                v.tangent.x = vertex.tangent.x;
                v.tangent.y = vertex.tangent.y;
                v.tangent.z = vertex.tangent.z;
                v.tangent.w = vertex.tangent.w;
                v.uv.x = vertex.uv.x;
                v.uv.y = vertex.uv.y;
                v.uv1.x = vertex.uv1.x;
                v.uv1.y = vertex.uv1.y;
                vertices.push_back(v);
            }
            state.vertices.upload(vertices.data(), vertices.size() * sizeof(rt_gpu_vertex));
            state.indices.upload(geometry.indices.data(), geometry.indices.size() * sizeof(std::uint32_t));
            state.material = geometry.material;
            engineCheck(rt_blas_build(state.vertices.data(), static_cast<std::uint32_t>(geometry.vertices.size()),
                                      sizeof(rt_gpu_vertex), static_cast<const std::uint32_t*>(state.indices.data()),
                                      static_cast<std::uint32_t>(geometry.indices.size() / 3), nullptr, &state.blas),
                        "rt_blas_build");
        }
        hipCheck(hipDeviceSynchronize());
    }

    void buildInstances(const Scene& scene) {  // renderer.cpp:398-434
        std::vector<rt_instance_desc> instances(scene.instances.size());
        for (std::size_t index = 0; index < scene.instances.size(); ++index) {
            const auto& source = scene.instances[index];
            auto& target = instances[index];
            for (std::size_t row = 0; row < 3; ++row)
                for (std::size_t column = 0; column < 4; ++column)
                    target.transform[row * 4 + column] = source.transform.values[column * 4 + row];
            target.instanceId = static_cast<std::uint32_t>(index);
            target.geometry = source.geometry;
            target.blas = geometries.at(source.geometry).blas;
        }
        engineCheck(rt_scene_build(instances.data(), static_cast<std::uint32_t>(instances.size()), nullptr, &sceneHandle),
                    "rt_scene_build");
        hipCheck(hipDeviceSynchronize());
    }

    void buildLights(const Scene& scene) {  // renderer.cpp:436-507
        std::vector<rt_gpu_light> source;
        for (std::size_t instanceIndex = 0; instanceIndex < scene.instances.size(); ++instanceIndex) {
            const auto& instance = scene.instances[instanceIndex];
            const auto& geometry = scene.geometries[instance.geometry];
            const auto& material = scene.materials[geometry.material];
            const rt_float3 emission = f3(material.emissive.x * material.emissiveStrength,
                                          material.emissive.y * material.emissiveStrength,
                                          material.emissive.z * material.emissiveStrength);
            if (luminance(emission) <= 0.0f)
                continue;
            for (std::size_t index = 0; index < geometry.indices.size(); index += 3) {
                const rt_float3 a = point(instance.transform, geometry.vertices[geometry.indices[index]].position);
                const rt_float3 b = point(instance.transform, geometry.vertices[geometry.indices[index + 1]].position);
                const rt_float3 c = point(instance.transform, geometry.vertices[geometry.indices[index + 2]].position);
                const rt_float3 areaVector = crossProduct(difference(b, a), difference(c, a));
                const float twiceArea = length(areaVector);
                if (twiceArea <= 1e-8f)
                    continue;
                const float area = 0.5f * twiceArea;
                rt_gpu_light light;
                std::memset(&light, 0, sizeof(light));
                light.a = a;
                light.b = b;
                light.c = c;
                light.normal = f3(areaVector.x / twiceArea, areaVector.y / twiceArea, areaVector.z / twiceArea);
                light.emission = emission;
                light.area = area;
                light.weight = area * luminance(emission);
                light.instance = static_cast<std::uint32_t>(instanceIndex);
                light.primitive = static_cast<std::uint32_t>(index / 3);
                light.type = 3U;
                source.push_back(light);
            }
        }
        for (const auto& light : scene.lights) {
            const auto& m = light.transform.values;
            const rt_float3 emission =
                f3(light.color.x * light.intensity, light.color.y * light.intensity, light.color.z * light.intensity);
            const rt_float3 direction = normalized(-m[8], -m[9], -m[10]);
            const float solidAngle = light.type == 1U ? 6.2831853f * (1.0f - std::cos(light.outerCone))
                                     : light.type == 2U ? 12.566371f
                                                        : 1.0f;
            rt_gpu_light record;
            std::memset(&record, 0, sizeof(record));
            record.a = f3(m[12], m[13], m[14]);
            record.b = direction;
            record.emission = emission;
            record.weight = luminance(emission) * solidAngle;
            record.range = light.range;
            record.innerCone = light.innerCone;
            record.outerCone = light.outerCone;
            record.instance = 0xffffffffU;
            record.primitive = 0xffffffffU;
            record.type = light.type;
            source.push_back(record);
        }
        if (environmentPower > 0.0f) {
            rt_gpu_light record;
            std::memset(&record, 0, sizeof(record));
            record.weight = environmentPower;
            record.instance = 0xffffffffU;
            record.primitive = 0xffffffffU;
            record.type = 4U;
            source.push_back(record);
        }
        launch.lightWeight = 0.0f;
        for (const auto& light : source)
            launch.lightWeight += light.weight;
        launch.lightCount = static_cast<std::uint32_t>(source.size());
        if (!source.empty()) {
            lights.upload(source.data(), source.size() * sizeof(rt_gpu_light));
            launch.lights = static_cast<const rt_gpu_light*>(lights.data());
        }
    }

    void configureCaustics(const Scene& scene) {  // renderer.cpp:509-551
        const float limit = std::numeric_limits<float>::max();
        rt_float3 minimum = f3(limit, limit, limit);
        rt_float3 maximum = f3(-limit, -limit, -limit);
        for (const auto& instance : scene.instances)
            for (const auto& vertex : scene.geometries[instance.geometry].vertices) {
                const rt_float3 value = point(instance.transform, vertex.position);
                minimum = f3(std::min(minimum.x, value.x), std::min(minimum.y, value.y), std::min(minimum.z, value.z));
                maximum = f3(std::max(maximum.x, value.x), std::max(maximum.y, value.y), std::max(maximum.z, value.z));
            }
        launch.sceneCenter =
            f3((minimum.x + maximum.x) * 0.5f, (minimum.y + maximum.y) * 0.5f, (minimum.z + maximum.z) * 0.5f);
        launch.sceneRadius = length(difference(maximum, minimum)) * 0.525f;
        if (launch.photonBucketCount == 0U) {
            launch.photonBucketCount = 1U << 22U;
            launch.photonCapacity = 1U << 23U;
            launch.photonEmissions = 1U << 24U;
        }
        allocatePhotonBuffers();
    }

    void allocatePhotonBuffers() {  // renderer.cpp:537-550
        const auto counterBytes = static_cast<std::size_t>(launch.photonBucketCount + 1U) * sizeof(std::uint32_t);
        photons.resize(static_cast<std::size_t>(launch.photonCapacity) * sizeof(rt_photon));
        sortedPhotons.resize(photons.size());
        photonCellCounts.resize(counterBytes);
        photonCellStart.resize(counterBytes);
        photonCounter.resize(sizeof(std::uint32_t));
        photonScan.resize(photonScanStorage(launch.photonBucketCount));
        // until the first build every bucket is empty
        hipCheck(hipMemset(photonCellStart.data(), 0, counterBytes));
        // one bit per bucket, set where the bucket holds photons: the gather asks it before a range
        const std::size_t bitBytes = std::max<std::size_t>(1, launch.photonBucketCount / 32U) * sizeof(std::uint32_t);
        photonBucketBits.resize(bitBytes);
        hipCheck(hipMemset(photonBucketBits.data(), 0, bitBytes));
        bindBucketBits();
        launch.photons = static_cast<rt_photon*>(photons.data());
        launch.photonCount = static_cast<std::uint32_t*>(photonCounter.data());
        launch.sortedPhotons = static_cast<const rt_photon*>(sortedPhotons.data());
        launch.photonCellStart = static_cast<const std::uint32_t*>(photonCellStart.data());
    }

    void bindBucketBits() {
        for (auto& t : tiles)
            if (t.wavefront)
                engineCheck(rt_wavefront_set_bucket_bitmap(
                                t.wavefront, static_cast<const std::uint32_t*>(photonBucketBits.data())),
                            "rt_wavefront_set_bucket_bitmap");
    }

    void createBindingTable() {  // renderer.cpp:633-670: one record per geometry
        std::vector<rt_hit_group_data> records(geometries.size());
        for (std::size_t index = 0; index < geometries.size(); ++index) {
            records[index].vertices = static_cast<const rt_gpu_vertex*>(geometries[index].vertices.data());
            records[index].indices = static_cast<const rt_uint3*>(geometries[index].indices.data());
            records[index].material = geometries[index].material;
        }
        engineCheck(rt_scene_set_geometry(sceneHandle, records.data(), static_cast<std::uint32_t>(records.size())),
                    "rt_scene_set_geometry");
    }

    void configureCamera(const Camera& camera) {  // renderer.cpp:672-714
        const auto& matrix = camera.transform.values;
        const rt_float3 right = normalized(matrix[0], matrix[1], matrix[2]);
        const rt_float3 up = normalized(matrix[4], matrix[5], matrix[6]);
        const rt_float3 direction = normalized(matrix[8], matrix[9], matrix[10]);
        const rt_float3 forward = f3(-direction.x, -direction.y, -direction.z);
        const float aspect =
            camera.aspectRatio > 0.0f ? camera.aspectRatio : static_cast<float>(width) / static_cast<float>(height);
        const float scale = std::tan(camera.verticalFov * 0.5f);
        launch.eye = f3(matrix[12], matrix[13], matrix[14]);
        launch.cameraU = f3(right.x * scale * aspect, right.y * scale * aspect, right.z * scale * aspect);
        launch.cameraV = f3(up.x * scale, up.y * scale, up.z * scale);
        launch.cameraW = forward;
        launch.lensU = right;
        launch.lensV = up;
        launch.aperture = camera.aperture;
        launch.focusDistance = camera.focusDistance;
        const float depth = std::clamp(camera.focusDistance, camera.nearPlane, launch.sceneRadius * 2.0f);
        causticRadius =
            std::max(2.0f * CausticRadiusPixels * depth * scale / static_cast<float>(height), 0.0005f);
        launch.width = width;
        launch.height = height;
        launch.accumulation = static_cast<rt_float4*>(aov[0].data());
        launch.diffuse = static_cast<rt_float4*>(aov[1].data());
        launch.reflection = static_cast<rt_float4*>(aov[2].data());
        launch.refraction = static_cast<rt_float4*>(aov[3].data());
        launch.caustics = static_cast<rt_float4*>(aov[4].data());
        launch.albedoGuide = static_cast<rt_float4*>(aov[5].data());
        launch.normalGuide = static_cast<rt_float4*>(aov[6].data());
        launch.materials = static_cast<const rt_gpu_material*>(materials.data());
        launch.textures = static_cast<const rt_gpu_texture*>(textures.data());
        launch.scene = rt_scene_handle(sceneHandle);
    }

    void reset() {  // renderer.cpp:716-728
        for (auto& buffer : aov)
            hipCheck(hipMemset(buffer.data(), 0, buffer.size()));
        launch.photonPass = 0;
        launch.photonRadius = causticsEnabled ? causticRadius : 0.0f;
        sample = 0;
    }

    void beginPhotonPass() {  // renderer.cpp:742-745
        launch.photonRadius = causticRadius;
        hipCheck(hipMemset(photonCounter.data(), 0, photonCounter.size()));
    }

    std::uint32_t tracePhotonRange(std::uint32_t first, std::uint32_t count) {  // renderer.cpp:746-769
        // The photon pass has wavefront state of its own, sized for a whole pass (at
        // most 2^24 paths, 328 bytes each): every batch ends with the drain of its
        // ray launches, so few large batches beat many frame-sized ones.
        const std::uint32_t wanted = std::min<std::uint32_t>(std::max<std::uint32_t>(count, 1U), 1U << 24);
        if (!photonWavefront || photonWavefrontCapacity < wanted) {
            hipCheck(hipDeviceSynchronize());
            if (photonWavefront)
                rt_wavefront_destroy(photonWavefront);
            photonWavefront = nullptr;
            engineCheck(rt_wavefront_create(wanted, &photonWavefront), "rt_wavefront_create");
            photonWavefrontCapacity = wanted;
        }
        engineCheck(rt_trace_photons(photonWavefront, &launch, first, count, nullptr), "rt_trace_photons");
        hipCheck(hipDeviceSynchronize());
        std::uint32_t stored = 0;
        hipCheck(hipMemcpy(&stored, photonCounter.data(), sizeof(stored), hipMemcpyDeviceToHost));
        if (stored > launch.photonCapacity && !photonOverflowWarned) {
            photonOverflowWarned = true;
            std::fprintf(stderr,
                         "warning: photon buffer full (%u of %u stored); "
                         "the caustic will be darker than it should be\n",
                         stored, launch.photonCapacity);
        }
        return std::min(stored, launch.photonCapacity);
    }

    void finishPhotonPass(std::uint32_t stored) {  // renderer.cpp:770-776
        stored = std::min(stored, launch.photonCapacity);
        const int status = buildPhotonGrid(
            static_cast<const Photon*>(photons.data()), stored, launch.photonRadius, launch.photonBucketCount,
            static_cast<std::uint32_t*>(photonCellCounts.data()), static_cast<std::uint32_t*>(photonCellStart.data()),
            photonScan.data(), photonScan.size(), static_cast<Photon*>(sortedPhotons.data()));
        engineCheck(status, "buildPhotonGrid");
        engineCheck(rt_photon_bucket_bitmap(static_cast<const std::uint32_t*>(photonCellStart.data()),
                                            launch.photonBucketCount,
                                            static_cast<std::uint32_t*>(photonBucketBits.data()), nullptr),
                    "rt_photon_bucket_bitmap");
        ++launch.photonPass;
        photonPassPending = false;
    }

    bool photonPassDue() const {  // renderer.cpp:780
        return launch.photonRadius > 0.0f && sample % PhotonRebuildInterval == 0U;
    }

    void buildCaustics() {  // renderer.cpp:741-777
        beginPhotonPass();
        const std::uint32_t stored = tracePhotonRange(0, launch.photonEmissions);
        finishPhotonPass(stored);
    }

    void render(void* target) {  // renderer.cpp:779-796
        if (photonPassDue()) {
            if (!manualPhotonPasses)
                buildCaustics();
            else if (photonPassPending)
                throw std::runtime_error("a photon pass is due: run beginPhotonPass/tracePhotonRange/finishPhotonPass");
        }
        launch.display = nullptr;
        launch.display2 = nullptr;
        launch.display3 = nullptr;
        launch.composite = nullptr;
        launch.output = static_cast<rt_uchar4*>(target ? target : output.data());
        launch.sample = sample;
        // The pixel range is cut into one contiguous part per tile; the parts run
        // on their own streams, so the slow rays that end one part's ray launch
        // overlap with the other parts' work instead of idling the GPU.  The
        // streams are ordinary (blocking) ones: work on the default stream before
        // and after still orders against them.
        const std::uint32_t count = pixelEnd - pixelBegin;
        const std::uint32_t parts = static_cast<std::uint32_t>(tiles.size());
        for (std::uint32_t k = 0; k < parts; ++k) {
            const std::uint32_t begin = pixelBegin + static_cast<std::uint32_t>(std::uint64_t(count) * k / parts);
            const std::uint32_t end = pixelBegin + static_cast<std::uint32_t>(std::uint64_t(count) * (k + 1) / parts);
            engineCheck(rt_render_sample_seeded(tiles[k].wavefront, &launch, begin, end,
                                                sampleFirst + sample * sampleStride, tiles[k].stream),
                        "rt_render_sample");
        }
        ++sample;
        photonPassPending = true;
    }

    std::uint32_t width = 0;
    std::uint32_t height = 0;
    std::uint32_t sample = 0;
    std::uint32_t pixelBegin = 0, pixelEnd = 0;
    std::uint32_t sampleFirst = 0, sampleStride = 1;
    rt_scene* sceneHandle = nullptr;
    // Wavefront state and stream per concurrent part of the frame.
    struct Tile {
        rt_wavefront* wavefront = nullptr;
        hipStream_t stream = nullptr;
    };
    static constexpr std::uint32_t DefaultTiles = 3;
    std::vector<Tile> tiles;
    rt_wavefront* photonWavefront = nullptr;  // created by the first photon pass
    std::uint32_t photonWavefrontCapacity = 0;
    std::uint32_t measurementFlags = 0;

    void destroyTiles() {
        for (auto& t : tiles) {
            if (t.wavefront)
                rt_wavefront_destroy(t.wavefront);
            if (t.stream)
                (void)hipStreamDestroy(t.stream);
        }
        tiles.clear();
    }

    void createTiles(std::uint32_t count) {
        hipCheck(hipDeviceSynchronize());
        destroyTiles();
        const std::size_t pixels = static_cast<std::size_t>(width) * height;
        count = std::max<std::uint32_t>(1U, std::min<std::uint32_t>(count, 8U));
        const std::size_t capacity = (pixels + count - 1) / count;
        tiles.resize(count);
        if (count > 1)
            pickStreams();
        for (auto& t : tiles) {
            engineCheck(rt_wavefront_create(static_cast<std::uint32_t>(capacity), &t.wavefront), "rt_wavefront_create");
            engineCheck(rt_wavefront_configure(t.wavefront, measurementFlags), "rt_wavefront_configure");
            // a single tile stays on the default stream; the parts of a split frame take half the
            // wave slots per ray launch: two of them then trace side by side, each filling the slots
            // the other's last rays leave idle
            if (count > 1)
                engineCheck(rt_wavefront_set_ray_share(t.wavefront, 1, 2), "rt_wavefront_set_ray_share");
        }
        bindBucketBits();
    }

    // One stream per tile, chosen so that they run side by side.  A process has a
    // few hardware queues (four by default) which its streams are dealt onto in
    // turn; when other code of the process (a collective library, the caller's own
    // streams) has taken some, two of a fresh set of streams can land on one queue
    // and their parts of the frame then run one after the other (measured: 5.3
    // instead of 3.7 ms per step).  So a set is probed, and a set that shares a
    // queue is parked - which moves the next set one place on - until one is clean
    // or a handful of sets has been tried; the last set is kept in that case.
    void pickStreams() {
        std::vector<hipStream_t> parked;
        for (int attempt = 0;; ++attempt) {
            std::vector<void*> set;
            for (auto& t : tiles) {
                hipCheck(hipStreamCreate(&t.stream));
                set.push_back(t.stream);
            }
            int concurrent = 0;
            engineCheck(rt_streams_run_concurrently(set.data(), static_cast<std::uint32_t>(set.size()), &concurrent),
                        "rt_streams_run_concurrently");
            if (concurrent || attempt == 5)
                break;
            // keep one stream of the failed set alive so that the next set starts one queue further
            parked.push_back(tiles[0].stream);
            for (std::size_t k = 1; k < tiles.size(); ++k)
                (void)hipStreamDestroy(tiles[k].stream);
            for (auto& t : tiles)
                t.stream = nullptr;
        }
        for (hipStream_t s : parked)
            (void)hipStreamDestroy(s);
    }

    std::vector<GeometryState> geometries;
    std::vector<ImageState> images;
    Buffer materials, textures, lights, environment, environmentCdf;
    Buffer aov[7];  // accumulation, diffuse, reflection, refraction, caustics, albedo guide, normal guide
    Buffer output;
    Buffer photons, sortedPhotons, photonCellCounts, photonCellStart, photonCounter, photonScan, photonBucketBits;
    rt_launch_params launch;
    float environmentPower = 0.0f;
    float causticRadius = 0.0f;
    bool causticsEnabled = false;
    bool photonOverflowWarned = false;
    bool manualPhotonPasses = false;
    bool photonPassPending = true;
};

Renderer::Renderer(const Scene& scene, std::uint32_t width, std::uint32_t height, Profile profile)
    : impl_(std::make_unique<Impl>()) {
    impl_->initialize(scene, width, height, profile);
}

Renderer::~Renderer() = default;

void Renderer::render(void* output) { impl_->render(output); }

void Renderer::denoise() {
    throw std::runtime_error("denoise: the OptiX denoiser has no counterpart in this engine");
}

void Renderer::setCamera(const Camera& camera) {  // renderer.cpp:730-733
    impl_->configureCamera(camera);
    impl_->reset();
}

void Renderer::setProfile(Profile profile) {  // renderer.cpp:735-739
    impl_->launch.maxDepth = profile == Profile::Final ? 32U : 5U;
    impl_->causticsEnabled = profile == Profile::Final;
    impl_->reset();
}

void Renderer::copyOutput(void* output) const {  // renderer.cpp:888-891
    hipCheck(hipMemcpy(output, impl_->output.data(), impl_->output.size(), hipMemcpyDeviceToDevice));
}

std::vector<std::uint8_t> Renderer::pixels() const {  // renderer.cpp:881-886
    std::vector<std::uint8_t> result(impl_->output.size());
    hipCheck(hipMemcpy(result.data(), impl_->output.data(), impl_->output.size(), hipMemcpyDeviceToHost));
    return result;
}

std::vector<float> Renderer::linearPixels() const {  // renderer.cpp:893-899
    const Buffer& source = impl_->aov[0];
    std::vector<float> result(source.size() / sizeof(float));
    hipCheck(hipMemcpy(result.data(), source.data(), source.size(), hipMemcpyDeviceToHost));
    return result;
}

std::uint32_t Renderer::samples() const { return impl_->sample; }

void Renderer::setPhotonBudget(std::uint32_t emissions, std::uint32_t capacity, std::uint32_t buckets) {
    if (emissions == 0 || capacity == 0 || buckets == 0 || (buckets & (buckets - 1U)) != 0)
        throw std::runtime_error("photon budget: counts must be positive and the bucket count a power of two");
    hipCheck(hipDeviceSynchronize());
    impl_->launch.photonEmissions = emissions;
    impl_->launch.photonCapacity = capacity;
    impl_->launch.photonBucketCount = buckets;
    impl_->allocatePhotonBuffers();
    impl_->reset();
}

void Renderer::setSampleSequence(std::uint32_t first, std::uint32_t stride) {
    impl_->sampleFirst = first;
    impl_->sampleStride = stride;
    impl_->reset();
}

void Renderer::setPixelRange(std::uint32_t begin, std::uint32_t end) {
    if (end < begin || end > impl_->width * impl_->height)
        throw std::runtime_error("pixel range outside the image");
    impl_->pixelBegin = begin;
    impl_->pixelEnd = end;
}

void Renderer::setManualPhotonPasses(bool manual) { impl_->manualPhotonPasses = manual; }
bool Renderer::photonPassDue() const { return impl_->photonPassDue(); }
void Renderer::beginPhotonPass() { impl_->beginPhotonPass(); }
std::uint32_t Renderer::tracePhotonRange(std::uint32_t firstEmission, std::uint32_t emissionCount) {
    return impl_->tracePhotonRange(firstEmission, emissionCount);
}
void* Renderer::photonBuffer() const { return impl_->photons.data(); }
std::uint32_t Renderer::photonCapacity() const { return impl_->launch.photonCapacity; }
std::uint32_t Renderer::photonEmissions() const { return impl_->launch.photonEmissions; }
void Renderer::finishPhotonPass(std::uint32_t totalPhotons) { impl_->finishPhotonPass(totalPhotons); }
void* Renderer::accumulationBuffer(int index) const {
    if (index < 0 || index > 6)
        throw std::runtime_error("accumulation buffer index");
    return impl_->aov[index].data();
}
void Renderer::statistics(std::uint64_t out[8]) {
    for (int k = 0; k < 8; ++k)
        out[k] = 0;
    for (auto& t : impl_->tiles) {
        std::uint64_t part[8] = {};
        engineCheck(rt_wavefront_stats(t.wavefront, part, 1, t.stream), "rt_wavefront_stats");
        for (int k = 0; k < 8; ++k)
            out[k] += part[k];
    }
    if (impl_->photonWavefront) {
        std::uint64_t part[8] = {};
        engineCheck(rt_wavefront_stats(impl_->photonWavefront, part, 1, nullptr), "rt_wavefront_stats");
        for (int k = 0; k < 8; ++k)
            out[k] += part[k];
    }
}
void Renderer::setMeasurement(bool timeStages, bool countTraversal, bool gatherPerLane) {
    impl_->measurementFlags = (timeStages ? RT_WAVEFRONT_TIME_STAGES : 0U) |
                              (countTraversal ? RT_WAVEFRONT_COUNT_TRAVERSAL : 0U) |
                              (gatherPerLane ? RT_WAVEFRONT_GATHER_PER_LANE : 0U);
    for (auto& t : impl_->tiles)
        engineCheck(rt_wavefront_configure(t.wavefront, impl_->measurementFlags), "rt_wavefront_configure");
}
void Renderer::stageTimings(float milliseconds[5], std::uint32_t launches[5]) {
    for (int k = 0; k < 5; ++k) {
        milliseconds[k] = 0.0f;
        launches[k] = 0;
    }
    for (auto& t : impl_->tiles) {
        float ms[5] = {};
        std::uint32_t n[5] = {};
        engineCheck(rt_wavefront_timings(t.wavefront, ms, n, t.stream), "rt_wavefront_timings");
        for (int k = 0; k < 5; ++k) {
            milliseconds[k] += ms[k];
            launches[k] += n[k];
        }
    }
}
void Renderer::setConcurrentTiles(std::uint32_t count) { impl_->createTiles(count); }
void Renderer::setRayShare(std::uint32_t numerator, std::uint32_t denominator) {
    for (auto& t : impl_->tiles)
        engineCheck(rt_wavefront_set_ray_share(t.wavefront, numerator, denominator), "rt_wavefront_set_ray_share");
}
std::uint32_t Renderer::concurrentTiles() const { return static_cast<std::uint32_t>(impl_->tiles.size()); }
void Renderer::synchronize() const { hipCheck(hipDeviceSynchronize()); }
const rt_launch_params& Renderer::launchParams() const { return impl_->launch; }
float Renderer::causticRadius() const { return impl_->causticRadius; }

}  // namespace rt
