// C binding of the host layer (rt::Scene + rt::Renderer) for callers that cannot
// use the C++ classes directly: the Python mirror, the tests and the benchmark.
// Exceptions become a non-zero return and a message (rth_last_error()).
#include "image.hpp"
#include "renderer.hpp"

#include "rt_types.h"

#include <cstdio>
#include <cstring>
#include <exception>
#include <stdexcept>
#include <vector>
#include <string>

namespace {

thread_local std::string lastError;

template <typename F>
int guarded(F&& body) {
    try {
        body();
        return 0;
    } catch (const std::exception& e) {
        lastError = e.what();
        return 1;
    }
}

rt::Mat4 matrix(const float* m16) {
    rt::Mat4 m;
    std::memcpy(m.values.data(), m16, sizeof(float) * 16);
    return m;
}

rt::TextureRef textureRef(const rt_gpu_texture_ref& r) {
    rt::TextureRef t;
    t.texture = r.texture;
    t.texCoord = r.texCoord;
    t.offset = {r.offset.x, r.offset.y};
    t.scale = {r.scale.x, r.scale.y};
    t.rotation = r.rotation;
    t.strength = r.strength;
    return t;
}

}  // namespace

extern "C" {

const char* rth_last_error(void) { return lastError.c_str(); }

rt::Scene* rth_scene_create(void) { return new rt::Scene(); }
void rth_scene_destroy(rt::Scene* scene) { delete scene; }

// Material fields as laid out in rt_gpu_material (the scene.hpp:52-73 members).
int rth_scene_add_material(rt::Scene* scene, const rt_gpu_material* m) {
    int index = -1;
    const int status = guarded([&] {
        rt::Material out;
        out.baseColor = {m->baseColor.x, m->baseColor.y, m->baseColor.z, m->baseColor.w};
        out.emissive = {m->emissive.x, m->emissive.y, m->emissive.z};
        out.attenuationColor = {m->attenuationColor.x, m->attenuationColor.y, m->attenuationColor.z};
        out.metallic = m->metallic;
        out.roughness = m->roughness;
        out.transmission = m->transmission;
        out.ior = m->ior;
        out.thickness = m->thickness;
        out.attenuationDistance = m->attenuationDistance;
        out.emissiveStrength = m->emissiveStrength;
        out.dispersion = m->dispersion;
        out.baseColorTexture = textureRef(m->baseColorTexture);
        out.metallicRoughnessTexture = textureRef(m->metallicRoughnessTexture);
        out.normalTexture = textureRef(m->normalTexture);
        out.emissiveTexture = textureRef(m->emissiveTexture);
        out.transmissionTexture = textureRef(m->transmissionTexture);
        out.thicknessTexture = textureRef(m->thicknessTexture);
        scene->materials.push_back(out);
        index = static_cast<int>(scene->materials.size()) - 1;
    });
    return status == 0 ? index : -1;
}

int rth_scene_add_geometry(rt::Scene* scene, const rt_gpu_vertex* vertices, uint32_t vertexCount,
                           const uint32_t* indices, uint32_t triangleCount, uint32_t material) {
    int index = -1;
    const int status = guarded([&] {
        rt::Geometry g;
        g.vertices.resize(vertexCount);
        for (uint32_t i = 0; i < vertexCount; ++i) {
            const rt_gpu_vertex& v = vertices[i];
            g.vertices[i] = {{v.position.x, v.position.y, v.position.z},
                             {v.normal.x, v.normal.y, v.normal.z},
                             {v.tangent.x, v.tangent.y, v.tangent.z, v.tangent.w},
                             {v.uv.x, v.uv.y},
                             {v.uv1.x, v.uv1.y}};
        }
        g.indices.assign(indices, indices + 3 * static_cast<size_t>(triangleCount));
        g.material = material;
        scene->geometries.push_back(std::move(g));
        index = static_cast<int>(scene->geometries.size()) - 1;
    });
    return status == 0 ? index : -1;
}

int rth_scene_add_instance(rt::Scene* scene, uint32_t geometry, const float* m16) {
    int index = -1;
    const int status = guarded([&] {
        scene->instances.push_back({"", matrix(m16), geometry});
        index = static_cast<int>(scene->instances.size()) - 1;
    });
    return status == 0 ? index : -1;
}

int rth_scene_add_camera(rt::Scene* scene, const float* m16, float verticalFov, float aspectRatio, float nearPlane,
                         float aperture, float focusDistance) {
    int index = -1;
    const int status = guarded([&] {
        rt::Camera c;
        c.transform = matrix(m16);
        c.verticalFov = verticalFov;
        c.aspectRatio = aspectRatio;
        c.nearPlane = nearPlane;
        c.aperture = aperture;
        c.focusDistance = focusDistance;
        scene->cameras.push_back(c);
        index = static_cast<int>(scene->cameras.size()) - 1;
    });
    return status == 0 ? index : -1;
}

int rth_scene_add_light(rt::Scene* scene, const float* m16, const float* color3, float intensity, float range,
                        float innerCone, float outerCone, uint32_t type) {
    int index = -1;
    const int status = guarded([&] {
        rt::Light l;
        l.transform = matrix(m16);
        l.color = {color3[0], color3[1], color3[2]};
        l.intensity = intensity;
        l.range = range;
        l.innerCone = innerCone;
        l.outerCone = outerCone;
        l.type = type;
        scene->lights.push_back(l);
        index = static_cast<int>(scene->lights.size()) - 1;
    });
    return status == 0 ? index : -1;
}

int rth_scene_add_image(rt::Scene* scene, const uint8_t* rgba, uint32_t width, uint32_t height) {
    int index = -1;
    const int status = guarded([&] {
        rt::Image image;
        image.rgba.assign(rgba, rgba + static_cast<size_t>(width) * height * 4);
        image.width = width;
        image.height = height;
        scene->images.push_back(std::move(image));
        index = static_cast<int>(scene->images.size()) - 1;
    });
    return status == 0 ? index : -1;
}

int rth_scene_add_texture(rt::Scene* scene, int32_t image, uint32_t wrapU, uint32_t wrapV) {
    int index = -1;
    const int status = guarded([&] {
        rt::Texture t;
        t.image = image;
        t.wrapU = wrapU;
        t.wrapV = wrapV;
        scene->textures.push_back(t);
        index = static_cast<int>(scene->textures.size()) - 1;
    });
    return status == 0 ? index : -1;
}

void rth_scene_set_environment(rt::Scene* scene, const float* rgba, uint32_t width, uint32_t height, float rotation,
                               float strength, float exposure) {
    auto& e = scene->environment;
    e.pixels.resize(static_cast<size_t>(width) * height);
    for (size_t i = 0; i < e.pixels.size(); ++i)
        e.pixels[i] = {rgba[4 * i], rgba[4 * i + 1], rgba[4 * i + 2], rgba[4 * i + 3]};
    e.width = width;
    e.height = height;
    e.rotation = rotation;
    e.strength = strength;
    e.exposure = exposure;
}

// ---- scene files and read access (tests, tools) ----------------------------

rt::Scene* rth_scene_load(const char* path) {
    rt::Scene* scene = nullptr;
    guarded([&] { scene = new rt::Scene(rt::loadScene(path)); });
    return scene;
}

// counts8: geometries, instances, materials, images, textures, cameras, lights, environment pixels
void rth_scene_counts(const rt::Scene* scene, uint32_t* counts8) {
    counts8[0] = static_cast<uint32_t>(scene->geometries.size());
    counts8[1] = static_cast<uint32_t>(scene->instances.size());
    counts8[2] = static_cast<uint32_t>(scene->materials.size());
    counts8[3] = static_cast<uint32_t>(scene->images.size());
    counts8[4] = static_cast<uint32_t>(scene->textures.size());
    counts8[5] = static_cast<uint32_t>(scene->cameras.size());
    counts8[6] = static_cast<uint32_t>(scene->lights.size());
    counts8[7] = static_cast<uint32_t>(scene->environment.pixels.size());
}

// sizes3: vertices, indices, material.  With non-null pointers also copies the
// vertices (14 floats each: position, normal, tangent, uv, uv1) and the indices.
int rth_scene_get_geometry(const rt::Scene* scene, uint32_t index, uint32_t* sizes3, float* vertices14,
                           uint32_t* indices, char* name, uint32_t nameBytes) {
    return guarded([&] {
        const rt::Geometry& g = scene->geometries.at(index);
        sizes3[0] = static_cast<uint32_t>(g.vertices.size());
        sizes3[1] = static_cast<uint32_t>(g.indices.size());
        sizes3[2] = g.material;
        static_assert(sizeof(rt::Vertex) == 14 * sizeof(float), "Vertex is 14 packed floats");
        if (vertices14)
            std::memcpy(vertices14, g.vertices.data(), g.vertices.size() * sizeof(rt::Vertex));
        if (indices)
            std::memcpy(indices, g.indices.data(), g.indices.size() * sizeof(uint32_t));
        if (name && nameBytes)
            std::snprintf(name, nameBytes, "%s", g.name.c_str());
    });
}

int rth_scene_get_instance(const rt::Scene* scene, uint32_t index, float* m16, uint32_t* geometry) {
    return guarded([&] {
        const rt::Instance& i = scene->instances.at(index);
        std::memcpy(m16, i.transform.values.data(), sizeof(float) * 16);
        *geometry = i.geometry;
    });
}

// The material in rt_gpu_material layout (what rth_scene_add_material takes) plus alphaCutoff and flags.
int rth_scene_get_material(const rt::Scene* scene, uint32_t index, rt_gpu_material* out, float* alphaCutoff,
                           uint32_t* flags) {
    return guarded([&] {
        const rt::Material& m = scene->materials.at(index);
        std::memset(out, 0, sizeof(*out));
        auto ref = [](const rt::TextureRef& t) {
            rt_gpu_texture_ref r{};
            r.texture = t.texture;
            r.texCoord = t.texCoord;
            r.offset = {t.offset.x, t.offset.y};
            r.scale = {t.scale.x, t.scale.y};
            r.rotation = t.rotation;
            r.strength = t.strength;
            return r;
        };
        out->baseColor = {m.baseColor.x, m.baseColor.y, m.baseColor.z, m.baseColor.w};
        out->emissive = {m.emissive.x, m.emissive.y, m.emissive.z};
        out->attenuationColor = {m.attenuationColor.x, m.attenuationColor.y, m.attenuationColor.z};
        out->metallic = m.metallic;
        out->roughness = m.roughness;
        out->transmission = m.transmission;
        out->ior = m.ior;
        out->thickness = m.thickness;
        out->attenuationDistance = m.attenuationDistance;
        out->emissiveStrength = m.emissiveStrength;
        out->dispersion = m.dispersion;
        out->baseColorTexture = ref(m.baseColorTexture);
        out->metallicRoughnessTexture = ref(m.metallicRoughnessTexture);
        out->normalTexture = ref(m.normalTexture);
        out->emissiveTexture = ref(m.emissiveTexture);
        out->transmissionTexture = ref(m.transmissionTexture);
        out->thicknessTexture = ref(m.thicknessTexture);
        *alphaCutoff = m.alphaCutoff;
        *flags = m.flags;
    });
}

// values8: verticalFov, aspectRatio, nearPlane, farPlane, aperture, focusDistance, 0, 0
int rth_scene_get_camera(const rt::Scene* scene, uint32_t index, float* m16, float* values8) {
    return guarded([&] {
        const rt::Camera& c = scene->cameras.at(index);
        std::memcpy(m16, c.transform.values.data(), sizeof(float) * 16);
        const float values[8] = {c.verticalFov, c.aspectRatio, c.nearPlane, c.farPlane, c.aperture, c.focusDistance, 0, 0};
        std::memcpy(values8, values, sizeof(values));
    });
}

// values8: color rgb, intensity, range, innerCone, outerCone, type
int rth_scene_get_light(const rt::Scene* scene, uint32_t index, float* m16, float* values8) {
    return guarded([&] {
        const rt::Light& l = scene->lights.at(index);
        std::memcpy(m16, l.transform.values.data(), sizeof(float) * 16);
        const float values[8] = {l.color.x, l.color.y, l.color.z, l.intensity,
                                 l.range, l.innerCone, l.outerCone, static_cast<float>(l.type)};
        std::memcpy(values8, values, sizeof(values));
    });
}

// sizes4: width, height, mimeType, encoded bytes.  rgba (width * height * 4 bytes) is filled when not null.
int rth_scene_get_image(const rt::Scene* scene, uint32_t index, uint32_t* sizes4, uint8_t* rgba) {
    return guarded([&] {
        const rt::Image& i = scene->images.at(index);
        sizes4[0] = i.width;
        sizes4[1] = i.height;
        sizes4[2] = i.mimeType;
        sizes4[3] = static_cast<uint32_t>(i.encoded.size());
        if (rgba)
            std::memcpy(rgba, i.rgba.data(), i.rgba.size());
    });
}

// values6: image, magFilter, minFilter, wrapU, wrapV, 0
int rth_scene_get_texture(const rt::Scene* scene, uint32_t index, int32_t* values6) {
    return guarded([&] {
        const rt::Texture& t = scene->textures.at(index);
        values6[0] = t.image;
        values6[1] = static_cast<int32_t>(t.magFilter);
        values6[2] = static_cast<int32_t>(t.minFilter);
        values6[3] = static_cast<int32_t>(t.wrapU);
        values6[4] = static_cast<int32_t>(t.wrapV);
        values6[5] = 0;
    });
}

// values5: width, height, rotation, strength, exposure.  pixels (RGBA float) is filled when not null.
int rth_scene_get_environment(const rt::Scene* scene, float* values5, float* pixels) {
    return guarded([&] {
        const rt::Environment& e = scene->environment;
        values5[0] = static_cast<float>(e.width);
        values5[1] = static_cast<float>(e.height);
        values5[2] = e.rotation;
        values5[3] = e.strength;
        values5[4] = e.exposure;
        if (pixels)
            std::memcpy(pixels, e.pixels.data(), e.pixels.size() * sizeof(rt::Vec4));
    });
}

// The loader's image readers on a byte buffer: returns 0 and the size; call again with `out` to get the pixels.
int rth_decode_png(const uint8_t* bytes, size_t size, uint32_t* width, uint32_t* height, uint8_t* out) {
    return guarded([&] {
        std::vector<std::byte> file(reinterpret_cast<const std::byte*>(bytes), reinterpret_cast<const std::byte*>(bytes) + size);
        std::vector<std::uint8_t> rgba;
        if (!rt::decodePngImage(file, rgba, *width, *height))
            throw std::runtime_error("not a PNG this reader handles");
        if (out)
            std::memcpy(out, rgba.data(), rgba.size());
    });
}

int rth_decode_jpeg(const uint8_t* bytes, size_t size, uint32_t* width, uint32_t* height, uint8_t* out) {
    return guarded([&] {
        std::vector<std::byte> file(reinterpret_cast<const std::byte*>(bytes), reinterpret_cast<const std::byte*>(bytes) + size);
        std::vector<std::uint8_t> rgba;
        if (!rt::decodeJpegImage(file, rgba, *width, *height))
            throw std::runtime_error("not a JPEG this reader handles");
        if (out)
            std::memcpy(out, rgba.data(), rgba.size());
    });
}

int rth_decode_hdr(const uint8_t* bytes, size_t size, uint32_t* width, uint32_t* height, float* out) {
    return guarded([&] {
        std::vector<std::byte> file(reinterpret_cast<const std::byte*>(bytes), reinterpret_cast<const std::byte*>(bytes) + size);
        std::vector<rt::Vec4> pixels;
        if (!rt::decodeRadianceImage(file, pixels, *width, *height))
            throw std::runtime_error("not a Radiance .hdr file this reader handles");
        if (out)
            std::memcpy(out, pixels.data(), pixels.size() * sizeof(rt::Vec4));
    });
}

// The image writers of the command-line renderer (row 0 = bottom row, as the renderer hands pixels out).
int rth_write_png(const char* path, uint32_t width, uint32_t height, const uint8_t* rgba) {
    return guarded([&] { rt::writePng(path, width, height, {rgba, static_cast<size_t>(width) * height * 4}); });
}
int rth_write_exr(const char* path, uint32_t width, uint32_t height, const float* rgba) {
    return guarded([&] { rt::writeExr(path, width, height, {rgba, static_cast<size_t>(width) * height * 4}); });
}

rt::Renderer* rth_renderer_create(const rt::Scene* scene, uint32_t width, uint32_t height, int finalProfile) {
    rt::Renderer* renderer = nullptr;
    guarded([&] {
        renderer = new rt::Renderer(*scene, width, height, finalProfile ? rt::Profile::Final : rt::Profile::Preview);
    });
    return renderer;
}

void rth_renderer_destroy(rt::Renderer* r) { delete r; }
int rth_renderer_render(rt::Renderer* r, void* output) { return guarded([&] { r->render(output); }); }
int rth_renderer_denoise(rt::Renderer* r) { return guarded([&] { r->denoise(); }); }
int rth_renderer_set_camera(rt::Renderer* r, const float* m16, float verticalFov, float aspectRatio, float nearPlane,
                            float aperture, float focusDistance) {
    rt::Camera c;
    c.transform = matrix(m16);
    c.verticalFov = verticalFov;
    c.aspectRatio = aspectRatio;
    c.nearPlane = nearPlane;
    c.aperture = aperture;
    c.focusDistance = focusDistance;
    return guarded([&] { r->setCamera(c); });
}
int rth_renderer_set_profile(rt::Renderer* r, int finalProfile) {
    return guarded([&] { r->setProfile(finalProfile ? rt::Profile::Final : rt::Profile::Preview); });
}
int rth_renderer_copy_output(rt::Renderer* r, void* deviceOutput) { return guarded([&] { r->copyOutput(deviceOutput); }); }
int rth_renderer_pixels(rt::Renderer* r, uint8_t* out) {
    return guarded([&] {
        const auto p = r->pixels();
        std::memcpy(out, p.data(), p.size());
    });
}
int rth_renderer_linear_pixels(rt::Renderer* r, float* out) {
    return guarded([&] {
        const auto p = r->linearPixels();
        std::memcpy(out, p.data(), p.size() * sizeof(float));
    });
}
uint32_t rth_renderer_samples(rt::Renderer* r) { return r->samples(); }

int rth_renderer_set_photon_budget(rt::Renderer* r, uint32_t emissions, uint32_t capacity, uint32_t buckets) {
    return guarded([&] { r->setPhotonBudget(emissions, capacity, buckets); });
}
int rth_renderer_set_sample_sequence(rt::Renderer* r, uint32_t first, uint32_t stride) {
    return guarded([&] { r->setSampleSequence(first, stride); });
}
int rth_renderer_set_pixel_range(rt::Renderer* r, uint32_t begin, uint32_t end) {
    return guarded([&] { r->setPixelRange(begin, end); });
}
int rth_renderer_set_row_interleave(rt::Renderer* r, uint32_t first, uint32_t stride) {
    return guarded([&] { r->setRowInterleave(first, stride); });
}
int rth_renderer_set_concurrent_tiles(rt::Renderer* r, uint32_t count) {
    return guarded([&] { r->setConcurrentTiles(count); });
}
int rth_renderer_set_parts_and_depth(rt::Renderer* r, uint32_t count, uint32_t samplesInFlight) {
    return guarded([&] { r->setConcurrentTiles(count, samplesInFlight); });
}
uint32_t rth_renderer_samples_in_flight(rt::Renderer* r) { return r->samplesInFlight(); }
int rth_renderer_set_sample_batch(rt::Renderer* r, uint32_t samples) {
    return guarded([&] { r->setSampleBatch(samples); });
}
uint32_t rth_renderer_sample_batch(rt::Renderer* r) { return r->sampleBatch(); }
uint32_t rth_renderer_concurrent_tiles(rt::Renderer* r) { return r->concurrentTiles(); }
int rth_renderer_set_ray_share(rt::Renderer* r, uint32_t numerator, uint32_t denominator) {
    return guarded([&] { r->setRayShare(numerator, denominator); });
}
void rth_renderer_set_manual_photon_passes(rt::Renderer* r, int manual) { r->setManualPhotonPasses(manual != 0); }
int rth_renderer_photon_pass_due(rt::Renderer* r) { return r->photonPassDue() ? 1 : 0; }
int rth_renderer_begin_photon_pass(rt::Renderer* r) { return guarded([&] { r->beginPhotonPass(); }); }
int rth_renderer_trace_photon_range(rt::Renderer* r, uint32_t first, uint32_t count, int intoStaging, uint32_t* stored) {
    return guarded([&] { *stored = r->tracePhotonRange(first, count, intoStaging != 0); });
}
void* rth_renderer_photon_buffer(rt::Renderer* r) { return r->photonBuffer(); }
void* rth_renderer_photon_staging_buffer(rt::Renderer* r) { return r->photonStagingBuffer(); }
// several consecutive photon passes at once (renderer.hpp: beginPhotonPasses and its companions)
uint32_t rth_renderer_photon_passes_together(rt::Renderer* r) {
    uint32_t count = 0;
    guarded([&] { count = r->photonPassesTogether(); });
    return count;
}
int rth_renderer_begin_photon_passes(rt::Renderer* r, uint32_t wanted, uint32_t* begun) {
    return guarded([&] { *begun = r->beginPhotonPasses(wanted); });
}
int rth_renderer_trace_photon_range_passes(rt::Renderer* r, uint32_t first, uint32_t count, int intoStaging, uint32_t* stored) {
    return guarded([&] { r->tracePhotonRangePasses(first, count, intoStaging != 0, stored); });
}
int rth_renderer_finish_photon_passes(rt::Renderer* r, const uint32_t* totals) {
    return guarded([&] { r->finishPhotonPasses(totals); });
}
void* rth_renderer_photon_buffer_of(rt::Renderer* r, uint32_t pass) {
    void* buffer = nullptr;
    guarded([&] { buffer = r->photonBuffer(pass); });
    return buffer;
}
void* rth_renderer_photon_staging_buffer_of(rt::Renderer* r, uint32_t pass) {
    void* buffer = nullptr;
    guarded([&] { buffer = r->photonStagingBuffer(pass); });
    return buffer;
}
void* rth_renderer_output_buffer(rt::Renderer* r) { return r->outputBuffer(); }
uint32_t rth_renderer_photon_capacity(rt::Renderer* r) { return r->photonCapacity(); }
uint32_t rth_renderer_photon_emissions(rt::Renderer* r) { return r->photonEmissions(); }
int rth_renderer_finish_photon_pass(rt::Renderer* r, uint32_t total) {
    return guarded([&] { r->finishPhotonPass(total); });
}
void* rth_renderer_accumulation_buffer(rt::Renderer* r, int index) {
    void* p = nullptr;
    guarded([&] { p = r->accumulationBuffer(index); });
    return p;
}
int rth_renderer_statistics(rt::Renderer* r, uint64_t* out8) { return guarded([&] { r->statistics(out8); }); }
int rth_renderer_set_measurement(rt::Renderer* r, int timeStages, int countTraversal, int gatherPerLane) {
    return guarded([&] { r->setMeasurement(timeStages != 0, countTraversal != 0, gatherPerLane != 0); });
}
int rth_renderer_stage_timings(rt::Renderer* r, float* milliseconds5, uint32_t* launches5) {
    return guarded([&] { r->stageTimings(milliseconds5, launches5); });
}
int rth_renderer_synchronize(rt::Renderer* r) { return guarded([&] { r->synchronize(); }); }
void rth_renderer_launch_params(rt::Renderer* r, rt_launch_params* out) { *out = r->launchParams(); }
float rth_renderer_caustic_radius(rt::Renderer* r) { return r->causticRadius(); }

}  // extern "C"
