// Host scene description consumed by rt::Renderer.
//
// Same structs, member names, defaults and conventions as the reference's
// src/scene.hpp:13-145 (column-major Mat4, 56-byte Vertex, uint32 indices, one
// material per geometry, instance = geometry + transform), so code that fills a
// reference rt::Scene fills this one unchanged.  One addition: an Image may
// carry already decoded RGBA8 pixels.
#pragma once

#include <array>
#include <cstddef>
#include <cstdint>
#include <filesystem>
#include <limits>
#include <string>
#include <vector>

namespace rt {

struct Vec2 { float x, y; };
struct Vec3 { float x, y, z; };
struct Vec4 { float x, y, z, w; };

struct Mat4 {
    std::array<float, 16> values;  // column-major
};

struct Vertex {
    Vec3 position;
    Vec3 normal;
    Vec4 tangent;
    Vec2 uv;
    Vec2 uv1;
};

struct TextureRef {
    std::int32_t texture = -1;
    std::uint32_t texCoord = 0;
    Vec2 offset{0.0f, 0.0f};
    Vec2 scale{1.0f, 1.0f};
    float rotation = 0.0f;
    float strength = 1.0f;
};

struct Material {
    std::string name;
    Vec4 baseColor{1.0f, 1.0f, 1.0f, 1.0f};
    Vec3 emissive{0.0f, 0.0f, 0.0f};
    Vec3 attenuationColor{1.0f, 1.0f, 1.0f};
    float metallic = 1.0f;
    float roughness = 1.0f;
    float transmission = 0.0f;
    float ior = 1.5f;
    float thickness = 0.0f;
    float attenuationDistance = std::numeric_limits<float>::infinity();
    float emissiveStrength = 1.0f;
    float dispersion = 0.0f;
    float alphaCutoff = 0.5f;
    std::uint32_t flags = 0;
    TextureRef baseColorTexture;
    TextureRef metallicRoughnessTexture;
    TextureRef normalTexture;
    TextureRef emissiveTexture;
    TextureRef transmissionTexture;
    TextureRef thicknessTexture;
};

struct Geometry {
    std::string name;
    std::vector<Vertex> vertices;
    std::vector<std::uint32_t> indices;
    std::uint32_t material = 0;
};

struct Instance {
    std::string name;
    Mat4 transform;
    std::uint32_t geometry;
};

struct Image {
    std::string name;
    std::vector<std::byte> encoded;
    std::uint32_t mimeType = 0;
    // Decoded pixels (RGBA8, row 0 first).  When set they are uploaded as they
    // are and `encoded` is not looked at.
    std::vector<std::uint8_t> rgba;
    std::uint32_t width = 0;
    std::uint32_t height = 0;
};

struct Texture {
    std::string name;
    std::int32_t image = -1;
    std::uint32_t magFilter = 0;
    std::uint32_t minFilter = 0;
    std::uint32_t wrapU = 10497;
    std::uint32_t wrapV = 10497;
};

struct Camera {
    std::string name;
    Mat4 transform;
    float verticalFov = 0.7853982f;
    float aspectRatio = 0.0f;
    float nearPlane = 0.01f;
    float farPlane = std::numeric_limits<float>::infinity();
    float aperture = 0.0f;
    float focusDistance = 10.0f;
};

struct Light {
    std::string name;
    Mat4 transform;
    Vec3 color;
    float intensity;
    float range;
    float innerCone;
    float outerCone;
    std::uint32_t type;  // 0 directional, 1 spot, 2 point
};

struct Environment {
    std::vector<Vec4> pixels;
    std::uint32_t width = 0;
    std::uint32_t height = 0;
    float rotation = 0.0f;
    float strength = 1.0f;
    float exposure = 0.0f;
};

struct Scene {
    std::vector<Geometry> geometries;
    std::vector<Instance> instances;
    std::vector<Material> materials;
    std::vector<Image> images;
    std::vector<Texture> textures;
    std::vector<Camera> cameras;
    std::vector<Light> lights;
    Environment environment;
};

// glTF 2.0 (.gltf with external or embedded buffers, or .glb), as the
// reference's loadScene (src/scene.hpp:145, src/scene.cpp:332-406): TRIANGLES
// primitives, KHR_lights_punctual, the KHR_materials_{ior, transmission, volume,
// dispersion, emissive_strength, unlit} and KHR_texture_transform extensions,
// KHR_mesh_quantization attributes, the `raytracer_*` extras.  PNG images come
// back decoded (Image::rgba); other formats keep their encoded bytes.  The
// environment named by `raytracer_hdri` must be a Radiance .hdr file.
Scene loadScene(const std::filesystem::path& path);

// The image readers the loader uses, exposed for tests and tools.
// PNG -> RGBA8, row 0 = top row; false when the file is not a PNG this reader handles.
bool decodePngImage(const std::vector<std::byte>& file, std::vector<std::uint8_t>& rgba, std::uint32_t& width,
                    std::uint32_t& height);
// JPEG (baseline, extended sequential or progressive Huffman, grey or three components) -> RGBA8,
// row 0 = top row, alpha 255 (jpeg.cpp).
bool decodeJpegImage(const std::vector<std::byte>& file, std::vector<std::uint8_t>& rgba, std::uint32_t& width,
                     std::uint32_t& height);
// Radiance RGBE -> float RGBA (alpha 1), rows in file order.
bool decodeRadianceImage(const std::vector<std::byte>& file, std::vector<Vec4>& pixels, std::uint32_t& width,
                         std::uint32_t& height);

}  // namespace rt
