// CPU oracle: the camera path integrator and the photon tracer, restated from the
// reference's device programs (src/device.cu) as scalar IEEE code on top of the
// oracle's own ray queries.  Test infrastructure only (see oracle.h).
//
// PARITY UNPINNED: the reference has no CPU path, tests or golden images, and its
// device code is compiled with --use_fast_math (CMakeLists.txt:53), which this
// restatement deliberately does not imitate.  Random-number draw order, branch
// structure and every constant follow the cited lines.
#include "oracle.h"
#include "oracle_internal.h"

#include "rt_math.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <thread>
#include <vector>

namespace {

// The elementary functions come from rt_math.h (sequences of correctly rounded binary64 operations, the same
// bits on every IEEE machine), which the device code uses as well; -DRT_LIBM_MATH takes the host libm's.
#ifdef RT_LIBM_MATH
float mathSin(float x) { return std::sin(x); }
float mathCos(float x) { return std::cos(x); }
float mathPow(float x, float y) { return std::pow(x, y); }
float mathAtan2(float y, float x) { return std::atan2(y, x); }
float mathAcos(float x) { return std::acos(x); }
float mathExp2(float x) { return std::exp2(x); }
#else
float mathSin(float x) { return rtm::sine(x); }
float mathCos(float x) { return rtm::cosine(x); }
float mathPow(float x, float y) { return rtm::power(x, y); }
float mathAtan2(float y, float x) { return rtm::arcTangent2(y, x); }
float mathAcos(float x) { return rtm::arcCosine(x); }
float mathExp2(float x) { return rtm::exp2(x); }
#endif

struct F3 {
    float x, y, z;
};

// device.cu:12-75
F3 operator+(F3 a, F3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
F3 operator-(F3 a, F3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
F3 operator-(F3 a) { return {-a.x, -a.y, -a.z}; }
F3 operator*(F3 a, F3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
F3 operator*(F3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
F3 operator*(float s, F3 a) { return a * s; }
F3 operator/(F3 a, float s) { return a * (1.0f / s); }  // device.cu:40-42
F3 operator/(F3 a, F3 b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }
F3 operator+(F3 a, float s) { return {a.x + s, a.y + s, a.z + s}; }
void operator+=(F3& a, F3 b) { a = a + b; }
void operator*=(F3& a, F3 b) { a = a * b; }
void operator*=(F3& a, float s) { a = a * s; }
void operator/=(F3& a, float s) { a = a / s; }
float dot(F3 a, F3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
F3 cross(F3 a, F3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
F3 normalize(F3 v) { return v / std::sqrt(dot(v, v)); }
F3 f3(const rt_float3& v) { return {v.x, v.y, v.z}; }
F3 rgb(const rt_float4& v) { return {v.x, v.y, v.z}; }
rt_float3 r3(F3 v) { return {v.x, v.y, v.z}; }

struct Ctx {
    const orc_scene* scene;
    const rt_launch_params* p;
};

float rnd(uint32_t& state) { return orc_random(&state); }

// device.cu:81-84
float linear(float value) {
    return value <= 0.04045f ? value / 12.92f : mathPow((value + 0.055f) / 1.055f, 2.4f);
}

// device.cu:86-98
int wrapped(int value, unsigned size, unsigned mode) {
    if (mode == 33071U)
        return std::min(std::max(value, 0), static_cast<int>(size) - 1);
    if (mode == 33648U) {
        const int period = static_cast<int>(size) * 2;
        const int coordinate = (value % period + period) % period;
        return coordinate < static_cast<int>(size) ? coordinate : period - coordinate - 1;
    }
    return (value % static_cast<int>(size) + static_cast<int>(size)) % static_cast<int>(size);
}

struct F4 {
    float x, y, z, w;
};

// device.cu:100-107
F4 textureTexel(const rt_gpu_texture& texture, int x, int y) {
    const rt_uchar4 value = texture.pixels[wrapped(y, texture.height, texture.wrapV) * texture.width +
                                           wrapped(x, texture.width, texture.wrapU)];
    return {value.x / 255.0f, value.y / 255.0f, value.z / 255.0f, value.w / 255.0f};
}

// device.cu:109-147
F4 texture(const Ctx& c, const rt_gpu_texture_ref& reference, const rt_hit& hit, bool srgb) {
    if (reference.texture < 0)
        return {1.0f, 1.0f, 1.0f, 1.0f};
    const rt_gpu_texture source = c.p->textures[reference.texture];
    const rt_float2 input = reference.texCoord == 1U ? hit.uv1 : hit.uv;
    const float sx = input.x * reference.scale.x, sy = input.y * reference.scale.y;
    const float cosine = mathCos(reference.rotation);
    const float sine = mathSin(reference.rotation);
    const float u = reference.offset.x + cosine * sx - sine * sy;
    const float v = reference.offset.y + sine * sx + cosine * sy;
    const float x = u * source.width - 0.5f;
    const float y = v * source.height - 0.5f;
    const int x0 = static_cast<int>(std::floor(x));
    const int y0 = static_cast<int>(std::floor(y));
    const float tx = x - std::floor(x);
    const float ty = y - std::floor(y);
    const F4 a = textureTexel(source, x0, y0);
    const F4 b = textureTexel(source, x0 + 1, y0);
    const F4 cc = textureTexel(source, x0, y0 + 1);
    const F4 d = textureTexel(source, x0 + 1, y0 + 1);
    F4 value{(a.x * (1.0f - tx) + b.x * tx) * (1.0f - ty) + (cc.x * (1.0f - tx) + d.x * tx) * ty,
             (a.y * (1.0f - tx) + b.y * tx) * (1.0f - ty) + (cc.y * (1.0f - tx) + d.y * tx) * ty,
             (a.z * (1.0f - tx) + b.z * tx) * (1.0f - ty) + (cc.z * (1.0f - tx) + d.z * tx) * ty,
             (a.w * (1.0f - tx) + b.w * tx) * (1.0f - ty) + (cc.w * (1.0f - tx) + d.w * tx) * ty};
    if (srgb) {
        value.x = linear(value.x);
        value.y = linear(value.y);
        value.z = linear(value.z);
    }
    return value;
}

// device.cu:149-166
rt_gpu_material textured(const Ctx& c, rt_gpu_material material, const rt_hit& hit) {
    const F4 base = texture(c, material.baseColorTexture, hit, true);
    material.baseColor.x *= base.x;
    material.baseColor.y *= base.y;
    material.baseColor.z *= base.z;
    material.baseColor.w *= base.w;
    const F4 pbr = texture(c, material.metallicRoughnessTexture, hit, false);
    material.roughness *= pbr.y;
    material.metallic *= pbr.z;
    const F4 emissive = texture(c, material.emissiveTexture, hit, true);
    material.emissive.x *= emissive.x;
    material.emissive.y *= emissive.y;
    material.emissive.z *= emissive.z;
    material.transmission *= texture(c, material.transmissionTexture, hit, false).x;
    material.thickness *= texture(c, material.thicknessTexture, hit, false).y;
    return material;
}

// device.cu:168-182
F3 mappedNormal(const Ctx& c, const rt_hit& hit, const rt_gpu_material& material) {
    if (material.normalTexture.texture < 0)
        return f3(hit.normal);
    const F4 sample = texture(c, material.normalTexture, hit, false);
    const F3 tangent = normalize(F3{hit.tangent.x, hit.tangent.y, hit.tangent.z});
    const F3 local = normalize(F3{(sample.x * 2.0f - 1.0f) * material.normalTexture.strength,
                                  (sample.y * 2.0f - 1.0f) * material.normalTexture.strength,
                                  sample.z * 2.0f - 1.0f});
    return normalize(tangent * local.x + cross(f3(hit.normal), tangent) * (local.y * hit.tangent.w) +
                     f3(hit.normal) * local.z);
}

float maximum(F3 v) { return std::fmax(std::fmax(v.x, v.y), v.z); }               // device.cu:184-186
float saturate(float v) { return std::fmin(std::fmax(v, 0.0f), 1.0f); }             // device.cu:188-190
F3 reflect(F3 direction, F3 normal) { return direction - 2.0f * dot(direction, normal) * normal; }  // :192-194

// device.cu:196-205
bool refract(F3 direction, F3 normal, float eta, F3& result) {
    const float cosine = std::fmin(dot(-direction, normal), 1.0f);
    const F3 perpendicular = eta * (direction + cosine * normal);
    const float parallelSquared = 1.0f - dot(perpendicular, perpendicular);
    if (parallelSquared < 0.0f)
        return false;
    result = perpendicular - std::sqrt(parallelSquared) * normal;
    return true;
}

F3 sky(F3) { return {0.1f, 0.1f, 0.1f}; }  // device.cu:235-237

// device.cu:239-245
void environmentUv(const Ctx& c, F3 direction, float& u, float& v) {
    u = mathAtan2(direction.z, direction.x) * 0.15915494f + 0.5f + c.p->environmentRotation * 0.15915494f;
    u -= std::floor(u);
    v = mathAcos(std::fmin(std::fmax(direction.y, -1.0f), 1.0f)) * 0.31830989f;
}

// device.cu:247-253
F3 environmentTexel(const Ctx& c, int x, int y) {
    const int w = static_cast<int>(c.p->environmentWidth);
    x = (x % w + w) % w;
    y = std::min(std::max(y, 0), static_cast<int>(c.p->environmentHeight) - 1);
    return rgb(c.p->environment[y * c.p->environmentWidth + x]);
}

// device.cu:255-271
F3 environment(const Ctx& c, F3 direction) {
    if (!c.p->environment)
        return sky(direction);
    float u, v;
    environmentUv(c, direction, u, v);
    const float x = u * c.p->environmentWidth - 0.5f;
    const float y = v * c.p->environmentHeight - 0.5f;
    const int x0 = static_cast<int>(std::floor(x));
    const int y0 = static_cast<int>(std::floor(y));
    const float tx = x - std::floor(x);
    const float ty = y - std::floor(y);
    const F3 a = environmentTexel(c, x0, y0) * (1.0f - tx) + environmentTexel(c, x0 + 1, y0) * tx;
    const F3 b = environmentTexel(c, x0, y0 + 1) * (1.0f - tx) + environmentTexel(c, x0 + 1, y0 + 1) * tx;
    return (a * (1.0f - ty) + b * ty) * c.p->environmentStrength;
}

F3 basisTangent(F3 normal) {  // device.cu:278-280, 295-297, 634-636, 645-647
    return normalize(std::fabs(normal.x) > 0.5f ? cross(F3{0.0f, 1.0f, 0.0f}, normal)
                                                : cross(F3{1.0f, 0.0f, 0.0f}, normal));
}

// device.cu:273-284 (phi first, then the radius)
F3 cosineDirection(F3 normal, uint32_t& rng) {
    const float phi = 6.2831853f * rnd(rng);
    const float radius = std::sqrt(rnd(rng));
    const float z = std::sqrt(std::fmax(0.0f, 1.0f - radius * radius));
    const F3 tangent = basisTangent(normal);
    const F3 bitangent = cross(normal, tangent);
    return normalize(tangent * (radius * mathCos(phi)) + bitangent * (radius * mathSin(phi)) + normal * z);
}

float roughnessAlpha(float roughness) { return std::fmax(roughness * roughness, 1e-4f); }  // device.cu:286-288

// device.cu:293-323 (radius first, then phi)
F3 ggxNormal(F3 view, F3 normal, float alpha, uint32_t& rng) {
    const F3 tangent = basisTangent(normal);
    const F3 bitangent = cross(normal, tangent);
    const F3 local{dot(view, tangent), dot(view, bitangent), dot(view, normal)};
    const F3 stretched = normalize(F3{alpha * local.x, alpha * local.y, local.z});
    const float lengthSquared = stretched.x * stretched.x + stretched.y * stretched.y;
    const F3 basisX = lengthSquared > 0.0f ? F3{-stretched.y, stretched.x, 0.0f} / std::sqrt(lengthSquared)
                                           : F3{1.0f, 0.0f, 0.0f};
    const F3 basisY = cross(stretched, basisX);
    const float radius = std::sqrt(rnd(rng));
    const float phi = 6.2831853f * rnd(rng);
    const float x = radius * mathCos(phi);
    float y = radius * mathSin(phi);
    const float lerp = 0.5f * (1.0f + stretched.z);
    y = (1.0f - lerp) * std::sqrt(std::fmax(0.0f, 1.0f - x * x)) + lerp * y;
    const F3 hemisphere = basisX * x + basisY * y + stretched * std::sqrt(std::fmax(0.0f, 1.0f - x * x - y * y));
    const F3 micro = normalize(F3{alpha * hemisphere.x, alpha * hemisphere.y, std::fmax(0.0f, hemisphere.z)});
    return normalize(tangent * micro.x + bitangent * micro.y + normal * micro.z);
}

// device.cu:325-329
float smithLambda(float cosine, float alpha) {
    const float squared = cosine * cosine;
    const float tangentSquared = std::fmax(1.0f - squared, 0.0f) / std::fmax(squared, 1e-8f);
    return 0.5f * (std::sqrt(1.0f + alpha * alpha * tangentSquared) - 1.0f);
}

// device.cu:333-337
float maskingRatio(float viewCosine, float lightCosine, float alpha) {
    const float view = smithLambda(viewCosine, alpha);
    return (1.0f + view) / (1.0f + view + smithLambda(lightCosine, alpha));
}

// device.cu:339-343
float fresnel(float cosine, float ior) {
    const float r = (1.0f - ior) / (1.0f + ior);
    const float r2 = r * r;
    return r2 + (1.0f - r2) * mathPow(std::fmax(1.0f - cosine, 0.0f), 5.0f);
}

// device.cu:345-350
F3 schlick(F3 f0, float cosine) {
    const float scale = mathPow(std::fmax(1.0f - cosine, 0.0f), 5.0f);
    return {f0.x + (1.0f - f0.x) * scale, f0.y + (1.0f - f0.y) * scale, f0.z + (1.0f - f0.z) * scale};
}

// device.cu:352-358
F3 baseReflectance(const rt_gpu_material& m) {
    const F3 color = rgb(m.baseColor);
    return {0.04f + (color.x - 0.04f) * m.metallic, 0.04f + (color.y - 0.04f) * m.metallic,
            0.04f + (color.z - 0.04f) * m.metallic};
}

// device.cu:360-370
F3 absorption(const rt_gpu_material& m, float distance) {
    if (!std::isfinite(m.attenuationDistance) || m.attenuationDistance <= 0.0f)
        return {1.0f, 1.0f, 1.0f};
    const float scale = distance / m.attenuationDistance;
    return {mathPow(std::fmax(m.attenuationColor.x, 1e-4f), scale),
            mathPow(std::fmax(m.attenuationColor.y, 1e-4f), scale),
            mathPow(std::fmax(m.attenuationColor.z, 1e-4f), scale)};
}

// device.cu:372-377
F3 lensSample(const Ctx& c, uint32_t& rng) {
    const float radius = std::sqrt(rnd(rng));
    const float angle = 6.2831853f * rnd(rng);
    return f3(c.p->lensU) * (radius * mathCos(angle) * c.p->aperture) +
           f3(c.p->lensV) * (radius * mathSin(angle) * c.p->aperture);
}

// device.cu:379-386
F3 aces(F3 color) {
    color = color * (2.51f * color + 0.03f) / (color * (2.43f * color + 0.59f) + 0.14f);
    return {mathPow(std::fmin(std::fmax(color.x, 0.0f), 1.0f), 1.0f / 2.2f),
            mathPow(std::fmin(std::fmax(color.y, 0.0f), 1.0f), 1.0f / 2.2f),
            mathPow(std::fmin(std::fmax(color.z, 0.0f), 1.0f), 1.0f / 2.2f)};
}

// device.cu:388-397
rt_hit trace(const Ctx& c, F3 origin, F3 direction, float maximumDistance = 1e16f) {
    const float o[3] = {origin.x, origin.y, origin.z};
    const float d[3] = {direction.x, direction.y, direction.z};
    return orc::traceHit(c.scene, o, d, maximumDistance);
}

// device.cu:399-403
float powerHeuristic(float a, float b) {
    const float a2 = a * a;
    const float b2 = b * b;
    return a2 / std::fmax(a2 + b2, 1e-12f);
}

// device.cu:408-413
float specularProbability(const rt_gpu_material& m, F3 view, F3 normal) {
    const F3 reflectance = schlick(baseReflectance(m), std::fmax(dot(view, normal), 0.0f));
    return std::fmin(std::fmax(maximum(reflectance), 0.05f), 0.95f);
}

struct LightSample {  // device.cu:415-425
    F3 direction{0, 0, 0};
    F3 radiance{0, 0, 0};
    float distance = 0.0f;
    float pdf = 0.0f;
    uint32_t instance = 0;
    uint32_t primitive = 0;
    bool delta = false;
    bool environment = false;
    bool valid = false;
};

struct EnvironmentSample {  // device.cu:427-431
    F3 direction;
    F3 radiance;
    float pdf;
};

// device.cu:433-447
const rt_gpu_light* selectLight(const Ctx& c, uint32_t& rng, float& probability) {
    if (c.p->lightCount == 0U || c.p->lightWeight <= 0.0f)
        return nullptr;
    float target = rnd(rng) * c.p->lightWeight;
    const rt_gpu_light* selected = &c.p->lights[c.p->lightCount - 1U];
    for (unsigned index = 0; index < c.p->lightCount; ++index) {
        selected = &c.p->lights[index];
        target -= selected->weight;
        if (target <= 0.0f)
            break;
    }
    probability = selected->weight / c.p->lightWeight;
    return selected;
}

// device.cu:449-478
EnvironmentSample sampleEnvironment(const Ctx& c, uint32_t& rng) {
    const unsigned count = c.p->environmentWidth * c.p->environmentHeight;
    const float target = rnd(rng);
    unsigned first = 0;
    unsigned last = count - 1U;
    while (first < last) {
        const unsigned middle = (first + last) / 2U;
        if (c.p->environmentCdf[middle] < target)
            first = middle + 1U;
        else
            last = middle;
    }
    const unsigned index = first;
    const float probability = c.p->environmentCdf[index] - (index ? c.p->environmentCdf[index - 1U] : 0.0f);
    const float u = (static_cast<float>(index % c.p->environmentWidth) + rnd(rng)) / c.p->environmentWidth;
    const float v = (static_cast<float>(index / c.p->environmentWidth) + rnd(rng)) / c.p->environmentHeight;
    const float theta = 3.14159265f * v;
    const float phi = 6.2831853f * (u - 0.5f) - c.p->environmentRotation;
    const float sine = mathSin(theta);
    const F3 direction{mathCos(phi) * sine, mathCos(theta), mathSin(phi) * sine};
    const float solidAngle = 19.7392088f * std::fmax(sine, 1e-6f) / static_cast<float>(count);
    return {direction, environment(c, direction), probability / solidAngle};
}

// device.cu:480-546
LightSample sampleLight(const Ctx& c, F3 position, uint32_t& rng) {
    LightSample sample;
    float choice = 0.0f;
    const rt_gpu_light* selected = selectLight(c, rng, choice);
    if (!selected)
        return sample;
    if (selected->type == 4U) {
        const EnvironmentSample source = sampleEnvironment(c, rng);
        sample.direction = source.direction;
        sample.radiance = source.radiance;
        sample.distance = 1e16f;
        sample.pdf = choice * source.pdf;
        sample.environment = true;
        sample.valid = true;
        return sample;
    }
    if (selected->type == 3U) {
        const float root = std::sqrt(rnd(rng));
        const float u = 1.0f - root;
        const float v = rnd(rng) * root;
        const F3 point = f3(selected->a) * u + f3(selected->b) * v + f3(selected->c) * (1.0f - u - v);
        const F3 offset = point - position;
        const float distanceSquared = dot(offset, offset);
        sample.distance = std::sqrt(distanceSquared);
        sample.direction = offset / sample.distance;
        const float cosine = std::fabs(dot(f3(selected->normal), -sample.direction));
        if (cosine <= 1e-6f)
            return sample;
        sample.pdf = choice * distanceSquared / (selected->area * cosine);
        sample.radiance = f3(selected->emission);
        sample.instance = selected->instance;
        sample.primitive = selected->primitive;
        sample.valid = true;
        return sample;
    }
    sample.delta = true;
    sample.pdf = choice;
    sample.radiance = f3(selected->emission);
    if (selected->type == 0U) {
        sample.direction = -f3(selected->b);
        sample.distance = 1e16f;
    } else {
        const F3 offset = f3(selected->a) - position;
        const float distanceSquared = dot(offset, offset);
        sample.distance = std::sqrt(distanceSquared);
        if (sample.distance >= selected->range)
            return sample;
        sample.direction = offset / sample.distance;
        sample.radiance /= distanceSquared;
        if (selected->type == 1U) {
            const float cone = dot(f3(selected->b), -sample.direction);
            const float outer = mathCos(selected->outerCone);
            const float inner = mathCos(selected->innerCone);
            const float falloff = saturate((cone - outer) / std::fmax(inner - outer, 1e-5f));
            sample.radiance = sample.radiance * (falloff * falloff);
            if (falloff <= 0.0f)
                return sample;
        }
    }
    sample.valid = true;
    return sample;
}

// device.cu:548-559
bool visible(const Ctx& c, F3 position, F3 normal, const LightSample& light) {
    const float limit = light.delta ? light.distance - 0.002f : light.distance + 0.002f;
    const rt_hit blocker = trace(c, position + normal * 0.001f, light.direction, limit);
    if (light.environment)
        return !blocker.found;
    if (light.delta)
        return !blocker.found;
    return blocker.found && blocker.instance == light.instance && blocker.primitive == light.primitive;
}

// device.cu:561-578
float environmentPdf(const Ctx& c, F3 direction) {
    if (!c.p->environment || c.p->lightWeight <= 0.0f)
        return 0.0f;
    float u, v;
    environmentUv(c, direction, u, v);
    const unsigned x = std::min(static_cast<unsigned>(u * c.p->environmentWidth), c.p->environmentWidth - 1U);
    const unsigned y = std::min(static_cast<unsigned>(v * c.p->environmentHeight), c.p->environmentHeight - 1U);
    const unsigned index = y * c.p->environmentWidth + x;
    const float probability = c.p->environmentCdf[index] - (index ? c.p->environmentCdf[index - 1U] : 0.0f);
    const float theta = 3.14159265f * (static_cast<float>(y) + 0.5f) / c.p->environmentHeight;
    const float solidAngle = 19.7392088f * std::fmax(mathSin(theta), 1e-6f) /
                             static_cast<float>(c.p->environmentWidth * c.p->environmentHeight);
    return c.p->environmentWeight / c.p->lightWeight * probability / solidAngle;
}

// device.cu:580-605
F3 directLighting(const Ctx& c, const rt_hit& hit, const rt_gpu_material& m, F3 view, uint32_t& rng) {
    const float diffuseWeight = (1.0f - m.metallic) * (1.0f - m.transmission);
    if (diffuseWeight <= 0.0f)
        return {0.0f, 0.0f, 0.0f};
    const LightSample light = sampleLight(c, f3(hit.position), rng);
    const float cosine = dot(f3(hit.normal), light.direction);
    if (!light.valid || cosine <= 0.0f || !visible(c, f3(hit.position), f3(hit.normal), light))
        return {0.0f, 0.0f, 0.0f};
    const F3 reflectance = schlick(baseReflectance(m), std::fmax(dot(view, f3(hit.normal)), 0.0f));
    const F3 diffuse{1.0f - reflectance.x, 1.0f - reflectance.y, 1.0f - reflectance.z};
    const F3 bsdf = rgb(m.baseColor) * diffuse * (diffuseWeight / 3.14159265f);
    const float bsdfPdf =
        (1.0f - m.transmission) * (1.0f - specularProbability(m, view, f3(hit.normal))) * cosine / 3.14159265f;
    const float weight = light.delta ? 1.0f : powerHeuristic(light.pdf, bsdfPdf);
    return bsdf * light.radiance * (cosine * weight / light.pdf);
}

// device.cu:607-621
float emissivePdf(const Ctx& c, F3 origin, const rt_hit& hit) {
    const F3 offset = f3(hit.position) - origin;
    const float distanceSquared = dot(offset, offset);
    const F3 direction = offset / std::sqrt(distanceSquared);
    for (unsigned index = 0; index < c.p->lightCount; ++index) {
        const rt_gpu_light& light = c.p->lights[index];
        if (light.type != 3U || light.instance != hit.instance || light.primitive != hit.primitive)
            continue;
        const float cosine = std::fabs(dot(f3(light.normal), -direction));
        return light.weight / c.p->lightWeight * distanceSquared / std::fmax(light.area * cosine, 1e-8f);
    }
    return 0.0f;
}

struct PhotonEmission {  // device.cu:623-628
    F3 origin{0, 0, 0};
    F3 direction{0, 0, 0};
    F3 power{0, 0, 0};
    bool valid = false;
};

// device.cu:630-639
F3 directionAround(F3 axis, float cosine, uint32_t& rng) {
    const float sine = std::sqrt(std::fmax(0.0f, 1.0f - cosine * cosine));
    const float angle = 6.2831853f * rnd(rng);
    const F3 tangent = basisTangent(axis);
    return normalize(axis * cosine + tangent * (sine * mathCos(angle)) +
                     cross(axis, tangent) * (sine * mathSin(angle)));
}

// device.cu:641-650
F3 diskOffset(F3 normal, float radius, uint32_t& rng) {
    const float distance = std::sqrt(rnd(rng)) * radius;
    const float angle = 6.2831853f * rnd(rng);
    const F3 tangent = basisTangent(normal);
    return tangent * (distance * mathCos(angle)) + cross(normal, tangent) * (distance * mathSin(angle));
}

// device.cu:652-712
PhotonEmission emitPhoton(const Ctx& c, uint32_t& rng) {
    PhotonEmission photon;
    float choice = 0.0f;
    const rt_gpu_light* light = selectLight(c, rng, choice);
    if (!light)
        return photon;
    const float normalization = 1.0f / (static_cast<float>(c.p->photonEmissions) * choice);
    const F3 center = f3(c.p->sceneCenter);
    const float sceneRadius = c.p->sceneRadius;
    if (light->type == 4U) {
        const EnvironmentSample source = sampleEnvironment(c, rng);
        photon.direction = -source.direction;
        photon.origin = center - photon.direction * sceneRadius + diskOffset(photon.direction, sceneRadius, rng);
        photon.power = source.radiance * (3.14159265f * sceneRadius * sceneRadius * normalization / source.pdf);
    } else if (light->type == 3U) {
        const float root = std::sqrt(rnd(rng));
        const float u = 1.0f - root;
        const float v = rnd(rng) * root;
        photon.origin = f3(light->a) * u + f3(light->b) * v + f3(light->c) * (1.0f - u - v);
        const F3 side = rnd(rng) < 0.5f ? f3(light->normal) : -f3(light->normal);
        photon.direction = cosineDirection(side, rng);
        photon.origin = photon.origin + photon.direction * 0.001f;
        photon.power = f3(light->emission) * (6.2831853f * light->area * normalization);
    } else if (light->type == 0U) {
        photon.direction = normalize(f3(light->b));
        photon.origin = center - photon.direction * sceneRadius + diskOffset(photon.direction, sceneRadius, rng);
        photon.power = f3(light->emission) * (3.14159265f * sceneRadius * sceneRadius * normalization);
    } else {
        photon.origin = f3(light->a);
        if (light->type == 1U) {
            const float outer = mathCos(light->outerCone);
            const float cosine = 1.0f - rnd(rng) * (1.0f - outer);
            photon.direction = directionAround(normalize(f3(light->b)), cosine, rng);
            const float inner = mathCos(light->innerCone);
            const float falloff = saturate((cosine - outer) / std::fmax(inner - outer, 1e-5f));
            photon.power = f3(light->emission) * (6.2831853f * (1.0f - outer) * falloff * falloff * normalization);
        } else {
            const float cosine = 1.0f - 2.0f * rnd(rng);
            photon.direction = directionAround(F3{0.0f, 1.0f, 0.0f}, cosine, rng);
            photon.power = f3(light->emission) * (12.566371f * normalization);
        }
        photon.origin = photon.origin + photon.direction * 0.001f;
    }
    photon.valid = maximum(photon.power) > 0.0f;
    return photon;
}

// device.cu:722-766
F3 causticLighting(const Ctx& c, const rt_hit& hit, const rt_gpu_material& m) {
    const float diffuseWeight = (1.0f - m.metallic) * (1.0f - m.transmission);
    if (c.p->photonRadius <= 0.0f || diffuseWeight <= 0.0f)
        return {0.0f, 0.0f, 0.0f};
    float flux[3];
    orc_photon_gather(c.p->sortedPhotons, c.p->photonCellStart, c.p->photonBucketCount, c.p->photonRadius,
                      &hit.position.x, &hit.normal.x, 1, flux, 1);
    const float radiusSquared = c.p->photonRadius * c.p->photonRadius;
    const float kernel = 2.0f / (3.14159265f * radiusSquared * 3.14159265f);
    return rgb(m.baseColor) * (diffuseWeight * kernel) * F3{flux[0], flux[1], flux[2]};
}

// The specular-transmission branch shared by the camera path (device.cu:954-997)
// and the photon tracer (device.cu:798-841).  Returns false when the path ends.
bool transmit(const rt_gpu_material& m, const rt_hit& hit, uint32_t& rng, F3& origin, F3& direction, F3& weight,
              int& medium) {
    float ior = m.ior;
    if (m.dispersion > 0.0f) {
        const unsigned channel = static_cast<unsigned>(rnd(rng) * 3.0f) % 3U;
        const float spread = (m.ior - 1.0f) * m.dispersion * 0.5f;
        ior += (static_cast<float>(channel) - 1.0f) * spread;
        weight *= channel == 0U ? F3{3.0f, 0.0f, 0.0f} : channel == 1U ? F3{0.0f, 3.0f, 0.0f} : F3{0.0f, 0.0f, 3.0f};
    }
    const F3 view = -direction;
    const F3 normal = f3(hit.normal);
    const float alpha = roughnessAlpha(m.roughness);
    const F3 microNormal = ggxNormal(view, normal, alpha, rng);
    const float eta = hit.frontFace ? 1.0f / ior : ior;
    const float cosine = std::fmin(std::fmax(dot(view, microNormal), 0.0f), 1.0f);
    F3 transmitted{0, 0, 0};
    const bool totalReflection = !refract(direction, microNormal, eta, transmitted);
    float reflectance = 1.0f;
    if (!totalReflection) {
        transmitted = normalize(transmitted);
        reflectance = fresnel(eta > 1.0f ? std::fabs(dot(transmitted, microNormal)) : cosine, ior);
    }
    const float viewCosine = std::fabs(dot(view, normal));
    if (totalReflection || rnd(rng) < reflectance) {
        direction = reflect(direction, microNormal);
        if (dot(direction, normal) <= 0.0f)
            return false;
        origin = f3(hit.position) + normal * 0.001f;
    } else {
        direction = transmitted;
        origin = f3(hit.position) - normal * 0.001f;
        medium = hit.frontFace ? static_cast<int>(hit.material) : -1;
    }
    weight *= maskingRatio(viewCosine, std::fabs(dot(direction, normal)), alpha);
    return true;
}

// device.cu:768-861 for one emission index; returns true when a photon is stored.
bool tracePhoton(const Ctx& c, uint32_t index, rt_photon& out) {
    uint32_t rng = orc_seeded(index, c.p->photonPass);
    const PhotonEmission emission = emitPhoton(c, rng);
    if (!emission.valid)
        return false;
    F3 origin = emission.origin;
    F3 direction = emission.direction;
    F3 power = emission.power;
    int medium = -1;
    bool specularPath = false;
    for (unsigned depth = 0; depth < 10U; ++depth) {
        rt_hit hit = trace(c, origin, direction);
        if (!hit.found)
            return false;
        const rt_gpu_material m = textured(c, c.p->materials[hit.material], hit);
        hit.normal = r3(mappedNormal(c, hit, m));
        if (medium >= 0)
            power *= absorption(c.p->materials[medium], hit.distance);
        const float diffuseWeight = (1.0f - m.metallic) * (1.0f - m.transmission);
        if (specularPath && diffuseWeight > 0.0f) {
            out.position = hit.position;  // device.cu:714-720
            out.power = r3(power);
            out.normal = hit.normal;
            return true;
        }
        if (m.transmission > 0.0f && rnd(rng) < m.transmission) {
            if (!transmit(m, hit, rng, origin, direction, power, medium))
                return false;
            specularPath = true;
            continue;
        }
        if (m.metallic > 0.5f && m.roughness < 0.3f) {
            const F3 view = -direction;
            const F3 normal = f3(hit.normal);
            const float alpha = roughnessAlpha(m.roughness);
            const F3 microNormal = ggxNormal(view, normal, alpha, rng);
            const F3 reflected = reflect(direction, microNormal);
            const float lightCosine = dot(reflected, normal);
            if (lightCosine <= 0.0f)
                return false;
            power *= schlick(baseReflectance(m), std::fmax(dot(view, microNormal), 0.0f)) *
                     maskingRatio(std::fabs(dot(view, normal)), lightCosine, alpha);
            direction = reflected;
            origin = f3(hit.position) + normal * 0.001f;
            specularPath = true;
            continue;
        }
        return false;
    }
    return false;
}

// device.cu:882-1074 for one pixel
void renderPixel(const Ctx& c, uint32_t px, uint32_t py) {
    const rt_launch_params& p = *c.p;
    const uint32_t index = py * p.width + px;
    uint32_t rng = index * 9781U + p.sample * 6271U + 0x68bc21ebU;
    const float jx = rnd(rng), jy = rnd(rng);
    const float sx = (static_cast<float>(px) + jx) / static_cast<float>(p.width);
    const float sy = (static_cast<float>(py) + jy) / static_cast<float>(p.height);
    const F3 pinhole = normalize(f3(p.cameraW) + (2.0f * sx - 1.0f) * f3(p.cameraU) + (2.0f * sy - 1.0f) * f3(p.cameraV));
    const F3 focalPoint = f3(p.eye) + pinhole * (p.focusDistance / std::fmax(dot(pinhole, f3(p.cameraW)), 1e-4f));
    F3 origin = f3(p.eye) + lensSample(c, rng);
    F3 direction = normalize(focalPoint - origin);
    F3 throughput{1.0f, 1.0f, 1.0f};
    F3 radiance[3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    F3 causticRadiance{0, 0, 0}, guideAlbedo{0, 0, 0}, guideNormal{0, 0, 0};
    unsigned lobe = 0;
    int medium = -1;
    float lastPdf = 0.0f;
    F3 lastOrigin = origin;
    bool lastDelta = true, primaryChain = true, guidePending = true;

    for (unsigned depth = 0; depth < p.maxDepth; ++depth) {
        rt_hit hit = trace(c, origin, direction);
        if (!hit.found) {
            const float pdf = lastDelta ? 0.0f : environmentPdf(c, direction);
            const float weight = lastDelta ? 1.0f : powerHeuristic(lastPdf, pdf);
            radiance[lobe] += throughput * environment(c, direction) * weight;
            break;
        }
        const rt_gpu_material m = textured(c, p.materials[hit.material], hit);
        hit.normal = r3(mappedNormal(c, hit, m));
        const F3 normal = f3(hit.normal);
        const F3 view = -direction;
        if (depth == 0U) {
            lobe = m.transmission > 0.5f ? 2U : m.metallic > 0.5f ? 1U : 0U;
            guideNormal = {dot(normal, f3(p.lensU)), dot(normal, f3(p.lensV)), dot(normal, f3(p.cameraW))};
        }
        if (guidePending) {
            const bool clearGlass = m.transmission > 0.95f && m.roughness < 0.1f;
            if (!clearGlass) {
                if (m.transmission < 0.05f)
                    guideAlbedo = rgb(m.baseColor);
                guidePending = false;
            }
        }
        if (medium >= 0)
            throughput *= absorption(p.materials[medium], hit.distance);
        const float lightPdf = lastDelta ? 0.0f : emissivePdf(c, lastOrigin, hit);
        const float emissionWeight = lastDelta ? 1.0f : powerHeuristic(lastPdf, lightPdf);
        radiance[lobe] += throughput * f3(m.emissive) * (m.emissiveStrength * emissionWeight);
        radiance[lobe] += throughput * directLighting(c, hit, m, view, rng);
        if (primaryChain)
            causticRadiance += throughput * causticLighting(c, hit, m);

        const bool transmissive = m.transmission > 0.0f && rnd(rng) < m.transmission;
        if (transmissive) {
            lastDelta = true;
            if (!transmit(m, hit, rng, origin, direction, throughput, medium))
                break;
        } else {
            const F3 color = rgb(m.baseColor);
            const F3 f0 = baseReflectance(m);
            const float viewCosine = std::fmax(dot(view, normal), 0.0f);
            const float probability = specularProbability(m, view, normal);
            if (rnd(rng) < probability) {
                lastDelta = true;
                const float alpha = roughnessAlpha(m.roughness);
                const F3 microNormal = ggxNormal(view, normal, alpha, rng);
                direction = reflect(direction, microNormal);
                const float lightCosine = dot(direction, normal);
                if (lightCosine <= 0.0f)
                    break;
                throughput *= schlick(f0, std::fmax(dot(view, microNormal), 0.0f)) *
                              (maskingRatio(viewCosine, lightCosine, alpha) / probability);
            } else {
                lastDelta = false;
                primaryChain = false;
                direction = cosineDirection(normal, rng);
                const F3 reflectance = schlick(f0, viewCosine);
                const F3 diffuse{1.0f - reflectance.x, 1.0f - reflectance.y, 1.0f - reflectance.z};
                throughput *= color * diffuse * ((1.0f - m.metallic) / (1.0f - probability));
                lastPdf = (1.0f - m.transmission) * (1.0f - probability) * std::fmax(dot(normal, direction), 0.0f) /
                          3.14159265f;
            }
            origin = f3(hit.position) + normal * 0.001f;
        }
        lastOrigin = f3(hit.position);
        if (depth > 2) {
            const float survival = std::fmin(std::fmax(maximum(throughput), 0.05f), 0.95f);
            if (rnd(rng) > survival)
                break;
            throughput /= survival;
        }
    }

    // device.cu:1043-1074
    auto blend = [&](rt_float4* buffer, F3 value, float weight) {
        const F3 previous = rgb(buffer[index]);
        const F3 next = previous + (value - previous) * weight;
        buffer[index] = rt_float4{next.x, next.y, next.z, 1.0f};
        return next;
    };
    const float weight = 1.0f / static_cast<float>(p.sample + 1U);
    const F3 sample = radiance[0] + radiance[1] + radiance[2] + causticRadiance;
    const F3 accumulated = blend(p.accumulation, sample, weight);
    blend(p.diffuse, radiance[0], weight);
    blend(p.reflection, radiance[1], weight);
    blend(p.refraction, radiance[2], weight);
    blend(p.caustics, causticRadiance, weight);
    blend(p.albedoGuide, guideAlbedo, weight);
    blend(p.normalGuide, guideNormal, weight);
    if (p.output) {
        const F3 mapped = aces(accumulated * mathExp2(p.exposure));
        p.output[index] = rt_uchar4{static_cast<unsigned char>(mapped.x * 255.0f + 0.5f),
                                    static_cast<unsigned char>(mapped.y * 255.0f + 0.5f),
                                    static_cast<unsigned char>(mapped.z * 255.0f + 0.5f), 255};
    }
}

template <typename F>
void parallelRange(uint32_t begin, uint32_t end, int threads, F&& body) {
    if (threads <= 1 || end - begin < 2) {
        body(begin, end, 0);
        return;
    }
    std::vector<std::thread> pool;
    const uint32_t n = end - begin;
    const uint32_t chunk = (n + threads - 1) / threads;
    for (int t = 0; t < threads; ++t) {
        const uint32_t b = begin + static_cast<uint32_t>(std::min<uint64_t>(n, static_cast<uint64_t>(t) * chunk));
        const uint32_t e = begin + static_cast<uint32_t>(std::min<uint64_t>(n, static_cast<uint64_t>(t + 1) * chunk));
        if (b < e)
            pool.emplace_back(body, b, e, t);
    }
    for (auto& t : pool)
        t.join();
}

}  // namespace

extern "C" {

// One camera sample (params->sample) for the pixels [pixelBegin, pixelEnd) in
// row-major order: the body of __raygen__render, device.cu:882-1074.  Every
// pointer in params is a HOST pointer; params->scene is ignored (the oracle
// scene is passed explicitly).
void orc_render_sample(const orc_scene* scene, const rt_launch_params* params, uint32_t pixelBegin, uint32_t pixelEnd,
                       int threads) {
    const Ctx c{scene, params};
    parallelRange(pixelBegin, pixelEnd, threads, [&](uint32_t b, uint32_t e, int) {
        for (uint32_t i = b; i < e; ++i)
            renderPixel(c, i % params->width, i / params->width);
    });
}

// The photon pass of __raygen__photons (device.cu:768-861) for emission indices
// [first, first + count): stored photons are appended to `out` in emission
// order (the canonical order; the reference appends in atomic order,
// device.cu:716).  Returns the number of photons the tracer wanted to store;
// at most `capacity` are written.
uint32_t orc_trace_photons_indexed(const orc_scene* scene, const rt_launch_params* params, uint32_t first,
                                   uint32_t count, rt_photon* out, uint32_t* emission, uint32_t capacity, int threads) {
    const Ctx c{scene, params};
    const int workers = std::max(1, threads);
    std::vector<std::vector<std::pair<uint32_t, rt_photon>>> parts(workers);
    parallelRange(first, first + count, workers, [&](uint32_t b, uint32_t e, int t) {
        for (uint32_t i = b; i < e; ++i) {
            rt_photon photon;
            if (tracePhoton(c, i, photon))
                parts[t].emplace_back(i, photon);
        }
    });
    uint32_t stored = 0;
    for (const auto& part : parts)
        for (const auto& entry : part) {
            if (stored < capacity) {
                out[stored] = entry.second;
                if (emission)
                    emission[stored] = entry.first;  // which emission stored this photon
            }
            ++stored;
        }
    return stored;
}

uint32_t orc_trace_photons(const orc_scene* scene, const rt_launch_params* params, uint32_t first, uint32_t count,
                           rt_photon* out, uint32_t capacity, int threads) {
    return orc_trace_photons_indexed(scene, params, first, count, out, nullptr, capacity, threads);
}

void orc_light_walks(const rt_gpu_light* lights, uint32_t lightCount, float lightWeight, const float* uniform,
                     const uint32_t* instance, const uint32_t* primitive, uint32_t n, uint32_t* chosen, uint32_t* found,
                     int threads) {
    if (lightCount == 0U)
        return;
    auto work = [&](uint32_t begin, uint32_t end) {
        for (uint32_t i = begin; i < end; ++i) {
            if (uniform && chosen) {  // device.cu:433-447
                float target = uniform[i] * lightWeight;
                uint32_t selected = lightCount - 1U;
                for (uint32_t index = 0; index < lightCount; ++index) {
                    target -= lights[index].weight;
                    if (target <= 0.0f) {
                        selected = index;
                        break;
                    }
                }
                chosen[i] = selected;
            }
            if (instance && primitive && found) {  // device.cu:607-621
                uint32_t light = 0xffffffffU;
                for (uint32_t index = 0; index < lightCount; ++index)
                    if (lights[index].type == 3U && lights[index].instance == instance[i] &&
                        lights[index].primitive == primitive[i]) {
                        light = index;
                        break;
                    }
                found[i] = light;
            }
        }
    };
    const uint32_t workers = static_cast<uint32_t>(threads < 1 ? 1 : threads);
    std::vector<std::thread> pool;
    for (uint32_t t = 0; t < workers; ++t)
        pool.emplace_back(work, static_cast<uint32_t>(static_cast<uint64_t>(n) * t / workers),
                          static_cast<uint32_t>(static_cast<uint64_t>(n) * (t + 1U) / workers));
    for (std::thread& t : pool)
        t.join();
}

}  // extern "C"

// The shared elementary functions over arrays, for the tests that pin them: which = 0 sine, 1 cosine,
// 2 power(x, y), 3 arcTangent2(x = ordinate, y = abscissa), 4 arcCosine, 5 exp2.
extern "C" void orc_shared_math(int which, const float* x, const float* y, uint32_t n, float* out) {
    for (uint32_t i = 0; i < n; ++i) {
        switch (which) {
        case 0: out[i] = rtm::sine(x[i]); break;
        case 1: out[i] = rtm::cosine(x[i]); break;
        case 2: out[i] = rtm::power(x[i], y[i]); break;
        case 3: out[i] = rtm::arcTangent2(x[i], y[i]); break;
        case 4: out[i] = rtm::arcCosine(x[i]); break;
        default: out[i] = rtm::exp2(x[i]); break;
        }
    }
}
