// CPU oracle: ray/scene closest-hit and any-hit queries and hit attribute
// reconstruction.  Test infrastructure only (see oracle.h).
//
// The reference reaches geometry through one optixTrace call (device.cu:388-397)
// whose traversal and ray/triangle arithmetic live in the closed OptiX driver, so
// there is nothing to restate line by line for the intersection itself: PARITY
// UNPINNED for t, barycentrics and tie resolution.  The conventions of the call
// are kept: rays live in world space, instances carry a 3x4 object->world
// transform (renderer.cpp:398-412), traversal is two-sided (no cull flags,
// renderer.cpp:365, 410), the ray direction is not renormalised in object space
// so t is the same parameter in both spaces, and the hit interval is
// tmin < t <= tmax.
//
// Intersection arithmetic (the spec the HIP kernels implement bit for bit), all
// binary32, every operation individually rounded (-ffp-contract=off), in the
// written association order:
//   object-space ray:  o' = ((m0*ox + m1*oy) + m2*oz) + m3 per row of the
//                      world->object 3x4, d' likewise without the translation.
//   watertight test (Woop, Benthin, Wald 2013), per ray and space:
//     kz = 0; if |d.y| > |d[kz]| kz = 1; if |d.z| > |d[kz]| kz = 2;
//     kx = (kz+1)%3; ky = (kx+1)%3; if d[kz] < 0 swap(kx, ky);
//     Sx = d[kx]/d[kz]; Sy = d[ky]/d[kz]; Sz = 1/d[kz];
//   per triangle: A = v0 - o', B = v1 - o', C = v2 - o';
//     Ax = A[kx] - Sx*A[kz]; Ay = A[ky] - Sy*A[kz]; same for B, C;
//     U = Cx*By - Cy*Bx; V = Ax*Cy - Ay*Cx; W = Bx*Ay - By*Ax;
//     if any of U, V, W == 0: all three recomputed with the products and the
//       difference in binary64, rounded to binary32;
//     miss if (U<0 || V<0 || W<0) && (U>0 || V>0 || W>0);
//     det = (U + V) + W; miss if det == 0;
//     T = (U*(Sz*A[kz]) + V*(Sz*B[kz])) + W*(Sz*C[kz]);
//     rcp = 1/det; t = T*rcp; miss unless t > tmin && t <= tmax;
//     u = V*rcp (weight of v1), v = W*rcp (weight of v2).
//   closest hit: smallest t; exact ties go to the lowest (instance, primitive).
#include "oracle.h"
#include "oracle_internal.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <thread>
#include <vector>

namespace {

struct V3 {
    float x, y, z;
    float operator[](int i) const { return i == 0 ? x : i == 1 ? y : z; }
};

V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }

struct Box {
    float lo[3], hi[3];
};

struct BvhNode {
    Box box;
    uint32_t left;   // internal: index of left child (right = left + 1); leaf: first triangle slot
    uint32_t count;  // 0 for internal nodes
};

struct Geometry {
    std::vector<rt_gpu_vertex> vertices;
    std::vector<uint32_t> indices;
    uint32_t material = 0;
    std::vector<BvhNode> nodes;
    std::vector<uint32_t> order;  // triangle ids in BVH leaf order
};

struct Instance {
    uint32_t geometry;
    float toWorld[12];   // row-major 3x4, renderer.cpp:403-406
    float toObject[12];  // row-major 3x4 inverse
    Box worldBox;
};

struct RaySetup {
    int kx, ky, kz;
    float Sx, Sy, Sz;
    V3 o, d;
};

struct Best {
    float t, u, v;
    uint32_t prim, inst;
};

// Inverse of the affine 3x4 in binary64, rounded once to binary32.
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
        out[row * 4 + 3] = static_cast<float>(-(inv[row * 3 + 0] * tx + inv[row * 3 + 1] * ty + inv[row * 3 + 2] * tz));
    }
}

V3 xformPoint(const float* m, V3 p) {
    return {((m[0] * p.x + m[1] * p.y) + m[2] * p.z) + m[3],
            ((m[4] * p.x + m[5] * p.y) + m[6] * p.z) + m[7],
            ((m[8] * p.x + m[9] * p.y) + m[10] * p.z) + m[11]};
}

V3 xformVector(const float* m, V3 p) {
    return {(m[0] * p.x + m[1] * p.y) + m[2] * p.z,
            (m[4] * p.x + m[5] * p.y) + m[6] * p.z,
            (m[8] * p.x + m[9] * p.y) + m[10] * p.z};
}

RaySetup setupRay(V3 o, V3 d) {
    RaySetup s;
    s.o = o;
    s.d = d;
    int kz = 0;
    if (std::fabs(d.y) > std::fabs(d[kz])) kz = 1;
    if (std::fabs(d.z) > std::fabs(d[kz])) kz = 2;
    int kx = (kz + 1) % 3;
    int ky = (kx + 1) % 3;
    if (d[kz] < 0.0f) std::swap(kx, ky);
    s.kx = kx;
    s.ky = ky;
    s.kz = kz;
    s.Sx = d[kx] / d[kz];
    s.Sy = d[ky] / d[kz];
    s.Sz = 1.0f / d[kz];
    return s;
}

bool intersectTriangle(const RaySetup& s, V3 v0, V3 v1, V3 v2, float tmin, float tmax, float& t, float& u, float& v) {
    const V3 A = sub(v0, s.o), B = sub(v1, s.o), C = sub(v2, s.o);
    const float Ax = A[s.kx] - s.Sx * A[s.kz];
    const float Ay = A[s.ky] - s.Sy * A[s.kz];
    const float Bx = B[s.kx] - s.Sx * B[s.kz];
    const float By = B[s.ky] - s.Sy * B[s.kz];
    const float Cx = C[s.kx] - s.Sx * C[s.kz];
    const float Cy = C[s.ky] - s.Sy * C[s.kz];
    float U = Cx * By - Cy * Bx;
    float V = Ax * Cy - Ay * Cx;
    float W = Bx * Ay - By * Ax;
    if (U == 0.0f || V == 0.0f || W == 0.0f) {
        U = static_cast<float>(static_cast<double>(Cx) * static_cast<double>(By) -
                               static_cast<double>(Cy) * static_cast<double>(Bx));
        V = static_cast<float>(static_cast<double>(Ax) * static_cast<double>(Cy) -
                               static_cast<double>(Ay) * static_cast<double>(Cx));
        W = static_cast<float>(static_cast<double>(Bx) * static_cast<double>(Ay) -
                               static_cast<double>(By) * static_cast<double>(Ax));
    }
    if ((U < 0.0f || V < 0.0f || W < 0.0f) && (U > 0.0f || V > 0.0f || W > 0.0f))
        return false;
    const float det = (U + V) + W;
    if (det == 0.0f)
        return false;
    const float Az = s.Sz * A[s.kz];
    const float Bz = s.Sz * B[s.kz];
    const float Cz = s.Sz * C[s.kz];
    const float T = (U * Az + V * Bz) + W * Cz;
    const float rcp = 1.0f / det;
    const float tt = T * rcp;
    if (!(tt > tmin && tt <= tmax))
        return false;
    t = tt;
    u = V * rcp;
    v = W * rcp;
    return true;
}

V3 position(const Geometry& g, uint32_t vertex) {
    const rt_float3& p = g.vertices[vertex].position;
    return {p.x, p.y, p.z};
}

void consider(Best& best, float t, float u, float v, uint32_t inst, uint32_t prim) {
    if (t < best.t || (t == best.t && (inst < best.inst || (inst == best.inst && prim < best.prim)))) {
        best.t = t;
        best.u = u;
        best.v = v;
        best.inst = inst;
        best.prim = prim;
    }
}

// Conservative slab test: the interval is widened by a relative 2^-20 so that a
// triangle accepted by the watertight test is never culled by its box.
bool hitBox(const Box& b, V3 o, const float* invd, float tmin, float tmax) {
    float tn = tmin, tf = tmax;
    const float oo[3] = {o.x, o.y, o.z};
    for (int a = 0; a < 3; ++a) {
        float t0 = (b.lo[a] - oo[a]) * invd[a];
        float t1 = (b.hi[a] - oo[a]) * invd[a];
        if (t0 > t1) std::swap(t0, t1);
        // NaN (0 * inf) keeps the previous bound
        if (t0 > tn) tn = t0;
        if (t1 < tf) tf = t1;
    }
    return tn * (1.0f - 0x1p-20f) <= tf * (1.0f + 0x1p-20f);
}

// The margin comes from the largest coordinate of the box and goes on every axis: the distance the triangle test
// computes is off by a few ulps of the triangle's size, not of the distance, and a box without extent along an axis
// (a floor at y = 0) would otherwise have no margin there at all.
void padBox(Box& b) {
    float m = 0.0f;
    for (int a = 0; a < 3; ++a)
        m = std::max(m, std::max(std::max(std::fabs(b.lo[a]), std::fabs(b.hi[a])), b.hi[a] - b.lo[a]));
    const float pad = std::max(m * 0x1p-18f, 1e-30f);
    for (int a = 0; a < 3; ++a) {
        b.lo[a] -= pad;
        b.hi[a] += pad;
    }
}

Box triangleBox(const Geometry& g, uint32_t tri) {
    Box b;
    for (int a = 0; a < 3; ++a) {
        b.lo[a] = INFINITY;
        b.hi[a] = -INFINITY;
    }
    for (int k = 0; k < 3; ++k) {
        const V3 p = position(g, g.indices[3 * static_cast<size_t>(tri) + k]);
        for (int a = 0; a < 3; ++a) {
            b.lo[a] = std::min(b.lo[a], p[a]);
            b.hi[a] = std::max(b.hi[a], p[a]);
        }
    }
    return b;
}

void grow(Box& b, const Box& o) {
    for (int a = 0; a < 3; ++a) {
        b.lo[a] = std::min(b.lo[a], o.lo[a]);
        b.hi[a] = std::max(b.hi[a], o.hi[a]);
    }
}

float halfArea(const Box& b) {
    const float x = b.hi[0] - b.lo[0], y = b.hi[1] - b.lo[1], z = b.hi[2] - b.lo[2];
    return x * y + y * z + z * x;
}

// Binned-SAH binary BVH over one geometry (object space).  The oracle's own
// acceleration structure; unrelated to the product's GPU builder.
void buildBvh(Geometry& g) {
    const uint32_t n = static_cast<uint32_t>(g.indices.size() / 3);
    g.order.resize(n);
    std::vector<Box> boxes(n);
    std::vector<V3> centers(n);
    for (uint32_t i = 0; i < n; ++i) {
        g.order[i] = i;
        boxes[i] = triangleBox(g, i);
        centers[i] = {0.5f * (boxes[i].lo[0] + boxes[i].hi[0]), 0.5f * (boxes[i].lo[1] + boxes[i].hi[1]),
                      0.5f * (boxes[i].lo[2] + boxes[i].hi[2])};
    }
    g.nodes.clear();
    g.nodes.reserve(2 * static_cast<size_t>(n) + 1);
    g.nodes.push_back({});
    struct Task {
        uint32_t node, first, count;
    };
    std::vector<Task> stack{{0, 0, n}};
    constexpr int Bins = 16;
    while (!stack.empty()) {
        const Task task = stack.back();
        stack.pop_back();
        Box box, cbox;
        for (int a = 0; a < 3; ++a) {
            box.lo[a] = cbox.lo[a] = INFINITY;
            box.hi[a] = cbox.hi[a] = -INFINITY;
        }
        for (uint32_t i = task.first; i < task.first + task.count; ++i) {
            const uint32_t tri = g.order[i];
            grow(box, boxes[tri]);
            for (int a = 0; a < 3; ++a) {
                cbox.lo[a] = std::min(cbox.lo[a], centers[tri][a]);
                cbox.hi[a] = std::max(cbox.hi[a], centers[tri][a]);
            }
        }
        Box padded = box;
        padBox(padded);
        g.nodes[task.node].box = padded;
        if (task.count <= 4) {
            g.nodes[task.node].left = task.first;
            g.nodes[task.node].count = task.count;
            continue;
        }
        int axis = 0;
        for (int a = 1; a < 3; ++a)
            if (cbox.hi[a] - cbox.lo[a] > cbox.hi[axis] - cbox.lo[axis]) axis = a;
        const float extent = cbox.hi[axis] - cbox.lo[axis];
        uint32_t mid = task.first + task.count / 2;
        bool split = false;
        if (extent > 0.0f) {
            Box binBox[Bins];
            uint32_t binCount[Bins] = {};
            for (auto& b : binBox)
                for (int a = 0; a < 3; ++a) {
                    b.lo[a] = INFINITY;
                    b.hi[a] = -INFINITY;
                }
            const float scale = Bins / extent;
            auto binOf = [&](uint32_t tri) {
                return std::min(Bins - 1, std::max(0, static_cast<int>((centers[tri][axis] - cbox.lo[axis]) * scale)));
            };
            for (uint32_t i = task.first; i < task.first + task.count; ++i) {
                const int b = binOf(g.order[i]);
                ++binCount[b];
                grow(binBox[b], boxes[g.order[i]]);
            }
            float rightArea[Bins];
            uint32_t rightCount[Bins];
            Box acc;
            for (int a = 0; a < 3; ++a) {
                acc.lo[a] = INFINITY;
                acc.hi[a] = -INFINITY;
            }
            uint32_t cnt = 0;
            for (int b = Bins - 1; b > 0; --b) {
                if (binCount[b]) grow(acc, binBox[b]);
                cnt += binCount[b];
                rightArea[b] = cnt ? halfArea(acc) : 0.0f;
                rightCount[b] = cnt;
            }
            for (int a = 0; a < 3; ++a) {
                acc.lo[a] = INFINITY;
                acc.hi[a] = -INFINITY;
            }
            cnt = 0;
            float bestCost = INFINITY;
            int bestBin = -1;
            for (int b = 0; b < Bins - 1; ++b) {
                if (binCount[b]) grow(acc, binBox[b]);
                cnt += binCount[b];
                if (!cnt || !rightCount[b + 1]) continue;
                const float cost = halfArea(acc) * cnt + rightArea[b + 1] * rightCount[b + 1];
                if (cost < bestCost) {
                    bestCost = cost;
                    bestBin = b;
                }
            }
            if (bestBin >= 0) {
                auto* begin = g.order.data() + task.first;
                auto* end = begin + task.count;
                auto* m = std::partition(begin, end, [&](uint32_t tri) { return binOf(tri) <= bestBin; });
                mid = task.first + static_cast<uint32_t>(m - begin);
                split = mid > task.first && mid < task.first + task.count;
            }
        }
        if (!split) {
            mid = task.first + task.count / 2;
            std::nth_element(g.order.begin() + task.first, g.order.begin() + mid,
                             g.order.begin() + task.first + task.count,
                             [&](uint32_t a, uint32_t b) { return centers[a][axis] < centers[b][axis]; });
        }
        const uint32_t left = static_cast<uint32_t>(g.nodes.size());
        g.nodes.push_back({});
        g.nodes.push_back({});
        g.nodes[task.node].left = left;
        g.nodes[task.node].count = 0;
        stack.push_back({left, task.first, mid - task.first});
        stack.push_back({left + 1, mid, task.first + task.count - mid});
    }
}

struct Scene {
    std::vector<Geometry> geometries;
    std::vector<Instance> instances;
};

void testTriangle(const Geometry& g, const RaySetup& s, uint32_t tri, uint32_t inst, float tmin, Best& best) {
    float t, u, v;
    const size_t base = 3 * static_cast<size_t>(tri);
    if (intersectTriangle(s, position(g, g.indices[base]), position(g, g.indices[base + 1]),
                          position(g, g.indices[base + 2]), tmin, best.t, t, u, v))
        consider(best, t, u, v, inst, tri);
}

void traceInstance(const Scene& scene, uint32_t index, V3 o, V3 d, float tmin, bool useBvh, bool any, Best& best) {
    const Instance& instance = scene.instances[index];
    const Geometry& g = scene.geometries[instance.geometry];
    const RaySetup s = setupRay(xformPoint(instance.toObject, o), xformVector(instance.toObject, d));
    const uint32_t triangles = static_cast<uint32_t>(g.indices.size() / 3);
    if (!useBvh) {
        for (uint32_t tri = 0; tri < triangles; ++tri) {
            testTriangle(g, s, tri, index, tmin, best);
            if (any && best.prim != 0xffffffffU) return;
        }
        return;
    }
    if (g.nodes.empty()) return;
    const float invd[3] = {1.0f / s.d.x, 1.0f / s.d.y, 1.0f / s.d.z};
    uint32_t stack[128];
    int top = 0;
    stack[top++] = 0;
    while (top) {
        const BvhNode& node = g.nodes[stack[--top]];
        if (!hitBox(node.box, s.o, invd, tmin, best.t)) continue;
        if (node.count) {
            for (uint32_t i = 0; i < node.count; ++i) {
                testTriangle(g, s, g.order[node.left + i], index, tmin, best);
                if (any && best.prim != 0xffffffffU) return;
            }
        } else if (top + 2 <= 128) {
            stack[top++] = node.left;
            stack[top++] = node.left + 1;
        }
    }
}

Best traceScene(const Scene& scene, const float* org, const float* dir, bool useBvh, bool any) {
    const V3 o{org[0], org[1], org[2]};
    const V3 d{dir[0], dir[1], dir[2]};
    // A ray starts at its origin at the earliest: a tmin below zero (or one that is no number) counts as zero.  The
    // reference's traversal takes no such ray at all (OptiX reports a negative or NaN tmin as an invalid ray).
    const float tmin = std::fmax(org[3], 0.0f);
    Best best{dir[3], 0.0f, 0.0f, 0xffffffffU, 0xffffffffU};
    const float invd[3] = {1.0f / d.x, 1.0f / d.y, 1.0f / d.z};
    for (uint32_t i = 0; i < scene.instances.size(); ++i) {
        if (useBvh && !hitBox(scene.instances[i].worldBox, o, invd, tmin, best.t)) continue;
        traceInstance(scene, i, o, d, tmin, useBvh, any, best);
        if (any && best.prim != 0xffffffffU) break;
    }
    return best;
}

template <typename F>
void parallelFor(uint32_t n, int threads, F&& body) {
    if (threads <= 1) {
        body(0U, n);
        return;
    }
    std::vector<std::thread> pool;
    const uint32_t chunk = (n + threads - 1) / threads;
    for (int t = 0; t < threads; ++t) {
        const uint32_t begin = static_cast<uint32_t>(std::min<uint64_t>(n, static_cast<uint64_t>(t) * chunk));
        const uint32_t end = static_cast<uint32_t>(std::min<uint64_t>(n, static_cast<uint64_t>(begin) + chunk));
        if (begin < end) pool.emplace_back(body, begin, end);
    }
    for (auto& t : pool) t.join();
}

V3 scale(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
V3 add(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
// device.cu:40-42, 73-75: divide = multiply by the reciprocal
V3 normalize(V3 a) { return scale(a, 1.0f / std::sqrt(dot(a, a))); }
V3 f3(const rt_float3& a) { return {a.x, a.y, a.z}; }

// device.cu:1081-1145.  optixTransformNormalFromObjectToWorldSpace multiplies
// by the transpose of the world->object 3x3; optixTransformVectorFrom... by the
// object->world 3x3.  Their internal association order is not published; the
// row-by-row (a*x + b*y) + c*z order is used here.
void shadeOne(const Scene& scene, V3 origin, V3 ray, float t, float bx, float by, uint32_t prim, uint32_t inst,
              rt_hit& hit) {
    std::memset(&hit, 0, sizeof(hit));  // device.cu:390
    if (prim == 0xffffffffU) return;    // device.cu:1077-1079
    const Instance& instance = scene.instances[inst];
    const Geometry& g = scene.geometries[instance.geometry];
    const size_t base = 3 * static_cast<size_t>(prim);
    const rt_gpu_vertex& v0 = g.vertices[g.indices[base]];
    const rt_gpu_vertex& v1 = g.vertices[g.indices[base + 1]];
    const rt_gpu_vertex& v2 = g.vertices[g.indices[base + 2]];
    const float b0 = 1.0f - bx - by;  // device.cu:1086
    V3 normal = add(add(scale(f3(v0.normal), b0), scale(f3(v1.normal), bx)), scale(f3(v2.normal), by));
    if (dot(normal, normal) < 1e-12f)
        normal = cross(sub(f3(v1.position), f3(v0.position)), sub(f3(v2.position), f3(v0.position)));
    const V3 objectNormal = normalize(normal);
    const float* w2o = instance.toObject;
    // transpose of the world->object 3x3
    normal = normalize(V3{(w2o[0] * normal.x + w2o[4] * normal.y) + w2o[8] * normal.z,
                          (w2o[1] * normal.x + w2o[5] * normal.y) + w2o[9] * normal.z,
                          (w2o[2] * normal.x + w2o[6] * normal.y) + w2o[10] * normal.z});
    hit.frontFace = dot(normal, ray) < 0.0f;  // device.cu:1099
    const V3 facing = hit.frontFace ? normal : scale(normal, -1.0f);
    hit.normal = {facing.x, facing.y, facing.z};
    float tw = v0.tangent.w * b0 + v1.tangent.w * bx + v2.tangent.w * by;
    V3 tangent{v0.tangent.x * b0 + v1.tangent.x * bx + v2.tangent.x * by,
               v0.tangent.y * b0 + v1.tangent.y * bx + v2.tangent.y * by,
               v0.tangent.z * b0 + v1.tangent.z * bx + v2.tangent.z * by};
    if (dot(tangent, tangent) < 1e-12f) {  // device.cu:1111-1123
        const float d1x = v1.uv.x - v0.uv.x, d1y = v1.uv.y - v0.uv.y;
        const float d2x = v2.uv.x - v0.uv.x, d2y = v2.uv.y - v0.uv.y;
        const float determinant = d1x * d2y - d1y * d2x;
        if (std::fabs(determinant) > 1e-8f) {
            const V3 e1 = sub(f3(v1.position), f3(v0.position));
            const V3 e2 = sub(f3(v2.position), f3(v0.position));
            tangent = scale(sub(scale(e1, d2y), scale(e2, d1y)), 1.0f / determinant);
        } else {
            tangent = cross(std::fabs(objectNormal.x) > 0.5f ? V3{0.0f, 1.0f, 0.0f} : V3{1.0f, 0.0f, 0.0f},
                            objectNormal);
        }
        tw = 1.0f;
    }
    tangent = xformVector(instance.toWorld, tangent);  // device.cu:1124-1125
    tangent = normalize(sub(tangent, scale(facing, dot(tangent, facing))));
    hit.tangent.x = tangent.x;
    hit.tangent.y = tangent.y;
    hit.tangent.z = tangent.z;
    hit.tangent.w = tw < 0.0f ? -1.0f : 1.0f;
    hit.uv.x = v0.uv.x * b0 + v1.uv.x * bx + v2.uv.x * by;
    hit.uv.y = v0.uv.y * b0 + v1.uv.y * bx + v2.uv.y * by;
    hit.uv1.x = v0.uv1.x * b0 + v1.uv1.x * bx + v2.uv1.x * by;
    hit.uv1.y = v0.uv1.y * b0 + v1.uv1.y * bx + v2.uv1.y * by;
    const V3 p = add(origin, scale(ray, t));  // device.cu:1139
    hit.position = {p.x, p.y, p.z};
    hit.distance = t;
    hit.material = g.material;
    hit.instance = inst;
    hit.primitive = prim;
    hit.found = 1;
}

}  // namespace

struct orc_scene : Scene {};

namespace orc {

std::atomic<uint64_t> raysTraced{0};

// trace() of device.cu:388-397 followed by the closest-hit / miss programs.
rt_hit traceHit(const orc_scene* scene, const float origin[3], const float direction[3], float tmax) {
    raysTraced.fetch_add(1, std::memory_order_relaxed);
    const float o4[4] = {origin[0], origin[1], origin[2], 0.001f};
    const float d4[4] = {direction[0], direction[1], direction[2], tmax};
    const Best best = traceScene(*scene, o4, d4, true, false);
    rt_hit hit;
    shadeOne(*scene, V3{origin[0], origin[1], origin[2]}, V3{direction[0], direction[1], direction[2]}, best.t, best.u,
             best.v, best.prim, best.inst, hit);
    return hit;
}

const rt_gpu_vertex* geometryVertices(const orc_scene* scene, uint32_t geometry, uint32_t* count) {
    *count = static_cast<uint32_t>(scene->geometries[geometry].vertices.size());
    return scene->geometries[geometry].vertices.data();
}

}  // namespace orc

extern "C" {

uint64_t orc_ray_counter(int reset) {
    return reset ? orc::raysTraced.exchange(0) : orc::raysTraced.load();
}

orc_scene* orc_scene_create(void) { return new orc_scene(); }
void orc_scene_destroy(orc_scene* s) { delete s; }

int orc_scene_add_geometry(orc_scene* s, const rt_gpu_vertex* vertices, uint32_t vertexCount, const uint32_t* indices,
                           uint32_t triangleCount, uint32_t material) {
    Geometry g;
    g.vertices.assign(vertices, vertices + vertexCount);
    g.indices.assign(indices, indices + 3 * static_cast<size_t>(triangleCount));
    g.material = material;
    s->geometries.push_back(std::move(g));
    return static_cast<int>(s->geometries.size()) - 1;
}

int orc_scene_add_instance(orc_scene* s, uint32_t geometry, const float* transform16) {
    Instance instance{};
    instance.geometry = geometry;
    // renderer.cpp:403-406: row-major 3x4 from the column-major 4x4
    for (int row = 0; row < 3; ++row)
        for (int column = 0; column < 4; ++column)
            instance.toWorld[row * 4 + column] = transform16[column * 4 + row];
    s->instances.push_back(instance);
    return static_cast<int>(s->instances.size()) - 1;
}

void orc_scene_commit(orc_scene* s) {
    for (auto& g : s->geometries)
        if (g.nodes.empty()) buildBvh(g);
    for (auto& instance : s->instances) {
        invertAffine(instance.toWorld, instance.toObject);
        const Geometry& g = s->geometries[instance.geometry];
        Box box;
        for (int a = 0; a < 3; ++a) {
            box.lo[a] = INFINITY;
            box.hi[a] = -INFINITY;
        }
        for (const auto& vertex : g.vertices) {
            const V3 p = xformPoint(instance.toWorld, f3(vertex.position));
            for (int a = 0; a < 3; ++a) {
                box.lo[a] = std::min(box.lo[a], p[a]);
                box.hi[a] = std::max(box.hi[a], p[a]);
            }
        }
        // the object-space test sees the ray through the rounded inverse, so
        // leave a wider margin than for object-space boxes
        float m = 0.0f;
        for (int a = 0; a < 3; ++a)
            m = std::max(m, std::max(std::fabs(box.lo[a]), std::fabs(box.hi[a])) + (box.hi[a] - box.lo[a]));
        const float pad = std::max(m * 0x1p-16f, 1e-30f);
        for (int a = 0; a < 3; ++a) {
            box.lo[a] -= pad;
            box.hi[a] += pad;
        }
        instance.worldBox = box;
    }
}

void orc_scene_inverse(const orc_scene* s, uint32_t instance, float* out12) {
    std::memcpy(out12, s->instances[instance].toObject, sizeof(float) * 12);
}

void orc_trace_closest(const orc_scene* s, const float* org_tmin, const float* dir_tmax, uint32_t n, int use_bvh,
                       int threads, float* tuv, uint32_t* prim, uint32_t* inst) {
    parallelFor(n, threads, [=](uint32_t begin, uint32_t end) {
        for (uint32_t i = begin; i < end; ++i) {
            const Best best = traceScene(*s, org_tmin + 4 * static_cast<size_t>(i), dir_tmax + 4 * static_cast<size_t>(i),
                                         use_bvh != 0, false);
            const bool found = best.prim != 0xffffffffU;
            tuv[3 * static_cast<size_t>(i) + 0] = found ? best.t : -1.0f;
            tuv[3 * static_cast<size_t>(i) + 1] = found ? best.u : 0.0f;
            tuv[3 * static_cast<size_t>(i) + 2] = found ? best.v : 0.0f;
            prim[i] = best.prim;
            inst[i] = best.inst;
        }
    });
}

void orc_trace_any(const orc_scene* s, const float* org_tmin, const float* dir_tmax, uint32_t n, int use_bvh,
                   int threads, uint8_t* occluded) {
    parallelFor(n, threads, [=](uint32_t begin, uint32_t end) {
        for (uint32_t i = begin; i < end; ++i) {
            const Best best = traceScene(*s, org_tmin + 4 * static_cast<size_t>(i), dir_tmax + 4 * static_cast<size_t>(i),
                                         use_bvh != 0, true);
            occluded[i] = best.prim != 0xffffffffU;
        }
    });
}

void orc_shade_hits(const orc_scene* s, const float* org_tmin, const float* dir_tmax, const float* tuv,
                    const uint32_t* prim, const uint32_t* inst, uint32_t n, rt_hit* hits) {
    for (uint32_t i = 0; i < n; ++i) {
        const size_t k = static_cast<size_t>(i);
        const V3 origin{org_tmin[4 * k], org_tmin[4 * k + 1], org_tmin[4 * k + 2]};
        const V3 ray{dir_tmax[4 * k], dir_tmax[4 * k + 1], dir_tmax[4 * k + 2]};
        shadeOne(*s, origin, ray, tuv[3 * k], tuv[3 * k + 1], tuv[3 * k + 2], prim[i], inst[i], hits[i]);
    }
}

}  // extern "C"
