// Acceleration-structure layout shared by the builder and the traversal kernels.
//
// Two levels, as the reference's IAS over GAS (renderer.cpp:343-434): a
// top-level tree over instances and one bottom-level tree per geometry, both
// made of the same 8-wide node with child boxes quantised to 8 bits per bound.
#pragma once

#include "common.h"

#include <atomic>
#include <mutex>
#include <vector>

namespace rt {

constexpr int kWide = 8;          // children per node
constexpr int kFeedStride = 32;   // words between the cursors of a persistent ray feed
#ifndef RT_FEED_SPLIT
#define RT_FEED_SPLIT 4
#endif
constexpr int kFeedSplit = RT_FEED_SPLIT;  // cursors (and chunks of the index range) per XCD
constexpr int kFeedChunks = 8 * kFeedSplit;
constexpr int kFeedWords = kFeedChunks * kFeedStride;
constexpr int kFeedSlots = 16;    // cursor sets a scene keeps for overlapping batched queries
#ifndef RT_MAX_LEAF
#define RT_MAX_LEAF 2
#endif
constexpr int kMaxLeaf = RT_MAX_LEAF;  // triangles per leaf child (the count field of a meta byte holds up to 7,
                                         // the offset field 31: eight leaves of four)
constexpr uint32_t kNoHit = 0xffffffffU;

// 80 bytes, five 16-byte loads per visit.
//   child box = origin + q * 2^(exp - 127) per axis, lo rounded down, hi up.
//   imask bit i set  : child i is a node, index childBase + meta[i]
//   imask bit i clear: meta[i] >> 5 = item count (0 = empty slot),
//                      meta[i] & 31 = first item relative to primBase
//   items are triangles in a bottom-level tree, instances in the top level.
//   An empty slot carries the inverted box qlo = 255, qhi = 0 on every axis, which the slab test of the
//   traversal never enters, so the test needs no look at the slot's meta byte.  (Where the node is so
//   small against its distance from the ray that 255 steps vanish in the rounding of the plane distances,
//   the slot can pass the test; its reference then is a leaf of no items, which the walk drops.)
struct alignas(16) WideNode {
    float origin[3];
    uint8_t exp[3];
    uint8_t imask;
    uint32_t childBase;
    uint32_t primBase;
    uint8_t meta[kWide];
    uint8_t qlo[3][kWide];
    uint8_t qhi[3][kWide];
};
static_assert(sizeof(WideNode) == 80, "wide node is 80 bytes");

// A triangle as the traversal reads it: three 16-byte loads.
struct alignas(16) PackedTriangle {
    float v0[3];
    uint32_t primitive;  // geometry-local triangle index
    float v1[3];
    uint32_t pad1;
    float v2[3];
    uint32_t pad2;
};
static_assert(sizeof(PackedTriangle) == 48, "packed triangle is 48 bytes");

// A top-level leaf item is an instance or a single triangle in world space (a folded triangle): an instance
// whose transform is the identity, bit for bit, and whose mesh is small costs a ray more to enter - transform,
// shear constants, the mesh's one or two nodes - than its triangles cost to test, so rt_scene_build puts such
// triangles into the top-level tree themselves.  The identity transform leaves every coordinate as it is, so the
// intersection arithmetic and its results are those of the instance.  Both kinds of item are 80-byte records of
// one array; a folded triangle is a PackedTriangle (pad1 = its instance's id) in the first 48 bytes, and the leaf
// reference that leads to it carries kFoldedBit.
constexpr uint32_t kFoldedBit = 0x40000000U;

struct alignas(16) InstanceRecord {
    float toObject[12];           // world->object, row-major 3x4
    const WideNode* nodes;        // bottom-level root
    const PackedTriangle* triangles;
    uint32_t instanceId;
    uint32_t geometry;
    uint32_t nodeCount;           // entries of `nodes` / `triangles`: the traversal checks references against them
    uint32_t triangleCount;
};
static_assert(sizeof(InstanceRecord) == 80, "instance record is 80 bytes");

// Shading-side record of an instance, indexed by instance id.
struct alignas(16) ShadeInstance {
    float toObject[12];
    float toWorld[12];
    uint32_t geometry;
    // the geometry's record (rt_scene_set_geometry), kept here as well so that a hit finds its
    // triangle with one dependent load less
    uint32_t material;
    const rt_gpu_vertex* vertices;
    const rt_uint3* indices;
    uint32_t pad[2];
};
static_assert(sizeof(ShadeInstance) == 128, "shade instance is 128 bytes");

struct Aabb {
    float lo[3];
    float hi[3];
};

}  // namespace rt

namespace rt {
// Entries the traversal stack holds per lane (the first ones in LDS, the rest in private memory).
constexpr uint32_t kTraversalStackEntries = 128;
// Every visited node leaves at most kWide - 1 entries behind (two entries of slack on top); a tree
// deeper than this would overflow the stack, and entries that do not fit are dropped, i.e. hits
// would be missed without any error.
inline bool fitsTraversalStack(uint32_t topDepth, uint32_t bottomDepth) {
    return static_cast<uint64_t>(kWide - 1) * (static_cast<uint64_t>(topDepth) + bottomDepth) + 2U <= kTraversalStackEntries;
}
}  // namespace rt

struct rt_blas {
    int device = 0;
    rt::WideNode* nodes = nullptr;
    rt::PackedTriangle* triangles = nullptr;
    uint64_t nodeCount = 0;
    uint64_t triangleCount = 0;
    uint32_t depth = 0;
    rt::Aabb bounds{};            // object-space root bounds (host copy)
    float buildMilliseconds = 0.0f;
};

struct rt_scene {
    int device = 0;
    rt::WideNode* nodes = nullptr;          // top-level tree
    rt::InstanceRecord* instances = nullptr;
    uint64_t nodeCount = 0;
    uint32_t instanceCount = 0;
    uint32_t itemCount = 0;                 // records behind `instances`: instances that are entered + folded triangles
    uint32_t foldedTriangles = 0, foldedInstances = 0;
    uint32_t depth = 0;
    uint64_t blasNodes = 0, blasNodeBytes = 0, triangles = 0, triangleBytes = 0;
    uint32_t blasDepth = 0;
    rt::ShadeInstance* shadeInstances = nullptr;  // indexed by instance id
    std::vector<rt::ShadeInstance> shadeHost;     // host copy (rt_scene_set_geometry completes it)
    uint32_t shadeInstanceCount = 0;
    rt_hit_group_data* hitGroups = nullptr;       // one record per geometry (rt_scene_set_geometry)
    uint32_t hitGroupCount = 0;
    std::vector<float> inverses;            // 12 floats per instance (host copy)
    float buildMilliseconds = 0.0f;
    // kFeedSlots sets of ray-feed cursors for the batched queries.  A set is handed to one launch at a
    // time: the event of a slot is recorded behind the kernel that uses it, and the next launch that is
    // dealt the slot makes its stream wait for that event before it zeroes the cursors, so launches on
    // different streams never share a set however many are in flight.
    uint32_t* feedCursors = nullptr;
    rt::Aabb bounds{};                       // world-space box of everything in the scene (host copy)
    // Workspace of the ordering step of large batched queries (see orderRays in traverse.hip): grown on demand,
    // handed to one launch at a time under feedMutex, the next user's stream waits for orderEvent.
    mutable void* orderData = nullptr;
    mutable size_t orderBytes = 0;
    mutable hipEvent_t orderEvent = nullptr;
    mutable bool orderEventRecorded = false;
    mutable hipEvent_t feedEvents[rt::kFeedSlots] = {};
    mutable bool feedEventRecorded[rt::kFeedSlots] = {};
    mutable uint32_t feedNext = 0;
    mutable std::mutex feedMutex;
};
