// bvh.h — host-side 4-wide BVH build (see bvh.cpp).
#pragma once
#include <cmath>
#include <string>
#include <vector>

#include "../../include/slrhip.h"
#include "device_types.h"

namespace slrhip {

struct QBVH {
    std::vector<QNode> nodes;        // breadth-first; nodes[0] is the root
    std::vector<QNodeQ> quantized;   // same nodes, 8-bit child boxes (quantizeNodes); empty unless asked for
    std::vector<QNode8> nodes8;      // the same binary tree collapsed to eight-wide quantized nodes (buildQBVH's wide8 argument); same leaf packets
    uint32_t depth8 = 0;
    std::vector<LeafTri> leafTris;   // leaf packets, contiguous per leaf
    uint32_t depth = 0;              // levels of 4-wide nodes
    uint64_t spatialSplits = 0;      // spatial-split build only: splits in space, and leaf references (>= triangles: duplicates)
    uint64_t references = 0;
};

// ---- binary builders: both produce BNode arrays + a primitive list that buildQBVH collapses to 4-wide nodes --------------------
struct Box {
    float lo[3], hi[3];
    void reset() { for (int a = 0; a < 3; ++a) { lo[a] = INFINITY; hi[a] = -INFINITY; } }
    void grow(const Box& b) { for (int a = 0; a < 3; ++a) { lo[a] = std::fmin(lo[a], b.lo[a]); hi[a] = std::fmax(hi[a], b.hi[a]); } }
    void grow(const float* p) { for (int a = 0; a < 3; ++a) { lo[a] = std::fmin(lo[a], p[a]); hi[a] = std::fmax(hi[a], p[a]); } }
    float area() const {
        float d[3] = {hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2]};
        if (d[0] < 0 || d[1] < 0 || d[2] < 0) return 0.0f;
        return 2.0f * (d[0] * d[1] + d[1] * d[2] + d[2] * d[0]);
    }
};

struct BNode {
    Box box;
    uint32_t left, right;    // children (inner)
    uint32_t first, count;   // primitive range (leaf: count > 0)
};

unsigned hostThreads();

// Binned-SAH binary BVH with SPATIAL SPLITS (Stich et al. 2009; the reference's default accelerator is this kind of tree:
// libSLR/Accelerator/SBVH.h:57-348 with Triangle::choppedBounds / splitBounds, Surface/TriangleMesh.cpp:19-125).  A triangle whose
// box straddles a spatial split plane is referenced from both children with its box clipped to each side, so `prims` may name a
// triangle more than once (at most refBudget x numTris entries).  Leaves hold at most kMaxLeafTris references.
struct SbvhStats { uint64_t spatialSplits = 0, objectSplits = 0, references = 0; uint32_t depth = 0; };
void buildBinarySBVH(const slrhip_vertex* verts, const slrhip_triangle* tris, uint32_t numTris, float refBudget,
                     std::vector<BNode>* nodes, std::vector<uint32_t>* prims, SbvhStats* stats);

// Fills out->quantized from out->nodes (conservative: every dequantized box contains the float box).
void quantizeNodes(QBVH* out);

// The same records built on the GPU (bvh_device.hip): LBVH over 63-bit Morton codes, the same 4-wide collapse, quantized nodes,
// leaf packets and per-triangle shading records.  All pointers are DEVICE memory from hipMalloc, owned by the caller afterwards.
// lightTris = scene indices of the emitting triangles in light-list order (their ShadeTri::light is patched in).
struct DeviceGeometry {
    QNode* nodes = nullptr;
    QNodeQ* nodesQ = nullptr;        // nullptr unless asked for
    LeafTri* leafTris = nullptr;
    ShadeTri* shadeTris = nullptr;
    uint32_t numNodes = 0, numLeafTris = 0, depth = 0;
    double secondsUpload = 0, secondsSort = 0, secondsHierarchy = 0, secondsCollapse = 0, secondsRecords = 0;
};
int buildGeometryDevice(const slrhip_vertex* verts, uint32_t numVerts, const slrhip_triangle* tris, uint32_t numTris, const uint32_t* lightTris,
                        uint32_t numLights, bool wantQuantized, DeviceGeometry* out, std::string* err);

// Returns 0 on success.
int buildQBVH(const slrhip_vertex* verts, const slrhip_triangle* tris, uint32_t numTris, QBVH* out, bool spatialSplits = false, bool wide8 = false);

// Scenes with instanced meshes (slrhip_instance): one tree per distinct mesh in its local space and a top-level tree over the loose
// triangles and the instances' world boxes, all in ONE node array and ONE leaf array (the top level first); an instance is a child
// reference of its own kind (device_types.h).  depth = the top level's + 1 + the deepest mesh's.
int buildInstancedQBVH(const slrhip_vertex* verts, const slrhip_triangle* tris, uint32_t numTris, const slrhip_instance* instances, uint32_t numInstances,
                       QBVH* out, std::vector<DevInstance>* devInstances, std::string* err);

} // namespace slrhip
