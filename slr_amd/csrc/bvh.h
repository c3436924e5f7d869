// bvh.h — host-side 4-wide BVH build (see bvh.cpp).
#pragma once
#include <vector>

#include "../../include/slrhip.h"
#include "device_types.h"

namespace slrhip {

struct QBVH {
    std::vector<QNode> nodes;        // breadth-first; nodes[0] is the root
    std::vector<QNodeQ> quantized;   // same nodes, 8-bit child boxes (quantizeNodes); empty unless asked for
    std::vector<LeafTri> leafTris;   // leaf packets, contiguous per leaf
    uint32_t depth = 0;              // levels of 4-wide nodes
};

// Fills out->quantized from out->nodes (conservative: every dequantized box contains the float box).
void quantizeNodes(QBVH* out);

// Returns 0 on success.
int buildQBVH(const slrhip_vertex* verts, const slrhip_triangle* tris, uint32_t numTris, QBVH* out);

} // namespace slrhip
