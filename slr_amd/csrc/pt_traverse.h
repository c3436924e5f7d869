// pt_traverse.h — one lane = one ray: the closest-hit / any-hit traversal as a device function (gfx950, wave64).
// Used by the 64-ray-batch kernels of pt_trace.hip and by the tail kernel of pt_tail_kernels.h.  Same semantics as
// Scene::intersect / testVisibility (Core/SurfaceObject.cpp:408-430), QBVH::intersect (Accelerator/QBVH.h:295-339) and
// Triangle::intersect (Surface/TriangleMesh.cpp:131-178).
#pragma once
#include <hip/hip_runtime.h>

#include "pt_device.h"
#include "pt_kernels.h"
#include "pt_tex.h"

namespace slrhip {

static const int kTraceBlock = 256;
static const int kLdsStack = 8;
static const int kSpillStack = 56;     // 8 + 56 = the reference's 64-entry stack (QBVH.h:299)
static const int kTopNodes = 128;      // levels 0-3 of the 4-wide tree (1 + 4 + 16 + 64 = 85) and a bit
static const int kTopStride = 9;       // float4 per staged node: 8 + 1 pad -> 144 B, spreads nodes over the LDS banks

struct HitRec {
    uint32_t tri;
    float t, b1, b2;       // Moller-Trumbore's own barycentrics: b0 = 1 - b1 - b2 is re-derived where it is needed (pt_shade_kernels.h)
    int32_t inst;          // INST: the instance the hit went through, -1 = a loose triangle
};
struct TravCount {
    uint32_t nodes, tris;
};

// INST: scenes with instanced meshes (DevScene::instances; see wsConsume in pt_trace_ws.hip for the scheme).  Here the world ray
// stays in registers while the lane is inside an instance.
template <bool ANY_HIT, bool COUNT, bool INST = false>
__device__ __forceinline__ bool traverse(const DevScene& sc, const float4* __restrict__ nodes4, const float4* __restrict__ tris4, const float4* topNodes,
                                         uint32_t numTop, V3 org, V3 dir, float tmin, float tmax, HitRec* hit,
                                         uint32_t* ldsStack /* [kLdsStack][blockDim], this lane's column */, TravCount* cnt,
                                         uint32_t* errorWord = nullptr) {
    float idx = 1.0f / dir.x, idy = 1.0f / dir.y, idz = 1.0f / dir.z;      // Vector3.h:60 reciprocal()
    // float4 index inside a node: 0..2 = min xyz, 3..5 = max xyz, 6 = children
    int nx = idx > 0.0f ? 0 : 3, fx = 3 - nx;
    int ny = idy > 0.0f ? 1 : 4, fy = 5 - ny;
    int nz = idz > 0.0f ? 2 : 5, fz = 7 - nz;
    const V3 worldOrg = org, worldDir = dir;
    int32_t inst = -1;
    hit->inst = -1;

    uint32_t spill[kSpillStack];
    int sp = 0;
    uint32_t cur = 0;                      // root
    bool found = false;
    hit->tri = 0xFFFFFFFFu; hit->t = INFINITY; hit->b1 = 0.0f; hit->b2 = 0.0f;

    for (;;) {
        if (INST && (cur & kLeafFlag) && (((cur >> kLeafCountShift) & 0xFu) == 0u || cur == kPopInstance)) {
            const bool leaving = cur == kPopInstance;
            if (leaving) { org = worldOrg; dir = worldDir; inst = -1; }
            else {
                // TransformedSurfaceObject::intersect (Core/SurfaceObject.cpp:307-317): invert(sampledTF) * ray
                const uint32_t k = cur & kLeafIndexMask;
                const float4* rec = sc.instances + (size_t)k * 9u + 4u;
                const float4 c0 = rec[0], c1 = rec[1], c2 = rec[2], c3 = rec[3], meta = rec[4];
                org = V3(c0.x * worldOrg.x + c1.x * worldOrg.y + c2.x * worldOrg.z + c3.x * 1.0f,
                         c0.y * worldOrg.x + c1.y * worldOrg.y + c2.y * worldOrg.z + c3.y * 1.0f,
                         c0.z * worldOrg.x + c1.z * worldOrg.y + c2.z * worldOrg.z + c3.z * 1.0f);
                dir = V3(c0.x * worldDir.x + c1.x * worldDir.y + c2.x * worldDir.z, c0.y * worldDir.x + c1.y * worldDir.y + c2.y * worldDir.z,
                         c0.z * worldDir.x + c1.z * worldDir.y + c2.z * worldDir.z);
                inst = (int32_t)k;
                if (sp < kLdsStack) ldsStack[sp * kTraceBlock] = kPopInstance;
                else if (sp < kLdsStack + kSpillStack) spill[sp - kLdsStack] = kPopInstance;
                else { if (errorWord) atomicOr(errorWord, ERR_STACK_OVERFLOW); --sp; }
                ++sp;
                cur = __float_as_uint(meta.x);      // the root of the mesh's tree
            }
            idx = 1.0f / dir.x; idy = 1.0f / dir.y; idz = 1.0f / dir.z;
            nx = idx > 0.0f ? 0 : 3; fx = 3 - nx;
            ny = idy > 0.0f ? 1 : 4; fy = 5 - ny;
            nz = idz > 0.0f ? 2 : 5; fz = 7 - nz;
            if (!leaving) continue;
            if (sp == 0) break;
            --sp;
            cur = sp < kLdsStack ? ldsStack[sp * kTraceBlock] : spill[sp - kLdsStack];
            continue;
        }
        if (cur & kLeafFlag) {
            const uint32_t first = cur & kLeafIndexMask;
            const uint32_t count = (cur >> kLeafCountShift) & 0xF;
            for (uint32_t k = 0; k < count; ++k) {
                const float4* tp = tris4 + (size_t)(first + k) * 3;
                const float4 a = tp[0], b = tp[1], c = tp[2];
                const V3 v0(a.x, a.y, a.z), e1(b.x, b.y, b.z), e2(c.x, c.y, c.z);
                const uint32_t triIdx = __float_as_uint(a.w);
                if (COUNT) ++cnt->tris;
                // Moller-Trumbore exactly as TriangleMesh.cpp:139-160
                V3 p = cross(dir, e2);
                float det = dot(e1, p);
                if (det == 0.0f) continue;
                float invDet = 1.0f / det;
                V3 d = org - v0;
                float b1 = dot(d, p) * invDet;
                if (b1 < 0.0f || b1 > 1.0f) continue;
                V3 q = cross(d, e1);
                float b2 = dot(dir, q) * invDet;
                if (b2 < 0.0f || b1 + b2 > 1.0f) continue;
                float tt = dot(e2, q) * invDet;
                if (tt < tmin || tt > tmax) continue;
                // alpha texture of the triangle (TriangleMesh.cpp:162-167); LeafTri::alpha rides in e1's fourth word
                if (__float_as_uint(b.w) != kNoAlpha && !alphaPasses(sc.alphaTris, sc.textures, __float_as_uint(b.w), b1, b2)) continue;
                if (ANY_HIT) return true;
                // equal distance: the larger scene index wins (tree-independent tie rule)
                if (tt == tmax && found && (INST ? (inst < hit->inst || (inst == hit->inst && triIdx < hit->tri)) : triIdx < hit->tri)) continue;
                found = true;
                tmax = tt;                                  // ray.distMax = isect->dist (QBVH.h:335)
                hit->tri = triIdx;
                hit->inst = inst;
                hit->t = tt;
                hit->b1 = b1;                               // Intersection::u = 1 - b1 - b2, ::v = b1 (TriangleMesh.cpp:159,172-173)
                hit->b2 = b2;
            }
            if (sp == 0) break;
            --sp;
            cur = sp < kLdsStack ? ldsStack[sp * kTraceBlock] : spill[sp - kLdsStack];
            continue;
        }

        if (COUNT) ++cnt->nodes;
        float4 nX, nY, nZ, fX, fY, fZ, ch;
        if (cur < numTop) {
            const float4* n = topNodes + cur * kTopStride;
            nX = n[nx]; nY = n[ny]; nZ = n[nz]; fX = n[fx]; fY = n[fy]; fZ = n[fz]; ch = n[6];
        }
        else {
            const float4* n = nodes4 + (size_t)cur * 8;
            nX = n[nx]; nY = n[ny]; nZ = n[nz]; fX = n[fx]; fY = n[fy]; fZ = n[fz]; ch = n[6];
        }
        // slab test of QBVH::Node::intersect (QBVH.h:55-76): tNear <= tFar
        const float tn0 = fmaxf(fmaxf((nX.x - org.x) * idx, (nY.x - org.y) * idy), fmaxf((nZ.x - org.z) * idz, tmin));
        const float tn1 = fmaxf(fmaxf((nX.y - org.x) * idx, (nY.y - org.y) * idy), fmaxf((nZ.y - org.z) * idz, tmin));
        const float tn2 = fmaxf(fmaxf((nX.z - org.x) * idx, (nY.z - org.y) * idy), fmaxf((nZ.z - org.z) * idz, tmin));
        const float tn3 = fmaxf(fmaxf((nX.w - org.x) * idx, (nY.w - org.y) * idy), fmaxf((nZ.w - org.z) * idz, tmin));
        const float tf0 = fminf(fminf((fX.x - org.x) * idx, (fY.x - org.y) * idy), fminf((fZ.x - org.z) * idz, tmax));
        const float tf1 = fminf(fminf((fX.y - org.x) * idx, (fY.y - org.y) * idy), fminf((fZ.y - org.z) * idz, tmax));
        const float tf2 = fminf(fminf((fX.z - org.x) * idx, (fY.z - org.y) * idy), fminf((fZ.z - org.z) * idz, tmax));
        const float tf3 = fminf(fminf((fX.w - org.x) * idx, (fY.w - org.y) * idy), fminf((fZ.w - org.z) * idz, tmax));
        const uint32_t c0 = __float_as_uint(ch.x), c1 = __float_as_uint(ch.y), c2 = __float_as_uint(ch.z), c3 = __float_as_uint(ch.w);
        const bool h0 = tn0 <= tf0 && c0 != kInvalidChild;
        const bool h1 = tn1 <= tf1 && c1 != kInvalidChild;
        const bool h2 = tn2 <= tf2 && c2 != kInvalidChild;
        const bool h3 = tn3 <= tf3 && c3 != kInvalidChild;

        // nearest hit child is visited next; the rest go on the stack
        float best = INFINITY;
        uint32_t next = kInvalidChild;
        if (h0) { best = tn0; next = c0; }
        if (h1 && tn1 < best) { best = tn1; next = c1; }
        if (h2 && tn2 < best) { best = tn2; next = c2; }
        if (h3 && tn3 < best) { best = tn3; next = c3; }
#define SLR_PUSH(cond, ref)                                                              \
        if ((cond) && (ref) != next) {                                                   \
            if (sp < kLdsStack) ldsStack[sp * kTraceBlock] = (ref);                      \
            else if (sp < kLdsStack + kSpillStack) spill[sp - kLdsStack] = (ref);        \
            else { if (errorWord) atomicOr(errorWord, ERR_STACK_OVERFLOW); --sp; }       \
            ++sp;                                                                        \
        }
        SLR_PUSH(h0, c0)
        SLR_PUSH(h1, c1)
        SLR_PUSH(h2, c2)
        SLR_PUSH(h3, c3)
#undef SLR_PUSH
        if (next != kInvalidChild) { cur = next; continue; }
        if (sp == 0) break;
        --sp;
        cur = sp < kLdsStack ? ldsStack[sp * kTraceBlock] : spill[sp - kLdsStack];
    }
    return found;
}

struct TraceLds {
    float4 top[kTopNodes * kTopStride];
    uint32_t stack[kLdsStack * kTraceBlock];
};

// Stage the top of the tree: coalesced 16-byte loads, padded rows in LDS.
__device__ __forceinline__ uint32_t stageTopNodes(const DevScene& sc, TraceLds& lds) {
    const uint32_t numTop = min(sc.numNodes, (uint32_t)kTopNodes);
    for (uint32_t i = threadIdx.x; i < numTop * 8; i += kTraceBlock) lds.top[(i >> 3) * kTopStride + (i & 7)] = sc.nodes[i];
    __syncthreads();
    return numTop;
}

} // namespace slrhip
