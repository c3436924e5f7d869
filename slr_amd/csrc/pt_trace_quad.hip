// pt_trace_quad.hip — the lane-mapping experiment SURVEY §7 names: FOUR LANES = ONE RAY (one child box / one leaf triangle per
// lane, DPP min / ballot to order the children, 16 rays per wave) against ONE LANE = ONE RAY (pt_trace.hip's batch kernel).
// Same closest-hit semantics (Scene::intersect, SurfaceObject.cpp:408-416; QBVH::intersect, QBVH.h:295-339;
// Triangle::intersect, TriangleMesh.cpp:131-178; the tree-independent tie rule of DESIGN.md), so both mappings must return the
// same hits bit for bit — which the GPU tests check — and slrhip_trace_rays_timed measures them on the same rays.
//
// Layouts made for this mapping (built at the first call from the uploaded tree):
//   QNode4     128 B: lo[4], hi[4]; lo[k] = (minx, miny, minz, maxx) of child k, hi[k] = (maxy, maxz, child ref, -).  The four
//              lanes of a ray read lo[k] in ONE 64-byte request and hi[k] in another: 2 L1 look-ups per node visit instead of
//              the 7 scattered 16-byte loads of the lane-per-ray kernels.
//   LeafPacket 192 B: v0[4], e1[4], e2[4] of a leaf's (up to 4) triangles: lane k tests triangle k, 3 look-ups per LEAF instead of
//              3 per triangle.
// The whole 64-entry stack (QBVH.h:299) of a ray lives in LDS ([entry][ray]): 16 rays per wave need 4 KiB.
#include <hip/hip_runtime.h>

#include <vector>

#include "pt_device.h"
#include "pt_kernels.h"
#include "pt_tex.h"

namespace slrhip {

namespace {

__device__ __forceinline__ int dppXor1(int v) { return __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xF, 0xF, false); }   // quad_perm [1,0,3,2]
__device__ __forceinline__ int dppXor2(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xF, 0xF, false); }   // quad_perm [2,3,0,1]
__device__ __forceinline__ float quadMin(float v) {
    v = fminf(v, __int_as_float(dppXor1(__float_as_int(v))));
    return fminf(v, __int_as_float(dppXor2(__float_as_int(v))));
}
__device__ __forceinline__ uint32_t quadMaxU(uint32_t v) {
    v = max(v, (uint32_t)dppXor1((int)v));
    return max(v, (uint32_t)dppXor2((int)v));
}
__device__ __forceinline__ uint32_t quadOr(uint32_t v) {
    v |= (uint32_t)dppXor1((int)v);
    return v | (uint32_t)dppXor2((int)v);
}

static const int kQuadBlock = 256;

__global__ __launch_bounds__(kQuadBlock) void k_trace_quad(DevScene sc, const float4* __restrict__ nodes, const float4* __restrict__ packets,
                                                           const float4* __restrict__ org, const float4* __restrict__ dir,
                                                           float4* __restrict__ out, uint32_t n) {
    __shared__ uint32_t stackLds[kQuadBlock / 64][64][16];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t k = lane & 3u, q = lane >> 2, quadShift = lane & ~3u;
    uint32_t* stack = &stackLds[wave][0][q];                       // entry e of this ray at stack[e * 16]
    const uint32_t kbit = 1u << k, below = kbit - 1u;

    for (uint32_t base = (blockIdx.x * (kQuadBlock / 64) + wave) * 16u; base < n; base += gridDim.x * (kQuadBlock / 64) * 16u) {
        const uint32_t ray = base + q;
        bool done = ray >= n;
        const float4 o = done ? make_float4(0, 0, 0, 0) : org[ray];
        const float4 d = done ? make_float4(1, 1, 1, 1) : dir[ray];
        const float ox = o.x, oy = o.y, oz = o.z, tmin = o.w, dx = d.x, dy = d.y, dz = d.z;
        float tmax = d.w;
        const float idx = 1.0f / dx, idy = 1.0f / dy, idz = 1.0f / dz;      // Vector3.h:60 reciprocal()
        uint32_t cur = 0;
        int sp = 0;
        uint32_t hitTri = 0xFFFFFFFFu;
        float hitT = INFINITY, hitB1 = 0.0f, hitB2 = 0.0f;

        while (__ballot(!done)) {
            if (!done && !(cur & kLeafFlag)) {
                const float4 lo = nodes[(size_t)cur * 8 + k], hi = nodes[(size_t)cur * 8 + 4 + k];
                const uint32_t child = __float_as_uint(hi.z);
                // near / far planes of THIS child (QBVH.h:66-71), then the slab test of QBVH::Node::intersect (:55-76), one child per lane
                const float nx = idx > 0.0f ? lo.x : lo.w, fx = idx > 0.0f ? lo.w : lo.x;
                const float ny = idy > 0.0f ? lo.y : hi.x, fy = idy > 0.0f ? hi.x : lo.y;
                const float nz = idz > 0.0f ? lo.z : hi.y, fz = idz > 0.0f ? hi.y : lo.z;
                const float tn = fmaxf(fmaxf((nx - ox) * idx, (ny - oy) * idy), fmaxf((nz - oz) * idz, tmin));
                const float tf = fminf(fminf((fx - ox) * idx, (fy - oy) * idy), fminf((fz - oz) * idz, tmax));
                const bool h = tn <= tf && child != kInvalidChild;
                const float m = quadMin(h ? tn : INFINITY);
                const uint32_t hitBits = (uint32_t)(__ballot(h) >> quadShift) & 0xFu;
                const uint32_t winBits = (uint32_t)(__ballot(h && tn == m) >> quadShift) & 0xFu;
                if (hitBits == 0u) {
                    if (sp == 0) done = true;
                    else { --sp; cur = stack[sp * 16]; }
                }
                else {
                    const uint32_t kn = (uint32_t)__ffs((int)winBits) - 1u;      // nearest child, lowest index among equals
                    const uint32_t next = quadOr(k == kn ? child : 0u);
                    const uint32_t pushBits = hitBits & ~(1u << kn);
                    if (pushBits & kbit) stack[(sp + (int)__popc(pushBits & below)) * 16] = child;     // depth <= 64 entries: checked at upload
                    sp += (int)__popc(pushBits);
                    cur = next;
                }
            }
            if (!done && (cur & kLeafFlag)) {
                const uint32_t p = cur & kLeafIndexMask, count = (cur >> kLeafCountShift) & 0xFu;
                bool accept = false;
                float tt = INFINITY, b1 = 0.0f, b2 = 0.0f;
                uint32_t triIdx = 0u;
                if (k < count) {
                    const float4 a = packets[(size_t)p * 12 + k], b = packets[(size_t)p * 12 + 4 + k], c = packets[(size_t)p * 12 + 8 + k];
                    const V3 v0(a.x, a.y, a.z), e1(b.x, b.y, b.z), e2(c.x, c.y, c.z), o3(ox, oy, oz), d3(dx, dy, dz);
                    triIdx = __float_as_uint(a.w);
                    // Moller-Trumbore exactly as TriangleMesh.cpp:139-160
                    const V3 pv = cross(d3, e2);
                    const float det = dot(e1, pv);
                    accept = det != 0.0f;
                    const float invDet = 1.0f / det;
                    const V3 dd = o3 - v0;
                    b1 = dot(dd, pv) * invDet;
                    accept = accept && !(b1 < 0.0f || b1 > 1.0f);
                    const V3 qv = cross(dd, e1);
                    b2 = dot(d3, qv) * invDet;
                    accept = accept && !(b2 < 0.0f || b1 + b2 > 1.0f);
                    tt = dot(e2, qv) * invDet;
                    accept = accept && !(tt < tmin || tt > tmax);
                    if (accept && __float_as_uint(b.w) != kNoAlpha) accept = alphaPasses(sc.alphaTris, sc.textures, __float_as_uint(b.w), b1, b2);
                }
                // the leaf's best candidate: smallest t; among equal t the larger scene index (tie rule)
                const float m = quadMin(accept ? tt : INFINITY);
                const bool cand = accept && tt == m;
                const uint32_t winKey = quadMaxU(cand ? triIdx + 1u : 0u);
                if (winKey != 0u) {
                    const uint32_t winTri = winKey - 1u;
                    // the sequential rule (one triangle after the other with a shrinking distMax): a hit AT the current
                    // distance replaces the current one only if its index is larger
                    if (!(m == tmax && hitTri != 0xFFFFFFFFu && winTri < hitTri)) {
                        const bool win = cand && triIdx == winTri;
                        hitB1 = __uint_as_float(quadOr(win ? __float_as_uint(b1) : 0u));      // Intersection::u = 1 - b1 - b2, ::v = b1
                        hitB2 = __uint_as_float(quadOr(win ? __float_as_uint(b2) : 0u));
                        tmax = m;                                                                       // ray.distMax = isect->dist (QBVH.h:335)
                        hitT = m;
                        hitTri = winTri;
                    }
                }
                if (sp == 0) done = true;
                else { --sp; cur = stack[sp * 16]; }
            }
        }
        if (ray < n && k == 0u) out[ray] = make_float4(__uint_as_float(hitTri), hitT, hitB1, hitB2);
    }
}

} // namespace

static inline float hostBitsToFloat(uint32_t u) { float f; __builtin_memcpy(&f, &u, 4); return f; }

// Host side: the quad layouts from the uploaded tree (QNode array + LeafTri array as uploaded by slrhip_upload_scene).
void buildQuadLayouts(const std::vector<QNode>& nodes, const std::vector<LeafTri>& leafTris, std::vector<float4>* nodes4, std::vector<float4>* packets) {
    nodes4->assign(nodes.size() * 8, make_float4(0, 0, 0, 0));
    packets->clear();
    for (size_t i = 0; i < nodes.size(); ++i) {
        const QNode& nd = nodes[i];
        for (int k = 0; k < 4; ++k) {
            uint32_t child = nd.child[k];
            if (child != kInvalidChild && (child & kLeafFlag)) {
                const uint32_t first = child & kLeafIndexMask, count = (child >> kLeafCountShift) & 0xFu;
                const uint32_t p = (uint32_t)(packets->size() / 12);
                packets->resize(packets->size() + 12, make_float4(0, 0, 0, 0));
                for (uint32_t t = 0; t < count && t < 4; ++t) {
                    const LeafTri& lt = leafTris[first + t];
                    (*packets)[(size_t)p * 12 + t] = make_float4(lt.v0[0], lt.v0[1], lt.v0[2], hostBitsToFloat(lt.tri));
                    (*packets)[(size_t)p * 12 + 4 + t] = make_float4(lt.e1[0], lt.e1[1], lt.e1[2], hostBitsToFloat(lt.alpha));
                    (*packets)[(size_t)p * 12 + 8 + t] = make_float4(lt.e2[0], lt.e2[1], lt.e2[2], 0.0f);
                }
                child = kLeafFlag | (count << kLeafCountShift) | p;
            }
            (*nodes4)[i * 8 + k] = make_float4(nd.minx[k], nd.miny[k], nd.minz[k], nd.maxx[k]);
            (*nodes4)[i * 8 + 4 + k] = make_float4(nd.maxy[k], nd.maxz[k], hostBitsToFloat(child), 0.0f);
        }
    }
}

void launchTraceQuad(const DevScene& sc, const float4* nodes4, const float4* packets, const float4* org, const float4* dir, float4* out, uint32_t n, hipStream_t stream) {
    uint32_t blocks = (n + 63) / 64;             // 16 rays per wave, 4 waves per block
    if (blocks > 2048) blocks = 2048;            // 8 resident blocks per CU (16 KiB of LDS each), striding over the rays
    hipLaunchKernelGGL(k_trace_quad, dim3(blocks), dim3(kQuadBlock), 0, stream, sc, nodes4, packets, org, dir, out, n);
}

} // namespace slrhip
