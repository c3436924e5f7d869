// pt_kernels.hip — wavefront path-tracing kernels for MI355X (gfx950, wave64).
//
// One path SLOT per (pixel, sample stripe).  A slot carries one light path at a time through
// the reference's bounce loop (Renderers/PathTracingRenderer.cpp:137-262); when the path ends
// the slot adds weight*C to its pixel accumulator in pass order (the Kahan sum of
// RGBStorage::add, RGBTypes.h:176-179) and regenerates the next sample of the same pixel in
// the same kernel, so every live slot emits exactly one extension ray per iteration and the
// framebuffer needs no atomics.  One wavefront iteration =
//     trace_closest (extension-ray queue)  ->  trace_shadow (shadow-ray queue)  ->  shade.
// Queues are slot-index lists in HBM built with wave ballots + one atomic per wave; all path
// state is SoA (16-byte records) so a wave's loads are 1 KiB coalesced bursts.
#include <hip/hip_runtime.h>

#include "pt_device.h"
#include "pt_kernels.h"

namespace slrhip {

// =============================================================================================
// Traversal.  Replaces Scene::intersect -> SurfaceObjectAggregate::intersect -> QBVH::intersect
// (Core/SurfaceObject.cpp:408-416,267-269; Accelerator/QBVH.h:295-339) and Triangle::intersect
// (Surface/TriangleMesh.cpp:131-178).  One lane = one ray; four child slabs per node from six
// 16-byte loads whose near/far selection (QBVH.h:66-71: invRayDir > 0 ? min : max) is folded
// into per-ray load offsets; the nearest hit child is descended into directly, the others are
// pushed on a per-lane stack that lives in LDS ([entry][lane], conflict-free) and spills to
// scratch beyond kLdsStack entries.
// =============================================================================================
static const int kTraceBlock = 256;
static const int kLdsStack = 16;
static const int kSpillStack = 48;     // 16 + 48 = the reference's 64-entry stack (QBVH.h:299)

struct HitRec {
    uint32_t tri;
    float t, b0, b1;
};

struct TravCount {
    uint32_t nodes, tris;
};

template <bool ANY_HIT, bool COUNT>
__device__ __forceinline__ bool traverse(const float4* __restrict__ nodes4, const float4* __restrict__ tris4, V3 org, V3 dir, float tmin,
                                         float tmax, HitRec* hit, uint32_t* ldsStack /* [kLdsStack][blockDim] at this lane */,
                                         uint32_t ldsStride, TravCount* cnt) {
    const float idx = 1.0f / dir.x, idy = 1.0f / dir.y, idz = 1.0f / dir.z;      // Vector3.h:60 reciprocal()
    // float4 index inside a node: 0..2 = min xyz, 3..5 = max xyz, 6 = children
    const int nx = idx > 0.0f ? 0 : 3, fx = 3 - nx;
    const int ny = idy > 0.0f ? 1 : 4, fy = 5 - ny;
    const int nz = idz > 0.0f ? 2 : 5, fz = 7 - nz;

    uint32_t spill[kSpillStack];
    int sp = 0;
    uint32_t cur = 0;                      // root
    bool found = false;
    hit->tri = 0xFFFFFFFFu; hit->t = INFINITY; hit->b0 = 0.0f; hit->b1 = 0.0f;

    for (;;) {
        if (cur & kLeafFlag) {
            uint32_t first = cur & kLeafIndexMask;
            uint32_t count = (cur >> kLeafCountShift) & 0xF;
            for (uint32_t k = 0; k < count; ++k) {
                const float4* tp = tris4 + (size_t)(first + k) * 3;
                float4 a = tp[0], b = tp[1], c = tp[2];
                V3 v0(a.x, a.y, a.z), e1(b.x, b.y, b.z), e2(c.x, c.y, c.z);
                uint32_t triIdx = __float_as_uint(a.w);
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
                if (ANY_HIT) return true;
                // equal distance: the larger scene index wins (tree-independent tie rule)
                if (tt == tmax && found && triIdx < hit->tri) continue;
                found = true;
                tmax = tt;                                  // ray.distMax = isect->dist (QBVH.h:335)
                hit->tri = triIdx;
                hit->t = tt;
                hit->b0 = 1.0f - b1 - b2;                   // TriangleMesh.cpp:162,172-173
                hit->b1 = b1;
            }
            if (sp == 0) break;
            --sp;
            cur = sp < kLdsStack ? ldsStack[sp * ldsStride] : spill[sp - kLdsStack];
            continue;
        }

        if (COUNT) ++cnt->nodes;
        const float4* n = nodes4 + (size_t)cur * 8;
        const float4 nX = n[nx], nY = n[ny], nZ = n[nz];
        const float4 fX = n[fx], fY = n[fy], fZ = n[fz];
        const float4 ch = n[6];
        // slab test of QBVH::Node::intersect (QBVH.h:55-76): tNear <= tFar
        float tn0 = fmaxf(fmaxf((nX.x - org.x) * idx, (nY.x - org.y) * idy), fmaxf((nZ.x - org.z) * idz, tmin));
        float tn1 = fmaxf(fmaxf((nX.y - org.x) * idx, (nY.y - org.y) * idy), fmaxf((nZ.y - org.z) * idz, tmin));
        float tn2 = fmaxf(fmaxf((nX.z - org.x) * idx, (nY.z - org.y) * idy), fmaxf((nZ.z - org.z) * idz, tmin));
        float tn3 = fmaxf(fmaxf((nX.w - org.x) * idx, (nY.w - org.y) * idy), fmaxf((nZ.w - org.z) * idz, tmin));
        float tf0 = fminf(fminf((fX.x - org.x) * idx, (fY.x - org.y) * idy), fminf((fZ.x - org.z) * idz, tmax));
        float tf1 = fminf(fminf((fX.y - org.x) * idx, (fY.y - org.y) * idy), fminf((fZ.y - org.z) * idz, tmax));
        float tf2 = fminf(fminf((fX.z - org.x) * idx, (fY.z - org.y) * idy), fminf((fZ.z - org.z) * idz, tmax));
        float tf3 = fminf(fminf((fX.w - org.x) * idx, (fY.w - org.y) * idy), fminf((fZ.w - org.z) * idz, tmax));
        const uint32_t c0 = __float_as_uint(ch.x), c1 = __float_as_uint(ch.y), c2 = __float_as_uint(ch.z), c3 = __float_as_uint(ch.w);
        bool h0 = tn0 <= tf0 && c0 != kInvalidChild;
        bool h1 = tn1 <= tf1 && c1 != kInvalidChild;
        bool h2 = tn2 <= tf2 && c2 != kInvalidChild;
        bool h3 = tn3 <= tf3 && c3 != kInvalidChild;

        // nearest hit child is visited next; the rest go on the stack
        float best = INFINITY;
        uint32_t next = kInvalidChild;
        if (h0) { best = tn0; next = c0; }
        if (h1 && tn1 < best) { best = tn1; next = c1; }
        if (h2 && tn2 < best) { best = tn2; next = c2; }
        if (h3 && tn3 < best) { best = tn3; next = c3; }
#define SLR_PUSH(cond, ref)                                                              \
        if ((cond) && (ref) != next) {                                                   \
            if (sp < kLdsStack) ldsStack[sp * ldsStride] = (ref);                        \
            else if (sp < kLdsStack + kSpillStack) spill[sp - kLdsStack] = (ref);        \
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
        cur = sp < kLdsStack ? ldsStack[sp * ldsStride] : spill[sp - kLdsStack];
    }
    return found;
}

// One 64-bit atomic per wave for the instrumented (COUNT) builds.
__device__ __forceinline__ void waveAdd(uint64_t* dst, uint32_t v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    if ((threadIdx.x & 63u) == 0 && v) atomicAdd((unsigned long long*)dst, (unsigned long long)v);
}

// Extension rays: closest hit.  Grid-stride over the queue the previous shade pass built.
template <bool COUNT>
__global__ __launch_bounds__(kTraceBlock) void k_trace_closest(DevScene sc, PathBuffers pb, uint32_t parity) {
    __shared__ uint32_t stack[kLdsStack * kTraceBlock];
    const uint32_t n = pb.queueCount[parity * 2 + 0];
    TravCount cnt = {0, 0};
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        // fold this iteration's queue sizes into the totals and clear the counters the next shade pass fills
        pb.totals[0] += n;
        pb.totals[1] += pb.queueCount[parity * 2 + 1];
        pb.queueCount[(parity ^ 1) * 2 + 0] = 0;
        pb.queueCount[(parity ^ 1) * 2 + 1] = 0;
    }
    for (uint32_t i = blockIdx.x * kTraceBlock + threadIdx.x; i < n; i += gridDim.x * kTraceBlock) {
        const uint32_t slot = pb.extQueue[i];
        const float4 o = pb.rayOrg[slot];
        const float4 d = pb.rayDir[slot];
        HitRec hit;
        traverse<false, COUNT>(sc.nodes, sc.leafTris, V3(o.x, o.y, o.z), V3(d.x, d.y, d.z), o.w, d.w, &hit, stack + threadIdx.x, kTraceBlock, &cnt);
        pb.hit[slot] = make_float4(__uint_as_float(hit.tri), hit.t, hit.b0, hit.b1);
    }
    if (COUNT) { waveAdd(&pb.totals[4], cnt.nodes); waveAdd(&pb.totals[5], cnt.tris); }
}

// Shadow rays: Scene::testVisibility (SurfaceObject.cpp:418-430) = "no hit in [eps, d(1-eps)]".
template <bool COUNT>
__global__ __launch_bounds__(kTraceBlock) void k_trace_shadow(DevScene sc, PathBuffers pb, uint32_t parity) {
    __shared__ uint32_t stack[kLdsStack * kTraceBlock];
    const uint32_t n = pb.queueCount[parity * 2 + 1];
    TravCount cnt = {0, 0};
    for (uint32_t i = blockIdx.x * kTraceBlock + threadIdx.x; i < n; i += gridDim.x * kTraceBlock) {
        const uint32_t slot = pb.shadowQueue[i];
        const float4 o = pb.rayOrg[slot];
        const float4 d = pb.shadowDir[slot];
        HitRec hit;
        bool occluded = traverse<true, COUNT>(sc.nodes, sc.leafTris, V3(o.x, o.y, o.z), V3(d.x, d.y, d.z), kRayEpsilon, d.w, &hit,
                                              stack + threadIdx.x, kTraceBlock, &cnt);
        pb.visible[slot] = occluded ? 0u : 1u;
    }
    if (COUNT) { waveAdd(&pb.totals[6], cnt.nodes); waveAdd(&pb.totals[7], cnt.tris); }
}

// Stand-alone closest-hit batch (parity tests of the traversal alone; not on the render path).
__global__ __launch_bounds__(kTraceBlock) void k_trace_batch(DevScene sc, const float4* org, const float4* dir, float4* out, uint32_t n) {
    __shared__ uint32_t stack[kLdsStack * kTraceBlock];
    for (uint32_t i = blockIdx.x * kTraceBlock + threadIdx.x; i < n; i += gridDim.x * kTraceBlock) {
        const float4 o = org[i], d = dir[i];
        HitRec hit;
        TravCount cnt = {0, 0};
        bool found = traverse<false, false>(sc.nodes, sc.leafTris, V3(o.x, o.y, o.z), V3(d.x, d.y, d.z), o.w, d.w, &hit, stack + threadIdx.x, kTraceBlock, &cnt);
        out[i] = found ? make_float4(__uint_as_float(hit.tri), hit.t, hit.b0, hit.b1) : make_float4(__uint_as_float(0xFFFFFFFFu), INFINITY, 0.f, 0.f);
    }
}

// =============================================================================================
// Shading.
// =============================================================================================
struct SurfPt {       // Core/geometry.h:239-258 (fields the path uses)
    V3 p;
    V3 gNormal;
    Frame frame;
    uint32_t material;
    int32_t light;
    float areaPDF;
};

// Triangle::getSurfacePoint, Surface/TriangleMesh.cpp:180-215.  isect.p = org + dir * t (:170).
__device__ __forceinline__ void getSurfacePoint(const DevScene& sc, uint32_t tri, float b0, float b1, V3 p, SurfPt* sp) {
    const float4* st = reinterpret_cast<const float4*>(sc.shadeTris) + (size_t)tri * 6;
    const float4 q0 = st[0], q1 = st[1], q2 = st[2], q3 = st[3], q4 = st[4], q5 = st[5];
    sp->p = p;
    sp->gNormal = V3(q3.w, q4.w, q5.w);
    sp->material = __float_as_uint(q0.w);
    sp->light = (int32_t)__float_as_uint(q1.w);
    sp->areaPDF = q2.w;
    float b2 = 1.0f - b0 - b1;
    sp->frame.z = normalize(b0 * xyz(q0) + b1 * xyz(q1) + b2 * xyz(q2));
    sp->frame.x = normalize(b0 * xyz(q3) + b1 * xyz(q4) + b2 * xyz(q5));
    float dotNT = dot(sp->frame.z, sp->frame.x);
    if (fabsf(dotNT) >= 0.01f)
        sp->frame.x = normalize(sp->frame.x - dotNT * sp->frame.z);
    sp->frame.y = cross(sp->frame.z, sp->frame.x);
}

// DiffuseEDF::evaluate, EDFs/basic_EDFs.cpp:19-23: (dir.z > 0 ? 1.0f / M_PI : 0.0f) -> float
__device__ __forceinline__ float diffuseEDF(V3 dir) { return dir.z > 0.0f ? (float)(1.0 / kPi) : 0.0f; }

struct BsdfSample {
    V3 dir_sn;
    float dirPDF;
    uint32_t dirType;
};

// Per-material lobe type (basic_BSDFs.h:27,43,59-61; dispersive = !wls.lambdaSelected(),
// basic_SurfaceMaterials.cpp:42)
__device__ __forceinline__ uint32_t bsdfType(uint32_t matType, uint32_t wlFlags) {
    switch (matType) {
    case SLRHIP_MATERIAL_MATTE: return DT_Reflection | DT_LowFreq;
    case SLRHIP_MATERIAL_METAL: return DT_Reflection | DT_Delta0D;
    case SLRHIP_MATERIAL_GLASS: return DT_Reflection | DT_Transmission | DT_Delta0D | ((wlFlags & 1u) ? 0u : (uint32_t)DT_Dispersive);
    default: return 0;
    }
}

// BSDF::sample (DDF.h:231-246) over sampleInternal of LambertianBRDF / SpecularBRDF / SpecularBSDF
// (BSDFs/basic_BSDFs.cpp:12-26, 61-71, 95-149); query.flags = All, adjoint = false.
__device__ __forceinline__ RGB bsdfSample(const DevMaterial& m, uint32_t type, V3 dirOut, V3 gNorm, uint32_t wl, float uComp, float u0,
                                          float u1, BsdfSample* res) {
    res->dirPDF = 0.0f;
    res->dirType = 0;
    if (!dtMatches(type, DT_All)) return RGB();
    RGB fs_sn;
    switch (m.type) {
    case SLRHIP_MATERIAL_MATTE: {
        res->dir_sn = cosineSampleHemisphere(u0, u1);
        res->dirPDF = (float)((double)res->dir_sn.z / kPi);
        res->dirType = type;
        res->dir_sn.z *= dot(dirOut, gNorm) > 0 ? 1 : -1;
        fs_sn = rgb4(m.a) / (float)kPi;
        break;
    }
    case SLRHIP_MATERIAL_METAL: {
        res->dir_sn = V3(-dirOut.x, -dirOut.y, dirOut.z);
        res->dirPDF = 1.0f;
        res->dirType = type;
        fs_sn = rgb4(m.a) * fresnelConductor(rgb4(m.b), rgb4(m.c), dirOut.z) / fabsf(dirOut.z);
        break;
    }
    case SLRHIP_MATERIAL_GLASS: {
        RGB F = fresnelDielectric(rgb4(m.b), rgb4(m.c), dirOut.z);
        float reflectProb = importance(F, wl);
        if (uComp < reflectProb) {
            if (dirOut.z == 0.0f) return RGB();
            res->dir_sn = V3(-dirOut.x, -dirOut.y, dirOut.z);
            res->dirPDF = reflectProb;
            res->dirType = DT_Reflection | DT_Delta0D;
            fs_sn = rgb4(m.a) * F / fabsf(dirOut.z);
        }
        else {
            bool entering = dirOut.z > 0.0f;
            float etaExtW = rgb4(m.b).comp(wl), etaIntW = rgb4(m.c).comp(wl);
            float eEnter = entering ? etaExtW : etaIntW;
            float eExit = entering ? etaIntW : etaExtW;
            float sinEnter2 = 1.0f - dirOut.z * dirOut.z;
            float rrEta = eEnter / eExit;
            float sinExit2 = rrEta * rrEta * sinEnter2;
            if (sinExit2 >= 1.0f) return RGB();
            float cosExit = sqrtf(fmaxf(0.0f, 1.0f - sinExit2));
            if (entering) cosExit = -cosExit;
            res->dir_sn = V3(rrEta * -dirOut.x, rrEta * -dirOut.y, cosExit);
            res->dirPDF = 1.0f - reflectProb;
            res->dirType = DT_Transmission | DT_Delta0D | ((type & DT_Dispersive) ? (uint32_t)DT_Dispersive : 0u);
            float v = rgb4(m.a).comp(wl) * (1.0f - F.comp(wl));
            v *= (eEnter * eEnter) / (eExit * eExit);
            RGB ret(wl == 0 ? v : 0.0f, wl == 1 ? v : 0.0f, wl == 2 ? v : 0.0f);
            fs_sn = ret / fabsf(cosExit);
        }
        break;
    }
    default:
        return RGB();
    }
    if (res->dirPDF == 0.0f) return RGB();
    float snCorrection = fabsf(res->dir_sn.z / dot(res->dir_sn, gNorm));
    return fs_sn * snCorrection;
}

// BSDF::evaluate (DDF.h:247-267) + evaluatePDF (:268-279) for the NEE direction.
__device__ __forceinline__ RGB bsdfEvaluate(const DevMaterial& m, uint32_t type, V3 dirOut, V3 gNorm, V3 dir, float* pdf) {
    *pdf = 0.0f;
    if (dtMatches(type, DT_All) && m.type == SLRHIP_MATERIAL_MATTE) {
        // LambertianBRDF::evaluatePDFInternal basic_BSDFs.cpp:41-50
        if (!(dirOut.z * dir.z <= 0.0f)) *pdf = (float)((double)fabsf(dir.z) / kPi);
    }
    bool reflect = dot(gNorm, dirOut) * dot(gNorm, dir) > 0;                       // sideTest DDF.h:213-216
    uint32_t flags = DT_All & (DT_AllFreq | (reflect ? DT_Reflection : DT_Transmission));
    if (!dtMatches(type, flags)) return RGB();
    RGB fs_sn;
    if (m.type == SLRHIP_MATERIAL_MATTE) {
        // LambertianBRDF::evaluateInternal basic_BSDFs.cpp:28-39
        if (dirOut.z * dir.z <= 0.0f) fs_sn = RGB();
        else fs_sn = rgb4(m.a) / (float)kPi;
    }
    float snCorrection = fabsf(dir.z / dot(dir, gNorm));
    return fs_sn * snCorrection;
}

// RegularConstantDiscrete1D::sample, Core/distributions.cpp:97-107
__device__ __forceinline__ uint32_t selectLight(const DevScene& sc, float u, float* prob) {
    int idx = (int)sc.numLights;
    for (int d = (int)sc.lightPow2; d > 0; d >>= 1)
        if (idx - d > 0 && sc.lightCDF[idx - d] >= u) idx -= d;
    --idx;
    *prob = sc.lightPMF[idx];
    return (uint32_t)idx;
}

enum : uint32_t {
    ST_IDLE = 0,            // no more samples for this slot
    ST_START = 1,           // needs its first camera ray
    ST_FIRST_HIT = 2,       // camera ray in flight        (PathTracingRenderer.cpp:147)
    ST_NEXT_HIT = 3,        // BSDF-sampled ray in flight  (:225)
    ST_FINISH = 4           // path ended while a shadow ray was still pending
};
// flags word: [2:0] state | [9:3] pathLength | [11:10] selectedLambda | [12] wlFlags.LambdaIsSelected
//             | [13] previous direction was delta | [14] shadow ray pending
#define F_STATE(f) ((f) & 7u)
#define F_PATHLEN(f) (((f) >> 3) & 127u)
#define F_WL(f) (((f) >> 10) & 3u)
#define F_WLSEL(f) (((f) >> 12) & 1u)
#define F_DELTA(f) (((f) >> 13) & 1u)
#define F_SHADOW(f) (((f) >> 14) & 1u)
#define F_MAKE(state, len, wl, wlsel, delta, shadow) \
    ((state) | ((len) << 3) | ((wl) << 10) | ((wlsel) << 12) | ((delta) << 13) | ((shadow) << 14))

static const int kShadeBlock = 256;

__global__ __launch_bounds__(kShadeBlock) void k_shade(DevScene sc, PathBuffers pb, RenderParams rp, uint32_t parity) {
    const uint32_t slot = blockIdx.x * kShadeBlock + threadIdx.x;
    bool emitExt = false, emitShadow = false;
    bool becameIdle = false;
    uint32_t finished = 0;

    if (slot < rp.numSlots) {
        uint32_t flags = pb.flags[slot];
        uint32_t state = F_STATE(flags);
        if (state != ST_IDLE) {
            // ---- load path state -------------------------------------------------------------
            Rng rng;
            { uint4 r = pb.rng[slot]; rng.s0 = r.x; rng.s1 = r.y; rng.s2 = r.z; rng.s3 = r.w; }
            float4 ao = pb.alpha[slot];
            RGB alpha(ao.x, ao.y, ao.z);
            float bsdfPDFprev = ao.w;
            float4 s0 = pb.spR[slot], s1 = pb.spC[slot];
            RGB spR(s0.x, s0.y, s0.z), spC(s1.x, s1.y, s1.z);
            float camWeight = s0.w;
            uint32_t pathLength = F_PATHLEN(flags), wl = F_WL(flags), wlSel = F_WLSEL(flags);
            uint32_t sampleIdx = pb.sampleIdx[slot];
            V3 rayOrg, rayDir;
            float rayTmin = 0.0f;
            SurfPt surf;
            V3 dirOut_sn;
            bool haveSurf = false;
            bool finish = false;

            // ---- 1. resolve the pending next-event estimate (:180,202) ---------------------------
            if (F_SHADOW(flags)) {
                if (pb.visible[slot]) {
                    float4 c = pb.nee[slot];
                    kahanAdd(spR, spC, RGB(c.x, c.y, c.z));
                }
            }

            // ---- 2. the hit that just came back --------------------------------------------------
            if (state == ST_FINISH) {
                finish = true;
            }
            else if (state == ST_FIRST_HIT || state == ST_NEXT_HIT) {
                float4 h = pb.hit[slot];
                uint32_t tri = __float_as_uint(h.x);
                float4 o4 = pb.rayOrg[slot], d4 = pb.rayDir[slot];
                rayOrg = V3(o4.x, o4.y, o4.z);
                rayDir = V3(d4.x, d4.y, d4.z);
                if (tri == 0xFFFFFFFFu) {
                    finish = true;                      // :148 return Zero / :226 break
                }
                else {
                    getSurfacePoint(sc, tri, h.z, h.w, rayOrg + rayDir * h.y, &surf);
                    haveSurf = true;
                    dirOut_sn = surf.frame.toLocal(-rayDir);
                    if (surf.light >= 0) {
                        const DevMaterial& m = sc.materials[surf.material];
                        RGB Le = rgb4(m.emittance) * RGB(diffuseEDF(dirOut_sn));
                        if (state == ST_FIRST_HIT) {
                            kahanAdd(spR, spC, alpha * Le);                    // :152-156
                        }
                        else {
                            // implicit light sampling with MIS :232-249
                            float lightProb = sc.lightPMF[surf.light] * 1.0f;  // SurfaceObject.cpp:295-298, :78-80
                            float dist2 = sqLength(rayOrg - surf.p);
                            float lightPDF = lightProb * surf.areaPDF * dist2 / absDot(rayDir, surf.gNormal);
                            float MISWeight = 1.0f;
                            if (!F_DELTA(flags))
                                MISWeight = (bsdfPDFprev * bsdfPDFprev) / (lightPDF * lightPDF + bsdfPDFprev * bsdfPDFprev);
                            kahanAdd(spR, spC, alpha * Le * MISWeight);
                        }
                    }
                    if (state == ST_NEXT_HIT) {
                        // Russian roulette :254-258 (initY = importance(One) evaluated like the reference)
                        float initY = importance(RGB(1.0f), wl);
                        float continueProb = fminf(importance(alpha, wl) / initY, 1.0f);
                        if (rng.nextFloat() < continueProb) alpha = alpha / continueProb;
                        else finish = true;
                    }
                }
            }

            // ---- 3. next bounce: NEE + BSDF sampling (:161-221) ----------------------------------
            if (!finish && haveSurf) {
                ++pathLength;
                if (pathLength >= 100) {
                    finish = true;
                }
                else {
                    V3 gNorm_sn = surf.frame.toLocal(surf.gNormal);
                    const DevMaterial& m = sc.materials[surf.material];
                    uint32_t type = bsdfType(m.type, wlSel);
                    if (dtMatches(type, DT_WholeSphere | DT_NonDelta)) {
                        float lightProb;
                        uint32_t li = selectLight(sc, rng.nextFloat(), &lightProb);
                        lightProb *= 1.0f;
                        float lu0 = rng.nextFloat();
                        float lu1 = rng.nextFloat();
                        // Triangle::sample TriangleMesh.cpp:224-255
                        const float4* lt = reinterpret_cast<const float4*>(sc.lightTris) + (size_t)li * 9;
                        float4 l0 = lt[0], l1 = lt[1], l2 = lt[2], l3 = lt[3], l4 = lt[4], l5 = lt[5], l6 = lt[6], l7 = lt[7], l8 = lt[8];
                        float su1 = sqrtf(lu0);
                        float b0 = 1.0f - su1;
                        float b1 = lu1 * su1;
                        float b2 = 1.0f - b0 - b1;
                        V3 lp = b0 * xyz(l0) + b1 * xyz(l1) + b2 * xyz(l2);
                        V3 lgn(l3.w, l4.w, l5.w);
                        Frame lf;
                        lf.z = normalize(b0 * xyz(l3) + b1 * xyz(l4) + b2 * xyz(l5));
                        lf.x = normalize(b0 * xyz(l6) + b1 * xyz(l7) + b2 * xyz(l8));
                        lf.y = cross(lf.z, lf.x);
                        float areaPDF = l2.w;
                        RGB M = rgb4(sc.materials[__float_as_uint(l1.w)].emittance);
                        // shadow ray of Scene::testVisibility SurfaceObject.cpp:425-426
                        float dist = length(surf.p - lp);
                        V3 sdir = (lp - surf.p) / dist;
                        pb.shadowDir[slot] = make_float4(sdir.x, sdir.y, sdir.z, dist * (1 - kRayEpsilon));
                        emitShadow = true;
                        // contribution if visible :181-202
                        V3 dvec = lp - surf.p;
                        float dist2 = sqLength(dvec);
                        V3 shadowDir = dvec / sqrtf(dist2);
                        V3 shadowDir_l = lf.toLocal(-shadowDir);
                        V3 shadowDir_sn = surf.frame.toLocal(shadowDir);
                        RGB Le = M * RGB(diffuseEDF(shadowDir_l));
                        float lightPDF = lightProb * areaPDF;
                        float pdfDir;
                        RGB fs = bsdfEvaluate(m, type, dirOut_sn, gNorm_sn, shadowDir_sn, &pdfDir);
                        float cosLight = absDot(-shadowDir, lgn);
                        float bsdfPDF = pdfDir * cosLight / dist2;
                        float MISWeight = 1.0f;
                        if (!isinf(areaPDF))
                            MISWeight = (lightPDF * lightPDF) / (lightPDF * lightPDF + bsdfPDF * bsdfPDF);
                        float G = absDot(shadowDir_sn, gNorm_sn) * cosLight / dist2;
                        RGB contrib = alpha * Le * fs * (G * MISWeight / lightPDF);
                        pb.nee[slot] = make_float4(contrib.r, contrib.g, contrib.b, 0.0f);
                    }
                    float uComp = rng.nextFloat();
                    float u0 = rng.nextFloat();
                    float u1 = rng.nextFloat();
                    BsdfSample bs;
                    RGB fs = bsdfSample(m, type, dirOut_sn, gNorm_sn, wl, uComp, u0, u1, &bs);
                    if (fs.isZero() || bs.dirPDF == 0.0f) {
                        finish = true;                                         // :209
                    }
                    else {
                        if (bs.dirType & DT_Dispersive) {                      // :211-214
                            bs.dirPDF /= 3;
                            wlSel = 1;
                        }
                        alpha = alpha * (fs * absDot(bs.dir_sn, gNorm_sn) / bs.dirPDF);     // :215
                        V3 dirIn = surf.frame.fromLocal(bs.dir_sn);
                        rayOrg = surf.p;                                       // :221 Ray(p, dirIn, time, eps)
                        rayDir = dirIn;
                        rayTmin = kRayEpsilon;
                        bsdfPDFprev = bs.dirPDF;
                        flags = F_MAKE((uint32_t)ST_NEXT_HIT, pathLength, wl, wlSel, dtIsDelta(bs.dirType) ? 1u : 0u, emitShadow ? 1u : 0u);
                        emitExt = true;
                    }
                    if (emitShadow) {
                        // the shadow ray starts at the shading point, which is also the next ray's origin
                        float4 o = make_float4(surf.p.x, surf.p.y, surf.p.z, rayTmin);
                        pb.rayOrg[slot] = o;
                    }
                }
            }

            // ---- 4. path finished: accumulate and regenerate ---------------------------------------
            if (finish && emitShadow) {
                // the NEE of this bounce is still in flight: finish next iteration
                flags = F_MAKE((uint32_t)ST_FINISH, pathLength, wl, wlSel, 0u, 1u);
                finish = false;
            }
            bool regenerate = (state == ST_START);
            if (finish) {
                // sensor->add(p.x, p.y, wls, weight * C)  PathTracingRenderer.cpp:126-130
                float4 a0 = pb.accR[slot], a1 = pb.accC[slot];
                RGB accR(a0.x, a0.y, a0.z), accC(a1.x, a1.y, a1.z);
                RGB weight = (RGB(1.0f) * RGB(1.0f)) * camWeight;
                kahanAdd(accR, accC, weight * spR);
                pb.accR[slot] = make_float4(accR.r, accR.g, accR.b, 0.0f);
                pb.accC[slot] = make_float4(accC.r, accC.g, accC.b, 0.0f);
                ++sampleIdx;
                ++finished;
                regenerate = true;
            }
            if (regenerate) {
                const uint32_t stripe = slot / rp.numPixels;
                const uint32_t pix = slot - stripe * rp.numPixels;
                const uint32_t pass = rp.sppBegin + stripe + sampleIdx * rp.stripes;
                if (pass >= rp.sppBegin + rp.sppCount) {
                    flags = F_MAKE((uint32_t)ST_IDLE, 0u, 0u, 0u, 0u, 0u);
                    becameIdle = true;
                }
                else {
                    // Job::kernel PathTracingRenderer.cpp:100-120, draws in source (left-to-right) order
                    const uint32_t xy = pb.pixelXY[pix];
                    const uint32_t px = xy & 0xFFFFu, py = xy >> 16;
                    rng.seed(sampleSeed(rp.rngSeed, px, py, pass));
                    float v = rng.nextFloat();
                    float time = rp.timeStart * (1 - v) + rp.timeEnd * v;
                    (void)time;
                    float pxx = px + rng.nextFloat();
                    float pyy = py + rng.nextFloat();
                    rng.nextFloat();                                           // wavelength offset (unused in RGB, RGBTypes.h:37-45)
                    float uLambda = rng.nextFloat();
                    wl = min((uint32_t)(uint16_t)(3 * uLambda), 2u);
                    wlSel = 0;
                    float lu0 = rng.nextFloat();
                    float lu1 = rng.nextFloat();
                    // PerspectiveCamera::sample PerspectiveCamera.cpp:33-57
                    float lx, ly;
                    concentricSampleDisk(lu0, lu1, &lx, &ly);
                    V3 orgLocal(sc.camera.lensRadius * lx, sc.camera.lensRadius * ly, 0.0f);
                    V3 lensP = mulPoint(sc.camera.mat, orgLocal);
                    V3 lensN = mulNormal(sc.camera.matInv, V3(0, 0, 1));
                    Frame lf;
                    lf.z = lensN;
                    lf.x = mulVector(sc.camera.mat, V3(1, 0, 0));
                    lf.y = cross(lf.z, lf.x);
                    // PerspectiveIDF::sample :63-74 with IDFSample(p.x / W, p.y / H)
                    float sx = pxx / (float)rp.imageWidth;
                    float sy = pyy / (float)rp.imageHeight;
                    V3 pFocus(sc.camera.opWidth * (0.5f - sx), sc.camera.opHeight * (0.5f - sy), sc.camera.objPlaneDistance);
                    V3 dirLocal = normalize(pFocus - orgLocal);
                    float dirPDF = sc.camera.imgPlaneDistance * sc.camera.imgPlaneDistance /
                                   ((dirLocal.z * dirLocal.z * dirLocal.z) * sc.camera.imgPlaneArea);
                    rayOrg = lensP;
                    rayDir = lf.fromLocal(dirLocal);
                    rayTmin = 0.0f;
                    // weight :126 (selectWLPDF = 1 in RGB mode)
                    camWeight = absDot(rayDir, lensN) / (sc.camera.areaPDF * dirPDF * 1.0f);
                    alpha = RGB(1.0f);
                    spR = RGB(); spC = RGB();
                    bsdfPDFprev = 0.0f;
                    flags = F_MAKE((uint32_t)ST_FIRST_HIT, 0u, wl, 0u, 0u, 0u);
                    emitExt = true;
                    emitShadow = false;
                }
            }

            // ---- store path state --------------------------------------------------------------------
            pb.flags[slot] = flags;
            pb.sampleIdx[slot] = sampleIdx;
            pb.rng[slot] = make_uint4(rng.s0, rng.s1, rng.s2, rng.s3);
            pb.alpha[slot] = make_float4(alpha.r, alpha.g, alpha.b, bsdfPDFprev);
            pb.spR[slot] = make_float4(spR.r, spR.g, spR.b, camWeight);
            pb.spC[slot] = make_float4(spC.r, spC.g, spC.b, 0.0f);
            if (emitExt) {
                pb.rayOrg[slot] = make_float4(rayOrg.x, rayOrg.y, rayOrg.z, rayTmin);
                pb.rayDir[slot] = make_float4(rayDir.x, rayDir.y, rayDir.z, INFINITY);
            }
        }
    }

    // ---- wave-level stream compaction: ballot + popcount prefix, one atomic per wave per queue ------
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t below = (1ull << lane) - 1ull;
    {
        const uint64_t m = __ballot(emitExt);
        if (m) {
            uint32_t base = 0;
            if (lane == (uint32_t)__ffsll((long long)m) - 1u) base = atomicAdd(&pb.queueCount[(parity ^ 1) * 2 + 0], (uint32_t)__popcll(m));
            base = __shfl(base, __ffsll((long long)m) - 1);
            if (emitExt) pb.extQueue[base + __popcll(m & below)] = slot;
        }
    }
    {
        const uint64_t m = __ballot(emitShadow);
        if (m) {
            uint32_t base = 0;
            if (lane == (uint32_t)__ffsll((long long)m) - 1u) base = atomicAdd(&pb.queueCount[(parity ^ 1) * 2 + 1], (uint32_t)__popcll(m));
            base = __shfl(base, __ffsll((long long)m) - 1);
            if (emitShadow) pb.shadowQueue[base + __popcll(m & below)] = slot;
        }
    }
    {
        const uint64_t mi = __ballot(becameIdle);
        const uint64_t mf = __ballot(finished != 0);
        if (lane == 0) {
            if (mi) atomicAdd(&pb.activeSlots[0], (uint32_t)(0u - (uint32_t)__popcll(mi)));
            if (mf) atomicAdd((unsigned long long*)&pb.totals[2], (unsigned long long)__popcll(mf));
        }
    }
    if (rp.countSlots) {
        const uint64_t ma = __ballot(slot < rp.numSlots && (emitExt || emitShadow || finished || becameIdle));
        if (lane == 0 && ma) atomicAdd((unsigned long long*)&pb.totals[3], (unsigned long long)__popcll(ma));
    }
}

// Start of a render() call: every slot of the shard goes to ST_START with sample counter 0
// (accumulators are kept: render() continues the image begun by render_begin()).
__global__ void k_reset_slots(PathBuffers pb, RenderParams rp, uint32_t clearAccumulators) {
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= rp.numSlots) return;
    pb.flags[slot] = F_MAKE((uint32_t)ST_START, 0u, 0u, 0u, 0u, 0u);
    pb.sampleIdx[slot] = 0;
    pb.visible[slot] = 0;
    if (clearAccumulators) {
        pb.accR[slot] = make_float4(0, 0, 0, 0);
        pb.accC[slot] = make_float4(0, 0, 0, 0);
    }
    if (slot == 0) {
        pb.queueCount[0] = pb.queueCount[1] = pb.queueCount[2] = pb.queueCount[3] = 0;
        pb.activeSlots[0] = rp.numSlots;
    }
}

// ImageSensor read-out: [H][W][3] linear sums; stripes of one pixel are added in stripe order.
__global__ void k_resolve(PathBuffers pb, RenderParams rp, float* dst) {
    const uint32_t pix = blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= rp.numPixels) return;
    const uint32_t xy = pb.pixelXY[pix];
    const uint32_t px = xy & 0xFFFFu, py = xy >> 16;
    float4 a = pb.accR[pix];
    RGB sum(a.x, a.y, a.z);
    for (uint32_t s = 1; s < rp.stripes; ++s) {
        float4 b = pb.accR[(size_t)s * rp.numPixels + pix];
        sum = sum + RGB(b.x, b.y, b.z);
    }
    float* o = dst + ((size_t)py * rp.imageWidth + px) * 3;
    o[0] = sum.r; o[1] = sum.g; o[2] = sum.b;
}

// ---- host-callable launchers -------------------------------------------------------------------------
void launchResetSlots(const PathBuffers& pb, const RenderParams& rp, bool clearAcc, hipStream_t stream) {
    hipLaunchKernelGGL(k_reset_slots, dim3((rp.numSlots + 255) / 256), dim3(256), 0, stream, pb, rp, clearAcc ? 1u : 0u);
}

void launchTraceClosest(const DevScene& sc, const PathBuffers& pb, uint32_t parity, uint32_t blocks, bool count, hipStream_t stream) {
    if (count) hipLaunchKernelGGL(k_trace_closest<true>, dim3(blocks), dim3(kTraceBlock), 0, stream, sc, pb, parity);
    else hipLaunchKernelGGL(k_trace_closest<false>, dim3(blocks), dim3(kTraceBlock), 0, stream, sc, pb, parity);
}
void launchTraceShadow(const DevScene& sc, const PathBuffers& pb, uint32_t parity, uint32_t blocks, bool count, hipStream_t stream) {
    if (count) hipLaunchKernelGGL(k_trace_shadow<true>, dim3(blocks), dim3(kTraceBlock), 0, stream, sc, pb, parity);
    else hipLaunchKernelGGL(k_trace_shadow<false>, dim3(blocks), dim3(kTraceBlock), 0, stream, sc, pb, parity);
}
void launchShade(const DevScene& sc, const PathBuffers& pb, const RenderParams& rp, uint32_t parity, hipStream_t stream) {
    hipLaunchKernelGGL(k_shade, dim3((rp.numSlots + kShadeBlock - 1) / kShadeBlock), dim3(kShadeBlock), 0, stream, sc, pb, rp, parity);
}

void launchResolve(const PathBuffers& pb, const RenderParams& rp, float* dst, hipStream_t stream) {
    hipLaunchKernelGGL(k_resolve, dim3((rp.numPixels + 255) / 256), dim3(256), 0, stream, pb, rp, dst);
}

void launchTraceBatch(const DevScene& sc, const float4* org, const float4* dir, float4* out, uint32_t n, hipStream_t stream) {
    uint32_t blocks = (n + kTraceBlock - 1) / kTraceBlock;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_trace_batch, dim3(blocks), dim3(kTraceBlock), 0, stream, sc, org, dir, out, n);
}

} // namespace slrhip
