// pt_shade.hip — the shading half of the wavefront path tracer (gfx950, wave64).
//
// One path SLOT per (pixel, sample stripe).  A slot carries one light path at a time through the
// reference's bounce loop (Renderers/PathTracingRenderer.cpp:137-262).  Per wavefront iteration:
//
//   k_regen          finished slots (compacted queue): add weight*C to the slot's pixel accumulator in
//                    pass order — the Kahan sum of RGBStorage::add (RGBTypes.h:176-179), so the
//                    framebuffer needs no atomics — then start the next sample of the same pixel:
//                    Job::kernel's camera-ray half (PathTracingRenderer.cpp:100-120).
//   k_trace_closest  extension-ray queue   (pt_trace.hip)
//   k_trace_shadow   shadow-ray queue      (pt_trace.hip)
//   k_logic          every live slot: resolve the pending next-event estimate, shade the hit
//                    (getSurfacePoint, emission + MIS, Russian roulette), then the next bounce: light
//                    sampling + BSDF sampling (:161-221).  Emits the next extension ray, a shadow ray
//                    and/or the slot index into the regen queue.
//
// Queues are slot-index lists in HBM built by wave ballot + popcount prefix with one atomic per wave
// per queue; all path state is SoA in 16-byte records so a wave's loads are 1 KiB bursts.  The state
// loads of k_logic are issued together at the top (the kernel is latency-bound: PMC shows > 80 % of
// wave cycles waiting on memory), and the material / light tables live in LDS.
#include <hip/hip_runtime.h>

#include "pt_bsdf.h"
#include "pt_kernels.h"

namespace slrhip {

enum : uint32_t {
    ST_IDLE = 0,            // no more samples for this slot
    ST_REGEN = 1,           // in the regen queue: accumulate (if a path just ended) and start the next sample
    ST_FIRST_HIT = 2,       // camera ray in flight        (PathTracingRenderer.cpp:147)
    ST_NEXT_HIT = 3,        // BSDF-sampled ray in flight  (:225)
    ST_FINISH = 4           // path ended while a shadow ray was still pending
};
// flags word: [2:0] state | [9:3] pathLength | [11:10] selectedLambda | [12] wlFlags.LambdaIsSelected
//             | [13] previous direction was delta | [14] shadow ray pending | [15] a finished path awaits accumulation
#define F_STATE(f) ((f) & 7u)
#define F_PATHLEN(f) (((f) >> 3) & 127u)
#define F_WL(f) (((f) >> 10) & 3u)
#define F_WLSEL(f) (((f) >> 12) & 1u)
#define F_DELTA(f) (((f) >> 13) & 1u)
#define F_SHADOW(f) (((f) >> 14) & 1u)
#define F_HASPATH(f) (((f) >> 15) & 1u)
#define F_MAKE(state, len, wl, wlsel, delta, shadow) \
    ((state) | ((len) << 3) | ((wl) << 10) | ((wlsel) << 12) | ((delta) << 13) | ((shadow) << 14))

static const int kShadeBlock = 256;
static const int kLdsMaterials = 32;
static const int kLdsLights = 16;

// Append `slot` to the workgroup's region of up to two queues: wave ballots + popcount prefixes, the four
// wave counts meet in LDS, ONE atomic per queue per workgroup (on the region's own counter line).
struct PushLds {
    uint32_t count[Q_KINDS][4];
    uint32_t base[Q_KINDS];
};
__device__ __forceinline__ void blockPush(PushLds& pl, bool emit0, bool emit1, uint32_t slot, uint32_t* queue0, uint32_t* queue1,
                                          uint32_t* counters /* set being filled */, uint32_t shardCapacity) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t shard = blockIdx.x % kShards;
    const uint64_t m0 = __ballot(emit0), m1 = __ballot(emit1);
    if (lane == 0) { pl.count[0][wave] = (uint32_t)__popcll(m0); pl.count[1][wave] = (uint32_t)__popcll(m1); }
    __syncthreads();
    if (threadIdx.x < Q_KINDS) {
        const uint32_t q = threadIdx.x;
        const uint32_t total = pl.count[q][0] + pl.count[q][1] + pl.count[q][2] + pl.count[q][3];
        pl.base[q] = total ? atomicAdd(&counters[(q * kShards + shard) * kCounterStride], total) : 0u;
    }
    __syncthreads();
    const uint64_t below = (1ull << lane) - 1ull;
    if (emit0) {
        uint32_t off = pl.base[0];
        for (uint32_t w = 0; w < wave; ++w) off += pl.count[0][w];
        queue0[(size_t)shard * shardCapacity + off + __popcll(m0 & below)] = slot;
    }
    if (emit1) {
        uint32_t off = pl.base[1];
        for (uint32_t w = 0; w < wave; ++w) off += pl.count[1][w];
        queue1[(size_t)shard * shardCapacity + off + __popcll(m1 & below)] = slot;
    }
}

struct SurfPt {       // Core/geometry.h:239-258 (fields the path uses)
    V3 p;
    V3 gNormal;
    Frame frame;
    uint32_t material;
    int32_t light;
    float areaPDF;
};

// RegularConstantDiscrete1D::sample, Core/distributions.cpp:97-107
__device__ __forceinline__ uint32_t selectLight(const DevScene& sc, const float* cdf, const float* pmf, float u, float* prob) {
    int idx = (int)sc.numLights;
    for (int d = (int)sc.lightPow2; d > 0; d >>= 1)
        if (idx - d > 0 && cdf[idx - d] >= u) idx -= d;
    --idx;
    *prob = pmf[idx];
    return (uint32_t)idx;
}

struct ShadeLds {
    float4 mats[kLdsMaterials * 5];        // DevMaterial = 5 x float4
    float4 lights[kLdsLights * 9];         // LightTri   = 9 x float4
    float lightPMF[kLdsLights];
    float lightCDF[kLdsLights + 1];
};

template <bool LDS_TABLES, bool MF>
__global__ __launch_bounds__(kShadeBlock) void k_logic(DevScene sc, PathBuffers pb, RenderParams rp, uint32_t parity) {
    __shared__ ShadeLds lds;
    __shared__ PushLds pushLds;
    if (LDS_TABLES) {
        const float4* gm = reinterpret_cast<const float4*>(sc.materials);
        for (uint32_t i = threadIdx.x; i < sc.numMaterials * 5; i += kShadeBlock) lds.mats[i] = gm[i];
        const float4* gl = reinterpret_cast<const float4*>(sc.lightTris);
        for (uint32_t i = threadIdx.x; i < sc.numLights * 9; i += kShadeBlock) lds.lights[i] = gl[i];
        if (threadIdx.x < sc.numLights) lds.lightPMF[threadIdx.x] = sc.lightPMF[threadIdx.x];
        if (threadIdx.x <= sc.numLights) lds.lightCDF[threadIdx.x] = sc.lightCDF[threadIdx.x];
        __syncthreads();
    }
    const float* lightPMF = LDS_TABLES ? lds.lightPMF : sc.lightPMF;
    const float* lightCDF = LDS_TABLES ? lds.lightCDF : sc.lightCDF;

    const uint32_t slot = blockIdx.x * kShadeBlock + threadIdx.x;
    bool emitExt = false, emitShadow = false, emitRegen = false;
    uint32_t* qw = pb.queueCount + (parity ^ 1) * kQueueSetWords;

    if (slot < rp.numSlots) {
        // ---- all state loads up front: one memory round trip instead of a dependent chain -----------
        uint32_t flags = pb.flags[slot];
        const uint4 r4 = pb.rng[slot];
        const float4 ao = pb.alpha[slot];
        const float4 s0 = pb.spR[slot], s1 = pb.spC[slot];
        const float4 h = pb.hit[slot];
        const float4 o4 = pb.rayOrg[slot], d4 = pb.rayDir[slot];
        const float4 neeC = pb.nee[slot];
        const uint32_t vis = pb.visible[slot];

        const uint32_t state = F_STATE(flags);
        if (state == ST_FIRST_HIT || state == ST_NEXT_HIT || state == ST_FINISH) {
            Rng rng;
            rng.s0 = r4.x; rng.s1 = r4.y; rng.s2 = r4.z; rng.s3 = r4.w;
            RGB alpha(ao.x, ao.y, ao.z);
            float bsdfPDFprev = ao.w;
            RGB spR(s0.x, s0.y, s0.z), spC(s1.x, s1.y, s1.z);
            const float camWeight = s0.w;
            uint32_t pathLength = F_PATHLEN(flags), wlSel = F_WLSEL(flags);
            const uint32_t wl = F_WL(flags);
            V3 rayOrg(o4.x, o4.y, o4.z), rayDir(d4.x, d4.y, d4.z);
            float rayTmin = 0.0f;
            SurfPt surf;
            V3 dirOut_sn;
            bool haveSurf = false;
            bool finish = false;
            const uint32_t tri = __float_as_uint(h.x);

            // the hit triangle's shading record: issued before anything else is computed
            float4 q0, q1, q2, q3, q4, q5;
            const bool hasHit = state != ST_FINISH && tri != 0xFFFFFFFFu;
            if (hasHit) {
                const float4* st = reinterpret_cast<const float4*>(sc.shadeTris) + (size_t)tri * 6;
                q0 = st[0]; q1 = st[1]; q2 = st[2]; q3 = st[3]; q4 = st[4]; q5 = st[5];
            }

            // ---- 1. resolve the pending next-event estimate (:180,202) ---------------------------------
            if (F_SHADOW(flags) && vis) kahanAdd(spR, spC, RGB(neeC.x, neeC.y, neeC.z));

            // ---- 2. the hit that just came back ------------------------------------------------------------
            Mat m;
            if (!hasHit) {
                finish = true;                      // ST_FINISH, or a miss: :148 return Zero / :226 break
            }
            else {
                // Triangle::getSurfacePoint, Surface/TriangleMesh.cpp:180-215.  isect.p = org + dir * t (:170)
                surf.p = rayOrg + rayDir * h.y;
                surf.gNormal = V3(q3.w, q4.w, q5.w);
                surf.material = __float_as_uint(q0.w);
                surf.light = (int32_t)__float_as_uint(q1.w);
                surf.areaPDF = q2.w;
                m = loadMat(LDS_TABLES ? reinterpret_cast<const DevMaterial*>(lds.mats) + surf.material : sc.materials + surf.material);
                const float b0 = h.z, b1 = h.w;
                const float b2 = 1.0f - b0 - b1;
                surf.frame.z = normalize(b0 * xyz(q0) + b1 * xyz(q1) + b2 * xyz(q2));
                surf.frame.x = normalize(b0 * xyz(q3) + b1 * xyz(q4) + b2 * xyz(q5));
                const float dotNT = dot(surf.frame.z, surf.frame.x);
                if (fabsf(dotNT) >= 0.01f) surf.frame.x = normalize(surf.frame.x - dotNT * surf.frame.z);
                surf.frame.y = cross(surf.frame.z, surf.frame.x);
                haveSurf = true;
                dirOut_sn = surf.frame.toLocal(-rayDir);
                if (surf.light >= 0) {
                    RGB Le = m.emittance * RGB(diffuseEDF(dirOut_sn));
                    if (state == ST_FIRST_HIT) {
                        kahanAdd(spR, spC, alpha * Le);                        // :152-156
                    }
                    else {
                        // implicit light sampling with MIS :232-249
                        float lightProb = lightPMF[surf.light] * 1.0f;          // SurfaceObject.cpp:295-298, :78-80
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

            // ---- 3. next bounce: NEE + BSDF sampling (:161-221) ----------------------------------------------
            if (!finish && haveSurf) {
                ++pathLength;
                if (pathLength >= 100) {
                    finish = true;
                }
                else {
                    V3 gNorm_sn = surf.frame.toLocal(surf.gNormal);
                    uint32_t type = bsdfType(m.type, wlSel);
                    if (dtMatches(type, DT_WholeSphere | DT_NonDelta)) {
                        float lightProb;
                        uint32_t li = selectLight(sc, lightCDF, lightPMF, rng.nextFloat(), &lightProb);
                        lightProb *= 1.0f;
                        float lu0 = rng.nextFloat();
                        float lu1 = rng.nextFloat();
                        // Triangle::sample TriangleMesh.cpp:224-255
                        const float4* lt = (LDS_TABLES ? lds.lights : reinterpret_cast<const float4*>(sc.lightTris)) + (size_t)li * 9;
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
                        const uint32_t lmat = __float_as_uint(l1.w);
                        RGB M = loadMat(LDS_TABLES ? reinterpret_cast<const DevMaterial*>(lds.mats) + lmat : sc.materials + lmat).emittance;
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
                        RGB fs = bsdfEvaluate<MF>(m, type, dirOut_sn, gNorm_sn, shadowDir_sn, wl, &pdfDir);
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
                    RGB fs = bsdfSample<MF>(m, type, dirOut_sn, gNorm_sn, wl, uComp, u0, u1, &bs);
                    if (fs.isZero() || bs.dirPDF == 0.0f) {
                        finish = true;                                         // :209
                    }
                    else {
                        if (bs.dirType & DT_Dispersive) {                      // :211-214
                            bs.dirPDF /= 3;
                            wlSel = 1;
                        }
                        alpha = alpha * (fs * absDot(bs.dir_sn, gNorm_sn) / bs.dirPDF);     // :215
                        rayDir = surf.frame.fromLocal(bs.dir_sn);
                        rayOrg = surf.p;                                       // :221 Ray(p, dirIn, time, eps)
                        rayTmin = kRayEpsilon;
                        bsdfPDFprev = bs.dirPDF;
                        flags = F_MAKE((uint32_t)ST_NEXT_HIT, pathLength, wl, wlSel, dtIsDelta(bs.dirType) ? 1u : 0u, emitShadow ? 1u : 0u);
                        emitExt = true;
                    }
                    // the shadow ray starts at the shading point, which is also the next ray's origin
                    if (emitShadow && !emitExt) pb.rayOrg[slot] = make_float4(surf.p.x, surf.p.y, surf.p.z, kRayEpsilon);
                }
            }

            // ---- 4. path finished ----------------------------------------------------------------------------
            if (finish && emitShadow) {
                // the NEE of this bounce is still in flight: finish next iteration
                flags = F_MAKE((uint32_t)ST_FINISH, pathLength, wl, wlSel, 0u, 1u);
            }
            else if (finish) {
                flags = F_MAKE((uint32_t)ST_REGEN, 0u, 0u, 0u, 0u, 0u) | (1u << 15);
                emitRegen = true;
            }

            // ---- store path state ---------------------------------------------------------------------------
            pb.flags[slot] = flags;
            pb.spR[slot] = make_float4(spR.r, spR.g, spR.b, camWeight);
            if (!emitRegen) {
                pb.spC[slot] = make_float4(spC.r, spC.g, spC.b, 0.0f);
                pb.rng[slot] = make_uint4(rng.s0, rng.s1, rng.s2, rng.s3);
                pb.alpha[slot] = make_float4(alpha.r, alpha.g, alpha.b, bsdfPDFprev);
            }
            if (emitExt) {
                pb.rayOrg[slot] = make_float4(rayOrg.x, rayOrg.y, rayOrg.z, rayTmin);
                pb.rayDir[slot] = make_float4(rayDir.x, rayDir.y, rayDir.z, INFINITY);
            }
        }
    }

    // ---- stream compaction of the shadow rays and of the finished slots -----------------------------------------
    (void)emitExt;     // extension rays need no queue: the traversal kernel reads the state flag
    blockPush(pushLds, emitShadow, emitRegen, slot, pb.shadowQueue, pb.regenQueue, qw, rp.shardCapacity);
    if (rp.countSlots) {
        const uint64_t ma = __ballot(emitExt || emitShadow || emitRegen);
        if ((threadIdx.x & 63u) == 0 && ma)
            atomicAdd((unsigned long long*)&pb.totals[totalIndex(T_SLOT_VISITS, blockIdx.x % kShards)], (unsigned long long)__popcll(ma));
    }
}

// Finished (or brand-new) slots, dense: sensor->add + the camera-ray half of Job::kernel.
// Workgroup b serves chunk b / kShards of queue region b % kShards; surplus workgroups exit at once.
__global__ __launch_bounds__(kShadeBlock) void k_regen(DevScene sc, PathBuffers pb, RenderParams rp, uint32_t parity) {
    const uint32_t shard = blockIdx.x % kShards;
    const uint32_t n = pb.queueCount[queueCounterIndex(parity, Q_REGEN, shard)];
    const uint32_t i = (blockIdx.x / kShards) * kShadeBlock + threadIdx.x;
    if ((blockIdx.x / kShards) * kShadeBlock >= n) return;
    bool becameIdle = false;
    uint32_t slot = 0;
    if (i < n) {
        slot = pb.regenQueue[(size_t)shard * rp.shardCapacity + i];
        const uint32_t flags = pb.flags[slot];
        uint32_t sampleIdx = pb.sampleIdx[slot];
        if (F_HASPATH(flags)) {
            // sensor->add(p.x, p.y, wls, weight * C)  PathTracingRenderer.cpp:126-130
            const float4 s0 = pb.spR[slot];
            const float4 a0 = pb.accR[slot], a1 = pb.accC[slot];
            RGB accR(a0.x, a0.y, a0.z), accC(a1.x, a1.y, a1.z);
            const RGB weight = (RGB(1.0f) * RGB(1.0f)) * s0.w;
            kahanAdd(accR, accC, weight * RGB(s0.x, s0.y, s0.z));
            pb.accR[slot] = make_float4(accR.r, accR.g, accR.b, 0.0f);
            pb.accC[slot] = make_float4(accC.r, accC.g, accC.b, 0.0f);
            ++sampleIdx;
        }
        const uint32_t stripe = slot / rp.numPixels;
        const uint32_t pix = slot - stripe * rp.numPixels;
        const uint32_t pass = rp.sppBegin + stripe + sampleIdx * rp.stripes;
        if (pass >= rp.sppBegin + rp.sppCount) {
            pb.flags[slot] = F_MAKE((uint32_t)ST_IDLE, 0u, 0u, 0u, 0u, 0u);
            becameIdle = true;
        }
        else {
            // Job::kernel PathTracingRenderer.cpp:100-120, draws in source (left-to-right) order
            const uint32_t xy = pb.pixelXY[pix];
            const uint32_t px = xy & 0xFFFFu, py = xy >> 16;
            Rng rng;
            rng.seed(sampleSeed(rp.rngSeed, px, py, pass));
            float v = rng.nextFloat();
            float time = rp.timeStart * (1 - v) + rp.timeEnd * v;
            (void)time;
            float pxx = px + rng.nextFloat();
            float pyy = py + rng.nextFloat();
            rng.nextFloat();                                           // wavelength offset (unused in RGB, RGBTypes.h:37-45)
            float uLambda = rng.nextFloat();
            const uint32_t wl = min((uint32_t)(uint16_t)(3 * uLambda), 2u);
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
            V3 rayDir = lf.fromLocal(dirLocal);
            // weight :126 (selectWLPDF = 1 in RGB mode)
            float camWeight = absDot(rayDir, lensN) / (sc.camera.areaPDF * dirPDF * 1.0f);
            pb.flags[slot] = F_MAKE((uint32_t)ST_FIRST_HIT, 0u, wl, 0u, 0u, 0u);
            pb.rng[slot] = make_uint4(rng.s0, rng.s1, rng.s2, rng.s3);
            pb.alpha[slot] = make_float4(1.0f, 1.0f, 1.0f, 0.0f);
            pb.spR[slot] = make_float4(0.0f, 0.0f, 0.0f, camWeight);
            pb.spC[slot] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            pb.rayOrg[slot] = make_float4(lensP.x, lensP.y, lensP.z, 0.0f);
            pb.rayDir[slot] = make_float4(rayDir.x, rayDir.y, rayDir.z, INFINITY);
        }
        pb.sampleIdx[slot] = sampleIdx;
    }
    // slots run out of samples only at the very end of a render() call, so this atomic is rare
    const uint64_t mi = __ballot(becameIdle);
    if ((threadIdx.x & 63u) == 0 && mi) atomicAdd(&pb.activeSlots[0], (uint32_t)(0u - (uint32_t)__popcll(mi)));
}

// Start of a render() call: every slot of the shard enters the regen queue with sample counter 0
// (accumulators are kept unless asked: render() continues the image begun by render_begin()).
__global__ void k_reset_slots(PathBuffers pb, RenderParams rp, uint32_t clearAccumulators) {
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;     // blockDim == kShadeBlock
    if (slot < rp.numSlots) {
        pb.flags[slot] = F_MAKE((uint32_t)ST_REGEN, 0u, 0u, 0u, 0u, 0u);
        pb.sampleIdx[slot] = 0;
        pb.visible[slot] = 0;
        // slot block b goes to region b % kShards at chunk b / kShards, exactly as k_logic would append it
        pb.regenQueue[(size_t)(blockIdx.x % kShards) * rp.shardCapacity + (blockIdx.x / kShards) * kShadeBlock + threadIdx.x] = slot;
        if (clearAccumulators) {
            pb.accR[slot] = make_float4(0, 0, 0, 0);
            pb.accC[slot] = make_float4(0, 0, 0, 0);
        }
    }
    if (blockIdx.x == 0) {
        for (uint32_t k = threadIdx.x; k < 2 * kQueueSetWords; k += blockDim.x) pb.queueCount[k] = 0;
        __syncthreads();
        if (threadIdx.x < kShards) {
            // entries of region r: full blocks r, r + kShards, ... ; the last block of the grid may be partial
            const uint32_t numBlocks = (rp.numSlots + kShadeBlock - 1) / kShadeBlock;
            const uint32_t r = threadIdx.x;
            uint32_t cnt = 0;
            if (r < numBlocks) {
                const uint32_t blocksInRegion = (numBlocks - 1 - r) / kShards + 1;
                cnt = blocksInRegion * kShadeBlock;
                const uint32_t lastBlock = numBlocks - 1;
                if (lastBlock % kShards == r) cnt -= numBlocks * kShadeBlock - rp.numSlots;
            }
            pb.queueCount[queueCounterIndex(0, Q_REGEN, r)] = cnt;
        }
        if (threadIdx.x == 0) pb.activeSlots[0] = rp.numSlots;
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

// ---- host-callable launchers -----------------------------------------------------------------------------------
void launchResetSlots(const PathBuffers& pb, const RenderParams& rp, bool clearAcc, hipStream_t stream) {
    hipLaunchKernelGGL(k_reset_slots, dim3((rp.numSlots + kShadeBlock - 1) / kShadeBlock), dim3(kShadeBlock), 0, stream, pb, rp,
                       clearAcc ? 1u : 0u);
}
void launchRegen(const DevScene& sc, const PathBuffers& pb, const RenderParams& rp, uint32_t parity, hipStream_t stream) {
    // queue lengths are only known on the device: launch for the worst case, surplus workgroups exit at once
    hipLaunchKernelGGL(k_regen, dim3(rp.shardCapacity / kShadeBlock * kShards), dim3(kShadeBlock), 0, stream, sc, pb, rp, parity);
}
void launchLogic(const DevScene& sc, const PathBuffers& pb, const RenderParams& rp, uint32_t parity, hipStream_t stream) {
    const dim3 grid((rp.numSlots + kShadeBlock - 1) / kShadeBlock), block(kShadeBlock);
    const bool ldsTables = sc.numMaterials <= (uint32_t)kLdsMaterials && sc.numLights <= (uint32_t)kLdsLights;
    // four instantiations: the microfacet (GGX) code costs ~45 VGPRs, so scenes without such lobes get a kernel without it
    if (ldsTables && !sc.hasMicrofacet) hipLaunchKernelGGL((k_logic<true, false>), grid, block, 0, stream, sc, pb, rp, parity);
    else if (ldsTables) hipLaunchKernelGGL((k_logic<true, true>), grid, block, 0, stream, sc, pb, rp, parity);
    else if (!sc.hasMicrofacet) hipLaunchKernelGGL((k_logic<false, false>), grid, block, 0, stream, sc, pb, rp, parity);
    else hipLaunchKernelGGL((k_logic<false, true>), grid, block, 0, stream, sc, pb, rp, parity);
}
void launchResolve(const PathBuffers& pb, const RenderParams& rp, float* dst, hipStream_t stream) {
    hipLaunchKernelGGL(k_resolve, dim3((rp.numPixels + 255) / 256), dim3(256), 0, stream, pb, rp, dst);
}

} // namespace slrhip
