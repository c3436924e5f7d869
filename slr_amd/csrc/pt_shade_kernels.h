// pt_shade_kernels.h — kernel templates of the shading half of the wavefront path tracer (gfx950, wave64).
//
// A path SLOT carries one light path at a time through the reference's bounce loop (Renderers/PathTracingRenderer.cpp:137-262);
// it is not bound to a pixel: whenever its path ends it takes the next sample of its wave's work queue (pt_kernels.h WorkItem).
// One wavefront iteration is two launches:
//
//   k_shade          every live slot: resolve the pending next-event estimate, shade the hit (getSurfacePoint, emission +
//                    MIS, Russian roulette), then the next bounce: light sampling + BSDF sampling (:161-221); emits the next
//                    extension ray in place and a shadow ray (slot index into the shadow queue).  A slot whose path ENDS
//                    here is finished in the same launch: weight * C goes to the sample's entry of the result window
//                    (writeResult; sensor->add itself, in pass order, is k_fold's), the wave's queue hands the slot its next
//                    (pixel, pass), and the new samples of the workgroup are started by its FIRST lanes, compacted
//                    (Job::kernel's camera half, :100-120: the 50-draw stream seeding runs on full waves).
//   k_trace_ws       every slot with a ray in flight (state flag) + the shadow-ray queue, one launch (pt_trace_ws.hip)
//
// (Rounds 1-2 ran the restart as a third kernel, k_regen, over a queue of finished slots, and kept the pixel's accumulator in
// the slot; DESIGN.md section 7 has the before / after of both.)
//
// The shadow queue is a slot-index list in HBM, 16 regions (one per blockIdx % 16), filled by wave ballot + popcount prefix
// with ONE atomic per workgroup on a counter that has its own 128-byte line.  All path state is SoA in 16-byte records so a
// wave's loads are 1 KiB bursts; the state loads are issued together at the top, the material / light / spectrum tables live in LDS.
#pragma once
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include <string>

#include "pt_bsdf.h"
#include "pt_bsdf_multi.h"
#include "pt_kernels.h"
#include "pt_tex.h"

namespace slrhip {

enum : uint32_t {
    ST_IDLE = 0,            // no more samples for this slot
    ST_REGEN = 1,           // start the next sample: every slot after k_reset_slots; in the tail kernel also "a path just ended" (bit 15)
    ST_FIRST_HIT = 2,       // camera ray in flight        (PathTracingRenderer.cpp:147)
    ST_NEXT_HIT = 3,        // BSDF-sampled ray in flight  (:225)
    ST_FINISH = 4           // path ended while a shadow ray was still pending
};
// flags word: [2:0] state | [9:3] pathLength | [12] wlFlags.LambdaIsSelected | [13] previous direction was delta
//             | [14] shadow ray pending | [15] a finished path awaits accumulation | [19:16] selectedLambda
#define F_STATE(f) ((f) & 7u)
#define F_PATHLEN(f) (((f) >> 3) & 127u)
#define F_WL(f) (((f) >> 16) & 15u)
#define F_WLSEL(f) (((f) >> 12) & 1u)
#define F_DELTA(f) (((f) >> 13) & 1u)
#define F_SHADOW(f) (((f) >> 14) & 1u)
#define F_HASPATH(f) (((f) >> 15) & 1u)
#define F_SPVALID(f) (((f) >> 10) & 1u)      // spectral mode: the path's radiance sum in HBM has been written since the path began
#define F_MAKE(state, len, wl, wlsel, delta, shadow) \
    ((state) | ((len) << 3) | ((wl) << 16) | ((wlsel) << 12) | ((delta) << 13) | ((shadow) << 14))

static const int kShadeBlock = 256;

// Spectrum-valued path state in HBM.  RGB: one float4 per slot, the scalar that travels with it in .w.
// Spectral: four float4 planes per array (plane p of slot i at [p * numSlots + i], so every plane is a coalesced
// stream) and the scalar in an array of its own.
template <class S> struct SpecIO;
template <> struct SpecIO<RGB> {
    static __device__ __forceinline__ void load(const float4* a, const float* /*scalars*/, uint32_t slot, uint32_t /*n*/, RGB& v, float& w) {
        const float4 q = a[slot];
        v = RGB(q.x, q.y, q.z); w = q.w;
    }
    static __device__ __forceinline__ void store(float4* a, float* /*scalars*/, uint32_t slot, uint32_t /*n*/, const RGB& v, float w) {
        a[slot] = make_float4(v.r, v.g, v.b, w);
    }
};
template <> struct SpecIO<Spec16> {
    static __device__ __forceinline__ void load(const float4* a, const float* scalars, uint32_t slot, uint32_t n, Spec16& v, float& w) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const float4 q = a[(size_t)p * n + slot];
            v.c[4 * p] = q.x; v.c[4 * p + 1] = q.y; v.c[4 * p + 2] = q.z; v.c[4 * p + 3] = q.w;
        }
        w = scalars ? scalars[slot] : 0.0f;
    }
    static __device__ __forceinline__ void store(float4* a, float* scalars, uint32_t slot, uint32_t n, const Spec16& v, float w) {
#pragma unroll
        for (int p = 0; p < 4; ++p) a[(size_t)p * n + slot] = make_float4(v.c[4 * p], v.c[4 * p + 1], v.c[4 * p + 2], v.c[4 * p + 3]);
        if (scalars) scalars[slot] = w;
    }
};

// Material access per mode
template <class S> struct MatIO;
template <> struct MatIO<RGB> {
    template <bool LDS>
    static __device__ __forceinline__ Mat<RGB> load(const DevScene& sc, const float4* ldsMats, uint32_t idx, float) {
        return loadMat(LDS ? reinterpret_cast<const DevMaterial*>(ldsMats) + idx : sc.materials + idx);
    }
    template <bool LDS>
    static __device__ __forceinline__ RGB emittance(const DevScene& sc, const float4* ldsMats, uint32_t idx, float) {
        return loadEmittance(LDS ? reinterpret_cast<const DevMaterial*>(ldsMats) + idx : sc.materials + idx);
    }
};
// spectral mode, LDS tables: [DevMaterialS x kLdsMaterials][DevSpectrum x kLdsSpectra][sample pool, kLdsPoolFloats floats]
__device__ __forceinline__ const float* ldsPoolOf(const float4* ldsMats) {
    return reinterpret_cast<const float*>(ldsMats + 2 * kLdsMaterials + 2 * kLdsSpectra);
}
template <class S> struct MatIOSpectral {
    // spectral mode: the LDS table holds the DevMaterialS records (2 x float4 each) followed by the DevSpectrum records
    template <bool LDS>
    static __device__ __forceinline__ Mat<S> load(const DevScene& sc, const float4* ldsMats, uint32_t idx, float wlOffset) {
        const DevMaterialS* mats = LDS ? reinterpret_cast<const DevMaterialS*>(ldsMats) : sc.materialsS;
        const DevSpectrum* spectra = LDS ? reinterpret_cast<const DevSpectrum*>(ldsMats + 2 * kLdsMaterials) : sc.spectra;
        return loadMatSpectral<S>(mats, idx, spectra, LDS ? ldsPoolOf(ldsMats) : sc.spectrumPool, wlOffset);
    }
    template <bool LDS>
    static __device__ __forceinline__ S emittance(const DevScene& sc, const float4* ldsMats, uint32_t idx, float wlOffset) {
        const DevMaterialS* mats = LDS ? reinterpret_cast<const DevMaterialS*>(ldsMats) : sc.materialsS;
        const DevSpectrum* spectra = LDS ? reinterpret_cast<const DevSpectrum*>(ldsMats + 2 * kLdsMaterials) : sc.spectra;
        return evalSpectrum<S>(spectra, LDS ? ldsPoolOf(ldsMats) : sc.spectrumPool, mats[idx].spec[3], wlOffset);
    }
};
template <> struct MatIO<Spec16> : MatIOSpectral<Spec16> {};

// Append `slot` to the workgroup's region of the shadow queue: wave ballots + popcount prefixes, the four wave counts meet
// in LDS, ONE atomic per workgroup (on the region's own counter line).
struct PushLds {
    uint32_t count[4];
    uint32_t base;
};
__device__ __forceinline__ void blockPush(PushLds& pl, bool emit, uint32_t slot, uint32_t* queue, uint32_t* counters /* set being filled */,
                                          uint32_t shardCapacity, uint32_t* errorWord) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t shard = blockIdx.x % kShards;
    const uint64_t m = __ballot(emit);
    if (lane == 0) pl.count[wave] = (uint32_t)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t total = pl.count[0] + pl.count[1] + pl.count[2] + pl.count[3];
        pl.base = total ? atomicAdd(&counters[(Q_SHADOW * kShards + shard) * kCounterStride], total) : 0u;
        if (pl.base + total > shardCapacity) atomicOr(errorWord, ERR_QUEUE_OVERFLOW);     // cannot happen: a region holds every slot of its blocks
    }
    __syncthreads();
    if (emit) {
        uint32_t off = pl.base;
        for (uint32_t w = 0; w < wave; ++w) off += pl.count[w];
        queue[(size_t)shard * shardCapacity + off + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = slot;
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

// ---- environment sphere (RGB mode) ---------------------------------------------------------------------------------------
// RegularConstantContinuous1D::sample, Core/distributions.cpp:168-179
__device__ __forceinline__ float sampleContinuous1D(const float* cdf, const float* pdfTable, uint32_t numValues, float u, float* pdf) {
    int idx = (int)numValues;
    uint32_t p2 = numValues;
    p2 |= p2 >> 1; p2 |= p2 >> 2; p2 |= p2 >> 4; p2 |= p2 >> 8; p2 |= p2 >> 16; p2 -= p2 >> 1;       // prevPowerOf2
    for (int d = (int)p2; d > 0; d >>= 1)
        if (idx - d > 0 && cdf[idx - d] >= u) idx -= d;
    --idx;
    *pdf = pdfTable[idx];
    float t = (u - cdf[idx]) / (cdf[idx + 1] - cdf[idx]);
    return ((float)idx + t) / (float)numValues;
}
// ImageSpectrumTexture::evaluate (Textures/image_textures.cpp:13-20,57-63) + IBLEmission::emittance (IBLEmission.cpp:15-17)
__device__ __forceinline__ RGB envEmittance(const DevScene& sc, float tcU, float tcV) {
    float u = fmodf(tcU, 1.0f);
    float v = fmodf(tcV, 1.0f);
    u += u < 0 ? 1.0f : 0.0f;
    v += v < 0 ? 1.0f : 0.0f;
    uint32_t px = min((uint32_t)((float)sc.envWidth * u), sc.envWidth - 1);
    uint32_t py = min((uint32_t)((float)sc.envHeight * v), sc.envHeight - 1);
    const float* t = sc.envTexels + ((size_t)py * sc.envWidth + px) * 3;
    return ((float)kPi * RGB(t[0], t[1], t[2])) * sc.envScale;
}
// UpsampledContinuousSpectrumTemplate::evaluate with its grid look-up (SpectrumTypes.h:239-339) at run time: the spectral
// build's environment texels are (u, v, s) (image_textures.cpp:23-32).  Cell search and weights are per path (scalar), the
// 16-wavelength interpolation goes through S::make like every other spectrum.
template <class S>
__device__ __forceinline__ S evaluateUpsampledRuntime(const DevScene& sc, float u, float v, float scale, float wlOffset) {
    if (u < 0.0f || u >= (float)sc.gridWidth || v < 0.0f || v >= (float)sc.gridHeight) return S();
    const int32_t ui = (int32_t)u, vi = (int32_t)v;
    const uint8_t* cell = sc.gridCells + (size_t)(ui + (int32_t)sc.gridWidth * vi) * 8;
    const uint2 cw = *reinterpret_cast<const uint2*>(cell);                      // inside, num_points, idx[0..5]
    const uint32_t inside = cw.x & 0xFFu, numPoints = (cw.x >> 8) & 0xFFu;
    const uint32_t idx6[6] = {(cw.x >> 16) & 0xFFu, cw.x >> 24, cw.y & 0xFFu, (cw.y >> 8) & 0xFFu, (cw.y >> 16) & 0xFFu, cw.y >> 24};
    uint32_t used0 = 255u, used1 = 255u, used2 = 255u, used3 = 255u;
    float w0 = 0, w1 = 0, w2 = 0, w3 = 0;
    if (inside) {
        const float s = u - (float)ui, t = v - (float)vi;
        w0 = (1 - s) * (1 - t); w1 = s * (1 - t); w2 = (1 - s) * t; w3 = s * t;
        used0 = idx6[0]; used1 = idx6[1]; used2 = idx6[2]; used3 = idx6[3];
    }
    else {
        const float2* uv = reinterpret_cast<const float2*>(sc.pointUV);
        const float2 p0 = uv[idx6[0]], p1 = uv[idx6[1]];
        const float ex = u - p0.x, ey = v - p0.y;
        float e0x = p1.x - p0.x, e0y = p1.y - p0.y;
        float uu = e0x * ey - ex * e0y;
        for (uint32_t i = 1; i < numPoints; ++i) {
            const uint32_t k = i % (numPoints - 1) + 1;
            // idx6[k] with a run-time k: select chain over the six bytes
            const uint32_t idx = k == 1 ? idx6[1] : k == 2 ? idx6[2] : k == 3 ? idx6[3] : k == 4 ? idx6[4] : idx6[5];
            const uint32_t idxI = i == 1 ? idx6[1] : i == 2 ? idx6[2] : i == 3 ? idx6[3] : i == 4 ? idx6[4] : idx6[5];
            const float2 pk = uv[idx];
            const float e1x = pk.x - p0.x, e1y = pk.y - p0.y;
            const float vv = ex * e1y - e1x * ey;
            const float area = e0x * e1y - e1x * e0y;
            const float bu = uu / area, bv = vv / area;
            const float bw = 1.0f - bu - bv;
            if ((double)bu < -1e-6 || (double)bv < -1e-6 || (double)bw < -1e-6) {
                uu = -vv;
                e0x = e1x;
                e0y = e1y;
                continue;
            }
            w0 = bu; w1 = bv; w2 = bw;
            used0 = idx; used1 = idxI; used2 = idx6[0];
            break;
        }
    }
    if (used0 == 255u) return S();
    const uint32_t nw = 95;
    const float* t0 = sc.pointSpectrum + (size_t)used0 * nw;
    const float* t1 = sc.pointSpectrum + (size_t)used1 * nw;
    const float* t2 = sc.pointSpectrum + (size_t)used2 * nw;
    const float* t3 = sc.pointSpectrum + (size_t)(used3 == 255u ? used0 : used3) * nw;
    const bool four = used3 != 255u;
    S ret = S::make([&](int i) {
        float p = (wavelengthOf(i, wlOffset) - 360.0f) / (830.0f - 360.0f);
        float sBinF = p * (float)(nw - 1);
        uint32_t sBin = (uint32_t)sBinF;
        uint32_t sBinNext = (sBin + 1 < nw) ? (sBin + 1) : (nw - 1);
        float t = sBinF - (float)sBin;
        float r = 0.0f;
        r += w0 * (t0[sBin] * (1 - t) + t0[sBinNext] * t);
        r += w1 * (t1[sBin] * (1 - t) + t1[sBinNext] * t);
        r += w2 * (t2[sBin] * (1 - t) + t2[sBinNext] * t);
        if (four) r += w3 * (t3[sBin] * (1 - t) + t3[sBinNext] * t);
        return r;
    });
    return ret * scale;
}
template <class S>
__device__ __forceinline__ S envEmittanceSpectral(const DevScene& sc, float tcU, float tcV, float wlOffset) {
    float u = fmodf(tcU, 1.0f);
    float v = fmodf(tcV, 1.0f);
    u += u < 0 ? 1.0f : 0.0f;
    v += v < 0 ? 1.0f : 0.0f;
    uint32_t px = min((uint32_t)((float)sc.envWidth * u), sc.envWidth - 1);
    uint32_t py = min((uint32_t)((float)sc.envHeight * v), sc.envHeight - 1);
    const float* t = sc.envTexels + ((size_t)py * sc.envWidth + px) * 3;
    const float kEqualEnergyReflectance = 0.009355121400914532f;                 // Upsampling::EqualEnergyReflectance
    const S tex = evaluateUpsampledRuntime<S>(sc, t[0], t[1], t[2] / kEqualEnergyReflectance, wlOffset);
    return ((float)kPi * tex) * sc.envScale;                                     // IBLEmission::emittance
}
template <class S> __device__ __forceinline__ S envEmittanceS(const DevScene& sc, float u, float v, float wlOffset);
template <> __device__ __forceinline__ RGB envEmittanceS<RGB>(const DevScene& sc, float u, float v, float) { return envEmittance(sc, u, v); }
template <> __device__ __forceinline__ Spec16 envEmittanceS<Spec16>(const DevScene& sc, float u, float v, float o) { return envEmittanceSpectral<Spec16>(sc, u, v, o); }
// InfiniteSphereSurfaceObject::evaluateAreaPDF, SurfaceObject.cpp:217-222 (RegularConstantContinuous2D::evaluatePDF :218-224)
__device__ __forceinline__ float envAreaPDF(const DevScene& sc, float phi, float theta) {
    float d0 = (float)((double)phi / (2 * kPi)), d1 = (float)((double)theta / kPi);
    uint32_t idx1D = min((uint32_t)((float)sc.envMapHeight * d1), sc.envMapHeight - 1);
    // the reference indexes with no clamp; phi / 2pi can round up to 1.0f, which would read past the table
    uint32_t iTop = min((uint32_t)(int32_t)(d1 * (float)sc.envMapHeight), sc.envMapHeight - 1);
    uint32_t iRow = min((uint32_t)(int32_t)(d0 * (float)sc.envMapWidth), sc.envMapWidth - 1);
    float uvPDF = sc.envTopPDF[iTop] * sc.envRowPDF[(size_t)idx1D * sc.envMapWidth + iRow];
    return (float)((double)uvPDF / (2 * kPi * kPi * (double)slrSin(theta)));
}

// SampledSpectrumSum sp of Job::contribution (PathTracingRenderer.cpp:141): see the note at its use in logicSlot.
template <class S> struct SpAcc;
// A path starts with sp = 0 (and alpha = 1, no previous PDF): startSample does not write those records, the first visit
// (state FIRST_HIT) supplies the values instead of what it loaded.  RGB keeps the pair in registers and always stores it,
// so it is valid from then on; the spectral variants update HBM only when a contribution arrives and track that in `valid`
// (flag bit 10), which writeResult consults before reading the sum.
#ifndef SLR_SP_LAZY
#define SLR_SP_LAZY 1      // 0 (variant builds): read and write the RGB radiance sum, its compensation and the pending light sample at every visit
#endif
template <> struct SpAcc<RGB> {
    // The Kahan pair is read only once the path has written it (flag bit 10) and written back only by a visit that added to it;
    // the pending light sample only by a visit that has one (flag bit 14).  Most visits do neither: a first hit adds only on an
    // emitter, a later one when its shadow ray came back visible or the ray found a light.
    RGB r, c, nee;
    bool valid, changed;
    __device__ __forceinline__ void begin(const PathBuffers& pb, uint32_t slot, uint32_t n, uint32_t flags, bool /*resolves*/) {
        float unused;
        valid = !SLR_SP_LAZY || (F_STATE(flags) != ST_FIRST_HIT && F_SPVALID(flags));
        changed = !SLR_SP_LAZY;
        r = RGB(); c = RGB(); nee = RGB();
        if (valid) {
            SpecIO<RGB>::load(pb.spR, nullptr, slot * pb.spStride, n * pb.spStride, r, unused);
            SpecIO<RGB>::load(pb.spC, nullptr, slot * pb.spStride, n * pb.spStride, c, unused);
        }
        if (!SLR_SP_LAZY || F_SHADOW(flags)) SpecIO<RGB>::load(pb.nee, nullptr, slot, n, nee, unused);
    }
    __device__ __forceinline__ void startPath(bool first, uint32_t) { if (first) { r = RGB(); c = RGB(); } }
    __device__ __forceinline__ uint32_t validBits() const { return (valid || changed) ? 1u << 10 : 0u; }
    __device__ __forceinline__ RGB total() const { return r; }
    __device__ __forceinline__ void addPendingNee(const PathBuffers&, uint32_t, uint32_t) { kahanAdd(r, c, nee); changed = true; }
    __device__ __forceinline__ void add(const PathBuffers&, uint32_t, uint32_t, const RGB& v) { kahanAdd(r, c, v); changed = true; }
    __device__ __forceinline__ void end(const PathBuffers& pb, uint32_t slot, uint32_t n, bool pathContinues) {
        if (!changed) return;
        SpecIO<RGB>::store(pb.spR, nullptr, slot * pb.spStride, n * pb.spStride, r, 0.0f);
        if (pathContinues) SpecIO<RGB>::store(pb.spC, nullptr, slot * pb.spStride, n * pb.spStride, c, 0.0f);       // a finished path only hands over the sum
    }
};
#ifndef SLR_SPEC_PREFETCH
#define SLR_SPEC_PREFETCH 1      // 0 (variant builds): the pending light sample and the radiance sum are fetched when the resolve runs
#endif
template <> struct SpAcc<Spec16> {
    // The Kahan pair (2 x 64 B) and the pending light sample (64 B) stay in HBM and are updated in place when a contribution
    // arrives.  The one contribution that is known before the visit starts — the pending light sample of a slot whose shadow ray
    // came back visible (flag bit 14 + the visibility word, both read before the state loads are issued) — has its twelve 16-byte
    // operands requested WITH the state loads, so that the resolve costs no memory round trip of its own.
    bool valid;
    float4 pn[4], pr[4], pc[4];
    bool fetched;
    __device__ __forceinline__ void begin(const PathBuffers& pb, uint32_t slot, uint32_t n, uint32_t flags, bool resolves) {
        fetched = false;
        if (SLR_SPEC_PREFETCH && resolves) {
            const bool had = F_STATE(flags) != ST_FIRST_HIT && F_SPVALID(flags);
            const float4 zero = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                pn[p] = pb.nee[(size_t)p * n + slot];
                pr[p] = had ? pb.spR[((size_t)p * n + slot) * pb.spStride] : zero;
                pc[p] = had ? pb.spC[((size_t)p * n + slot) * pb.spStride] : zero;
            }
            fetched = true;
        }
    }
    __device__ __forceinline__ void startPath(bool first, uint32_t flags) { valid = !first && F_SPVALID(flags); }
    __device__ __forceinline__ uint32_t validBits() const { return valid ? 1u << 10 : 0u; }
    __device__ __forceinline__ Spec16 total() const { return Spec16(); }      // the sum is in HBM (flag bit 10 says whether it was ever written)
    __device__ __forceinline__ void add(const PathBuffers& pb, uint32_t slot, uint32_t n, const Spec16& v) {
        // plane by plane: 4 components of the Kahan pair in flight at a time
        const float4 zero = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            float4 r = zero, c = zero;
            if (valid) { r = pb.spR[((size_t)p * n + slot) * pb.spStride]; c = pb.spC[((size_t)p * n + slot) * pb.spStride]; }
            kahanAdd(r.x, c.x, v.c[4 * p]); kahanAdd(r.y, c.y, v.c[4 * p + 1]);
            kahanAdd(r.z, c.z, v.c[4 * p + 2]); kahanAdd(r.w, c.w, v.c[4 * p + 3]);
            pb.spR[((size_t)p * n + slot) * pb.spStride] = r;
            pb.spC[((size_t)p * n + slot) * pb.spStride] = c;
        }
        valid = true;
    }
    __device__ __forceinline__ void addPendingNee(const PathBuffers& pb, uint32_t slot, uint32_t n) {
        const float4 zero = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            float4 v, r = zero, c = zero;
            if (fetched) { v = pn[p]; r = pr[p]; c = pc[p]; }
            else {
                v = pb.nee[(size_t)p * n + slot];
                if (valid) { r = pb.spR[((size_t)p * n + slot) * pb.spStride]; c = pb.spC[((size_t)p * n + slot) * pb.spStride]; }
            }
            kahanAdd(r.x, c.x, v.x); kahanAdd(r.y, c.y, v.y); kahanAdd(r.z, c.z, v.z); kahanAdd(r.w, c.w, v.w);
            pb.spR[((size_t)p * n + slot) * pb.spStride] = r;
            pb.spC[((size_t)p * n + slot) * pb.spStride] = c;
        }
        valid = true;
    }
    __device__ __forceinline__ void end(const PathBuffers&, uint32_t, uint32_t, bool) {}
};

template <bool SPECTRAL>
struct ShadeLds {
    // RGB: DevMaterial = 5 x float4 each.  Spectral: DevMaterialS (2 x float4 each), DevSpectrum (2 x float4 each), then the
    // spectrum sample pool: evaluating a spectrum is 16 x 2 scattered 16-byte reads per slot, which the vector L1 serves at
    // one cache line per clock (measured: the look-ups were 1.0-1.6 ms of a 1.4-2.0 ms launch); LDS serves them in banks
    float4 mats[SPECTRAL ? kLdsMaterials * 2 + kLdsSpectra * 2 + kLdsPoolFloats / 4 : kLdsMaterials * 5];
    float4 lights[kLdsLights * 9];         // LightTri   = 9 x float4
    float lightPMF[kLdsLights];
    float lightCDF[kLdsLights + 4];        // kLdsLights + 1 entries, padded to whole float4
};
static_assert(kLdsLights % 4 == 0, "the light PMF is staged as whole float4");

// Stage the shading tables of the scene (DevScene::shadeTables: one packed array, segments in the order of ShadeLds' members) in
// LDS.  One flat copy: every thread requests its (up to four) float4 first and writes them afterwards, so the whole table costs
// ONE memory round trip — the per-table loops this replaces waited for each table in turn, four to six dependent round trips at
// the top of every workgroup.  The caller's barrier publishes the tables.
template <bool SPECTRAL>
__device__ __forceinline__ void stageShadeTables(const DevScene& sc, ShadeLds<SPECTRAL>& lds) {
    constexpr uint32_t kSegments = SPECTRAL ? 6u : 4u;
    // float4 index of every segment's first element inside ShadeLds
    constexpr uint32_t matsF4 = SPECTRAL ? kLdsMaterials * 2 + kLdsSpectra * 2 + kLdsPoolFloats / 4 : kLdsMaterials * 5;
    const uint32_t dst[6] = {0u,
                             SPECTRAL ? (uint32_t)(kLdsMaterials * 2) : matsF4,
                             SPECTRAL ? (uint32_t)(kLdsMaterials * 2 + kLdsSpectra * 2) : matsF4 + kLdsLights * 9,
                             SPECTRAL ? matsF4 : matsF4 + kLdsLights * 9 + kLdsLights / 4,
                             matsF4 + kLdsLights * 9,
                             matsF4 + kLdsLights * 9 + kLdsLights / 4};
    float4* out = reinterpret_cast<float4*>(&lds);
    const uint32_t total = sc.tableEnd[kSegments - 1];
    for (uint32_t base = threadIdx.x; base < total; base += 4 * kShadeBlock) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t i = base + u * kShadeBlock;
            v[u] = sc.shadeTables[min(i, total - 1u)];      // unconditional (clamped), so that the four requests leave back to back
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) asm volatile("" : "+v"(v[u].x), "+v"(v[u].y), "+v"(v[u].z), "+v"(v[u].w));      // ... and are not sunk into the guarded writes below
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t i = base + u * kShadeBlock;
            if (i >= total) continue;
            uint32_t begin = 0, to = dst[0];
#pragma unroll
            for (uint32_t k = 0; k + 1 < kSegments; ++k)
                if (i >= sc.tableEnd[k]) { begin = sc.tableEnd[k]; to = dst[k + 1]; }
            out[to + (i - begin)] = v[u];
        }
    }
}

// Occupancy floor of k_shade (waves per SIMD): the register allocator spills to scratch to stay under 512 / N registers.
// Measured on the spectral GGX scene: the 330-register allocation (1 wave per SIMD, 63 % of its cycles waiting on memory,
// PMC) ran at 1 393 us per launch; held to 256 registers (300 B of scratch per lane, 2 waves) it runs at 765 us.
// A floor of 3 waves costs the same kernel 988 us (more scratch than the extra wave hides), but pays on the variant
// without the glossy lobes (190 registers: 832 -> 768 us at <= 170).  The RGB variants allocate 116-161 registers
// (3-4 waves) on their own; forcing 5 or 6 waves slows them (220 -> 326 / 505 us), so their floor is left below that.
#ifndef SLR_WAVES_SPECTRAL_GLOSSY
#define SLR_WAVES_SPECTRAL_GLOSSY 2
#endif
#ifndef SLR_WAVES_SPECTRAL
#define SLR_WAVES_SPECTRAL 3
#endif
#ifndef SLR_WAVES_RGB
#define SLR_WAVES_RGB 2
#endif
// The texture behind a material's spectrum slot, evaluated at the hit's texture coordinate.
//   CheckerBoardSpectrumTexture (checker_board_textures.h:21-24): one of its two constant spectra.  RGB build: their values are in
//   the texture record; spectral build: evaluated at the path's wavelengths like any other constant (tables in HBM: textured
//   scenes run the kernel variant without LDS tables).
//   ImageSpectrumTexture (Textures/image_textures.cpp:13-79): the nearest texel; RGB build: its three floats; spectral build:
//   UpsampledContinuousSpectrum(u, v, s / EqualEnergyReflectance) with its grid look-up at run time (:23-32), like the environment map.
template <class S> struct TexSpectrum {
    static __device__ __forceinline__ S eval(const DevScene& sc, const DevTexture& t, float texU, float texV, float wlOffset) {
        if (t.kind == SLRHIP_TEXTURE_IMAGE_SPECTRUM) {
            const float* x = sc.texTexels + (size_t)imageTexel(t, texU, texV) * 3;
            const float kEqualEnergyReflectance = 0.009355121400914532f;                 // Upsampling::EqualEnergyReflectance
            return evaluateUpsampledRuntime<S>(sc, x[0], x[1], x[2] / kEqualEnergyReflectance, wlOffset);
        }
        return evalSpectrum<S>(sc.spectra, sc.spectrumPool, checkerIndex(t, texU, texV) ? t.spec1 : t.spec0, wlOffset);
    }
};
template <> struct TexSpectrum<RGB> {
    static __device__ __forceinline__ RGB eval(const DevScene& sc, const DevTexture& t, float texU, float texV, float) {
        if (t.kind == SLRHIP_TEXTURE_IMAGE_SPECTRUM) {
            const float* x = sc.texTexels + (size_t)imageTexel(t, texU, texV) * 3;
            return RGB(x[0], x[1], x[2]);
        }
        return checkerIndex(t, texU, texV) ? RGB(t.rgb1[0], t.rgb1[1], t.rgb1[2]) : RGB(t.rgb0[0], t.rgb0[1], t.rgb0[2]);
    }
};

// Textured scenes: replaces the constants of the textured spectrum slots of material record `matIndex` by the textures' values
// at texture coordinate (texU, texV) and returns the material's normal map (-1 = none).
template <class S>
__device__ __forceinline__ int32_t texturizeMat(const DevScene& sc, Mat<S>& mm, uint32_t matIndex, float texU, float texV, float wlOffset) {
    mm.type &= ~kMatTexturedBit;
    const int4 mt = *reinterpret_cast<const int4*>(sc.matTex + matIndex);
    if (mt.x >= 0) { const DevTexture t = loadTexture(sc.textures, (uint32_t)mt.x); mm.a = 1.0f * TexSpectrum<S>::eval(sc, t, texU, texV, wlOffset); }
    if (mt.y >= 0) { const DevTexture t = loadTexture(sc.textures, (uint32_t)mt.y); mm.b = TexSpectrum<S>::eval(sc, t, texU, texV, wlOffset); }
    if (mt.z >= 0) { const DevTexture t = loadTexture(sc.textures, (uint32_t)mt.z); mm.c = TexSpectrum<S>::eval(sc, t, texU, texV, wlOffset); }
    return mt.w;
}

// One visit of the shade kernel to one slot: everything between the state loads and the state stores.  A device function so
// that the tail kernel (pt_tail_kernels.h) runs the very same code on the last paths of a render call.
// FUSED (k_shade): a path that ends here is accumulated and restarted by the caller in the same launch — the slot's flags and
// radiance sum are handed back (`flags`, `radiance`; RGB keeps the sum in registers) instead of being stored.  Not FUSED
// (k_tail): the slot is left in ST_REGEN with bit 15 set and its sum in HBM, for the lane's next turn.
// The records of a slot that a visit reads whatever the slot's state is.  k_shade requests them at the very top of the kernel,
// together with the slot's flags and the tables it stages in LDS, so that they travel in ONE memory round trip before the
// workgroup's barrier instead of in three dependent ones (flags -> tables -> state); the tail kernel requests them just before
// its visit.
template <class S>
struct SlotLoads {
    uint4 r4;                  // xorshift128 state
    S alpha;                   // path throughput
    float pdfPrev;
    float4 h, o4, d4;          // hit record, ray origin, ray direction
    int32_t hitInstance;       // instanced scenes: the TransformedSurfaceObject of the hit
    float wlOffset;            // spectral mode: the sample's wavelength offset (sample header)
    __device__ __forceinline__ void issue(const DevScene& sc, const PathBuffers& pb, const RenderParams& rp, uint32_t slot) {
        r4 = pb.rng[(size_t)slot * pb.hdrStride];
        SpecIO<S>::load(pb.alpha, pb.pdfPrev, slot, rp.numSlots, alpha, pdfPrev);
        h = pb.hit[slot];
        hitInstance = sc.instances ? pb.hitInstance[slot] : -1;
        o4 = pb.rayOrg[(size_t)slot * pb.rayStride];
        d4 = pb.rayDir[(size_t)slot * pb.rayStride];
        wlOffset = S::N == 3 ? 0.0f : __uint_as_float(pb.hdr[(size_t)slot * pb.hdrStride].z);
    }
};

template <class S, bool LDS_TABLES, bool MF, bool MULTI, bool TEX, bool FUSED>
__device__ __forceinline__ void logicSlot(const DevScene& sc, const PathBuffers& pb, const RenderParams& rp, const ShadeLds<S::N != 3>& lds,
                                          const float* lightPMF, const float* lightCDF, uint32_t slot, const SlotLoads<S>& in, uint32_t& flags, uint32_t vis,
                                          S& radiance, bool& emitExt, bool& emitShadow, bool& emitRegen) {
    constexpr bool leader = true;
    // ---- the state loads that depend on the flags; the others were requested by the caller (SlotLoads) -----------
    const uint4 r4 = in.r4;
    // The path's radiance sum (Kahan pair) and the pending light sample: RGB keeps them in registers (3 x 16 B,
    // requested with everything else); in spectral mode they are 3 x 64 B that most visits never touch, so they
    // stay in HBM and SpAcc updates them in place when a contribution actually arrives.
    S alpha = in.alpha;
    SpAcc<S> sp;
    float bsdfPDFprev = in.pdfPrev;
    sp.begin(pb, slot, rp.numSlots, flags, F_SHADOW(flags) && vis);
    const float4 h = in.h;
    const int32_t hitInstance = in.hitInstance;
    const float4 o4 = in.o4, d4 = in.d4;
    const float wlOffset = in.wlOffset;

    const uint32_t state = F_STATE(flags);
    if (state == ST_FIRST_HIT || state == ST_NEXT_HIT || state == ST_FINISH) {
        // a path's first visit: throughput 1, no previous PDF, empty radiance sum (startSample writes none of them)
        if (state == ST_FIRST_HIT) { alpha = S(1.0f); bsdfPDFprev = 0.0f; }
        sp.startPath(state == ST_FIRST_HIT, flags);
        Rng rng;
        rng.s0 = r4.x; rng.s1 = r4.y; rng.s2 = r4.z; rng.s3 = r4.w;
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
        if (F_SHADOW(flags) && vis) sp.addPendingNee(pb, slot, rp.numSlots);      // (its operands were requested with the state loads: SpAcc::begin)

        // ---- 2. the hit that just came back ------------------------------------------------------------
        Mat<S> m;
        float texU = 0.0f, texV = 0.0f;
        int32_t normalMap = -1;
        const auto texturize = [&](Mat<S>& mm, uint32_t matIndex) -> int32_t { return texturizeMat<S>(sc, mm, matIndex, texU, texV, wlOffset); };
        if (!hasHit) {
            finish = true;                      // ST_FINISH, or a miss: :148 return Zero / :226 break
            if (state != ST_FINISH && sc.hasEnv) {
                // the ray left the scene: Scene::intersect falls through to the environment sphere (SurfaceObject.cpp:411-414).
                // InfiniteSphere::intersect / getSurfacePoint (Surface/InfiniteSphere.cpp:34-59), Vector3::toPolarYUp (Vector3.h:72-75)
                float theta = slrAcos(fminf(1.0f, fmaxf(-1.0f, rayDir.y)));
                float phi = fmodf((float)((double)slrAtan2(-rayDir.x, rayDir.z) + 2 * kPi), (float)(2 * kPi));
                float texU = (float)((double)phi / (2 * kPi)), texV = (float)((double)theta / kPi);
                // emittance x IBLEDF::evaluate = 1 / pi (EDFs/IBLEDF.cpp:19-23)
                S Le = envEmittanceS<S>(sc, texU, texV, wlOffset) * S((float)(1.0 / kPi));
                if (state == ST_FIRST_HIT) {
                    sp.add(pb, slot, rp.numSlots, alpha * Le);             // :152-157, atInfinity -> return sp
                }
                else {
                    // implicit light sampling :232-250; the path ends at infinity before Russian roulette
                    float sumImps = sc.aggImportance + 1.0f;
                    float lightProb = 1.0f / sumImps;                        // Scene::evaluateProb SurfaceObject.cpp:456-457
                    V3 gN = -rayDir;
                    float lightPDF = lightProb * envAreaPDF(sc, phi, theta) * 1.0f / absDot(rayDir, gN);
                    float MISWeight = 1.0f;
                    if (!F_DELTA(flags))
                        MISWeight = (bsdfPDFprev * bsdfPDFprev) / (lightPDF * lightPDF + bsdfPDFprev * bsdfPDFprev);
                    sp.add(pb, slot, rp.numSlots, alpha * Le * MISWeight);
                }
            }
        }
        else {
            // Triangle::getSurfacePoint, Surface/TriangleMesh.cpp:180-215.  isect.p = org + dir * t (:170) — of the ray the triangle
            // was tested with: inside an instance that is the LOCAL ray, invert(sampledTF) * ray (SurfaceObject.cpp:307-311), and the
            // whole surface point is taken to world space at the end (TransformedSurfaceObject::getSurfacePoint, below)
            const float* instMats = hitInstance >= 0 ? reinterpret_cast<const float*>(sc.instances + (size_t)hitInstance * 9u) : nullptr;
            if (instMats) surf.p = mulPoint(instMats + 16, rayOrg) + mulVector(instMats + 16, rayDir) * h.y;
            else surf.p = rayOrg + rayDir * h.y;
            surf.gNormal = V3(q3.w, q4.w, q5.w);
            surf.material = __float_as_uint(q0.w);
            surf.light = (int32_t)__float_as_uint(q1.w);
            surf.areaPDF = q2.w;
            m = MatIO<S>::template load<LDS_TABLES>(sc, lds.mats, surf.material, wlOffset);
            // the hit record carries Moller-Trumbore's (b1, b2); Intersection::u = b0 = 1 - b1 - b2 as Triangle::intersect
            // computes it (TriangleMesh.cpp:159), and getSurfacePoint re-derives ITS b2 from (u, v) (:190-191)
            const float b1 = h.z, b2hit = h.w;
            const float b0 = 1.0f - b1 - b2hit;
            const float b2 = 1.0f - b0 - b1;
            if constexpr (TEX) {
                // texCoord from the original barycentrics (TriangleMesh.cpp:160-161), then the textures of this material
                const float4 uvA = sc.triUV[(size_t)tri * 2], uvB = sc.triUV[(size_t)tri * 2 + 1];
                hitTexCoord(uvA, uvB, b1, b2hit, &texU, &texV);
                if (m.type & kMatTexturedBit) normalMap = texturize(m, surf.material);
            }
            surf.frame.z = normalize(b0 * xyz(q0) + b1 * xyz(q1) + b2 * xyz(q2));
            surf.frame.x = normalize(b0 * xyz(q3) + b1 * xyz(q4) + b2 * xyz(q5));
            const float dotNT = dot(surf.frame.z, surf.frame.x);
            if (fabsf(dotNT) >= 0.01f) surf.frame.x = normalize(surf.frame.x - dotNT * surf.frame.z);
            surf.frame.y = cross(surf.frame.z, surf.frame.x);
            if constexpr (TEX) {
                if (normalMap >= 0) {
                    // BumpSingleSurfaceObject::getSurfacePoint, Core/SurfaceObject.cpp:123-134
                    const DevTexture nt = loadTexture(sc.textures, (uint32_t)normalMap);
                    float uc, vc;
                    checkerNormalComponents(nt, texU, texV, &uc, &vc);
                    const V3 nLocal = normalize(V3(uc, vc, 1.0f));
                    const V3 tLocal = V3(1.0f, 0.0f, 0.0f) - dot(nLocal, V3(1.0f, 0.0f, 0.0f)) * nLocal;
                    const V3 bLocal = V3(0.0f, 1.0f, 0.0f) - dot(nLocal, V3(0.0f, 1.0f, 0.0f)) * nLocal;
                    const V3 tt = normalize(surf.frame.fromLocal(tLocal));
                    const V3 bb = normalize(surf.frame.fromLocal(bLocal));
                    const V3 nn = normalize(surf.frame.fromLocal(nLocal));
                    surf.frame.x = tt; surf.frame.y = bb; surf.frame.z = nn;
                }
            }
            if (instMats) {
                // *surfPt = sampledTF * *surfPt (SurfaceObject.cpp:329-336; SurfacePoint x StaticTransform, geometry.cpp:63-78): p as a
                // point, the geometric normal through the inverse transpose (Transform.h:47-52), the frame's axes as vectors, re-normalised
                surf.p = mulPoint(instMats, surf.p);
                surf.gNormal = normalize(mulNormal(instMats + 16, surf.gNormal));
                surf.frame.x = normalize(mulVector(instMats, surf.frame.x));
                surf.frame.y = normalize(mulVector(instMats, surf.frame.y));
                surf.frame.z = normalize(mulVector(instMats, surf.frame.z));
            }
            haveSurf = true;
            dirOut_sn = surf.frame.toLocal(-rayDir);
            if (surf.light >= 0) {
                S Le = MatIO<S>::template emittance<LDS_TABLES>(sc, lds.mats, surf.material, wlOffset) * S(diffuseEDF(dirOut_sn));
                if (state == ST_FIRST_HIT) {
                    sp.add(pb, slot, rp.numSlots, alpha * Le);             // :152-156
                }
                else {
                    // implicit light sampling with MIS :232-249
                    float lightProb = lightPMF[surf.light] * 1.0f;          // SurfaceObject.cpp:295-298, :78-80
                    if (sc.hasEnv) lightProb = sc.aggImportance / (sc.aggImportance + 1.0f) * lightProb;   // Scene::evaluateProb :459
                    float dist2 = sqLength(rayOrg - surf.p);
                    float lightPDF = lightProb * surf.areaPDF * dist2 / absDot(rayDir, surf.gNormal);
                    float MISWeight = 1.0f;
                    if (!F_DELTA(flags))
                        MISWeight = (bsdfPDFprev * bsdfPDFprev) / (lightPDF * lightPDF + bsdfPDFprev * bsdfPDFprev);
                    sp.add(pb, slot, rp.numSlots, alpha * Le * MISWeight);
                }
            }
            if (state == ST_NEXT_HIT) {
                // Russian roulette :254-258 (initY = importance(One) evaluated like the reference)
                float initY = importance(S(1.0f), wl);
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
                // SLRHIP_MATERIAL_MULTI: a MultiBSDF whose components are fetched from the material table on demand
                const auto loadComponent = [&](uint32_t idx) {
                    Mat<S> cm = MatIO<S>::template load<LDS_TABLES>(sc, lds.mats, idx, wlOffset);
                    if constexpr (TEX) { if (cm.type & kMatTexturedBit) (void)texturize(cm, idx); }
                    return cm;
                };
                const bool isMulti = MULTI && m.type == SLRHIP_MATERIAL_MULTI;
                MultiTree multiTree = {};
                if constexpr (MULTI) {
                    if (isMulti) {
                        multiTree = buildMultiTree<S>(decodeMulti(m), loadComponent);
                        type = multiType(multiTree, wlSel);
                    }
                }
                if (dtMatches(type, DT_WholeSphere | DT_NonDelta)) {
                    // Scene::selectLight, SurfaceObject.cpp:432-450 (+ aggregate :279-286)
                    float lightProb;
                    float uSel = rng.nextFloat();
                    bool pickEnv = false;
                    if (sc.hasEnv) {
                        float sumImps = sc.aggImportance + 1.0f;
                        float su = sumImps * uSel;
                        if (su < sc.aggImportance) uSel = uSel / (sc.aggImportance / sumImps);
                        else pickEnv = true;
                    }
                    float lu0 = rng.nextFloat();
                    float lu1 = rng.nextFloat();
                    V3 lp, lgn;
                    Frame lf;
                    float areaPDF;
                    S M;
                    float shadowTmax;
                    V3 sdir;
                    if (pickEnv) {
                        lightProb = 1.0f * (1.0f / (sc.aggImportance + 1.0f));
                        // InfiniteSphereSurfaceObject::sample, SurfaceObject.cpp:158-185
                        float topPDF, rowPDF;
                        float d1 = sampleContinuous1D(sc.envTopCDF, sc.envTopPDF, sc.envMapHeight, lu1, &topPDF);
                        uint32_t idx1D = min((uint32_t)((float)sc.envMapHeight * d1), sc.envMapHeight - 1);
                        float d0 = sampleContinuous1D(sc.envRowCDF + (size_t)idx1D * (sc.envMapWidth + 1), sc.envRowPDF + (size_t)idx1D * sc.envMapWidth,
                                                      sc.envMapWidth, lu0, &rowPDF);
                        float uvPDF = rowPDF * topPDF;
                        float phi = (float)((double)d0 * (2 * kPi));
                        float theta = (float)((double)d1 * kPi);
                        lp = V3(-slrSin(phi) * slrSin(theta), slrCos(theta), slrCos(phi) * slrSin(theta));
                        lgn = -lp;
                        lf.x = normalize(V3(-slrCos(phi), 0.0f, -slrSin(phi)));
                        lf.z = lgn;
                        lf.y = cross(lf.z, lf.x);
                        areaPDF = (float)((double)uvPDF / (2 * kPi * kPi * (double)slrSin(theta)));
                        M = envEmittanceS<S>(sc, (float)((double)phi / (2 * kPi)), (float)((double)theta / kPi), wlOffset);
                        sdir = normalize(lp);                              // Scene::testVisibility :421-423: [eps, FLT_MAX]
                        shadowTmax = 3.402823466e+38f;
                    }
                    else {
                        uint32_t li = selectLight(sc, lightCDF, lightPMF, uSel, &lightProb);
                        lightProb *= 1.0f;
                        if (sc.hasEnv) lightProb *= sc.aggImportance / (sc.aggImportance + 1.0f);
                        // Triangle::sample TriangleMesh.cpp:224-255
                        const float4* lt = (LDS_TABLES ? lds.lights : reinterpret_cast<const float4*>(sc.lightTris)) + (size_t)li * 9;
                        float4 l0 = lt[0], l1 = lt[1], l2 = lt[2], l3 = lt[3], l4 = lt[4], l5 = lt[5], l6 = lt[6], l7 = lt[7], l8 = lt[8];
                        float su1 = sqrtf(lu0);
                        float b0 = 1.0f - su1;
                        float b1 = lu1 * su1;
                        float b2 = 1.0f - b0 - b1;
                        lp = b0 * xyz(l0) + b1 * xyz(l1) + b2 * xyz(l2);
                        lgn = V3(l3.w, l4.w, l5.w);
                        lf.z = normalize(b0 * xyz(l3) + b1 * xyz(l4) + b2 * xyz(l5));
                        lf.x = normalize(b0 * xyz(l6) + b1 * xyz(l7) + b2 * xyz(l8));
                        lf.y = cross(lf.z, lf.x);
                        areaPDF = l2.w;
                        const uint32_t lmat = __float_as_uint(l1.w);
                        M = MatIO<S>::template emittance<LDS_TABLES>(sc, lds.mats, lmat, wlOffset);
                        // shadow ray of Scene::testVisibility SurfaceObject.cpp:425-426
                        float dist = length(surf.p - lp);
                        sdir = (lp - surf.p) / dist;
                        shadowTmax = dist * (1 - kRayEpsilon);
                    }
                    if (leader) pb.shadowDir[slot] = make_float4(sdir.x, sdir.y, sdir.z, shadowTmax);
                    emitShadow = true;
                    // contribution if visible :181-202
                    float dist2;
                    V3 shadowDir;
                    if (pickEnv) { dist2 = 1.0f; shadowDir = normalize(lp); }   // SurfacePoint::getDirectionFrom geometry.cpp:32-37
                    else {
                        V3 dvec = lp - surf.p;
                        dist2 = sqLength(dvec);
                        shadowDir = dvec / sqrtf(dist2);
                    }
                    V3 shadowDir_l = lf.toLocal(-shadowDir);
                    V3 shadowDir_sn = surf.frame.toLocal(shadowDir);
                    S Le = M * S(pickEnv ? (float)(1.0 / kPi) : diffuseEDF(shadowDir_l));
                    float lightPDF = lightProb * areaPDF;
                    float pdfDir = 0.0f;
                    S fs;
                    bool evaluated = false;
                    if constexpr (MULTI) {
                        if (isMulti) {
                            const MultiBSDF<S, decltype(loadComponent)> multi = {multiTree, wlSel, loadComponent};
                            fs = multi.evaluate(type, dirOut_sn, gNorm_sn, shadowDir_sn, wl, &pdfDir);
                            evaluated = true;
                        }
                    }
                    if (!evaluated) fs = bsdfEvaluate<S, MF>(m, type, dirOut_sn, gNorm_sn, shadowDir_sn, wl, &pdfDir);
                    float cosLight = absDot(-shadowDir, lgn);
                    float bsdfPDF = pdfDir * cosLight / dist2;
                    float MISWeight = 1.0f;
                    if (!isinf(areaPDF))
                        MISWeight = (lightPDF * lightPDF) / (lightPDF * lightPDF + bsdfPDF * bsdfPDF);
                    float G = absDot(shadowDir_sn, gNorm_sn) * cosLight / dist2;
                    S contrib = alpha * Le * fs * (G * MISWeight / lightPDF);
                    SpecIO<S>::store(pb.nee, nullptr, slot, rp.numSlots, contrib, 0.0f);
                }
                float uComp = rng.nextFloat();
                float u0 = rng.nextFloat();
                float u1 = rng.nextFloat();
                BsdfSample bs;
                S fs;
                bool sampled = false;
                if constexpr (MULTI) {
                    if (isMulti) {
                        const MultiBSDF<S, decltype(loadComponent)> multi = {multiTree, wlSel, loadComponent};
                        fs = multi.sample(type, dirOut_sn, gNorm_sn, wl, uComp, u0, u1, &bs);
                        sampled = true;
                    }
                }
                if (!sampled) fs = bsdfSample<S, MF>(m, type, dirOut_sn, gNorm_sn, wl, uComp, u0, u1, &bs);
                if (fs.isZero() || bs.dirPDF == 0.0f) {
                    finish = true;                                         // :209
                }
                else {
                    if (bs.dirType & DT_Dispersive) {                      // :211-214
                        bs.dirPDF /= S::N;                                 // WavelengthSamples::NumComponents
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
                if (emitShadow && !emitExt && leader) pb.rayOrg[(size_t)slot * pb.rayStride] = make_float4(surf.p.x, surf.p.y, surf.p.z, kRayEpsilon);
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
        flags |= sp.validBits();
        if (FUSED && emitRegen) {
            radiance = sp.total();                 // accumulated by the caller: nothing of this path needs to reach HBM any more
        }
        else {
            pb.flags[slot] = flags;
            sp.end(pb, slot, rp.numSlots, !emitRegen);
        }
        if (!emitRegen) {
            pb.rng[(size_t)slot * pb.hdrStride] = make_uint4(rng.s0, rng.s1, rng.s2, rng.s3);
            SpecIO<S>::store(pb.alpha, pb.pdfPrev, slot, rp.numSlots, alpha, bsdfPDFprev);
        }
        if (emitExt) {
            pb.rayOrg[(size_t)slot * pb.rayStride] = make_float4(rayOrg.x, rayOrg.y, rayOrg.z, rayTmin);
            pb.rayDir[(size_t)slot * pb.rayStride] = make_float4(rayDir.x, rayDir.y, rayDir.z, INFINITY);
        }
    }
}

// SpectrumStorage::add.  RGB: the sample is Kahan-added to the pixel (RGBTypes.h:176-179).  Spectral: every component goes
// to the storage bin of its wavelength scaled by the reciprocal bin width, then the 16-bin addend is Kahan-added
// (SpectrumTypes.h:818-836).  Bin selection through compare-selects keeps the addend in registers.
__device__ __forceinline__ RGB storageAddend(const RGB& val, float) { return val; }
__device__ __forceinline__ Spec16 storageAddend(const Spec16& val, float wlOffset) {
    const float recBinWidth = 16 / (830.0f - 360.0f);
    uint32_t sBin[16];
    float v[16];
    bool near = true;           // every component lands in its own bin or a neighbour (it always does up to rounding: lambda_i = 360 + 470 (i + u) / 16)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        sBin[i] = min((uint32_t)((wavelengthOf(i, wlOffset) - 360.0f) / (830.0f - 360.0f) * 16), 15u);
        v[i] = val.c[i] * recBinWidth;
        near = near && (sBin[i] + 1u >= (uint32_t)i) && (sBin[i] <= (uint32_t)i + 1u);
    }
    Spec16 addend;
    if (near) {
        // bin b can only receive components b-1, b, b+1: three guarded adds in component order instead of sixteen
#pragma unroll
        for (int b = 0; b < 16; ++b) {
            float a = 0.0f;
            if (b > 0) a = (sBin[b - 1] == (uint32_t)b) ? a + v[b - 1] : a;
            a = (sBin[b] == (uint32_t)b) ? a + v[b] : a;
            if (b < 15) a = (sBin[b + 1] == (uint32_t)b) ? a + v[b + 1] : a;
            addend.c[b] = a;
        }
    }
    else {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
#pragma unroll
            for (int b = 0; b < 16; ++b) addend.c[b] = (sBin[i] == (uint32_t)b) ? addend.c[b] + v[i] : addend.c[b];
        }
    }
    return addend;
}

// Job::kernel's camera half (PathTracingRenderer.cpp:100-120): seeds the sample's stream, draws in source (left-to-right)
// order and leaves the slot with a camera ray in flight; writes the slot's sample header (k_shade's compacted start lanes;
// the tail kernel for the passes left when it takes over).
template <class S>
__device__ __forceinline__ void startSample(const DevScene& sc, const PathBuffers& pb, const RenderParams& rp, uint32_t slot, uint32_t pix,
                                            uint32_t passOfWindow) {
    // Job::kernel PathTracingRenderer.cpp:100-120, draws in source (left-to-right) order
    const uint32_t xy = pb.pixelXY[pix];
    const uint32_t px = xy & 0xFFFFu, py = xy >> 16;
    Rng rng;
    rng.seed(sampleSeed(rp.rngSeed, px, py, rp.sppBegin + passOfWindow));
    float v = rng.nextFloat();
    float time = rp.timeStart * (1 - v) + rp.timeEnd * v;
    (void)time;
    float pxx = px + rng.nextFloat();
    float pyy = py + rng.nextFloat();
    // createWithEqualOffsets: RGBTypes.h:37-45 (offset unused, PDF 1) / SpectrumTypes.h:54-64 (PDF N / 470)
    const float wlOffset = rng.nextFloat();
    float uLambda = rng.nextFloat();
    const uint32_t wl = min((uint32_t)(uint16_t)(S::N * uLambda), (uint32_t)(S::N - 1));
    const float selectWLPDF = S::N == 3 ? 1.0f : S::N / (830.0f - 360.0f);
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
    // weight :126
    float camWeight = absDot(rayDir, lensN) / (sc.camera.areaPDF * dirPDF * selectWLPDF);
    pb.flags[slot] = F_MAKE((uint32_t)ST_FIRST_HIT, 0u, wl, 0u, 0u, 0u);
    pb.rng[(size_t)slot * pb.hdrStride] = make_uint4(rng.s0, rng.s1, rng.s2, rng.s3);
    // alpha = 1, pdfPrev = 0 and sp = 0 are implied by ST_FIRST_HIT (see SpAcc): 48 B (RGB) / 196 B (spectral) not written
    pb.hdr[(size_t)slot * pb.hdrStride] = make_uint4(pix, __float_as_uint(camWeight), __float_as_uint(wlOffset), passOfWindow);
    pb.rayOrg[(size_t)slot * pb.rayStride] = make_float4(lensP.x, lensP.y, lensP.z, 0.0f);
    pb.rayDir[(size_t)slot * pb.rayStride] = make_float4(rayDir.x, rayDir.y, rayDir.z, INFINITY);
}

// The finished sample's contribution, weight * C of sensor->add(p.x, p.y, wls, weight * C) (PathTracingRenderer.cpp:126-130) as
// the sensor's storage takes it (ImageSensor::add -> SpectrumStorage::add: RGB as it is, spectral spread over the 16 bins),
// written to the sample's entry of the result window; k_fold adds the entries of a pixel in pass order.  `inRegisters`: C is
// handed over by the caller (RGB in k_shade); else it is read from the slot's radiance sum in HBM, if the path ever wrote it
// (flag bit 10).
__device__ __forceinline__ void storeResult(float4* results, size_t entry, const RGB& v) { results[entry] = make_float4(v.r, v.g, v.b, 0.0f); }
__device__ __forceinline__ void storeResult(float4* results, size_t entry, const Spec16& v) {
#pragma unroll
    for (int k = 0; k < 4; ++k) results[entry * 4 + k] = make_float4(v.c[4 * k], v.c[4 * k + 1], v.c[4 * k + 2], v.c[4 * k + 3]);
}
template <class S>
__device__ __forceinline__ void writeResult(const PathBuffers& pb, const RenderParams& rp, uint32_t slot, uint32_t flags, const uint4& hdr,
                                            bool inRegisters, S C) {
    float unusedW;
    const float camW = __uint_as_float(hdr.y);
    if (!inRegisters) {
        C = S();
        if (F_SPVALID(flags)) SpecIO<S>::load(pb.spR, nullptr, slot * pb.spStride, rp.numSlots * pb.spStride, C, unusedW);     // else the path gathered nothing: C = 0
    }
    const S weight = (S(1.0f) * S(1.0f)) * camW;
    storeResult(pb.results, (size_t)hdr.w * rp.numPixels + hdr.x, storageAddend(weight * C, S::N == 3 ? 0.0f : __uint_as_float(hdr.z)));
}

// What the finishing lanes of a k_shade workgroup hand to its first lanes: the samples to start.
struct StartLds {
    uint32_t lane[kShadeBlock];                 // compacted: slot = workgroup base + lane
    uint32_t pix[kShadeBlock];
    uint32_t pass[kShadeBlock];
    uint32_t waveBase[kShadeBlock / 64 + 1];
    uint32_t wentIdle;                          // lanes of the workgroup that found the queue exhausted in this launch
};

#ifndef SLR_SHADE_EARLY
#define SLR_SHADE_EARLY 1          // 0 (variant builds): the slot's state records are requested after the table barrier
#endif
template <class S, bool LDS_TABLES, bool MF, bool MULTI = false, bool TEX = false>
__global__ __launch_bounds__(kShadeBlock)
__attribute__((amdgpu_waves_per_eu(S::N == 3 ? SLR_WAVES_RGB : (MF ? SLR_WAVES_SPECTRAL_GLOSSY : SLR_WAVES_SPECTRAL)))) void k_shade(DevScene sc, PathBuffers pb, RenderParams rp, uint32_t parity) {
    __shared__ ShadeLds<S::N != 3> lds;
    __shared__ PushLds pushLds;
    __shared__ StartLds start;
    const uint32_t slot = blockIdx.x * kShadeBlock + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;

    // ---- the end of a render: nothing left anywhere / nothing left in this block ------------------------------------------
    {
        // every slot is out of passes / the tail kernel takes over / this block is finished (all uniform); the three words are
        // requested together: evaluated one after the other they were three dependent scalar-cache round trips
        const uint32_t liveSlots = pb.activeSlots[0], tailMode = pb.tailMode[0], dead = pb.blockDead[blockIdx.x];
        if ((liveSlots == 0u) | (tailMode != 0u) | (dead != 0u)) return;
    }

    // ONE round trip for everything the visit needs that does not depend on the slot's state: flags, visibility word, the pixel's
    // pool counter, the state records (SlotLoads) and the tables staged below; ONE barrier publishes the tables and decides
    // whether the block has any work left.  (SLR_SHADE_EARLY 0, variant builds: the state records are requested after the barrier.)
    uint32_t flags = pb.flags[slot];                                      // numSlots = 256 x workgroups: always in range
    const uint32_t vis = pb.visible[slot];
    SlotLoads<S> in;
    if (SLR_SHADE_EARLY) in.issue(sc, pb, rp, slot);
    const uint32_t state0 = F_STATE(flags);
    const uint32_t globalWave = blockIdx.x * (kShadeBlock / 64) + wave;
    const uint32_t taken = pb.cursor[globalWave];                         // samples this wave has taken from its queue (wave-uniform)
    if (threadIdx.x == 0) start.wentIdle = 0u;
    if (LDS_TABLES) stageShadeTables<S::N != 3>(sc, lds);
    {
        const int anyWork = __syncthreads_or(state0 != ST_IDLE);
        if (!anyWork) {
            // every slot here is idle — its queue is exhausted and a slot never leaves ST_IDLE within a window: the block is
            // finished for the rest of the window; say so, and the scanning kernels stop reading its state
            if (threadIdx.x == 0) pb.blockDead[blockIdx.x] = 1u;
            return;
        }
    }
    if (!SLR_SHADE_EARLY) in.issue(sc, pb, rp, slot);
    const float* lightPMF = LDS_TABLES ? lds.lightPMF : sc.lightPMF;
    const float* lightCDF = LDS_TABLES ? lds.lightCDF : sc.lightCDF;

    bool emitExt = false, emitShadow = false, pathEnded = false;
    S radiance;
    if (state0 == ST_FIRST_HIT || state0 == ST_NEXT_HIT || state0 == ST_FINISH)
        logicSlot<S, LDS_TABLES, MF, MULTI, TEX, true>(sc, pb, rp, lds, lightPMF, lightCDF, slot, in, flags, vis, radiance, emitExt, emitShadow, pathEnded);

    // ---- a path that ended: its contribution goes to the result window, in the same launch ------------------------------------
    if (pathEnded) {
        const uint4 hdr = pb.hdr[(size_t)slot * pb.hdrStride];
        writeResult<S>(pb, rp, slot, flags, hdr, S::N == 3, radiance);
    }
    // ---- stream compaction of the shadow rays (extension rays need no queue: the traversal kernel reads the state flag) ----------
    blockPush(pushLds, emitShadow, slot, pb.shadowQueue, pb.queueCount + parity * kQueueSetWords, rp.shardCapacity, pb.errorWord);
    if (rp.countSlots) {
        const uint64_t ma = __ballot(emitExt || emitShadow || pathEnded);
        if (lane == 0 && ma) atomicAdd((unsigned long long*)&pb.totals[totalIndex(T_SLOT_VISITS, blockIdx.x % kShards)], (unsigned long long)__popcll(ma));
    }

    // ---- the next sample of every slot that needs one (its path ended, or it has not started yet: ST_REGEN after k_reset_slots):
    //      the wanting lanes of a wave, in lane order, take the next items of the wave's queue (pt_kernels.h WorkItem) --------------
    const bool wants = pathEnded || state0 == ST_REGEN;
    {
        const uint64_t mw = __ballot(wants);
        WorkItem w;
        w.valid = false;
        if (wants) w = workItemOf(rp, globalWave, taken + (uint32_t)__popcll(mw & ((1ull << lane) - 1ull)));
        if (lane == 0 && mw) pb.cursor[globalWave] = taken + (uint32_t)__popcll(mw);
        const bool becameIdle = wants && !w.valid;
        if (becameIdle) pb.flags[slot] = F_MAKE((uint32_t)ST_IDLE, 0u, 0u, 0u, 0u, 0u);
        const uint64_t mi = __ballot(becameIdle);
        if (lane == 0 && mi) atomicAdd(&start.wentIdle, (uint32_t)__popcll(mi));
        // compaction of the samples to start over the workgroup: wave counts in LDS, then lanes 0 .. n-1 run startSample, so
        // that the 50-draw seeding of the stream (the seeding contract, ~400 integer operations) and the camera arithmetic run
        // on full waves instead of on the quarter of the lanes whose path has just ended
        const bool starts = wants && w.valid;
        const uint64_t ms = __ballot(starts);
        if (lane == 0) start.waveBase[wave] = (uint32_t)__popcll(ms);
        __syncthreads();
        uint32_t base = 0;
        for (uint32_t k = 0; k < wave; ++k) base += start.waveBase[k];
        if (starts) {
            const uint32_t i = base + (uint32_t)__popcll(ms & ((1ull << lane) - 1ull));
            start.lane[i] = threadIdx.x;
            start.pix[i] = w.pix;
            start.pass[i] = w.pass;
        }
        __syncthreads();
        // the live count: one atomic per workgroup and launch, on one of kShards lines (PathBuffers::idleShards)
        if (threadIdx.x == 0 && start.wentIdle) atomicAdd(&pb.idleShards[(blockIdx.x % kShards) * kCounterStride], start.wentIdle);
        const uint32_t n = start.waveBase[0] + start.waveBase[1] + start.waveBase[2] + start.waveBase[3];
        if (threadIdx.x < n)
            startSample<S>(sc, pb, rp, blockIdx.x * kShadeBlock + start.lane[threadIdx.x], start.pix[threadIdx.x], start.pass[threadIdx.x]);
    }
}

// Start of a render window: every slot wants a sample (ST_REGEN), the queues are at their beginning.  `clearSensor`: the first
// window after slrhip_render_begin also zeroes the sensor (later calls continue the image, like ImageSensor::add does).
template <class S>
__global__ void k_reset_slots(PathBuffers pb, RenderParams rp, uint32_t clearSensor) {
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;     // blockDim == kShadeBlock, numSlots = 256 x gridDim
    pb.flags[slot] = F_MAKE((uint32_t)ST_REGEN, 0u, 0u, 0u, 0u, 0u);
    pb.hdr[(size_t)slot * pb.hdrStride] = make_uint4(0u, 0u, 0u, 0u);
    pb.visible[slot] = 0;
    if (threadIdx.x == 0) pb.blockDead[blockIdx.x] = 0u;
    if ((threadIdx.x & 63u) == 0u) pb.cursor[slot >> 6] = 0u;
    if (clearSensor) {
        constexpr uint32_t planes = S::N == 3 ? 1u : 4u;
        for (size_t e = slot; e < (size_t)rp.numPixels * planes; e += (size_t)gridDim.x * blockDim.x) {
            pb.fbSum[e] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            pb.fbComp[e] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
    }
    if (blockIdx.x == 0) {
        for (uint32_t k = threadIdx.x; k < 2 * kQueueSetWords; k += blockDim.x) pb.queueCount[k] = 0;
        if (threadIdx.x < kShards) pb.idleShards[threadIdx.x * kCounterStride] = 0u;
        if (threadIdx.x == 0) {
            pb.activeSlots[0] = rp.numSlots; pb.errorWord[0] = rp.injectError ? ERR_QUEUE_OVERFLOW : 0u;
            pb.tailMode[0] = 0u; pb.tailWords[0] = 0u; pb.tailWords[1] = 0u; pb.tailIdled[0] = 0u; pb.windowSamples[0] = 0u;
        }
    }
}

// ImageSensor read-out: [H][W][N] linear sums.
template <class S>
__global__ void k_resolve(PathBuffers pb, RenderParams rp, float* dst) {
    const uint32_t pix = blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= rp.numPixels) return;
    const uint32_t xy = pb.pixelXY[pix];
    const uint32_t px = xy & 0xFFFFu, py = xy >> 16;
    float* o = dst + ((size_t)py * rp.imageWidth + px) * S::N;
    if (S::N == 3) {
        const float4 v = pb.fbSum[pix];
        o[0] = v.x; o[1] = v.y; o[2] = v.z;
    }
    else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float4 v = pb.fbSum[(size_t)pix * 4 + k];
            o[4 * k] = v.x; o[4 * k + 1] = v.y; o[4 * k + 2] = v.z; o[4 * k + 3] = v.w;
        }
    }
}

// ---- host-callable launchers -----------------------------------------------------------------------------------
// Diagnostic (slrhip_bsdf_queries): the three BSDF entry points exactly as logicSlot calls them, one query per lane.
// geo[i] = (sampled dir_sn, dirPDF); misc[i] = (dirType, evaluatePDF, 0, 0); fsSample / fsEval in the SpecIO layout.
template <class S>
__global__ void __launch_bounds__(64) k_bsdf_queries(DevScene sc, uint32_t material, uint32_t n, const float* __restrict__ in, float wlOffset,
                                                     uint32_t wl, float4* __restrict__ geo, float4* __restrict__ misc,
                                                     float4* __restrict__ fsSample, float4* __restrict__ fsEval) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* q = in + 12 * (size_t)i;
    Mat<S> m = MatIO<S>::template load<false>(sc, nullptr, material, wlOffset);
    if (m.type & kMatTexturedBit) (void)texturizeMat<S>(sc, m, material, 0.0f, 0.0f, wlOffset);     // textures at texCoord (0, 0)
    const V3 dirOut(q[0], q[1], q[2]), gNorm(q[3], q[4], q[5]), dirIn(q[6], q[7], q[8]);
    BsdfSample bs;
    bs.dir_sn = V3(0, 0, 0);
    S fs, fe;
    float pdf;
    if (m.type == SLRHIP_MATERIAL_MULTI) {
        const auto loadComponent = [&](uint32_t idx) {
            Mat<S> cm = MatIO<S>::template load<false>(sc, nullptr, idx, wlOffset);
            if (cm.type & kMatTexturedBit) (void)texturizeMat<S>(sc, cm, idx, 0.0f, 0.0f, wlOffset);
            return cm;
        };
        const MultiBSDF<S, decltype(loadComponent)> multi = {buildMultiTree<S>(decodeMulti(m), loadComponent), 0u, loadComponent};
        const uint32_t type = multiType(multi.tr, 0u);
        fs = multi.sample(type, dirOut, gNorm, wl, q[9], q[10], q[11], &bs);
        fe = multi.evaluate(type, dirOut, gNorm, dirIn, wl, &pdf);
    }
    else {
        const uint32_t type = bsdfType(m.type, 0u);
        fs = bsdfSample<S, true>(m, type, dirOut, gNorm, wl, q[9], q[10], q[11], &bs);
        fe = bsdfEvaluate<S, true>(m, type, dirOut, gNorm, dirIn, wl, &pdf);
    }
    if (bs.dirPDF == 0.0f) { bs.dir_sn = V3(0, 0, 0); bs.dirType = 0; fs = S(); }
    geo[i] = make_float4(bs.dir_sn.x, bs.dir_sn.y, bs.dir_sn.z, bs.dirPDF);
    misc[i] = make_float4((float)bs.dirType, pdf, 0.0f, 0.0f);
    SpecIO<S>::store(fsSample, nullptr, i, n, fs, 0.0f);
    SpecIO<S>::store(fsEval, nullptr, i, n, fe, 0.0f);
}

// k_shade is instantiated in several translation units (pt_shade_{rgb,spec16,multi_*,tex_*}.hip) so that the build
// parallelises; launchShade (pt_shade.hip) picks one.  `lds`: material / light / spectrum tables staged in LDS;
// `glossy`: GGX / Ward / Ashikhmin lobes compiled in.
void launchShadeRGB(const DevScene& sc, const PathBuffers& pb, const RenderParams& rp, uint32_t parity, bool lds, bool glossy, hipStream_t stream);
void launchShadeSpec16(const DevScene& sc, const PathBuffers& pb, const RenderParams& rp, uint32_t parity, bool lds, bool glossy, hipStream_t stream);
void launchShadeMulti(const DevScene& sc, const PathBuffers& pb, const RenderParams& rp, uint32_t parity, hipStream_t stream);

} // namespace slrhip
