// pt_shade.hip — launchers of the shading half of the wavefront path tracer (gfx950, wave64) and the instantiations of its
// small kernels; the kernel templates are in pt_shade_kernels.h, the k_shade instantiations in pt_shade_{rgb,spec16,multi*,tex*}.hip.
#include <algorithm>

#include "pt_shade_kernels.h"

namespace slrhip {

void launchResetSlots(const PathBuffers& pb, const RenderParams& rp, bool clearAcc, hipStream_t stream) {
    const dim3 grid(rp.numSlots / kShadeBlock), block(kShadeBlock);
    if (rp.spectral) hipLaunchKernelGGL(k_reset_slots<Spec16>, grid, block, 0, stream, pb, rp, clearAcc ? 1u : 0u);
    else hipLaunchKernelGGL(k_reset_slots<RGB>, grid, block, 0, stream, pb, rp, clearAcc ? 1u : 0u);
}
void launchShade(const DevScene& sc, const PathBuffers& pb, const RenderParams& rp, uint32_t parity, hipStream_t stream) {
    const bool ldsTables = sc.shadeTables != nullptr;      // slrhip_upload_scene packs them when the scene fits the LDS limits (shadeTablesFit)
    // The microfacet (GGX) code costs ~45 VGPRs, so scenes without such lobes get kernels without it.
    const bool glossy = sc.hasMicrofacet != 0;
    if (sc.hasMulti || sc.numTextures) {
        // MultiBSDF scenes: one kernel per mode, tables in HBM (a component is re-read per use), all lobes compiled in
        launchShadeMulti(sc, pb, rp, parity, stream);
        return;
    }
    if (rp.spectral) launchShadeSpec16(sc, pb, rp, parity, ldsTables, glossy, stream);
    else launchShadeRGB(sc, pb, rp, parity, ldsTables, glossy, stream);
}
void launchBsdfQueries(const DevScene& sc, bool spectral, uint32_t material, uint32_t n, const float* in, float wlOffset, uint32_t wl,
                       float4* geo, float4* misc, float4* fsSample, float4* fsEval, hipStream_t stream) {
    const dim3 grid((n + 63) / 64), block(64);
    if (spectral) hipLaunchKernelGGL(k_bsdf_queries<Spec16>, grid, block, 0, stream, sc, material, n, in, wlOffset, wl, geo, misc, fsSample, fsEval);
    else hipLaunchKernelGGL(k_bsdf_queries<RGB>, grid, block, 0, stream, sc, material, n, in, wlOffset, wl, geo, misc, fsSample, fsEval);
}
// Samples rendered in the window that just ended = what the waves took from their queues (PathBuffers::cursor, capped at
// the queue's length: a cursor runs past the end by the lanes that found nothing).  This is the device's own account of the
// work handed out — the host's numPixels x passes would be a tautology.  One atomic per workgroup on the sharded totals.
__global__ __launch_bounds__(kShadeBlock) void k_count_samples(PathBuffers pb, RenderParams rp) {
    __shared__ uint32_t red[kShadeBlock / 64];
    uint32_t n = 0;
    for (uint32_t w = blockIdx.x * kShadeBlock + threadIdx.x; w < rp.numWaves; w += gridDim.x * kShadeBlock) n += workSamplesTaken(rp, w, pb.cursor[w]);
    for (int off = 32; off > 0; off >>= 1) n += __shfl_down(n, off);
    if ((threadIdx.x & 63u) == 0) red[threadIdx.x >> 6] = n;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
        for (int w = 0; w < kShadeBlock / 64; ++w) t += red[w];
        if (t) {
            atomicAdd((unsigned long long*)&pb.totals[totalIndex(T_SAMPLES, blockIdx.x % kShards)], (unsigned long long)t);
            atomicAdd(pb.windowSamples, t);
        }
    }
}

// ImageSensor::add for the passes of a finished window (Core/ImageSensor.cpp:52-62 -> SpectrumStorage::add): per pixel and bin,
// sum += value with the compensated sum of BasicTypes/CompensatedSum.h:24-30, the entries taken in PASS ORDER — the order in which
// one thread of the reference adds the samples of a pixel, whatever slot rendered them here.  One thread per float4 of the
// sensor (RGB: one per pixel, the fourth component idles; spectral: four per pixel); the window is pass-major, so the threads of
// a wave read consecutive 16-byte entries at every pass: a streaming kernel (window bytes / HBM rate).
__global__ __launch_bounds__(256) void k_fold(PathBuffers pb, uint32_t elems, uint32_t passes) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= elems) return;
    float4 s = pb.fbSum[e], c = pb.fbComp[e];
    const float4* r = pb.results + e;
    uint32_t p = 0;
    for (; p + 4 <= passes; p += 4) {
        // four entries requested together; added one after the other
        const float4 v0 = r[(size_t)p * elems], v1 = r[(size_t)(p + 1) * elems], v2 = r[(size_t)(p + 2) * elems], v3 = r[(size_t)(p + 3) * elems];
        kahanAdd(s.x, c.x, v0.x); kahanAdd(s.y, c.y, v0.y); kahanAdd(s.z, c.z, v0.z); kahanAdd(s.w, c.w, v0.w);
        kahanAdd(s.x, c.x, v1.x); kahanAdd(s.y, c.y, v1.y); kahanAdd(s.z, c.z, v1.z); kahanAdd(s.w, c.w, v1.w);
        kahanAdd(s.x, c.x, v2.x); kahanAdd(s.y, c.y, v2.y); kahanAdd(s.z, c.z, v2.z); kahanAdd(s.w, c.w, v2.w);
        kahanAdd(s.x, c.x, v3.x); kahanAdd(s.y, c.y, v3.y); kahanAdd(s.z, c.z, v3.z); kahanAdd(s.w, c.w, v3.w);
    }
    for (; p < passes; ++p) {
        const float4 v = r[(size_t)p * elems];
        kahanAdd(s.x, c.x, v.x); kahanAdd(s.y, c.y, v.y); kahanAdd(s.z, c.z, v.z); kahanAdd(s.w, c.w, v.w);
    }
    pb.fbSum[e] = s;
    pb.fbComp[e] = c;
}

void launchFold(const PathBuffers& pb, const RenderParams& rp, hipStream_t stream) {
    const uint32_t elems = rp.numPixels * (rp.spectral ? 4u : 1u);
    if (elems == 0 || rp.sppCount == 0) return;
    hipLaunchKernelGGL(k_fold, dim3((elems + 255) / 256), dim3(256), 0, stream, pb, elems, rp.sppCount);
}
void launchCountSamples(const PathBuffers& pb, const RenderParams& rp, hipStream_t stream) {
    if (rp.numSlots == 0) return;
    const uint32_t blocks = std::min<uint32_t>((rp.numWaves + kShadeBlock - 1) / kShadeBlock, 2048u);
    hipLaunchKernelGGL(k_count_samples, dim3(blocks), dim3(kShadeBlock), 0, stream, pb, rp);
}
void launchResolve(const PathBuffers& pb, const RenderParams& rp, float* dst, hipStream_t stream) {
    const dim3 grid((rp.numPixels + 255) / 256), block(256);
    if (rp.spectral) hipLaunchKernelGGL(k_resolve<Spec16>, grid, block, 0, stream, pb, rp, dst);
    else hipLaunchKernelGGL(k_resolve<RGB>, grid, block, 0, stream, pb, rp, dst);
}

} // namespace slrhip
