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
// Samples accumulated into pixels by the render call that just ended = sum over the slots of the per-slot count k_shade keeps
// in the sample header (hdr.x, restarted by k_reset_slots).  This is the device's own account of the work done: the host's
// numPixels x spp would be a tautology.  One atomic per workgroup on the sharded totals.
__global__ __launch_bounds__(kShadeBlock) void k_count_samples(PathBuffers pb, RenderParams rp) {
    __shared__ uint32_t red[kShadeBlock / 64];
    uint32_t n = 0;
    for (uint32_t slot = blockIdx.x * kShadeBlock + threadIdx.x; slot < rp.numSlots; slot += gridDim.x * kShadeBlock) n += pb.hdr[(size_t)slot * pb.hdrStride].x;
    for (int off = 32; off > 0; off >>= 1) n += __shfl_down(n, off);
    if ((threadIdx.x & 63u) == 0) red[threadIdx.x >> 6] = n;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
        for (int w = 0; w < kShadeBlock / 64; ++w) t += red[w];
        if (t) atomicAdd((unsigned long long*)&pb.totals[totalIndex(T_SAMPLES, blockIdx.x % kShards)], (unsigned long long)t);
    }
}

void launchCountSamples(const PathBuffers& pb, const RenderParams& rp, hipStream_t stream) {
    if (rp.numSlots == 0) return;
    const uint32_t blocks = std::min<uint32_t>((rp.numSlots + kShadeBlock - 1) / kShadeBlock, 2048u);
    hipLaunchKernelGGL(k_count_samples, dim3(blocks), dim3(kShadeBlock), 0, stream, pb, rp);
}
void launchResolve(const PathBuffers& pb, const RenderParams& rp, float* dst, hipStream_t stream) {
    const dim3 grid((rp.numPixels + 255) / 256), block(256);
    if (rp.spectral) hipLaunchKernelGGL(k_resolve<Spec16>, grid, block, 0, stream, pb, rp, dst);
    else hipLaunchKernelGGL(k_resolve<RGB>, grid, block, 0, stream, pb, rp, dst);
}

} // namespace slrhip
