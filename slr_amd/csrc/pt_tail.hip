// pt_tail.hip — launcher of the tail kernel (pt_tail_kernels.h): picks the instantiation that matches the k_shade variant of the
// scene (launchShade, pt_shade.hip); the instantiations are in pt_tail_{rgb,spec16}.hip and pt_tail_{multi,tex}_{rgb,spec}.hip.
#include <algorithm>

#include "pt_tail_kernels.h"

namespace slrhip {

// Live slots -> pb.tailList (numSlots entries).
__global__ __launch_bounds__(kShadeBlock) void k_tail_collect(PathBuffers pb, RenderParams rp) {
    __shared__ uint32_t waveCount[kShadeBlock / 64];
    __shared__ uint32_t base;
    if (pb.blockDead[blockIdx.x]) return;
    const uint32_t slot = blockIdx.x * kShadeBlock + threadIdx.x;
    const bool live = F_STATE(pb.flags[slot]) != ST_IDLE;
    const uint64_t m = __ballot(live);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (lane == 0) waveCount[wave] = (uint32_t)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t total = waveCount[0] + waveCount[1] + waveCount[2] + waveCount[3];
        base = total ? atomicAdd(&pb.tailWords[0], total) : 0u;
    }
    __syncthreads();
    if (live) {
        uint32_t off = base;
        for (uint32_t w = 0; w < wave; ++w) off += waveCount[w];
        pb.tailList[off + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = slot;
    }
}

void launchTailRGB(const DevScene& sc, const PathBuffers& pb, const RenderParams& rp, bool lds, bool glossy, uint32_t blocks, hipStream_t stream);
void launchTailSpec16(const DevScene& sc, const PathBuffers& pb, const RenderParams& rp, bool lds, bool glossy, uint32_t blocks, hipStream_t stream);
void launchTailMultiRGB(const DevScene& sc, const PathBuffers& pb, const RenderParams& rp, uint32_t blocks, hipStream_t stream);
void launchTailTexRGB(const DevScene& sc, const PathBuffers& pb, const RenderParams& rp, uint32_t blocks, hipStream_t stream);
void launchTailMultiSpec(const DevScene& sc, const PathBuffers& pb, const RenderParams& rp, uint32_t blocks, hipStream_t stream);
void launchTailTexSpec(const DevScene& sc, const PathBuffers& pb, const RenderParams& rp, uint32_t blocks, hipStream_t stream);

bool tailKernelAvailable(const DevScene&, bool) { return true; }      // every shade-kernel variant has its tail kernel

void launchTail(const DevScene& sc, const PathBuffers& pb, const RenderParams& rp, uint32_t liveSlots, int numCUs, hipStream_t stream) {
    if (rp.numSlots == 0 || liveSlots == 0) return;
    hipLaunchKernelGGL(k_tail_collect, dim3(rp.numSlots / kShadeBlock), dim3(kShadeBlock), 0, stream, pb, rp);
    // one lane per listed slot; lanes take further slots from the list when theirs goes idle, so a grid smaller than the list is fine
    const uint32_t blocks = std::min<uint32_t>((liveSlots + kShadeBlock - 1) / kShadeBlock, (uint32_t)numCUs * 4u);
    const bool ldsTables = sc.shadeTables != nullptr;      // slrhip_upload_scene packs them when the scene fits the LDS limits (shadeTablesFit)
    const bool glossy = sc.hasMicrofacet != 0;
    if (sc.numTextures) { if (rp.spectral) launchTailTexSpec(sc, pb, rp, blocks, stream); else launchTailTexRGB(sc, pb, rp, blocks, stream); }
    else if (sc.hasMulti) { if (rp.spectral) launchTailMultiSpec(sc, pb, rp, blocks, stream); else launchTailMultiRGB(sc, pb, rp, blocks, stream); }
    else if (rp.spectral) launchTailSpec16(sc, pb, rp, ldsTables, glossy, blocks, stream);
    else launchTailRGB(sc, pb, rp, ldsTables, glossy, blocks, stream);
}

} // namespace slrhip
