// pt_tail.hip — launcher of the tail kernel (pt_tail_kernels.h): picks the instantiation that matches the k_logic variant of the
// scene (launchLogic, pt_shade.hip); the instantiations are in pt_tail_{rgb,spec16}.hip and pt_tail_{multi,tex}_rgb.hip.
#include <algorithm>

#include "pt_tail_kernels.h"

namespace slrhip {

// Live slots -> pb.regenQueue used as one flat list (the queues are dead in tail mode; capacity kShards x shardCapacity >= numSlots).
// The stripe-0 slots also retire their pixel's sample-pool mask of queue set `parity`: the k_regen of the iteration in which
// the tail took over has handed those passes out, the k_logic that would have advanced the counter no longer runs.
__global__ __launch_bounds__(kShadeBlock) void k_tail_collect(PathBuffers pb, RenderParams rp, uint32_t parity) {
    __shared__ uint32_t waveCount[kShadeBlock / 64];
    __shared__ uint32_t base;
    if (pb.blockDead[blockIdx.x]) return;             // never set for a block that holds stripe-0 slots
    const uint32_t slot = blockIdx.x * kShadeBlock + threadIdx.x;
    if (slot < rp.numPixels) {
        unsigned long long* done = pb.finishedMask + (size_t)parity * rp.numPixels + slot;
        const unsigned long long m = *done;
        if (m) {
            pb.nextSample[slot] += (uint32_t)__popcll(m);
            *done = 0ull;
        }
    }
    const bool live = slot < rp.numSlots && F_STATE(pb.flags[slot]) != ST_IDLE;
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
        pb.regenQueue[off + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = slot;
    }
}

void launchTailRGB(const DevScene& sc, const PathBuffers& pb, const RenderParams& rp, bool lds, bool glossy, uint32_t blocks, hipStream_t stream);
void launchTailSpec16(const DevScene& sc, const PathBuffers& pb, const RenderParams& rp, bool lds, bool glossy, uint32_t blocks, hipStream_t stream);
void launchTailMultiRGB(const DevScene& sc, const PathBuffers& pb, const RenderParams& rp, uint32_t blocks, hipStream_t stream);
void launchTailTexRGB(const DevScene& sc, const PathBuffers& pb, const RenderParams& rp, uint32_t blocks, hipStream_t stream);

bool tailKernelAvailable(const DevScene& sc, bool spectral) { return !(spectral && (sc.hasMulti || sc.numTextures)); }

void launchTail(const DevScene& sc, const PathBuffers& pb, const RenderParams& rp, uint32_t liveSlots, uint32_t parity, int numCUs, hipStream_t stream) {
    if (rp.numSlots == 0 || liveSlots == 0) return;
    hipLaunchKernelGGL(k_tail_collect, dim3((rp.numSlots + kShadeBlock - 1) / kShadeBlock), dim3(kShadeBlock), 0, stream, pb, rp, parity);
    // one lane per listed slot; lanes take further slots from the list when theirs goes idle, so a grid smaller than the list is fine
    const uint32_t blocks = std::min<uint32_t>((liveSlots + kShadeBlock - 1) / kShadeBlock, (uint32_t)numCUs * 4u);
    const bool ldsTables = sc.numMaterials <= (uint32_t)kLdsMaterials && sc.numLights <= (uint32_t)kLdsLights &&
                           (!rp.spectral || (sc.numSpectra <= (uint32_t)kLdsSpectra && sc.numSpectrumData <= (uint32_t)kLdsPoolFloats));
    const bool glossy = sc.hasMicrofacet != 0;
    // (spectral scenes with MultiBSDF materials or textures never enter tail mode — tailKernelAvailable: that instantiation alone
    // takes nine minutes to compile for a 1 % gain on scenes no BASELINE config has)
    if (sc.numTextures) launchTailTexRGB(sc, pb, rp, blocks, stream);
    else if (sc.hasMulti) launchTailMultiRGB(sc, pb, rp, blocks, stream);
    else if (rp.spectral) launchTailSpec16(sc, pb, rp, ldsTables, glossy, blocks, stream);
    else launchTailRGB(sc, pb, rp, ldsTables, glossy, blocks, stream);
}

} // namespace slrhip
