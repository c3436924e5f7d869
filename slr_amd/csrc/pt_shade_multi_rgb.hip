// pt_shade_multi_rgb.hip — one k_logic instantiation (see pt_shade_kernels.h); one per file: each takes a minute or more to compile
#include "pt_shade_kernels.h"

namespace slrhip {

void launchLogicMultiRGB(const DevScene& sc, const PathBuffers& pb, const RenderParams& rp, uint32_t parity, hipStream_t stream) {
    const dim3 grid((rp.numSlots + kShadeBlock - 1) / kShadeBlock), block(kShadeBlock);
    hipLaunchKernelGGL((k_logic<RGB, false, true, true>), grid, block, 0, stream, sc, pb, rp, parity);
}

} // namespace slrhip
