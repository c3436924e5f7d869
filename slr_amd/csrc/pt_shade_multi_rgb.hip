// pt_shade_multi_rgb.hip — one k_shade instantiation (see pt_shade_kernels.h); one per file: each takes a minute or more to compile
#include "pt_shade_kernels.h"

namespace slrhip {

void launchShadeMultiRGB(const DevScene& sc, const PathBuffers& pb, const RenderParams& rp, uint32_t parity, hipStream_t stream) {
    const dim3 grid(rp.numSlots / kShadeBlock), block(kShadeBlock);
    hipLaunchKernelGGL((k_shade<RGB, false, true, true>), grid, block, 0, stream, sc, pb, rp, parity);
}

} // namespace slrhip
