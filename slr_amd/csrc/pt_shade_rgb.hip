// pt_shade_rgb.hip — k_shade instantiations (see pt_shade_kernels.h)
#include "pt_shade_kernels.h"

namespace slrhip {

void launchShadeRGB(const DevScene& sc, const PathBuffers& pb, const RenderParams& rp, uint32_t parity, bool lds, bool glossy, hipStream_t stream) {
    const dim3 grid(rp.numSlots / kShadeBlock), block(kShadeBlock);
    if (lds && !glossy) hipLaunchKernelGGL((k_shade<RGB, true, false>), grid, block, 0, stream, sc, pb, rp, parity);
    else if (lds) hipLaunchKernelGGL((k_shade<RGB, true, true>), grid, block, 0, stream, sc, pb, rp, parity);
    else if (!glossy) hipLaunchKernelGGL((k_shade<RGB, false, false>), grid, block, 0, stream, sc, pb, rp, parity);
    else hipLaunchKernelGGL((k_shade<RGB, false, true>), grid, block, 0, stream, sc, pb, rp, parity);
}

} // namespace slrhip
