// pt_tail_rgb.hip — k_tail instantiations (see pt_tail_kernels.h)
#include "pt_tail_kernels.h"

namespace slrhip {

void launchTailRGB(const DevScene& sc, const PathBuffers& pb, const RenderParams& rp, bool lds, bool glossy, uint32_t blocks, hipStream_t stream) {
    const dim3 grid(blocks), block(kShadeBlock);
    if (lds && !glossy) hipLaunchKernelGGL((k_tail<RGB, true, false>), grid, block, 0, stream, sc, pb, rp);
    else if (lds) hipLaunchKernelGGL((k_tail<RGB, true, true>), grid, block, 0, stream, sc, pb, rp);
    else if (!glossy) hipLaunchKernelGGL((k_tail<RGB, false, false>), grid, block, 0, stream, sc, pb, rp);
    else hipLaunchKernelGGL((k_tail<RGB, false, true>), grid, block, 0, stream, sc, pb, rp);
}

} // namespace slrhip
