// pt_tail_multi.hip — k_tail instantiations (see pt_tail_kernels.h)
#include "pt_tail_kernels.h"

namespace slrhip {

void launchTailMulti(const DevScene& sc, const PathBuffers& pb, const RenderParams& rp, uint32_t blocks, hipStream_t stream) {
    const dim3 grid(blocks), block(kShadeBlock);
    if (sc.numTextures) {
        if (rp.spectral) hipLaunchKernelGGL((k_tail<Spec16, false, true, true, true>), grid, block, 0, stream, sc, pb, rp);
        else hipLaunchKernelGGL((k_tail<RGB, false, true, true, true>), grid, block, 0, stream, sc, pb, rp);
        return;
    }
    if (rp.spectral) hipLaunchKernelGGL((k_tail<Spec16, false, true, true>), grid, block, 0, stream, sc, pb, rp);
    else hipLaunchKernelGGL((k_tail<RGB, false, true, true>), grid, block, 0, stream, sc, pb, rp);
}

} // namespace slrhip
