// pt_shade_multi.hip — k_logic instantiations (see pt_shade_kernels.h)
#include "pt_shade_kernels.h"

namespace slrhip {

void launchLogicMulti(const DevScene& sc, const PathBuffers& pb, const RenderParams& rp, uint32_t parity, hipStream_t stream) {
    const dim3 grid((rp.numSlots + kShadeBlock - 1) / kShadeBlock), block(kShadeBlock);
    if (sc.numTextures) {
        // textured scenes (checkerboard reflectances, bump): the same all-lobes kernel with the texture code
        if (rp.spectral) hipLaunchKernelGGL((k_logic<Spec16, false, true, true, true>), grid, block, 0, stream, sc, pb, rp, parity);
        else hipLaunchKernelGGL((k_logic<RGB, false, true, true, true>), grid, block, 0, stream, sc, pb, rp, parity);
        return;
    }
    if (rp.spectral) hipLaunchKernelGGL((k_logic<Spec16, false, true, true>), grid, block, 0, stream, sc, pb, rp, parity);
    else hipLaunchKernelGGL((k_logic<RGB, false, true, true>), grid, block, 0, stream, sc, pb, rp, parity);
}

} // namespace slrhip
