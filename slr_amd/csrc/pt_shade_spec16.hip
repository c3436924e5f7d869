// pt_shade_spec16.hip — k_logic instantiations (see pt_shade_kernels.h)
#include "pt_shade_kernels.h"

namespace slrhip {

void launchLogicSpec16(const DevScene& sc, const PathBuffers& pb, const RenderParams& rp, uint32_t parity, bool lds, bool glossy, hipStream_t stream) {
    const dim3 grid((rp.numSlots + kShadeBlock - 1) / kShadeBlock), block(kShadeBlock);
    if (lds && !glossy) hipLaunchKernelGGL((k_logic<Spec16, true, false>), grid, block, 0, stream, sc, pb, rp, parity);
    else if (lds) hipLaunchKernelGGL((k_logic<Spec16, true, true>), grid, block, 0, stream, sc, pb, rp, parity);
    else if (!glossy) hipLaunchKernelGGL((k_logic<Spec16, false, false>), grid, block, 0, stream, sc, pb, rp, parity);
    else hipLaunchKernelGGL((k_logic<Spec16, false, true>), grid, block, 0, stream, sc, pb, rp, parity);
}

} // namespace slrhip
