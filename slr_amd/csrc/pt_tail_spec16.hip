// pt_tail_spec16.hip — k_tail instantiations (see pt_tail_kernels.h)
#include "pt_tail_kernels.h"

namespace slrhip {

void launchTailSpec16(const DevScene& sc, const PathBuffers& pb, const RenderParams& rp, bool lds, bool glossy, uint32_t blocks, hipStream_t stream) {
    const dim3 grid(blocks), block(kShadeBlock);
    if (lds && !glossy) hipLaunchKernelGGL((k_tail<Spec16, true, false>), grid, block, 0, stream, sc, pb, rp);
    else if (lds) hipLaunchKernelGGL((k_tail<Spec16, true, true>), grid, block, 0, stream, sc, pb, rp);
    else if (!glossy) hipLaunchKernelGGL((k_tail<Spec16, false, false>), grid, block, 0, stream, sc, pb, rp);
    else hipLaunchKernelGGL((k_tail<Spec16, false, true>), grid, block, 0, stream, sc, pb, rp);
}

} // namespace slrhip
