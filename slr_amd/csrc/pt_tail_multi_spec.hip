// pt_tail_multi_spec.hip — one k_tail instantiation (see pt_tail_kernels.h); one per file: each takes minutes to compile
#include "pt_tail_kernels.h"

namespace slrhip {

void launchTailMultiSpec(const DevScene& sc, const PathBuffers& pb, const RenderParams& rp, uint32_t blocks, hipStream_t stream) {
    hipLaunchKernelGGL((k_tail<Spec16, false, true, true, false>), dim3(blocks), dim3(kShadeBlock), 0, stream, sc, pb, rp);
}

} // namespace slrhip
