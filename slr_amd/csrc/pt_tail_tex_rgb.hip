// pt_tail_tex_rgb.hip — one k_tail instantiation (see pt_tail_kernels.h); one per file: each takes minutes to compile
#include "pt_tail_kernels.h"

namespace slrhip {

void launchTailTexRGB(const DevScene& sc, const PathBuffers& pb, const RenderParams& rp, uint32_t blocks, hipStream_t stream) {
    hipLaunchKernelGGL((k_tail<RGB, false, true, true, true>), dim3(blocks), dim3(kShadeBlock), 0, stream, sc, pb, rp);
}

} // namespace slrhip
