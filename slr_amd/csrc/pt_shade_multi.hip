// pt_shade_multi.hip — dispatch to the all-lobes k_logic instantiations (MultiBSDF scenes, textured scenes); the instantiations
// themselves are one per file (pt_shade_{multi,tex}_{rgb,spec}.hip): each takes a minute or more to compile
#include "pt_shade_kernels.h"

namespace slrhip {

void launchLogicMultiRGB(const DevScene& sc, const PathBuffers& pb, const RenderParams& rp, uint32_t parity, hipStream_t stream);
void launchLogicMultiSpec(const DevScene& sc, const PathBuffers& pb, const RenderParams& rp, uint32_t parity, hipStream_t stream);
void launchLogicTexRGB(const DevScene& sc, const PathBuffers& pb, const RenderParams& rp, uint32_t parity, hipStream_t stream);
void launchLogicTexSpec(const DevScene& sc, const PathBuffers& pb, const RenderParams& rp, uint32_t parity, hipStream_t stream);

void launchLogicMulti(const DevScene& sc, const PathBuffers& pb, const RenderParams& rp, uint32_t parity, hipStream_t stream) {
    if (sc.numTextures) {
        // textured scenes (checkerboard reflectances, bump): the same all-lobes kernel with the texture code
        if (rp.spectral) launchLogicTexSpec(sc, pb, rp, parity, stream);
        else launchLogicTexRGB(sc, pb, rp, parity, stream);
        return;
    }
    if (rp.spectral) launchLogicMultiSpec(sc, pb, rp, parity, stream);
    else launchLogicMultiRGB(sc, pb, rp, parity, stream);
}

} // namespace slrhip
