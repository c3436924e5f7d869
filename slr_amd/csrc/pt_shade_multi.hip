// pt_shade_multi.hip — dispatch to the all-lobes k_shade instantiations (MultiBSDF scenes, textured scenes); the instantiations
// themselves are one per file (pt_shade_{multi,tex}_{rgb,spec}.hip): each takes a minute or more to compile
#include "pt_shade_kernels.h"

namespace slrhip {

void launchShadeMultiRGB(const DevScene& sc, const PathBuffers& pb, const RenderParams& rp, uint32_t parity, hipStream_t stream);
void launchShadeMultiSpec(const DevScene& sc, const PathBuffers& pb, const RenderParams& rp, uint32_t parity, hipStream_t stream);
void launchShadeTexRGB(const DevScene& sc, const PathBuffers& pb, const RenderParams& rp, uint32_t parity, hipStream_t stream);
void launchShadeTexSpec(const DevScene& sc, const PathBuffers& pb, const RenderParams& rp, uint32_t parity, hipStream_t stream);

void launchShadeMulti(const DevScene& sc, const PathBuffers& pb, const RenderParams& rp, uint32_t parity, hipStream_t stream) {
    if (sc.numTextures) {
        // textured scenes (checkerboard reflectances, bump): the same all-lobes kernel with the texture code
        if (rp.spectral) launchShadeTexSpec(sc, pb, rp, parity, stream);
        else launchShadeTexRGB(sc, pb, rp, parity, stream);
        return;
    }
    if (rp.spectral) launchShadeMultiSpec(sc, pb, rp, parity, stream);
    else launchShadeMultiRGB(sc, pb, rp, parity, stream);
}

} // namespace slrhip
