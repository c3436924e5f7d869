/*
 * slrhip_debug.h — diagnostic exports of libslrhip.so that are NOT part of the drop-in boundary (include/slrhip.h).
 * They exist for the parity tests: function-level checks of the device code against the reference's own answers.
 */
#ifndef SLRHIP_DEBUG_H
#define SLRHIP_DEBUG_H

#include "slrhip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Function-level BSDF queries against material `material` of the uploaded scene,
 * through the same device functions the shading kernel calls: BSDF::sample / evaluate /
 * evaluatePDF (libSLR/Core/directional_distribution_functions.h:231-279; flags = All, non-adjoint) on the
 * BSDF that SurfaceMaterial::getBSDF (libSLR/Core/surface_material.h:22) builds for wavelengths
 * WavelengthSamples::createWithEqualOffsets(wl_offset, u_lambda) (libSLR/BasicTypes/SpectrumTypes.h:54-64).
 *   queries[12 i ..]     = dirOut_sn[3], gNormal_sn[3], dirIn_sn[3], uComponent, uDir[2]
 *   out[(6 + 2C) i ..]   = sampled dir_sn[3], dirPDF, dirType, fs(sample)[C], fs(evaluate)[C],
 *                          evaluatePDF        (C = slrhip_components; zeros when dirPDF == 0)
 * Directions are in the shading frame (z = shading normal).  Host arrays; synchronises.        */
int slrhip_bsdf_queries(slrhip_ctx* ctx, uint32_t material, uint32_t n, const float* queries,
                        float wl_offset, float u_lambda, float* out);

/* The work distribution of a render window (slr_amd/csrc/pt_kernels.h, WorkItem), evaluated on the HOST with the very function
 * the kernels call: the samples (pixel, pass) of a window of `num_passes` passes over `num_pixels` pixels are dealt to the
 * `num_slots / 64` wave queues in runs of `run_length` passes of a pixel.  counts[pass * num_pixels + pixel] receives how many
 * times the sample comes up over all queues (the caller checks: exactly once); queue_lengths[wave] the samples of each queue.
 * No GPU is touched.  Returns SLRHIP_OK, or SLRHIP_ERR_INVALID_ARGUMENT when run_length does not divide num_passes.            */
int slrhip_debug_work_distribution(uint32_t num_pixels, uint32_t num_slots, uint32_t num_passes, uint32_t run_length,
                                   uint32_t* counts, uint32_t* queue_lengths);

#ifdef __cplusplus
}
#endif
#endif /* SLRHIP_DEBUG_H */
