/*
 * slr_oracle.h — C interface of the CPU ORACLE for the path-tracing hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (slr_amd/, include/) may include,
 * link or call this; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg do.  The same interface is implemented twice:
 *   - oracle/slr_oracle.cpp      a scalar restatement of the reference algorithm
 *                                (prefix slr_oracle_), travels to the GPU box;
 *   - oracle/ref_build/ref_shim.cpp  the compiled reference libSLR itself driven through
 *                                its C++ API (prefix slr_ref_), built into oracle/_ref/
 *                                only where /root/reference exists.
 * Both consume the flat scene of include/slrhip.h and the same per-(pixel,sample)
 * seeding contract (slrhip_sample_seed), so their outputs are comparable bit for bit.
 */
#ifndef SLR_ORACLE_H
#define SLR_ORACLE_H

#include "../include/slrhip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct slr_oracle_scene slr_oracle_scene;

typedef struct slr_oracle_counters {
    uint64_t samples;
    uint64_t extension_rays;   /* Scene::intersect calls      */
    uint64_t shadow_rays;      /* Scene::testVisibility calls */
    uint64_t loop_iterations;  /* bounce-loop bodies entered  */
    uint64_t rng_draws;
    uint64_t nodes_visited;    /* oracle's binary BVH (not comparable with the 4-wide tree) */
    uint64_t tris_tested;
} slr_oracle_counters;

/* One ray of a traversal-only parity batch. */
typedef struct slr_oracle_ray {
    float org[3];
    float dir[3];
    float dist_min;
    float dist_max;
} slr_oracle_ray;
typedef struct slr_oracle_hit {
    uint32_t triangle;   /* 0xFFFFFFFF = miss */
    float dist;
    float b0, b1;        /* Intersection::u, ::v (TriangleMesh.cpp:172-173) */
} slr_oracle_hit;

#define SLR_ORACLE_DECLARE(P)                                                                      \
    slr_oracle_scene* P##create(const slrhip_scene_desc* scene, int mode);                         \
    void P##destroy(slr_oracle_scene* s);                                                          \
    /* Passes [spp_begin, spp_begin+spp_count) of the shard's pixels, per-(pixel,sample) seeding.  \
     * fb_sum / fb_comp: [H][W][C] Kahan sum and compensation (in/out; fresh render = zeros).      \
     * threads <= 0: all hardware threads. Returns 0 on success.                               */ \
    int P##render(slr_oracle_scene* s, const slrhip_render_settings* settings, slrhip_shard shard, \
                  uint32_t spp_begin, uint32_t spp_count, int threads, float* fb_sum,              \
                  float* fb_comp, slr_oracle_counters* counters);                                  \
    /* One sample: out[0..C) = weight*C, out[C], out[C+1] = pixel position p.x, p.y.            */ \
    int P##sample(slr_oracle_scene* s, const slrhip_render_settings* settings, uint32_t px,        \
                  uint32_t py, uint32_t pass, float* out);                                         \
    /* Closest-hit queries (Scene::intersect on the aggregate only).                           */ \
    int P##trace(slr_oracle_scene* s, const slr_oracle_ray* rays, uint32_t n, slr_oracle_hit* hits);\
    /* xorshift128 known answers: n raw uint32 draws then n float draws from `seed`.            */ \
    void P##rng(int32_t seed, uint32_t n, uint32_t* uints, float* floats);                         \
    int P##components(const slr_oracle_scene* s);                                                  \
    /* Function-level known answers: BSDF::sample / evaluate / evaluatePDF             \
     * (DDF.h:231-279, flags = All, non-adjoint) of scene material `material` for n queries under  \
     * WavelengthSamples::createWithEqualOffsets(wl_offset, u_lambda).                             \
     * in[12 i ..]  = dirOut_sn[3], gNormal_sn[3], dirIn_sn[3], uComponent, uDir[2]                \
     * out[(6+2C) i ..] = sampled dir_sn[3], dirPDF, dirType, fs(sample)[C], fs(evaluate)[C],      \
     * evaluatePDF; the five sample fields and fs(sample) are zero when dirPDF == 0.           */ \
    int P##bsdf_kat(slr_oracle_scene* s, uint32_t material, uint32_t n, const float* in,           \
                    float wl_offset, float u_lambda, float* out);                                  \
    /* Scene::selectLight + Light::sample as the integrator calls them                             \
     * (PathTracingRenderer.cpp:169-177).  in[3 i ..] = uLightSelection, uPos[2]                   \
     * out[(16+C) i ..] = light (triangle index, -1 = environment sphere), lightProb, p[3],        \
     * gNormal[3], shadingFrame.x[3], shadingFrame.z[3], areaPDF, atInfinity, M[C]              */ \
    int P##light_kat(slr_oracle_scene* s, uint32_t n, const float* in, float wl_offset,            \
                     float u_lambda, float* out);

SLR_ORACLE_DECLARE(slr_oracle_)
SLR_ORACLE_DECLARE(slr_ref_)

/* The reference's own serial mode (one xorshift stream over all pixels and passes, tiles in
 * row-major order: PathTracingRenderer.cpp:33-38,72-81 with numThreads = 1).  Implemented by
 * both so that the restatement can be pinned against the UNMODIFIED reference render().     */
int slr_oracle_render_serial(slr_oracle_scene* s, const slrhip_render_settings* settings, uint32_t spp,
                             float* fb_sum, slr_oracle_counters* counters);
int slr_ref_render_serial(slr_oracle_scene* s, const slrhip_render_settings* settings, uint32_t spp,
                          float* fb_sum, slr_oracle_counters* counters);

/* Timing only: the reference's own multi-threaded render() exactly as shipped (one ThreadPool
 * per pass, 8x8 tile jobs, one xorshift stream per worker: PathTracingRenderer.cpp:27-98) on
 * `threads` workers (<= 0: all online CPUs).  Not reproducible run to run, so its
 * image is only sanity-checked.  Reference build only.                                        */
int slr_ref_render_native(slr_oracle_scene* s, const slrhip_render_settings* settings, uint32_t spp, int threads,
                          float* fb_sum, double* seconds);

/* ImageSensor::saveImage (ImageSensor.cpp:138-186) on a given linear framebuffer: fills the reference's
 * sensor with `fb` ([H][W][C] sums), calls saveImage(path, scale) and returns 0.  `sensitivity` is the
 * ImageSensor constructor argument.  Reference build only; pins the product's slrhip_tonemap_bgr8.     */
int slr_ref_save_image(const float* fb, uint32_t width, uint32_t height, float sensitivity, float scale, const char* path);

/* Published numeric tables the spectral path needs and that only exist inside the reference tree:
 * the Meng-15 upsampling grid (Spectrum.h:199-575: cells + data points), the CIE 1931 2-degree CMFs
 * (xbar/ybar/zbar_2deg, 471 samples), D65 (common_spectra.cpp:185) and the IOR tables of
 * spectrum_library.cpp.  tools/extract_spectral_tables.py dumps them once into slr_amd/data/ as data.
 * what: 0 grid cells (GridWidth*GridHeight x 8 bytes), 1 data points (n x (2+2+95) floats), 2 CMFs (3 x 471),
 * 3 D65 (531), 4+i: IOR table i as [numSamples, min, max, regular?, lambdas..., etas..., ks...].
 * Returns the number of floats/bytes written (or needed when dst is NULL), -1 for an unknown table.  */
long slr_ref_dump_table(int what, const char* name, void* dst, long capacity);

/* SampledSpectrum evaluation of one scene spectrum at the wavelengths of createWithEqualOffsets(offset, .)
 * (ConstantSpectrumTexture::evaluate -> InputSpectrum::evaluate).  out: 16 floats.  Spectral builds.    */
int slr_ref_eval_spectrum(const slrhip_scene_desc* scene, uint32_t spectrum_index, float offset, float* out);
int slr_oracle_eval_spectrum(const slrhip_scene_desc* scene, uint32_t spectrum_index, float offset, float* out);

/* UpsampledContinuousSpectrum(spType, space, e0, e1, e2) constructor (SpectrumTypes.h:180-237): out = u, v, scale.
 * sp_type: 0 Reflectance, 1 Illuminant, 2 IndexOfRefraction; space: 0 sRGB, 1 sRGB_NonLinear, 2 xyY, 3 XYZ.   */
int slr_ref_upsample(int sp_type, int space, float e0, float e1, float e2, float* uvs);

#ifdef __cplusplus
}
#endif
#endif
