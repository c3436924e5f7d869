/*
 * slrhip.h — C ABI of the MI355X-native unidirectional path-tracing integrator.
 *
 * This is the drop-in boundary for ONE hot path of goofoo/SLR:
 *     SLR::PathTracingRenderer::render(const Scene&, const RenderSettings&) const
 *     (reference: libSLR/Renderers/PathTracingRenderer.cpp:27-98, entered from
 *      HostProgram/main.cpp:59 through the Renderer vtable libSLR/Core/Renderer.h:15-19).
 *
 * The reference's Scene is an opaque pointer graph with no accessor for its nodes,
 * triangles or materials (libSLR/Core/SurfaceObject.h:187-204,239-260), so the host
 * layer owns a FLAT scene description and hands it across this ABI.  Every struct
 * below names the reference type it flattens.  Plain pointers and sizes only; no
 * C++ or torch types; every entry point returns an int status (0 = OK) and never
 * throws.  One context per GPU; a context is single-caller (not re-entrant), like
 * the reference's render() (PathTracingRenderer.cpp:27, called once from main).
 */
#ifndef SLRHIP_H
#define SLRHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SLRHIP_VERSION 7   /* 2: slrhip_material::param2, slrhip_scene_desc::upsampling, Ward / Ashikhmin lobes;
                            * 3: SLRHIP_MATERIAL_MULTI, slrhip_bsdf_queries (additive: version-2 callers are unaffected);
                            * 4: stripes > 64 rejected, device error word, samples counted on the device (additive);
                            * 5: slrhip_texture (checkerboard textures, bump, alpha), appended to slrhip_scene_desc; host spectrum
                            *    construction; slrhip_reduce_framebuffer;
                            * 6: SLRHIP_KERNEL_TAIL (slrhip_profile grows by one kernel class), SLRHIP_FLAG_TAIL_KERNEL;
                            * 7: the measured-slower traversal / shading schedules, their flags and the measurement exports are gone
                            *    (TRACE_BATCH, TRACE_POOL, SPECTRAL_QUAD, QUAD_LAYOUT, slrhip_trace_rays_timed, slrhip_debug_read_rays);
                            *    kernel classes of slrhip_profile = {TRACE, SHADE, TAIL}: a wavefront iteration is two launches;
                            *    slrhip_bsdf_queries moved to slrhip_debug.h.  The number is frozen here: later additions append.   */

/* ---- status codes -------------------------------------------------------------- */
enum {
    SLRHIP_OK = 0,
    SLRHIP_ERR_INVALID_ARGUMENT = 1,
    SLRHIP_ERR_NO_DEVICE = 2,      /* no HIP device / HIP runtime failure at create      */
    SLRHIP_ERR_HIP = 3,            /* a HIP call failed; see slrhip_last_error_string    */
    SLRHIP_ERR_NO_SCENE = 4,       /* render before upload_scene                         */
    SLRHIP_ERR_UNSUPPORTED = 5,    /* scene uses a feature outside the hot path          */
    SLRHIP_ERR_OUT_OF_MEMORY = 6
};

/* ---- colour representation ------------------------------------------------------- */
/* The reference selects RGB vs 16-sample spectral rendering at COMPILE time
 * (libSLR/defines.h:160 -> typedefs in libSLR/references.h:39-59).  Here it is a
 * per-context run-time mode.                                                         */
enum {
    SLRHIP_MODE_RGB = 0,           /* SampledSpectrum = RGBTemplate<float>, 3 components  */
    SLRHIP_MODE_SPECTRAL = 1       /* SampledSpectrumTemplate<float,16>, 16-bin storage   */
};
#define SLRHIP_RGB_COMPONENTS 3
#define SLRHIP_SPECTRAL_COMPONENTS 16   /* NumSpectralSamples / NumStrataForStorage, references.h:39-40 */

/* ---- geometry -------------------------------------------------------------------- */
/* SLR::Vertex, libSLR/Core/geometry.h:147-155 (44 bytes, same field order). */
typedef struct slrhip_vertex {
    float position[3];
    float normal[3];
    float tangent[3];
    float texcoord[2];
} slrhip_vertex;

/* One SLR::Triangle (libSLR/Surface/TriangleMesh.h:16-36) wrapped in its
 * SingleSurfaceObject (libSLR/Core/SurfaceObject.h:115-153): three vertex indices and
 * the material of the owning object.  The order of this array is the order in which
 * the reference would receive the objects (it defines light indices and the tie-break
 * among equal-distance hits).                                          */
typedef struct slrhip_triangle {
    uint32_t v[3];
    uint32_t material;
} slrhip_triangle;

/* ---- spectra --------------------------------------------------------------------- */
/* A scene-constant input spectrum (reference InputSpectrum, references.h:51,57).
 * RGB mode reads only `rgb` (RGBTemplate, libSLR/BasicTypes/RGBTypes.h:51-143).
 * Spectral mode evaluates at the path's wavelengths per hit like
 * ConstantSpectrumTexture::evaluate (libSLR/Textures/constant_textures.h:16-31):
 *   kind UPSAMPLED : UpsampledContinuousSpectrum(u, v, scale)  SpectrumTypes.h:180-339
 *   kind REGULAR   : RegularContinuousSpectrum(min,max,values) SpectrumTypes.h:70-118
 *   kind IRREGULAR : IrregularContinuousSpectrum(lambdas,values) SpectrumTypes.h:121-170
 * Sample tables live in slrhip_scene_desc::spectrum_data at [data_offset, +num_samples)
 * (IRREGULAR: num_samples wavelengths followed by num_samples values).
 * UPSAMPLED: the grid-cell lookup of evaluate() (SpectrumTypes.h:241-312) depends only on
 * (u, v), so the caller resolves it once: `reserved` = number of data points (0 = outside
 * the grid, 3 or 4), and the payload at data_offset (a multiple of 4 floats) is 4
 * interpolation weights followed by num_samples (= 95) records of 4 floats: the samples of
 * the data-point spectra interleaved per wavelength bin, [bin][point] (unused points 0).
 * slr_amd/spectra.py:resolve_upsampled builds it.                                       */
enum {
    SLRHIP_SPECTRUM_RGB_ONLY = 0,
    SLRHIP_SPECTRUM_UPSAMPLED = 1,
    SLRHIP_SPECTRUM_REGULAR = 2,
    SLRHIP_SPECTRUM_IRREGULAR = 3
};
typedef struct slrhip_spectrum {
    uint32_t kind;
    float rgb[3];
    float u, v, scale;          /* UPSAMPLED */
    float lambda_min, lambda_max; /* REGULAR */
    uint32_t num_samples;
    uint32_t data_offset;
    uint32_t reserved;
} slrhip_spectrum;

/* ---- materials ------------------------------------------------------------------- */
/* Material -> BSDF factories evaluated per hit in the reference
 * (SurfacePoint::createBSDF, libSLR/Core/geometry.cpp:56-58).                        */
enum {
    /* DiffuseReflection basic_SurfaceMaterials.cpp:15-25: sigma < 0 -> LambertianBRDF
     * (basic_BSDFs.cpp:12-57), sigma >= 0 -> OrenNayerBRDF (OrenNayerBRDF.cpp:12-73).
     * spectrum[0] = reflectance, param = sigma.                                       */
    SLRHIP_MATERIAL_MATTE = 0,
    /* SpecularReflection basic_SurfaceMaterials.cpp:29-34 -> SpecularBRDF + FresnelConductor.
     * spectrum = {coeffR, eta, k}.                                                    */
    SLRHIP_MATERIAL_METAL = 1,
    /* SpecularScattering basic_SurfaceMaterials.cpp:38-43 -> SpecularBSDF + FresnelDielectric.
     * spectrum = {coeff, etaExt, etaInt}.                                             */
    SLRHIP_MATERIAL_GLASS = 2,
    /* MicrofacetReflection MicrofacetSurfaceMaterial.cpp:14-19 -> MicrofacetBRDF(GGX, conductor).
     * spectrum = {-, eta, k}, param = alpha_g.                                        */
    SLRHIP_MATERIAL_MICROFACET_METAL = 3,
    /* MicrofacetScattering MicrofacetSurfaceMaterial.cpp:23-28 -> MicrofacetBSDF(GGX, dielectric).
     * spectrum = {-, etaExt, etaInt}, param = alpha_g.                                */
    SLRHIP_MATERIAL_MICROFACET_GLASS = 4,
    /* ModifiedWardDurReflection SurfaceMaterials/ModifiedWardDurReflection.cpp:14-19 -> ModifiedWardDurBRDF
     * (BSDFs/ModifiedWardDurBRDF.cpp:11-87).  spectrum = {R, -, -}, param = anisoX, param2 = anisoY.        */
    SLRHIP_MATERIAL_WARD = 5,
    /* AshikhminShirleyReflection SurfaceMaterials/AshikhminShirleyReflection.cpp:14-20 -> AshikhminShirleyBRDF
     * (BSDFs/AshikhminShirleyBRDF.cpp:12-170).  spectrum = {Rs, Rd, -}, param = nu, param2 = nv.            */
    SLRHIP_MATERIAL_ASHIKHMIN = 6,
    /* SummedSurfaceMaterial / MixedSurfaceMaterial (SurfaceMaterials/SummedSurfaceMaterial.cpp:13-20,
     * MixedSurfaceMaterial.cpp:14-22) -> MultiBSDF (BSDFs/MultiBSDF.cpp:12-217) over two component BSDFs, either
     * of which may be wrapped in InverseBSDF (InverseSurfaceMaterial basic_SurfaceMaterials.cpp:47-50,
     * basic_BSDFs.cpp:172-203).
     *   spectrum[0], spectrum[1] = indices of the two component MATERIALS: earlier entries of the material
     *                              table; single-lobe types, or MULTI records whose own components are
     *                              single lobes (one level of nesting = up to four lobes, MultiBSDF.h:17:
     *                              sum(mix(a, b), c), mix(sum(a, b), sum(c, d)), ...); their emittance is ignored
     *   spectrum[2]              = SLRHIP_MULTI_INVERSE_0 | SLRHIP_MULTI_INVERSE_1 bits
     *   param, param2            = the `scale` each component's getBSDF receives: 1, 1 for "sum";
     *                              1 - f, f for "mix" with a constant factor f
     * InverseBSDF is limited to single lobes of the reflection-only types (MATTE, METAL, MICROFACET_METAL, WARD, ASHIKHMIN):
     * the two-sided lobes read query.flags inside sampleInternal, which this path fixes at All.             */
    SLRHIP_MATERIAL_MULTI = 7
};
#define SLRHIP_MULTI_INVERSE_0 1
#define SLRHIP_MULTI_INVERSE_1 2
typedef struct slrhip_material {
    uint32_t type;
    int32_t spectrum[3];   /* indices into slrhip_scene_desc::spectra, -1 = unused, SLRHIP_TEXTURE_REF(t) = texture t */
    float param;
    /* EmitterSurfaceMaterial(mat, DiffuseEmission(emittance)) surface_material.h:55-69,
     * DiffuseEmission.cpp:15-21: index of the emittance spectrum, or -1 if not emitting. */
    int32_t emittance;
    float param2;          /* second scalar of the anisotropic lobes (Ward anisoY, Ashikhmin nv), else 0 */
    uint32_t reserved;     /* 0, or SLRHIP_MATERIAL_NORMAL_MAP(t) | SLRHIP_MATERIAL_ALPHA_MAP(t') */
} slrhip_material;

/* ---- textures ------------------------------------------------------- */
/* The procedural textures the reference's libSLR itself holds (Textures/checker_board_textures.{h,cpp}), evaluated per hit at
 * the hit's texture coordinate (Triangle::intersect interpolates it from the ORIGINAL barycentrics, TriangleMesh.cpp:160-161)
 * through a Texture2DMapping (Core/textures.h:16-42): (u, v) -> ((u + offset[0]) * scale[0], (v + offset[1]) * scale[1]);
 * offset 0 / scale 1 is the default mapping.
 *   CHECKER_SPECTRUM  CheckerBoardSpectrumTexture: spectrum[((int)(2 x) + (int)(2 y)) % 2]           checker_board_textures.h:15-27
 *   CHECKER_FLOAT     CheckerBoardFloatTexture:    value[...same index...]                           :43-53
 *   CHECKER_NORMAL    CheckerBoardNormal3DTexture(stepWidth = value[0], reverse = value[1] != 0)     checker_board_textures.cpp:16-43
 * A material's spectrum slot refers to a texture with SLRHIP_TEXTURE_REF(t); its normal map (BumpSingleSurfaceObject,
 * Core/SurfaceObject.cpp:123-134) and its alpha texture (Triangle::m_alphaTex, TriangleMesh.cpp:163-167: a hit where the alpha
 * value is 0 does not occur) ride in slrhip_material::reserved, as the material group of libSLRSceneGraph/TriangleMeshNode
 * pairs them.                                                                                                              */
/*   IMAGE_SPECTRUM    ImageSpectrumTexture: the texel nearest to the mapped coordinate, wrapped by fmod                Textures/image_textures.cpp:13-79
 *                     reserved[0], reserved[1] = width, height; reserved[2] = index of the image's first texel in
 *                     slrhip_scene_desc::texture_texels (3 floats per texel, row-major).  RGB mode: what the look-up of the
 *                     reference's RGB build returns (8-bit formats: byte / 255, RGBA16F: the halves).  Spectral mode: (u, v, s) as
 *                     the spectral build stores the image (Core/Image.h:39-40,265-310), evaluated per hit like
 *                     UpsampledContinuousSpectrum(u, v, s / EqualEnergyReflectance) (image_textures.cpp:23-32) — needs
 *                     slrhip_scene_desc::upsampling.  The image decoding and colour conversion stay with the caller (libSLRSceneGraph's
 *                     image loaders are outside this boundary).                                                                  */
enum { SLRHIP_TEXTURE_CHECKER_SPECTRUM = 0, SLRHIP_TEXTURE_CHECKER_FLOAT = 1, SLRHIP_TEXTURE_CHECKER_NORMAL = 2, SLRHIP_TEXTURE_IMAGE_SPECTRUM = 3 };
typedef struct slrhip_texture {
    uint32_t kind;
    float offset[2];
    float scale[2];
    int32_t spectrum[2];       /* CHECKER_SPECTRUM: indices into slrhip_scene_desc::spectra                    */
    float value[2];            /* CHECKER_FLOAT: the two values; CHECKER_NORMAL: stepWidth in (0, 1], reverse   */
    uint32_t reserved[3];      /* IMAGE_SPECTRUM: width, height, first texel; else 0                                 */
} slrhip_texture;
#define SLRHIP_TEXTURE_REF(t) (-2 - (int32_t)(t))              /* value of slrhip_material::spectrum[k] naming texture t */
#define SLRHIP_MATERIAL_NORMAL_MAP(t) ((uint32_t)(t) + 1u)      /* OR into slrhip_material::reserved: bits 0..15  */
#define SLRHIP_MATERIAL_ALPHA_MAP(t) (((uint32_t)(t) + 1u) << 16)   /* bits 16..31                              */

/* ---- camera ---------------------------------------------------------------------- */
/* SLR::PerspectiveCamera (libSLR/Cameras/PerspectiveCamera.cpp:15-24) with its
 * StaticTransform (libSLR/Core/Transform.h:38-87).  Matrices are column-major
 * (Matrix4x4Template: m[c*4+r], libSLR/BasicTypes/Matrix4x4.h:22-45); both the
 * matrix and its inverse are given because StaticTransform stores both.
 * sensitivity <= 0 selects 1/(pi r^2) as PerspectiveCamera.cpp:23 does.               */
typedef struct slrhip_camera {
    float local_to_world[16];
    float world_to_local[16];
    float aspect;
    float fov_y;
    float lens_radius;
    float img_plane_distance;
    float obj_plane_distance;
    float sensitivity;
} slrhip_camera;

/* ---- environment light ------------------------------------------------------------ */
/* InfiniteSphereSurfaceObject + IBLEmission over an image texture (SurfaceObject.cpp:137-222,
 * SurfaceMaterials/IBLEmission.cpp:15-25, Textures/image_textures.cpp:13-79): a lat-long radiance map looked up
 * at the nearest texel (row 0 = theta 0 = +Y; u = phi / 2 pi), times `scale`, times pi.
 *   texels     : width * height * 3 floats, row-major.  RGB mode: (r, g, b) — the reference stores RGBA16F, use
 *                half-representable values.  Spectral mode: (u, v, s) as the reference's spectral build stores an
 *                RGB image (Upsampling::sRGB_to_uvs per texel, BasicTypes/Spectrum.h:148-171, kept as halves:
 *                Core/Image.h:39-40); every look-up evaluates UpsampledContinuousSpectrum(u, v, s / EqualEnergyReflectance)
 *                at the path's wavelengths (image_textures.cpp:23-32), which needs `upsampling` in the scene description.
 *   importance : map_width * map_height floats = the area-averaged luminance of each map cell, i.e. what
 *                ImageSpectrumTexture::createIBLImportanceMap's pickFunc computes BEFORE the sin(theta) factor
 *                (image_textures.cpp:81-132; map = quarter resolution).  It is an input because it is image
 *                preprocessing (Image2D::areaAverage, Core/Image.cpp:19-120); the library applies sin(theta) and
 *                builds the RegularConstantContinuous2D exactly like Core/distributions.cpp:127-224.            */
typedef struct slrhip_envmap {
    uint32_t width, height;
    const float* texels;
    float scale;
    uint32_t map_width, map_height;
    const float* importance;
} slrhip_envmap;

/* The Meng-15 RGB-upsampling tables (BasicTypes/Spectrum.h:197-575) for spectra whose (u, v) is only known at run time,
 * i.e. environment-map texels in spectral mode: grid_width x grid_height cells of 8 bytes {inside, num_points, idx[6]}
 * (row-major), num_points data points with their (u, v) and their 95-sample spectra (360-830 nm).  Constant spectra do
 * not need it: their cell look-up is resolved by the caller (slrhip_spectrum, kind UPSAMPLED).                       */
typedef struct slrhip_upsampling_tables {
    uint32_t grid_width, grid_height;      /* 12 x 14 */
    const uint8_t* cells;                  /* grid_width * grid_height * 8 bytes */
    uint32_t num_points;
    const float* point_uv;                 /* num_points * 2 */
    const float* point_spectrum;           /* num_points * 95 */
} slrhip_upsampling_tables;

/* ---- instancing ------------------------------------------------------------------- */
/* TransformedSurfaceObject over a mesh's aggregate (libSLR/Core/SurfaceObject.cpp:303-392, StaticTransform only:
 * Transform.h:38-87): the mesh is a RANGE of slrhip_scene_desc::triangles given in the mesh's local space; a ray is taken to
 * local space with world_to_local (`invert(sampledTF) * ray`: origin as a point, direction as a vector, NOT renormalised, so
 * distances stay world distances), intersected with the mesh there, and the surface point comes back through local_to_world
 * (`sampledTF * surfPt`, geometry.cpp:63-78: p as a point, the geometric normal through the inverse transpose, the shading
 * frame's axes as vectors, each re-normalised).  Ranges of two instances are either equal (one mesh, many placements — the
 * mesh's tree is built once) or disjoint; triangles inside an instanced range are not objects of the top-level aggregate.
 * Instanced triangles must not emit.  Matrices are column-major like slrhip_camera's.                                       */
typedef struct slrhip_instance {
    uint32_t first_triangle, num_triangles;
    float local_to_world[16];
    float world_to_local[16];
} slrhip_instance;

/* ---- scene ----------------------------------------------------------------------- */
typedef struct slrhip_scene_desc {
    const slrhip_vertex* vertices;
    uint32_t num_vertices;
    const slrhip_triangle* triangles;
    uint32_t num_triangles;
    const slrhip_material* materials;
    uint32_t num_materials;
    const slrhip_spectrum* spectra;
    uint32_t num_spectra;
    const float* spectrum_data;
    uint32_t num_spectrum_data;
    slrhip_camera camera;
    const slrhip_envmap* env;     /* NULL = no environment sphere (Scene::build envSphere = nullptr) */
    const slrhip_upsampling_tables* upsampling;   /* needed only with an environment map in spectral mode, else may be NULL */
    const slrhip_texture* textures;               /* NULL / 0 = no textured material (version 5)                             */
    uint32_t num_textures;
    const float* texture_texels;                  /* texels of the IMAGE_SPECTRUM textures, 3 floats each (appended in version 7) */
    uint32_t num_texture_texels;                  /* number of TEXELS                                                        */
    const slrhip_instance* instances;             /* NULL / 0 = no instanced mesh (appended in version 7)                    */
    uint32_t num_instances;
} slrhip_scene_desc;

/* ---- render settings -------------------------------------------------------------- */
/* SLR::RenderSettings as read by the path tracer (libSLR/Core/RenderSettings.h:15-22;
 * read at PathTracingRenderer.cpp:33,54-59,88).                                       */
typedef struct slrhip_render_settings {
    int32_t image_width;
    int32_t image_height;
    float time_start;
    float time_end;
    float brightness;
    int32_t rng_seed;
} slrhip_render_settings;

/* ---- image-plane shard ------------------------------------------------------------ */
/* The tile partition of PathTracingRenderer.cpp:74-79 (8x8 tiles, ImageSensor.cpp:12-14)
 * generalised to N devices: this context renders the tiles whose row-major index t
 * satisfies t % shard_count == shard_index.  {0,1} = whole image.                     */
typedef struct slrhip_shard {
    uint32_t shard_index;
    uint32_t shard_count;
} slrhip_shard;

/* ---- context configuration -------------------------------------------------------- */
#define SLRHIP_MAX_STRIPES 64u
typedef struct slrhip_config {
    int32_t device;            /* HIP device ordinal                                         */
    int32_t mode;              /* SLRHIP_MODE_*                                              */
    uint32_t stripes;          /* paths kept in flight PER PIXEL of the shard (the number of path slots = pixels x stripes):
                                * 0 = auto, else 1 .. SLRHIP_MAX_STRIPES; larger values are rejected by slrhip_create with
                                * SLRHIP_ERR_INVALID_ARGUMENT.  It sizes memory and parallelism only: a slot is not bound to
                                * a pixel, and the image is the same to the last bit for every value                      */
    uint32_t flags;            /* SLRHIP_FLAG_*                                              */
} slrhip_config;

/* ---- counters --------------------------------------------------------------------- */
typedef struct slrhip_counters {
    uint64_t samples;            /* (pixel, sample) pairs rendered since slrhip_render_begin, counted on the device
                                  * (what the wave queues handed out)                                       */
    uint64_t extension_rays;     /* Scene::intersect calls         PathTracingRenderer.cpp:147,225 */
    uint64_t shadow_rays;        /* Scene::testVisibility calls    PathTracingRenderer.cpp:180     */
    uint64_t iterations;         /* wavefront iterations launched                            */
    uint64_t bvh_nodes;          /* 4-wide nodes in the flattened tree                       */
    uint64_t bvh_depth;
    double   build_seconds;      /* host BVH build + upload                                  */
    uint64_t bvh_leaf_references;/* triangles referenced from leaves: = triangle count, or more with spatial splits */
} slrhip_counters;

/* ---- per-kernel timing and traversal statistics (measurement) ------------- */
/* Kernel classes of one wavefront iteration. */
enum {
    SLRHIP_KERNEL_TRACE = 0,           /* k_trace_ws: Scene::intersect and Scene::testVisibility of an iteration, one launch */
    SLRHIP_KERNEL_SHADE = 1,           /* k_shade: getSurfacePoint .. bsdf->sample (PathTracingRenderer.cpp:149-258) and, for a path that
                                        * ends, its sample's contribution + Job::kernel's camera ray of the slot's next sample (:100-130) */
    SLRHIP_KERNEL_TAIL = 2,            /* k_tail: the last paths of a render window, each taken to its end by one lane (all of the above
                                        * in one launch, once few slots are live)                        */
    SLRHIP_KERNEL_COUNT = 3
};
typedef struct slrhip_profile {
    uint64_t launches[SLRHIP_KERNEL_COUNT];
    double   milliseconds[SLRHIP_KERNEL_COUNT];   /* sum of HIP-event durations on the render stream  */
    uint64_t rays[2];                             /* [0] extension (closest-hit), [1] shadow rays traced */
    uint64_t nodes[2];                            /* 4-wide nodes fetched (128 B each; 64 B quantized)  */
    uint64_t triangles[2];                        /* leaf triangles tested (48 B each)                 */
    uint64_t slot_visits;                         /* live slots processed by SHADE                     */
} slrhip_profile;

/* config.flags */
#define SLRHIP_FLAG_TAIL_KERNEL 128u    /* with a FIXED slot count (slrhip_config::stripes > 0): also hand the last <= 2^18 live slots of a render
                                         * window (never more than an eighth of the slots) to the tail kernel — one launch instead of the
                                         * last wavefront iterations.  Same samples, same frame to the last bit (the sensor adds a pixel's
                                         * samples in pass order whoever rendered them).  Always on with the automatic slot count
                                         * (stripes = 0); a caller who fixes the count gets the pure wavefront schedule unless he asks      */
#define SLRHIP_FLAG_BVH_SPATIAL_SPLITS 64u /* build the tree with spatial splits (sbvh.cpp; the reference's SBVH, Accelerator/SBVH.h:57-348):
                                         * a triangle straddling a split plane is referenced from both sides with clipped boxes.
                                         * Same hits; fewer triangle tests, more node visits: measured slower with these kernels on
                                         * every BASELINE scene, so the object-split SAH tree stays the default          */
#define SLRHIP_FLAG_BVH_DEVICE_BUILD 4u /* build the accelerator AND the per-triangle records on the GPU (bvh_device.hip: LBVH over 63-bit Morton codes,
                                         * the host build's 4-wide collapse, quantized nodes): 10 M triangles in a fraction of a second instead of
                                         * seconds on the host cores, at the price of a tree of lower quality (more nodes per ray; DESIGN.md has both
                                         * figures).  Same hits.  Automatic from 2^20 triangles on (SLRHIP_BVH=host in the environment keeps the host build);
                                         * scenes with alpha-textured triangles, or fewer than 1024 triangles, always use the host build */
#define SLRHIP_FLAG_TEST_DEVICE_ERROR 16u /* test hook: the next slrhip_render raises the device-side error word, so that the
                                         * error path (SLRHIP_ERR_HIP + message) can be exercised; renders nothing useful */
#define SLRHIP_FLAG_TIME_KERNELS   1u   /* bracket every launch with HIP events (a few us per launch)        */
#define SLRHIP_FLAG_COUNT_TRAVERSAL 2u  /* count nodes / triangles per ray (instrumented kernels, slower)    */

typedef struct slrhip_ctx slrhip_ctx;

/* Replaces: `new PathTracingRenderer(spp)` libSLRSceneGraph/API.cpp:1015-1020 (device side). */
int slrhip_create(const slrhip_config* config, slrhip_ctx** out_ctx);
int slrhip_destroy(slrhip_ctx* ctx);

/* Replaces: the Scene pointer graph handed to render() (SurfaceObjectAggregate ctor
 * SurfaceObject.cpp:226-250 builds the accelerator and light list; Scene::build :396-406).
 * Copies everything; the caller keeps ownership of the host arrays.                       */
int slrhip_upload_scene(slrhip_ctx* ctx, const slrhip_scene_desc* scene);

/* Replaces: sensor->init(W,H) PathTracingRenderer.cpp:67 (ImageSensor.cpp:35-51) plus the
 * per-render setup :33-61.  Clears the accumulation state of this context's shard.        */
int slrhip_render_begin(slrhip_ctx* ctx, const slrhip_render_settings* settings, slrhip_shard shard);

/* Replaces: the pass loop PathTracingRenderer.cpp:72-81 for passes
 * [spp_begin, spp_begin+spp_count).  Sample s of pixel (x,y) draws from the xorshift128
 * stream seeded with slrhip_sample_seed(rng_seed, x, y, s), and the sensor adds the samples
 * of a pixel in pass order (ImageSensor::add's order for one thread), so the image does not
 * depend on the shard layout, the slot count or scheduling — not even in the last bit.  Every
 * sample's contribution is kept until its window of passes is complete (16 B per pixel and
 * pass, 64 B in spectral mode; windows of at most 16 GiB, SLRHIP_RESULT_WINDOW_MB overrides;
 * longer calls are rendered window after window).  The work is ORDERED on `stream` (a hipStream_t, or
 * NULL for the default stream): it starts after what the caller queued there before.  The
 * call itself BLOCKS the host until the passes are done — the number of wavefront iterations
 * is data dependent, so the host polls a device-side "live slots" word between blocks of
 * iterations — and returns SLRHIP_ERR_HIP if a kernel raised the device error word (a
 * bounded spin that gave up, a dropped stack push: never expected, never silent).         */
int slrhip_render(slrhip_ctx* ctx, uint32_t spp_begin, uint32_t spp_count, void* stream);

/* Resolve the accumulated radiance into a linear float framebuffer
 * [height][width][components] = un-normalised SUM over samples of weight*C, exactly what
 * ImageSensor holds (ImageSensor.cpp:124-129; spectral bins already carry the 16/470
 * factor of SpectrumTypes.h:826-835).  Pixels outside the shard are written as 0, so a
 * sum-reduce over shards equals the full image.  `device_dst` is a DEVICE pointer
 * (e.g. a torch tensor's data_ptr, which is how bench.py feeds RCCL).                     */
int slrhip_resolve_framebuffer(slrhip_ctx* ctx, float* device_dst, size_t num_floats, void* stream);

/* Same, to HOST memory (synchronises).  Replaces reading camera->getSensor()->pixel(x,y)
 * (ImageSensor.cpp:88-95).                                                                */
int slrhip_read_framebuffer(slrhip_ctx* ctx, float* host_dst, size_t num_floats);

/* The single exchange of the multi-GPU path: one process per GPU renders its tile shard (slrhip_render_begin's `shard`), then all
 * ranks call this: the resolved frames (zeros outside a rank's tiles) are sum-reduced onto rank `root` with ONE ncclReduce over
 * `nccl_comm` (an ncclComm_t of RCCL the caller created, e.g. ncclCommInitRank; the call is stream-ordered on `stream`).
 * `device_dst` (DEVICE memory, num_floats >= width * height * components) receives the full image on `root`; other ranks may pass
 * NULL.  The reference has no distributed path (one process, a thread pool over 8x8 tiles: PathTracingRenderer.cpp:72-81); this
 * replaces "all tiles of the sensor are filled by this process" (ImageSensor.cpp:124-129) for a sensor spread over ranks.
 * librccl.so is loaded on first use.                                                                                     */
int slrhip_reduce_framebuffer(slrhip_ctx* ctx, void* nccl_comm, int root, float* device_dst, size_t num_floats, void* stream);

int slrhip_synchronize(slrhip_ctx* ctx);
int slrhip_get_counters(slrhip_ctx* ctx, slrhip_counters* out);
int slrhip_components(const slrhip_ctx* ctx);   /* 3 or 16 */
/* Kernel times accumulate over the life of the context; ray / node / triangle totals restart at
 * slrhip_render_begin.  Needs the matching config.flags.                                       */
int slrhip_get_profile(slrhip_ctx* ctx, slrhip_profile* out);

/* Closest-hit queries against the uploaded scene, the aggregate part of
 * Scene::intersect (SurfaceObject.cpp:267-269,408-416).  rays: n x {org[3], dir[3], dist_min,
 * dist_max}; hits: n x {triangle index as uint32 bits (0xFFFFFFFF = miss), dist, b0, b1}
 * (Intersection::dist, ::u, ::v; TriangleMesh.cpp:169-173).  Host arrays; synchronises.        */
int slrhip_trace_rays(slrhip_ctx* ctx, const float* rays, uint32_t n, float* hits);

/* The per-(pixel, sample) seeding contract (pure function, also used by the oracle).      */
int32_t slrhip_sample_seed(int32_t rng_seed, uint32_t pixel_x, uint32_t pixel_y, uint32_t pass);

/* ---- host-side construction of spectral-mode spectra ------------------------------------------------------------------ */
/* SpectrumType / ColorSpace of the reference (BasicTypes/Spectrum.h:17-35), as the scene language's Spectrum(...) passes them. */
enum { SLRHIP_SPECTRUMTYPE_REFLECTANCE = 0, SLRHIP_SPECTRUMTYPE_ILLUMINANT = 1, SLRHIP_SPECTRUMTYPE_IOR = 2 };
enum { SLRHIP_COLORSPACE_SRGB = 0, SLRHIP_COLORSPACE_SRGB_NONLINEAR = 1, SLRHIP_COLORSPACE_XYY = 2, SLRHIP_COLORSPACE_XYZ = 3 };
#define SLRHIP_UPSAMPLING_SAMPLES 95u    /* samples per data-point spectrum, 360-830 nm (Spectrum.h:197) */
/* Replaces: the UpsampledContinuousSpectrum(spType, space, e0, e1, e2) constructor (BasicTypes/SpectrumTypes.h:180-237), which
 * `Spectrum(r, g, b)` of the scene language calls (libSLRSceneGraph/API.cpp:286-441, :1139-1147): writes (u, v, scale).        */
int slrhip_upsample(int32_t spectrum_type, int32_t color_space, float e0, float e1, float e2, float uvs[3]);
/* Replaces: the (u, v)-only half of UpsampledContinuousSpectrum::evaluate (SpectrumTypes.h:241-312).  Writes the number of data
 * points (0, 3 or 4 -> slrhip_spectrum::reserved) and the 4 + 4 * SLRHIP_UPSAMPLING_SAMPLES floats of the UPSAMPLED payload. */
int slrhip_resolve_upsampled(const slrhip_upsampling_tables* tables, float u, float v, uint32_t* num_points, float* payload);

/* Replaces: Spectrum::create(spType, minLambda, maxLambda, values, n) / (spType, lambdas, values, n) of the reference's RGB build
 * (libSLRSceneGraph/API.cpp:1149-1278,1326-1369): integrates the sampled spectrum against the CIE 2-degree colour-matching
 * functions (trapezoids over the union of both sample grids, Kahan sums), normalises by integralCMF, converts XYZ -> linear sRGB
 * (type ILLUMINANT) or sRGB_E (REFLECTANCE, IOR) and clamps negative components: the `rgb` of a slrhip_spectrum of kind REGULAR
 * (lambdas == NULL; min / max wavelength given) or IRREGULAR (lambdas given), i.e. what RGB mode renders named spectra such as
 * Spectrum("ID": "D65") or the refractive-index tables with.                                                              */
int slrhip_spectrum_to_rgb(int32_t spectrum_type, const float* lambdas, float lambda_min, float lambda_max, const float* values,
                           uint32_t num_samples, float rgb[3]);

/* Host-side helpers on the float framebuffer.
 * slrhip_tonemap_bgr8: ImageSensor::saveImage (ImageSensor.cpp:138-186) pixel pipeline:
 * scale*sensitivity -> (spectral: XYZ->sRGB) -> 1-exp(-Y) tone map -> sRGB gamma -> 8-bit BGR,
 * bottom-up rows as the BMP writer stores them.                                            */
int slrhip_tonemap_bgr8(const float* framebuffer, int32_t width, int32_t height, int32_t components,
                        float scale, uint8_t* dst_bgr, size_t dst_bytes);
int slrhip_save_bmp(const char* path, const uint8_t* bgr_bottom_up, int32_t width, int32_t height);

const char* slrhip_last_error_string(void);
int slrhip_version(void);

#ifdef __cplusplus
}
#endif
#endif /* SLRHIP_H */
