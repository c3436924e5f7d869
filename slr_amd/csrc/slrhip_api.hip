// slrhip_api.hip — the C ABI of include/slrhip.h over the HIP kernels.
//
// Host-side responsibilities that the reference spreads over SurfaceObjectAggregate's
// constructor (Core/SurfaceObject.cpp:226-250: accelerator + light list), Scene::build
// (:396-406), PerspectiveCamera's constructor (Cameras/PerspectiveCamera.cpp:15-24) and
// PathTracingRenderer::render's set-up (Renderers/PathTracingRenderer.cpp:27-70) live here:
// flatten, build the 4-wide BVH, upload once, then drive the wavefront iterations.
// There is no CPU fallback: without a HIP device every entry point fails loudly.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/slrhip.h"
#include "bvh.h"
#include "pt_kernels.h"

using namespace slrhip;

namespace {

const int kDefaultPairs = 1;                   // SLRHIP_PAIRS: the ray pair pays, the radiance-sum pair does not
const uint32_t kStatusWords = 8;               // PathBuffers::activeSlots .. tailWords: one small array, read back in one copy
const uint32_t kDefaultRunLength = 64;         // SLRHIP_RUN_LENGTH: passes of a pixel a wave takes in a row (pt_kernels.h WorkItem; measured: DESIGN.md)
#ifndef SLR_TAIL_DIVISOR
#define SLR_TAIL_DIVISOR 8u      // the tail kernel never takes more than this fraction of the slots (1u in a variant build: the tail kernel as the whole renderer, measured in DESIGN.md)
#endif
const uint32_t kDefaultTailSlots = 1u << 18;   // SLRHIP_TAIL_SLOTS: measured on the headline frame and its N = 8 shard

thread_local std::string g_lastError;

int fail(int code, const std::string& msg) {
    g_lastError = msg;
    return code;
}

// Device-side error word (PathBuffers::errorWord): every bounded spin that gives up and every dropped stack push sets a bit, so a
// logic error in a kernel fails the render instead of returning a wrong image with status 0.
int deviceError(uint32_t bits) {
    std::string what = "slrhip_render: device-side error word set:";
    if (bits & ERR_RING_SPACE) what += " [producer gave up waiting for ray-ring space]";
    if (bits & ERR_RING_RELEASE) what += " [consumer gave up waiting for the ring's release watermark]";
    if (bits & ERR_CONSUMER_IDLE) what += " [consumer wave gave up waiting for rays]";
    if (bits & ERR_STACK_OVERFLOW) what += " [traversal stack overflow: a push was dropped]";
    if (bits & ERR_QUEUE_OVERFLOW) what += " [queue region overflow]";
    return fail(SLRHIP_ERR_HIP, what);
}

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess)                                                                           \
            return fail(SLRHIP_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));             \
    } while (0)

// hipMalloc returns 2 MiB-aligned blocks, so the record of slot i would sit at the same offset modulo
// the HBM channel interleave in every per-slot array, and a wave that loads its ten state records back to
// back would queue all of them on one channel ("partition camping").  Each array therefore starts at its
// own skew: a distinct odd multiple of 256 B plus a few KiB.
static std::atomic<size_t> g_skewCounter{0};     // shared by every context of the process; contexts may be set up from several threads

template <typename T>
struct DevArray {
    T* ptr = nullptr;
    void* base = nullptr;
    size_t count = 0, capacity = 0;
    ~DevArray() { release(); }
    void release() { if (base) { (void)hipFree(base); base = nullptr; ptr = nullptr; count = 0; capacity = 0; } }
    hipError_t alloc(size_t n, bool skew = false) {
        if (n == 0) n = 1;
        if (ptr && n <= capacity) { count = n; return hipSuccess; }     // reuse across render_begin calls
        release();
        capacity = n;
        size_t offset = 0;
        if (skew) { size_t k = ++g_skewCounter; offset = (k % 61) * 4352 + (k % 7) * 256; }
        hipError_t e = hipMalloc(&base, n * sizeof(T) + offset);
        if (e == hipSuccess) { ptr = reinterpret_cast<T*>(static_cast<char*>(base) + offset); count = n; }
        return e;
    }
    void adopt(T* devicePtr, size_t n) { release(); base = devicePtr; ptr = devicePtr; count = n; capacity = n; }      // takes ownership of a hipMalloc block
    hipError_t upload(const std::vector<T>& v) {
        hipError_t e = alloc(v.size());
        if (e != hipSuccess || v.empty()) return e;
        return hipMemcpy(ptr, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
    }
};

uint32_t prevPowerOf2(uint32_t x) {   // defines.h:136-143
    x |= x >> 1; x |= x >> 2; x |= x >> 4; x |= x >> 8; x |= x >> 16;
    return x - (x >> 1);
}

// Kahan sum, BasicTypes/CompensatedSum.h:24-30
struct KahanF {
    float result = 0.0f, comp = 0.0f;
    void add(float value) {
        float cInput = value - comp;
        float sumTemp = result + cInput;
        comp = (sumTemp - result) - cInput;
        result = sumTemp;
    }
};

} // namespace

struct slrhip_ctx {
    slrhip_config config;
    int device = 0;
    int numCUs = 256;
    bool haveScene = false;
    bool haveRender = false;

    // scene
    DevArray<QNode> nodes;
    DevArray<QNodeQ> nodesQ;
    DevArray<QNode8> nodes8;
    DevArray<DevTexture> textures;
    DevArray<DevMatTex> matTex;
    DevArray<float4> triUV, alphaTris;
    DevArray<float> texTexels;
    DevArray<LeafTri> leafTris;
    DevArray<ShadeTri> shadeTris;
    DevArray<LightTri> lightTris;
    DevArray<DevInstance> instances;
    DevArray<DevMaterial> materials;
    DevArray<DevMaterialS> materialsS;
    DevArray<DevSpectrum> spectra;
    DevArray<float> spectrumPool;
    DevArray<float> lightPMF, lightCDF;
    DevArray<float4> shadeTables;
    std::vector<float> hostSpectrumPool;      // the padded pool as uploaded (goes into the packed shade tables)
    DevArray<float> envTexels, envTopPDF, envTopCDF, envRowPDF, envRowCDF;
    DevArray<uint8_t> gridCells;
    DevArray<float> pointUV, pointSpectrum;
    DevScene scene;
    uint32_t bvhDepth = 0;
    uint64_t bvhLeafRefs = 0;
    double buildSeconds = 0.0;

    // render state
    slrhip_render_settings settings;
    slrhip_shard shard;
    RenderParams params;
    DevArray<uint32_t> pixelXY;
    DevArray<uint4> rng;
    DevArray<float4> rayOrg, rayDir, hit, alpha, spR, spC, nee, shadowDir;
    DevArray<float4> results, fbSum, fbComp;      // result window of the current render call; the sensor (per-pixel Kahan sums)
    DevArray<uint32_t> cursor, idleShards;
    DevArray<float> pdfPrev;
    DevArray<uint4> hdr;
    DevArray<int32_t> hitInstance;
    DevArray<uint32_t> flags, visible, shadowQueue, tailList, queueCount, activeSlots, blockDead;
    DevArray<uint64_t> totals;
    DevArray<float> resolveScratch;
    PathBuffers buffers;
    uint64_t iterations = 0;
    bool firstRenderCall = true;

    // SLRHIP_FLAG_TIME_KERNELS: 3 events per iteration (before shade, after shade, after trace)
    std::vector<hipEvent_t> events;
    // hipGraph of one block of iterations (slrhip_render): captured on the context's own stream, replayed until no slot is live
    hipStream_t workStream = nullptr;
    hipEvent_t userReady = nullptr;
    uint64_t profLaunches[SLRHIP_KERNEL_COUNT] = {};
    double profMs[SLRHIP_KERNEL_COUNT] = {};
    ~slrhip_ctx() {
        for (hipEvent_t e : events) (void)hipEventDestroy(e);
        if (userReady) (void)hipEventDestroy(userReady);
        if (workStream) (void)hipStreamDestroy(workStream);
    }
};

// Sum the sharded statistics words (pt_kernels.h: totalIndex).
static int readTotals(slrhip_ctx* ctx, uint64_t* out) {
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipDeviceSynchronize());
    std::vector<uint64_t> raw(ctx->totals.count);
    HIP_TRY(hipMemcpy(raw.data(), ctx->totals.ptr, raw.size() * sizeof(uint64_t), hipMemcpyDeviceToHost));
    for (uint32_t k = 0; k < T_KINDS; ++k) {
        out[k] = 0;
        for (uint32_t sh = 0; sh < kShards; ++sh) out[k] += raw[totalIndex(k, sh)];
    }
    if (tuningEnv("SLRHIP_DEBUG_WS") && out[T_WS_STEPS])
        fprintf(stderr, "ws closest: rays %llu nodes %llu tris %llu | consumer wave-steps %llu refills %llu idle spins %llu | "
                        "consumer wave cycles %llu x64, idle %llu x64 | producer waits %llu | node blocks %llu tri blocks %llu active lanes %llu | shadow rays %llu nodes %llu tris %llu\n",
                (unsigned long long)out[T_EXT_RAYS], (unsigned long long)out[T_NODES_CLOSEST], (unsigned long long)out[T_TRIS_CLOSEST],
                (unsigned long long)out[T_WS_STEPS], (unsigned long long)out[T_WS_REFILLS], (unsigned long long)out[T_WS_IDLE_SPINS],
                (unsigned long long)out[T_WS_CYCLES], (unsigned long long)out[T_WS_IDLE_CYCLES], (unsigned long long)out[T_WS_PRODUCER_WAITS],
                (unsigned long long)out[T_WS_NODE_BLOCKS], (unsigned long long)out[T_WS_TRI_BLOCKS], (unsigned long long)out[T_WS_ACTIVE_LANES],
                (unsigned long long)out[T_SHADOW_RAYS], (unsigned long long)out[T_NODES_SHADOW], (unsigned long long)out[T_TRIS_SHADOW]);
    return SLRHIP_OK;
}

// The kernels' hit record is (triangle, t, b1, b2) — Moller-Trumbore's barycentrics; the ABI reports Intersection::u, ::v =
// (b0, b1) with b0 = 1 - b1 - b2 exactly as Triangle::intersect computes it (TriangleMesh.cpp:159,172-173).
static void hitsToUV(float* hits, uint32_t n) {
    for (uint32_t i = 0; i < n; ++i) {
        float* h = hits + (size_t)i * 4;
        const float b1 = h[2], b2 = h[3];
        const float b0 = 1.0f - b1 - b2;
        uint32_t tri; std::memcpy(&tri, h, 4);
        h[2] = tri == 0xFFFFFFFFu ? 0.0f : b0;
        h[3] = tri == 0xFFFFFFFFu ? 0.0f : b1;
    }
}

extern "C" {

const char* slrhip_last_error_string(void) { return g_lastError.c_str(); }
int slrhip_version(void) { return SLRHIP_VERSION; }

int slrhip_create(const slrhip_config* config, slrhip_ctx** out) {
    if (!config || !out) return fail(SLRHIP_ERR_INVALID_ARGUMENT, "slrhip_create: null argument");
    *out = nullptr;
    if (config->mode != SLRHIP_MODE_RGB && config->mode != SLRHIP_MODE_SPECTRAL)
        return fail(SLRHIP_ERR_INVALID_ARGUMENT, "slrhip_create: unknown mode");
    // the per-pixel sample pool tracks the stripes of a pixel in a 64-bit mask (PathBuffers::finishedMask)
    if (config->stripes > SLRHIP_MAX_STRIPES)
        return fail(SLRHIP_ERR_INVALID_ARGUMENT, "slrhip_create: at most 64 sample stripes per pixel (slrhip_config::stripes)");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(SLRHIP_ERR_NO_DEVICE, std::string("slrhip_create: no HIP device (") + hipGetErrorString(e) + ")");
    if (config->device < 0 || config->device >= n) return fail(SLRHIP_ERR_INVALID_ARGUMENT, "slrhip_create: device ordinal out of range");
    HIP_TRY(hipSetDevice(config->device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, config->device));
    slrhip_ctx* ctx = new slrhip_ctx();
    ctx->config = *config;
    ctx->device = config->device;
    ctx->numCUs = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    *out = ctx;
    return SLRHIP_OK;
}

int slrhip_destroy(slrhip_ctx* ctx) {
    if (!ctx) return SLRHIP_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    delete ctx;
    return SLRHIP_OK;
}

int slrhip_components(const slrhip_ctx* ctx) { return ctx && ctx->config.mode == SLRHIP_MODE_SPECTRAL ? 16 : 3; }

int slrhip_upload_scene(slrhip_ctx* ctx, const slrhip_scene_desc* d) {
    if (!ctx || !d) return fail(SLRHIP_ERR_INVALID_ARGUMENT, "slrhip_upload_scene: null argument");
    if (!d->vertices || !d->triangles || !d->materials || !d->spectra || d->num_triangles == 0 || d->num_vertices == 0)
        return fail(SLRHIP_ERR_INVALID_ARGUMENT, "slrhip_upload_scene: empty scene");
    // the Meng-15 tables, wherever they are given: the kernels index with these bytes
    if (ctx->config.mode == SLRHIP_MODE_SPECTRAL && d->upsampling) {
        const slrhip_upsampling_tables* t = d->upsampling;
        if (!t->cells || !t->point_uv || !t->point_spectrum || t->grid_width == 0 || t->grid_height == 0 || t->num_points == 0 || t->num_points > 255)
            return fail(SLRHIP_ERR_INVALID_ARGUMENT, "slrhip_upload_scene: incomplete slrhip_scene_desc::upsampling");
        for (size_t c = 0; c < (size_t)t->grid_width * t->grid_height; ++c) {
            const uint8_t* cell = t->cells + c * 8;
            if (cell[1] > 6) return fail(SLRHIP_ERR_INVALID_ARGUMENT, "slrhip_upload_scene: upsampling cell with more than 6 points");
            for (int k = 0; k < (cell[0] ? 4 : cell[1]); ++k)
                if (cell[2 + k] >= t->num_points) return fail(SLRHIP_ERR_INVALID_ARGUMENT, "slrhip_upload_scene: upsampling point index out of range");
        }
    }
    if (d->env) {
        const slrhip_envmap& e = *d->env;
        if (ctx->config.mode == SLRHIP_MODE_SPECTRAL && !d->upsampling)
            return fail(SLRHIP_ERR_INVALID_ARGUMENT, "slrhip_upload_scene: an environment map in spectral mode needs slrhip_scene_desc::upsampling");
        if (!e.texels || !e.importance || e.width == 0 || e.height == 0 || e.map_width == 0 || e.map_height == 0 ||
            e.width > 32768 || e.height > 32768 || e.map_width > 32768 || e.map_height > 32768)
            return fail(SLRHIP_ERR_INVALID_ARGUMENT, "slrhip_upload_scene: bad environment map");
    }
    HIP_TRY(hipSetDevice(ctx->device));
    auto t0 = std::chrono::steady_clock::now();

    // --- validate indices (a bad index would fault the GPU) --------------------------------------
    for (uint32_t i = 0; i < d->num_triangles; ++i) {
        const slrhip_triangle& t = d->triangles[i];
        if (t.v[0] >= d->num_vertices || t.v[1] >= d->num_vertices || t.v[2] >= d->num_vertices || t.material >= d->num_materials)
            return fail(SLRHIP_ERR_INVALID_ARGUMENT, "slrhip_upload_scene: triangle index out of range");
    }
    const bool spectral = ctx->config.mode == SLRHIP_MODE_SPECTRAL;
    // --- textures: checkerboard spectrum / float / normal textures ----------------------------------------
    const uint32_t numTextures = d->textures ? d->num_textures : 0u;
    if (numTextures > 32767u) return fail(SLRHIP_ERR_UNSUPPORTED, "slrhip_upload_scene: more than 32767 textures");
    std::vector<DevTexture> devTextures(numTextures);
    bool anyImageTexture = false;
    for (uint32_t i = 0; i < numTextures; ++i) {
        const slrhip_texture& t = d->textures[i];
        DevTexture dt;
        std::memset(&dt, 0, sizeof(dt));
        dt.kind = t.kind; dt.ox = t.offset[0]; dt.oy = t.offset[1]; dt.sx = t.scale[0]; dt.sy = t.scale[1];
        dt.v0 = t.value[0]; dt.v1 = t.value[1]; dt.spec0 = dt.spec1 = -1;
        if (t.kind == SLRHIP_TEXTURE_CHECKER_SPECTRUM) {
            for (int k = 0; k < 2; ++k) {
                if (t.spectrum[k] < 0 || (uint32_t)t.spectrum[k] >= d->num_spectra)
                    return fail(SLRHIP_ERR_INVALID_ARGUMENT, "slrhip_upload_scene: texture names a spectrum out of range");
                if (spectral && d->spectra[t.spectrum[k]].kind == SLRHIP_SPECTRUM_RGB_ONLY)
                    return fail(SLRHIP_ERR_INVALID_ARGUMENT, "slrhip_upload_scene: spectral mode needs a spectral descriptor for every spectrum in use");
            }
            dt.spec0 = t.spectrum[0]; dt.spec1 = t.spectrum[1];
            for (int c = 0; c < 3; ++c) { dt.rgb0[c] = d->spectra[t.spectrum[0]].rgb[c]; dt.rgb1[c] = d->spectra[t.spectrum[1]].rgb[c]; }
        }
        else if (t.kind == SLRHIP_TEXTURE_CHECKER_NORMAL) {
            if (!(t.value[0] > 0.0f && t.value[0] <= 1.0f))        // SLRAssert of the reference's constructor
                return fail(SLRHIP_ERR_INVALID_ARGUMENT, "slrhip_upload_scene: checkerboard normal texture needs stepWidth in (0, 1]");
        }
        else if (t.kind == SLRHIP_TEXTURE_IMAGE_SPECTRUM) {
            // ImageSpectrumTexture: width, height and the first texel travel in the record (DevTexture, device_types.h)
            const uint64_t w = t.reserved[0], h = t.reserved[1], first = t.reserved[2];
            if (w == 0 || h == 0 || w > 65535 || h > 65535 || !d->texture_texels || first + w * h > d->num_texture_texels)
                return fail(SLRHIP_ERR_INVALID_ARGUMENT, "slrhip_upload_scene: image texture outside slrhip_scene_desc::texture_texels");
            if (spectral && !d->upsampling)
                return fail(SLRHIP_ERR_INVALID_ARGUMENT, "slrhip_upload_scene: image textures in spectral mode need slrhip_scene_desc::upsampling");
            dt.spec0 = (int32_t)w; dt.spec1 = (int32_t)h; dt.pad = (uint32_t)first;
            anyImageTexture = true;
        }
        else if (t.kind != SLRHIP_TEXTURE_CHECKER_FLOAT) return fail(SLRHIP_ERR_UNSUPPORTED, "slrhip_upload_scene: unknown texture kind");
        devTextures[i] = dt;
    }
    std::vector<DevMatTex> matTex(d->num_materials);
    std::vector<int32_t> alphaOfMaterial(d->num_materials, -1);
    bool anyAlpha = false;
    for (uint32_t i = 0; i < d->num_materials; ++i) {
        const slrhip_material& m = d->materials[i];
        DevMatTex mt = {{-1, -1, -1}, -1};
        const uint32_t nmap = m.reserved & 0xFFFFu, amap = m.reserved >> 16;
        if (nmap > numTextures || amap > numTextures) return fail(SLRHIP_ERR_INVALID_ARGUMENT, "slrhip_upload_scene: material names a texture out of range");
        if (nmap) {
            if (d->textures[nmap - 1].kind != SLRHIP_TEXTURE_CHECKER_NORMAL) return fail(SLRHIP_ERR_INVALID_ARGUMENT, "slrhip_upload_scene: a normal map must be a CHECKER_NORMAL texture");
            mt.normalMap = (int32_t)nmap - 1;
        }
        if (amap) {
            if (d->textures[amap - 1].kind != SLRHIP_TEXTURE_CHECKER_FLOAT) return fail(SLRHIP_ERR_INVALID_ARGUMENT, "slrhip_upload_scene: an alpha map must be a CHECKER_FLOAT texture");
            alphaOfMaterial[i] = (int32_t)amap - 1;
            anyAlpha = true;
        }
        if (m.emittance < -1) return fail(SLRHIP_ERR_UNSUPPORTED, "slrhip_upload_scene: textured emittance is not supported");
        for (int k = 0; k < 3 && m.type != SLRHIP_MATERIAL_MULTI; ++k) {
            if (m.spectrum[k] >= -1) continue;
            const uint32_t t = (uint32_t)(-2 - m.spectrum[k]);
            if (t >= numTextures || (d->textures[t].kind != SLRHIP_TEXTURE_CHECKER_SPECTRUM && d->textures[t].kind != SLRHIP_TEXTURE_IMAGE_SPECTRUM))
                return fail(SLRHIP_ERR_INVALID_ARGUMENT, "slrhip_upload_scene: material spectrum slot names a texture that is not a spectrum texture");
            mt.slot[k] = (int32_t)t;
        }
        matTex[i] = mt;
    }
    std::vector<DevMaterial> mats(d->num_materials);
    std::vector<DevMaterialS> matsS(d->num_materials);
    std::vector<char> emitting(d->num_materials, 0);
    for (uint32_t i = 0; i < d->num_materials; ++i) {
        const slrhip_material& m = d->materials[i];
        DevMaterial dm;
        std::memset(&dm, 0, sizeof(dm));
        dm.type = m.type;
        dm.param = m.param;
        if (m.type > SLRHIP_MATERIAL_MULTI)
            return fail(SLRHIP_ERR_UNSUPPORTED, "slrhip_upload_scene: unknown material type");
        if (m.type == SLRHIP_MATERIAL_MULTI) {
            // MultiBSDF of two earlier materials — single lobes, or MULTI records of single lobes (include/slrhip.h); the record carries indices, scales, flags
            if ((uint32_t)m.spectrum[2] > 3u)
                return fail(SLRHIP_ERR_INVALID_ARGUMENT, "slrhip_upload_scene: MULTI material with unknown inverse bits");
            uint32_t childType[2];
            for (int k = 0; k < 2; ++k) {
                if (m.spectrum[k] < 0 || (uint32_t)m.spectrum[k] >= i)
                    return fail(SLRHIP_ERR_INVALID_ARGUMENT, "slrhip_upload_scene: MULTI component must be an earlier entry of the material table");
                if ((uint32_t)m.spectrum[k] > kMultiMaxChildIndex)
                    return fail(SLRHIP_ERR_UNSUPPORTED, "slrhip_upload_scene: MULTI component index beyond 1023");
                const slrhip_material& cmat = d->materials[m.spectrum[k]];
                childType[k] = cmat.type;
                if (childType[k] == SLRHIP_MATERIAL_MULTI) {
                    // one level of nesting: the components of a component are single lobes (four lobes in all, the reference's
                    // MultiBSDF::maxNumElems); an InverseBSDF over a MultiBSDF is not supported
                    if (d->materials[cmat.spectrum[0]].type >= SLRHIP_MATERIAL_MULTI || d->materials[cmat.spectrum[1]].type >= SLRHIP_MATERIAL_MULTI)
                        return fail(SLRHIP_ERR_UNSUPPORTED, "slrhip_upload_scene: MULTI materials nest one level deep (at most four lobes)");
                    if ((m.spectrum[2] >> k) & 1)
                        return fail(SLRHIP_ERR_UNSUPPORTED, "slrhip_upload_scene: inverse of a MULTI component is not supported");
                }
                if (((m.spectrum[2] >> k) & 1) && (childType[k] == SLRHIP_MATERIAL_GLASS || childType[k] == SLRHIP_MATERIAL_MICROFACET_GLASS))
                    return fail(SLRHIP_ERR_UNSUPPORTED, "slrhip_upload_scene: inverse of a two-sided lobe (glass, microfacet glass) is not supported");
            }
            if (m.emittance >= 0 && (uint32_t)m.emittance >= d->num_spectra)
                return fail(SLRHIP_ERR_INVALID_ARGUMENT, "slrhip_upload_scene: spectrum index out of range");
            if (spectral && m.emittance >= 0 && d->spectra[m.emittance].kind == SLRHIP_SPECTRUM_RGB_ONLY)
                return fail(SLRHIP_ERR_INVALID_ARGUMENT, "slrhip_upload_scene: spectral mode needs a spectral descriptor for every spectrum in use");
            dm.param = 1.0f * m.param;          // `scale * (1.0f - factor)` / `scale * factor` with scale = 1 (MixedSurfaceMaterial.cpp:16-17)
            dm.onA = 1.0f * m.param2;
            const uint32_t bits = packMultiBits((uint32_t)m.spectrum[0], (uint32_t)m.spectrum[1], (uint32_t)m.spectrum[2], childType[0], childType[1]);
            std::memcpy(&dm.onB, &bits, sizeof(bits));
            if (m.emittance >= 0)
                for (int k = 0; k < 3; ++k) dm.emittance[k] = d->spectra[m.emittance].rgb[k];
            emitting[i] = m.emittance >= 0;
            mats[i] = dm;
            DevMaterialS ds;
            ds.type = dm.type; ds.param = dm.param; ds.onA = dm.onA; ds.onB = dm.onB;
            ds.spec[0] = ds.spec[1] = ds.spec[2] = -1; ds.spec[3] = m.emittance;
            matsS[i] = ds;
            continue;
        }
        if (m.type == SLRHIP_MATERIAL_WARD || m.type == SLRHIP_MATERIAL_ASHIKHMIN) {
            dm.onA = m.param2;          // the lobe's second scalar travels in the Oren-Nayar slot (unused by these types)
            if (!(m.param > 0.0f) || !(m.param2 > 0.0f))
                return fail(SLRHIP_ERR_INVALID_ARGUMENT, "slrhip_upload_scene: Ward / Ashikhmin need param > 0 and param2 > 0");
        }
        if (m.type == SLRHIP_MATERIAL_MATTE && m.param >= 0.0f) {
            // OrenNayerBRDF ctor, OrenNayerBRDF.h:28-30: double literals in a float expression
            const float sigma = m.param;
            dm.onA = (float)(1.0f - 0.5f * sigma * sigma / (sigma * sigma + 0.33));
            dm.onB = (float)(0.45 * sigma * sigma / (sigma * sigma + 0.09));
        }
        if ((m.type == SLRHIP_MATERIAL_MICROFACET_METAL || m.type == SLRHIP_MATERIAL_MICROFACET_GLASS) && !(m.param > 0.0f))
            return fail(SLRHIP_ERR_INVALID_ARGUMENT, "slrhip_upload_scene: microfacet material needs alpha_g > 0");
        auto fetch = [&](int32_t idx, float* dst) -> bool {
            if (idx < 0) return true;
            if ((uint32_t)idx >= d->num_spectra) return false;
            for (int k = 0; k < 3; ++k) dst[k] = d->spectra[idx].rgb[k];   // RGBTemplate::evaluate RGBTypes.h:124-126
            return true;
        };
        // `scale * spectrum` with scale = 1.0f (basic_SurfaceMaterials.cpp:22,33,42) is exact
        if (!fetch(m.spectrum[0], dm.a) || !fetch(m.spectrum[1], dm.b) || !fetch(m.spectrum[2], dm.c) || !fetch(m.emittance, dm.emittance))
            return fail(SLRHIP_ERR_INVALID_ARGUMENT, "slrhip_upload_scene: spectrum index out of range");
        if (m.spectrum[0] == -1 && m.type <= SLRHIP_MATERIAL_GLASS)
            return fail(SLRHIP_ERR_INVALID_ARGUMENT, "slrhip_upload_scene: material without its first spectrum");
        if (m.type == SLRHIP_MATERIAL_ASHIKHMIN && (m.spectrum[0] == -1 || m.spectrum[1] == -1))
            return fail(SLRHIP_ERR_INVALID_ARGUMENT, "slrhip_upload_scene: Ashikhmin needs Rs and Rd");
        if (m.type == SLRHIP_MATERIAL_WARD && m.spectrum[0] == -1)
            return fail(SLRHIP_ERR_INVALID_ARGUMENT, "slrhip_upload_scene: Ward needs R");
        if (m.type >= SLRHIP_MATERIAL_METAL && m.type <= SLRHIP_MATERIAL_MICROFACET_GLASS && (m.spectrum[1] == -1 || m.spectrum[2] == -1))
            return fail(SLRHIP_ERR_INVALID_ARGUMENT, "slrhip_upload_scene: material without its eta / k spectra");
        emitting[i] = m.emittance >= 0;
        mats[i] = dm;
        DevMaterialS ds;
        ds.type = dm.type; ds.param = dm.param; ds.onA = dm.onA; ds.onB = dm.onB;
        ds.spec[0] = m.spectrum[0]; ds.spec[1] = m.spectrum[1]; ds.spec[2] = m.spectrum[2]; ds.spec[3] = m.emittance;
        matsS[i] = ds;
        if (spectral)
            for (int k = 0; k < 4; ++k)
                if (ds.spec[k] >= 0 && d->spectra[ds.spec[k]].kind == SLRHIP_SPECTRUM_RGB_ONLY)
                    return fail(SLRHIP_ERR_INVALID_ARGUMENT, "slrhip_upload_scene: spectral mode needs a spectral descriptor for every spectrum in use");
    }
    for (uint32_t i = 0; i < d->num_materials; ++i) {
        const DevMatTex& mt = matTex[i];
        if (mt.slot[0] >= 0 || mt.slot[1] >= 0 || mt.slot[2] >= 0 || mt.normalMap >= 0) { mats[i].type |= kMatTexturedBit; matsS[i].type |= kMatTexturedBit; }
    }
    // spectrum table (spectral mode): descriptors + the float pool, bounds-checked here because the kernels index it
    std::vector<DevSpectrum> devSpectra(d->num_spectra);
    for (uint32_t i = 0; i < d->num_spectra && spectral; ++i) {
        const slrhip_spectrum& sp = d->spectra[i];
        DevSpectrum ds;
        std::memset(&ds, 0, sizeof(ds));
        ds.kind = sp.kind; ds.numPoints = sp.reserved; ds.numSamples = sp.num_samples; ds.dataOffset = sp.data_offset;
        ds.scale = sp.scale; ds.lambdaMin = sp.lambda_min; ds.lambdaMax = sp.lambda_max;
        ds.cellOffset = 0xFFFFFFFFu;
        size_t need = 0;
        if (sp.kind == SLRHIP_SPECTRUM_REGULAR) need = sp.num_samples;
        else if (sp.kind == SLRHIP_SPECTRUM_IRREGULAR) need = 2 * (size_t)sp.num_samples;
        else if (sp.kind == SLRHIP_SPECTRUM_UPSAMPLED) need = 4 + 4 * (size_t)sp.num_samples;
        if (sp.kind != SLRHIP_SPECTRUM_RGB_ONLY) {
            if (sp.num_samples < 2 || (size_t)sp.data_offset + need > d->num_spectrum_data || !d->spectrum_data)
                return fail(SLRHIP_ERR_INVALID_ARGUMENT, "slrhip_upload_scene: spectrum data out of range");
            if (sp.kind == SLRHIP_SPECTRUM_UPSAMPLED && sp.data_offset % 4 != 0)
                return fail(SLRHIP_ERR_INVALID_ARGUMENT, "slrhip_upload_scene: upsampled spectrum payload must start at a multiple of 4 floats");
            if (sp.kind == SLRHIP_SPECTRUM_UPSAMPLED && sp.reserved != 0 && sp.reserved != 3 && sp.reserved != 4)
                return fail(SLRHIP_ERR_INVALID_ARGUMENT, "slrhip_upload_scene: upsampled spectrum must resolve to 0, 3 or 4 points");
        }
        devSpectra[i] = ds;
    }

    // --- accelerator -------------------------------------------------------------------------------
    // Host build (binned SAH, bvh.cpp / spatial splits, sbvh.cpp) unless the context asks for the device build (bvh_device.hip: LBVH,
    // the same collapse; for scenes of millions of triangles, where the host build takes seconds).  The device build also writes the
    // per-triangle records on the GPU; it does not cover alpha-textured triangles (their leaf entries are patched on the host).
    static const std::string envBuild = [] { const char* e = getenv("SLRHIP_BVH"); return std::string(e ? e : ""); }();      // "host" / "device": override
    // automatic: from 2^20 triangles on (host build of 10 M triangles: 2.9 s on 16 cores; device: 0.13 s, traversal 5 % slower)
    const bool wantDevice = envBuild == "device" || (ctx->config.flags & SLRHIP_FLAG_BVH_DEVICE_BUILD) != 0 ||
                            (envBuild != "host" && envBuild != "sbvh" && !(ctx->config.flags & SLRHIP_FLAG_BVH_SPATIAL_SPLITS) && d->num_triangles >= (1u << 20));
    const uint32_t numInstances = d->instances ? d->num_instances : 0u;
    const bool deviceBuild = wantDevice && !anyAlpha && d->num_triangles >= 1024 && numInstances == 0;
    QBVH bvh;
    std::vector<DevInstance> devInstances;
    if (numInstances) {
        // instanced meshes (TransformedSurfaceObject, Core/SurfaceObject.cpp:303-392): two-level tree in one node array (bvh.h)
        for (uint32_t k = 0; k < numInstances; ++k) {
            const slrhip_instance& in = d->instances[k];
            if (in.num_triangles == 0 || (uint64_t)in.first_triangle + in.num_triangles > d->num_triangles)
                return fail(SLRHIP_ERR_INVALID_ARGUMENT, "slrhip_upload_scene: instance names triangles out of range");
            for (uint32_t t = 0; t < in.num_triangles; ++t)
                if (emitting[d->triangles[in.first_triangle + t].material])
                    return fail(SLRHIP_ERR_UNSUPPORTED, "slrhip_upload_scene: instanced triangles must not emit");
        }
        std::string err;
        if (buildInstancedQBVH(d->vertices, d->triangles, d->num_triangles, d->instances, numInstances, &bvh, &devInstances, &err) != 0)
            return fail(SLRHIP_ERR_INVALID_ARGUMENT, "slrhip_upload_scene: " + err);
        if (3 * bvh.depth + 1 > 64)
            return fail(SLRHIP_ERR_UNSUPPORTED, "slrhip_upload_scene: two-level tree deeper than the 64-entry traversal stack (QBVH.h:299)");
        if ((uint64_t)bvh.nodes.size() * sizeof(QNode) >= (1ull << 32) || (uint64_t)bvh.leafTris.size() * sizeof(LeafTri) >= (1ull << 32))
            return fail(SLRHIP_ERR_UNSUPPORTED, "slrhip_upload_scene: node or leaf array beyond the 4 GiB the traversal kernels address with 32-bit offsets");
    }
    else if (!deviceBuild) {
        static const bool wide8 = [] { const char* e = tuningEnv("SLRHIP_WIDE8"); return e && std::string(e) == "1"; }();      // measurement: the eight-wide quantized tree
        if (buildQBVH(d->vertices, d->triangles, d->num_triangles, &bvh, (ctx->config.flags & SLRHIP_FLAG_BVH_SPATIAL_SPLITS) != 0, wide8) != 0)
            return fail(SLRHIP_ERR_INVALID_ARGUMENT, "slrhip_upload_scene: BVH build failed");
        if (3 * bvh.depth + 1 > 64)
            return fail(SLRHIP_ERR_UNSUPPORTED, "slrhip_upload_scene: tree deeper than the 64-entry traversal stack (QBVH.h:299)");
        if ((uint64_t)bvh.nodes.size() * sizeof(QNode) >= (1ull << 32) || (uint64_t)bvh.leafTris.size() * sizeof(LeafTri) >= (1ull << 32))
            return fail(SLRHIP_ERR_UNSUPPORTED, "slrhip_upload_scene: node or leaf array beyond the 4 GiB the traversal kernels address with 32-bit offsets");
    }

    // --- per-triangle shading records and the light list (SurfaceObject.cpp:232-249) ---------------------
    std::vector<ShadeTri> shade(deviceBuild ? 0 : d->num_triangles);      // the device build writes these records itself (k_shade_tris)
    std::vector<LightTri> lights;
    std::vector<uint32_t> lightTriangles;
    std::vector<float> importances;
    float lightIntegral = 0.0f;
    for (uint32_t i = 0; i < d->num_triangles; ++i) {
        const slrhip_triangle& t = d->triangles[i];
        if (deviceBuild && !emitting[t.material]) continue;
        const slrhip_vertex &v0 = d->vertices[t.v[0]], &v1 = d->vertices[t.v[1]], &v2 = d->vertices[t.v[2]];
        ShadeTri s;
        std::memset(&s, 0, sizeof(s));
        float e1[3], e2[3];
        for (int a = 0; a < 3; ++a) {
            s.n0[a] = v0.normal[a]; s.n1[a] = v1.normal[a]; s.n2[a] = v2.normal[a];
            s.t0[a] = v0.tangent[a]; s.t1[a] = v1.tangent[a]; s.t2[a] = v2.tangent[a];
            e1[a] = v1.position[a] - v0.position[a];
            e2[a] = v2.position[a] - v0.position[a];
        }
        // normalize(cross(edge01, edge02)) TriangleMesh.cpp:171 — same float ops as the reference, on the host
        float cx = e1[1] * e2[2] - e1[2] * e2[1], cy = e1[2] * e2[0] - e1[0] * e2[2], cz = e1[0] * e2[1] - e1[1] * e2[0];
        float len = std::sqrt(cx * cx + cy * cy + cz * cz);
        float r = 1.0f / len;
        s.gnx = cx * r; s.gny = cy * r; s.gnz = cz * r;
        s.areaPDF = 1.0f / (0.5f * len);                    // 1 / Triangle::area() :217-222
        s.material = t.material;
        s.light = -1;
        if (emitting[t.material]) {
            s.light = (int32_t)lights.size();
            LightTri l;
            std::memset(&l, 0, sizeof(l));
            for (int a = 0; a < 3; ++a) {
                l.p0[a] = v0.position[a]; l.p1[a] = v1.position[a]; l.p2[a] = v2.position[a];
                l.n0[a] = v0.normal[a]; l.n1[a] = v1.normal[a]; l.n2[a] = v2.normal[a];
                l.t0[a] = v0.tangent[a]; l.t1[a] = v1.tangent[a]; l.t2[a] = v2.tangent[a];
            }
            l.tri = i; l.material = t.material; l.areaPDF = s.areaPDF;
            l.gnx = s.gnx; l.gny = s.gny; l.gnz = s.gnz;
            lights.push_back(l);
            lightTriangles.push_back(i);
            importances.push_back(1.0f);                    // SingleSurfaceObject::importance :69-71
        }
        if (!deviceBuild) shade[i] = s;
    }
    if (lights.empty() && !d->env) return fail(SLRHIP_ERR_UNSUPPORTED, "slrhip_upload_scene: scene has no emitting triangle and no environment light");

    // RegularConstantDiscrete1D ctor, Core/distributions.cpp:76-95
    std::vector<float> pmf = importances, cdf(importances.size() + 1, 0.0f);
    {
        KahanF sum;
        for (size_t i = 0; i < pmf.size(); ++i) { sum.add(pmf[i]); cdf[i + 1] = sum.result; }
        float integral = sum.result;
        lightIntegral = integral;
        for (size_t i = 0; i < pmf.size(); ++i) { pmf[i] /= integral; cdf[i + 1] /= integral; }
    }

    // --- environment sphere: texels + the importance distribution -------------------------------------------
    // InfiniteSphereSurfaceObject ctor (SurfaceObject.cpp:137-141) -> IBLEmission::createIBLImportanceMap
    // (IBLEmission.cpp:11-13) -> RegularConstantContinuous2D (Core/distributions.cpp:186-212) over
    // sin(pi (y + 0.5) / mapHeight) * importance (Textures/image_textures.cpp:126-133).
    std::vector<float> envTexels, envTopPDF, envTopCDF, envRowPDF, envRowCDF;
    if (d->env) {
        const slrhip_envmap& e = *d->env;
        const uint32_t mw = e.map_width, mh = e.map_height;
        envTexels.assign(e.texels, e.texels + (size_t)e.width * e.height * 3);
        envRowPDF.resize((size_t)mw * mh);
        envRowCDF.assign((size_t)(mw + 1) * mh, 0.0f);
        envTopPDF.resize(mh);
        envTopCDF.assign(mh + 1, 0.0f);
        // RegularConstantContinuous1D ctor, distributions.cpp:127-147
        auto build1D = [](float* PDF, float* CDF, uint32_t n) -> float {
            KahanF sum;
            CDF[0] = 0.0f;
            for (uint32_t i = 0; i < n; ++i) { sum.add(PDF[i] / n); CDF[i + 1] = sum.result; }
            for (uint32_t i = 0; i < n; ++i) { PDF[i] /= sum.result; CDF[i + 1] /= sum.result; }
            return sum.result;
        };
        for (uint32_t y = 0; y < mh; ++y) {
            float* row = envRowPDF.data() + (size_t)y * mw;
            for (uint32_t x = 0; x < mw; ++x)
                row[x] = (float)(std::sin(M_PI * (y + 0.5f) / mh) * e.importance[(size_t)y * mw + x]);
            envTopPDF[y] = build1D(row, envRowCDF.data() + (size_t)y * (mw + 1), mw);
        }
        build1D(envTopPDF.data(), envTopCDF.data(), mh);
    }

    // --- camera constants, PerspectiveCamera.cpp:15-24, :55 -------------------------------------------------
    DevCamera cam;
    std::memcpy(cam.mat, d->camera.local_to_world, sizeof(cam.mat));
    std::memcpy(cam.matInv, d->camera.world_to_local, sizeof(cam.matInv));
    cam.lensRadius = d->camera.lens_radius;
    cam.imgPlaneDistance = d->camera.img_plane_distance;
    cam.objPlaneDistance = d->camera.obj_plane_distance;
    cam.opHeight = 2.0f * cam.objPlaneDistance * std::tan(d->camera.fov_y * 0.5f);
    cam.opWidth = cam.opHeight * d->camera.aspect;
    cam.imgPlaneArea = (float)((double)(cam.opWidth * cam.opHeight) * std::pow((double)(cam.imgPlaneDistance / cam.objPlaneDistance), 2.0));
    cam.areaPDF = cam.lensRadius > 0.0f ? (float)(1.0f / (M_PI * (double)cam.lensRadius * (double)cam.lensRadius)) : 1.0f;
    cam.sensitivity = d->camera.sensitivity > 0 ? d->camera.sensitivity
                                                : (float)(1.0f / (M_PI * (double)cam.lensRadius * (double)cam.lensRadius));

    // --- upload ----------------------------------------------------------------------------------------------
    // trees beyond the L2 (>= 64 Ki nodes = 8 MiB) are also stored with 8-bit child boxes: half the bytes per node visit
    static const bool noQuant = [] { const char* e = tuningEnv("SLRHIP_QUANT"); return e && std::string(e) == "0"; }();
    static const bool forceQuant = [] { const char* e = tuningEnv("SLRHIP_QUANT"); return e && std::string(e) == "1"; }();   // experiment: small trees too
    bool quant = false, useWide8 = false;
    uint32_t numNodes = 0, treeDepth = 0;
    uint64_t leafRefs = 0;
    if (deviceBuild) {
        // the whole geometry on the GPU: tree, quantized nodes, leaf packets, shading records (bvh_device.hip)
        DeviceGeometry g;
        std::string err;
        // whether the tree will pass 64 Ki nodes is not known before it is built: ask for the quantized records whenever it could
        const bool wantQ = (d->num_triangles >= 65536 * 2 || forceQuant) && !noQuant;
        if (buildGeometryDevice(d->vertices, d->num_vertices, d->triangles, d->num_triangles, lightTriangles.data(), (uint32_t)lightTriangles.size(), wantQ, &g, &err) != 0)
            return fail(SLRHIP_ERR_HIP, "slrhip_upload_scene: " + err);
        ctx->nodes.adopt(g.nodes, g.numNodes);
        ctx->leafTris.adopt(g.leafTris, g.numLeafTris);
        ctx->shadeTris.adopt(g.shadeTris, d->num_triangles);
        quant = g.nodesQ != nullptr && (g.numNodes >= 65536 || forceQuant);
        if (g.nodesQ) { if (quant) ctx->nodesQ.adopt(g.nodesQ, g.numNodes); else (void)hipFree(g.nodesQ); }
        numNodes = g.numNodes; treeDepth = g.depth; leafRefs = g.numLeafTris;
        if (3 * treeDepth + 1 > 64)
            return fail(SLRHIP_ERR_UNSUPPORTED, "slrhip_upload_scene: device-built tree deeper than the 64-entry traversal stack (QBVH.h:299); use the host build");
        if ((uint64_t)numNodes * sizeof(QNode) >= (1ull << 32) || leafRefs * sizeof(LeafTri) >= (1ull << 32))
            return fail(SLRHIP_ERR_UNSUPPORTED, "slrhip_upload_scene: node or leaf array beyond the 4 GiB the traversal kernels address with 32-bit offsets");
    }
    else {
        HIP_TRY(ctx->nodes.upload(bvh.nodes));
        quant = (bvh.nodes.size() >= 65536 || forceQuant) && !noQuant && numInstances == 0;      // instanced scenes traverse float nodes
        if (quant) { quantizeNodes(&bvh); HIP_TRY(ctx->nodesQ.upload(bvh.quantized)); }
        numNodes = (uint32_t)bvh.nodes.size(); treeDepth = bvh.depth; leafRefs = bvh.leafTris.size();
        useWide8 = !bvh.nodes8.empty() && 7 * bvh.depth8 + 1 <= 64;          // up to seven pushes per level on the 64-entry stack
        if (useWide8) HIP_TRY(ctx->nodes8.upload(bvh.nodes8));
    }
    // alpha textures (Triangle::m_alphaTex): one record per triangle that has one, named by its leaf entries; texture coordinates
    // of every triangle for the textured shading kernels
    std::vector<float4> alphaTris, triUV;
    if (anyAlpha) {
        std::vector<uint32_t> recordOf(d->num_triangles, kNoAlpha);
        for (uint32_t i = 0; i < d->num_triangles; ++i) {
            const int32_t a = alphaOfMaterial[d->triangles[i].material];
            if (a < 0) continue;
            recordOf[i] = (uint32_t)(alphaTris.size() / 2);
            const slrhip_triangle& t = d->triangles[i];
            const float* u0 = d->vertices[t.v[0]].texcoord; const float* u1 = d->vertices[t.v[1]].texcoord; const float* u2 = d->vertices[t.v[2]].texcoord;
            float texIdx; const uint32_t bits = (uint32_t)a; std::memcpy(&texIdx, &bits, 4);
            alphaTris.push_back(make_float4(u0[0], u0[1], u1[0], u1[1]));
            alphaTris.push_back(make_float4(u2[0], u2[1], texIdx, 0.0f));
        }
        for (LeafTri& lt : bvh.leafTris) lt.alpha = recordOf[lt.tri];
    }
    if (numTextures) {
        triUV.resize((size_t)d->num_triangles * 2);
        for (uint32_t i = 0; i < d->num_triangles; ++i) {
            const slrhip_triangle& t = d->triangles[i];
            const float* u0 = d->vertices[t.v[0]].texcoord; const float* u1 = d->vertices[t.v[1]].texcoord; const float* u2 = d->vertices[t.v[2]].texcoord;
            triUV[(size_t)i * 2] = make_float4(u0[0], u0[1], u1[0], u1[1]);
            triUV[(size_t)i * 2 + 1] = make_float4(u2[0], u2[1], 0.0f, 0.0f);
        }
    }
    if (!deviceBuild) HIP_TRY(ctx->leafTris.upload(bvh.leafTris));
    HIP_TRY(ctx->textures.upload(devTextures));
    {
        std::vector<float> texels;
        if (anyImageTexture) texels.assign(d->texture_texels, d->texture_texels + (size_t)d->num_texture_texels * 3);
        HIP_TRY(ctx->texTexels.upload(texels));
    }
    HIP_TRY(ctx->matTex.upload(matTex));
    HIP_TRY(ctx->triUV.upload(triUV));
    HIP_TRY(ctx->alphaTris.upload(alphaTris));
    if (!deviceBuild) HIP_TRY(ctx->shadeTris.upload(shade));
    HIP_TRY(ctx->lightTris.upload(lights));
    HIP_TRY(ctx->instances.upload(devInstances));
    HIP_TRY(ctx->materials.upload(mats));
    HIP_TRY(ctx->materialsS.upload(matsS));
    {
        std::vector<float> pool;
        if (spectral && d->spectrum_data) pool.assign(d->spectrum_data, d->spectrum_data + d->num_spectrum_data);
        // Irregular spectra are evaluated with std::lower_bound per wavelength (SpectrumTypes.h:143-146): on the GPU that is a
        // chain of dependent loads per component.  A path's wavelengths lie in [360, 830], so for every 1-nm cell the host
        // stores lower_bound(cell start) as one byte: the device starts there and walks at most a step or two — the same index.
        for (uint32_t i = 0; i < d->num_spectra && spectral; ++i) {
            const slrhip_spectrum& sp = d->spectra[i];
            if (sp.kind != SLRHIP_SPECTRUM_IRREGULAR || sp.num_samples > 255) continue;
            while (pool.size() % 4) pool.push_back(0.0f);
            const float* lambdas = d->spectrum_data + sp.data_offset;
            const uint32_t cells = 472;                                   // 360 + j, j = 0 .. 471
            std::vector<uint32_t> words(cells / 4, 0u);
            for (uint32_t j = 0; j < cells; ++j) {
                const float start = 360.0f + (float)j;
                const uint32_t lb = (uint32_t)(std::lower_bound(lambdas, lambdas + sp.num_samples, start) - lambdas);
                words[j / 4] |= lb << (8 * (j % 4));
            }
            devSpectra[i].cellOffset = (uint32_t)pool.size();
            for (uint32_t w : words) { float f; std::memcpy(&f, &w, 4); pool.push_back(f); }
        }
        while (pool.size() % 4) pool.push_back(0.0f);          // the shade kernel stages the pool into LDS 16 bytes at a time
        ctx->scene.numSpectrumData = (uint32_t)pool.size();
        HIP_TRY(ctx->spectrumPool.upload(pool));
        ctx->hostSpectrumPool.swap(pool);
    }
    HIP_TRY(ctx->spectra.upload(devSpectra));
    HIP_TRY(ctx->lightPMF.upload(pmf));
    HIP_TRY(ctx->lightCDF.upload(cdf));
    {
        // the tables the shade kernels stage in LDS, packed in the order of ShadeLds' segments (DevScene::shadeTables)
        const bool fits = mats.size() <= (size_t)kLdsMaterials && lights.size() <= (size_t)kLdsLights &&
                          (!spectral || (devSpectra.size() <= (size_t)kLdsSpectra && ctx->scene.numSpectrumData <= (uint32_t)kLdsPoolFloats));
        std::vector<float4> blob;
        uint32_t seg = 0;
        for (uint32_t k = 0; k < 6; ++k) ctx->scene.tableEnd[k] = 0;
        const auto append = [&](const void* src, size_t bytes) {
            const size_t n = (bytes + 15) / 16, at = blob.size();
            blob.resize(at + n, make_float4(0.0f, 0.0f, 0.0f, 0.0f));
            if (bytes) std::memcpy(blob.data() + at, src, bytes);
            ctx->scene.tableEnd[seg++] = (uint32_t)blob.size();
        };
        if (fits) {
            if (spectral) {
                append(matsS.data(), matsS.size() * sizeof(DevMaterialS));
                append(devSpectra.data(), devSpectra.size() * sizeof(DevSpectrum));
                append(ctx->hostSpectrumPool.data(), ctx->hostSpectrumPool.size() * sizeof(float));
            }
            else append(mats.data(), mats.size() * sizeof(DevMaterial));
            append(lights.data(), lights.size() * sizeof(LightTri));
            append(pmf.data(), pmf.size() * sizeof(float));
            append(cdf.data(), cdf.size() * sizeof(float));
            while (seg < 6) { ctx->scene.tableEnd[seg] = (uint32_t)blob.size(); ++seg; }
        }
        HIP_TRY(ctx->shadeTables.upload(blob));
        ctx->scene.shadeTables = fits ? ctx->shadeTables.ptr : nullptr;
    }
    HIP_TRY(ctx->envTexels.upload(envTexels));
    HIP_TRY(ctx->envTopPDF.upload(envTopPDF)); HIP_TRY(ctx->envTopCDF.upload(envTopCDF));
    HIP_TRY(ctx->envRowPDF.upload(envRowPDF)); HIP_TRY(ctx->envRowCDF.upload(envRowCDF));
    DevScene& sc = ctx->scene;
    sc.nodes = reinterpret_cast<const float4*>(ctx->nodes.ptr);
    sc.leafTris = reinterpret_cast<const float4*>(ctx->leafTris.ptr);
    sc.shadeTris = ctx->shadeTris.ptr;
    sc.lightTris = ctx->lightTris.ptr;
    sc.materials = ctx->materials.ptr;
    sc.materialsS = ctx->materialsS.ptr;
    sc.spectra = ctx->spectra.ptr;
    sc.spectrumPool = ctx->spectrumPool.ptr;
    sc.lightPMF = ctx->lightPMF.ptr;
    sc.lightCDF = ctx->lightCDF.ptr;
    sc.textures = ctx->textures.ptr; sc.matTex = ctx->matTex.ptr; sc.triUV = ctx->triUV.ptr; sc.alphaTris = ctx->alphaTris.ptr; sc.texTexels = ctx->texTexels.ptr;
    sc.numTextures = numTextures;
    sc.instances = numInstances ? reinterpret_cast<const float4*>(ctx->instances.ptr) : nullptr;
    sc.numInstances = numInstances;
    sc.numNodes = numNodes;
    sc.nodesQ = quant ? reinterpret_cast<const float4*>(ctx->nodesQ.ptr) : nullptr;
    sc.nodes8 = useWide8 ? reinterpret_cast<const float4*>(ctx->nodes8.ptr) : nullptr;
    sc.numMaterials = (uint32_t)mats.size();
    sc.numSpectra = (uint32_t)devSpectra.size();
    sc.numLights = (uint32_t)lights.size();
    sc.hasMicrofacet = 0;
    sc.hasMulti = 0;
    for (const DevMaterial& dm : mats) {
        if ((dm.type & 0xFFu) >= SLRHIP_MATERIAL_MICROFACET_METAL) sc.hasMicrofacet = 1;     // GGX, Ward, Ashikhmin: the kernels with the glossy-lobe code
        if ((dm.type & 0xFFu) == SLRHIP_MATERIAL_MULTI) sc.hasMulti = 1;
    }
    sc.lightPow2 = prevPowerOf2(sc.numLights);
    sc.hasEnv = d->env ? 1u : 0u;
    sc.aggImportance = lightIntegral;
    if (d->env) {
        sc.envWidth = d->env->width; sc.envHeight = d->env->height;
        sc.envMapWidth = d->env->map_width; sc.envMapHeight = d->env->map_height;
        sc.envScale = d->env->scale;
    }
    {
        std::vector<uint8_t> cells;
        std::vector<float> puv, psp;
        sc.gridWidth = sc.gridHeight = 0;
        if ((d->env || anyImageTexture) && spectral) {
            const slrhip_upsampling_tables* t = d->upsampling;
            cells.assign(t->cells, t->cells + (size_t)t->grid_width * t->grid_height * 8);
            puv.assign(t->point_uv, t->point_uv + (size_t)t->num_points * 2);
            psp.assign(t->point_spectrum, t->point_spectrum + (size_t)t->num_points * 95);
            sc.gridWidth = t->grid_width; sc.gridHeight = t->grid_height;
        }
        HIP_TRY(ctx->gridCells.upload(cells)); HIP_TRY(ctx->pointUV.upload(puv)); HIP_TRY(ctx->pointSpectrum.upload(psp));
        sc.gridCells = ctx->gridCells.ptr; sc.pointUV = ctx->pointUV.ptr; sc.pointSpectrum = ctx->pointSpectrum.ptr;
    }
    sc.envTexels = ctx->envTexels.ptr;
    sc.envTopPDF = ctx->envTopPDF.ptr; sc.envTopCDF = ctx->envTopCDF.ptr;
    sc.envRowPDF = ctx->envRowPDF.ptr; sc.envRowCDF = ctx->envRowCDF.ptr;
    sc.camera = cam;
    ctx->bvhDepth = treeDepth;
    ctx->bvhLeafRefs = leafRefs;
    ctx->haveScene = true;
    ctx->haveRender = false;
    ctx->buildSeconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return SLRHIP_OK;
}

int slrhip_render_begin(slrhip_ctx* ctx, const slrhip_render_settings* st, slrhip_shard shard) {
    if (!ctx || !st) return fail(SLRHIP_ERR_INVALID_ARGUMENT, "slrhip_render_begin: null argument");
    if (!ctx->haveScene) return fail(SLRHIP_ERR_NO_SCENE, "slrhip_render_begin: no scene uploaded");
    if (st->image_width <= 0 || st->image_height <= 0 || st->image_width > 65535 || st->image_height > 65535)
        return fail(SLRHIP_ERR_INVALID_ARGUMENT, "slrhip_render_begin: image size out of range");
    if (shard.shard_count == 0 || shard.shard_index >= shard.shard_count)
        return fail(SLRHIP_ERR_INVALID_ARGUMENT, "slrhip_render_begin: bad shard");
    HIP_TRY(hipSetDevice(ctx->device));
    const uint32_t W = (uint32_t)st->image_width, H = (uint32_t)st->image_height;
    const uint32_t tilesX = (W + 7) >> 3, tilesY = (H + 7) >> 3;     // ImageSensor.cpp:43-44, 8x8 tiles
    // pixel list of this shard: tiles t with t % count == index, row-major inside each tile, so 64
    // consecutive slots (one wavefront) are one 8x8 tile
    std::vector<uint32_t> pixels;
    for (uint32_t t = shard.shard_index; t < tilesX * tilesY; t += shard.shard_count) {
        uint32_t tx = t % tilesX, ty = t / tilesX;
        for (uint32_t ly = 0; ly < 8; ++ly)
            for (uint32_t lx = 0; lx < 8; ++lx) {
                uint32_t x = tx * 8 + lx, y = ty * 8 + ly;
                if (x < W && y < H) pixels.push_back(x | (y << 16));
            }
    }
    if (pixels.empty()) pixels.push_back(0xFFFFFFFFu);   // an empty shard keeps the buffers valid; numPixels = 0 below
    const uint32_t numPixels = pixels[0] == 0xFFFFFFFFu ? 0u : (uint32_t)pixels.size();
    uint32_t stripes = ctx->config.stripes;
    if (stripes == 0) {
        // Paths in flight: throughput keeps rising with the slot count (longer launches amortise the per-wave tail of the
        // traversal kernel: 807 / 1146 / 1267 Msamples/s at 0.9 / 3.7 / 7.4 M slots on the 1280x720 Cornell scene in round 1),
        // at ~200 B of HBM per slot; the per-pixel sample pool keeps the stripes of a pixel finishing together, so fewer, fuller
        // iterations keep paying.  The count is a power of two (the stripes of a pixel then fill whole lane groups of the shade
        // workgroup, PathBuffers): the smallest that reaches ~22 M slots in RGB mode, ~7.4 M in spectral mode (492 B per slot; its
        // shade kernel is latency-bound, not launch-bound), at most 64 (the width of the pool's mask; also the best count measured
        // for the eighth of the image a rank owns at N = 8).  Measured with the fused shade kernel, 16 vs 32 stripes at 1280x720
        // (profiles/r03_e_*): Cornell 2 709 vs 2 719, environment light 5 630 vs 6 061, 10 M-triangle grid 1 954 vs 1 981 Msamples/s.
        const uint32_t target = ctx->config.mode == SLRHIP_MODE_SPECTRAL ? 7372800u : 22118400u;
        static const long envStripes = [] { const char* e = tuningEnv("SLRHIP_AUTO_STRIPES"); return e ? atol(e) : 0L; }();      // measurement: force the automatic choice
        stripes = 1u;
        while (stripes < 64u && (uint64_t)numPixels * stripes < target) stripes *= 2u;
        if (numPixels == 0) stripes = 1u;
        if (envStripes >= 1 && envStripes <= 64) stripes = (uint32_t)envStripes;
    }
    // Slots are paths in flight, not places in the image (pt_kernels.h): `stripes` only sizes their number
    const size_t numSlots = std::max<size_t>(((size_t)numPixels * stripes + 255u) / 256u, 1) * 256u;
    if (numSlots > 0x7FFFFFFFull) return fail(SLRHIP_ERR_INVALID_ARGUMENT, "slrhip_render_begin: too many path slots");

    const bool spectral = ctx->config.mode == SLRHIP_MODE_SPECTRAL;
    const size_t planes = spectral ? 4 : 1;
    HIP_TRY(ctx->pixelXY.upload(pixels));
    // SLRHIP_PAIRS (bit 0: ray origin + direction, bit 1: the path's radiance sum + its compensation, bit 2: sample header + RNG state): the two records of a pair
    // interleaved in one array, one 32-byte sector per slot (PathBuffers::rayStride / spStride)
    static const int envPairs = [] { const char* e = tuningEnv("SLRHIP_PAIRS"); return e ? atoi(e) : kDefaultPairs; }();
    const uint32_t rayStride = (envPairs & 1) ? 2u : 1u, spStride = (envPairs & 2) ? 2u : 1u, hdrStride = (envPairs & 4) ? 2u : 1u;
    HIP_TRY(ctx->rayOrg.alloc(numSlots * rayStride, true)); HIP_TRY(ctx->rayDir.alloc(rayStride == 2 ? 1 : numSlots, true)); HIP_TRY(ctx->hit.alloc(numSlots, true));
    HIP_TRY(ctx->alpha.alloc(numSlots * planes, true)); HIP_TRY(ctx->spR.alloc(numSlots * planes * spStride, true)); HIP_TRY(ctx->spC.alloc(spStride == 2 ? 1 : numSlots * planes, true));
    HIP_TRY(ctx->fbSum.alloc((size_t)std::max(numPixels, 1u) * planes, true)); HIP_TRY(ctx->fbComp.alloc((size_t)std::max(numPixels, 1u) * planes, true));
    HIP_TRY(ctx->nee.alloc(numSlots * planes, true));
    HIP_TRY(ctx->shadowDir.alloc(numSlots, true));
    HIP_TRY(ctx->pdfPrev.alloc(spectral ? numSlots : 1, true));
    HIP_TRY(ctx->hdr.alloc(numSlots * hdrStride, true)); HIP_TRY(ctx->rng.alloc(hdrStride == 2 ? 1 : numSlots, true));
    HIP_TRY(ctx->cursor.alloc(numSlots / 64u)); HIP_TRY(ctx->idleShards.alloc(kShards * kCounterStride));
    HIP_TRY(ctx->flags.alloc(numSlots, true)); HIP_TRY(ctx->visible.alloc(numSlots, true));
    if (ctx->scene.instances) HIP_TRY(ctx->hitInstance.alloc(numSlots, true));
    // queue regions: slot block b appends to region b % kShards, so a region holds at most ceil(numBlocks / kShards) blocks
    const uint32_t numBlocks = (uint32_t)((numSlots + 255) / 256);
    const uint32_t shardCapacity = ((numBlocks + kShards - 1) / kShards) * 256;
    HIP_TRY(ctx->shadowQueue.alloc((size_t)shardCapacity * kShards, true)); HIP_TRY(ctx->tailList.alloc(numSlots, true));
    HIP_TRY(ctx->blockDead.alloc(numBlocks));
    HIP_TRY(ctx->queueCount.alloc(2 * kQueueSetWords)); HIP_TRY(ctx->activeSlots.alloc(kStatusWords));      // [0] live slots, [1] device error word, [2] unused, [3] tail mode (1 + parity), [4..5] tail list length / cursor
    HIP_TRY(ctx->totals.alloc((size_t)T_KINDS * kShards * kTotalStride));
    // The statistics restart here.  A memset of device memory is only ordered on the null stream, and slrhip_render may be
    // given a NON-BLOCKING stream, which the null stream does not wait for and which does not wait for it: the memset has to
    // be complete before this call returns.  (Round 1, gpurun_out/overlap.log: this function also cleared queueCount with a
    // second null-stream memset and did not wait; on a non-blocking stream k_reset_slots — which writes the initial regen
    // counts into queueCount — could run BEFORE that memset landed, the counts were wiped, no slot ever started a sample and
    // slrhip_render ran into its iteration bound.  queueCount is now written by k_reset_slots alone, in stream order.)
    HIP_TRY(hipMemsetAsync(ctx->totals.ptr, 0, ctx->totals.count * sizeof(uint64_t), nullptr));
    HIP_TRY(hipMemsetAsync(ctx->activeSlots.ptr, 0, kStatusWords * sizeof(uint32_t), nullptr));
    HIP_TRY(hipStreamSynchronize(nullptr));

    PathBuffers& pb = ctx->buffers;
    pb.rng = hdrStride == 2 ? ctx->hdr.ptr + 1 : ctx->rng.ptr; pb.hdrStride = hdrStride; pb.rayOrg = ctx->rayOrg.ptr; pb.rayDir = rayStride == 2 ? ctx->rayOrg.ptr + 1 : ctx->rayDir.ptr; pb.hit = ctx->hit.ptr;
    pb.rayStride = rayStride; pb.spStride = spStride;
    pb.alpha = ctx->alpha.ptr; pb.spR = ctx->spR.ptr; pb.spC = spStride == 2 ? ctx->spR.ptr + 1 : ctx->spC.ptr; pb.fbSum = ctx->fbSum.ptr; pb.fbComp = ctx->fbComp.ptr; pb.results = ctx->results.ptr; pb.cursor = ctx->cursor.ptr;
    pb.nee = ctx->nee.ptr;
    pb.pdfPrev = spectral ? ctx->pdfPrev.ptr : nullptr; pb.hdr = ctx->hdr.ptr; pb.shadowDir = ctx->shadowDir.ptr; pb.flags = ctx->flags.ptr;
    pb.hitInstance = ctx->scene.instances ? ctx->hitInstance.ptr : nullptr;
    pb.visible = ctx->visible.ptr; pb.shadowQueue = ctx->shadowQueue.ptr; pb.tailList = ctx->tailList.ptr;
    pb.queueCount = ctx->queueCount.ptr; pb.activeSlots = ctx->activeSlots.ptr; pb.errorWord = ctx->activeSlots.ptr + 1; pb.tailMode = ctx->activeSlots.ptr + 3; pb.tailWords = ctx->activeSlots.ptr + 4; pb.tailIdled = ctx->activeSlots.ptr + 2; pb.windowSamples = ctx->activeSlots.ptr + 6; pb.idleShards = ctx->idleShards.ptr; pb.blockDead = ctx->blockDead.ptr; pb.totals = ctx->totals.ptr;
    pb.pixelXY = ctx->pixelXY.ptr;

    RenderParams& rp = ctx->params;
    rp.numSlots = numPixels ? (uint32_t)numSlots : 0u; rp.numBlocks = rp.numSlots / 256u; rp.numPixels = numPixels; rp.stripes = stripes;
    rp.sppBegin = 0; rp.sppCount = 0; rp.workItems = 0; rp.numWaves = rp.numSlots / 64u; rp.runLength = 1; rp.numRuns = 0;
    rp.rngSeed = st->rng_seed; rp.timeStart = st->time_start; rp.timeEnd = st->time_end;
    rp.imageWidth = W; rp.imageHeight = H;
    rp.countSlots = (ctx->config.flags & SLRHIP_FLAG_COUNT_TRAVERSAL) ? 1u : 0u;
    rp.shardCapacity = shardCapacity;
    rp.spectral = spectral ? 1u : 0u;
    rp.injectError = (ctx->config.flags & SLRHIP_FLAG_TEST_DEVICE_ERROR) ? 1u : 0u;
    ctx->settings = *st;
    ctx->shard = shard;
    ctx->iterations = 0;
    ctx->firstRenderCall = true;
    ctx->haveRender = true;
    return SLRHIP_OK;
}

// One window of passes [sppBegin, sppBegin + sppCount): every sample of the window rendered into the result window, then folded
// into the sensor in pass order.  slrhip_render sizes the windows.
static int renderWindow(slrhip_ctx* ctx, uint32_t sppBegin, uint32_t sppCount, hipStream_t stream) {
    RenderParams& rp = ctx->params;
    rp.sppBegin = sppBegin;
    rp.sppCount = sppCount;
    rp.workItems = rp.numPixels * sppCount;                        // < 2^32: slrhip_render
    // passes per run (pt_kernels.h WorkItem): the largest power of two <= the default that divides the window's pass count
    static const uint32_t envRun = [] { const char* e = getenv("SLRHIP_RUN_LENGTH"); const long v = e ? atol(e) : 0L; return v >= 1 && v <= 4096 ? (uint32_t)v : 0u; }();
    rp.runLength = envRun ? envRun : kDefaultRunLength;
    while (rp.runLength > 1 && (sppCount % rp.runLength) != 0) rp.runLength /= 2;
    if (sppCount && sppCount % rp.runLength) rp.runLength = 1;
    rp.numRuns = rp.numPixels * (sppCount / std::max(rp.runLength, 1u));

    // the first window after render_begin also clears the sensor (the buffers are reused across render_begin calls), even when
    // it is asked for zero passes
    launchResetSlots(ctx->buffers, rp, ctx->firstRenderCall, stream);
    ctx->firstRenderCall = false;
    if (sppCount == 0) return SLRHIP_OK;
    // persistent traversal workgroups of the wave-specialised kernel (pt_trace_ws.hip): a fixed number per CU
    const uint32_t traceBlocks = (uint32_t)ctx->numCUs * (uint32_t)traceWsBlocksPerCU(ctx->scene.nodesQ != nullptr);

    uint32_t active = rp.numSlots;
    uint32_t status[4] = {active, 0u, 0u, 0u};       // device words: live slots, error bits, slots idled by the tail kernel, tail mode (PathBuffers)
    // The end of the window (pt_tail_kernels.h): once at most tailSlots slots are alive the traversal kernel raises the tail-mode
    // word instead of tracing, the rest of the block of iterations is no-ops, and the tail kernel finishes every remaining path
    // and sample in one launch.  The image does not depend on who finishes a sample (the sensor adds in pass order).  On with
    // the automatic slot count (slrhip_config::stripes = 0) and on request (SLRHIP_FLAG_TAIL_KERNEL, SLRHIP_TAIL_SLOTS=n); a
    // caller who fixes the slot count gets the pure wavefront schedule unless he asks (the parity tests compare the two).  Never
    // for more than an eighth of the slots (the wavefront kernels are the efficient way to advance many paths) and not in the
    // counting build (its per-ray figures come from the wavefront kernels).  SLRHIP_TAIL_SLOTS=0 turns it off.
    static const long envTail = [] { const char* e = getenv("SLRHIP_TAIL_SLOTS"); return e ? atol(e) : -1L; }();
    {
        const bool asked = (ctx->config.flags & SLRHIP_FLAG_TAIL_KERNEL) != 0 || envTail > 0 || ctx->config.stripes == 0;
        const uint32_t bound = envTail > 0 ? (uint32_t)std::min<long>(envTail, 0x7FFFFFFFL) : kDefaultTailSlots;
        const bool off = !asked || envTail == 0 || !tailKernelAvailable(ctx->scene, rp.spectral != 0) || (ctx->config.flags & SLRHIP_FLAG_COUNT_TRAVERSAL) != 0;
        rp.tailSlots = off ? 0u : std::min(bound, active / SLR_TAIL_DIVISOR);
    }
    // The window is complete only if the queues handed out exactly one sample per pixel and pass: the device's own count
    // (k_count_samples over the queues' cursors) against the host's arithmetic.  A lost or repeated sample would leave a stale
    // or overwritten entry in the result window — never silent.
    const auto checkWindow = [&](hipStream_t s) -> int {
        uint32_t words[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        HIP_TRY(hipMemcpyAsync(words, ctx->activeSlots.ptr, kStatusWords * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        if (words[1]) return deviceError(words[1]);
        if (words[6] != rp.workItems)
            return fail(SLRHIP_ERR_HIP, "slrhip_render: the work queues handed out " + std::to_string(words[6]) + " samples for a window of " +
                                            std::to_string(rp.workItems) + " (internal error)");
        return SLRHIP_OK;
    };
    // tail mode seen in the status words: list the live slots, finish them, read the words again (live slots must be 0 then)
    const auto runTail = [&](hipStream_t s, bool timed) -> int {
        hipEvent_t t0 = nullptr, t1 = nullptr;
        if (timed) { HIP_TRY(hipEventCreate(&t0)); HIP_TRY(hipEventCreate(&t1)); HIP_TRY(hipEventRecord(t0, s)); }
        launchTail(ctx->scene, ctx->buffers, rp, status[0], ctx->numCUs, s);
        if (timed) HIP_TRY(hipEventRecord(t1, s));
        HIP_TRY(hipGetLastError());
        const uint32_t liveBefore = status[0];
        uint32_t words[6] = {0, 0, 0, 0, 0, 0};
        HIP_TRY(hipMemcpyAsync(words, ctx->activeSlots.ptr, 6 * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        std::memcpy(status, words, 4 * sizeof(uint32_t));
        if (timed) {
            float ms = 0.0f;
            HIP_TRY(hipEventElapsedTime(&ms, t0, t1));
            ctx->profMs[SLRHIP_KERNEL_TAIL] += ms;
            ++ctx->profLaunches[SLRHIP_KERNEL_TAIL];
            (void)hipEventDestroy(t0); (void)hipEventDestroy(t1);
        }
        if (status[1]) return deviceError(status[1]);
        if (words[2] != words[4])
            return fail(SLRHIP_ERR_HIP, "slrhip_render: the tail kernel left " + std::to_string(words[4] - words[2]) + " of " + std::to_string(words[4]) +
                                            " listed slots live (live count before: " + std::to_string(liveBefore) + ", cursor " + std::to_string(words[5]) + "; internal error)");
        // every listed slot ended idle and the wavefront kernels are off (tail mode): nothing is live any more
        HIP_TRY(hipMemsetAsync(ctx->activeSlots.ptr, 0, sizeof(uint32_t), s));
        status[0] = 0;
        return SLRHIP_OK;
    };
    // One wavefront iteration = k_shade (advance every live path by one vertex; finish and restart the paths that end) then
    // k_trace_ws (the extension and shadow rays that left).  Each check of the live-slot word costs one small copy + stream
    // sync; 16 iterations between checks keeps it < 1 %.
    const int kCheckEvery = 16;
    const int kEv = 3;    // events per iteration: before shade, after shade, after trace
    const bool timeKernels = (ctx->config.flags & SLRHIP_FLAG_TIME_KERNELS) != 0;
    const bool count = (ctx->config.flags & SLRHIP_FLAG_COUNT_TRAVERSAL) != 0;
    if (timeKernels && ctx->events.empty()) {
        ctx->events.resize((size_t)kCheckEvery * kEv);
        for (hipEvent_t& e : ctx->events) HIP_TRY(hipEventCreate(&e));
    }
    const uint64_t maxIterations = ((uint64_t)rp.workItems / rp.numSlots + 2) * 128 + 1024;   // paths are <= 100 vertices long
    uint64_t it = 0;

    // The block of kCheckEvery iterations is the same sequence of launches every time (the parity alternates inside it and is
    // back to 0 at its end), so it is captured ONCE per call into a hipGraph and replayed: one submission per block instead
    // of 32 launches with their dispatch gaps — what is left of the cost of the nearly empty iterations at the end of a
    // render.  Capture needs a real stream, so the work runs on the context's own stream, ordered after the caller's by an
    // event; render() returns only after that stream is idle, which orders the caller's later work after it.
    static const bool noGraph = [] { const char* e = getenv("SLRHIP_GRAPH"); return e && std::string(e) == "0"; }();
    if (!timeKernels && !noGraph && rp.numSlots >= (1u << 18)) {
        if (!ctx->workStream) {
            HIP_TRY(hipStreamCreateWithFlags(&ctx->workStream, hipStreamNonBlocking));
            HIP_TRY(hipEventCreateWithFlags(&ctx->userReady, hipEventDisableTiming));
        }
        hipStream_t ws = ctx->workStream;
        HIP_TRY(hipEventRecord(ctx->userReady, stream));          // the reset kernel above and whatever the caller queued before
        HIP_TRY(hipStreamWaitEvent(ws, ctx->userReady, 0));
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        HIP_TRY(hipStreamBeginCapture(ws, hipStreamCaptureModeThreadLocal));
        for (int k = 0; k < kCheckEvery; ++k) {
            launchShade(ctx->scene, ctx->buffers, rp, (uint32_t)(k & 1), ws);
            launchTraceWs(ctx->scene, ctx->buffers, rp, (uint32_t)(k & 1), traceBlocks, count, ws);
        }
        hipError_t ce = hipStreamEndCapture(ws, &graph);
        if (ce == hipSuccess) ce = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        if (ce != hipSuccess) {
            if (graph) (void)hipGraphDestroy(graph);
            return fail(SLRHIP_ERR_HIP, std::string("slrhip_render: hipGraph capture failed: ") + hipGetErrorString(ce));
        }
        int rc = SLRHIP_OK;
        while (active > 0 && rc == SLRHIP_OK) {
            hipError_t e = hipGraphLaunch(exec, ws);
            if (e == hipSuccess) e = hipMemcpyAsync(status, ctx->activeSlots.ptr, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, ws);
            if (e == hipSuccess) e = hipStreamSynchronize(ws);
            if (e != hipSuccess) rc = fail(SLRHIP_ERR_HIP, std::string("slrhip_render: ") + hipGetErrorString(e));
            it += kCheckEvery;
            if (rc == SLRHIP_OK && status[1]) rc = deviceError(status[1]);
            if (rc == SLRHIP_OK && status[3] && status[0]) rc = runTail(ws, false);
            active = status[0];
            if (rc == SLRHIP_OK && it > maxIterations) rc = fail(SLRHIP_ERR_HIP, "slrhip_render: iteration bound exceeded (internal error)");
        }
        (void)hipGraphExecDestroy(exec);
        (void)hipGraphDestroy(graph);
        if (rc != SLRHIP_OK) return rc;
        ctx->iterations += it;
        launchCountSamples(ctx->buffers, rp, ws);      // samples rendered in this window, counted on the device (T_SAMPLES)
        launchFold(ctx->buffers, rp, ws);              // sensor->add, in pass order
        HIP_TRY(hipGetLastError());
        return checkWindow(ws);
    }

    // SLRHIP_ITER_LOG=path (with SLRHIP_FLAG_TIME_KERNELS): per-iteration kernel times of this call, one line per iteration
    // "iteration shade_ms trace_ms live_slots_at_block_end" — how the drain of a render's last paths was measured
    static const char* iterLog = getenv("SLRHIP_ITER_LOG");
    std::vector<float> iterMs;
    std::vector<uint32_t> iterActive;
    uint32_t parity = 0;
    while (active > 0) {
        for (int k = 0; k < kCheckEvery; ++k) {
            hipEvent_t* ev = timeKernels ? &ctx->events[(size_t)k * kEv] : nullptr;
            if (ev) HIP_TRY(hipEventRecord(ev[0], stream));
            launchShade(ctx->scene, ctx->buffers, rp, parity, stream);
            if (ev) HIP_TRY(hipEventRecord(ev[1], stream));
            launchTraceWs(ctx->scene, ctx->buffers, rp, parity, traceBlocks, count, stream);
            if (ev) HIP_TRY(hipEventRecord(ev[2], stream));
            parity ^= 1;
            ++it;
        }
        HIP_TRY(hipGetLastError());                  // a failed launch surfaces here, not at the end of the render
        HIP_TRY(hipMemcpyAsync(status, ctx->activeSlots.ptr, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        active = status[0];
        if (status[1]) return deviceError(status[1]);
        if (timeKernels) {
            static const int cls[2] = {SLRHIP_KERNEL_SHADE, SLRHIP_KERNEL_TRACE};
            for (int k = 0; k < kCheckEvery; ++k) {
                hipEvent_t* ev = &ctx->events[(size_t)k * kEv];
                for (int j = 0; j < 2; ++j) {
                    float ms = 0.0f;
                    HIP_TRY(hipEventElapsedTime(&ms, ev[j], ev[j + 1]));
                    ctx->profMs[cls[j]] += ms;
                    ++ctx->profLaunches[cls[j]];
                    if (iterLog) iterMs.push_back(ms);
                }
                if (iterLog) iterActive.push_back(active);
            }
        }
        if (status[3] && status[0]) {
            const int rc = runTail(stream, timeKernels);
            if (rc != SLRHIP_OK) return rc;
            active = status[0];
        }
        if (it > maxIterations) return fail(SLRHIP_ERR_HIP, "slrhip_render: iteration bound exceeded (internal error)");
    }
    ctx->iterations += it;
    launchCountSamples(ctx->buffers, rp, stream);       // samples rendered in this window, counted on the device (T_SAMPLES)
    launchFold(ctx->buffers, rp, stream);               // sensor->add, in pass order
    HIP_TRY(hipGetLastError());
    {
        const int rc = checkWindow(stream);
        if (rc != SLRHIP_OK) return rc;
    }
    if (iterLog && timeKernels && !iterActive.empty()) {
        if (FILE* f = fopen(iterLog, "a")) {
            const size_t per = iterMs.size() / iterActive.size();
            fprintf(f, "# render: %u slots, %u passes from %u\n", rp.numSlots, sppCount, sppBegin);
            for (size_t i = 0; i < iterActive.size(); ++i) {
                fprintf(f, "%zu", i);
                for (size_t j = 0; j < per; ++j) fprintf(f, " %.4f", iterMs[i * per + j]);
                fprintf(f, " %u\n", iterActive[i]);
            }
            fclose(f);
        }
    }
    return SLRHIP_OK;
}

int slrhip_render(slrhip_ctx* ctx, uint32_t sppBegin, uint32_t sppCount, void* streamPtr) {
    if (!ctx) return fail(SLRHIP_ERR_INVALID_ARGUMENT, "slrhip_render: null context");
    if (!ctx->haveRender) return fail(SLRHIP_ERR_NO_SCENE, "slrhip_render: call slrhip_render_begin first");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t stream = (hipStream_t)streamPtr;
    RenderParams& rp = ctx->params;
    if (rp.numSlots == 0) { ctx->firstRenderCall = false; return SLRHIP_OK; }
    // The result window (PathBuffers::results) holds one entry per pixel and pass: 16 B (RGB) / 64 B (spectral).  A call of more
    // passes than fit the budget — 16 GiB by default, SLRHIP_RESULT_WINDOW_MB overrides — or than 2^32 samples is rendered as
    // several windows, one after the other; the sensor adds in pass order either way, so the image does not depend on the split.
    const uint64_t entryBytes = (rp.spectral ? 4u : 1u) * sizeof(float4);
    uint64_t budget = 16ull << 30;
    if (const char* e = getenv("SLRHIP_RESULT_WINDOW_MB")) { const long mb = atol(e); if (mb > 0) budget = (uint64_t)mb << 20; }
    const uint64_t maxPasses = std::max<uint64_t>(1, std::min<uint64_t>(budget / ((uint64_t)rp.numPixels * entryBytes), 0xF0000000ull / rp.numPixels));      // (run ids + one round of waves stay inside 32 bits)
    // whole runs (RenderParams::runLength passes of a pixel in a row, pt_kernels.h) wherever the call is long enough: a window of
    // an odd number of passes would fall back to runs of one pass and lose the coherence of a wave's slots
    uint32_t window = (uint32_t)std::min<uint64_t>(maxPasses, std::max<uint32_t>(sppCount, 1u));
    if (window >= kDefaultRunLength) window -= window % kDefaultRunLength;
    HIP_TRY(ctx->results.alloc((size_t)window * rp.numPixels * (rp.spectral ? 4u : 1u)));
    ctx->buffers.results = ctx->results.ptr;
    if (sppCount == 0) return renderWindow(ctx, sppBegin, 0, stream);
    for (uint32_t done = 0; done < sppCount; done += window) {
        const int rc = renderWindow(ctx, sppBegin + done, std::min(window, sppCount - done), stream);
        if (rc != SLRHIP_OK) return rc;
    }
    return SLRHIP_OK;
}

int slrhip_resolve_framebuffer(slrhip_ctx* ctx, float* deviceDst, size_t numFloats, void* streamPtr) {
    if (!ctx || !deviceDst) return fail(SLRHIP_ERR_INVALID_ARGUMENT, "slrhip_resolve_framebuffer: null argument");
    if (!ctx->haveRender) return fail(SLRHIP_ERR_NO_SCENE, "slrhip_resolve_framebuffer: nothing rendered");
    const RenderParams& rp = ctx->params;
    const size_t need = (size_t)rp.imageWidth * rp.imageHeight * (rp.spectral ? 16 : 3);
    if (numFloats < need) return fail(SLRHIP_ERR_INVALID_ARGUMENT, "slrhip_resolve_framebuffer: destination too small");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t stream = (hipStream_t)streamPtr;
    HIP_TRY(hipMemsetAsync(deviceDst, 0, need * sizeof(float), stream));
    if (rp.numPixels) launchResolve(ctx->buffers, rp, deviceDst, stream);
    HIP_TRY(hipGetLastError());
    return SLRHIP_OK;
}

// The one exchange step of the multi-GPU path: every rank resolves its shard (zeros outside its tiles) and the
// frames are summed onto `root` — disjoint supports, so the sum is a gather.  RCCL is loaded on first use (dlopen), so a
// single-GPU host does not need librccl at all; the communicator is the caller's (one process per GPU, ncclCommInitRank).
int slrhip_reduce_framebuffer(slrhip_ctx* ctx, void* ncclComm, int root, float* deviceDst, size_t numFloats, void* streamPtr) {
    if (!ctx || !ncclComm) return fail(SLRHIP_ERR_INVALID_ARGUMENT, "slrhip_reduce_framebuffer: null argument");
    if (!ctx->haveRender) return fail(SLRHIP_ERR_NO_SCENE, "slrhip_reduce_framebuffer: nothing rendered");
    const RenderParams& rp = ctx->params;
    const size_t need = (size_t)rp.imageWidth * rp.imageHeight * (rp.spectral ? 16 : 3);
    if (deviceDst && numFloats < need) return fail(SLRHIP_ERR_INVALID_ARGUMENT, "slrhip_reduce_framebuffer: destination too small");
    typedef int (*reduce_fn)(const void*, void*, size_t, int, int, int, void*, hipStream_t);
    typedef int (*rank_fn)(void*, int*);
    static void* rccl = [] {
        void* h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        return h ? h : dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    }();
    static reduce_fn ncclReduceFn = rccl ? reinterpret_cast<reduce_fn>(dlsym(rccl, "ncclReduce")) : nullptr;
    static rank_fn ncclCommUserRankFn = rccl ? reinterpret_cast<rank_fn>(dlsym(rccl, "ncclCommUserRank")) : nullptr;
    if (!ncclReduceFn || !ncclCommUserRankFn) return fail(SLRHIP_ERR_UNSUPPORTED, "slrhip_reduce_framebuffer: librccl.so (ncclReduce) not found");
    // only the root receives: it must hand over a destination (RCCL would fault on a null recvbuff, not return an error)
    int myRank = -1;
    if (ncclCommUserRankFn(ncclComm, &myRank) != 0) return fail(SLRHIP_ERR_HIP, "slrhip_reduce_framebuffer: ncclCommUserRank failed");
    if (myRank == root && !deviceDst) return fail(SLRHIP_ERR_INVALID_ARGUMENT, "slrhip_reduce_framebuffer: the root rank needs a destination buffer");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(ctx->resolveScratch.alloc(need));
    int rc = slrhip_resolve_framebuffer(ctx, ctx->resolveScratch.ptr, need, streamPtr);
    if (rc != SLRHIP_OK) return rc;
    const int kNcclFloat32 = 7, kNcclSum = 0;      // rccl.h: ncclDataType_t / ncclRedOp_t
    const int nrc = ncclReduceFn(ctx->resolveScratch.ptr, deviceDst, need, kNcclFloat32, kNcclSum, root, ncclComm, (hipStream_t)streamPtr);
    if (nrc != 0) return fail(SLRHIP_ERR_HIP, "slrhip_reduce_framebuffer: ncclReduce failed (" + std::to_string(nrc) + ")");
    return SLRHIP_OK;
}

int slrhip_read_framebuffer(slrhip_ctx* ctx, float* hostDst, size_t numFloats) {
    if (!ctx || !hostDst) return fail(SLRHIP_ERR_INVALID_ARGUMENT, "slrhip_read_framebuffer: null argument");
    if (!ctx->haveRender) return fail(SLRHIP_ERR_NO_SCENE, "slrhip_read_framebuffer: nothing rendered");
    const RenderParams& rp = ctx->params;
    const size_t need = (size_t)rp.imageWidth * rp.imageHeight * (rp.spectral ? 16 : 3);
    if (numFloats < need) return fail(SLRHIP_ERR_INVALID_ARGUMENT, "slrhip_read_framebuffer: destination too small");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(ctx->resolveScratch.alloc(need));
    int rc = slrhip_resolve_framebuffer(ctx, ctx->resolveScratch.ptr, need, nullptr);
    if (rc != SLRHIP_OK) return rc;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(hostDst, ctx->resolveScratch.ptr, need * sizeof(float), hipMemcpyDeviceToHost));
    return SLRHIP_OK;
}

int slrhip_synchronize(slrhip_ctx* ctx) {
    if (!ctx) return fail(SLRHIP_ERR_INVALID_ARGUMENT, "slrhip_synchronize: null context");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipDeviceSynchronize());
    return SLRHIP_OK;
}

int slrhip_get_counters(slrhip_ctx* ctx, slrhip_counters* out) {
    if (!ctx || !out) return fail(SLRHIP_ERR_INVALID_ARGUMENT, "slrhip_get_counters: null argument");
    std::memset(out, 0, sizeof(*out));
    out->bvh_nodes = ctx->nodes.count;
    out->bvh_depth = ctx->bvhDepth;
    out->bvh_leaf_references = ctx->bvhLeafRefs;
    out->build_seconds = ctx->buildSeconds;
    out->iterations = ctx->iterations;
    if (ctx->haveRender) {
        uint64_t t[T_KINDS];
        int rc = readTotals(ctx, t);
        if (rc != SLRHIP_OK) return rc;
        out->extension_rays = t[T_EXT_RAYS];
        out->shadow_rays = t[T_SHADOW_RAYS];
        out->samples = t[T_SAMPLES];          // counted on the device from the slots' sample headers (k_count_samples)
    }
    return SLRHIP_OK;
}

int slrhip_get_profile(slrhip_ctx* ctx, slrhip_profile* out) {
    if (!ctx || !out) return fail(SLRHIP_ERR_INVALID_ARGUMENT, "slrhip_get_profile: null argument");
    std::memset(out, 0, sizeof(*out));
    for (int k = 0; k < SLRHIP_KERNEL_COUNT; ++k) { out->launches[k] = ctx->profLaunches[k]; out->milliseconds[k] = ctx->profMs[k]; }
    if (ctx->haveRender) {
        uint64_t t[T_KINDS];
        int rc = readTotals(ctx, t);
        if (rc != SLRHIP_OK) return rc;
        out->rays[0] = t[T_EXT_RAYS]; out->rays[1] = t[T_SHADOW_RAYS];
        out->nodes[0] = t[T_NODES_CLOSEST]; out->triangles[0] = t[T_TRIS_CLOSEST];
        out->nodes[1] = t[T_NODES_SHADOW]; out->triangles[1] = t[T_TRIS_SHADOW];
        out->slot_visits = t[T_SLOT_VISITS];
    }
    return SLRHIP_OK;
}

// Diagnostic entry point: closest-hit queries against the uploaded scene (host arrays in and out).
// rays: n x {org[3], dir[3], dist_min, dist_max}; hits: n x {triangle, dist, b0, b1}.
int slrhip_trace_rays(slrhip_ctx* ctx, const float* rays, uint32_t n, float* hits) {
    if (!ctx || !rays || !hits) return fail(SLRHIP_ERR_INVALID_ARGUMENT, "slrhip_trace_rays: null argument");
    if (!ctx->haveScene) return fail(SLRHIP_ERR_NO_SCENE, "slrhip_trace_rays: no scene uploaded");
    if (n == 0) return SLRHIP_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    std::vector<float4> org(n), dir(n);
    for (uint32_t i = 0; i < n; ++i) {
        const float* r = rays + (size_t)i * 8;
        org[i] = make_float4(r[0], r[1], r[2], r[6]);
        dir[i] = make_float4(r[3], r[4], r[5], r[7]);
    }
    DevArray<float4> dOrg, dDir, dOut;
    HIP_TRY(dOrg.upload(org));
    HIP_TRY(dDir.upload(dir));
    HIP_TRY(dOut.alloc(n));
    launchTraceBatch(ctx->scene, dOrg.ptr, dDir.ptr, dOut.ptr, n, nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(hits, dOut.ptr, (size_t)n * sizeof(float4), hipMemcpyDeviceToHost));
    hitsToUV(hits, n);
    return SLRHIP_OK;
}

// Diagnostic (include/slrhip_debug.h): function-level BSDF queries through the device functions the shade kernel calls.
int slrhip_debug_work_distribution(uint32_t numPixels, uint32_t numSlots, uint32_t numPasses, uint32_t runLength, uint32_t* counts,
                                   uint32_t* queueLengths) {
    if (!counts || !queueLengths || numPixels == 0 || numSlots < 64 || numSlots % 64 || runLength == 0 || numPasses % runLength ||
        (uint64_t)numPixels * numPasses > 0xFFFFFFFFull)
        return fail(SLRHIP_ERR_INVALID_ARGUMENT, "slrhip_debug_work_distribution: bad arguments");
    RenderParams rp;
    std::memset(&rp, 0, sizeof(rp));
    rp.numSlots = numSlots; rp.numWaves = numSlots / 64u; rp.numPixels = numPixels; rp.sppCount = numPasses;
    rp.workItems = numPixels * numPasses; rp.runLength = runLength; rp.numRuns = numPixels * (numPasses / runLength);
    for (uint32_t w = 0; w < rp.numWaves; ++w) {
        uint32_t taken = 0;
        for (;; ++taken) {
            const WorkItem it = workItemOf(rp, w, taken);
            if (!it.valid) break;
            ++counts[(size_t)it.pass * numPixels + it.pix];
        }
        queueLengths[w] = taken;
        if (workSamplesTaken(rp, w, taken + 7u) != taken) return fail(SLRHIP_ERR_HIP, "slrhip_debug_work_distribution: workSamplesTaken disagrees with the queue");
    }
    return SLRHIP_OK;
}

int slrhip_bsdf_queries(slrhip_ctx* ctx, uint32_t material, uint32_t n, const float* queries, float wl_offset, float u_lambda, float* out) {
    if (!ctx || !queries || !out) return fail(SLRHIP_ERR_INVALID_ARGUMENT, "slrhip_bsdf_queries: null argument");
    if (!ctx->haveScene) return fail(SLRHIP_ERR_NO_SCENE, "slrhip_bsdf_queries: no scene uploaded");
    if (material >= ctx->scene.numMaterials) return fail(SLRHIP_ERR_INVALID_ARGUMENT, "slrhip_bsdf_queries: material index out of range");
    if (!(wl_offset >= 0.0f && wl_offset < 1.0f) || !(u_lambda >= 0.0f && u_lambda < 1.0f))
        return fail(SLRHIP_ERR_INVALID_ARGUMENT, "slrhip_bsdf_queries: wl_offset and u_lambda must be in [0, 1)");
    if (n == 0) return SLRHIP_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    const bool spectral = ctx->config.mode == SLRHIP_MODE_SPECTRAL;
    const uint32_t C = spectral ? 16u : 3u, planes = spectral ? 4u : 1u;
    // WavelengthSamples::createWithEqualOffsets (SpectrumTypes.h:60, RGBTypes.h:41)
    const uint32_t wl = std::min<uint32_t>((uint16_t)(C * u_lambda), C - 1);
    std::vector<float> in(queries, queries + (size_t)n * 12);
    DevArray<float> dIn;
    DevArray<float4> dGeo, dMisc, dFsS, dFsE;
    HIP_TRY(dIn.upload(in));
    HIP_TRY(dGeo.alloc(n));
    HIP_TRY(dMisc.alloc(n));
    HIP_TRY(dFsS.alloc((size_t)planes * n));
    HIP_TRY(dFsE.alloc((size_t)planes * n));
    launchBsdfQueries(ctx->scene, spectral, material, n, dIn.ptr, wl_offset, wl, dGeo.ptr, dMisc.ptr, dFsS.ptr, dFsE.ptr, nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    std::vector<float4> geo(n), misc(n), fsS((size_t)planes * n), fsE((size_t)planes * n);
    HIP_TRY(hipMemcpy(geo.data(), dGeo.ptr, geo.size() * sizeof(float4), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(misc.data(), dMisc.ptr, misc.size() * sizeof(float4), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(fsS.data(), dFsS.ptr, fsS.size() * sizeof(float4), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(fsE.data(), dFsE.ptr, fsE.size() * sizeof(float4), hipMemcpyDeviceToHost));
    const uint32_t stride = 6 + 2 * C;
    for (uint32_t i = 0; i < n; ++i) {
        float* o = out + (size_t)stride * i;
        o[0] = geo[i].x; o[1] = geo[i].y; o[2] = geo[i].z; o[3] = geo[i].w; o[4] = misc[i].x;
        for (uint32_t k = 0; k < C; ++k) {
            const float* a = reinterpret_cast<const float*>(&fsS[(size_t)(k / 4) * n + i]);
            const float* b = reinterpret_cast<const float*>(&fsE[(size_t)(k / 4) * n + i]);
            o[5 + k] = a[k % 4];
            o[5 + C + k] = b[k % 4];
        }
        o[5 + 2 * C] = misc[i].y;
    }
    return SLRHIP_OK;
}

} // extern "C"
