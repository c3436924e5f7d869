// pt_kernels.h — launch interface between the C-ABI layer (slrhip_api.hip) and the kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_types.h"

namespace slrhip {

// Limits of the shading tables staged in LDS by k_shade / k_tail (ShadeLds, pt_shade_kernels.h); larger scenes read them from HBM.
static const int kLdsMaterials = 32;
static const int kLdsLights = 16;
static const int kLdsSpectra = 96;
static const int kLdsPoolFloats = 6144;     // 24 KiB: the spectrum sample tables of a scene (spectral mode), staged per workgroup

// Scene arrays resident in HBM (uploaded once by slrhip_upload_scene).
struct DevScene {
    const float4* nodes;          // QNode as 8 x float4, breadth-first
    const float4* leafTris;       // LeafTri as 3 x float4
    const ShadeTri* shadeTris;    // indexed by scene triangle index
    const LightTri* lightTris;    // indexed by light index
    const DevMaterial* materials;  // RGB mode
    const DevMaterialS* materialsS; // spectral mode: spectrum indices instead of constants
    const DevSpectrum* spectra;
    const float* spectrumPool;
    const float* lightPMF;        // RegularConstantDiscrete1D of the aggregate's light list
    const float* lightCDF;        // numLights + 1 entries
    uint32_t numNodes;
    const float4* nodesQ;         // QNodeQ array (4 x float4 per node) or nullptr: large scenes traverse this one
    const float4* nodes8;         // QNode8 array (8 x float4 per node) or nullptr: the eight-wide quantized tree (k_trace_ws only)
    uint32_t numMaterials;
    uint32_t numLights;
    uint32_t lightPow2;           // prevPowerOf2(numLights)
    uint32_t hasMicrofacet;       // any SLRHIP_MATERIAL_MICROFACET_* in the scene
    uint32_t hasMulti;            // any SLRHIP_MATERIAL_MULTI: k_shade with the MultiBSDF code
    // textures: null / 0 when the scene has none
    const DevTexture* textures;
    const DevMatTex* matTex;      // per material
    const float4* triUV;          // per scene triangle: (u0, v0, u1, v1), (u2, v2, -, -)
    const float4* alphaTris;      // per alpha record (LeafTri::alpha): the same two float4 + the alpha texture index
    const float* texTexels;       // texels of the image textures, 3 floats each: (r, g, b), spectral mode (u, v, s)
    // The tables k_shade / k_tail stage in LDS, packed by slrhip_upload_scene into ONE array in the order of ShadeLds'
    // segments — RGB: materials, lights, light PMF, light CDF; spectral: materials, spectra, sample pool, lights, PMF, CDF — each
    // padded to whole float4; tableEnd[k] = end of segment k in float4 (cumulative).  nullptr when the scene exceeds the LDS limits.
    const float4* shadeTables;
    uint32_t tableEnd[6];
    const float4* instances;      // DevInstance as 9 x float4, or nullptr: the scene has no instanced mesh
    uint32_t numInstances;
    uint32_t numTextures;
    uint32_t numSpectra;
    uint32_t numSpectrumData;     // floats in spectrumPool (padded to a multiple of 4 on upload)
    // environment sphere (InfiniteSphereSurfaceObject, SurfaceObject.cpp:137-222); RGB mode
    uint32_t hasEnv;
    uint32_t envWidth, envHeight, envMapWidth, envMapHeight;
    float envScale;
    float aggImportance;          // SurfaceObjectAggregate::importance() = integral of the light list distribution
    const float* envTexels;       // [height][width][3]
    const float* envTopPDF;       // RegularConstantContinuous2D: top distribution over rows (mapHeight, +1 for the CDF)
    const float* envTopCDF;
    const float* envRowPDF;       // [mapHeight][mapWidth]
    const float* envRowCDF;       // [mapHeight][mapWidth + 1]
    // spectral mode: the environment texels are (u, v, s); the Meng-15 tables to look them up at run time (slrhip_upsampling_tables)
    uint32_t gridWidth, gridHeight;
    const uint8_t* gridCells;     // 8 bytes per cell: inside, num_points, idx[6]
    const float* pointUV;         // 2 per data point
    const float* pointSpectrum;   // 95 per data point
    DevCamera camera;
};

// Queue counters.  A counter word that every wave bumps saturates near 88 atomics/us on this chip
// (MI355X_MICROARCH.md, rows "dequeue" / "fanin"), which made the first version of the shade kernel
// atomic-bound (14 400 waves -> 164 us).  So: (1) there is NO extension-ray queue — nearly every live
// slot has one, the traversal kernel walks all slots and reads the state flag; (2) there is no queue of
// finished paths either — the shade kernel restarts them itself (k_shade); (3) the shadow-ray queue is
// split into kShards regions, a workgroup appends to region (blockIdx % kShards) with ONE atomic per
// workgroup, and every counter sits on its own 128-byte line.
// One set of counters per iteration parity: k_shade(parity) FILLS set `parity`, k_trace_ws(parity) READS it
// and clears set `parity ^ 1` for the next k_shade.
static const uint32_t kShards = 16;
static const uint32_t kCounterStride = 32;                 // words: one 128-byte line per counter
enum { Q_SHADOW = 0, Q_KINDS = 1 };
static const uint32_t kQueueSetWords = Q_KINDS * kShards * kCounterStride;
__host__ __device__ inline uint32_t queueCounterIndex(uint32_t parity, uint32_t kind, uint32_t shard) {
    return parity * kQueueSetWords + (kind * kShards + shard) * kCounterStride;
}
// statistics, sharded the same way: [kind][shard] 64-bit words on separate lines
enum { T_EXT_RAYS = 0, T_SHADOW_RAYS = 1, T_NODES_CLOSEST = 2, T_TRIS_CLOSEST = 3, T_NODES_SHADOW = 4, T_TRIS_SHADOW = 5,
       T_SLOT_VISITS = 6,
       // schedule diagnostics of the wave-specialised closest-hit kernel (counting builds only; SLRHIP_DEBUG_WS prints them)
       T_WS_STEPS = 7, T_WS_IDLE_SPINS = 8, T_WS_CYCLES = 9, T_WS_IDLE_CYCLES = 10, T_WS_REFILLS = 11, T_WS_PRODUCER_WAITS = 12,
       T_WS_NODE_BLOCKS = 13, T_WS_TRI_BLOCKS = 14, T_WS_ACTIVE_LANES = 15,
       T_SAMPLES = 16,          // finished samples accumulated into pixels, summed from the slots' headers at the end of a render call
       T_KINDS = 17 };
// device error word bits (PathBuffers::errorWord): set by any kernel path that gives up or drops work
enum : uint32_t { ERR_RING_SPACE = 1u, ERR_RING_RELEASE = 2u, ERR_CONSUMER_IDLE = 4u, ERR_STACK_OVERFLOW = 8u, ERR_QUEUE_OVERFLOW = 16u };
static const uint32_t kTotalStride = 16;                   // 64-bit words: one 128-byte line
__host__ __device__ inline uint32_t totalIndex(uint32_t kind, uint32_t shard) { return (kind * kShards + shard) * kTotalStride; }

// Path state, SoA, one record per slot.  A slot is a path in flight, NOT a place in the image: whenever its path ends it takes the
// next sample — a (pixel, pass) pair — from the work queue of its wave of the shade kernel (WorkItem below) and writes the finished
// sample's contribution to the result window; the sensor's per-pixel Kahan sum (ImageSensor::add, in pass order) is done by
// k_fold once the window's passes are complete.  The image therefore does not depend on the number of slots, on which slot
// rendered which sample or on the shard size, and the slots stay busy until the LAST sample of the call has been handed out.
struct PathBuffers {
    uint4* rng;                   // xorshift128 state
    float4* rayOrg;               // extension / shadow ray origin, w = distMin
    float4* rayDir;               // extension ray direction, w = distMax
    float4* hit;                  // x = triangle (bits), y = t, z = b1, w = b2 (Moller-Trumbore's barycentrics)
    int32_t* hitInstance;         // instanced scenes only (else nullptr): the instance the hit went through, -1 = a loose triangle
    // spectrum-valued records: RGB = one float4 per slot (scalar in .w); spectral = 4 planes of numSlots float4
    float4* alpha;                // path throughput (+ pdf of the sampled direction)
    float4* spR;                  // path radiance Kahan sum (sp) (+ camera weight)
    float4* spC;                  // its compensation
    // Result window of the render call: one entry per (pass of the window, pixel of the shard), pass-major — RGB one float4,
    // spectral four (the sixteen bins of SpectrumStorage) — written once by the slot whose path was that sample; then the
    // sensor itself: per pixel the Kahan sum and its compensation (CompensatedSum, BasicTypes/CompensatedSum.h:24-30), kept
    // across the render calls of one slrhip_render_begin.
    float4* results;
    float4* fbSum;
    float4* fbComp;
    float4* nee;                  // pending next-event contribution
    float4* shadowDir;            // shadow ray direction, w = distMax
    float* pdfPrev;               // spectral mode only: the scalar that rides in alpha.w in RGB mode
    // per-slot sample header, written when a sample starts: x = the sample's pixel (index into the shard's pixel list), y = its
    // camera weight (bits), z = its wavelength offset (bits; lambda_i = 360 + 470 (i + offset) / 16, spectral mode), w = its pass
    // relative to the first pass of the window
    uint4* hdr;
    uint32_t* flags;
    // Work queues (WorkItem below): per wave of the shade kernel (64 slots) the number of samples it has taken from its queue so
    // far.  Read and advanced by that wave alone in k_shade (no atomics); by atomics in the tail kernel, where the slots of a
    // wave may be held by lanes of several waves.
    uint32_t* cursor;
    uint32_t* visible;            // result of the shadow ray
    uint32_t* shadowQueue;        // kShards regions of shardCapacity slot indices: shadow rays of this iteration
    uint32_t* tailList;           // tail mode: the live slots, listed by k_tail_collect (numSlots entries)
    uint32_t* queueCount;         // [parity][kind][shard], see queueCounterIndex
    // Live slots.  A slot that finds its queue exhausted goes idle; k_shade counts them per workgroup and adds the count to one of
    // kShards words on separate lines (idleShards: the queues run out at about the same iteration everywhere, and ~10^5 atomics
    // per launch on ONE word would cost a millisecond each time).  Every workgroup of the traversal launch that follows sums the
    // shards (liveSlots below) for the tail-mode decision; its first workgroup publishes the sum in activeSlots[0] for the next
    // k_shade launch (which only asks "is anything left?") and for the host.
    uint32_t* activeSlots;
    uint32_t* idleShards;         // [kShards] x kCounterStride words
    // one word per block of 256 consecutive slots: set by k_shade once every slot of the block has run out of passes (a slot
    // never leaves ST_IDLE within a render call), cleared by k_reset_slots; the scanning kernels skip such blocks without
    // touching their state — the last iterations of a render, in which a few long paths are left, then cost launch overhead
    uint32_t* blockDead;
    uint32_t* errorWord;          // ERR_* bits, sticky until the next slrhip_render call; read back with activeSlots
    // The end of a render call (pt_tail_kernels.h): as soon as at most RenderParams::tailSlots slots are still live (a slot
    // goes idle only when its pixel has no pass left, so by then nearly every pixel is done), the traversal kernel of that
    // iteration raises tailMode (= 1 + the iteration's parity) instead of tracing: every later wavefront launch of the call is
    // a no-op and the host runs the tail kernels, which take each remaining path — and the passes the remaining pixels still
    // have to hand out — to the end in one launch.  tailWords = {list length, list cursor}.
    uint32_t* tailMode;
    uint32_t* tailWords;
    uint32_t* tailIdled;          // slots the tail kernel left idle: the host checks it against the list length (every listed slot must end idle)
    uint32_t* windowSamples;      // samples the queues handed out in this window (k_count_samples): the host checks it against pixels x passes
    uint64_t* totals;             // [kind][shard], see totalIndex
    const uint32_t* pixelXY;      // pixel-of-shard -> x | y << 16
    // Records that the same kernels touch at the same slots share a 32-byte sector when these strides are 2 (rayDir = rayOrg + 1,
    // spC = spR + 1, element index = slot x stride): a kernel that visits a fraction of the slots then moves one sector per
    // visited slot and pair instead of most sectors of two arrays.  1 = separate arrays.
    uint32_t rayStride, spStride, hdrStride;      // hdrStride 2: rng = hdr + 1 (sample header + RNG state of a slot in one sector)
};

struct RenderParams {
    uint32_t numSlots;            // 256 x shade workgroups
    uint32_t numBlocks;           // numSlots / 256
    uint32_t numPixels, stripes;  // stripes: slots per pixel the slot count was sized for (slrhip_config::stripes or the automatic choice)
    uint32_t sppBegin, sppCount;  // the passes of the current window
    uint32_t workItems;           // numPixels x sppCount (< 2^32: slrhip_render sizes the windows)
    uint32_t numWaves;            // numSlots / 64: the owners of the work queues
    uint32_t runLength;           // passes per run (divides sppCount)
    uint32_t numRuns;             // runs of the window = numPixels x sppCount / runLength
    int32_t rngSeed;
    float timeStart, timeEnd;
    uint32_t imageWidth, imageHeight;
    uint32_t countSlots;          // SLRHIP_FLAG_COUNT_TRAVERSAL: also count live slots per logic launch
    uint32_t shardCapacity;       // entries per queue region = ceil(numBlocks / kShards) * 256
    uint32_t spectral;            // 0 = RGB (3 components), 1 = 16 wavelength samples
    uint32_t injectError;         // SLRHIP_FLAG_TEST_DEVICE_ERROR: the reset kernel raises the device error word
    uint32_t tailSlots;           // enter tail mode once at most this many slots are live; 0 = never
};

// Work distribution.  An item is one sample, a (pixel, pass) pair.  The unit that is dealt out is a RUN: runLength consecutive
// passes of ONE pixel.  Run r = (pass group, pixel) with the pixel fastest; queues belong to WAVES of the shade kernel (64
// consecutive slots): the j-th run of wave w is run j x numWaves + (w + j x kWorkRotate) mod numWaves — every run belongs to
// exactly one wave, and the rotation walks each wave over the image, so that its queue is a sample of the whole frame and the
// queues carry the same work to within a few per cent.  The lanes of a wave that need a sample take the next items of the
// wave's queue in lane order, so the 64 slots of a wave hold passes of one pixel (two while it moves on): their camera rays are
// all but identical and their first bounces start in the same place, which is what keeps the lanes of a traversal wave in step
// (measured: with the slots of a wave spread over an 8 x 8 tile the traversal launch took 2.1 ms instead of 1.76).  Static
// queues need no global counter: one word that every wave bumps saturates near 88 atomics per microsecond (above).
static const uint32_t kWorkRotate = 40503u;
struct WorkItem {
    uint32_t pix, pass;           // pass relative to RenderParams::sppBegin
    bool valid;                   // false: the queue is exhausted (every later item of this queue is invalid too)
};
__host__ __device__ inline uint32_t workRun(const RenderParams& rp, uint32_t wave, uint32_t j) {
    return j * rp.numWaves + (uint32_t)(((uint64_t)wave + (uint64_t)j * kWorkRotate) % rp.numWaves);
}
__host__ __device__ inline WorkItem workItemOf(const RenderParams& rp, uint32_t wave, uint32_t taken) {
    const uint32_t j = taken / rp.runLength, r = taken - j * rp.runLength;
    WorkItem w;
    w.pix = 0; w.pass = 0; w.valid = false;
    if (j > rp.numRuns / rp.numWaves) return w;                                      // (also keeps j x numWaves inside 32 bits)
    const uint32_t run = workRun(rp, wave, j);
    if (run >= rp.numRuns) return w;
    const uint32_t group = run / rp.numPixels;
    w.pix = run - group * rp.numPixels;
    w.pass = group * rp.runLength + r;
    w.valid = true;
    return w;
}
// Samples among the first `taken` items of a wave's queue (what k_count_samples adds up: a cursor runs past the end of its
// queue by the lanes that found nothing).
__host__ __device__ inline uint32_t workSamplesTaken(const RenderParams& rp, uint32_t wave, uint32_t taken) {
    uint32_t runs = 0;
    for (uint32_t j = 0; j <= rp.numRuns / rp.numWaves; ++j) if (workRun(rp, wave, j) < rp.numRuns) ++runs;
    const uint32_t length = runs * rp.runLength;
    return taken < length ? taken : length;
}

// Evaluated by every workgroup of the traversal launch of an iteration: its input is stable during that launch (activeSlots
// is written by k_shade only), so all workgroups agree; the first one records the decision for the kernels that follow.
__device__ __forceinline__ uint32_t liveSlots(const PathBuffers& pb, uint32_t numSlots) {
    uint32_t idle = 0;
#pragma unroll
    for (uint32_t k = 0; k < kShards; ++k) idle += pb.idleShards[k * kCounterStride];
    return numSlots - idle;
}
__device__ __forceinline__ bool tailModeBegins(const PathBuffers& pb, uint32_t live, uint32_t tailSlots, uint32_t parity) {
    if (tailSlots == 0u) return false;
    if (pb.tailMode[0]) return true;
    if (live > tailSlots) return false;
    if (blockIdx.x == 0 && threadIdx.x == 0) pb.tailMode[0] = 1u + parity;
    return true;
}

void launchResetSlots(const PathBuffers& pb, const RenderParams& rp, bool clearSensor, hipStream_t stream);
// ImageSensor::add for the window's passes, in pass order: the result window folded into the per-pixel Kahan sums
void launchFold(const PathBuffers& pb, const RenderParams& rp, hipStream_t stream);
// one wavefront iteration = launchShade(parity) then launchTraceWs(parity)
void launchShade(const DevScene& sc, const PathBuffers& pb, const RenderParams& rp, uint32_t parity, hipStream_t stream);
void launchCountSamples(const PathBuffers& pb, const RenderParams& rp, hipStream_t stream);
// the rest of a render call in one launch (after tailMode was raised): list the live slots, then one lane per path to its end
void launchTail(const DevScene& sc, const PathBuffers& pb, const RenderParams& rp, uint32_t liveSlots, int numCUs, hipStream_t stream);
bool tailKernelAvailable(const DevScene& sc, bool spectral);      // not built for spectral scenes with MultiBSDF materials or textures
void launchResolve(const PathBuffers& pb, const RenderParams& rp, float* dst, hipStream_t stream);
void launchBsdfQueries(const DevScene& sc, bool spectral, uint32_t material, uint32_t n, const float* in, float wlOffset, uint32_t wl,
                       float4* geo, float4* misc, float4* fsSample, float4* fsEval, hipStream_t stream);
// closest-hit ray queries (slrhip_trace_rays): one lane per ray, 64-ray batches (pt_trace.hip)
void launchTraceBatch(const DevScene& sc, const float4* org, const float4* dir, float4* out, uint32_t n, hipStream_t stream);
// wave-specialised traversal (pt_trace_ws.hip): ONE launch for the extension and the shadow rays of an iteration
void launchTraceWs(const DevScene& sc, const PathBuffers& pb, const RenderParams& rp, uint32_t parity, uint32_t blocks, bool count,
                   hipStream_t stream);
int traceWsBlocksPerCU(bool quantizedTree);

} // namespace slrhip
