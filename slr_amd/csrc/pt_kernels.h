// pt_kernels.h — launch interface between the C-ABI layer (slrhip_api.hip) and the kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_types.h"

namespace slrhip {

// Scene arrays resident in HBM (uploaded once by slrhip_upload_scene).
struct DevScene {
    const float4* nodes;          // QNode as 8 x float4
    const float4* leafTris;       // LeafTri as 3 x float4
    const ShadeTri* shadeTris;    // indexed by scene triangle index
    const LightTri* lightTris;    // indexed by light index
    const DevMaterial* materials;
    const float* lightPMF;        // RegularConstantDiscrete1D of the aggregate's light list
    const float* lightCDF;        // numLights + 1 entries
    uint32_t numLights;
    uint32_t lightPow2;           // prevPowerOf2(numLights)
    DevCamera camera;
};

// Path state, SoA, one record per slot (slot = stripe * numPixels + pixel-of-shard).
struct PathBuffers {
    uint4* rng;                   // xorshift128 state
    float4* rayOrg;               // extension / shadow ray origin, w = distMin
    float4* rayDir;               // extension ray direction, w = distMax
    float4* hit;                  // x = triangle (bits), y = t, z = b0, w = b1
    float4* alpha;                // path throughput, w = pdf of the sampled direction
    float4* spR;                  // path radiance Kahan sum (sp), w = camera weight
    float4* spC;                  // its compensation
    float4* accR;                 // pixel accumulator Kahan sum (the ImageSensor pixel)
    float4* accC;
    float4* nee;                  // pending next-event contribution
    float4* shadowDir;            // shadow ray direction, w = distMax
    uint32_t* flags;
    uint32_t* sampleIdx;
    uint32_t* visible;            // result of the shadow ray
    uint32_t* extQueue;           // slot indices with an extension ray this iteration
    uint32_t* shadowQueue;        // slot indices with a shadow ray this iteration
    uint32_t* queueCount;         // [parity][ext, shadow]
    uint32_t* activeSlots;        // slots that still have samples to do
    uint64_t* totals;             // [0] extension rays, [1] shadow rays, [2] finished samples, [3] live slot visits,
                                  // [4..7] nodes/tris fetched by closest / shadow traversal (COUNT builds)
    const uint32_t* pixelXY;      // pixel-of-shard -> x | y << 16
};

struct RenderParams {
    uint32_t numSlots, numPixels, stripes;
    uint32_t sppBegin, sppCount;
    int32_t rngSeed;
    float timeStart, timeEnd;
    uint32_t imageWidth, imageHeight;
    uint32_t countSlots;          // SLRHIP_FLAG_COUNT_TRAVERSAL: also count live slots per shade launch
};

void launchResetSlots(const PathBuffers& pb, const RenderParams& rp, bool clearAccumulators, hipStream_t stream);
void launchTraceClosest(const DevScene& sc, const PathBuffers& pb, uint32_t parity, uint32_t blocks, bool count, hipStream_t stream);
void launchTraceShadow(const DevScene& sc, const PathBuffers& pb, uint32_t parity, uint32_t blocks, bool count, hipStream_t stream);
void launchShade(const DevScene& sc, const PathBuffers& pb, const RenderParams& rp, uint32_t parity, hipStream_t stream);
void launchResolve(const PathBuffers& pb, const RenderParams& rp, float* dst, hipStream_t stream);
void launchTraceBatch(const DevScene& sc, const float4* org, const float4* dir, float4* out, uint32_t n, hipStream_t stream);

} // namespace slrhip
