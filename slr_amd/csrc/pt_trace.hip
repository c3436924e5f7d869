// pt_trace.hip — BVH traversal kernels for MI355X (gfx950, wave64).
//
// Replaces Scene::intersect -> SurfaceObjectAggregate::intersect -> QBVH::intersect
// (Core/SurfaceObject.cpp:408-416,267-269; Accelerator/QBVH.h:295-339), Triangle::intersect
// (Surface/TriangleMesh.cpp:131-178) and Scene::testVisibility (SurfaceObject.cpp:418-430).
//
// The 64-ray-BATCH schedule (SLRHIP_FLAG_TRACE_BATCH; the default is the wave-specialised kernel of pt_trace_ws.hip, same
// results): persistent workgroups (a fixed number per CU) stage the top of the tree — the first kTopNodes breadth-first
// nodes — in LDS once, then stride over the slots (extension rays) or their region of the shadow queue, 64 rays per wave at a
// time.  Also serves slrhip_trace_rays (k_trace_batch).  One lane = one ray; per node four child slabs are tested from six 16-byte loads
// whose near/far selection (QBVH.h:66-71: invRayDir > 0 ? min : max) is folded into per-ray load
// offsets.  The nearest hit child is descended into directly, the others go on a per-lane stack in
// LDS ([entry][lane]: conflict-free) that spills to scratch beyond kLdsStack entries.
#include <hip/hip_runtime.h>

#include "pt_traverse.h"

namespace slrhip {

static const int kBlocksPerCU = 6;     // (8 KiB stack + 18 KiB nodes) x 6 = 156 KiB of the CU's 160 KiB LDS

int traceBlocksPerCU() { return kBlocksPerCU; }

// Statistics: reduce over the workgroup through LDS, then ONE atomic per workgroup on the shard's line.
__device__ __forceinline__ void blockAdd(uint64_t* totals, uint32_t kind, uint32_t v, uint32_t* scratch /* 4 words of LDS */) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    __syncthreads();
    if ((threadIdx.x & 63u) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t t = scratch[0] + scratch[1] + scratch[2] + scratch[3];
        if (t) atomicAdd((unsigned long long*)&totals[totalIndex(kind, blockIdx.x % kShards)], (unsigned long long)t);
    }
}

// Extension rays: closest hit.  Walks ALL slots (no queue): a slot has a ray in flight iff its state is
// FIRST_HIT or NEXT_HIT (flag values 2 and 3, pt_shade_kernels.h).
template <bool COUNT>
__global__ __launch_bounds__(kTraceBlock) void k_trace_closest(DevScene sc, PathBuffers pb, uint32_t numSlots, uint32_t parity, uint32_t tailSlots) {
    __shared__ TraceLds lds;
    __shared__ uint32_t red[4];
    if (pb.activeSlots[0] == 0) return;            // every slot is out of passes (uniform)
    if (tailModeBegins(pb, tailSlots, parity)) return;      // the last paths go to the tail kernel (pt_kernels.h)
    if (blockIdx.x == 0 && threadIdx.x < Q_KINDS * kShards) {
        // clear the counter set the logic kernel of this iteration fills
        pb.queueCount[queueCounterIndex(parity ^ 1, threadIdx.x / kShards, threadIdx.x % kShards)] = 0;
    }
    const uint32_t numTop = stageTopNodes(sc, lds);
    TravCount cnt = {0, 0};
    uint32_t rays = 0;
    const uint32_t stride = gridDim.x * kTraceBlock;
    for (uint32_t slot = blockIdx.x * kTraceBlock + threadIdx.x; slot < numSlots; slot += stride) {
        const uint32_t state = pb.flags[slot] & 7u;
        if (state == 2u || state == 3u) {
            const float4 o = pb.rayOrg[(size_t)slot * pb.rayStride];
            const float4 d = pb.rayDir[(size_t)slot * pb.rayStride];
            HitRec hit;
            traverse<false, COUNT>(sc, sc.nodes, sc.leafTris, lds.top, numTop, V3(o.x, o.y, o.z), V3(d.x, d.y, d.z), o.w, d.w, &hit,
                                   lds.stack + threadIdx.x, &cnt, pb.errorWord);
            pb.hit[slot] = make_float4(__uint_as_float(hit.tri), hit.t, hit.b1, hit.b2);
            ++rays;
        }
    }
    blockAdd(pb.totals, T_EXT_RAYS, rays, red);
    if (COUNT) { blockAdd(pb.totals, T_NODES_CLOSEST, cnt.nodes, red); blockAdd(pb.totals, T_TRIS_CLOSEST, cnt.tris, red); }
}

// Shadow rays: Scene::testVisibility (SurfaceObject.cpp:418-430) = "no hit in [eps, d(1-eps)]".
// Workgroup b serves queue region b % kShards (gridDim is a multiple of kShards).
template <bool COUNT>
__global__ __launch_bounds__(kTraceBlock) void k_trace_shadow(DevScene sc, PathBuffers pb, uint32_t shardCapacity, uint32_t parity) {
    __shared__ TraceLds lds;
    __shared__ uint32_t red[4];
    if (pb.activeSlots[0] == 0 || pb.tailMode[0]) return;
    const uint32_t shard = blockIdx.x % kShards;
    const uint32_t n = pb.queueCount[queueCounterIndex(parity, Q_SHADOW, shard)];
    const uint32_t numTop = stageTopNodes(sc, lds);
    TravCount cnt = {0, 0};
    uint32_t rays = 0;
    const uint32_t* queue = pb.shadowQueue + (size_t)shard * shardCapacity;
    const uint32_t stride = (gridDim.x / kShards) * kTraceBlock;
    for (uint32_t i = (blockIdx.x / kShards) * kTraceBlock + threadIdx.x; i < n; i += stride) {
        const uint32_t slot = queue[i];
        const float4 o = pb.rayOrg[(size_t)slot * pb.rayStride];
        const float4 d = pb.shadowDir[slot];
        HitRec hit;
        const bool occluded = traverse<true, COUNT>(sc, sc.nodes, sc.leafTris, lds.top, numTop, V3(o.x, o.y, o.z), V3(d.x, d.y, d.z),
                                                    kRayEpsilon, d.w, &hit, lds.stack + threadIdx.x, &cnt, pb.errorWord);
        pb.visible[slot] = occluded ? 0u : 1u;
        ++rays;
    }
    blockAdd(pb.totals, T_SHADOW_RAYS, rays, red);
    if (COUNT) { blockAdd(pb.totals, T_NODES_SHADOW, cnt.nodes, red); blockAdd(pb.totals, T_TRIS_SHADOW, cnt.tris, red); }
}

// Stand-alone closest-hit batch (parity tests of the traversal alone; not on the render path).
__global__ __launch_bounds__(kTraceBlock) void k_trace_batch(DevScene sc, const float4* org, const float4* dir, float4* out, uint32_t n) {
    __shared__ TraceLds lds;
    const uint32_t numTop = stageTopNodes(sc, lds);
    TravCount cnt = {0, 0};
    for (uint32_t i = blockIdx.x * kTraceBlock + threadIdx.x; i < n; i += gridDim.x * kTraceBlock) {
        const float4 o = org[i], d = dir[i];
        HitRec hit;
        const bool found = traverse<false, false>(sc, sc.nodes, sc.leafTris, lds.top, numTop, V3(o.x, o.y, o.z), V3(d.x, d.y, d.z), o.w, d.w,
                                                  &hit, lds.stack + threadIdx.x, &cnt);
        out[i] = found ? make_float4(__uint_as_float(hit.tri), hit.t, hit.b1, hit.b2) : make_float4(__uint_as_float(0xFFFFFFFFu), INFINITY, 0.f, 0.f);
    }
}

void launchTraceClosest(const DevScene& sc, const PathBuffers& pb, const RenderParams& rp, uint32_t parity, uint32_t blocks, bool count,
                        hipStream_t stream) {
    if (count) hipLaunchKernelGGL(k_trace_closest<true>, dim3(blocks), dim3(kTraceBlock), 0, stream, sc, pb, rp.numSlots, parity, rp.tailSlots);
    else hipLaunchKernelGGL(k_trace_closest<false>, dim3(blocks), dim3(kTraceBlock), 0, stream, sc, pb, rp.numSlots, parity, rp.tailSlots);
}
void launchTraceShadow(const DevScene& sc, const PathBuffers& pb, const RenderParams& rp, uint32_t parity, uint32_t blocks, bool count,
                       hipStream_t stream) {
    blocks = (blocks + kShards - 1) / kShards * kShards;
    if (count) hipLaunchKernelGGL(k_trace_shadow<true>, dim3(blocks), dim3(kTraceBlock), 0, stream, sc, pb, rp.shardCapacity, parity);
    else hipLaunchKernelGGL(k_trace_shadow<false>, dim3(blocks), dim3(kTraceBlock), 0, stream, sc, pb, rp.shardCapacity, parity);
}
void launchTraceBatch(const DevScene& sc, const float4* org, const float4* dir, float4* out, uint32_t n, hipStream_t stream) {
    uint32_t blocks = (n + kTraceBlock - 1) / kTraceBlock;
    if (blocks > 1536) blocks = 1536;
    hipLaunchKernelGGL(k_trace_batch, dim3(blocks), dim3(kTraceBlock), 0, stream, sc, org, dir, out, n);
}

} // namespace slrhip
