// pt_trace.hip — BVH traversal kernels for MI355X (gfx950, wave64).
//
// Replaces Scene::intersect -> SurfaceObjectAggregate::intersect -> QBVH::intersect
// (Core/SurfaceObject.cpp:408-416,267-269; Accelerator/QBVH.h:295-339), Triangle::intersect
// (Surface/TriangleMesh.cpp:131-178) and Scene::testVisibility (SurfaceObject.cpp:418-430).
//
// The stand-alone ray-query kernel behind slrhip_trace_rays (k_trace_batch); the render path traces with the wave-specialised
// kernel of pt_trace_ws.hip and, for the last paths of a call, with the same device function inside the tail kernel.
// Workgroups stage the top of the tree — the first kTopNodes breadth-first nodes — in LDS once, then stride over the rays, 64 per
// wave at a time.  One lane = one ray; per node four child slabs are tested from six 16-byte loads
// whose near/far selection (QBVH.h:66-71: invRayDir > 0 ? min : max) is folded into per-ray load
// offsets.  The nearest hit child is descended into directly, the others go on a per-lane stack in
// LDS ([entry][lane]: conflict-free) that spills to scratch beyond kLdsStack entries.
#include <hip/hip_runtime.h>

#include "pt_traverse.h"

namespace slrhip {

// Stand-alone closest-hit batch (parity tests of the traversal alone; not on the render path).
template <bool INST>
__global__ __launch_bounds__(kTraceBlock) void k_trace_batch(DevScene sc, const float4* org, const float4* dir, float4* out, uint32_t n) {
    __shared__ TraceLds lds;
    const uint32_t numTop = stageTopNodes(sc, lds);
    TravCount cnt = {0, 0};
    for (uint32_t i = blockIdx.x * kTraceBlock + threadIdx.x; i < n; i += gridDim.x * kTraceBlock) {
        const float4 o = org[i], d = dir[i];
        HitRec hit;
        const bool found = traverse<false, false, INST>(sc, sc.nodes, sc.leafTris, lds.top, numTop, V3(o.x, o.y, o.z), V3(d.x, d.y, d.z), o.w, d.w,
                                                  &hit, lds.stack + threadIdx.x, &cnt);
        out[i] = found ? make_float4(__uint_as_float(hit.tri), hit.t, hit.b1, hit.b2) : make_float4(__uint_as_float(0xFFFFFFFFu), INFINITY, 0.f, 0.f);
    }
}

void launchTraceBatch(const DevScene& sc, const float4* org, const float4* dir, float4* out, uint32_t n, hipStream_t stream) {
    uint32_t blocks = (n + kTraceBlock - 1) / kTraceBlock;
    if (blocks > 1536) blocks = 1536;
    if (sc.instances) hipLaunchKernelGGL(k_trace_batch<true>, dim3(blocks), dim3(kTraceBlock), 0, stream, sc, org, dir, out, n);
    else hipLaunchKernelGGL(k_trace_batch<false>, dim3(blocks), dim3(kTraceBlock), 0, stream, sc, org, dir, out, n);
}

} // namespace slrhip
