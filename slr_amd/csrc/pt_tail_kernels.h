// pt_tail_kernels.h — the end of a render window in ONE launch (gfx950, wave64).
//
// Why: the longest single paths run ~100 bounces beyond the average, so once the work queues are exhausted the last iterations of
// a window advance an ever smaller number of live slots, and each of those iterations still costs two launches that scan or
// skip every slot block.
//
// What: once at most RenderParams::tailSlots slots are live the wavefront stops (PathBuffers::tailMode) and
//   k_tail_collect  lists the live slots (one atomic per workgroup);
//   k_tail          gives every lane one listed slot and runs it until the slot has nothing left to do: per bounce the extension
//                   ray and the pending shadow ray through the one-lane-per-ray traversal of pt_traverse.h, then logicSlot —
//                   the very function k_shade calls; when the path ends, writeResult and, if the queue of the slot's wave has
//                   a sample left, startSample — what k_shade does for a finished path.  A lane whose slot went idle takes
//                   the next listed one (one atomic per wave and refill).
// A sample's contribution is a function of (pixel, pass) alone and the sensor adds the contributions in pass order (k_fold), so
// the image is the same to the last bit whether the tail kernel or the wavefront kernels finish the window, at any slot count.
#pragma once
#include "pt_shade_kernels.h"
#include "pt_traverse.h"

namespace slrhip {

static const uint32_t kTailNone = 0xFFFFFFFFu;

#ifndef SLR_TAIL_WAVES
#define SLR_TAIL_WAVES 1
#endif
template <class S, bool LDS_TABLES, bool MF, bool MULTI = false, bool TEX = false>
__global__ __launch_bounds__(kShadeBlock) __attribute__((amdgpu_waves_per_eu(SLR_TAIL_WAVES))) void k_tail(DevScene sc, PathBuffers pb, RenderParams rp) {
    __shared__ ShadeLds<S::N != 3> lds;
    __shared__ TraceLds tlds;
    __shared__ uint32_t red[4];
    if (LDS_TABLES) stageShadeTables<S::N != 3>(sc, lds);
    const uint32_t numTop = stageTopNodes(sc, tlds);          // ends with the barrier that also covers the tables above
    const float* lightPMF = LDS_TABLES ? lds.lightPMF : sc.lightPMF;
    const float* lightCDF = LDS_TABLES ? lds.lightCDF : sc.lightCDF;

    const uint32_t n = pb.tailWords[0];
    const uint32_t* list = pb.tailList;
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t below = (1ull << lane) - 1ull;
    uint32_t slot = kTailNone;
    uint32_t extRays = 0, shadowRays = 0, wentIdle = 0;
    bool listDrained = false;                                  // wave-uniform: the cursor is past the end of the list
    TravCount cnt = {0, 0};
    uint32_t* stack = tlds.stack + threadIdx.x;

    // <= 102 turns per sample; the samples a lane can still be handed are bounded by the longest queue (a constant
    // bound failed one-lane renders of very many passes)
    const uint64_t guardTurns = 104ull * ((uint64_t)(rp.numRuns / rp.numWaves + 2u) * rp.runLength + 2ull) + 1024ull;
    for (uint64_t guard = 0; guard < guardTurns; ++guard) {
        // ---- lanes without a path take the next listed ones: one atomic per wave and refill --------------------------------
        const uint64_t idle = __ballot(slot == kTailNone);
        if (idle && !listDrained) {
            const uint32_t want = (uint32_t)__popcll(idle);
            uint32_t start = 0;
            if (lane == (uint32_t)__ffsll((long long)idle) - 1u) start = atomicAdd(&pb.tailWords[1], want);
            start = __shfl(start, __ffsll((long long)idle) - 1);
            if (slot == kTailNone) {
                const uint32_t i = start + (uint32_t)__popcll(idle & below);
                if (i < n) slot = list[i];
            }
            listDrained = start + want >= n;
        }
        if (__ballot(slot != kTailNone) == 0) break;
        if (slot == kTailNone) continue;

        const uint32_t flags = pb.flags[slot];
        const uint32_t state = F_STATE(flags);
        if (state == ST_IDLE) { slot = kTailNone; continue; }    // not expected (the list holds live slots): nothing to do
        if (state == ST_REGEN) {
            // the path ended at this lane's last turn, or the slot has not started a sample yet: the finished sample's result,
            // then the next item of the queue of the slot's wave (by atomic: the slots of a wave may be held by
            // lanes of different waves here)
            if (F_HASPATH(flags)) {
                const uint4 hdr = pb.hdr[(size_t)slot * pb.hdrStride];
                writeResult<S>(pb, rp, slot, flags, hdr, false, S());
            }
            const WorkItem w = workItemOf(rp, slot >> 6, atomicAdd(&pb.cursor[slot >> 6], 1u));
            if (!w.valid) {
                pb.flags[slot] = F_MAKE((uint32_t)ST_IDLE, 0u, 0u, 0u, 0u, 0u);
                ++wentIdle;
                slot = kTailNone;
            }
            else startSample<S>(sc, pb, rp, slot, w.pix, w.pass);
            continue;
        }
        // ---- one bounce: the two rays of this slot, then the logic visit -----------------------------------------------------
        const float4 o = pb.rayOrg[(size_t)slot * pb.rayStride];
        if (state != ST_FINISH) {
            const float4 d = pb.rayDir[(size_t)slot * pb.rayStride];
            HitRec hit;
            // (the instance steps are compiled in for every scene: they cost a test per leaf reference here, and the tail kernel is
            // a third of a per cent of a frame)
            traverse<false, false, true>(sc, sc.nodes, sc.leafTris, tlds.top, numTop, V3(o.x, o.y, o.z), V3(d.x, d.y, d.z), o.w, d.w, &hit, stack, &cnt,
                                         pb.errorWord);
            pb.hit[slot] = make_float4(__uint_as_float(hit.tri), hit.t, hit.b1, hit.b2);
            if (pb.hitInstance) pb.hitInstance[slot] = hit.inst;
            ++extRays;
        }
        if (F_SHADOW(flags)) {
            // Scene::testVisibility (SurfaceObject.cpp:418-430): the shadow ray starts where the extension ray does
            const float4 d = pb.shadowDir[slot];
            HitRec hit;
            const bool occluded = traverse<true, false, true>(sc, sc.nodes, sc.leafTris, tlds.top, numTop, V3(o.x, o.y, o.z), V3(d.x, d.y, d.z), kRayEpsilon, d.w,
                                                        &hit, stack, &cnt, pb.errorWord);
            pb.visible[slot] = occluded ? 0u : 1u;
            ++shadowRays;
        }
        bool emitExt = false, emitShadow = false, emitRegen = false;
        uint32_t fl = flags;
        S unusedSum;
        SlotLoads<S> in;
        in.issue(sc, pb, rp, slot);
        logicSlot<S, LDS_TABLES, MF, MULTI, TEX, false>(sc, pb, rp, lds, lightPMF, lightCDF, slot, in, fl, pb.visible[slot], unusedSum, emitExt, emitShadow, emitRegen);
        // a finished path is in ST_REGEN now: accumulated at the next turn of this loop
    }
    if (slot != kTailNone) atomicOr(pb.errorWord, ERR_CONSUMER_IDLE);      // the bound was hit: never expected, fails the render loudly

    // ---- statistics and the live-slot count: one atomic per workgroup each ------------------------------------------------------
    const auto blockSum = [&](uint32_t v) -> uint32_t {
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
        __syncthreads();
        if (lane == 0) red[threadIdx.x >> 6] = v;
        __syncthreads();
        return red[0] + red[1] + red[2] + red[3];
    };
    const uint32_t e = blockSum(extRays), sh = blockSum(shadowRays), idleNow = blockSum(wentIdle);
    if (threadIdx.x == 0) {
        if (e) atomicAdd((unsigned long long*)&pb.totals[totalIndex(T_EXT_RAYS, blockIdx.x % kShards)], (unsigned long long)e);
        if (sh) atomicAdd((unsigned long long*)&pb.totals[totalIndex(T_SHADOW_RAYS, blockIdx.x % kShards)], (unsigned long long)sh);
        if (idleNow) atomicAdd(&pb.tailIdled[0], idleNow);      // the host checks it against the list length, then clears the live count
    }
}

} // namespace slrhip
