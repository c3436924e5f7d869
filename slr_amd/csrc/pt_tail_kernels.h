// pt_tail_kernels.h — the end of a render call in ONE launch (gfx950, wave64).
//
// Why: pixels whose paths are long hand out their passes more slowly than the others, and the longest single paths run ~100
// bounces beyond the average, so the second half of the iterations of a frame advances an ever smaller number of live slots
// (per-iteration logs, profiles/r02_g_iter_*.txt: 22 M slots, 7.7 M live after iteration 104 of 208, 0.9 M after 112, 4 600 after
// 160), and each of those iterations still costs three launches that scan or skip every slot block: 38 of 394 ms at N = 1 and
// 11 of 56 ms at the N = 8 shard size.
//
// What: once at most RenderParams::tailSlots slots are live the wavefront stops (PathBuffers::tailMode) and
//   k_tail_collect  lists the live slots (one atomic per workgroup);
//   k_tail          gives every lane one listed slot and runs it until the slot has nothing left to do: per bounce the extension
//                   ray and the pending shadow ray through the one-lane-per-ray traversal of pt_traverse.h, then logicSlot —
//                   the very function k_shade calls; when the path ends, accumulateSample and, if the pixel has passes left,
//                   startSample — what k_shade does for a finished path.  A lane whose slot went idle takes the next listed
//                   one (one atomic per wave and refill).
// Passes in the tail: a slot goes idle only when its pixel has run out of passes, so a pixel that still has some has all its K
// stripes alive; stripe s takes the passes next + s, next + s + K, ... of its pixel (next = the pixel's counter when the tail
// took over).  Which stripe renders which pass therefore still depends on path lengths only: frames stay reproducible, every
// pass is rendered once, and the image differs from the pure wavefront schedule only in the grouping of a pixel's float sum
// over its stripes — not at all with one stripe.  But when the tail takes over depends on the number of live slots, hence on
// the shard size: with it the sum of the shards of a frame no longer equals the unsharded frame to the last bit at a fixed
// stripe count (tests/test_gpu_parity.py::test_eight_tile_shards_...): with a fixed stripe count it is opt-in
// (SLRHIP_FLAG_TAIL_KERNEL); with the automatic one — which itself depends on the shard size — it is always on.
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
    uint32_t taken = 0;                                        // passes this lane's slot has started in the tail
    uint32_t extRays = 0, shadowRays = 0, wentIdle = 0;
    bool listDrained = false;                                  // wave-uniform: the cursor is past the end of the list
    TravCount cnt = {0, 0};
    uint32_t* stack = tlds.stack + threadIdx.x;

    // <= 102 turns per sample; the passes a lane can still be handed are bounded by the call's pass count (a constant
    // bound failed one-lane renders of very many passes)
    const uint64_t guardTurns = 104ull * ((uint64_t)rp.sppCount + 2ull) + 1024ull;
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
                if (i < n) { slot = list[i]; taken = 0; }
            }
            listDrained = start + want >= n;
        }
        if (__ballot(slot != kTailNone) == 0) break;
        if (slot == kTailNone) continue;

        const uint32_t flags = pb.flags[slot];
        const uint32_t state = F_STATE(flags);
        if (state == ST_IDLE) { slot = kTailNone; continue; }    // not expected (the list holds live slots): nothing to do
        if (state == ST_REGEN) {
            // the path ended at this lane's last turn, or the slot has not started its first sample yet: sensor->add, next pass
            uint32_t samplesDone = 0;
            if (F_HASPATH(flags)) {
                const uint4 hdr = pb.hdr[(size_t)slot * pb.hdrStride];
                accumulateSample<S>(pb, rp, slot, flags, hdr, false, S());
                samplesDone = hdr.x + 1u;
            }
            const SlotAddr at = slotAddr(rp, slot);
            // a slot that has not started yet (possible only when the tail takes over at the first iteration) owns pass `stripe`
            const uint32_t pass = F_HASPATH(flags) ? pb.nextSample[at.pix] + at.stripe + rp.stripes * taken : at.stripe;      // relative to sppBegin
            if (pass >= rp.sppCount) {
                pb.flags[slot] = F_MAKE((uint32_t)ST_IDLE, 0u, 0u, 0u, 0u, 0u);
                ++wentIdle;
                pb.hdr[(size_t)slot * pb.hdrStride] = make_uint4(samplesDone, 0u, 0u, 0u);
                slot = kTailNone;
            }
            else {
                startSample<S>(sc, pb, rp, slot, at.pix, rp.sppBegin + pass, samplesDone);
                if (F_HASPATH(flags)) ++taken;
            }
            continue;
        }
        // ---- one bounce: the two rays of this slot, then the logic visit -----------------------------------------------------
        const float4 o = pb.rayOrg[(size_t)slot * pb.rayStride];
        if (state != ST_FINISH) {
            const float4 d = pb.rayDir[(size_t)slot * pb.rayStride];
            HitRec hit;
            traverse<false, false>(sc, sc.nodes, sc.leafTris, tlds.top, numTop, V3(o.x, o.y, o.z), V3(d.x, d.y, d.z), o.w, d.w, &hit, stack, &cnt,
                                   pb.errorWord);
            pb.hit[slot] = make_float4(__uint_as_float(hit.tri), hit.t, hit.b1, hit.b2);
            ++extRays;
        }
        if (F_SHADOW(flags)) {
            // Scene::testVisibility (SurfaceObject.cpp:418-430): the shadow ray starts where the extension ray does
            const float4 d = pb.shadowDir[slot];
            HitRec hit;
            const bool occluded = traverse<true, false>(sc, sc.nodes, sc.leafTris, tlds.top, numTop, V3(o.x, o.y, o.z), V3(d.x, d.y, d.z), kRayEpsilon, d.w,
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
        if (idleNow) atomicAdd(&pb.activeSlots[0], 0u - idleNow);
    }
}

} // namespace slrhip
