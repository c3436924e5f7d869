// pt_trace_ws.hip — wave-specialised BVH traversal kernels (gfx950, wave64).
//
// Same results as pt_trace.hip (Scene::intersect / testVisibility, Core/SurfaceObject.cpp:408-430; QBVH::intersect,
// Accelerator/QBVH.h:295-339; Triangle::intersect, Surface/TriangleMesh.cpp:131-178), different schedule.
//
// Why: PMC on the batch kernels showed 19-21 % VALU lane utilisation (SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU)):
// a wave of 64 rays runs until its LONGEST ray is done, and in this scene most rays need 3-4 nodes while a few need 20+.
// Refilling idle lanes needs new rays without stalling the lanes that are still traversing, and loads return in order
// per wave (vmcnt), so a wave cannot prefetch from HBM behind its own node fetches.  Hence two roles per workgroup:
//
//   wave 0      PRODUCER: streams the slot arrays in order (coalesced 1 KiB bursts, four 64-slot sub-chunks in flight),
//               keeps the slots that have a ray, and appends them to a ring of rays in LDS.
//   waves 1..NC CONSUMERS (NC = 7; 15 on large trees; SLRHIP_WS_NC): one lane = one ray; whenever kRefill or more lanes are idle the wave reserves that many ring
//               entries (one LDS compare-and-swap by lane 0) and the idle lanes start on them, while the other lanes
//               carry on mid-traversal.  A traversal step is one node (four slab tests) or ONE triangle, so lanes in
//               different phases interleave at fine grain.
//
// All hand-offs are LDS atomics at workgroup scope; no global atomics, no inter-workgroup communication.  Every spin is
// bounded (kSpinLimit) so a logic error ends the kernel instead of hanging the GPU.
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include <algorithm>

#include "pt_device.h"
#include "pt_kernels.h"
#include "pt_tex.h"

namespace slrhip {

// NC = consumer waves per workgroup (3 -> 256 threads, 7 -> 512, 15 -> 1 024); the ring holds 64 x (NC + 1) rays (512 for NC = 15)
static const int kWsLdsStack = 11;          // + 1 trash row = 12 rows of 64 lanes = 3 KiB per consumer wave
static const int kWsSpill = 53;            // 11 + 53 = the reference's 64-entry stack (QBVH.h:299)
static const int kSub = 2;                 // 64-slot sub-chunks the producer keeps in flight
#ifndef SLR_WS_AHEAD
#define SLR_WS_AHEAD 4
#endif
static const int kAhead = SLR_WS_AHEAD;               // chunks whose state flags the producer reads in one round trip (see k_trace_ws)
static_assert(64 % kAhead == 0 && kAhead <= 32, "the producer's 64-chunk dead-block mask is consumed kAhead bits at a time");
static uint32_t g_refill = 20;             // idle lanes that trigger a refill (SLRHIP_WS_REFILL)
static int g_consumers = 0;                // SLRHIP_WS_NC: consumer waves per workgroup (3, 7 or 15); 0 = by tree size: 15 on quantized (large) trees, else 7
static const uint32_t kSpinLimit = 1u << 22;
#ifndef SLR_WS_CHAIN
#define SLR_WS_CHAIN 1
#endif
static const bool kChain = SLR_WS_CHAIN != 0;
#ifdef SLR_WS_NOSPILL
static const bool kNoSpill = true;
#else
static const bool kNoSpill = false;
#endif
static const uint32_t kIdle = 0xFFFFFFFFu;
// SLR_WS_INST_SAVE 1: the world ray of a lane inside an instance is kept in six more registers (under the 64-VGPR cap: in
// scratch, where the compiler decides) instead of being read again from the slot's record when the lane leaves the instance
#ifndef SLR_WS_INST_SAVE
#define SLR_WS_INST_SAVE 1      // measured on the 1 250-instance grid: traversal launch 6 350 -> 5 770 us (+8 % Msamples/s); 0: variant builds
#endif
// SLR_WS_TOP (variant builds; 0 = off, the default): the first SLR_WS_TOP nodes of the breadth-first tree — levels 0-3 of a
// four-wide tree are 85 nodes — are staged in LDS by every workgroup and read from there (north_star's "nodes staged through
// LDS"; DESIGN.md has the measurement that decided the default).  128-byte nodes in rows of 144 B, 64-byte quantized nodes in
// rows of 80 B, so that lanes at different nodes spread over the banks.
#ifndef SLR_WS_TOP
#define SLR_WS_TOP 0
#endif
static const uint32_t kTop = SLR_WS_TOP;

static int wsConsumers(bool quantized) { return g_consumers ? g_consumers : (quantized ? 15 : 7); }

template <int NC>
struct WsLds {
    static constexpr uint32_t kRing = NC == 15 ? 512u : 64u * (NC + 1);      // ray ring entries (power of two)
    float4 org[kRing];                     // xyz + tmin
    float4 dir[kRing];                     // xyz + tmax
    uint32_t slot[kRing];
    uint32_t stack[NC][(kWsLdsStack + 1) * 64];
    uint32_t tail;                         // entries published by the producer
    uint32_t reserved;                     // entries claimed by consumers
    uint32_t released;                     // entries consumers have finished reading (ring space)
    uint32_t done;                         // producer has published its last entry
    uint32_t red[NC + 1];
    float4 top[kTop ? kTop * 9 : 1];       // SLR_WS_TOP: staged nodes, row stride 9 (float nodes) or 5 (quantized) float4
};

int traceWsBlocksPerCU(bool quantized) {
    static const bool init = [] {
        if (const char* e = tuningEnv("SLRHIP_WS_REFILL")) g_refill = (uint32_t)atoi(e);
        if (const char* e = tuningEnv("SLRHIP_WS_NC")) { const int n = atoi(e); g_consumers = n == 3 ? 3 : n == 15 ? 15 : n == 7 ? 7 : 0; }
        if (g_refill < 1) g_refill = 1;
        if (g_refill > 64) g_refill = 64;
        return true;
    }();
    (void)init;
    if (const char* e = tuningEnv("SLRHIP_WS_BLOCKS_PER_CU")) { const int b = atoi(e); if (b >= 1 && b <= 8) return b; }   // experiments
    const int nc = wsConsumers(quantized);
    // 18.5 / 39.5 / 64.5 KiB of LDS per workgroup, 32 waves per CU either way (<= 64 VGPR); fewer where staged nodes (SLR_WS_TOP) need the room
    const size_t lds = nc == 15 ? sizeof(WsLds<15>) : nc == 7 ? sizeof(WsLds<7>) : sizeof(WsLds<3>);
    return std::min(nc == 15 ? 2 : nc == 7 ? 4 : 8, (int)(163840 / lds));
}

// streamed once per launch: keep them out of the vector L1 so that it stays with the BVH nodes
typedef float wsFloat4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ntLoad4(const float4* p) {
    const wsFloat4 v = __builtin_nontemporal_load(reinterpret_cast<const wsFloat4*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void ntStore4(float4* p, float4 q) {
    wsFloat4 v = {q.x, q.y, q.z, q.w};
    __builtin_nontemporal_store(v, reinterpret_cast<wsFloat4*>(p));
}

#define WS_LOAD(p, order) __hip_atomic_load((p), order, __HIP_MEMORY_SCOPE_WORKGROUP)
#define WS_STORE(p, v, order) __hip_atomic_store((p), (v), order, __HIP_MEMORY_SCOPE_WORKGROUP)

__device__ __forceinline__ void wsBlockAdd(uint64_t* totals, uint32_t kind, uint32_t v, uint32_t* scratch) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    __syncthreads();
    if ((threadIdx.x & 63u) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
        for (uint32_t w = 0; w < blockDim.x / 64; ++w) t += scratch[w];
        if (t) atomicAdd((unsigned long long*)&totals[totalIndex(kind, blockIdx.x % kShards)], (unsigned long long)t);
    }
}

// Append the rays of one 64-lane sub-chunk that are live; returns how many were appended (wave-uniform).
template <class LDS>
__device__ __forceinline__ uint32_t wsAppend(LDS& lds, uint32_t tailLocal, bool live, uint32_t slot, float4 o, float4 d) {
    const uint64_t mask = __ballot(live);
    if (live) {
        const uint32_t lane = threadIdx.x & 63u;
        const uint32_t pos = (tailLocal + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))) & (LDS::kRing - 1);
        lds.org[pos] = o;
        lds.dir[pos] = d;
        lds.slot[pos] = slot;
    }
    return (uint32_t)__popcll(mask);
}

// Wait until the ring can take `need` more entries.  Returns false if the bound was hit (never expected).
template <class LDS>
__device__ __forceinline__ bool wsWaitSpace(LDS& lds, uint32_t tailLocal, uint32_t need, uint32_t* waits) {
    for (uint32_t spin = 0; spin < kSpinLimit; ++spin) {
        const uint32_t rel = WS_LOAD(&lds.released, __ATOMIC_ACQUIRE);
        if (tailLocal + need - rel <= LDS::kRing) return true;
        ++*waits;
        __builtin_amdgcn_s_sleep(8);
    }
    return false;
}

// The consumer side: shared by the closest-hit and the any-hit kernels.
struct WsDebug {
    uint32_t steps = 0, idleSpins = 0, refills = 0, producerWaits = 0, nodeBlocks = 0, triBlocks = 0, activeLanes = 0;
    uint64_t cycles = 0, idleCycles = 0;
};

static const uint32_t kShadowBit = 0x80000000u;   // ring slot word: bit 31 = shadow (any-hit) ray, bits 30..0 = slot

struct WsCounts {
    uint32_t nodes[2] = {0, 0}, tris[2] = {0, 0};   // [0] closest, [1] shadow
};

// INST: the scene has instanced meshes (DevScene::instances).  A child reference that names an instance (leaf flag, count 0) takes
// the lane's ray to the mesh's local space — TransformedSurfaceObject::intersect, Core/SurfaceObject.cpp:307-317: origin as a point,
// direction as a vector and NOT renormalised, so distances stay world distances and tmin / tmax carry over — pushes kPopInstance
// and goes on at the root of the mesh's tree (same node and leaf arrays).  When kPopInstance comes off the stack the mesh is done
// and the world ray comes back from six registers of its own (they fit under the 64-VGPR cap without more scratch; reading it
// again from the slot's record — two 16-byte loads that miss the caches, the producer streamed them — was 8 % slower).
template <bool COUNT, int NC, bool QUANT, bool WIDE8, bool INST>
__device__ __forceinline__ void wsConsume(const DevScene& sc, const PathBuffers& pb, WsLds<NC>& lds, uint32_t refill, uint32_t numTop, WsCounts& cnt, WsDebug& dbg) {
    const uint64_t tStart = COUNT ? __builtin_readcyclecounter() : 0;
    constexpr uint32_t kRing = WsLds<NC>::kRing;
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t below = (1ull << lane) - 1ull;
    uint32_t* stack = lds.stack[(threadIdx.x >> 6) - 1] + lane;          // [entry][lane]
    const float4* __restrict__ nodes4 = sc.nodes;
    const float4* __restrict__ tris4 = sc.leafTris;

    uint32_t slot = kIdle;
    float ox = 0, oy = 0, oz = 0, dx = 0, dy = 0, dz = 0, idx = 0, idy = 0, idz = 0, tmin = 0, tmax = 0;
    uint32_t cur = 0;
    int sp = 0;
    uint32_t hitTri = 0xFFFFFFFFu;
    float hitT = INFINITY, hitB1 = 0.0f, hitB2 = 0.0f;
    int32_t inst = -1, hitInst = -1;           // INST: the instance the lane is inside / the one its closest hit went through
#if SLR_WS_INST_SAVE
    float wox = 0, woy = 0, woz = 0, wdx = 0, wdy = 0, wdz = 0;      // the world ray while the lane is inside an instance
#endif
    // a reference that is a packet of triangles (INST: not an instance reference, not kPopInstance)
    const auto isTriLeaf = [](uint32_t ref) -> bool {
        if (!INST) return (ref & kLeafFlag) != 0;
        const uint32_t n = (ref >> kLeafCountShift) & 0xFu;
        return (ref & kLeafFlag) != 0 && n != 0u && n != 15u;
    };
#ifdef SLR_WS_NOSPILL
    uint32_t* spill = nullptr;      // timing experiment only: no scratch, pushes beyond the LDS part are dropped
#else
    uint32_t spill[kWsSpill];
#endif
    uint32_t idleSpins = 0;

    for (;;) {
        const uint64_t idleMask = __ballot(slot == kIdle);
        const uint32_t nIdle = (uint32_t)__popcll(idleMask);
        if (nIdle >= refill) {
            uint32_t start = 0, take = 0;
            if (lane == 0) {
                // reserved <= tail at all times and tail only grows: reading reserved FIRST makes tail - reserved >= 0
                for (int attempt = 0; attempt < 64; ++attempt) {
                    uint32_t r = WS_LOAD(&lds.reserved, __ATOMIC_RELAXED);
                    const uint32_t t = WS_LOAD(&lds.tail, __ATOMIC_ACQUIRE);
                    take = min(t - r, nIdle);
                    if (take == 0) break;
                    if (__hip_atomic_compare_exchange_strong(&lds.reserved, &r, r + take, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                             __HIP_MEMORY_SCOPE_WORKGROUP)) { start = r; break; }
                    take = 0;
                }
            }
            start = __builtin_amdgcn_readfirstlane(start);
            take = __builtin_amdgcn_readfirstlane(take);
            if (take) {
                idleSpins = 0;
                if (COUNT) ++dbg.refills;
                const uint32_t rank = (uint32_t)__popcll(idleMask & below);
                if (slot == kIdle && rank < take) {
                    const uint32_t pos = (start + rank) & (kRing - 1);
                    const float4 o = lds.org[pos], d = lds.dir[pos];
                    slot = lds.slot[pos];
                    ox = o.x; oy = o.y; oz = o.z; tmin = o.w;
                    dx = d.x; dy = d.y; dz = d.z; tmax = d.w;
                    idx = 1.0f / dx; idy = 1.0f / dy; idz = 1.0f / dz;          // Vector3.h:60 reciprocal()
                    cur = 0; sp = 0;
                    hitTri = 0xFFFFFFFFu; hitT = INFINITY; hitB1 = 0.0f; hitB2 = 0.0f;
                    inst = -1; hitInst = -1;
                }
                // Ring space is handed back IN RESERVATION ORDER: `released` is a watermark, the producer overwrites
                // everything below it, so this wave may only move it past its own entries once every earlier
                // reservation (another wave, possibly still reading) has been released.  The wait is a few hundred
                // cycles at most: the earlier wave is between its compare-and-swap and the end of its own ring reads.
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");           // the ring reads above are complete
                if (lane == 0) {
                    uint32_t spin = 0;
                    for (; spin < kSpinLimit && WS_LOAD(&lds.released, __ATOMIC_ACQUIRE) != start; ++spin)
                        __builtin_amdgcn_s_sleep(1);
                    if (spin == kSpinLimit) atomicOr(pb.errorWord, ERR_RING_RELEASE);     // never expected; the render fails loudly
                    WS_STORE(&lds.released, start + take, __ATOMIC_RELEASE);
                }
            }
            else if (nIdle == 64) {
                // nothing in flight and nothing to take: finished, or the producer is behind
                if (WS_LOAD(&lds.done, __ATOMIC_ACQUIRE) && WS_LOAD(&lds.tail, __ATOMIC_ACQUIRE) == WS_LOAD(&lds.reserved, __ATOMIC_RELAXED)) break;
                if (++idleSpins > kSpinLimit) { if (lane == 0) atomicOr(pb.errorWord, ERR_CONSUMER_IDLE); break; }
                if (COUNT) {
                    const uint64_t t0 = __builtin_readcyclecounter();
                    __builtin_amdgcn_s_sleep(4);
                    dbg.idleCycles += __builtin_readcyclecounter() - t0;
                    ++dbg.idleSpins;
                }
                else __builtin_amdgcn_s_sleep(4);
                continue;
            }
        }

        if (COUNT) {
            ++dbg.steps;
            const uint64_t act = __ballot(slot != kIdle);
            dbg.activeLanes += (uint32_t)__popcll(act);
            if (__ballot(slot != kIdle && !(cur & kLeafFlag))) ++dbg.nodeBlocks;
        }
        if (INST && slot != kIdle && (cur & kLeafFlag) != 0 && !isTriLeaf(cur)) {
            // one step = entering or leaving an instance
            bool finished = false;
            if (cur == kPopInstance) {
#if SLR_WS_INST_SAVE
                ox = wox; oy = woy; oz = woz;
                dx = wdx; dy = wdy; dz = wdz;
#else
                const uint32_t s = slot & ~kShadowBit;
                const float4 o = pb.rayOrg[(size_t)s * pb.rayStride];
                const float4 d = (slot & kShadowBit) ? pb.shadowDir[s] : pb.rayDir[(size_t)s * pb.rayStride];
                ox = o.x; oy = o.y; oz = o.z;
                dx = d.x; dy = d.y; dz = d.z;
#endif
                inst = -1;
                if (sp == 0) finished = true;
                else {
                    --sp;
                    if (kNoSpill || sp < kWsLdsStack) cur = stack[sp * 64];
                    else cur = spill[sp - kWsLdsStack];
                }
            }
            else {
                const uint32_t k = cur & kLeafIndexMask;
                const float4* rec = sc.instances + (size_t)k * 9u + 4u;              // DevInstance::worldToLocal, columns
                const float4 c0 = rec[0], c1 = rec[1], c2 = rec[2], c3 = rec[3], meta = rec[4];
                // invert(sampledTF) * ray: Matrix4x4 x Point3D (Matrix4x4.h:75-81; the bottom row is 0 0 0 1, so w == 1 and there is
                // no division) and Matrix4x4 x Vector3D (:71-73)
                const float lx = c0.x * ox + c1.x * oy + c2.x * oz + c3.x * 1.0f;
                const float ly = c0.y * ox + c1.y * oy + c2.y * oz + c3.y * 1.0f;
                const float lz = c0.z * ox + c1.z * oy + c2.z * oz + c3.z * 1.0f;
                const float mx = c0.x * dx + c1.x * dy + c2.x * dz;
                const float my = c0.y * dx + c1.y * dy + c2.y * dz;
                const float mz = c0.z * dx + c1.z * dy + c2.z * dz;
#if SLR_WS_INST_SAVE
                wox = ox; woy = oy; woz = oz; wdx = dx; wdy = dy; wdz = dz;
#endif
                ox = lx; oy = ly; oz = lz;
                dx = mx; dy = my; dz = mz;
                if (sp < kWsLdsStack) { stack[sp * 64] = kPopInstance; ++sp; }
                else if (!kNoSpill && sp < kWsLdsStack + kWsSpill) { spill[sp - kWsLdsStack] = kPopInstance; ++sp; }
                else atomicOr(pb.errorWord, ERR_STACK_OVERFLOW);
                cur = __float_as_uint(meta.x);
                inst = (int32_t)k;
            }
            idx = 1.0f / dx; idy = 1.0f / dy; idz = 1.0f / dz;          // Vector3.h:60 reciprocal()
            if (finished) {
                if (slot & kShadowBit) pb.visible[slot & ~kShadowBit] = hitTri == 0xFFFFFFFFu ? 1u : 0u;
                else { pb.hit[slot] = make_float4(__uint_as_float(hitTri), hitT, hitB1, hitB2); pb.hitInstance[slot] = hitInst; }
                slot = kIdle;
            }
        }
        // (not `else`: a lane that has just entered or left an instance goes on with its node or triangle step in the same turn)
        if (slot != kIdle && !(INST && (cur & kLeafFlag) != 0 && !isTriLeaf(cur))) {
            bool finished = false;
            const bool leafAtTop = isTriLeaf(cur);
            if (!leafAtTop) {
                if (COUNT) { const bool sh = (slot & kShadowBit) != 0; cnt.nodes[0] += sh ? 0u : 1u; cnt.nodes[1] += sh ? 1u : 0u; }
                if (WIDE8) {
                    // Eight-wide quantized node (device_types.h QNode8): seven 16-byte loads, then the children one after the other —
                    // dequantize the near and far planes (fma(byte, scale, origin), rounded outwards by the host: supersets of the float
                    // boxes, same hits), slab test, keep the nearest hit child as the next node and push the others as they come
                    const char* nb = reinterpret_cast<const char*>(sc.nodes8);
                    const uint32_t nOff = cur * 128u;
                    const float4 v0 = *reinterpret_cast<const float4*>(nb + nOff);
                    const float4 v1 = *reinterpret_cast<const float4*>(nb + (nOff + 16u));
                    const float4 v2 = *reinterpret_cast<const float4*>(nb + (nOff + 32u));
                    const float4 v3 = *reinterpret_cast<const float4*>(nb + (nOff + 48u));
                    const float4 v4 = *reinterpret_cast<const float4*>(nb + (nOff + 64u));
                    const float4 ca = *reinterpret_cast<const float4*>(nb + (nOff + 80u));
                    const float4 cb = *reinterpret_cast<const float4*>(nb + (nOff + 96u));
                    const bool px = idx > 0.0f, py = idy > 0.0f, pz = idz > 0.0f;                   // QBVH.h:66-71
                    const uint32_t lox[2] = {__float_as_uint(v1.z), __float_as_uint(v1.w)}, loy[2] = {__float_as_uint(v2.x), __float_as_uint(v2.y)},
                                   loz[2] = {__float_as_uint(v2.z), __float_as_uint(v2.w)}, hix[2] = {__float_as_uint(v3.x), __float_as_uint(v3.y)},
                                   hiy[2] = {__float_as_uint(v3.z), __float_as_uint(v3.w)}, hiz[2] = {__float_as_uint(v4.x), __float_as_uint(v4.y)};
                    const uint32_t cc[8] = {__float_as_uint(ca.x), __float_as_uint(ca.y), __float_as_uint(ca.z), __float_as_uint(ca.w),
                                            __float_as_uint(cb.x), __float_as_uint(cb.y), __float_as_uint(cb.z), __float_as_uint(cb.w)};
                    float best = INFINITY;
                    uint32_t next = kInvalidChild;
#define WS_PUSH8(cond, ref)                                                             \
                    if (cond) {                                                         \
                        if (sp < kWsLdsStack) { stack[sp * 64] = (ref); ++sp; }         \
                        else if (!kNoSpill && sp < kWsLdsStack + kWsSpill) { spill[sp - kWsLdsStack] = (ref); ++sp; } \
                        else atomicOr(pb.errorWord, ERR_STACK_OVERFLOW);      /* the host falls back to the four-wide tree for deeper ones */ \
                    }
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        const int w = c >> 2, sh = 8 * (c & 3);
                        const uint32_t nqx = px ? lox[w] : hix[w], fqx = px ? hix[w] : lox[w];
                        const uint32_t nqy = py ? loy[w] : hiy[w], fqy = py ? hiy[w] : loy[w];
                        const uint32_t nqz = pz ? loz[w] : hiz[w], fqz = pz ? hiz[w] : loz[w];
                        const float nX = __builtin_fmaf((float)((nqx >> sh) & 0xFFu), v0.w, v0.x), fX = __builtin_fmaf((float)((fqx >> sh) & 0xFFu), v0.w, v0.x);
                        const float nY = __builtin_fmaf((float)((nqy >> sh) & 0xFFu), v1.x, v0.y), fY = __builtin_fmaf((float)((fqy >> sh) & 0xFFu), v1.x, v0.y);
                        const float nZ = __builtin_fmaf((float)((nqz >> sh) & 0xFFu), v1.y, v0.z), fZ = __builtin_fmaf((float)((fqz >> sh) & 0xFFu), v1.y, v0.z);
                        const float tn = fmaxf(fmaxf((nX - ox) * idx, (nY - oy) * idy), fmaxf((nZ - oz) * idz, tmin));
                        const float tf = fminf(fminf((fX - ox) * idx, (fY - oy) * idy), fminf((fZ - oz) * idz, tmax));
                        const bool hit = tn <= tf && cc[c] != kInvalidChild;
                        const bool closer = hit && tn < best;
                        const uint32_t displaced = closer ? next : cc[c];
                        const bool doPush = hit && displaced != kInvalidChild;
                        best = closer ? tn : best;
                        next = closer ? cc[c] : next;
                        WS_PUSH8(doPush, displaced)
                    }
#undef WS_PUSH8
                    if (next != kInvalidChild) cur = next;
                    else if (sp == 0) finished = true;
                    else {
                        --sp;
                        if (kNoSpill || sp < kWsLdsStack) cur = stack[sp * 64];
                        else cur = spill[sp - kWsLdsStack];
                    }
                }
                else {
                // float4 index inside a 128-byte node: 0..2 = min xyz, 3..5 = max xyz, 6 = children (QBVH.h:66-71 folded into offsets)
                float4 nX, nY, nZ, fX, fY, fZ, ch;
                if (QUANT) {
                    // 64-byte node, 8-bit child boxes (device_types.h QNodeQ): plane = fma(byte, scale, origin), rounded outwards by
                    // the host, so the boxes are supersets of the float boxes and the set of hits is unchanged
                    float4 v0, v1, v2;
                    if (kTop && cur < numTop) {
                        const float4* n = lds.top + cur * 5u;
                        v0 = n[0]; v1 = n[1]; v2 = n[2]; ch = n[3];
                    }
                    else {
                        const char* nb = reinterpret_cast<const char*>(sc.nodesQ);
                        const uint32_t nOff = cur * 64u;
                        v0 = *reinterpret_cast<const float4*>(nb + nOff);
                        v1 = *reinterpret_cast<const float4*>(nb + (nOff + 16u));
                        v2 = *reinterpret_cast<const float4*>(nb + (nOff + 32u));
                        ch = *reinterpret_cast<const float4*>(nb + (nOff + 48u));
                    }
                    const uint32_t qlox = __float_as_uint(v1.z), qloy = __float_as_uint(v1.w), qloz = __float_as_uint(v2.x);
                    const uint32_t qhix = __float_as_uint(v2.y), qhiy = __float_as_uint(v2.z), qhiz = __float_as_uint(v2.w);
                    const uint32_t nqx = idx > 0.0f ? qlox : qhix, fqx = idx > 0.0f ? qhix : qlox;      // QBVH.h:66-71
                    const uint32_t nqy = idy > 0.0f ? qloy : qhiy, fqy = idy > 0.0f ? qhiy : qloy;
                    const uint32_t nqz = idz > 0.0f ? qloz : qhiz, fqz = idz > 0.0f ? qhiz : qloz;
#define WS_DEQ(q, sh, scale, org) __builtin_fmaf((float)(((q) >> (sh)) & 0xFFu), scale, org)
#define WS_DEQ4(q, scale, org) make_float4(WS_DEQ(q, 0, scale, org), WS_DEQ(q, 8, scale, org), WS_DEQ(q, 16, scale, org), WS_DEQ(q, 24, scale, org))
                    nX = WS_DEQ4(nqx, v0.w, v0.x); fX = WS_DEQ4(fqx, v0.w, v0.x);
                    nY = WS_DEQ4(nqy, v1.x, v0.y); fY = WS_DEQ4(fqy, v1.x, v0.y);
                    nZ = WS_DEQ4(nqz, v1.y, v0.z); fZ = WS_DEQ4(fqz, v1.y, v0.z);
#undef WS_DEQ4
#undef WS_DEQ
                }
                else {
                    // 32-bit byte offsets from the (uniform) array base: SGPR-base + VGPR-offset addressing, one VALU op per load
                    // instead of a 64-bit address computation (node array < 4 GiB: checked at upload)
                    const int nx = idx > 0.0f ? 0 : 3, fx = 3 - nx;
                    const int ny = idy > 0.0f ? 1 : 4, fy = 5 - ny;
                    const int nz = idz > 0.0f ? 2 : 5, fz = 7 - nz;
                    if (kTop && cur < numTop) {
                        const float4* n = lds.top + cur * 9u;
                        nX = n[nx]; nY = n[ny]; nZ = n[nz]; fX = n[fx]; fY = n[fy]; fZ = n[fz]; ch = n[6];
                    }
                    else {
                    const char* nb = reinterpret_cast<const char*>(nodes4);
                    const uint32_t nOff = cur * 128u;
                    nX = *reinterpret_cast<const float4*>(nb + (nOff + (uint32_t)nx * 16u));
                    nY = *reinterpret_cast<const float4*>(nb + (nOff + (uint32_t)ny * 16u));
                    nZ = *reinterpret_cast<const float4*>(nb + (nOff + (uint32_t)nz * 16u));
                    fX = *reinterpret_cast<const float4*>(nb + (nOff + (uint32_t)fx * 16u));
                    fY = *reinterpret_cast<const float4*>(nb + (nOff + (uint32_t)fy * 16u));
                    fZ = *reinterpret_cast<const float4*>(nb + (nOff + (uint32_t)fz * 16u));
                    ch = *reinterpret_cast<const float4*>(nb + (nOff + 96u));
                    }
                }
                // slab test of QBVH::Node::intersect (QBVH.h:55-76): tNear <= tFar.  (Tried: fma(plane, id, -(o * id)) on
                // padded boxes, half the arithmetic — but for rays that start ON a surface the cancellation noise near t = 0
                // admits the boxes around the origin: 3x the triangle tests for shadow rays, 4x slower on the 10 M grid.)
                // (Tried in round 2: the 24 subtractions + 24 multiplications as packed fp32, v_pk_add_f32 / v_pk_mul_f32 — same
                // roundings, 110 -> 87 VALU in this block — measured 23 % SLOWER per launch: 1.55 -> 1.90 ms at 22 M slots.)
                const float tn0 = fmaxf(fmaxf((nX.x - ox) * idx, (nY.x - oy) * idy), fmaxf((nZ.x - oz) * idz, tmin));
                const float tn1 = fmaxf(fmaxf((nX.y - ox) * idx, (nY.y - oy) * idy), fmaxf((nZ.y - oz) * idz, tmin));
                const float tn2 = fmaxf(fmaxf((nX.z - ox) * idx, (nY.z - oy) * idy), fmaxf((nZ.z - oz) * idz, tmin));
                const float tn3 = fmaxf(fmaxf((nX.w - ox) * idx, (nY.w - oy) * idy), fmaxf((nZ.w - oz) * idz, tmin));
                const float tf0 = fminf(fminf((fX.x - ox) * idx, (fY.x - oy) * idy), fminf((fZ.x - oz) * idz, tmax));
                const float tf1 = fminf(fminf((fX.y - ox) * idx, (fY.y - oy) * idy), fminf((fZ.y - oz) * idz, tmax));
                const float tf2 = fminf(fminf((fX.z - ox) * idx, (fY.z - oy) * idy), fminf((fZ.z - oz) * idz, tmax));
                const float tf3 = fminf(fminf((fX.w - ox) * idx, (fY.w - oy) * idy), fminf((fZ.w - oz) * idz, tmax));
                const uint32_t c0 = __float_as_uint(ch.x), c1 = __float_as_uint(ch.y), c2 = __float_as_uint(ch.z), c3 = __float_as_uint(ch.w);
                const bool h0 = tn0 <= tf0 && c0 != kInvalidChild;
                const bool h1 = tn1 <= tf1 && c1 != kInvalidChild;
                const bool h2 = tn2 <= tf2 && c2 != kInvalidChild;
                const bool h3 = tn3 <= tf3 && c3 != kInvalidChild;
                // nearest hit child is visited next; the rest go on the stack
                float best = INFINITY;
                uint32_t next = kInvalidChild;
                if (h0) { best = tn0; next = c0; }
                if (h1 && tn1 < best) { best = tn1; next = c1; }
                if (h2 && tn2 < best) { best = tn2; next = c2; }
                if (h3 && tn3 < best) { best = tn3; next = c3; }
                // the hit children other than `next` go on the stack in child order.  Fast path (all of them fit in the LDS
                // part): branch-free, a child that is not pushed writes the trash row kWsLdsStack of this lane's column.
                const bool v0 = h0 && c0 != next, v1 = h1 && c1 != next, v2 = h2 && c2 != next, v3 = h3 && c3 != next;
                const int nPush = (int)v0 + (int)v1 + (int)v2 + (int)v3;
                if (sp + nPush <= kWsLdsStack) {
                    int p = sp;
                    stack[(v0 ? p : kWsLdsStack) * 64] = c0; p += (int)v0;
                    stack[(v1 ? p : kWsLdsStack) * 64] = c1; p += (int)v1;
                    stack[(v2 ? p : kWsLdsStack) * 64] = c2; p += (int)v2;
                    stack[(v3 ? p : kWsLdsStack) * 64] = c3; p += (int)v3;
                    sp = p;
                }
                else {
#define WS_PUSH(cond, ref)                                                              \
                    if (cond) {                                                         \
                        if (sp < kWsLdsStack) { stack[sp * 64] = (ref); ++sp; }         \
                        else if (!kNoSpill && sp < kWsLdsStack + kWsSpill) { spill[sp - kWsLdsStack] = (ref); ++sp; } \
                        else atomicOr(pb.errorWord, ERR_STACK_OVERFLOW);      /* the host rejects trees that could get here */ \
                    }
                    WS_PUSH(v0, c0)
                    WS_PUSH(v1, c1)
                    WS_PUSH(v2, c2)
                    WS_PUSH(v3, c3)
#undef WS_PUSH
                }
                if (next != kInvalidChild) cur = next;
                else if (sp == 0) finished = true;
                else {
                    --sp;
                    if (kNoSpill || sp < kWsLdsStack) cur = stack[sp * 64];
                    else cur = spill[sp - kWsLdsStack];
                }
                }
            }
            // kChain: a lane whose node step ended on a leaf tests that leaf's first triangle in the same iteration
            if (COUNT && __ballot(!finished && (kChain ? isTriLeaf(cur) : leafAtTop)) && (threadIdx.x & 63u) == (uint32_t)__ffsll((long long)__ballot(true)) - 1u)
                ++dbg.triBlocks;      // counted by the first active lane; summed over lanes at the end
            if (!finished && (kChain ? isTriLeaf(cur) : leafAtTop)) {
                // ONE triangle of the leaf packet per step; the reference tests them in order (QBVH.h:322-327)
                const uint32_t first = cur & kLeafIndexMask;
                const uint32_t count = (cur >> kLeafCountShift) & 0xF;
                const char* tb = reinterpret_cast<const char*>(tris4);
                const uint32_t tOff = first * 48u;
                const float4 a = *reinterpret_cast<const float4*>(tb + tOff);
                const float4 b = *reinterpret_cast<const float4*>(tb + (tOff + 16u));
                const float4 c = *reinterpret_cast<const float4*>(tb + (tOff + 32u));
                const V3 v0(a.x, a.y, a.z), e1(b.x, b.y, b.z), e2(c.x, c.y, c.z);
                const uint32_t triIdx = __float_as_uint(a.w);
                const V3 org(ox, oy, oz), dir(dx, dy, dz);
                const bool anyHit = (slot & kShadowBit) != 0;
                if (COUNT) { cnt.tris[0] += anyHit ? 0u : 1u; cnt.tris[1] += anyHit ? 1u : 0u; }
                // Moller-Trumbore exactly as TriangleMesh.cpp:139-160
                const V3 p = cross(dir, e2);
                const float det = dot(e1, p);
                bool accept = det != 0.0f;
                const float invDet = 1.0f / det;
                const V3 dd = org - v0;
                const float b1 = dot(dd, p) * invDet;
                accept = accept && !(b1 < 0.0f || b1 > 1.0f);
                const V3 q = cross(dd, e1);
                const float b2 = dot(dir, q) * invDet;
                accept = accept && !(b2 < 0.0f || b1 + b2 > 1.0f);
                const float tt = dot(e2, q) * invDet;
                accept = accept && !(tt < tmin || tt > tmax);
                // alpha texture of the triangle (TriangleMesh.cpp:162-167); LeafTri::alpha rides in e1's fourth word
                if (accept && __float_as_uint(b.w) != kNoAlpha) accept = alphaPasses(sc.alphaTris, sc.textures, __float_as_uint(b.w), b1, b2);
                if (accept) {
                    if (anyHit) {
                        hitTri = triIdx;
                        finished = true;
                    }
                    else if (!(tt == tmax && hitTri != 0xFFFFFFFFu && (INST ? (inst < hitInst || (inst == hitInst && triIdx < hitTri)) : triIdx < hitTri))) {
                        // equal distance: the larger scene index wins — the larger (instance, triangle) pair in instanced scenes
                        // (tree-independent tie rule)
                        tmax = tt;                                  // ray.distMax = isect->dist (QBVH.h:335)
                        hitTri = triIdx;
                        if (INST) hitInst = inst;
                        hitT = tt;
                        hitB1 = b1;                                 // Intersection::u = 1 - b1 - b2, ::v = b1 (TriangleMesh.cpp:159,172-173)
                        hitB2 = b2;
                    }
                }
                if (!finished) {
                    if (count > 1) cur = kLeafFlag | ((count - 1) << kLeafCountShift) | (first + 1);
                    else if (sp == 0) finished = true;
                    else {
                        --sp;
                        if (kNoSpill || sp < kWsLdsStack) cur = stack[sp * 64];
                        else cur = spill[sp - kWsLdsStack];
                    }
                }
            }
            if (finished) {
                if (slot & kShadowBit) pb.visible[slot & ~kShadowBit] = hitTri == 0xFFFFFFFFu ? 1u : 0u;      // testVisibility
                else {
                    pb.hit[slot] = make_float4(__uint_as_float(hitTri), hitT, hitB1, hitB2);
                    if (INST) pb.hitInstance[slot] = hitInst;
                }
                slot = kIdle;
            }
        }
    }
    if (COUNT) dbg.cycles = __builtin_readcyclecounter() - tStart;
}

// The producer wave (see the head of this file and k_trace_ws): phase 1 walks the slots for extension rays, phase 2 the shadow
// queue; both append to the ray ring of the workgroup.
template <class LDS>
__device__ __forceinline__ void wsProduce(const PathBuffers& pb, LDS& lds, uint32_t numSlots, uint32_t shardCapacity, uint32_t parity,
                                          uint32_t& extRays, uint32_t& shadowRays, WsDebug& dbg) {
    const uint32_t lane = threadIdx.x;
    uint32_t tailLocal = 0;
    const uint32_t chunk = kSub * 64;
    bool ok = true;
    {
        // The state flags of kAhead chunks are fetched in one round trip, the ray records only for chunks that have a ray:
        // near the end of a render most slots are idle, and a producer that pays one memory round trip per 128 slots
        // whether or not they hold a ray makes an almost empty launch last ~58 us (28 dependent round trips per workgroup).
        // Blocks of 256 slots whose slots have all run out of passes are marked by k_shade (PathBuffers::blockDead): their
        // state is not read at all.  One 64-lane load fetches the marks of this workgroup's next 64 chunks.
        const uint32_t numChunks = (numSlots + chunk - 1) / chunk;
        uint64_t deadMask = 0;
        uint32_t group = 0;
        for (uint32_t c0 = blockIdx.x; ok && c0 < numChunks; c0 += gridDim.x * kAhead, group += kAhead) {
            if ((group & 63u) == 0) {
                const uint64_t cl = (uint64_t)c0 + (uint64_t)lane * gridDim.x;       // this lane's chunk among the next 64
                deadMask = __ballot(cl >= numChunks || pb.blockDead[(cl * chunk) / 256u] != 0u);
            }
            const uint32_t deadBits = (uint32_t)(deadMask >> (group & 63u)) & ((1u << kAhead) - 1u);
            if (deadBits == (1u << kAhead) - 1u) continue;                            // wave-uniform
            uint32_t fl[kAhead][kSub];
#pragma unroll
            for (int a = 0; a < kAhead; ++a)
#pragma unroll
                for (int j = 0; j < kSub; ++j) {
                    const uint64_t s = (uint64_t)(c0 + a * gridDim.x) * chunk + j * 64 + lane;
                    fl[a][j] = (s < numSlots && !((deadBits >> a) & 1u)) ? __builtin_nontemporal_load(&pb.flags[s]) : 0u;
                }
#pragma unroll
            for (int a = 0; a < kAhead; ++a) {
                const uint32_t c = c0 + a * gridDim.x;
                bool live[kSub];
                bool any = false;
#pragma unroll
                for (int j = 0; j < kSub; ++j) {
                    const uint32_t state = fl[a][j] & 7u;
                    live[j] = state == 2u || state == 3u;
                    any = any || __ballot(live[j]) != 0;
                }
                if (!any) continue;       // wave-uniform (also true past the last chunk: those flags were read as 0)
                float4 o[kSub], d[kSub];
#pragma unroll
                for (int j = 0; j < kSub; ++j) {
                    const uint32_t s = c * chunk + j * 64 + lane;
                    o[j] = live[j] ? ntLoad4(&pb.rayOrg[(size_t)s * pb.rayStride]) : make_float4(0, 0, 0, 0);
                    d[j] = live[j] ? ntLoad4(&pb.rayDir[(size_t)s * pb.rayStride]) : make_float4(0, 0, 0, 0);
                }
                ok = wsWaitSpace(lds, tailLocal, chunk, &dbg.producerWaits);
                if (!ok) { if (lane == 0) atomicOr(pb.errorWord, ERR_RING_SPACE); break; }
#pragma unroll
                for (int j = 0; j < kSub; ++j) tailLocal += wsAppend(lds, tailLocal, live[j], c * chunk + j * 64 + lane, o[j], d[j]);
                WS_STORE(&lds.tail, tailLocal, __ATOMIC_RELEASE);
            }
        }
        extRays = tailLocal;
    }
    if (ok) {
        const uint32_t shard = blockIdx.x % kShards;
        const uint32_t n = pb.queueCount[queueCounterIndex(parity, Q_SHADOW, shard)];
        const uint32_t* queue = pb.shadowQueue + (size_t)shard * shardCapacity;
        const uint32_t numChunks = (n + chunk - 1) / chunk;
        for (uint32_t c = blockIdx.x / kShards; c < numChunks; c += gridDim.x / kShards) {
            uint32_t sl[kSub];
            float4 o[kSub], d[kSub];
#pragma unroll
            for (int j = 0; j < kSub; ++j) {
                const uint32_t i = c * chunk + j * 64 + lane;
                sl[j] = i < n ? __builtin_nontemporal_load(&queue[i]) : kIdle;
            }
#pragma unroll
            for (int j = 0; j < kSub; ++j) {
                const bool valid = sl[j] != kIdle;
                o[j] = valid ? ntLoad4(&pb.rayOrg[(size_t)sl[j] * pb.rayStride]) : make_float4(0, 0, 0, 0);
                d[j] = valid ? ntLoad4(&pb.shadowDir[sl[j]]) : make_float4(0, 0, 0, 0);
                o[j].w = kRayEpsilon;
            }
            if (!wsWaitSpace(lds, tailLocal, chunk, &dbg.producerWaits)) { if (lane == 0) atomicOr(pb.errorWord, ERR_RING_SPACE); break; }
#pragma unroll
            for (int j = 0; j < kSub; ++j) tailLocal += wsAppend(lds, tailLocal, sl[j] != kIdle, sl[j] | kShadowBit, o[j], d[j]);
            WS_STORE(&lds.tail, tailLocal, __ATOMIC_RELEASE);
        }
        shadowRays = tailLocal - extRays;
    }
    WS_STORE(&lds.done, 1u, __ATOMIC_RELEASE);
    if (lane != 0) { extRays = 0; shadowRays = 0; }
}

// ONE launch traces both ray kinds of an iteration, so the tail of the extension rays (the last long rays of a few
// waves) is filled by shadow rays instead of idle CUs; measured: ~70 + ~50 us of fixed cost per launch pair before.
//   phase 1, extension rays (closest hit): the producer walks ALL slots of its share (no queue); a slot has a ray in flight
//            iff its state is FIRST_HIT or NEXT_HIT (flag values 2 and 3, pt_shade_kernels.h).
//   phase 2, shadow rays: Scene::testVisibility (SurfaceObject.cpp:418-430) = "no hit in [eps, d(1-eps)]"; workgroup b
//            serves queue region b % kShards (gridDim is a multiple of kShards).
template <bool COUNT, int NC, bool QUANT, bool WIDE8 = false, bool INST = false>
__global__ __launch_bounds__(64 * (NC + 1)) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_trace_ws(DevScene sc, PathBuffers pb, uint32_t numSlots, uint32_t shardCapacity, uint32_t parity, uint32_t refill,
                                                                                                              uint32_t tailSlots) {
    __shared__ WsLds<NC> lds;
    const uint32_t live = liveSlots(pb, numSlots);                                   // as of the end of the k_shade launch before this one (uniform)
    if (blockIdx.x == 0 && threadIdx.x == 0) pb.activeSlots[0] = live;               // for the next k_shade launch and the host
    if (live == 0) return;                         // every slot is idle: nothing to trace
    if (tailModeBegins(pb, live, tailSlots, parity)) return;      // the last paths go to the tail kernel (uniform over the grid, pt_kernels.h)
    if (blockIdx.x == 0 && threadIdx.x < Q_KINDS * kShards) {
        // clear the counter set the logic kernel of this iteration fills
        pb.queueCount[queueCounterIndex(parity ^ 1, threadIdx.x / kShards, threadIdx.x % kShards)] = 0;
    }
    if (threadIdx.x == 0) { lds.tail = 0; lds.reserved = 0; lds.released = 0; lds.done = 0; }
    const uint32_t numTop = kTop ? min(kTop, sc.numNodes) : 0u;
    if (kTop) {
        if (QUANT) for (uint32_t i = threadIdx.x; i < numTop * 4u; i += blockDim.x) lds.top[(i >> 2) * 5u + (i & 3u)] = sc.nodesQ[i];
        else for (uint32_t i = threadIdx.x; i < numTop * 8u; i += blockDim.x) lds.top[(i >> 3) * 9u + (i & 7u)] = sc.nodes[i];
    }
    __syncthreads();
    uint32_t extRays = 0, shadowRays = 0;
    WsCounts cnt;
    WsDebug dbg;
    if (threadIdx.x < 64) wsProduce(pb, lds, numSlots, shardCapacity, parity, extRays, shadowRays, dbg);
    else {
        wsConsume<COUNT, NC, QUANT, WIDE8, INST>(sc, pb, lds, refill, numTop, cnt, dbg);
    }
    wsBlockAdd(pb.totals, T_EXT_RAYS, extRays, lds.red);
    wsBlockAdd(pb.totals, T_SHADOW_RAYS, shadowRays, lds.red);
    if (COUNT) {
        wsBlockAdd(pb.totals, T_NODES_CLOSEST, cnt.nodes[0], lds.red); wsBlockAdd(pb.totals, T_TRIS_CLOSEST, cnt.tris[0], lds.red);
        wsBlockAdd(pb.totals, T_NODES_SHADOW, cnt.nodes[1], lds.red); wsBlockAdd(pb.totals, T_TRIS_SHADOW, cnt.tris[1], lds.red);
        const bool l0 = (threadIdx.x & 63u) == 0;      // wave-level figures: lane 0 of each wave speaks; cycles in units of 64
        wsBlockAdd(pb.totals, T_WS_STEPS, l0 ? dbg.steps : 0u, lds.red);
        wsBlockAdd(pb.totals, T_WS_IDLE_SPINS, l0 ? dbg.idleSpins : 0u, lds.red);
        wsBlockAdd(pb.totals, T_WS_CYCLES, l0 ? (uint32_t)(dbg.cycles >> 6) : 0u, lds.red);
        wsBlockAdd(pb.totals, T_WS_IDLE_CYCLES, l0 ? (uint32_t)(dbg.idleCycles >> 6) : 0u, lds.red);
        wsBlockAdd(pb.totals, T_WS_REFILLS, l0 ? dbg.refills : 0u, lds.red);
        wsBlockAdd(pb.totals, T_WS_PRODUCER_WAITS, l0 ? dbg.producerWaits : 0u, lds.red);
        wsBlockAdd(pb.totals, T_WS_NODE_BLOCKS, l0 ? dbg.nodeBlocks : 0u, lds.red);
        wsBlockAdd(pb.totals, T_WS_TRI_BLOCKS, dbg.triBlocks, lds.red);
        wsBlockAdd(pb.totals, T_WS_ACTIVE_LANES, l0 ? dbg.activeLanes : 0u, lds.red);
    }
}


template <bool COUNT, int NC>
static void launchTraceWsT(const DevScene& sc, const PathBuffers& pb, const RenderParams& rp, uint32_t parity, uint32_t blocks, hipStream_t stream) {
    const dim3 grid(blocks), block(64 * (NC + 1));
#ifdef SLR_TUNING_KNOBS
    // the eight-wide quantized tree (measurement, DESIGN.md: -2 % on the headline, +3..6 % slower elsewhere): instantiated in variant builds only
    if (sc.nodes8) { hipLaunchKernelGGL((k_trace_ws<COUNT, NC, false, true>), grid, block, 0, stream, sc, pb, rp.numSlots, rp.shardCapacity, parity, g_refill, rp.tailSlots); return; }
#endif
    // instanced scenes: float nodes only (slrhip_upload_scene does not quantize them)
    if (sc.instances) { hipLaunchKernelGGL((k_trace_ws<COUNT, NC, false, false, true>), grid, block, 0, stream, sc, pb, rp.numSlots, rp.shardCapacity, parity, g_refill, rp.tailSlots); return; }
    if (sc.nodesQ) hipLaunchKernelGGL((k_trace_ws<COUNT, NC, true>), grid, block, 0, stream, sc, pb, rp.numSlots, rp.shardCapacity, parity, g_refill, rp.tailSlots);
    else hipLaunchKernelGGL((k_trace_ws<COUNT, NC, false>), grid, block, 0, stream, sc, pb, rp.numSlots, rp.shardCapacity, parity, g_refill, rp.tailSlots);
}
void launchTraceWs(const DevScene& sc, const PathBuffers& pb, const RenderParams& rp, uint32_t parity, uint32_t blocks, bool count,
                   hipStream_t stream) {
    blocks = (blocks + kShards - 1) / kShards * kShards;
    const int nc = wsConsumers(sc.nodesQ != nullptr);
    if (nc == 15) {
        if (count) launchTraceWsT<true, 15>(sc, pb, rp, parity, blocks, stream);
        else launchTraceWsT<false, 15>(sc, pb, rp, parity, blocks, stream);
    }
    else if (nc == 7) {
        if (count) launchTraceWsT<true, 7>(sc, pb, rp, parity, blocks, stream);
        else launchTraceWsT<false, 7>(sc, pb, rp, parity, blocks, stream);
    }
    else {
        if (count) launchTraceWsT<true, 3>(sc, pb, rp, parity, blocks, stream);
        else launchTraceWsT<false, 3>(sc, pb, rp, parity, blocks, stream);
    }
}

} // namespace slrhip
