// bvh_device.hip — the accelerator and the per-triangle records built ON THE GPU (gfx950), for large scenes.
//
// Role in the reference: SurfaceObjectAggregate's constructor builds the accelerator on the host (Core/SurfaceObject.cpp:226-230;
// SBVH::SBVH, Accelerator/SBVH.h:379-407 prints its build time; QBVH = a collapse of it, QBVH.h:85-202, :254-284).  The rendered
// image does not depend on the tree (closest-hit semantics), so for scenes of millions of triangles — where the host
// build of bvh.cpp takes seconds — the tree is built where the triangles are going anyway:
//
//   k_prim_bounds     triangle boxes + centroids, scene bounds (one atomic min / max per workgroup and axis)
//   k_morton          63-bit Morton code of the centroid (21 bits per axis)
//   rocPRIM radix sort (hipcub::DeviceRadixSort) of (code, triangle)
//   k_hierarchy       Karras 2012: the binary radix tree over the sorted codes, one thread per inner node, no atomics
//   k_refit_pass      boxes bottom-up, one launch per tree level from the leaves: a node is computed in the pass after both its
//                     children were (their pass numbers are read, never data written in the same launch — no in-kernel hand-off
//                     between CUs, whose L1s and per-XCD L2s are not coherent with each other)
//   k_collapse_*      the SAME 4-wide collapse as bvh.cpp (open the child with the largest surface area until four children or
//                     only leaf packets are left; a subtree of <= 4 triangles is a leaf packet — contiguous in sorted order),
//                     level by level so that nodes come out in breadth-first order; child slots are numbered by an exclusive
//                     scan over the level, so the layout is deterministic
//   k_emit_nodes      the 128-byte QNode records and, for trees beyond the L2, the 64-byte quantized ones (quantizeNodes of
//                     bvh.cpp: rounded outwards and checked with the traversal kernel's own fma)
//   k_leaf_tris       LeafTri packets in sorted order (v0, e1 = v1 - v0, e2 = v2 - v0: TriangleMesh.cpp:136-137)
//   k_shade_tris      ShadeTri records (Triangle::getSurfacePoint's inputs, TriangleMesh.cpp:180-215) — the same float operations
//                     as the host loop of slrhip_upload_scene, so the records are bit-identical
//
// LBVH trees are of lower quality than the binned-SAH tree (more nodes per ray: + 9 % on the 10 M-triangle grid, + 16 % on the
// Cornell scene): the host build stays the default below 2^20 triangles; above, and on request (SLRHIP_FLAG_BVH_DEVICE_BUILD,
// SLRHIP_BVH=device), this one is used (DESIGN.md has build time and nodes per ray side by side).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "bvh.h"

namespace slrhip {

namespace {

#define DB_TRY(expr)                                                                                              \
    do {                                                                                                          \
        hipError_t e_ = (expr);                                                                                   \
        if (e_ != hipSuccess) { *err = std::string(#expr) + ": " + hipGetErrorString(e_); return 1; }             \
    } while (0)

struct DBox {
    float lo[3], hi[3];
};

// floats ordered as unsigned integers (for atomicMin / atomicMax on scene bounds)
__device__ __forceinline__ uint32_t orderedBits(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ inline float fromOrdered(uint32_t u) {
    const uint32_t b = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u;
    float f;
#ifdef __HIP_DEVICE_COMPILE__
    f = __uint_as_float(b);
#else
    std::memcpy(&f, &b, 4);
#endif
    return f;
}

__global__ void k_prim_bounds(const slrhip_vertex* __restrict__ verts, const slrhip_triangle* __restrict__ tris, uint32_t n, DBox* __restrict__ boxes,
                              uint32_t* __restrict__ sceneBounds /* lo[3], hi[3] ordered bits of CENTROID bounds */) {
    __shared__ uint32_t sLo[3], sHi[3];
    if (threadIdx.x < 3) { sLo[threadIdx.x] = 0xFFFFFFFFu; sHi[threadIdx.x] = 0u; }
    __syncthreads();
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const slrhip_triangle t = tris[i];
        DBox b;
        for (int a = 0; a < 3; ++a) {
            const float p0 = verts[t.v[0]].position[a], p1 = verts[t.v[1]].position[a], p2 = verts[t.v[2]].position[a];
            b.lo[a] = fminf(fminf(p0, p1), p2);
            b.hi[a] = fmaxf(fmaxf(p0, p1), p2);
            const float c = 0.5f * (b.lo[a] + b.hi[a]);
            atomicMin(&sLo[a], orderedBits(c));
            atomicMax(&sHi[a], orderedBits(c));
        }
        boxes[i] = b;
    }
    __syncthreads();
    if (threadIdx.x < 3) { atomicMin(&sceneBounds[threadIdx.x], sLo[threadIdx.x]); atomicMax(&sceneBounds[3 + threadIdx.x], sHi[threadIdx.x]); }
}

__device__ __forceinline__ unsigned long long spread21(uint32_t v) {       // 21 bits -> every third bit of 63
    unsigned long long x = v & 0x1FFFFFull;
    x = (x | (x << 32)) & 0x1F00000000FFFFull;
    x = (x | (x << 16)) & 0x1F0000FF0000FFull;
    x = (x | (x << 8)) & 0x100F00F00F00F00Full;
    x = (x | (x << 4)) & 0x10C30C30C30C30C3ull;
    x = (x | (x << 2)) & 0x1249249249249249ull;
    return x;
}

__global__ void k_morton(const DBox* __restrict__ boxes, uint32_t n, const uint32_t* __restrict__ sceneBounds, unsigned long long* __restrict__ keys,
                         uint32_t* __restrict__ vals) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t q[3];
    for (int a = 0; a < 3; ++a) {
        const float lo = fromOrdered(sceneBounds[a]), hi = fromOrdered(sceneBounds[3 + a]);
        const float c = 0.5f * (boxes[i].lo[a] + boxes[i].hi[a]);
        const float ext = hi - lo;
        const float u = ext > 0.0f ? (c - lo) / ext : 0.0f;
        q[a] = (uint32_t)fminf(fmaxf(u * 2097152.0f, 0.0f), 2097151.0f);
    }
    keys[i] = (spread21(q[0]) << 2) | (spread21(q[1]) << 1) | spread21(q[2]);
    vals[i] = i;
}

// Common-prefix length of sorted keys i and j (Karras 2012, with the index as tie-break for equal codes); -1 outside the range.
__device__ __forceinline__ int delta(const unsigned long long* __restrict__ keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    const unsigned long long a = keys[i], b = keys[j];
    if (a == b) return 64 + __clz((uint32_t)i ^ (uint32_t)j);
    return __clzll((long long)(a ^ b));
}

// Binary nodes: inner nodes 0 .. n-2, leaves n-1 .. 2n-2 (leaf k = sorted position k).  range = sorted positions covered.
struct BinTree {
    uint32_t* left;
    uint32_t* right;
    uint32_t* first;        // inner: first sorted position; last = first + count - 1
    uint32_t* count;
    DBox* box;              // 2n-1 entries
    uint32_t* ready;        // 2n-1 entries: the refit pass that computed the node's box (leaves: 1), 0 = not yet
    uint32_t maxLeaf;       // triangles per leaf packet (<= kMaxLeafTris)
};

__global__ void k_hierarchy(const unsigned long long* __restrict__ keys, int n, BinTree t) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n - 1) return;
    const int d = (delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1)) >= 0 ? 1 : -1;
    const int dmin = delta(keys, n, i, i - d);
    int lmax = 2;
    while (delta(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int s = lmax / 2; s >= 1; s /= 2)
        if (delta(keys, n, i, i + (l + s) * d) > dmin) l += s;
    const int j = i + l * d;
    const int dnode = delta(keys, n, i, j);
    int s = 0;
    int tt = l;
    do {
        tt = (tt + 1) / 2;
        if (delta(keys, n, i, i + (s + tt) * d) > dnode) s += tt;
    } while (tt > 1);
    const int gamma = i + s * d + min(d, 0);
    const int lo = min(i, j), hi = max(i, j);
    const uint32_t l_ = lo == gamma ? (uint32_t)(n - 1 + gamma) : (uint32_t)gamma;
    const uint32_t r_ = hi == gamma + 1 ? (uint32_t)(n - 1 + gamma + 1) : (uint32_t)(gamma + 1);
    t.left[i] = l_;
    t.right[i] = r_;
    t.first[i] = (uint32_t)lo;
    t.count[i] = (uint32_t)(hi - lo + 1);
    t.ready[i] = 0u;
}

__global__ void k_refit_leaves(const DBox* __restrict__ primBoxes, const uint32_t* __restrict__ vals, int n, BinTree t) {
    const int k = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (k >= n) return;
    t.box[n - 1 + k] = primBoxes[vals[k]];
    t.ready[n - 1 + k] = 1u;
}
// pass p >= 2: inner nodes whose children were both computed in EARLIER passes (launches)
__global__ void k_refit_pass(int n, BinTree t, uint32_t pass) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n - 1 || t.ready[i] != 0u) return;
    const uint32_t l = t.left[i], r = t.right[i];
    const uint32_t rl = t.ready[l], rr = t.ready[r];
    if (rl == 0u || rl >= pass || rr == 0u || rr >= pass) return;
    const DBox a = t.box[l], b = t.box[r];
    DBox o;
    for (int ax = 0; ax < 3; ++ax) { o.lo[ax] = fminf(a.lo[ax], b.lo[ax]); o.hi[ax] = fmaxf(a.hi[ax], b.hi[ax]); }
    t.box[i] = o;
    t.ready[i] = pass;
}

__device__ __forceinline__ float boxArea(const DBox& b) {          // Box::area of bvh.h
    const float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
    if (dx < 0 || dy < 0 || dz < 0) return 0.0f;
    return 2.0f * (dx * dy + dy * dz + dz * dx);
}
// a binary node that becomes a leaf packet of the 4-wide tree: a single triangle, or an inner node over <= kMaxLeafTris of them
__device__ __forceinline__ bool isPacket(const BinTree& t, int n, uint32_t node) { return node >= (uint32_t)(n - 1) || t.count[node] <= t.maxLeaf; }

struct Kids {
    uint32_t b[4];
    uint32_t num;
};

// One level of the collapse, pass 1: the (up to) four children of every node of the level, and how many of them are inner nodes.
__global__ void k_collapse_count(BinTree t, int n, const uint32_t* __restrict__ frontier, uint32_t numFrontier, Kids* __restrict__ kidsOut,
                                 uint32_t* __restrict__ innerCount) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= numFrontier) return;
    const uint32_t bn = frontier[i];
    uint32_t kids[4] = {t.left[bn], t.right[bn], 0u, 0u};
    uint32_t nk = 2;
    while (nk < 4) {
        int best = -1;
        float bestArea = -1.0f;
        for (uint32_t c = 0; c < nk; ++c) {
            if (isPacket(t, n, kids[c])) continue;
            const float a = boxArea(t.box[kids[c]]);
            if (a > bestArea) { bestArea = a; best = (int)c; }
        }
        if (best < 0) break;
        const uint32_t c = kids[best];
        kids[best] = t.left[c];
        kids[nk++] = t.right[c];
    }
    Kids k;
    uint32_t inner = 0;
    for (uint32_t c = 0; c < 4; ++c) {
        k.b[c] = c < nk ? kids[c] : 0xFFFFFFFFu;
        if (c < nk && !isPacket(t, n, kids[c])) ++inner;
    }
    k.num = nk;
    kidsOut[i] = k;
    innerCount[i] = inner;
}

// what the traversal kernel computes for a quantized plane
__device__ __forceinline__ float dequantDev(uint32_t q, float scale, float origin) { return __builtin_fmaf((float)q, scale, origin); }

// pass 2: write the level's QNode (and QNodeQ) records; inner children are numbered nextBase + scan[i] + (rank among the inner
// children of this node), and listed in the next frontier in that order
__global__ void k_collapse_emit(BinTree t, int n, uint32_t numFrontier, uint32_t levelBase, uint32_t nextBase, const Kids* __restrict__ kidsIn,
                                const uint32_t* __restrict__ scan, uint32_t* __restrict__ nextFrontier, QNode* __restrict__ nodes, QNodeQ* __restrict__ nodesQ) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= numFrontier) return;
    const Kids k = kidsIn[i];
    QNode qn;
    uint32_t rank = 0;
    for (uint32_t c = 0; c < 4; ++c) {
        qn.pad[c] = 0u;
        if (c >= k.num) {
            // empty slot: an inverted box never passes the slab test
            qn.child[c] = kInvalidChild;
            qn.minx[c] = qn.miny[c] = qn.minz[c] = INFINITY;
            qn.maxx[c] = qn.maxy[c] = qn.maxz[c] = -INFINITY;
            continue;
        }
        const uint32_t bn = k.b[c];
        const DBox b = t.box[bn];
        qn.minx[c] = b.lo[0]; qn.miny[c] = b.lo[1]; qn.minz[c] = b.lo[2];
        qn.maxx[c] = b.hi[0]; qn.maxy[c] = b.hi[1]; qn.maxz[c] = b.hi[2];
        if (isPacket(t, n, bn)) {
            const uint32_t first = bn >= (uint32_t)(n - 1) ? bn - (uint32_t)(n - 1) : t.first[bn];
            const uint32_t cnt = bn >= (uint32_t)(n - 1) ? 1u : t.count[bn];
            qn.child[c] = kLeafFlag | (cnt << kLeafCountShift) | first;      // leaf packets are contiguous in sorted order
        }
        else {
            const uint32_t slot = scan[i] + rank++;
            qn.child[c] = nextBase + slot;
            nextFrontier[slot] = bn;
        }
    }
    nodes[levelBase + i] = qn;
    if (!nodesQ) return;
    // quantizeNodes of bvh.cpp, with the device's own fma as the check (rounded outwards: every dequantized box contains the float box)
    QNodeQ d;
    const float* mins[3] = {qn.minx, qn.miny, qn.minz};
    const float* maxs[3] = {qn.maxx, qn.maxy, qn.maxz};
    float org[3], scl[3];
    uint32_t qlo[3] = {0, 0, 0}, qhi[3] = {0, 0, 0};
    for (int a = 0; a < 3; ++a) {
        float bmin = INFINITY, bmax = -INFINITY;
        for (int c = 0; c < 4; ++c)
            if (qn.child[c] != kInvalidChild) { bmin = fminf(bmin, mins[a][c]); bmax = fmaxf(bmax, maxs[a][c]); }
        if (!(bmin <= bmax)) { bmin = 0.0f; bmax = 0.0f; }
        org[a] = bmin;
        float sc = (bmax - bmin) / 255.0f;
        for (int guard = 0; guard < 64 && sc > 0.0f && dequantDev(255, sc, bmin) < bmax; ++guard) sc = __uint_as_float(__float_as_uint(sc) + 1u);      // nextafter upwards (sc > 0)
        scl[a] = sc;
        for (int c = 0; c < 4; ++c) {
            uint32_t l = 255, h = 0;                       // empty slot: inverted, never entered
            if (qn.child[c] != kInvalidChild) {
                if (sc > 0.0f) {
                    l = (uint32_t)fminf(255.0f, fmaxf(0.0f, floorf((mins[a][c] - bmin) / sc)));
                    h = (uint32_t)fminf(255.0f, fmaxf(0.0f, ceilf((maxs[a][c] - bmin) / sc)));
                    while (l > 0 && dequantDev(l, sc, bmin) > mins[a][c]) --l;
                    while (h < 255 && dequantDev(h, sc, bmin) < maxs[a][c]) ++h;
                }
                else { l = 0; h = 0; }                    // flat box on this axis: origin is the plane, exactly
            }
            qlo[a] |= l << (8 * c);
            qhi[a] |= h << (8 * c);
        }
    }
    d.ox = org[0]; d.oy = org[1]; d.oz = org[2];
    d.sx = scl[0]; d.sy = scl[1]; d.sz = scl[2];
    d.qlox = qlo[0]; d.qloy = qlo[1]; d.qloz = qlo[2];
    d.qhix = qhi[0]; d.qhiy = qhi[1]; d.qhiz = qhi[2];
    for (int c = 0; c < 4; ++c) d.child[c] = qn.child[c];
    nodesQ[levelBase + i] = d;
}

__global__ void k_leaf_tris(const slrhip_vertex* __restrict__ verts, const slrhip_triangle* __restrict__ tris, const uint32_t* __restrict__ vals, uint32_t n,
                            LeafTri* __restrict__ out) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const uint32_t ti = vals[k];
    const slrhip_triangle t = tris[ti];
    LeafTri lt;
    for (int a = 0; a < 3; ++a) {
        const float p0 = verts[t.v[0]].position[a];
        lt.v0[a] = p0;
        lt.e1[a] = verts[t.v[1]].position[a] - p0;      // edge01, TriangleMesh.cpp:136
        lt.e2[a] = verts[t.v[2]].position[a] - p0;      // edge02, TriangleMesh.cpp:137
    }
    lt.tri = ti;
    lt.alpha = kNoAlpha;
    lt.pad1 = 0u;
    out[k] = lt;
}

// the per-triangle loop of slrhip_upload_scene, operation for operation (same float results)
__global__ void k_shade_tris(const slrhip_vertex* __restrict__ verts, const slrhip_triangle* __restrict__ tris, uint32_t n, ShadeTri* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const slrhip_triangle t = tris[i];
    const slrhip_vertex v0 = verts[t.v[0]], v1 = verts[t.v[1]], v2 = verts[t.v[2]];
    ShadeTri s;
    float e1[3], e2[3];
    for (int a = 0; a < 3; ++a) {
        s.n0[a] = v0.normal[a]; s.n1[a] = v1.normal[a]; s.n2[a] = v2.normal[a];
        s.t0[a] = v0.tangent[a]; s.t1[a] = v1.tangent[a]; s.t2[a] = v2.tangent[a];
        e1[a] = v1.position[a] - v0.position[a];
        e2[a] = v2.position[a] - v0.position[a];
    }
    // normalize(cross(edge01, edge02)) TriangleMesh.cpp:171
    const float cx = e1[1] * e2[2] - e1[2] * e2[1], cy = e1[2] * e2[0] - e1[0] * e2[2], cz = e1[0] * e2[1] - e1[1] * e2[0];
    const float len = sqrtf(cx * cx + cy * cy + cz * cz);
    const float r = 1.0f / len;
    s.gnx = cx * r; s.gny = cy * r; s.gnz = cz * r;
    s.areaPDF = 1.0f / (0.5f * len);                    // 1 / Triangle::area() :217-222
    s.material = t.material;
    s.light = -1;
    out[i] = s;
}
__global__ void k_patch_lights(const uint32_t* __restrict__ lightTri, uint32_t numLights, ShadeTri* __restrict__ shade) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < numLights) shade[lightTri[i]].light = (int32_t)i;
}

template <class T>
struct Tmp {
    T* p = nullptr;
    ~Tmp() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t n) { return hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(T)); }
};

} // namespace

int buildGeometryDevice(const slrhip_vertex* hVerts, uint32_t numVerts, const slrhip_triangle* hTris, uint32_t numTris, const uint32_t* lightTris,
                        uint32_t numLights, bool wantQuantized, DeviceGeometry* out, std::string* err) {
    const int n = (int)numTris;
    if (numTris < 8 || numTris > kLeafIndexMask) { *err = "device build: triangle count out of range"; return 1; }
    const auto t0 = std::chrono::steady_clock::now();
    Tmp<slrhip_vertex> verts;
    Tmp<slrhip_triangle> tris;
    DB_TRY(verts.alloc(numVerts));
    DB_TRY(tris.alloc(numTris));
    DB_TRY(hipMemcpy(verts.p, hVerts, (size_t)numVerts * sizeof(slrhip_vertex), hipMemcpyHostToDevice));
    DB_TRY(hipMemcpy(tris.p, hTris, (size_t)numTris * sizeof(slrhip_triangle), hipMemcpyHostToDevice));
    const auto t1 = std::chrono::steady_clock::now();

    const uint32_t B = 256, G = (numTris + B - 1) / B;
    Tmp<DBox> primBoxes;
    Tmp<uint32_t> bounds, vals, valsSorted;
    Tmp<unsigned long long> keys, keysSorted;
    DB_TRY(primBoxes.alloc(numTris)); DB_TRY(bounds.alloc(6)); DB_TRY(vals.alloc(numTris)); DB_TRY(valsSorted.alloc(numTris));
    DB_TRY(keys.alloc(numTris)); DB_TRY(keysSorted.alloc(numTris));
    const uint32_t boundsInit[6] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u};
    DB_TRY(hipMemcpy(bounds.p, boundsInit, sizeof(boundsInit), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_prim_bounds, dim3(G), dim3(B), 0, nullptr, verts.p, tris.p, numTris, primBoxes.p, bounds.p);
    hipLaunchKernelGGL(k_morton, dim3(G), dim3(B), 0, nullptr, primBoxes.p, numTris, bounds.p, keys.p, vals.p);
    {
        size_t tmpBytes = 0;
        DB_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tmpBytes, keys.p, keysSorted.p, vals.p, valsSorted.p, n, 0, 63));
        Tmp<char> tmp;
        DB_TRY(tmp.alloc(tmpBytes));
        DB_TRY(hipcub::DeviceRadixSort::SortPairs(tmp.p, tmpBytes, keys.p, keysSorted.p, vals.p, valsSorted.p, n, 0, 63));
        DB_TRY(hipDeviceSynchronize());
    }
    const auto t2 = std::chrono::steady_clock::now();

    BinTree bt;
    Tmp<uint32_t> left, right, first, count, ready;
    Tmp<DBox> boxes;
    DB_TRY(left.alloc(numTris)); DB_TRY(right.alloc(numTris)); DB_TRY(first.alloc(numTris));
    DB_TRY(count.alloc(numTris)); DB_TRY(ready.alloc(2 * (size_t)numTris)); DB_TRY(boxes.alloc(2 * (size_t)numTris));
    bt.left = left.p; bt.right = right.p; bt.first = first.p; bt.count = count.p; bt.box = boxes.p; bt.ready = ready.p;
    // Leaf packets: runs of up to maxLeaf triangles that are neighbours in Morton order.  Measured on the 10 M-triangle grid
    // (SLRHIP_LBVH_LEAF, profiles/r03_f_*): DESIGN.md has nodes / triangles per ray and the traversal time for each size.
    static const uint32_t envLeaf = [] { const char* e = tuningEnv("SLRHIP_LBVH_LEAF"); const int v = e ? atoi(e) : 0; return (uint32_t)(v >= 1 && v <= (int)kMaxLeafTris ? v : 0); }();
    bt.maxLeaf = envLeaf ? envLeaf : 2u;      // 10 M-triangle grid, traversal us per launch at 1 / 2 / 4: 3 590 / 3 498 / 3 866 (host SAH tree: 3 338)
    hipLaunchKernelGGL(k_hierarchy, dim3(G), dim3(B), 0, nullptr, keysSorted.p, n, bt);
    hipLaunchKernelGGL(k_refit_leaves, dim3(G), dim3(B), 0, nullptr, primBoxes.p, valsSorted.p, n, bt);
    {
        uint32_t rootReady = 0;
        for (uint32_t pass = 2; pass < 512 && !rootReady; ++pass) {
            hipLaunchKernelGGL(k_refit_pass, dim3(G), dim3(B), 0, nullptr, n, bt, pass);
            if ((pass & 3u) == 1u) DB_TRY(hipMemcpy(&rootReady, ready.p, sizeof(uint32_t), hipMemcpyDeviceToHost));
        }
        if (!rootReady) DB_TRY(hipMemcpy(&rootReady, ready.p, sizeof(uint32_t), hipMemcpyDeviceToHost));
        if (!rootReady) { *err = "device build: the binary tree is deeper than 512 levels"; return 1; }
    }
    DB_TRY(hipDeviceSynchronize());
    const auto t3 = std::chrono::steady_clock::now();

    // ---- collapse, level by level ------------------------------------------------------------------------------------------
    const size_t maxNodes = (size_t)numTris + 64;           // every 4-wide node consumes at least one inner node of the binary tree
    QNode* nodes = nullptr;
    QNodeQ* nodesQ = nullptr;
    DB_TRY(hipMalloc(&nodes, maxNodes * sizeof(QNode)));
    if (wantQuantized) {
        hipError_t e = hipMalloc(&nodesQ, maxNodes * sizeof(QNodeQ));
        if (e != hipSuccess) { (void)hipFree(nodes); *err = std::string("hipMalloc nodesQ: ") + hipGetErrorString(e); return 1; }
    }
    auto failFree = [&](const std::string& m) { (void)hipFree(nodes); if (nodesQ) (void)hipFree(nodesQ); *err = m; return 1; };
    Tmp<uint32_t> frontierA, frontierB, innerCount, scan;
    Tmp<Kids> kids;
    Tmp<char> scanTmp;
    size_t scanTmpBytes = 0;
    if (frontierA.alloc(maxNodes) != hipSuccess || frontierB.alloc(maxNodes) != hipSuccess || innerCount.alloc(maxNodes + 1) != hipSuccess ||
        scan.alloc(maxNodes + 1) != hipSuccess || kids.alloc(maxNodes) != hipSuccess)
        return failFree("device build: out of memory for the collapse");
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, scanTmpBytes, innerCount.p, scan.p, (int)maxNodes + 1);
    if (scanTmp.alloc(scanTmpBytes) != hipSuccess) return failFree("device build: out of memory for the scan");
    const uint32_t rootNode = 0u;
    if (hipMemcpy(frontierA.p, &rootNode, sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess) return failFree("device build: copy failed");
    uint32_t numFrontier = 1, levelBase = 0, depth = 0;
    uint32_t* cur = frontierA.p;
    uint32_t* next = frontierB.p;
    while (numFrontier > 0) {
        ++depth;
        if (depth > 64) return failFree("device build: tree deeper than 64 levels");
        if ((size_t)levelBase + numFrontier > maxNodes) return failFree("device build: node budget exceeded");
        const uint32_t g = (numFrontier + B - 1) / B;
        hipLaunchKernelGGL(k_collapse_count, dim3(g), dim3(B), 0, nullptr, bt, n, cur, numFrontier, kids.p, innerCount.p);
        // exclusive scan over numFrontier + 1 entries: entry [numFrontier] of the result is the number of inner children of the level
        if (hipMemsetAsync(innerCount.p + numFrontier, 0, sizeof(uint32_t), nullptr) != hipSuccess) return failFree("device build: memset failed");
        if (hipcub::DeviceScan::ExclusiveSum(scanTmp.p, scanTmpBytes, innerCount.p, scan.p, (int)numFrontier + 1) != hipSuccess)
            return failFree("device build: scan failed");
        uint32_t numNext = 0;
        if (hipMemcpy(&numNext, scan.p + numFrontier, sizeof(uint32_t), hipMemcpyDeviceToHost) != hipSuccess) return failFree("device build: copy failed");
        const uint32_t nextBase = levelBase + numFrontier;
        if ((size_t)nextBase + numNext > maxNodes) return failFree("device build: node budget exceeded");
        hipLaunchKernelGGL(k_collapse_emit, dim3(g), dim3(B), 0, nullptr, bt, n, numFrontier, levelBase, nextBase, kids.p, scan.p, next, nodes, nodesQ);
        levelBase = nextBase;
        numFrontier = numNext;
        std::swap(cur, next);
    }
    const uint32_t numNodes = levelBase;
    const auto t4 = std::chrono::steady_clock::now();

    LeafTri* leafTris = nullptr;
    ShadeTri* shadeTris = nullptr;
    if (hipMalloc(&leafTris, (size_t)numTris * sizeof(LeafTri)) != hipSuccess) return failFree("device build: out of memory for the leaf triangles");
    if (hipMalloc(&shadeTris, (size_t)numTris * sizeof(ShadeTri)) != hipSuccess) { (void)hipFree(leafTris); return failFree("device build: out of memory for the shading records"); }
    hipLaunchKernelGGL(k_leaf_tris, dim3(G), dim3(B), 0, nullptr, verts.p, tris.p, valsSorted.p, numTris, leafTris);
    hipLaunchKernelGGL(k_shade_tris, dim3(G), dim3(B), 0, nullptr, verts.p, tris.p, numTris, shadeTris);
    if (numLights) {
        Tmp<uint32_t> lt;
        if (lt.alloc(numLights) != hipSuccess || hipMemcpy(lt.p, lightTris, (size_t)numLights * sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess) {
            (void)hipFree(leafTris); (void)hipFree(shadeTris);
            return failFree("device build: light list upload failed");
        }
        hipLaunchKernelGGL(k_patch_lights, dim3((numLights + B - 1) / B), dim3(B), 0, nullptr, lt.p, numLights, shadeTris);
        (void)hipDeviceSynchronize();
    }
    const hipError_t fin = hipDeviceSynchronize();
    const hipError_t last = hipGetLastError();
    if (fin != hipSuccess || last != hipSuccess) {
        (void)hipFree(leafTris); (void)hipFree(shadeTris);
        return failFree(std::string("device build: ") + hipGetErrorString(fin != hipSuccess ? fin : last));
    }
    const auto t5 = std::chrono::steady_clock::now();
    out->nodes = nodes; out->nodesQ = nodesQ; out->leafTris = leafTris; out->shadeTris = shadeTris;
    out->numNodes = numNodes; out->numLeafTris = numTris; out->depth = depth;
    auto sec = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    out->secondsUpload = sec(t0, t1); out->secondsSort = sec(t1, t2); out->secondsHierarchy = sec(t2, t3); out->secondsCollapse = sec(t3, t4);
    out->secondsRecords = sec(t4, t5);
    if (getenv("SLRHIP_BVH_TIMING"))
        fprintf(stderr, "device build: upload %.3f s, boxes+morton+sort %.3f s, hierarchy+refit %.3f s, collapse+emit (%u levels, %u nodes) %.3f s, leaf+shading records %.3f s\n",
                out->secondsUpload, out->secondsSort, out->secondsHierarchy, depth, numNodes, out->secondsCollapse, out->secondsRecords);
    return 0;
}

} // namespace slrhip
