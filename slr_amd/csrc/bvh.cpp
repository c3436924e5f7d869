// bvh.cpp — host-side accelerator build for the HIP path tracer.
//
// Role in the reference: SurfaceObjectAggregate's constructor builds the accelerator
// (libSLR/Core/SurfaceObject.cpp:226-230; SBVH by default, QBVH = collapse of it,
// libSLR/Accelerator/QBVH.h:85-202).  The rendered image does not depend on the tree
// (closest-hit semantics only), so the tree here is built for the GPU:
// a binned-SAH binary BVH (16 bins, all three axes) collapsed to 4-wide nodes by repeatedly
// opening the child with the largest surface area, emitted in breadth-first order so that the
// first K nodes are the top of the tree (the part the traversal kernel stages in LDS).
#include "bvh.h"

#include <sched.h>
#include <stdio.h>
#include <stdlib.h>

#include <chrono>
#include <string>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <mutex>
#include <queue>
#include <thread>

namespace slrhip {
unsigned hostThreads() {
    unsigned nt = std::thread::hardware_concurrency();
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) nt = (unsigned)CPU_COUNT(&set);
    if (nt < 1) nt = 1;
    return nt > 32 ? 32 : nt;
}

namespace {

struct Builder {
    const std::vector<Box>& primBox;
    std::vector<float> cen;          // centroids, 3 per primitive
    std::vector<uint32_t> prims;
    std::vector<BNode> nodes;

    uint32_t firstAlone = 0xFFFFFFFFu;      // primitives from this index on (instances) are never packed with others in a leaf
    bool hasAlone(uint32_t begin, uint32_t end) const {
        if (firstAlone >= primBox.size()) return false;
        for (uint32_t k = begin; k < end; ++k) if (prims[k] >= firstAlone) return true;
        return false;
    }

    explicit Builder(const std::vector<Box>& pb) : primBox(pb) {
        size_t n = pb.size();
        cen.resize(3 * n);
        prims.resize(n);
        for (size_t i = 0; i < n; ++i) {
            prims[i] = (uint32_t)i;
            for (int a = 0; a < 3; ++a) cen[3 * i + a] = 0.5f * (pb[i].lo[a] + pb[i].hi[a]);
        }
    }

    // Subtrees are independent once their primitive range is fixed (std::partition works inside [begin, end)), so
    // jobs above kParallelGrain primitives go to a shared queue served by all host threads; smaller ones stay on the
    // worker's own stack.  Nodes come from one pre-sized array through an atomic cursor.
    std::atomic<uint32_t> nodeCursor{0};
    struct Job { uint32_t node, begin, end; };
    std::mutex queueMutex;
    std::condition_variable queueCv;
    std::deque<Job> shared;
    uint32_t busy = 0;
    static const uint32_t kParallelGrain = 1u << 14;

    void build() {
        nodes.resize(2 * prims.size());
        nodeCursor = 1;
        shared.push_back({0, 0, (uint32_t)prims.size()});
        unsigned nt = hostThreads();
        if (prims.size() < 4 * kParallelGrain) nt = 1;
        std::vector<std::thread> pool;
        for (unsigned i = 1; i < nt; ++i) pool.emplace_back([this] { worker(); });
        worker();
        for (std::thread& th : pool) th.join();
        nodes.resize(nodeCursor.load());
    }

    void worker() {
        for (;;) {
            Job j;
            {
                std::unique_lock<std::mutex> lock(queueMutex);
                queueCv.wait(lock, [this] { return !shared.empty() || busy == 0; });
                if (shared.empty()) { queueCv.notify_all(); return; }
                j = shared.front();
                shared.pop_front();
                ++busy;
            }
            run(j);
            {
                std::lock_guard<std::mutex> lock(queueMutex);
                --busy;
            }
            queueCv.notify_all();
        }
    }

    void run(Job first) {
        std::vector<Job> stack;
        stack.push_back(first);
        const int kBins = 16;
        // development knobs (tree-quality experiments): largest leaf and the leaf-versus-split bias
        static const uint32_t maxLeaf = [] { const char* e = tuningEnv("SLRHIP_BVH_MAXLEAF"); int v = e ? atoi(e) : (int)kMaxLeafTris; return (uint32_t)std::min((int)kMaxLeafTris, std::max(1, v)); }();
        static const float leafBias = [] { const char* e = tuningEnv("SLRHIP_BVH_LEAFBIAS"); return e ? (float)atof(e) : 0.125f; }();
        while (!stack.empty()) {
            Job j = stack.back();
            stack.pop_back();
            if (j.end - j.begin > kParallelGrain && !(j.node == first.node)) {
                { std::lock_guard<std::mutex> lock(queueMutex); shared.push_back(j); }
                queueCv.notify_one();
                continue;
            }
            Box box, cbox;
            box.reset(); cbox.reset();
            for (uint32_t k = j.begin; k < j.end; ++k) {
                box.grow(primBox[prims[k]]);
                cbox.grow(&cen[3 * (size_t)prims[k]]);
            }
            uint32_t n = j.end - j.begin;
            BNode nd;
            nd.box = box;
            nd.left = nd.right = 0;
            nd.first = j.begin;
            nd.count = n;
            if (n <= 1) { nodes[j.node] = nd; continue; }       // (a set of <= maxLeaf references may still be split: SAH decides below)
            // binned SAH over the three axes
            float bestCost = INFINITY;
            int bestAxis = -1, bestSplit = 0;
            for (int a = 0; a < 3; ++a) {
                float ext = cbox.hi[a] - cbox.lo[a];
                if (!(ext > 0.0f)) continue;
                Box bb[kBins];
                uint32_t bc[kBins];
                for (int b = 0; b < kBins; ++b) { bb[b].reset(); bc[b] = 0; }
                float scale = kBins / ext;
                for (uint32_t k = j.begin; k < j.end; ++k) {
                    uint32_t p = prims[k];
                    int b = std::min(kBins - 1, std::max(0, (int)((cen[3 * (size_t)p + a] - cbox.lo[a]) * scale)));
                    bb[b].grow(primBox[p]);
                    ++bc[b];
                }
                float rightArea[kBins];
                uint32_t rightCount[kBins];
                Box acc; acc.reset();
                uint32_t cnt = 0;
                for (int b = kBins - 1; b > 0; --b) {
                    acc.grow(bb[b]); cnt += bc[b];
                    rightArea[b] = acc.area(); rightCount[b] = cnt;
                }
                acc.reset(); cnt = 0;
                for (int b = 1; b < kBins; ++b) {
                    acc.grow(bb[b - 1]); cnt += bc[b - 1];
                    if (cnt == 0 || rightCount[b] == 0) continue;
                    float cost = acc.area() * cnt + rightArea[b] * rightCount[b];
                    if (cost < bestCost) { bestCost = cost; bestAxis = a; bestSplit = b; }
                }
            }
            uint32_t mid;
            if (bestAxis >= 0) {
                float leafCost = box.area() * n;
                if (n <= maxLeaf && bestCost + leafBias * box.area() >= leafCost && !hasAlone(j.begin, j.end)) { nodes[j.node] = nd; continue; }
                float ext = cbox.hi[bestAxis] - cbox.lo[bestAxis];
                float scale = kBins / ext;
                float lo = cbox.lo[bestAxis];
                int axis = bestAxis, split = bestSplit;
                auto it = std::partition(prims.begin() + j.begin, prims.begin() + j.end, [&](uint32_t p) {
                    int b = std::min(kBins - 1, std::max(0, (int)((cen[3 * (size_t)p + axis] - lo) * scale)));
                    return b < split;
                });
                mid = (uint32_t)(it - prims.begin());
            }
            else {
                if (n <= maxLeaf && !hasAlone(j.begin, j.end)) { nodes[j.node] = nd; continue; }
                mid = j.begin + n / 2;        // identical centroids: split the list
            }
            if (mid == j.begin || mid == j.end) mid = j.begin + n / 2;
            nd.count = 0;
            nd.left = nodeCursor.fetch_add(2);
            nd.right = nd.left + 1;
            nodes[j.node] = nd;
            stack.push_back({nd.right, mid, j.end});
            stack.push_back({nd.left, j.begin, mid});
        }
    }
};

} // namespace

namespace {
// what the device computes for a plane: fma(q, scale, origin), correctly rounded (double holds the exact sum)
inline float dequant(uint32_t q, float scale, float origin) { return (float)((double)q * (double)scale + (double)origin); }
}

void quantizeNodes(QBVH* out) {
    const size_t n = out->nodes.size();
    out->quantized.resize(n);
    auto work = [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; ++i) {
            const QNode& s = out->nodes[i];
            QNodeQ d;
            std::memset(&d, 0, sizeof(d));
            const float* mins[3] = {s.minx, s.miny, s.minz};
            const float* maxs[3] = {s.maxx, s.maxy, s.maxz};
            float org[3], scl[3];
            uint32_t qlo[3] = {0, 0, 0}, qhi[3] = {0, 0, 0};
            for (int a = 0; a < 3; ++a) {
                float bmin = INFINITY, bmax = -INFINITY;
                for (int c = 0; c < 4; ++c)
                    if (s.child[c] != kInvalidChild) { bmin = std::fmin(bmin, mins[a][c]); bmax = std::fmax(bmax, maxs[a][c]); }
                if (!(bmin <= bmax)) { bmin = 0.0f; bmax = 0.0f; }
                org[a] = bmin;
                float sc = (bmax - bmin) / 255.0f;
                // origin + 255 * scale must reach the top of the box under the device's rounding
                while (sc > 0.0f && dequant(255, sc, bmin) < bmax) sc = std::nextafter(sc, INFINITY);
                scl[a] = sc;
                for (int c = 0; c < 4; ++c) {
                    uint32_t l = 255, h = 0;                       // empty slot: inverted, never entered
                    if (s.child[c] != kInvalidChild) {
                        if (sc > 0.0f) {
                            l = (uint32_t)std::fmin(255.0f, std::fmax(0.0f, std::floor((mins[a][c] - bmin) / sc)));
                            h = (uint32_t)std::fmin(255.0f, std::fmax(0.0f, std::ceil((maxs[a][c] - bmin) / sc)));
                            while (l > 0 && dequant(l, sc, bmin) > mins[a][c]) --l;
                            while (h < 255 && dequant(h, sc, bmin) < maxs[a][c]) ++h;
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
            for (int c = 0; c < 4; ++c) d.child[c] = s.child[c];
            out->quantized[i] = d;
        }
    };
    unsigned nt = n < 65536 ? 1u : hostThreads();
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < nt; ++t) pool.emplace_back(work, n * t / nt, n * (t + 1) / nt);
    work(0, n / nt);
    for (std::thread& th : pool) th.join();
}

namespace {
// The build over a list of primitives: triangles (triIds, or 0 .. numTriPrims - 1 when null) followed by instBoxes.size() instances,
// each of which ends up alone in a leaf whose child reference is kLeafFlag | instance index (count 0, device_types.h).
int buildQBVHPrims(const slrhip_vertex* verts, const slrhip_triangle* tris, const uint32_t* triIds, uint32_t numTriPrims, const std::vector<Box>& instBoxes,
                   QBVH* out, bool spatialSplits, bool wide8);
}

int buildQBVH(const slrhip_vertex* verts, const slrhip_triangle* tris, uint32_t numTris, QBVH* out, bool spatialSplits, bool wide8) {
    if (!verts || !tris || numTris == 0 || numTris > kLeafIndexMask) return 1;
    return buildQBVHPrims(verts, tris, nullptr, numTris, std::vector<Box>(), out, spatialSplits, wide8);
}

namespace {
int buildQBVHPrims(const slrhip_vertex* verts, const slrhip_triangle* tris, const uint32_t* triIds, uint32_t numTriPrims, const std::vector<Box>& instBoxes,
                   QBVH* out, bool spatialSplits, bool wide8) {
    const uint32_t numTris = numTriPrims + (uint32_t)instBoxes.size();       // primitives
    if (numTris == 0) return 1;
    if (!instBoxes.empty() || triIds) { spatialSplits = false; wide8 = false; }
    const auto triOf = [&](uint32_t prim) -> uint32_t { return triIds ? triIds[prim] : prim; };
    std::vector<Box> primBox(numTris);
    for (uint32_t i = 0; i < numTriPrims; ++i) {
        primBox[i].reset();
        for (int k = 0; k < 3; ++k) primBox[i].grow(verts[tris[triOf(i)].v[k]].position);
    }
    for (size_t i = 0; i < instBoxes.size(); ++i) primBox[numTriPrims + i] = instBoxes[i];
    auto tA = std::chrono::steady_clock::now();
    // Binary tree: binned SAH over object partitions (default), or with spatial splits (sbvh.cpp; SLRHIP_BVH=sbvh, reference budget
    // SLRHIP_SBVH_BUDGET x the triangle count, default 1.3).  Both feed the same 4-wide collapse below.
    struct Binary { std::vector<BNode> nodes; std::vector<uint32_t> prims; } b;
    static const bool envSbvh = [] { const char* e = getenv("SLRHIP_BVH"); return e && std::string(e) == "sbvh"; }();
    const bool useSbvh = (spatialSplits || envSbvh) && instBoxes.empty() && !triIds;
    auto tB = tA;
    if (useSbvh) {
        const char* e = tuningEnv("SLRHIP_SBVH_BUDGET");
        SbvhStats st;
        buildBinarySBVH(verts, tris, numTris, e ? (float)atof(e) : 1.3f, &b.nodes, &b.prims, &st);
        out->spatialSplits = st.spatialSplits;
        out->references = st.references;
        if (getenv("SLRHIP_BVH_TIMING"))
            fprintf(stderr, "sbvh: %llu spatial + %llu object splits, %llu references for %u triangles, binary depth %u\n",
                    (unsigned long long)st.spatialSplits, (unsigned long long)st.objectSplits, (unsigned long long)st.references, numTris, st.depth);
    }
    else {
        Builder sah(primBox);
        if (!instBoxes.empty()) sah.firstAlone = numTriPrims;
        tB = std::chrono::steady_clock::now();
        sah.build();
        b.nodes.swap(sah.nodes);
        b.prims.swap(sah.prims);
    }
    auto tC = std::chrono::steady_clock::now();
    if (getenv("SLRHIP_BVH_TIMING")) fprintf(stderr, "bvh: boxes+centroids %.2f s, binary build %.2f s\n",
        std::chrono::duration<double>(tB - tA).count(), std::chrono::duration<double>(tC - tB).count());

    out->nodes.clear();
    out->leafTris.clear();
    out->depth = 0;

    // Pass 1 (serial, touches only the binary nodes): collapse to 4-wide nodes in breadth-first order, number the
    // nodes and the leaf packets.  Pass 2 (all host threads): fill the child boxes and the leaf triangles.
    struct Kids { uint32_t b[4]; };
    std::vector<Kids> kidsOf;
    std::vector<uint32_t> depthOf;
    uint32_t leafCursor = 0;
    std::vector<uint32_t> leafRefOf(wide8 ? b.nodes.size() : 0, 0u);      // the leaf packet of every binary leaf, for the eight-wide tree
    auto leafRef = [&](const BNode& leaf) -> uint32_t {
        if (leaf.count == 1 && b.prims[leaf.first] >= numTriPrims) return kLeafFlag | (b.prims[leaf.first] - numTriPrims);      // an instance
        uint32_t first = leafCursor;
        leafCursor += leaf.count;
        const uint32_t ref = kLeafFlag | (leaf.count << kLeafCountShift) | first;
        if (wide8) leafRefOf[&leaf - b.nodes.data()] = ref;
        return ref;
    };
    const BNode& root = b.nodes[0];
    if (root.count > 0) {
        // a single leaf: wrap it in one node
        QNode qn;
        std::memset(&qn, 0, sizeof(qn));
        qn.child[0] = leafRef(root);
        qn.child[1] = qn.child[2] = qn.child[3] = kInvalidChild;
        out->nodes.push_back(qn);
        kidsOf.push_back({{0, 0, 0, 0}});
        out->depth = 1;
    }
    else {
        out->nodes.reserve(b.nodes.size() / 2 + 1);
        kidsOf.reserve(b.nodes.size() / 2 + 1);
        out->nodes.push_back(QNode());
        kidsOf.push_back({{0, 0, 0, 0}});
        depthOf.push_back(1);
        std::vector<uint32_t> bnodeOf(1, 0u);           // binary node each 4-wide node was made from
        bnodeOf.reserve(b.nodes.size() / 2 + 1);
        depthOf.reserve(b.nodes.size() / 2 + 1);
        for (size_t qi = 0; qi < out->nodes.size(); ++qi) {     // the vector is the BFS queue
            const BNode& bn = b.nodes[bnodeOf[qi]];
            out->depth = std::max(out->depth, depthOf[qi]);
            uint32_t kids[4] = {bn.left, bn.right, 0, 0};
            int nk = 2;
            while (nk < 4) {
                int best = -1;
                float bestArea = -1.0f;
                for (int i = 0; i < nk; ++i) {
                    const BNode& c = b.nodes[kids[i]];
                    if (c.count > 0) continue;
                    float a = c.box.area();
                    if (a > bestArea) { bestArea = a; best = i; }
                }
                if (best < 0) break;
                const BNode& c = b.nodes[kids[best]];
                kids[best] = c.left;
                kids[nk++] = c.right;
            }
            QNode qn;
            std::memset(&qn, 0, sizeof(qn));
            Kids kd = {{0, 0, 0, 0}};
            for (int c = 0; c < 4; ++c) {
                if (c >= nk) { qn.child[c] = kInvalidChild; continue; }
                const BNode& cn = b.nodes[kids[c]];
                kd.b[c] = kids[c];
                if (cn.count > 0) qn.child[c] = leafRef(cn);
                else {
                    qn.child[c] = (uint32_t)out->nodes.size();
                    out->nodes.push_back(QNode());
                    kidsOf.push_back({{0, 0, 0, 0}});
                    bnodeOf.push_back(kids[c]);
                    depthOf.push_back(depthOf[qi] + 1);
                }
            }
            // children were appended after this node: re-index, the vector may have grown
            std::memcpy(out->nodes[qi].child, qn.child, sizeof(qn.child));
            kidsOf[qi] = kd;
        }
    }
    auto tD = std::chrono::steady_clock::now();
    out->leafTris.resize(leafCursor);

    auto fill = [&](size_t lo, size_t hi) {
        for (size_t qi = lo; qi < hi; ++qi) {
            QNode& qn = out->nodes[qi];
            for (int c = 0; c < 4; ++c) {
                if (qn.child[c] == kInvalidChild) {
                    // empty slot: an inverted box never passes the slab test
                    qn.minx[c] = qn.miny[c] = qn.minz[c] = INFINITY;
                    qn.maxx[c] = qn.maxy[c] = qn.maxz[c] = -INFINITY;
                    continue;
                }
                const BNode& cn = b.nodes[kidsOf[qi].b[c]];
                qn.minx[c] = cn.box.lo[0]; qn.miny[c] = cn.box.lo[1]; qn.minz[c] = cn.box.lo[2];
                qn.maxx[c] = cn.box.hi[0]; qn.maxy[c] = cn.box.hi[1]; qn.maxz[c] = cn.box.hi[2];
                if (!(qn.child[c] & kLeafFlag) || ((qn.child[c] >> kLeafCountShift) & 0xFu) == 0u) continue;
                uint32_t first = qn.child[c] & kLeafIndexMask;
                for (uint32_t k = 0; k < cn.count; ++k) {
                    uint32_t t = triOf(b.prims[cn.first + k]);
                    const float* p0 = verts[tris[t].v[0]].position;
                    const float* p1 = verts[tris[t].v[1]].position;
                    const float* p2 = verts[tris[t].v[2]].position;
                    LeafTri lt;
                    std::memset(&lt, 0, sizeof(lt));
                    for (int a = 0; a < 3; ++a) {
                        lt.v0[a] = p0[a];
                        lt.e1[a] = p1[a] - p0[a];      // edge01, TriangleMesh.cpp:136
                        lt.e2[a] = p2[a] - p0[a];      // edge02, TriangleMesh.cpp:137
                    }
                    lt.tri = t;
                    lt.alpha = kNoAlpha;               // set by slrhip_upload_scene for triangles whose material has an alpha texture
                    out->leafTris[first + k] = lt;
                }
            }
        }
    };
    {
        const size_t nq = out->nodes.size();
        unsigned nt = hostThreads();
        if (nq < 65536) nt = 1;
        std::vector<std::thread> pool;
        for (unsigned i = 1; i < nt; ++i) pool.emplace_back(fill, nq * i / nt, nq * (i + 1) / nt);
        fill(0, nq / nt);
        for (std::thread& th : pool) th.join();
    }
    if (getenv("SLRHIP_BVH_TIMING")) fprintf(stderr, "bvh: collapse %.2f s, emit %.2f s\n", std::chrono::duration<double>(tD - tC).count(),
        std::chrono::duration<double>(std::chrono::steady_clock::now() - tD).count());

    // ---- the same binary tree as EIGHT-wide quantized nodes (QNode8), breadth-first, over the same leaf packets ------------------
    out->nodes8.clear();
    out->depth8 = 0;
    if (wide8 && root.count == 0) {
        struct Pending { uint32_t bnode, depth; };
        std::vector<Pending> queue(1, Pending{0u, 1u});
        out->nodes8.reserve(b.nodes.size() / 4 + 1);
        for (size_t qi = 0; qi < queue.size(); ++qi) {
            const BNode& bn = b.nodes[queue[qi].bnode];
            out->depth8 = std::max(out->depth8, queue[qi].depth);
            uint32_t kids[8] = {bn.left, bn.right, 0, 0, 0, 0, 0, 0};
            int nk = 2;
            while (nk < 8) {          // open the inner child with the largest surface area, as the four-wide collapse does
                int best = -1;
                float bestArea = -1.0f;
                for (int i = 0; i < nk; ++i) {
                    const BNode& c = b.nodes[kids[i]];
                    if (c.count > 0) continue;
                    const float a = c.box.area();
                    if (a > bestArea) { bestArea = a; best = i; }
                }
                if (best < 0) break;
                const BNode& c = b.nodes[kids[best]];
                kids[best] = c.left;
                kids[nk++] = c.right;
            }
            QNode8 q;
            std::memset(&q, 0, sizeof(q));
            float org[3], scl[3];
            for (int a = 0; a < 3; ++a) {
                float bmin = INFINITY, bmax = -INFINITY;
                for (int c = 0; c < nk; ++c) { bmin = std::fmin(bmin, b.nodes[kids[c]].box.lo[a]); bmax = std::fmax(bmax, b.nodes[kids[c]].box.hi[a]); }
                org[a] = bmin;
                float sc = (bmax - bmin) / 255.0f;
                while (sc > 0.0f && dequant(255, sc, bmin) < bmax) sc = std::nextafter(sc, INFINITY);
                scl[a] = sc;
            }
            uint32_t* qlo[3] = {q.qlox, q.qloy, q.qloz};
            uint32_t* qhi[3] = {q.qhix, q.qhiy, q.qhiz};
            for (int c = 0; c < 8; ++c) {
                if (c >= nk) {
                    q.child[c] = kInvalidChild;
                    for (int a = 0; a < 3; ++a) { qlo[a][c >> 2] |= 255u << (8 * (c & 3)); }      // inverted box (hi stays 0): never entered
                    continue;
                }
                const BNode& cn = b.nodes[kids[c]];
                for (int a = 0; a < 3; ++a) {
                    uint32_t l = 0, h = 0;
                    if (scl[a] > 0.0f) {
                        l = (uint32_t)std::fmin(255.0f, std::fmax(0.0f, std::floor((cn.box.lo[a] - org[a]) / scl[a])));
                        h = (uint32_t)std::fmin(255.0f, std::fmax(0.0f, std::ceil((cn.box.hi[a] - org[a]) / scl[a])));
                        while (l > 0 && dequant(l, scl[a], org[a]) > cn.box.lo[a]) --l;
                        while (h < 255 && dequant(h, scl[a], org[a]) < cn.box.hi[a]) ++h;
                    }
                    qlo[a][c >> 2] |= l << (8 * (c & 3));
                    qhi[a][c >> 2] |= h << (8 * (c & 3));
                }
                if (cn.count > 0) q.child[c] = leafRefOf[kids[c]];
                else {
                    q.child[c] = (uint32_t)queue.size();
                    queue.push_back(Pending{kids[c], queue[qi].depth + 1});
                }
            }
            q.ox = org[0]; q.oy = org[1]; q.oz = org[2];
            q.sx = scl[0]; q.sy = scl[1]; q.sz = scl[2];
            out->nodes8.push_back(q);
        }
    }
    return 0;
}
} // namespace

// Matrix4x4 x Point3D (Matrix4x4.h:75-81), column-major m
static void mulPointHost(const float* m, const float* p, float* o) {
    float x = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12] * 1.0f;
    float y = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13] * 1.0f;
    float z = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14] * 1.0f;
    float w = m[3] * p[0] + m[7] * p[1] + m[11] * p[2] + m[15] * 1.0f;
    if (w != 1.0f) { float r = 1.0f / w; x *= r; y *= r; z *= r; }
    o[0] = x; o[1] = y; o[2] = z;
}

int buildInstancedQBVH(const slrhip_vertex* verts, const slrhip_triangle* tris, uint32_t numTris, const slrhip_instance* instances, uint32_t numInstances,
                       QBVH* out, std::vector<DevInstance>* devInstances, std::string* err) {
    if (!verts || !tris || numTris == 0 || numTris > kLeafIndexMask || !instances || numInstances == 0 || numInstances > kLeafIndexMask) { *err = "bad arguments"; return 1; }
    struct Mesh { uint32_t first, count, root, depth; Box box; };
    std::vector<Mesh> meshes;
    std::vector<uint32_t> meshOf(numInstances);
    std::vector<char> instanced(numTris, 0);
    std::vector<QBVH> trees;
    for (uint32_t k = 0; k < numInstances; ++k) {
        const slrhip_instance& in = instances[k];
        if (in.num_triangles == 0 || (uint64_t)in.first_triangle + in.num_triangles > numTris) { *err = "instance names triangles out of range"; return 1; }
        for (int r = 0; r < 2; ++r) {
            const float* m = r ? in.world_to_local : in.local_to_world;
            if (m[3] != 0.0f || m[7] != 0.0f || m[11] != 0.0f || m[15] != 1.0f) { *err = "instance transforms must be affine (bottom row 0 0 0 1)"; return 1; }
            for (int i = 0; i < 16; ++i) if (!std::isfinite(m[i])) { *err = "instance transform is not finite"; return 1; }
        }
        uint32_t m = 0;
        for (; m < meshes.size(); ++m) if (meshes[m].first == in.first_triangle && meshes[m].count == in.num_triangles) break;
        if (m == meshes.size()) {
            for (uint32_t t = 0; t < in.num_triangles; ++t) {
                if (instanced[in.first_triangle + t]) { *err = "instanced triangle ranges must be equal or disjoint"; return 1; }
                instanced[in.first_triangle + t] = 1;
            }
            Mesh mesh;
            mesh.first = in.first_triangle; mesh.count = in.num_triangles; mesh.root = 0; mesh.depth = 0;
            mesh.box.reset();
            std::vector<uint32_t> ids(in.num_triangles);
            for (uint32_t t = 0; t < in.num_triangles; ++t) {
                ids[t] = in.first_triangle + t;
                for (int v = 0; v < 3; ++v) mesh.box.grow(verts[tris[ids[t]].v[v]].position);
            }
            trees.emplace_back();
            if (buildQBVHPrims(verts, tris, ids.data(), in.num_triangles, std::vector<Box>(), &trees.back(), false, false) != 0) { *err = "mesh tree build failed"; return 1; }
            mesh.depth = trees.back().depth;
            meshes.push_back(mesh);
        }
        meshOf[k] = m;
    }
    // bounds() of a TransformedSurfaceObject = StaticTransform x BoundingBox3D (Transform.h:54-65): the box of the eight transformed
    // corners of the mesh's box; a hair of slack because the local-space traversal rounds differently from a world-space box test
    std::vector<Box> instBoxes(numInstances);
    for (uint32_t k = 0; k < numInstances; ++k) {
        const Box& mb = meshes[meshOf[k]].box;
        Box wb; wb.reset();
        for (int c = 0; c < 8; ++c) {
            const float p[3] = {(c & 4) ? mb.hi[0] : mb.lo[0], (c & 2) ? mb.hi[1] : mb.lo[1], (c & 1) ? mb.hi[2] : mb.lo[2]};
            float q[3];
            mulPointHost(instances[k].local_to_world, p, q);
            wb.grow(q);
        }
        for (int a = 0; a < 3; ++a) {
            const float pad = 1e-5f * std::fmax(1.0f, std::fmax(std::fabs(wb.lo[a]), std::fabs(wb.hi[a])));
            wb.lo[a] -= pad; wb.hi[a] += pad;
        }
        instBoxes[k] = wb;
    }
    std::vector<uint32_t> loose;
    for (uint32_t i = 0; i < numTris; ++i) if (!instanced[i]) loose.push_back(i);
    if (buildQBVHPrims(verts, tris, loose.data(), (uint32_t)loose.size(), instBoxes, out, false, false) != 0) { *err = "top-level tree build failed"; return 1; }
    // the mesh trees go behind the top-level tree in the same arrays; their inner child indices and leaf packets are rebased
    uint32_t maxMeshDepth = 0;
    for (size_t m = 0; m < meshes.size(); ++m) {
        const uint32_t nodeBase = (uint32_t)out->nodes.size(), leafBase = (uint32_t)out->leafTris.size();
        if ((uint64_t)leafBase + trees[m].leafTris.size() > kLeafIndexMask) { *err = "too many leaf entries"; return 1; }
        meshes[m].root = nodeBase;
        for (QNode qn : trees[m].nodes) {
            for (int c = 0; c < 4; ++c) {
                if (qn.child[c] == kInvalidChild) continue;
                if (qn.child[c] & kLeafFlag) qn.child[c] = (qn.child[c] & ~kLeafIndexMask) | ((qn.child[c] & kLeafIndexMask) + leafBase);
                else qn.child[c] += nodeBase;
            }
            out->nodes.push_back(qn);
        }
        out->leafTris.insert(out->leafTris.end(), trees[m].leafTris.begin(), trees[m].leafTris.end());
        maxMeshDepth = std::max(maxMeshDepth, meshes[m].depth);
    }
    out->depth += maxMeshDepth + 1;        // stack entries: the top level's, kPopInstance, the mesh's (slrhip_upload_scene checks 3 x depth + 1 <= 64)
    devInstances->resize(numInstances);
    for (uint32_t k = 0; k < numInstances; ++k) {
        DevInstance& di = (*devInstances)[k];
        std::memcpy(di.localToWorld, instances[k].local_to_world, sizeof(di.localToWorld));
        std::memcpy(di.worldToLocal, instances[k].world_to_local, sizeof(di.worldToLocal));
        di.rootNode = meshes[meshOf[k]].root; di.firstTriangle = instances[k].first_triangle; di.numTriangles = instances[k].num_triangles; di.mesh = meshOf[k];
    }
    return 0;
}

} // namespace slrhip
