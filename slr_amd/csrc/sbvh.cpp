// sbvh.cpp — binned-SAH binary BVH with spatial splits, host side.
//
// What it stands for in the reference: the default accelerator, libSLR/Accelerator/SBVH.h:57-348 (object binning, spatial
// binning with Triangle::choppedBounds, reference duplication with Triangle::splitBounds, Surface/TriangleMesh.cpp:19-125,
// overlap criterion alpha = 1e-5 against the root's surface area).  The rendered image does not depend on the tree (closest-hit
// semantics), so this is the algorithm of Stich, Friedrich and Dietrich (2009) written for this code base's needs:
//   * references = (clipped box, triangle); a node is split either by OBJECT (binned SAH on the reference boxes' centroids, 16 bins,
//     all three axes) or in SPACE (16 bins over the node's box on each axis, every straddling reference clipped into the bins it
//     spans); the spatial candidate is evaluated only if the object split's children overlap by more than alpha of the root area
//     and the reference budget allows duplicates;
//   * a reference that straddles the chosen plane goes to both sides with its box clipped (and may be put back on one side alone
//     if that is cheaper: "reference unsplitting");
//   * leaves hold at most kMaxLeafTris references (the leaf packet size of the traversal kernels);
//   * clipped boxes are padded outwards by a few ulps before they are intersected with the triangle's exact box: the clip
//     points carry rounding error, and a box that is too small by one ulp would lose a hit on the split plane.
// Subtrees above a grain size are built by separate host threads.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <thread>
#include <vector>

#include "bvh.h"

namespace slrhip {
namespace {

struct Ref {
    Box box;
    uint32_t prim;
};

const int kBins = 16;
const float kAlpha = 1e-5f;            // SBVH.h:176
const int kMaxDepth = 48;              // binary levels; the 4-wide collapse roughly halves it (the traversal stack holds 64 entries)

struct Ctx {
    const slrhip_vertex* verts;
    const slrhip_triangle* tris;
    float rootArea;
    std::vector<BNode>* nodes;
    std::atomic<uint32_t> nodeCursor{1};
    std::atomic<int64_t> refsLeft;      // duplicates still allowed
    std::atomic<int> threadsLeft;
    std::atomic<uint64_t> spatialSplits{0}, objectSplits{0};
    std::atomic<uint32_t> depth{0};
    // leaves are collected per thread and concatenated at the end
};

inline float pad(float v, float dir, float scale) {
    // a few ulps outwards, relative to the coordinate and to the scene scale (coordinates near zero)
    return v + dir * (std::fabs(v) * 4e-7f + scale * 1e-7f);
}

// Bounds of triangle `t` clipped to lo <= x[axis] <= hi (Sutherland-Hodgman against the two planes), padded, then
// intersected with `within` (the reference's current box).  Returns false if nothing is left.
bool clippedBounds(const Ctx& c, uint32_t t, int axis, float lo, float hi, const Box& within, Box* out) {
    float poly[8][3], tmp[8][3];
    int n = 3;
    for (int k = 0; k < 3; ++k) std::memcpy(poly[k], c.verts[c.tris[t].v[k]].position, 12);
    for (int pass = 0; pass < 2; ++pass) {
        const float plane = pass == 0 ? lo : hi;
        const float sign = pass == 0 ? 1.0f : -1.0f;           // keep sign * (x - plane) >= 0
        if (!std::isfinite(plane)) continue;
        int m = 0;
        for (int i = 0; i < n; ++i) {
            const float* a = poly[i];
            const float* b = poly[(i + 1) % n];
            const float da = sign * (a[axis] - plane), db = sign * (b[axis] - plane);
            if (da >= 0.0f) { std::memcpy(tmp[m++], a, 12); }
            if ((da >= 0.0f) != (db >= 0.0f)) {
                const float tpar = da / (da - db);
                for (int k = 0; k < 3; ++k) tmp[m][k] = a[k] + (b[k] - a[k]) * tpar;
                tmp[m][axis] = plane;
                ++m;
            }
        }
        n = m;
        if (n == 0) return false;
        std::memcpy(poly, tmp, sizeof(float) * 3 * n);
    }
    Box b;
    b.reset();
    for (int i = 0; i < n; ++i) b.grow(poly[i]);
    const float scale = std::sqrt(c.rootArea);
    for (int k = 0; k < 3; ++k) {
        if (k == axis) {            // exact on the clipping axis
            b.lo[k] = std::fmax(b.lo[k], lo); b.hi[k] = std::fmin(b.hi[k], hi);
        }
        else { b.lo[k] = pad(b.lo[k], -1.0f, scale); b.hi[k] = pad(b.hi[k], 1.0f, scale); }
        b.lo[k] = std::fmax(b.lo[k], within.lo[k]);
        b.hi[k] = std::fmin(b.hi[k], within.hi[k]);
        if (b.lo[k] > b.hi[k]) return false;
    }
    *out = b;
    return true;
}

struct Split {
    float cost = INFINITY;
    int axis = -1;
    int plane = 0;          // object: bin index; spatial: bin index (plane after bin `plane - 1`)
    float pos = 0.0f;       // spatial: coordinate of the plane
    Box left, right;
    uint32_t nLeft = 0, nRight = 0;
};

uint32_t build(Ctx& c, std::vector<Ref>& refs, uint32_t nodeIdx, int depth, std::vector<uint32_t>& leafPrims, std::vector<std::pair<uint32_t, uint32_t>>& leafOf);

void makeLeaf(Ctx& c, std::vector<Ref>& refs, const Box& box, uint32_t nodeIdx, std::vector<uint32_t>& leafPrims,
              std::vector<std::pair<uint32_t, uint32_t>>& leafOf) {
    BNode nd;
    nd.box = box;
    nd.left = nd.right = 0;
    nd.first = (uint32_t)leafPrims.size();           // thread-local offset, rebased when the lists are concatenated
    nd.count = (uint32_t)refs.size();
    for (const Ref& r : refs) leafPrims.push_back(r.prim);
    (*c.nodes)[nodeIdx] = nd;
    leafOf.push_back({nodeIdx, nd.first});
}

uint32_t build(Ctx& c, std::vector<Ref>& refs, uint32_t nodeIdx, int depth, std::vector<uint32_t>& leafPrims,
               std::vector<std::pair<uint32_t, uint32_t>>& leafOf) {
    {
        uint32_t d = c.depth.load();
        while ((uint32_t)depth > d && !c.depth.compare_exchange_weak(d, (uint32_t)depth)) {}
    }
    const uint32_t n = (uint32_t)refs.size();
    Box box, cbox;
    box.reset(); cbox.reset();
    for (const Ref& r : refs) {
        box.grow(r.box);
        const float cen[3] = {0.5f * (r.box.lo[0] + r.box.hi[0]), 0.5f * (r.box.lo[1] + r.box.hi[1]), 0.5f * (r.box.lo[2] + r.box.hi[2])};
        cbox.grow(cen);
    }
    if (n <= 1) { makeLeaf(c, refs, box, nodeIdx, leafPrims, leafOf); return nodeIdx; }

    // ---- object split: binned SAH on the centroids, all three axes ------------------------------------------------------
    Split obj;
    for (int a = 0; a < 3; ++a) {
        const float ext = cbox.hi[a] - cbox.lo[a];
        if (!(ext > 0.0f)) continue;
        Box bb[kBins];
        uint32_t bc[kBins];
        for (int b = 0; b < kBins; ++b) { bb[b].reset(); bc[b] = 0; }
        const float scale = kBins / ext;
        for (const Ref& r : refs) {
            const int b = std::min(kBins - 1, std::max(0, (int)((0.5f * (r.box.lo[a] + r.box.hi[a]) - cbox.lo[a]) * scale)));
            bb[b].grow(r.box);
            ++bc[b];
        }
        Box rightBox[kBins];
        uint32_t rightCount[kBins];
        Box acc; acc.reset();
        uint32_t cnt = 0;
        for (int b = kBins - 1; b > 0; --b) { acc.grow(bb[b]); cnt += bc[b]; rightBox[b] = acc; rightCount[b] = cnt; }
        acc.reset(); cnt = 0;
        for (int b = 1; b < kBins; ++b) {
            acc.grow(bb[b - 1]); cnt += bc[b - 1];
            if (cnt == 0 || rightCount[b] == 0) continue;
            const float cost = acc.area() * cnt + rightBox[b].area() * rightCount[b];
            if (cost < obj.cost) { obj.cost = cost; obj.axis = a; obj.plane = b; obj.left = acc; obj.right = rightBox[b]; obj.nLeft = cnt; obj.nRight = rightCount[b]; }
        }
    }

    // ---- spatial split: only where the object split's children overlap noticeably (SBVH.h:160-176) --------------------------
    Split spa;
    bool trySpatial = false;
    if (obj.axis >= 0 && depth < kMaxDepth - 8) {
        Box ov;
        for (int k = 0; k < 3; ++k) { ov.lo[k] = std::fmax(obj.left.lo[k], obj.right.lo[k]); ov.hi[k] = std::fmin(obj.left.hi[k], obj.right.hi[k]); }
        trySpatial = ov.lo[0] <= ov.hi[0] && ov.lo[1] <= ov.hi[1] && ov.lo[2] <= ov.hi[2] && ov.area() / c.rootArea > kAlpha;
    }
    else if (obj.axis < 0 && depth < kMaxDepth - 8) trySpatial = true;         // identical centroids: only space can separate them
    if (trySpatial && c.refsLeft.load() > 0) {
        for (int a = 0; a < 3; ++a) {
            const float lo = box.lo[a], ext = box.hi[a] - box.lo[a];
            if (!(ext > 0.0f)) continue;
            Box bb[kBins];
            uint32_t entries[kBins], exits[kBins];
            for (int b = 0; b < kBins; ++b) { bb[b].reset(); entries[b] = exits[b] = 0; }
            const float scale = kBins / ext, width = ext / kBins;
            for (const Ref& r : refs) {
                const int b0 = std::min(kBins - 1, std::max(0, (int)((r.box.lo[a] - lo) * scale)));
                const int b1 = std::min(kBins - 1, std::max(b0, (int)((r.box.hi[a] - lo) * scale)));
                ++entries[b0];
                ++exits[b1];
                if (b0 == b1) { bb[b0].grow(r.box); continue; }
                for (int b = b0; b <= b1; ++b) {
                    Box cb;
                    const float pl = b == 0 ? -INFINITY : lo + width * b, ph = b == kBins - 1 ? INFINITY : lo + width * (b + 1);
                    if (clippedBounds(c, r.prim, a, pl, ph, r.box, &cb)) bb[b].grow(cb);
                }
            }
            Box rightBox[kBins];
            uint32_t rightCount[kBins];
            Box acc; acc.reset();
            uint32_t cnt = 0;
            for (int b = kBins - 1; b > 0; --b) { acc.grow(bb[b]); cnt += exits[b]; rightBox[b] = acc; rightCount[b] = cnt; }
            acc.reset(); cnt = 0;
            for (int b = 1; b < kBins; ++b) {
                acc.grow(bb[b - 1]); cnt += entries[b - 1];
                if (cnt == 0 || rightCount[b] == 0) continue;
                const float cost = acc.area() * cnt + rightBox[b].area() * rightCount[b];
                if (cost < spa.cost) {
                    spa.cost = cost; spa.axis = a; spa.plane = b; spa.pos = lo + width * b;
                    spa.left = acc; spa.right = rightBox[b]; spa.nLeft = cnt; spa.nRight = rightCount[b];
                }
            }
        }
        // a spatial split must make progress and fit the budget
        if (spa.axis >= 0 && (spa.nLeft >= n || spa.nRight >= n || (int64_t)(spa.nLeft + spa.nRight - n) > c.refsLeft.load())) spa.cost = INFINITY;
    }

    const float leafCost = box.area() * n;
    const float bestCost = std::fmin(obj.cost, spa.cost);
    if (n <= kMaxLeafTris && !(bestCost + 0.125f * box.area() < leafCost)) { makeLeaf(c, refs, box, nodeIdx, leafPrims, leafOf); return nodeIdx; }

    std::vector<Ref> left, right;
    if (spa.cost < obj.cost) {
        left.reserve(spa.nLeft); right.reserve(spa.nRight);
        const int a = spa.axis;
        Box lb = spa.left, rb = spa.right;
        for (const Ref& r : refs) {
            if (r.box.hi[a] <= spa.pos) { left.push_back(r); continue; }
            if (r.box.lo[a] >= spa.pos) { right.push_back(r); continue; }
            Ref rl = r, rr = r;
            const bool okL = clippedBounds(c, r.prim, a, -INFINITY, spa.pos, r.box, &rl.box);
            const bool okR = clippedBounds(c, r.prim, a, spa.pos, INFINITY, r.box, &rr.box);
            if (okL && okR) {
                // reference unsplitting (Stich et al. 4.3): keep the whole reference on one side if that is cheaper
                Box lWhole = lb, rWhole = rb;
                lWhole.grow(r.box); rWhole.grow(r.box);
                const float cSplit = lb.area() * spa.nLeft + rb.area() * spa.nRight;
                const float cLeft = lWhole.area() * spa.nLeft + rb.area() * (spa.nRight - 1);
                const float cRight = lb.area() * (spa.nLeft - 1) + rWhole.area() * spa.nRight;
                if (cLeft < cSplit && cLeft <= cRight) { left.push_back(r); lb = lWhole; continue; }
                if (cRight < cSplit) { right.push_back(r); rb = rWhole; continue; }
                left.push_back(rl); right.push_back(rr);
            }
            else if (okL) left.push_back(rl);
            else if (okR) right.push_back(rr);
            else left.push_back(r);                  // degenerate: keep it somewhere
        }
        if (left.empty() || right.empty() || left.size() >= n || right.size() >= n) { left.clear(); right.clear(); }     // no progress: fall through
        else {
            c.refsLeft.fetch_sub((int64_t)(left.size() + right.size()) - (int64_t)n);
            ++c.spatialSplits;
        }
    }
    if (left.empty()) {
        if (obj.axis >= 0) {
            const int a = obj.axis;
            const float ext = cbox.hi[a] - cbox.lo[a], scale = kBins / ext, lo = cbox.lo[a];
            for (const Ref& r : refs) {
                const int b = std::min(kBins - 1, std::max(0, (int)((0.5f * (r.box.lo[a] + r.box.hi[a]) - lo) * scale)));
                (b < obj.plane ? left : right).push_back(r);
            }
        }
        if (left.empty() || right.empty()) {            // identical centroids (or a degenerate bin assignment): split the list
            left.assign(refs.begin(), refs.begin() + n / 2);
            right.assign(refs.begin() + n / 2, refs.end());
        }
        ++c.objectSplits;
    }
    std::vector<Ref>().swap(refs);                       // release this level's list before descending

    BNode nd;
    nd.box = box;
    nd.count = 0;
    nd.first = 0;
    nd.left = c.nodeCursor.fetch_add(2);
    nd.right = nd.left + 1;
    (*c.nodes)[nodeIdx] = nd;

    const size_t grain = 1u << 15;
    if (left.size() + right.size() > 4 * grain && right.size() > grain && c.threadsLeft.fetch_sub(1) > 0) {
        // the right subtree on its own thread, with its own leaf list (rebased by the caller chain through leafOf)
        std::vector<uint32_t> rp;
        std::vector<std::pair<uint32_t, uint32_t>> rl;
        std::thread th([&] { build(c, right, nd.right, depth + 1, rp, rl); });
        build(c, left, nd.left, depth + 1, leafPrims, leafOf);
        th.join();
        c.threadsLeft.fetch_add(1);
        const uint32_t base = (uint32_t)leafPrims.size();
        leafPrims.insert(leafPrims.end(), rp.begin(), rp.end());
        for (auto& e : rl) { (*c.nodes)[e.first].first = e.second + base; leafOf.push_back({e.first, e.second + base}); }
    }
    else {
        if (left.size() + right.size() > 4 * grain && right.size() > grain) c.threadsLeft.fetch_add(1);      // undo the failed reservation
        build(c, left, nd.left, depth + 1, leafPrims, leafOf);
        build(c, right, nd.right, depth + 1, leafPrims, leafOf);
    }
    return nodeIdx;
}

} // namespace

void buildBinarySBVH(const slrhip_vertex* verts, const slrhip_triangle* tris, uint32_t numTris, float refBudget,
                     std::vector<BNode>* nodes, std::vector<uint32_t>* prims, SbvhStats* stats) {
    std::vector<Ref> refs(numTris);
    Box root;
    root.reset();
    for (uint32_t i = 0; i < numTris; ++i) {
        refs[i].prim = i;
        refs[i].box.reset();
        for (int k = 0; k < 3; ++k) refs[i].box.grow(verts[tris[i].v[k]].position);
        root.grow(refs[i].box);
    }
    const uint64_t maxRefs = (uint64_t)std::ceil((double)numTris * (double)std::fmax(refBudget, 1.0f)) + 16;
    Ctx c;
    c.verts = verts;
    c.tris = tris;
    c.rootArea = std::fmax(root.area(), 1e-30f);
    c.nodes = nodes;
    c.refsLeft = (int64_t)(maxRefs - numTris);
    c.threadsLeft = (int)hostThreads() - 1;
    nodes->assign(2 * maxRefs + 2, BNode());
    prims->clear();
    prims->reserve(maxRefs);
    std::vector<std::pair<uint32_t, uint32_t>> leafOf;
    build(c, refs, 0, 1, *prims, leafOf);
    nodes->resize(c.nodeCursor.load());
    if (stats) {
        stats->spatialSplits = c.spatialSplits.load();
        stats->objectSplits = c.objectSplits.load();
        stats->references = prims->size();
        stats->depth = c.depth.load();
    }
}

} // namespace slrhip
