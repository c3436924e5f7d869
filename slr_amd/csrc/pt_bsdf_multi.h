// pt_bsdf_multi.h — MultiBSDF over two components: lobes (either of which may be an InverseBSDF) or MultiBSDFs of two lobes.
//
// Reference: SummedSurfaceMaterial::getBSDF / MixedSurfaceMaterial::getBSDF build a MultiBSDF of the two components'
// BSDFs (SurfaceMaterials/SummedSurfaceMaterial.cpp:13-20, MixedSurfaceMaterial.cpp:14-22, with `scale` handed down to
// the components); InverseSurfaceMaterial wraps its base BSDF in InverseBSDF (basic_SurfaceMaterials.cpp:47-50).
// MultiBSDF: BSDFs/MultiBSDF.cpp:12-217 (the NoRev variants: this integrator never asks for reverse values);
// InverseBSDF: BSDFs/basic_BSDFs.cpp:172-203.
//
// Only kernels instantiated with MULTI = true include this code (scenes with an SLRHIP_MATERIAL_MULTI record).
// A component is re-loaded from the material table each time it is needed instead of keeping two Mat<S> alive.
#pragma once
#include "pt_bsdf.h"

namespace slrhip {

SLR_DEV uint32_t dtFlip(uint32_t t) { return t ^ (uint32_t)DT_WholeSphere; }      // DirectionType::flip DDF.h:79

// luminance(): RGBTypes.h:94-100 (double coefficients, float result) / SpectrumTypes.h:504-509 (mean of the samples)
SLR_DEV float luminance(const RGB& s) { return (float)(0.222485 * (double)s.r + 0.716905 * (double)s.g + 0.060610 * (double)s.b); }
SLR_DEV float luminance(const Spec16& s) { return s.sum() / 16; }

// The MULTI record: param = scale of component 0, onA = scale of component 1, onB = packMultiBits(...) (device_types.h) as raw bits
// (written by slrhip_upload_scene; only moved, never used in arithmetic).  A component may itself be a MULTI record whose
// components are single lobes: the reference's material expressions are binary trees (SummedSurfaceMaterial / MixedSurfaceMaterial
// always add two BSDFs), here up to two levels = four lobes (MultiBSDF::maxNumElems, MultiBSDF.h:17).
struct MultiRec {
    uint32_t child[2];
    uint32_t childType[2];     // SLRHIP_MATERIAL_* of the components, so the union lobe type needs no table look-up
    bool inverse[2];
    float scale[2];
};
template <class S>
SLR_DEV MultiRec decodeMulti(const Mat<S>& m) {
    const uint32_t bits = __float_as_uint(m.onB);
    MultiRec r;
    r.child[0] = bits & 0x3FFu; r.child[1] = (bits >> 10) & 0x3FFu;
    r.inverse[0] = (bits >> 20) & 1u; r.inverse[1] = (bits >> 21) & 1u;
    r.childType[0] = (bits >> 22) & 7u; r.childType[1] = (bits >> 25) & 7u;
    r.scale[0] = m.param; r.scale[1] = m.onA;
    return r;
}

// The expression tree flattened: leaves 0 .. numLeaves-1 in order; the root's component 0 covers the first size0 of them
// (1 = a lobe, 2 = a nested MultiBSDF), component 1 the rest.
struct MultiTree {
    uint32_t leafMat[4];       // material table index
    uint32_t leafMatType[4];   // SLRHIP_MATERIAL_*
    float leafScale[4];        // the `scale` the lobe's getBSDF receives: the product down the tree, in the reference's order
    uint32_t inverseMask;      // bit i: leaf i is wrapped in InverseBSDF
    uint32_t numLeaves, size0;
};
// run-time index into a four-entry table without an indexed register file access
template <class T> SLR_DEV T sel4(const T (&a)[4], uint32_t i) {
    T r = a[0];
    r = i == 1u ? a[1] : r;
    r = i == 2u ? a[2] : r;
    r = i == 3u ? a[3] : r;
    return r;
}
template <class S, class Load>
SLR_DEV MultiTree buildMultiTree(const MultiRec& root, const Load& load) {
    MultiTree t;
    t.inverseMask = 0u;
    uint32_t n = 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        if (root.childType[i] == SLRHIP_MATERIAL_MULTI) {
            const MultiRec sub = decodeMulti(load(root.child[i]));
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                t.leafMat[n] = sub.child[j]; t.leafMatType[n] = sub.childType[j];
                t.leafScale[n] = root.scale[i] * sub.scale[j];          // `scale * (1.0f - factor)` / `scale * factor` one level down
                if (sub.inverse[j]) t.inverseMask |= 1u << n;
                ++n;
            }
        }
        else {
            t.leafMat[n] = root.child[i]; t.leafMatType[n] = root.childType[i]; t.leafScale[n] = root.scale[i];
            if (root.inverse[i]) t.inverseMask |= 1u << n;
            ++n;
        }
        if (i == 0) t.size0 = n;
    }
    t.numLeaves = n;
    for (uint32_t k = n; k < 4u; ++k) { t.leafMat[k] = t.leafMat[0]; t.leafMatType[k] = t.leafMatType[0]; t.leafScale[k] = 0.0f; }
    return t;
}
// m_type of a group of leaves: the union of its lobes' (MultiBSDF::add, MultiBSDF.cpp:16; InverseBSDF ctor basic_BSDFs.h:71)
SLR_DEV uint32_t multiGroupType(const MultiTree& t, uint32_t first, uint32_t count, uint32_t wlFlags) {
    uint32_t type = 0u;
    for (uint32_t i = first; i < first + count; ++i) {
        const uint32_t lt = bsdfType(sel4(t.leafMatType, i), wlFlags);
        type |= ((t.inverseMask >> i) & 1u) ? dtFlip(lt) : lt;
    }
    return type;
}
SLR_DEV uint32_t multiType(const MultiTree& t, uint32_t wlFlags) { return multiGroupType(t, 0u, t.numLeaves, wlFlags); }
// `scale * spectrum` in the components' getBSDF (basic_SurfaceMaterials.cpp:19,22,33,42, ModifiedWardDurReflection.cpp:18,
// AshikhminShirleyReflection.cpp:19); the microfacet materials ignore their scale (MicrofacetSurfaceMaterial.cpp:14-28).
// The loaders applied scale = 1 already, an exact multiply.
template <class S>
SLR_DEV void applyScale(Mat<S>& c, float scale) {
    switch (c.type) {
    case SLRHIP_MATERIAL_MATTE: case SLRHIP_MATERIAL_METAL: case SLRHIP_MATERIAL_GLASS: case SLRHIP_MATERIAL_WARD:
        c.a = scale * c.a;
        break;
    case SLRHIP_MATERIAL_ASHIKHMIN:
        c.a = scale * c.a;
        c.b = scale * c.b;
        break;
    default:
        break;
    }
}

// weightInternal of each lobe under BSDF::weight (DDF.h:280-289; non-adjoint: no correction)
template <class S>
SLR_DEV float bsdfWeight(const Mat<S>& m, uint32_t type, uint32_t flags, V3 dirOut, uint32_t wl) {
    if (!dtMatches(type, flags)) return 0.0f;
    switch (m.type) {
    case SLRHIP_MATERIAL_MATTE:                  // LambertianBRDF basic_BSDFs.cpp:51-53: importance;  OrenNayerBRDF.cpp:67-69: luminance
        return m.param >= 0.0f ? luminance(m.a) : importance(m.a, wl);
    case SLRHIP_MATERIAL_METAL:                  // basic_BSDFs.cpp:85-87
        return importance(m.a, wl) * importance(fresnelConductor(m.b, m.c, dirOut.z), wl);
    case SLRHIP_MATERIAL_GLASS:                  // basic_BSDFs.cpp:163-165
        return importance(m.a, wl);
    case SLRHIP_MATERIAL_MICROFACET_METAL:
    case SLRHIP_MATERIAL_MICROFACET_GLASS: {     // MicrofacetBSDF.cpp:102-106, 307-311
        GGX D_ = {m.param};
        float sign = dirOut.z >= 0.0f ? 1.0f : -1.0f;
        return D_.evaluateSmithG1(dirOut * sign, V3(0, 0, 1));
    }
    case SLRHIP_MATERIAL_WARD:                   // ModifiedWardDurBRDF.cpp:80-82
        return importance(m.a, wl);
    case SLRHIP_MATERIAL_ASHIKHMIN: {            // AshikhminShirleyBRDF.cpp:156-165
        float specularWeight, diffuseWeight;
        ashikhminWeights(m, wl, fabsf(dirOut.z), &specularWeight, &diffuseWeight);
        return specularWeight + diffuseWeight;
    }
    default:
        return 0.0f;
    }
}

// One component as MultiBSDF sees it: m_type, weight(), and the three *Internal calls — for an InverseBSDF those
// forward to the base's PUBLIC sample / evaluate / evaluatePDF / weight with a flipped query, exactly as the reference
// does (including evaluatePDFInternal, whose `mQuery.flags.flip();` discards its result, basic_BSDFs.cpp:193).
template <class S>
struct Component {
    Mat<S> base;
    uint32_t baseType;
    bool inverse;
    SLR_DEV uint32_t type() const { return inverse ? dtFlip(baseType) : baseType; }
    SLR_DEV bool matches(uint32_t flags) const { return dtMatches(type(), flags); }
    SLR_DEV float weight(uint32_t flags, V3 dirOut, uint32_t wl) const {
        if (!matches(flags)) return 0.0f;
        return bsdfWeight(base, baseType, inverse ? dtFlip(flags) : flags, dirOut, wl);
    }
    SLR_DEV S sampleInternal(uint32_t flags, V3 dirOut, V3 gNorm, uint32_t wl, float uComp, float u0, float u1, BsdfSample* res) const {
        if (!inverse) return bsdfSampleInternal<S, true>(base, baseType, dirOut, gNorm, wl, uComp, u0, u1, res);
        S ret = bsdfSample<S, true>(base, baseType, dirOut, gNorm, wl, uComp, u0, u1, res, dtFlip(flags));
        res->dirType = dtFlip(res->dirType);
        res->dir_sn.z *= -1;
        return ret;
    }
    SLR_DEV S evaluateInternal(uint32_t flags, V3 dirOut, V3 gNorm, V3 dir, uint32_t wl) const {
        if (!inverse) return bsdfEvaluateInternal<S, true>(base, flags, dirOut, dir, wl);
        V3 mDir = dir;
        mDir.z *= -1;
        return bsdfEvaluateOnly<S, true>(base, baseType, dirOut, gNorm, mDir, wl, dtFlip(flags));
    }
    SLR_DEV float evaluatePDFInternal(uint32_t flags, V3 dirOut, V3 dir, uint32_t wl) const {
        if (!inverse) return bsdfEvaluatePDFInternal<S, true>(base, dirOut, dir, wl);
        V3 mDir = dir;
        mDir.z *= -1;
        return bsdfEvaluatePDF<S, true>(base, baseType, dirOut, mDir, wl, flags);
    }
};

// Core/distributions.cpp:14-29 for two importances (its compensated sums equal the plain float sums for two items).
SLR_DEV uint32_t sampleDiscrete2(float w0, float w1, float* sumImportances, float* base, float u) {
    const float sum = w0 + w1;
    *sumImportances = sum;
    const float su = u * sum;
    *base = 0.0f;
    if (su < w0) return 0;
    *base = w0;
    if (su < sum) return 1;
    return 0;          // falls out of the loop: index 0 with the base of the last step
}

// `load(i)` returns material record i as a Mat<S> — a callable so the kernel decides where the tables live.
// The lobes are visited in loops with a run-time leaf index (one copy of the lobe code per operation), and the sums are taken
// in the reference's nesting order: a nested MultiBSDF's weight() = BSDF::weight over MultiBSDF::weightInternal (the sum of
// its two lobes' weights), its sampleInternal / evaluateInternal / evaluatePDFInternal = the same functions one level down.
template <class S, class Load>
struct MultiBSDF {
    MultiTree tr;
    uint32_t wlFlags;      // WavelengthSamples flags (dispersive = !lambdaSelected for the specular dielectric)
    Load load;

    SLR_DEV Component<S> leaf(uint32_t i) const {
        Component<S> c;
        c.base = load(sel4(tr.leafMat, i));
        applyScale(c.base, sel4(tr.leafScale, i));
        c.baseType = bsdfType(c.base.type, wlFlags);
        c.inverse = (tr.inverseMask >> i) & 1u;
        return c;
    }
    SLR_DEV uint32_t groupFirst(uint32_t g) const { return g ? tr.size0 : 0u; }
    SLR_DEV uint32_t groupSize(uint32_t g) const { return g ? tr.numLeaves - tr.size0 : tr.size0; }
    SLR_DEV bool groupMatches(uint32_t g, uint32_t flags) const { return dtMatches(multiGroupType(tr, groupFirst(g), groupSize(g), wlFlags), flags); }
    // BSDF::weight of root component g: a lobe's own, or matches ? w_a + w_b : 0 for a nested MultiBSDF (weightInternal, MultiBSDF.cpp:207-212)
    SLR_DEV float groupWeight(uint32_t g, uint32_t flags, const float (&w)[4]) const {
        const uint32_t f = groupFirst(g);
        if (groupSize(g) == 1u) return sel4(w, f);
        return groupMatches(g, flags) ? sel4(w, f) + sel4(w, f + 1u) : 0.0f;
    }

    // BSDF::sample (DDF.h:231-246) over MultiBSDF::sampleInternalNoRev (MultiBSDF.cpp:20-59); query.flags = All
    SLR_DEV S sample(uint32_t type, V3 dirOut, V3 gNorm, uint32_t wl, float uComp, float u0, float u1, BsdfSample* res) const {
        const uint32_t flags = DT_All;
        res->dirPDF = 0.0f;
        res->dirType = 0;
        if (!dtMatches(type, flags)) return S();
        float w[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll 1
        for (uint32_t i = 0; i < tr.numLeaves; ++i) {
            const float wi = leaf(i).weight(flags, dirOut, wl);
            w[0] = i == 0u ? wi : w[0]; w[1] = i == 1u ? wi : w[1]; w[2] = i == 2u ? wi : w[2]; w[3] = i == 3u ? wi : w[3];
        }
        const float W0 = groupWeight(0u, flags, w), W1 = groupWeight(1u, flags, w);
        float sumWeights, base;
        const uint32_t g = sampleDiscrete2(W0, W1, &sumWeights, &base, uComp);
        if (sumWeights == 0.0f) return S();
        const float Wsel = g == 0u ? W0 : W1, Wother = g == 0u ? W1 : W0;
        uComp = (uComp * sumWeights - base) / Wsel;
        // the selected component: a lobe, or a nested MultiBSDF that selects one of its two lobes the same way
        const uint32_t first = groupFirst(g), nested = groupSize(g) == 2u;
        uint32_t selLeaf = first;
        float innerSum = 0.0f, wa = 0.0f, wb = 0.0f;
        if (nested) {
            wa = sel4(w, first); wb = sel4(w, first + 1u);
            float ibase;
            const uint32_t ii = sampleDiscrete2(wa, wb, &innerSum, &ibase, uComp);
            if (innerSum == 0.0f) return S();                    // inner: dirPDF = 0, Zero; outer: dirPDF *= weight stays 0, Zero
            selLeaf = first + ii;
            uComp = (uComp * innerSum - ibase) / (ii == 0u ? wa : wb);
        }
        S value = leaf(selLeaf).sampleInternal(flags, dirOut, gNorm, wl, uComp, u0, u1, res);
        const bool delta = dtIsDelta(res->dirType);
        if (nested) {
            res->dirPDF *= selLeaf == first ? wa : wb;
            if (res->dirPDF == 0.0f) return S();
            if (!delta) {
                const uint32_t other = selLeaf == first ? first + 1u : first;
                const Component<S> c = leaf(other);
                if (c.matches(flags)) res->dirPDF += c.evaluatePDFInternal(flags, dirOut, res->dir_sn, wl) * (other == first ? wa : wb);
            }
            res->dirPDF /= innerSum;
        }
        res->dirPDF *= Wsel;
        if (res->dirPDF == 0.0f) return S();
        if (!delta) {
            // the other root component's PDF term (:41-45)
            const uint32_t og = 1u - g, of = groupFirst(og);
            if (groupMatches(og, flags)) {
                float pdfOther;
                if (groupSize(og) == 1u) pdfOther = leaf(of).evaluatePDFInternal(flags, dirOut, res->dir_sn, wl);
                else {
                    // MultiBSDF::evaluatePDFInternalNoRev (:151-169) of the nested component
                    const float oa = sel4(w, of), ob = sel4(w, of + 1u);
                    const float osum = oa + ob;
                    pdfOther = 0.0f;
                    if (osum != 0.0f) {
#pragma unroll 1
                        for (uint32_t k = 0; k < 2u; ++k) {
                            const float wk = k == 0u ? oa : ob;
                            if (wk > 0) pdfOther += leaf(of + k).evaluatePDFInternal(flags, dirOut, res->dir_sn, wl) * wk;
                        }
                        pdfOther /= osum;
                    }
                }
                res->dirPDF += pdfOther * Wother;
            }
            // the value: the sum over the matching components, each a lobe or the sum over ITS matching lobes (:47-55, :125-149)
            const uint32_t mflags = flags & sideTest(gNorm, dirOut, res->dir_sn);
            value = S();
            S groupValue;
            bool groupOn = groupMatches(0u, mflags);
#pragma unroll 1
            for (uint32_t i = 0; i < tr.numLeaves; ++i) {
                if (i == tr.size0) { if (groupOn) value = value + groupValue; groupValue = S(); groupOn = groupMatches(1u, mflags); }
                if (!groupOn) continue;
                const Component<S> c = leaf(i);
                if (c.matches(mflags)) groupValue = groupValue + c.evaluateInternal(mflags, dirOut, gNorm, res->dir_sn, wl);
            }
            if (groupOn) value = value + groupValue;
        }
        res->dirPDF /= sumWeights;
        const float snCorrection = fabsf(res->dir_sn.z / dot(res->dir_sn, gNorm));
        return value * snCorrection;
    }

    // BSDF::evaluate (DDF.h:247-267) over MultiBSDF::evaluateInternal (:125-149) and BSDF::evaluatePDF (DDF.h:268-279)
    // over evaluatePDFInternalNoRev (:151-169)
    SLR_DEV S evaluate(uint32_t type, V3 dirOut, V3 gNorm, V3 dir, uint32_t wl, float* pdf) const {
        const uint32_t queryFlags = DT_All;
        const uint32_t flags = queryFlags & sideTest(gNorm, dirOut, dir);
        const bool evalMatches = dtMatches(type, flags);
        S fs_sn, groupValue;
        float w[4] = {0.0f, 0.0f, 0.0f, 0.0f}, p[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        bool groupOn = evalMatches && groupMatches(0u, flags);
#pragma unroll 1
        for (uint32_t i = 0; i < tr.numLeaves; ++i) {
            if (i == tr.size0) { if (groupOn) fs_sn = fs_sn + groupValue; groupValue = S(); groupOn = evalMatches && groupMatches(1u, flags); }
            const Component<S> c = leaf(i);
            const float wi = c.weight(queryFlags, dirOut, wl);
            const float pi = wi > 0 ? c.evaluatePDFInternal(queryFlags, dirOut, dir, wl) : 0.0f;
            w[0] = i == 0u ? wi : w[0]; w[1] = i == 1u ? wi : w[1]; w[2] = i == 2u ? wi : w[2]; w[3] = i == 3u ? wi : w[3];
            p[0] = i == 0u ? pi : p[0]; p[1] = i == 1u ? pi : p[1]; p[2] = i == 2u ? pi : p[2]; p[3] = i == 3u ? pi : p[3];
            if (groupOn && c.matches(flags)) groupValue = groupValue + c.evaluateInternal(flags, dirOut, gNorm, dir, wl);
        }
        if (groupOn) fs_sn = fs_sn + groupValue;
        // PDFs in nesting order
        float W[2], P[2];
#pragma unroll
        for (uint32_t g = 0; g < 2u; ++g) {
            const uint32_t f = groupFirst(g);
            W[g] = groupWeight(g, queryFlags, w);
            if (groupSize(g) == 1u) P[g] = sel4(p, f);
            else {
                const float a = sel4(w, f), b = sel4(w, f + 1u), sum = a + b;
                float r = 0.0f;
                if (sum != 0.0f) {
                    if (a > 0) r += sel4(p, f) * a;
                    if (b > 0) r += sel4(p, f + 1u) * b;
                    r /= sum;
                }
                P[g] = r;
            }
        }
        const float sumWeights = W[0] + W[1];
        float retPDF = 0.0f;
        if (dtMatches(type, queryFlags) && sumWeights != 0.0f) {
            if (W[0] > 0) retPDF += P[0] * W[0];
            if (W[1] > 0) retPDF += P[1] * W[1];
            retPDF /= sumWeights;
        }
        *pdf = retPDF;
        if (!evalMatches) return S();
        const float snCorrection = fabsf(dir.z / dot(dir, gNorm));
        return fs_sn * snCorrection;
    }
};

} // namespace slrhip
