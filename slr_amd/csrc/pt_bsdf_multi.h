// pt_bsdf_multi.h — MultiBSDF over two component lobes, either of which may be an InverseBSDF.
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
// (written by slrhip_upload_scene; only moved, never used in arithmetic).
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
// m_type of the MultiBSDF: the union of its components' (MultiBSDF::add, MultiBSDF.cpp:16; InverseBSDF ctor basic_BSDFs.h:71)
SLR_DEV uint32_t multiType(const MultiRec& r, uint32_t wlFlags) {
    const uint32_t t0 = bsdfType(r.childType[0], wlFlags), t1 = bsdfType(r.childType[1], wlFlags);
    return (r.inverse[0] ? dtFlip(t0) : t0) | (r.inverse[1] ? dtFlip(t1) : t1);
}
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

// `load(i)` returns component i's Mat<S> (scale applied) — a callable so the kernel decides where the tables live.
template <class S, class Load>
struct MultiBSDF {
    MultiRec rec;
    uint32_t wlFlags;      // WavelengthSamples flags (dispersive = !lambdaSelected for the specular dielectric)
    Load load;

    SLR_DEV Component<S> component(int i) const {
        Component<S> c;
        c.base = load(rec.child[i]);
        applyScale(c.base, rec.scale[i]);
        c.baseType = bsdfType(c.base.type, wlFlags);
        c.inverse = rec.inverse[i];
        return c;
    }
    // BSDF::sample (DDF.h:231-246) over MultiBSDF::sampleInternalNoRev (MultiBSDF.cpp:20-59); query.flags = All
    SLR_DEV S sample(uint32_t type, V3 dirOut, V3 gNorm, uint32_t wl, float uComp, float u0, float u1, BsdfSample* res) const {
        const uint32_t flags = DT_All;
        res->dirPDF = 0.0f;
        res->dirType = 0;
        if (!dtMatches(type, flags)) return S();
        float weights[2];
        weights[0] = component(0).weight(flags, dirOut, wl);
        weights[1] = component(1).weight(flags, dirOut, wl);
        float sumWeights, base;
        const uint32_t idx = sampleDiscrete2(weights[0], weights[1], &sumWeights, &base, uComp);
        if (sumWeights == 0.0f) return S();
        const float wIdx = idx == 0 ? weights[0] : weights[1];
        const float wOther = idx == 0 ? weights[1] : weights[0];
        uComp = (uComp * sumWeights - base) / wIdx;
        S value;
        {
            const Component<S> sel = component((int)idx);
            value = sel.sampleInternal(flags, dirOut, gNorm, wl, uComp, u0, u1, res);
        }
        res->dirPDF *= wIdx;
        if (res->dirPDF == 0.0f) return S();
        if (!dtIsDelta(res->dirType)) {
            const uint32_t mflags = flags & sideTest(gNorm, dirOut, res->dir_sn);
            value = S();
            // the loops of :41-55 unrolled in component order; the PDF term belongs to the component that was not selected
            for (int i = 0; i < 2; ++i) {
                const Component<S> c = component(i);
                if (i != (int)idx && c.matches(flags)) res->dirPDF += c.evaluatePDFInternal(flags, dirOut, res->dir_sn, wl) * wOther;
                if (c.matches(mflags)) value = value + c.evaluateInternal(mflags, dirOut, gNorm, res->dir_sn, wl);
            }
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
        S fs_sn;
        float weights[2], pdfs[2] = {0.0f, 0.0f};
        for (int i = 0; i < 2; ++i) {
            const Component<S> c = component(i);
            weights[i] = c.weight(queryFlags, dirOut, wl);
            if (weights[i] > 0) pdfs[i] = c.evaluatePDFInternal(queryFlags, dirOut, dir, wl);
            if (evalMatches && c.matches(flags)) fs_sn = fs_sn + c.evaluateInternal(flags, dirOut, gNorm, dir, wl);
        }
        const float sumWeights = weights[0] + weights[1];
        float retPDF = 0.0f;
        if (dtMatches(type, queryFlags) && sumWeights != 0.0f) {
            if (weights[0] > 0) retPDF += pdfs[0] * weights[0];
            if (weights[1] > 0) retPDF += pdfs[1] * weights[1];
            retPDF /= sumWeights;
        }
        *pdf = retPDF;
        if (!evalMatches) return S();
        const float snCorrection = fabsf(dir.z / dot(dir, gNorm));
        return fs_sn * snCorrection;
    }
};

} // namespace slrhip
