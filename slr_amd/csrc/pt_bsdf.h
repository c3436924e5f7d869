// pt_bsdf.h — per-hit BSDF evaluation and sampling on the device.
//
// The reference builds a BSDF object per hit on an arena (SurfacePoint::createBSDF, Core/geometry.cpp:56-58
// -> SurfaceMaterial::getBSDF, SurfaceMaterials/basic_SurfaceMaterials.cpp:15-43) and calls it through a
// vtable.  Here the material record IS the BSDF: a switch on the material type, constants in registers.
#pragma once
#include "pt_device.h"

namespace slrhip {

struct Mat {
    uint32_t type;
    float param;
    RGB a, b, c, emittance;
};

SLR_DEV Mat loadMat(const DevMaterial* m) {
    const float4* q = reinterpret_cast<const float4*>(m);
    const float4 h = q[0], a = q[1], b = q[2], c = q[3], e = q[4];
    Mat r;
    r.type = __float_as_uint(h.x);
    r.param = h.y;
    r.a = RGB(a.x, a.y, a.z); r.b = RGB(b.x, b.y, b.z); r.c = RGB(c.x, c.y, c.z); r.emittance = RGB(e.x, e.y, e.z);
    return r;
}

// DiffuseEDF::evaluate, EDFs/basic_EDFs.cpp:19-23: (dir.z > 0 ? 1.0f / M_PI : 0.0f) -> float
SLR_DEV float diffuseEDF(V3 dir) { return dir.z > 0.0f ? (float)(1.0 / kPi) : 0.0f; }

struct BsdfSample {
    V3 dir_sn;
    float dirPDF;
    uint32_t dirType;
};

// Lobe type per material (basic_BSDFs.h:27,43,59-61; dispersive = !wls.lambdaSelected(),
// basic_SurfaceMaterials.cpp:42)
SLR_DEV uint32_t bsdfType(uint32_t matType, uint32_t wlFlags) {
    switch (matType) {
    case SLRHIP_MATERIAL_MATTE: return DT_Reflection | DT_LowFreq;
    case SLRHIP_MATERIAL_METAL: return DT_Reflection | DT_Delta0D;
    case SLRHIP_MATERIAL_GLASS: return DT_Reflection | DT_Transmission | DT_Delta0D | ((wlFlags & 1u) ? 0u : (uint32_t)DT_Dispersive);
    default: return 0;
    }
}

// BSDF::sample (DDF.h:231-246) over sampleInternal of LambertianBRDF / SpecularBRDF / SpecularBSDF
// (BSDFs/basic_BSDFs.cpp:12-26, 61-71, 95-149); query.flags = All, adjoint = false.
SLR_DEV RGB bsdfSample(const Mat& m, uint32_t type, V3 dirOut, V3 gNorm, uint32_t wl, float uComp, float u0, float u1, BsdfSample* res) {
    res->dirPDF = 0.0f;
    res->dirType = 0;
    if (!dtMatches(type, DT_All)) return RGB();
    RGB fs_sn;
    switch (m.type) {
    case SLRHIP_MATERIAL_MATTE: {
        res->dir_sn = cosineSampleHemisphere(u0, u1);
        res->dirPDF = (float)((double)res->dir_sn.z / kPi);
        res->dirType = type;
        res->dir_sn.z *= dot(dirOut, gNorm) > 0 ? 1 : -1;
        fs_sn = m.a / (float)kPi;
        break;
    }
    case SLRHIP_MATERIAL_METAL: {
        res->dir_sn = V3(-dirOut.x, -dirOut.y, dirOut.z);
        res->dirPDF = 1.0f;
        res->dirType = type;
        fs_sn = m.a * fresnelConductor(m.b, m.c, dirOut.z) / fabsf(dirOut.z);
        break;
    }
    case SLRHIP_MATERIAL_GLASS: {
        RGB F = fresnelDielectric(m.b, m.c, dirOut.z);
        float reflectProb = importance(F, wl);
        if (uComp < reflectProb) {
            if (dirOut.z == 0.0f) return RGB();
            res->dir_sn = V3(-dirOut.x, -dirOut.y, dirOut.z);
            res->dirPDF = reflectProb;
            res->dirType = DT_Reflection | DT_Delta0D;
            fs_sn = m.a * F / fabsf(dirOut.z);
        }
        else {
            bool entering = dirOut.z > 0.0f;
            float etaExtW = m.b.comp(wl), etaIntW = m.c.comp(wl);
            float eEnter = entering ? etaExtW : etaIntW;
            float eExit = entering ? etaIntW : etaExtW;
            float sinEnter2 = 1.0f - dirOut.z * dirOut.z;
            float rrEta = eEnter / eExit;
            float sinExit2 = rrEta * rrEta * sinEnter2;
            if (sinExit2 >= 1.0f) return RGB();
            float cosExit = sqrtf(fmaxf(0.0f, 1.0f - sinExit2));
            if (entering) cosExit = -cosExit;
            res->dir_sn = V3(rrEta * -dirOut.x, rrEta * -dirOut.y, cosExit);
            res->dirPDF = 1.0f - reflectProb;
            res->dirType = DT_Transmission | DT_Delta0D | ((type & DT_Dispersive) ? (uint32_t)DT_Dispersive : 0u);
            float v = m.a.comp(wl) * (1.0f - F.comp(wl));
            v *= (eEnter * eEnter) / (eExit * eExit);
            RGB ret(wl == 0 ? v : 0.0f, wl == 1 ? v : 0.0f, wl == 2 ? v : 0.0f);
            fs_sn = ret / fabsf(cosExit);
        }
        break;
    }
    default:
        return RGB();
    }
    if (res->dirPDF == 0.0f) return RGB();
    float snCorrection = fabsf(res->dir_sn.z / dot(res->dir_sn, gNorm));
    return fs_sn * snCorrection;
}

// BSDF::evaluate (DDF.h:247-267) + evaluatePDF (:268-279) for the NEE direction.
SLR_DEV RGB bsdfEvaluate(const Mat& m, uint32_t type, V3 dirOut, V3 gNorm, V3 dir, float* pdf) {
    *pdf = 0.0f;
    if (dtMatches(type, DT_All) && m.type == SLRHIP_MATERIAL_MATTE) {
        // LambertianBRDF::evaluatePDFInternal basic_BSDFs.cpp:41-50
        if (!(dirOut.z * dir.z <= 0.0f)) *pdf = (float)((double)fabsf(dir.z) / kPi);
    }
    bool reflect = dot(gNorm, dirOut) * dot(gNorm, dir) > 0;                       // sideTest DDF.h:213-216
    uint32_t flags = DT_All & (DT_AllFreq | (reflect ? DT_Reflection : DT_Transmission));
    if (!dtMatches(type, flags)) return RGB();
    RGB fs_sn;
    if (m.type == SLRHIP_MATERIAL_MATTE) {
        // LambertianBRDF::evaluateInternal basic_BSDFs.cpp:28-39
        if (dirOut.z * dir.z <= 0.0f) fs_sn = RGB();
        else fs_sn = m.a / (float)kPi;
    }
    float snCorrection = fabsf(dir.z / dot(dir, gNorm));
    return fs_sn * snCorrection;
}

} // namespace slrhip
