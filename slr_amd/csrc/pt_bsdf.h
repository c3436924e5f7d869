// pt_bsdf.h — per-hit BSDF evaluation and sampling on the device.
//
// The reference builds a BSDF object per hit on an arena (SurfacePoint::createBSDF, Core/geometry.cpp:56-58
// -> SurfaceMaterial::getBSDF, SurfaceMaterials/basic_SurfaceMaterials.cpp:15-43) and calls it through a
// vtable.  Here the material record IS the BSDF: a switch on the material type, constants in registers.
#pragma once
#include "pt_device.h"

namespace slrhip {

template <class S>
struct Mat {
    uint32_t type;
    float param;
    float onA, onB;
    S a, b, c;
};

// RGB mode: the material record holds the evaluated constants (RGBTemplate::evaluate returns itself, RGBTypes.h:124-126)
SLR_DEV Mat<RGB> loadMat(const DevMaterial* m) {
    const float4* q = reinterpret_cast<const float4*>(m);
    const float4 h = q[0], a = q[1], b = q[2], c = q[3];
    Mat<RGB> r;
    r.type = __float_as_uint(h.x);
    r.param = h.y;
    r.onA = h.z; r.onB = h.w;
    r.a = RGB(a.x, a.y, a.z); r.b = RGB(b.x, b.y, b.z); r.c = RGB(c.x, c.y, c.z);
    return r;
}
SLR_DEV RGB loadEmittance(const DevMaterial* m) {
    const float4 e = reinterpret_cast<const float4*>(m)[4];
    return RGB(e.x, e.y, e.z);
}

// ---- spectral mode: ContinuousSpectrum::evaluate at the path's 16 wavelengths (per hit, like the reference) ------------
// lambda_i of WavelengthSamples::createWithEqualOffsets, SpectrumTypes.h:54-64
SLR_DEV float wavelengthOf(int i, float offset) { return 360.0f + (830.0f - 360.0f) * ((float)i + offset) / 16.0f; }

SLR_DEV DevSpectrum loadSpectrumRecord(const DevSpectrum* spectra, int32_t idx) {
    const uint4* q = reinterpret_cast<const uint4*>(spectra + idx);
    const uint4 a = q[0], b = q[1];
    DevSpectrum sp;
    sp.kind = a.x; sp.numPoints = a.y; sp.numSamples = a.z; sp.dataOffset = a.w;
    sp.scale = __uint_as_float(b.x); sp.lambdaMin = __uint_as_float(b.y); sp.lambdaMax = __uint_as_float(b.z); sp.cellOffset = b.w;
    return sp;
}

template <class S>
SLR_DEV S evalSpectrum(const DevSpectrum* spectra, const float* __restrict__ pool, int32_t idx, float wlOffset) {
    const DevSpectrum sp = loadSpectrumRecord(spectra, idx);
    const float* data = pool + sp.dataOffset;
    switch (sp.kind) {
    case SLRHIP_SPECTRUM_REGULAR: {
        // RegularContinuousSpectrumTemplate::evaluate, SpectrumTypes.h:90-109
        const uint32_t numSamples = sp.numSamples;
        return S::make([&](int i) {
            float binF = (wavelengthOf(i, wlOffset) - sp.lambdaMin) / (sp.lambdaMax - sp.lambdaMin) * (float)(numSamples - 1);
            if (binF <= 0.0f) return data[0];
            if (binF >= (float)(numSamples - 1)) return data[numSamples - 1];
            int32_t bin = (int32_t)binF;
            float t = binF - (float)bin;
            return (1 - t) * data[bin] + t * data[bin + 1];
        });
    }
    case SLRHIP_SPECTRUM_IRREGULAR: {
        // IrregularContinuousSpectrumTemplate::evaluate, SpectrumTypes.h:139-160.  std::lower_bound restricted to
        // [searchBase, n) equals the unrestricted one because the wavelengths ascend, so each component searches alone.
        const int32_t n = (int32_t)sp.numSamples;
        const float* lambdas = data;
        const float* values = data + n;
        // lower_bound = first index with lambdas[idx] >= wl.  With the host's cell table (slrhip_api.hip: lower_bound of every
        // 1-nm cell start, one byte each) the search starts at the cell's entry and walks at most a step or two: two
        // dependent loads instead of a binary search per component.  Without one: binary search for a lane's first component,
        // then onwards from the previous position (the index never decreases as wl ascends).
        const uint32_t* cellWords = sp.cellOffset != 0xFFFFFFFFu ? reinterpret_cast<const uint32_t*>(pool + sp.cellOffset) : nullptr;
        int32_t lo = 0;
        bool first = true;
        return S::make([&](int i) {
            const float wl = wavelengthOf(i, wlOffset);
            if (cellWords) {
                const uint32_t cell = min((uint32_t)fmaxf(wl - 360.0f, 0.0f), 471u);
                lo = (int32_t)((cellWords[cell >> 2] >> (8u * (cell & 3u))) & 0xFFu);
                while (lo < n && lambdas[lo] < wl) ++lo;
            }
            else if (first) {
                int32_t hi = n;
                while (lo < hi) { int32_t mid = (lo + hi) >> 1; if (lambdas[mid] < wl) lo = mid + 1; else hi = mid; }
                first = false;
            }
            else {
                while (lo < n && lambdas[lo] < wl) ++lo;
            }
            int32_t lowIdx = max(lo - 1, 0);
            if (lowIdx >= n - 1) return values[n - 1];
            float t = (wl - lambdas[lowIdx]) / (lambdas[lowIdx + 1] - lambdas[lowIdx]);
            if (t <= 0.0f) return values[0];
            return (1 - t) * values[lowIdx] + t * values[lowIdx + 1];
        });
    }
    case SLRHIP_SPECTRUM_UPSAMPLED: {
        // UpsampledContinuousSpectrumTemplate::evaluate, SpectrumTypes.h:314-338: the wavelength loop; the cell lookup and
        // the weights (:241-312) are constants of the spectrum, resolved on the host (slr_amd/spectra.py).
        const uint32_t numPoints = sp.numPoints;
        if (numPoints == 0) return S();
        const uint32_t nw = sp.numSamples;                // 95
        const float w0 = data[0], w1 = data[1], w2 = data[2], w3 = data[3];
        const float4* rec = reinterpret_cast<const float4*>(data + 4);     // [bin][point], 16-byte aligned (checked at upload)
        S ret = S::make([&](int i) {
            float p = (wavelengthOf(i, wlOffset) - 360.0f) / (830.0f - 360.0f);
            float sBinF = p * (float)(nw - 1);
            uint32_t sBin = (uint32_t)sBinF;
            uint32_t sBinNext = (sBin + 1 < nw) ? (sBin + 1) : (nw - 1);
            float t = sBinF - (float)sBin;
            const float4 lo = rec[sBin], hi = rec[sBinNext];
            float r = 0.0f;
            r += w0 * (lo.x * (1 - t) + hi.x * t);
            r += w1 * (lo.y * (1 - t) + hi.y * t);
            r += w2 * (lo.z * (1 - t) + hi.z * t);
            if (numPoints == 4) r += w3 * (lo.w * (1 - t) + hi.w * t);
            return r;
        });
        return ret * sp.scale;
    }
    default:
        return S();
    }
}

// SurfaceMaterial::getBSDF in spectral mode (basic_SurfaceMaterials.cpp:15-43, MicrofacetSurfaceMaterial.cpp:14-28):
// evaluate the constant spectra the lobe needs at this path's wavelengths.
template <class S>
SLR_DEV Mat<S> loadMatSpectral(const DevMaterialS* mats, uint32_t idx, const DevSpectrum* spectra, const float* pool, float wlOffset) {
    const uint4* mq = reinterpret_cast<const uint4*>(mats + idx);
    const uint4 m0 = mq[0], m1 = mq[1];
    DevMaterialS m;
    m.type = m0.x; m.param = __uint_as_float(m0.y); m.onA = __uint_as_float(m0.z); m.onB = __uint_as_float(m0.w);
    m.spec[0] = (int32_t)m1.x; m.spec[1] = (int32_t)m1.y; m.spec[2] = (int32_t)m1.z; m.spec[3] = (int32_t)m1.w;
    Mat<S> r;
    r.type = m.type; r.param = m.param; r.onA = m.onA; r.onB = m.onB;
    if (m.spec[0] >= 0) r.a = 1.0f * evalSpectrum<S>(spectra, pool, m.spec[0], wlOffset);     // scale * spectrum, scale = 1
    if (m.spec[1] >= 0) r.b = evalSpectrum<S>(spectra, pool, m.spec[1], wlOffset);
    if (m.spec[2] >= 0) r.c = evalSpectrum<S>(spectra, pool, m.spec[2], wlOffset);
    return r;
}

// DiffuseEDF::evaluate, EDFs/basic_EDFs.cpp:19-23: (dir.z > 0 ? 1.0f / M_PI : 0.0f) -> float
SLR_DEV float diffuseEDF(V3 dir) { return dir.z > 0.0f ? (float)(1.0 / kPi) : 0.0f; }

// std::max / std::min as the reference uses them (NaN in the second argument yields the first)
SLR_DEV float stdmax(float a, float b) { return (a < b) ? b : a; }
SLR_DEV float stdmin(float a, float b) { return (b < a) ? b : a; }

// OrenNayerBRDF (BSDFs/OrenNayerBRDF.cpp:19-27,46-53): "sin" terms are 1 - z^2 without the square root
template <class S>
SLR_DEV S orenNayar(const Mat<S>& m, V3 dirI, V3 dirO, bool guardNonFinite) {
    float sinThetaI = 1.0f - dirI.z * dirI.z;
    float sinThetaO = 1.0f - dirO.z * dirO.z;
    float absTanThetaI = sinThetaI / fabsf(dirI.z);
    float absTanThetaO = sinThetaO / fabsf(dirO.z);
    float sinAlpha = stdmax(sinThetaI, sinThetaO);
    float tanBeta = stdmin(absTanThetaI, absTanThetaO);
    float cos_dAzimuth = (dirI.x * dirO.x + dirI.y * dirO.y) / (sinThetaI * sinThetaO);
    if (guardNonFinite && !isfinite(cos_dAzimuth)) cos_dAzimuth = 0.0f;
    return m.a * (float)((double)(m.onA + m.onB * stdmax(0.0f, cos_dAzimuth) * sinAlpha * tanBeta) / kPi);
}

// GGX, Core/directional_distribution_functions.cpp:162-268.  The float libm calls (acosf, atan2f, tanf, cosf,
// sinf) are the device library's: they can differ from the host's in the last ulp, so scenes with microfacet
// lobes are compared with a tolerance instead of bit for bit (tests/test_gpu_parity.py).
struct GGX {
    float alpha_g;
    SLR_DEV float evaluate(V3 m) const {                                         // :176-183
        if (m.z <= 0) return 0.0f;
        float theta_m = slrAcos(m.z);
        float cosTheta_m = m.z;
        float tanTheta_m = slrTan(theta_m);
        double c2 = (double)cosTheta_m * (double)cosTheta_m;
        double s = (double)(alpha_g * alpha_g + tanTheta_m * tanTheta_m);
        return (float)((double)(alpha_g * alpha_g) / (kPi * (c2 * c2) * (s * s)));      // std::pow(x, 4), std::pow(x, 2) in double
    }
    SLR_DEV float evaluateSmithG1(V3 v, V3 m) const {                            // :264-268
        float chi = (dot(v, m) / v.z) > 0 ? 1 : 0;
        float theta_v = slrAcos(fminf(1.0f, fmaxf(-1.0f, v.z)));
        double t = (double)(alpha_g * slrTan(theta_v));
        return (float)((double)(chi * 2) / (1 + sqrt(1 + t * t)));
    }
    SLR_DEV float evaluatePDF(V3 v, V3 m) const { return evaluateSmithG1(v, m) * absDot(v, m) * evaluate(m) / fabsf(v.z); }   // :260-262
    SLR_DEV float sample(V3 v, float u0, float u1, V3* m, float* normalPDF) const {     // :191-258
        float alpha_gx = alpha_g, alpha_gy = alpha_g;
        V3 sv = normalize(V3(alpha_gx * v.x, alpha_gy * v.y, v.z));
        float theta_sv = slrAcos(sv.z);
        float phi_sv = slrAtan2(sv.y, sv.x);
        if (sv.z > 0.99999f) { theta_sv = 0.0f; phi_sv = 0.0f; }
        float slope_x, slope_y;
        if ((double)theta_sv < 0.0001) {
            const float r = sqrtf(u0 / (1 - u0));
            const float phi = (float)(2 * kPi * (double)u1);
            slope_x = (float)((double)r * cos((double)phi));
            slope_y = (float)((double)r * sin((double)phi));
        }
        else {
            const float tan_theta_i = (float)tan((double)theta_sv);
            const float a = 1 / tan_theta_i;
            const float G1 = (float)(2 / (1 + sqrt(1.0 + 1.0 / (double)(a * a))));
            const float A = (float)(2.0 * (double)u0 / (double)G1 - 1.0);
            const float tmp = (float)(1.0 / ((double)(A * A) - 1.0));
            const float B = tan_theta_i;
            const float D = sqrtf(B * B * tmp * tmp - (A * A - B * B) * tmp);
            const float slope_x_1 = B * tmp - D;
            const float slope_x_2 = B * tmp + D;
            slope_x = (A < 0 || (double)slope_x_2 > 1.0 / (double)tan_theta_i) ? slope_x_1 : slope_x_2;
            if (u0 == 0) slope_x = 0;
            float S;
            if ((double)u1 > 0.5) { S = 1.0f; u1 = (float)(2.0 * ((double)u1 - 0.5)); }
            else { S = -1.0f; u1 = (float)(2.0 * (0.5 - (double)u1)); }
            const double w = (double)u1;
            const float z = (float)((w * (w * (w * 0.27385 - 0.73369) + 0.46341)) / (w * (w * (w * 0.093073 + 0.309420) - 1.000000) + 0.597999));
            slope_y = (float)((double)(S * z) * sqrt(1.0 + (double)(slope_x * slope_x)));
        }
        float tmp = slrCos(phi_sv) * slope_x - slrSin(phi_sv) * slope_y;
        slope_y = slrSin(phi_sv) * slope_x + slrCos(phi_sv) * slope_y;
        slope_x = tmp;
        slope_x *= alpha_gx;
        slope_y *= alpha_gy;
        *m = normalize(V3(-slope_x, -slope_y, 1));
        float D = evaluate(*m);
        *normalPDF = evaluateSmithG1(v, *m) * absDot(v, *m) * D / fabsf(v.z);
        return D;
    }
};

// FresnelDielectric::evaluate(cosEnter, wlIdx), DDF.cpp:113-129
template <class S>
SLR_DEV float fresnelDielectricWl(const S& etaExt, const S& etaInt, float cosEnter, uint32_t wl) {
    cosEnter = fminf(1.0f, fmaxf(-1.0f, cosEnter));
    bool entering = cosEnter > 0.0f;
    // wl is the index a make() callback was given (this lane's own component when the spectrum is spread over a quad)
    const float eEnter = entering ? etaExt.own((int)wl) : etaInt.own((int)wl);
    const float eExit = entering ? etaInt.own((int)wl) : etaExt.own((int)wl);
    return fresnelDielectric1(eEnter, eExit, sqrtf(fmaxf(0.0f, 1.0f - cosEnter * cosEnter)), fabsf(cosEnter));
}

// One wavelength of the Walter-07 transmission term (MicrofacetBSDF.cpp:170-181, :225-236)
template <class S>
SLR_DEV float mfTransmissionWl(const GGX& D_, const Mat<S>& m, const S& eEnter, const S& eExit, V3 dirOut, V3 dir, uint32_t wl) {
    const float ee = eEnter.own((int)wl), ex = eExit.own((int)wl);
    V3 m_wl = normalize(-(ee * dirOut + ex * dir));
    float dotHV_wl = dot(dirOut, m_wl);
    float dotHL_wl = dot(dir, m_wl);
    float F_wl = fresnelDielectricWl(m.b, m.c, dotHV_wl, wl);
    float G_wl = D_.evaluateSmithG1(dirOut, m_wl) * D_.evaluateSmithG1(dir, m_wl);
    float D_wl = D_.evaluate(m_wl);
    double den = (double)(ee * dotHV_wl + ex * dotHL_wl);
    return (float)((double)(fabsf(dotHV_wl * dotHL_wl) * (1 - F_wl) * G_wl * D_wl) / (den * den));
}

// ---- ModifiedWardDurBRDF (BSDFs/ModifiedWardDurBRDF.cpp:11-87): a = R, param = anisoX, onA = anisoY (the material's param2) ----
// ---- AshikhminShirleyBRDF (BSDFs/AshikhminShirleyBRDF.cpp:12-170): a = Rs, b = Rd, param = nu, onA = nv ---------------------------
// Both are float-libm lobes (exp, log, atan, atan2, acos, pow); std::pow(float, int) in the reference promotes to double.
SLR_DEV double pow5d(float x) { return pow((double)x, 5.0); }
template <class S>
SLR_DEV float wardNumerator(const Mat<S>& m, V3 halfv, V3 dirL, float* dotHI, float* dotHN) {
    float hx_ax = halfv.x / m.param;
    float hy_ay = halfv.y / m.onA;
    *dotHN = fabsf(halfv.z);
    *dotHI = dot(halfv, dirL);
    return slrExp(-(hx_ax * hx_ax + hy_ay * hy_ay) / (*dotHN * *dotHN));
}
template <class S>
SLR_DEV void ashikhminWeights(const Mat<S>& m, uint32_t wl, float absCos, float* specularWeight, float* diffuseWeight) {
    float iRs = importance(m.a, wl);
    float iRd = importance(m.b, wl);
    *specularWeight = (float)((double)iRs + (double)(1 - iRs) * pow5d(1.0f - absCos));
    float transmissionTerm = (float)(1.0 - pow5d(1 - absCos * 0.5f));
    *diffuseWeight = 28 * iRd / 23 * (1 - iRs) * transmissionTerm * transmissionTerm;
}
template <class S>
SLR_DEV float ashikhminCommon(const Mat<S>& m, V3 halfv, float dotHV) {
    float e = (m.param * halfv.x * halfv.x + m.onA * halfv.y * halfv.y) / (1 - halfv.z * halfv.z);
    return (float)((double)sqrtf((m.param + 1) * (m.onA + 1)) / (8 * kPi * (double)dotHV) * (double)slrPow(fabsf(halfv.z), e));
}
template <class S>
SLR_DEV S ashikhminFs(const Mat<S>& m, float commonTerm, float dotHV, float zQuery, float zDir) {
    S F = m.a + (S(1.0f) - m.a) * (float)pow5d(1.0f - dotHV);
    S specular_fs = (commonTerm / fmaxf(fabsf(zQuery), fabsf(zDir))) * F;
    S diffuse_fs = (28.0f * m.b) / (float)(23 * kPi) * (S(1.0f) - m.a) * (float)(1.0 - pow5d(1.0f - fabsf(zQuery) / 2)) *
                   (float)(1.0 - pow5d(1.0f - fabsf(zDir) / 2));
    return specular_fs + diffuse_fs;
}

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
    case SLRHIP_MATERIAL_MICROFACET_METAL: return DT_Reflection | DT_HighFreq;                     // MicrofacetBSDF.h:27-28
    case SLRHIP_MATERIAL_MICROFACET_GLASS: return DT_Reflection | DT_Transmission | DT_HighFreq;   // MicrofacetBSDF.h:44-46
    case SLRHIP_MATERIAL_WARD: return DT_Reflection | DT_HighFreq;                                 // ModifiedWardDurBRDF.h:29
    case SLRHIP_MATERIAL_ASHIKHMIN: return DT_Reflection | DT_HighFreq | DT_LowFreq;               // AshikhminShirleyBRDF.h:29
    default: return 0;
    }
}

// BSDF::sample (DDF.h:231-246) over sampleInternal of LambertianBRDF / SpecularBRDF / SpecularBSDF
// (BSDFs/basic_BSDFs.cpp:12-26, 61-71, 95-149); query.flags = All, adjoint = false.
template <class S, bool MF>
SLR_DEV S bsdfSampleInternal(const Mat<S>& m, uint32_t type, V3 dirOut, V3 gNorm, uint32_t wl, float uComp, float u0, float u1, BsdfSample* res) {
    res->dirPDF = 0.0f;       // every failure path below leaves it so
    S fs_sn;
    switch (m.type) {
    case SLRHIP_MATERIAL_MATTE: {
        // LambertianBRDF basic_BSDFs.cpp:12-26 / OrenNayerBRDF.cpp:12-34: same cosine sample, different value
        res->dir_sn = cosineSampleHemisphere(u0, u1);
        res->dirPDF = (float)((double)res->dir_sn.z / kPi);
        res->dirType = type;
        res->dir_sn.z *= dot(dirOut, gNorm) > 0 ? 1 : -1;
        fs_sn = m.param >= 0.0f ? orenNayar(m, res->dir_sn, dirOut, true) : m.a / (float)kPi;
        break;
    }
    case SLRHIP_MATERIAL_WARD: {
        if (!MF) return S();
        // ModifiedWardDurBRDF::sampleInternal :11-40
        float quad = (float)(2 * kPi * (double)u1);
        float phi_h = slrAtan2(m.onA * slrSin(quad), m.param * slrCos(quad));
        float cosphi_ax = slrCos(phi_h) / m.param;
        float sinphi_ay = slrSin(phi_h) / m.onA;
        float theta_h = slrAtan(sqrtf(-slrLog(1 - u0) / (cosphi_ax * cosphi_ax + sinphi_ay * sinphi_ay)));
        V3 halfv(slrSin(theta_h) * slrCos(phi_h), slrSin(theta_h) * slrSin(phi_h), slrCos(theta_h));
        halfv.z *= dirOut.z > 0 ? 1 : -1;
        res->dir_sn = (2 * dot(dirOut, halfv)) * halfv - dirOut;
        if (res->dir_sn.z * dirOut.z <= 0) { res->dirPDF = 0.0f; return S(); }
        float dotHI, dotHN;
        float numerator = wardNumerator(m, halfv, res->dir_sn, &dotHI, &dotHN);
        float commonDenom = (float)(4 * kPi * (double)m.param * (double)m.onA * (double)dotHI * (double)dotHN * (double)dotHN * (double)dotHN);
        res->dirPDF = numerator / commonDenom;
        res->dirType = type;
        fs_sn = m.a * (numerator / (commonDenom * dotHI * dotHN));
        break;
    }
    case SLRHIP_MATERIAL_ASHIKHMIN: {
        if (!MF) return S();
        // AshikhminShirleyBRDF::sampleInternal :12-92
        float specularWeight, diffuseWeight;
        ashikhminWeights(m, wl, fabsf(dirOut.z), &specularWeight, &diffuseWeight);
        float sumWeights = specularWeight + diffuseWeight;
        float specularDirPDF, diffuseDirPDF;
        if (uComp * sumWeights < specularWeight) {
            res->dirType = DT_Reflection | DT_HighFreq;
            float quad = (float)(2 * kPi * (double)u1);
            float phi_h = slrAtan2(sqrtf(m.param + 1) * slrSin(quad), sqrtf(m.onA + 1) * slrCos(quad));
            float cosphi = slrCos(phi_h);
            float sinphi = slrSin(phi_h);
            float theta_h = slrAcos(slrPow(1 - u0, 1.0f / (m.param * cosphi * cosphi + m.onA * sinphi * sinphi + 1)));
            if (dirOut.z < 0) theta_h = (float)(kPi - (double)theta_h);
            V3 halfv(slrSin(theta_h) * slrCos(phi_h), slrSin(theta_h) * slrSin(phi_h), slrCos(theta_h));
            res->dir_sn = (2 * dot(dirOut, halfv)) * halfv - dirOut;
            if (res->dir_sn.z * dirOut.z <= 0) { res->dirPDF = 0.0f; return S(); }
            float dotHV = dot(halfv, dirOut);
            specularDirPDF = ashikhminCommon(m, halfv, dotHV);
            diffuseDirPDF = (float)((double)fabsf(res->dir_sn.z) / kPi);
            fs_sn = ashikhminFs(m, specularDirPDF, dotHV, dirOut.z, res->dir_sn.z);
        }
        else {
            res->dirType = DT_Reflection | DT_LowFreq;
            res->dir_sn = cosineSampleHemisphere(u0, u1);
            diffuseDirPDF = (float)((double)res->dir_sn.z / kPi);
            res->dir_sn.z *= dot(dirOut, gNorm) > 0 ? 1 : -1;
            V3 halfv = normalize(dirOut + res->dir_sn);
            float dotHV = dot(halfv, dirOut);
            specularDirPDF = ashikhminCommon(m, halfv, dotHV);
            fs_sn = ashikhminFs(m, specularDirPDF, dotHV, dirOut.z, res->dir_sn.z);
        }
        res->dirPDF = (specularDirPDF * specularWeight + diffuseDirPDF * diffuseWeight) / sumWeights;
        break;
    }
    case SLRHIP_MATERIAL_MICROFACET_METAL: {
        if (!MF) return S();      // kernels instantiated for scenes without microfacet lobes carry none of this code
        // MicrofacetBRDF::sampleInternal, BSDFs/MicrofacetBSDF.cpp:11-45
        GGX D_ = {m.param};
        bool entering = dirOut.z >= 0.0f;
        float sign = entering ? 1.0f : -1.0f;
        V3 mm; float mPDF;
        float D = D_.sample(sign * dirOut, u0, u1, &mm, &mPDF);
        float dotHV = dot(dirOut, mm);
        if (dotHV * sign <= 0) return S();
        res->dir_sn = 2 * dotHV * mm - dirOut;
        if (res->dir_sn.z * dirOut.z <= 0) return S();
        float commonPDFTerm = 1.0f / (4 * dotHV * sign);
        res->dirPDF = commonPDFTerm * mPDF;
        res->dirType = type;
        S F = fresnelConductor(m.b, m.c, dotHV);
        float G = D_.evaluateSmithG1(dirOut, mm) * D_.evaluateSmithG1(res->dir_sn, mm);
        fs_sn = F * D * G / (4 * dirOut.z * res->dir_sn.z);
        break;
    }
    case SLRHIP_MATERIAL_MICROFACET_GLASS: {
        if (!MF) return S();
        // MicrofacetBSDF::sampleInternal, MicrofacetBSDF.cpp:113-196 (flags = All, adjoint = false)
        GGX D_ = {m.param};
        bool entering = dirOut.z >= 0.0f;
        float sign = entering ? 1.0f : -1.0f;
        const S eEnter = selectSpectrum(entering, m.b, m.c);     // by value: a reference select would pin m in scratch
        const S eExit = selectSpectrum(entering, m.c, m.b);
        V3 mm; float mPDF;
        float D = D_.sample(sign * dirOut, u0, u1, &mm, &mPDF);
        float dotHV = dot(dirOut, mm);
        if (dotHV * sign <= 0 || isnan(D)) return S();
        S F = fresnelDielectric(m.b, m.c, dotHV);
        float reflectProb = importance(F, wl);
        if (uComp < reflectProb) {
            res->dir_sn = 2 * dotHV * mm - dirOut;
            if (res->dir_sn.z * dirOut.z <= 0) return S();
            float commonPDFTerm = reflectProb / (4 * dotHV * sign);
            res->dirPDF = commonPDFTerm * mPDF;
            res->dirType = DT_Reflection | DT_HighFreq;
            float G = D_.evaluateSmithG1(dirOut, mm) * D_.evaluateSmithG1(res->dir_sn, mm);
            fs_sn = F * D * G / (4 * dirOut.z * res->dir_sn.z);
        }
        else {
            float ee = eEnter.comp(wl), ex = eExit.comp(wl);
            float recRelIOR = ee / ex;
            float innerRoot = 1 + recRelIOR * recRelIOR * (dotHV * dotHV - 1);
            if (innerRoot < 0) return S();
            res->dir_sn = (recRelIOR * dotHV - sign * sqrtf(innerRoot)) * mm - recRelIOR * dirOut;
            if (res->dir_sn.z * dirOut.z >= 0) return S();
            float dotHL = dot(res->dir_sn, mm);
            double den = (double)(ee * dotHV + ex * dotHL);
            float commonPDFTerm = (float)((double)(1 - reflectProb) / (den * den));
            float pdf = commonPDFTerm * mPDF * ex * ex * fabsf(dotHL);
            const V3 dirT = res->dir_sn;
            S ret = S::make([&](int i) { return mfTransmissionWl(D_, m, eEnter, eExit, dirOut, dirT, (uint32_t)i); });
            ret = ret / fabsf(dirOut.z * res->dir_sn.z);
            ret = ret * (eEnter * eEnter);
            res->dirPDF = pdf;
            res->dirType = DT_Transmission | DT_HighFreq;
            fs_sn = ret;
        }
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
        S F = fresnelDielectric(m.b, m.c, dirOut.z);
        float reflectProb = importance(F, wl);
        if (uComp < reflectProb) {
            if (dirOut.z == 0.0f) return S();
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
            if (sinExit2 >= 1.0f) return S();
            float cosExit = sqrtf(fmaxf(0.0f, 1.0f - sinExit2));
            if (entering) cosExit = -cosExit;
            res->dir_sn = V3(rrEta * -dirOut.x, rrEta * -dirOut.y, cosExit);
            res->dirPDF = 1.0f - reflectProb;
            res->dirType = DT_Transmission | DT_Delta0D | ((type & DT_Dispersive) ? (uint32_t)DT_Dispersive : 0u);
            float v = m.a.comp(wl) * (1.0f - F.comp(wl));
            v *= (eEnter * eEnter) / (eExit * eExit);
            S ret = S::make([&](int i) { return (uint32_t)i == wl ? v : 0.0f; });
            fs_sn = ret / fabsf(cosExit);
        }
        break;
    }
    default:
        return S();
    }
    return fs_sn;
}
// `flags` = query.flags: All on this path, except where InverseBSDF hands a flipped query to its base (pt_bsdf_multi.h)
template <class S, bool MF>
SLR_DEV S bsdfSample(const Mat<S>& m, uint32_t type, V3 dirOut, V3 gNorm, uint32_t wl, float uComp, float u0, float u1, BsdfSample* res,
                     uint32_t flags = DT_All) {
    res->dirPDF = 0.0f;
    res->dirType = 0;
    if (!dtMatches(type, flags)) return S();
    S fs_sn = bsdfSampleInternal<S, MF>(m, type, dirOut, gNorm, wl, uComp, u0, u1, res);
    if (res->dirPDF == 0.0f) return S();
    float snCorrection = fabsf(res->dir_sn.z / dot(res->dir_sn, gNorm));
    return fs_sn * snCorrection;
}

// evaluatePDFInternal of the lobes that have a non-delta component (the two-sided ones as for query.flags = All).
template <class S, bool MF>
SLR_DEV float bsdfEvaluatePDFInternal(const Mat<S>& m, V3 dirOut, V3 dir, uint32_t wl) {
    switch (m.type) {
    case SLRHIP_MATERIAL_MATTE:
        // LambertianBRDF::evaluatePDFInternal basic_BSDFs.cpp:41-50 == OrenNayerBRDF.cpp:58-66
        if (dirOut.z * dir.z <= 0.0f) return 0.0f;
        return (float)((double)fabsf(dir.z) / kPi);
    case SLRHIP_MATERIAL_WARD: {
        if (!MF) return 0.0f;
        // ModifiedWardDurBRDF::evaluatePDFInternal :61-77
        if (dir.z * dirOut.z <= 0) return 0.0f;
        V3 halfv = normalize(dirOut + dir);
        float dotHI, dotHN;
        float numerator = wardNumerator(m, halfv, dir, &dotHI, &dotHN);
        float denominator = (float)(4 * kPi * (double)m.param * (double)m.onA * (double)dotHI * (double)dotHN * (double)dotHN * (double)dotHN);
        return numerator / denominator;
    }
    case SLRHIP_MATERIAL_ASHIKHMIN: {
        if (!MF) return 0.0f;
        // AshikhminShirleyBRDF::evaluatePDFInternal :115-153
        if (dir.z * dirOut.z <= 0) return 0.0f;
        V3 halfv = normalize(dirOut + dir);
        float dotHV = dot(halfv, dirOut);
        float specularDirPDF = ashikhminCommon(m, halfv, dotHV);
        float diffuseDirPDF = (float)((double)fabsf(dir.z) / kPi);
        float specularWeight, diffuseWeight;
        ashikhminWeights(m, wl, fabsf(dirOut.z), &specularWeight, &diffuseWeight);
        return (specularDirPDF * specularWeight + diffuseDirPDF * diffuseWeight) / (specularWeight + diffuseWeight);
    }
    case SLRHIP_MATERIAL_MICROFACET_METAL: {
        if (!MF) return 0.0f;
        // MicrofacetBSDF.cpp:73-100
        if (dir.z * dirOut.z <= 0) return 0.0f;
        GGX D_ = {m.param};
        float sign = dirOut.z >= 0.0f ? 1.0f : -1.0f;
        V3 mm = sign * normalize(dirOut + dir);
        float dotHV = dot(dirOut, mm);
        if (dotHV * sign <= 0) return 0.0f;
        float mPDF = D_.evaluatePDF(sign * dirOut, mm);
        float commonPDFTerm = 1.0f / (4 * dotHV * sign);
        return commonPDFTerm * mPDF;
    }
    case SLRHIP_MATERIAL_MICROFACET_GLASS: {
        if (!MF) return 0.0f;
        // MicrofacetBSDF.cpp:249-303
        GGX D_ = {m.param};
        bool entering = dirOut.z >= 0.0f;
        float sign = entering ? 1.0f : -1.0f;
        float dotNVdotNL = dir.z * dirOut.z;
        if (dotNVdotNL == 0) return 0.0f;
        float ee = entering ? m.b.comp(wl) : m.c.comp(wl), ex = entering ? m.c.comp(wl) : m.b.comp(wl);
        V3 mm;
        if (dotNVdotNL > 0) mm = sign * normalize(dirOut + dir);
        else mm = normalize(-(ee * dirOut + ex * dir));
        float dotHV = dot(dirOut, mm);
        if (dotHV * sign <= 0) return 0.0f;
        float mPDF = D_.evaluatePDF(sign * dirOut, mm);
        S F = fresnelDielectric(m.b, m.c, dotHV);
        float reflectProb = importance(F, wl);
        if (dotNVdotNL > 0) {
            float commonPDFTerm = reflectProb / (4 * dotHV * sign);
            return commonPDFTerm * mPDF;
        }
        float dotHL = dot(dir, mm);
        double den = (double)(ee * dotHV + ex * dotHL);
        float commonPDFTerm = (float)((double)(1 - reflectProb) / (den * den));
        return commonPDFTerm * mPDF * ex * ex * fabsf(dotHL);
    }
    default:
        return 0.0f;
    }
}
// BSDF::evaluatePDF (DDF.h:268-279)
template <class S, bool MF>
SLR_DEV float bsdfEvaluatePDF(const Mat<S>& m, uint32_t type, V3 dirOut, V3 dir, uint32_t wl, uint32_t flags = DT_All) {
    if (!dtMatches(type, flags)) return 0.0f;
    return bsdfEvaluatePDFInternal<S, MF>(m, dirOut, dir, wl);
}

SLR_DEV uint32_t sideTest(V3 gNorm, V3 d0, V3 d1) {                                // DDF.h:209-212
    bool reflect = dot(gNorm, d0) * dot(gNorm, d1) > 0;
    return DT_AllFreq | (reflect ? DT_Reflection : DT_Transmission);
}

// evaluateInternal of each lobe; `flags` = the query's flags after the caller's side test.
template <class S, bool MF>
SLR_DEV S bsdfEvaluateInternal(const Mat<S>& m, uint32_t flags, V3 dirOut, V3 dir, uint32_t wl) {
    S fs_sn;
    switch (m.type) {
    case SLRHIP_MATERIAL_MATTE:
        // LambertianBRDF::evaluateInternal basic_BSDFs.cpp:28-39 / OrenNayerBRDF.cpp:36-56
        if (dirOut.z * dir.z <= 0.0f) fs_sn = S();
        else if (m.param >= 0.0f) fs_sn = orenNayar(m, dir, dirOut, false);
        else fs_sn = m.a / (float)kPi;
        break;
    case SLRHIP_MATERIAL_WARD: {
        if (!MF) break;
        // ModifiedWardDurBRDF::evaluateInternal :42-59
        if (dir.z * dirOut.z <= 0) break;
        V3 halfv = normalize(dirOut + dir);
        float dotHI, dotHN;
        float numerator = wardNumerator(m, halfv, dir, &dotHI, &dotHN);
        float denominator = (float)(4 * kPi * (double)m.param * (double)m.onA * (double)dotHI * (double)dotHI * (double)dotHN * (double)dotHN *
                                    (double)dotHN * (double)dotHN);
        fs_sn = m.a * numerator / denominator;
        break;
    }
    case SLRHIP_MATERIAL_ASHIKHMIN: {
        if (!MF) break;
        // AshikhminShirleyBRDF::evaluateInternal :94-113
        if (dir.z * dirOut.z <= 0) break;
        V3 halfv = normalize(dirOut + dir);
        float dotHV = dot(halfv, dirOut);
        fs_sn = ashikhminFs(m, ashikhminCommon(m, halfv, dotHV), dotHV, dirOut.z, dir.z);
        break;
    }
    case SLRHIP_MATERIAL_MICROFACET_METAL: {
        if (!MF) break;
        // MicrofacetBRDF::evaluateInternal MicrofacetBSDF.cpp:47-71
        if (dir.z * dirOut.z <= 0) break;
        GGX D_ = {m.param};
        float sign = dirOut.z >= 0.0f ? 1.0f : -1.0f;
        V3 mm = sign * normalize(dirOut + dir);
        float dotHV = dot(dirOut, mm);
        float D = D_.evaluate(mm);
        S F = fresnelConductor(m.b, m.c, dotHV);
        float G = D_.evaluateSmithG1(dirOut, mm) * D_.evaluateSmithG1(dir, mm);
        fs_sn = F * D * G / (4 * dirOut.z * dir.z);
        break;
    }
    case SLRHIP_MATERIAL_MICROFACET_GLASS: {
        if (!MF) break;
        // MicrofacetBSDF::evaluateInternal MicrofacetBSDF.cpp:198-247 (mQuery.flags after the side test)
        GGX D_ = {m.param};
        bool entering = dirOut.z >= 0.0f;
        float sign = entering ? 1.0f : -1.0f;
        float dotNVdotNL = dir.z * dirOut.z;
        if (dotNVdotNL > 0 && dtMatches(flags, DT_Reflection | DT_AllFreq)) {
            V3 mm = sign * normalize(dirOut + dir);
            float dotHV = dot(dirOut, mm);
            float D = D_.evaluate(mm);
            S F = fresnelDielectric(m.b, m.c, dotHV);
            float G = D_.evaluateSmithG1(dirOut, mm) * D_.evaluateSmithG1(dir, mm);
            fs_sn = F * D * G / (4 * dotNVdotNL);
        }
        else if (dotNVdotNL < 0 && dtMatches(flags, DT_Transmission | DT_AllFreq)) {
            const S eEnter = selectSpectrum(entering, m.b, m.c);
            const S eExit = selectSpectrum(entering, m.c, m.b);
            S ret = S::make([&](int i) { return mfTransmissionWl(D_, m, eEnter, eExit, dirOut, dir, (uint32_t)i); });
            ret = ret / fabsf(dotNVdotNL);
            fs_sn = ret * (eEnter * eEnter);
        }
        break;
    }
    default:   // SpecularBRDF / SpecularBSDF::evaluateInternal return Zero (basic_BSDFs.cpp:73-77,151-155)
        break;
    }
    return fs_sn;
}
// BSDF::evaluate (DDF.h:247-267)
template <class S, bool MF>
SLR_DEV S bsdfEvaluateOnly(const Mat<S>& m, uint32_t type, V3 dirOut, V3 gNorm, V3 dir, uint32_t wl, uint32_t queryFlags = DT_All) {
    uint32_t flags = queryFlags & sideTest(gNorm, dirOut, dir);
    if (!dtMatches(type, flags)) return S();
    S fs_sn = bsdfEvaluateInternal<S, MF>(m, flags, dirOut, dir, wl);
    float snCorrection = fabsf(dir.z / dot(dir, gNorm));
    return fs_sn * snCorrection;
}
// BSDF::evaluate + evaluatePDF for the NEE direction.
template <class S, bool MF>
SLR_DEV S bsdfEvaluate(const Mat<S>& m, uint32_t type, V3 dirOut, V3 gNorm, V3 dir, uint32_t wl, float* pdf) {
    *pdf = bsdfEvaluatePDF<S, MF>(m, type, dirOut, dir, wl);
    return bsdfEvaluateOnly<S, MF>(m, type, dirOut, gNorm, dir, wl);
}

} // namespace slrhip
