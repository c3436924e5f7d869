// host_util.cpp — host-only pieces of the C ABI: the seeding contract, and the image export of
// the reference (ImageSensor::saveImage, Core/ImageSensor.cpp:138-186; saveBMP,
// Helper/bmp_exporter.cpp:13-53) operating on the linear float framebuffer the GPU returns.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/slrhip.h"
#include "cmf16_table.h"
#include "cmf_2deg_table.h"

namespace {

inline uint32_t fmix32(uint32_t h) { h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16; return h; }

// sRGB_gamma, BasicTypes/Spectrum.cpp:15-21 (float instantiation; literals are double)
inline float sRGBGamma(float value) {
    if (value <= 0.0031308) return (float)(12.92 * value);
    return (float)(1.055 * std::pow((double)value, 1.0 / 2.4) - 0.055);
}

// sRGB_degamma, BasicTypes/Spectrum.cpp:24-30 (float instantiation; literals are double)
inline float sRGBDegamma(float value) {
    if (value <= 0.04045) return (float)(value / 12.92);
    return (float)std::pow((value + 0.055) / 1.055, 2.4);
}

} // namespace

extern "C" {

// UpsampledContinuousSpectrumTemplate<float, N> constructor (BasicTypes/SpectrumTypes.h:180-237): the colour-space cascade
// (non-linear sRGB -> sRGB -> XYZ -> xy + brightness), then Upsampling::xy_to_uv (Spectrum.h:136-139).  Double literals in
// float expressions are evaluated in double and stored as float, as the reference's template instantiation does.
int slrhip_upsample(int32_t spectrumType, int32_t colorSpace, float e0, float e1, float e2, float uvs[3]) {
    if (!uvs) return SLRHIP_ERR_INVALID_ARGUMENT;
    float x = 0.0f, y = 0.0f, brightness = 0.0f;
    switch (colorSpace) {
    case SLRHIP_COLORSPACE_SRGB_NONLINEAR:
        e0 = sRGBDegamma(e0); e1 = sRGBDegamma(e1); e2 = sRGBDegamma(e2);
        /* fall through */
    case SLRHIP_COLORSPACE_SRGB: {
        float X, Y, Z;
        if (spectrumType == SLRHIP_SPECTRUMTYPE_REFLECTANCE) {            // sRGB_E_to_XYZ, Spectrum.h:66-71
            X = (float)(0.4969 * e0 + 0.3391 * e1 + 0.1640 * e2);
            Y = (float)(0.2562 * e0 + 0.6782 * e1 + 0.0656 * e2);
            Z = (float)(0.0233 * e0 + 0.1130 * e1 + 0.8637 * e2);
        }
        else if (spectrumType == SLRHIP_SPECTRUMTYPE_ILLUMINANT) {        // sRGB_to_XYZ, Spectrum.h:53-57
            X = (float)(0.4124564 * e0 + 0.3575761 * e1 + 0.1804375 * e2);
            Y = (float)(0.2126729 * e0 + 0.7151522 * e1 + 0.0721750 * e2);
            Z = (float)(0.0193339 * e0 + 0.1191920 * e1 + 0.9503041 * e2);
        }
        else return SLRHIP_ERR_INVALID_ARGUMENT;                            // SLRAssert(false, "Invalid Spectrum Type")
        e0 = X; e1 = Y; e2 = Z;
    }   /* fall through */
    case SLRHIP_COLORSPACE_XYZ:
        brightness = e0 + e1 + e2;
        if (brightness == 0) { uvs[0] = 6; uvs[1] = 4; uvs[2] = 0; return SLRHIP_OK; }
        x = e0 / brightness;
        y = e1 / brightness;
        break;
    case SLRHIP_COLORSPACE_XYY:
        x = e0; y = e1;
        brightness = e2 / e1;
        break;
    default:
        return SLRHIP_ERR_INVALID_ARGUMENT;
    }
    uvs[2] = brightness / 0.009355121400914532f;                            // Upsampling::EqualEnergyReflectance
    uvs[0] = (float)(16.730260708356887 * x + 7.7801960340706 * y - 2.170152247475828);
    uvs[1] = (float)(-7.530081094743006 * x + 16.192422314095225 * y + 1.1125529268825947);
    return SLRHIP_OK;
}

// RGB build of the reference: a sampled spectrum -> the RGBInputSpectrum the scene language creates for it
// (libSLRSceneGraph/API.cpp:1149-1214 spectrum_to_XYZ for regular samples, :1216-1278 for irregular ones; :1326-1369 the
// XYZ -> sRGB conversion by spectrum type and the clamp of negative components).  The integration walks the union of the CMF's
// 1-nm grid and the spectrum's own sample positions with trapezoids and Kahan sums, all in float, then divides by integralCMF
// (BasicTypes/Spectrum.cpp:222-229: the Kahan-summed trapezoid integral of ybar_2deg, evaluated with double literals).
namespace {
struct KahanSum {     // BasicTypes/CompensatedSum.h:15-32
    float result = 0.0f, comp = 0.0f;
    void add(float value) { float cInput = value - comp; float sumTemp = result + cInput; comp = (sumTemp - result) - cInput; result = sumTemp; }
};
float integralCMF() {
    KahanSum cum;
    for (int i = 1; i < 471; ++i) cum.add((float)((kCmf2Deg[1][i - 1] + kCmf2Deg[1][i]) * 1 * 0.5));
    return cum.result;
}
}

int slrhip_spectrum_to_rgb(int32_t spectrumType, const float* lambdas, float minLambda, float maxLambda, const float* values, uint32_t numSamples,
                           float rgb[3]) {
    if (!values || !rgb || numSamples < 2) return SLRHIP_ERR_INVALID_ARGUMENT;
    if (spectrumType < SLRHIP_SPECTRUMTYPE_REFLECTANCE || spectrumType > SLRHIP_SPECTRUMTYPE_IOR) return SLRHIP_ERR_INVALID_ARGUMENT;
    const float WavelengthLowBound = 360.0f, WavelengthHighBound = 830.0f;
    const uint32_t NumCMFSamples = 471;
    const float* xbar_2deg = kCmf2Deg[0];
    const float* ybar_2deg = kCmf2Deg[1];
    const float* zbar_2deg = kCmf2Deg[2];
    const float CMFBinWidth = (WavelengthHighBound - WavelengthLowBound) / (NumCMFSamples - 1);
    const float binWidth = lambdas ? 0.0f : (maxLambda - minLambda) / (numSamples - 1);
    uint32_t curCMFIdx = 0, baseIdx = 0;
    float curWL = WavelengthLowBound;
    float prev_xbarVal = 0, prev_ybarVal = 0, prev_zbarVal = 0, prevValue = 0, halfWidth = 0;
    KahanSum X, Y, Z;
    for (uint32_t guard = 0; guard < 4u * (NumCMFSamples + numSamples); ++guard) {
        float xbarValue, ybarValue, zbarValue;
        if (curWL == WavelengthLowBound + curCMFIdx * CMFBinWidth) {
            xbarValue = xbar_2deg[curCMFIdx]; ybarValue = ybar_2deg[curCMFIdx]; zbarValue = zbar_2deg[curCMFIdx];
            ++curCMFIdx;
        }
        else if (curWL < WavelengthLowBound) {
            // A spectrum that starts below 360 nm (D65 at 300 nm, the metals' tables) makes the reference's walk step BACK from
            // 360 nm to the spectrum's first sample and climb up again; on the way it converts a negative float to unsigned
            // (undefined behaviour; the x86-64 clang build clamps the index to 470 and reads one element past xbar_2deg).
            // That is not reproducible.  Here the colour-matching functions are zero below 360 nm, which is what they are;
            // the detour then contributes only its first, negative-width trapezoid (< 1e-4 of the result).
            xbarValue = ybarValue = zbarValue = 0.0f;
        }
        else {
            uint32_t idx = std::min(uint32_t((curWL - WavelengthLowBound) / CMFBinWidth), NumCMFSamples - 1);
            if (idx + 1 >= NumCMFSamples) idx = NumCMFSamples - 2;          // the reference reads xbar_2deg[idx + 1]; stay inside the table
            float CMFBaseWL = WavelengthLowBound + idx * CMFBinWidth;
            float t = (curWL - CMFBaseWL) / CMFBinWidth;
            xbarValue = (1 - t) * xbar_2deg[idx] + t * xbar_2deg[idx + 1];
            ybarValue = (1 - t) * ybar_2deg[idx] + t * ybar_2deg[idx + 1];
            zbarValue = (1 - t) * zbar_2deg[idx] + t * zbar_2deg[idx + 1];
        }
        float value;
        if (lambdas) {                                                       // irregular samples, API.cpp:1243-1259
            if (curWL < lambdas[0]) value = values[0];
            else if (curWL > lambdas[numSamples - 1]) value = values[numSamples - 1];
            else if (baseIdx < numSamples && curWL == lambdas[baseIdx]) { value = values[baseIdx]; ++baseIdx; }
            else {
                const float* lb = std::lower_bound(lambdas + std::max((int32_t)baseIdx - 1, 0), lambdas + numSamples, curWL);
                uint32_t idx = (uint32_t)std::max(int32_t(lb - lambdas) - 1, 0);
                if (idx + 1 >= numSamples) idx = numSamples - 2;
                float t = (curWL - lambdas[idx]) / (lambdas[idx + 1] - lambdas[idx]);
                value = (1 - t) * values[idx] + t * values[idx + 1];
            }
        }
        else {                                                               // regular samples, API.cpp:1176-1192
            if (curWL < minLambda) value = values[0];
            else if (curWL > maxLambda) value = values[numSamples - 1];
            else if (curWL == minLambda + baseIdx * binWidth) { value = values[baseIdx]; ++baseIdx; }
            else {
                uint32_t idx = std::min(uint32_t((curWL - minLambda) / binWidth), numSamples - 1);
                if (idx + 1 >= numSamples) idx = numSamples - 2;
                float baseWL = minLambda + idx * binWidth;
                float t = (curWL - baseWL) / binWidth;
                value = (1 - t) * values[idx] + t * values[idx + 1];
            }
        }
        float avgValue = (prevValue + value) * 0.5f;
        X.add(avgValue * (prev_xbarVal + xbarValue) * halfWidth);
        Y.add(avgValue * (prev_ybarVal + ybarValue) * halfWidth);
        Z.add(avgValue * (prev_zbarVal + zbarValue) * halfWidth);
        prev_xbarVal = xbarValue; prev_ybarVal = ybarValue; prev_zbarVal = zbarValue;
        prevValue = value;
        float prevWL = curWL;
        const float nextSample = baseIdx < numSamples ? (lambdas ? lambdas[baseIdx] : (minLambda + baseIdx * binWidth)) : INFINITY;
        curWL = std::min(WavelengthLowBound + curCMFIdx * CMFBinWidth, nextSample);
        halfWidth = (curWL - prevWL) * 0.5f;
        if (curCMFIdx == NumCMFSamples) break;
    }
    const float norm = integralCMF();
    const float XYZ[3] = {X.result / norm, Y.result / norm, Z.result / norm};
    if (spectrumType == SLRHIP_SPECTRUMTYPE_ILLUMINANT) {                    // XYZ_to_sRGB, Spectrum.h:59-64
        rgb[0] = (float)(3.2404542 * XYZ[0] - 1.5371385 * XYZ[1] - 0.4985314 * XYZ[2]);
        rgb[1] = (float)(-0.9692660 * XYZ[0] + 1.8760108 * XYZ[1] + 0.0415560 * XYZ[2]);
        rgb[2] = (float)(0.0556434 * XYZ[0] - 0.2040259 * XYZ[1] + 1.0572252 * XYZ[2]);
    }
    else {                                                                   // XYZ_to_sRGB_E, Spectrum.h:73-78 (reflectances and IORs)
        rgb[0] = (float)(2.6897 * XYZ[0] - 1.2759 * XYZ[1] - 0.4138 * XYZ[2]);
        rgb[1] = (float)(-1.0221 * XYZ[0] + 1.9783 * XYZ[1] + 0.0438 * XYZ[2]);
        rgb[2] = (float)(0.0612 * XYZ[0] - 0.2245 * XYZ[1] + 1.1633 * XYZ[2]);
    }
    for (int i = 0; i < 3; ++i) rgb[i] = rgb[i] < 0.0f ? 0.0f : rgb[i];
    return SLRHIP_OK;
}

// The part of UpsampledContinuousSpectrumTemplate::evaluate that depends on (u, v) only (SpectrumTypes.h:241-312): the grid
// cell, the 3 or 4 data points it interpolates and their weights.  Writes the payload slrhip_spectrum (kind UPSAMPLED)
// carries: 4 weights, then SLRHIP_UPSAMPLING_SAMPLES records of 4 floats = the data points' spectra interleaved per bin.
int slrhip_resolve_upsampled(const slrhip_upsampling_tables* t, float u, float v, uint32_t* numPointsOut, float* payload) {
    if (!t || !t->cells || !t->point_uv || !t->point_spectrum || !numPointsOut || !payload || t->grid_width == 0 || t->grid_height == 0)
        return SLRHIP_ERR_INVALID_ARGUMENT;
    const uint32_t nw = SLRHIP_UPSAMPLING_SAMPLES;
    std::memset(payload, 0, sizeof(float) * (4 + 4 * nw));
    *numPointsOut = 0;
    if (u < 0.0f || u >= (float)t->grid_width || v < 0.0f || v >= (float)t->grid_height) return SLRHIP_OK;    // evaluates to zero
    const int32_t ui = (int32_t)u, vi = (int32_t)v;
    const uint8_t* cell = t->cells + (size_t)(ui + (int32_t)t->grid_width * vi) * 8;
    const uint32_t inside = cell[0], numPoints = cell[1];
    const uint8_t* idx = cell + 2;
    if (numPoints > 6) return SLRHIP_ERR_INVALID_ARGUMENT;
    for (uint32_t k = 0; k < (inside ? 4u : numPoints); ++k)
        if (idx[k] >= t->num_points) return SLRHIP_ERR_INVALID_ARGUMENT;
    uint32_t used[4] = {0, 0, 0, 0};
    uint32_t n = 0;
    float* w = payload;
    if (inside) {
        const float s = u - (float)ui, tt = v - (float)vi;
        w[0] = (1 - s) * (1 - tt); w[1] = s * (1 - tt); w[2] = (1 - s) * tt; w[3] = s * tt;
        for (int k = 0; k < 4; ++k) used[k] = idx[k];
        n = 4;
    }
    else if (numPoints >= 2) {
        const float* uv = t->point_uv;
        const float p0x = uv[2 * idx[0]], p0y = uv[2 * idx[0] + 1];
        const float ex = u - p0x, ey = v - p0y;
        float e0x = uv[2 * idx[1]] - p0x, e0y = uv[2 * idx[1] + 1] - p0y;
        float uu = e0x * ey - ex * e0y;
        for (uint32_t i = 1; i < numPoints; ++i) {
            const uint32_t j = idx[i % (numPoints - 1) + 1];
            const float e1x = uv[2 * j] - p0x, e1y = uv[2 * j + 1] - p0y;
            const float vv = ex * e1y - e1x * ey;
            const float area = e0x * e1y - e1x * e0y;
            const float bu = uu / area, bv = vv / area;
            const float bw = 1.0f - bu - bv;
            if ((double)bu < -1e-6 || (double)bv < -1e-6 || (double)bw < -1e-6) {
                uu = -vv;
                e0x = e1x;
                e0y = e1y;
                continue;
            }
            w[0] = bu; w[1] = bv; w[2] = bw;
            used[0] = j; used[1] = idx[i]; used[2] = idx[0];
            n = 3;
            break;
        }
    }
    *numPointsOut = n;
    for (uint32_t k = 0; k < n; ++k)
        for (uint32_t b = 0; b < nw; ++b) payload[4 + 4 * b + k] = t->point_spectrum[(size_t)used[k] * nw + b];
    return SLRHIP_OK;
}

// One xorshift128 stream per (pixel, sample): the reference has one stream per worker thread
// (PathTracingRenderer.cpp:33-38), which is neither reproducible nor parallel; this hash is the
// contract that replaces it (murmur3 finaliser over seed, pass, y, x).
int32_t slrhip_sample_seed(int32_t rngSeed, uint32_t px, uint32_t py, uint32_t pass) {
    uint32_t h = (uint32_t)rngSeed;
    h = fmix32(h ^ (pass * 0x9E3779B1u));
    h = fmix32(h ^ (py * 0x85EBCA77u + 0x165667B1u));
    h = fmix32(h ^ (px * 0xC2B2AE3Du + 0x27D4EB2Fu));
    return (int32_t)h;
}

int slrhip_tonemap_bgr8(const float* fb, int32_t width, int32_t height, int32_t components, float scale, uint8_t* dst,
                        size_t dstBytes) {
    if (!fb || !dst || width <= 0 || height <= 0) return SLRHIP_ERR_INVALID_ARGUMENT;
    if (components != 3 && components != 16) return SLRHIP_ERR_UNSUPPORTED;
    const uint32_t byteWidth = 3u * (uint32_t)width + (uint32_t)width % 4u;      // ImageSensor.cpp:149 (sic)
    if (dstBytes < (size_t)byteWidth * (size_t)height) return SLRHIP_ERR_INVALID_ARGUMENT;
    std::memset(dst, 0, (size_t)byteWidth * (size_t)height);
    for (int32_t i = 0; i < height; ++i) {
        for (int32_t j = 0; j < width; ++j) {
            const float* p = fb + ((size_t)i * width + j) * components;
            float RGB[3];
            if (components == 3) {
                RGB[0] = p[0] * scale; RGB[1] = p[1] * scale; RGB[2] = p[2] * scale;       // pixel(j, i) * scale
            }
            else {
                // DiscretizedSpectrum::getRGB, BasicTypes/SpectrumTypes.h:702-721: 16 storage bins -> XYZ -> sRGB
                float XYZ[3] = {0, 0, 0};
                for (int b = 0; b < 16; ++b) {
                    const float v = p[b] * scale;                                          // pixel(j, i) * scale
                    XYZ[0] += kCmfX16[b] * v;
                    XYZ[1] += kCmfY16[b] * v;
                    XYZ[2] += kCmfZ16[b] * v;
                }
                XYZ[0] /= kIntegralCmf16; XYZ[1] /= kIntegralCmf16; XYZ[2] /= kIntegralCmf16;
                // XYZ_to_sRGB, BasicTypes/Spectrum.h:60-64 (double literals, float operands and results)
                RGB[0] = (float)(3.2404542 * XYZ[0] - 1.5371385 * XYZ[1] - 0.4985314 * XYZ[2]);
                RGB[1] = (float)(-0.9692660 * XYZ[0] + 1.8760108 * XYZ[1] + 0.0415560 * XYZ[2]);
                RGB[2] = (float)(0.0556434 * XYZ[0] - 0.2040259 * XYZ[1] + 1.0572252 * XYZ[2]);
            }
            for (int k = 0; k < 3; ++k) RGB[k] = RGB[k] < 0.0f ? 0.0f : RGB[k];
            float Y = (float)(0.222485 * RGB[0] + 0.716905 * RGB[1] + 0.060610 * RGB[2]);
            float scaleY = Y != 0 ? (1.0f - std::exp(-Y)) / Y : 0.0f;
            for (int k = 0; k < 3; ++k) RGB[k] = std::fmin(scaleY * RGB[k], 1.0f);
            uint8_t* o = dst + (size_t)(height - i - 1) * byteWidth + 3 * (size_t)j;
            o[2] = (uint8_t)(256 * std::fmin(sRGBGamma(RGB[0]), 0.999f));
            o[1] = (uint8_t)(256 * std::fmin(sRGBGamma(RGB[1]), 0.999f));
            o[0] = (uint8_t)(256 * std::fmin(sRGBGamma(RGB[2]), 0.999f));
        }
    }
    return SLRHIP_OK;
}

int slrhip_save_bmp(const char* path, const uint8_t* pixels, int32_t width, int32_t height) {
    if (!path || !pixels || width <= 0 || height <= 0) return SLRHIP_ERR_INVALID_ARGUMENT;
    const uint32_t w = (uint32_t)width, h = (uint32_t)height;
    const uint32_t byteWidth = 3 * w + w % 4;
    const uint32_t kHeader = 14 + 40;
    uint8_t header[kHeader];
    std::memset(header, 0, sizeof(header));
    const uint32_t dataSize = byteWidth * h, fileSize = dataSize + kHeader, dataOffset = kHeader, infoSize = 40, one = 1, zero = 0;
    const uint16_t planes = 1, bits = 24;
    header[0] = 'B'; header[1] = 'M';
    std::memcpy(header + 2, &fileSize, 4);
    std::memcpy(header + 10, &dataOffset, 4);
    std::memcpy(header + 14, &infoSize, 4);
    std::memcpy(header + 18, &w, 4);
    std::memcpy(header + 22, &h, 4);
    std::memcpy(header + 26, &planes, 2);
    std::memcpy(header + 28, &bits, 2);
    std::memcpy(header + 30, &zero, 4);
    std::memcpy(header + 34, &dataSize, 4);
    std::memcpy(header + 38, &one, 4);
    std::memcpy(header + 42, &one, 4);
    FILE* fp = std::fopen(path, "wb");
    if (!fp) return SLRHIP_ERR_INVALID_ARGUMENT;
    std::fwrite(header, 1, kHeader, fp);
    std::fwrite(pixels, 1, dataSize, fp);
    std::fclose(fp);
    return SLRHIP_OK;
}

} // extern "C"
