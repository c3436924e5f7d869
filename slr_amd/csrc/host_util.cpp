// host_util.cpp — host-only pieces of the C ABI: the seeding contract, and the image export of
// the reference (ImageSensor::saveImage, Core/ImageSensor.cpp:138-186; saveBMP,
// Helper/bmp_exporter.cpp:13-53) operating on the linear float framebuffer the GPU returns.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/slrhip.h"
#include "cmf16_table.h"

namespace {

inline uint32_t fmix32(uint32_t h) { h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16; return h; }

// sRGB_gamma, BasicTypes/Spectrum.cpp:15-21 (float instantiation; literals are double)
inline float sRGBGamma(float value) {
    if (value <= 0.0031308) return (float)(12.92 * value);
    return (float)(1.055 * std::pow((double)value, 1.0 / 2.4) - 0.055);
}

} // namespace

extern "C" {

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
