// pt_tex.h — the reference's procedural textures on the device (libSLR/Textures/checker_board_textures.{h,cpp}) through a
// Texture2DMapping (Core/textures.h:16-42), and the texture coordinate of a hit (Triangle::intersect interpolates it from the
// ORIGINAL barycentrics, Surface/TriangleMesh.cpp:160-161).  Used by k_shade (spectrum textures, bump) and by the traversal
// kernels (alpha test).
#pragma once
#include <hip/hip_runtime.h>

#include "device_types.h"

namespace slrhip {

// texCoord = b0 * tc0 + b1 * tc1 + b2 * tc2 with b0 = 1 - b1 - b2 (TriangleMesh.cpp:159-161); uvA = (u0, v0, u1, v1), uvB = (u2, v2, -, -)
__device__ __forceinline__ void hitTexCoord(float4 uvA, float4 uvB, float b1, float b2, float* texU, float* texV) {
    const float b0 = 1.0f - b1 - b2;
    *texU = (b0 * uvA.x + b1 * uvA.z) + b2 * uvB.x;
    *texV = (b0 * uvA.y + b1 * uvA.w) + b2 * uvB.y;
}

__device__ __forceinline__ DevTexture loadTexture(const DevTexture* textures, uint32_t idx) {
    const float4* q = reinterpret_cast<const float4*>(textures + idx);
    const float4 a = q[0], b = q[1], c = q[2], d = q[3];
    DevTexture t;
    t.kind = __float_as_uint(a.x); t.ox = a.y; t.oy = a.z; t.sx = a.w;
    t.sy = b.x; t.v0 = b.y; t.v1 = b.z; t.spec0 = (int32_t)__float_as_uint(b.w);
    t.spec1 = (int32_t)__float_as_uint(c.x); t.rgb0[0] = c.y; t.rgb0[1] = c.z; t.rgb0[2] = c.w;
    t.rgb1[0] = d.x; t.rgb1[1] = d.y; t.rgb1[2] = d.z; t.pad = __float_as_uint(d.w);
    return t;
}

// OffsetAndScale2DMapping::map (Core/textures.h:37-41), then the index of CheckerBoard{Spectrum,Float}Texture::evaluate
// (checker_board_textures.h:23,49).  A negative sum would index the reference's two-element array with -1 (undefined): folded.
__device__ __forceinline__ int checkerIndex(const DevTexture& t, float texU, float texV) {
    const float x = (texU + t.ox) * t.sx, y = (texV + t.oy) * t.sy;
    const int idx = ((int)(x * 2) + (int)(y * 2)) % 2;
    return idx < 0 ? -idx : idx;
}

// ImageSpectrumTexture::evaluate's texel address (Textures/image_textures.cpp:14-20) after OffsetAndScale2DMapping::map: nearest
// texel, coordinates wrapped by fmod.  An image texture record carries width / height in spec0 / spec1 and its first texel in pad.
__device__ __forceinline__ uint32_t imageTexel(const DevTexture& t, float texU, float texV) {
    const float x = (texU + t.ox) * t.sx, y = (texV + t.oy) * t.sy;
    float u = fmodf(x, 1.0f);
    float v = fmodf(y, 1.0f);
    u += u < 0 ? 1.0f : 0.0f;
    v += v < 0 ? 1.0f : 0.0f;
    const uint32_t w = (uint32_t)t.spec0, h = (uint32_t)t.spec1;
    const uint32_t px = min((uint32_t)((float)w * u), w - 1u);
    const uint32_t py = min((uint32_t)((float)h * v), h - 1u);
    return t.pad + py * w + px;
}

// CheckerBoardNormal3DTexture::evaluate, checker_board_textures.cpp:16-43 (stepWidth = v0, reverse = v1 != 0); returns the
// UNNORMALISED (uComp, vComp, 1): the caller normalises like the reference does
__device__ __forceinline__ void checkerNormalComponents(const DevTexture& t, float texU, float texV, float* uComp, float* vComp) {
    const float x = (texU + t.ox) * t.sx, y = (texV + t.oy) * t.sy;
    const float halfWidth = t.v0 * 0.5f;
    float uc = 0.0f;
    const float absWrapU = fmodf(fabsf(x), 1.0f);
    if (absWrapU < halfWidth * 0.5f || absWrapU > 1.0f - halfWidth * 0.5f) uc = 1.0f;
    else if (absWrapU > 0.5f - halfWidth * 0.5f && absWrapU < 0.5f + halfWidth * 0.5f) uc = -1.0f;
    float vc = 0.0f;
    const float absWrapV = fmodf(fabsf(y), 1.0f);
    if (absWrapV < halfWidth * 0.5f || absWrapV > 1.0f - halfWidth * 0.5f) vc = 1.0f;
    else if (absWrapV > 0.5f - halfWidth * 0.5f && absWrapV < 0.5f + halfWidth * 0.5f) vc = -1.0f;
    if (absWrapV > 0.5f) uc *= -1;
    if (absWrapU > 0.5f) vc *= -1;
    if (t.v1 != 0.0f) { uc *= -1; vc *= -1; }
    *uComp = uc;
    *vComp = vc;
}

// Triangle::intersect's alpha test (TriangleMesh.cpp:162-167): a hit where the alpha texture evaluates to 0 does not occur.
// alphaTris: two float4 per record: (u0, v0, u1, v1), (u2, v2, texture index, -)
__device__ __forceinline__ bool alphaPasses(const float4* __restrict__ alphaTris, const DevTexture* __restrict__ textures, uint32_t alphaIdx, float b1, float b2) {
    const float4 uvA = alphaTris[(size_t)alphaIdx * 2], uvB = alphaTris[(size_t)alphaIdx * 2 + 1];
    float texU, texV;
    hitTexCoord(uvA, uvB, b1, b2, &texU, &texV);
    const DevTexture t = loadTexture(textures, __float_as_uint(uvB.z));
    return (checkerIndex(t, texU, texV) ? t.v1 : t.v0) != 0.0f;
}

} // namespace slrhip
