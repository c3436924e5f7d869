// pt_device.h — device-side math of the path tracer (HIP, gfx950).
//
// Every function states the reference code it replaces (paths relative to
// /root/reference/libSLR).  The arithmetic follows the reference operation by operation —
// including its float/double mixing through M_PI and unsuffixed literals and "x / s" as
// "x * (1.0f / s)" — and the file is compiled with -ffp-contract=off and IEEE-correct
// division / square root, so a sample's radiance is a pure function of (scene, seed) that
// the CPU oracle reproduces.  Only libm calls (double-precision cos/sin here) can differ
// from the host's in the last bit.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/slrhip.h"
#include "device_types.h"

namespace slrhip {

#define SLR_DEV __device__ __forceinline__

static const double kPi = 3.14159265358979323846;      // M_PI
static const double kPi4 = 0.78539816339744830962;     // M_PI_4
static const float kRayEpsilon = 0.0001f;              // Ray::Epsilon, Core/geometry.cpp:15

// ---- BasicTypes/Vector3.h:17-147 ------------------------------------------------------------
struct V3 {
    float x, y, z;
    SLR_DEV V3() : x(0), y(0), z(0) {}
    SLR_DEV V3(float xx, float yy, float zz) : x(xx), y(yy), z(zz) {}
};
SLR_DEV V3 operator+(V3 a, V3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
SLR_DEV V3 operator-(V3 a, V3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
SLR_DEV V3 operator-(V3 a) { return V3(-a.x, -a.y, -a.z); }
SLR_DEV V3 operator*(V3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
SLR_DEV V3 operator*(float s, V3 a) { return V3(s * a.x, s * a.y, s * a.z); }
SLR_DEV V3 operator/(V3 a, float s) { float r = 1.0f / s; return V3(a.x * r, a.y * r, a.z * r); }   // Vector3.h:31
SLR_DEV float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
SLR_DEV float absDot(V3 a, V3 b) { return fabsf(a.x * b.x + a.y * b.y + a.z * b.z); }
SLR_DEV V3 cross(V3 a, V3 b) { return V3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
SLR_DEV float length(V3 a) { return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }
SLR_DEV float sqLength(V3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
SLR_DEV V3 normalize(V3 a) { float l = length(a); return a / l; }
SLR_DEV V3 ld3(const float* p) { return V3(p[0], p[1], p[2]); }
SLR_DEV V3 xyz(float4 v) { return V3(v.x, v.y, v.z); }

// ---- Core/geometry.h:225-235 ReferenceFrame ---------------------------------------------------
struct Frame {
    V3 x, y, z;
    SLR_DEV V3 toLocal(V3 v) const { return V3(dot(x, v), dot(y, v), dot(z, v)); }
    SLR_DEV V3 fromLocal(V3 v) const {
        return V3(dot(V3(x.x, y.x, z.x), v), dot(V3(x.y, y.y, z.y), v), dot(V3(x.z, y.z, z.z), v));
    }
};

// ---- RGBTemplate<float>, BasicTypes/RGBTypes.h:51-143 -------------------------------------------
struct RGB {
    float r, g, b;
    SLR_DEV RGB() : r(0), g(0), b(0) {}
    SLR_DEV explicit RGB(float v) : r(v), g(v), b(v) {}
    SLR_DEV RGB(float rr, float gg, float bb) : r(rr), g(gg), b(bb) {}
    SLR_DEV float comp(uint32_t i) const { return i == 0 ? r : (i == 1 ? g : b); }
    SLR_DEV float own(int i) const { return comp((uint32_t)i); }      // component i for the index a make() callback was given
    static constexpr int LANES = 1;                                    // lanes that share one spectrum
    SLR_DEV bool isZero() const { return r == 0.0f && g == 0.0f && b == 0.0f; }
    // generic spectrum interface shared with Spec16
    static constexpr int N = 3;
    template <class F> SLR_DEV static RGB make(F f) { return RGB(f(0), f(1), f(2)); }
    SLR_DEV float sum() const { return r + g + b; }                      // 0 + r is exact: same bits as a running sum from 0
};
SLR_DEV RGB operator+(RGB a, RGB b) { return RGB(a.r + b.r, a.g + b.g, a.b + b.b); }
SLR_DEV RGB operator-(RGB a, RGB b) { return RGB(a.r - b.r, a.g - b.g, a.b - b.b); }
SLR_DEV RGB operator*(RGB a, RGB b) { return RGB(a.r * b.r, a.g * b.g, a.b * b.b); }
SLR_DEV RGB operator/(RGB a, RGB b) { return RGB(a.r / b.r, a.g / b.g, a.b / b.b); }
SLR_DEV RGB operator*(RGB a, float s) { return RGB(a.r * s, a.g * s, a.b * s); }
SLR_DEV RGB operator*(float s, RGB a) { return RGB(s * a.r, s * a.g, s * a.b); }
SLR_DEV RGB operator/(RGB a, float s) { float rc = 1.0f / s; return RGB(a.r * rc, a.g * rc, a.b * rc); }   // RGBTypes.h:67
SLR_DEV RGB operator+(RGB a, float s) { return a + RGB(s); }
SLR_DEV RGB operator-(RGB a, float s) { return a - RGB(s); }
SLR_DEV RGB rgb4(const float* p) { return RGB(p[0], p[1], p[2]); }

// ---- SampledSpectrumTemplate<float, 16>, BasicTypes/SpectrumTypes.h:348-554 ------------------------------
// Component access with a run-time index goes through a select chain so that the array stays in registers.
struct Spec16 {
    float c[16];
    static constexpr int N = 16;
    SLR_DEV Spec16() {
#pragma unroll
        for (int i = 0; i < 16; ++i) c[i] = 0.0f;
    }
    SLR_DEV explicit Spec16(float v) {
#pragma unroll
        for (int i = 0; i < 16; ++i) c[i] = v;
    }
    template <class F> SLR_DEV static Spec16 make(F f) {
        Spec16 r;
#pragma unroll
        for (int i = 0; i < 16; ++i) r.c[i] = f(i);
        return r;
    }
    SLR_DEV float comp(uint32_t idx) const {
        // A 4-level select tree on the bits of idx over register copies.  The empty asm makes each copy opaque: without
        // it the optimiser folds a select of two loads (or a chain of idx == i selects) back into ONE load through a
        // selected / indexed pointer, and that pins the whole spectrum — and the struct around it — in scratch memory.
        float t[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) { t[i] = c[i]; asm volatile("" : "+v"(t[i])); }
        const bool b0 = idx & 1u, b1 = idx & 2u, b2 = idx & 4u, b3 = idx & 8u;
        const float p0 = b0 ? t[1] : t[0], p1 = b0 ? t[3] : t[2], p2 = b0 ? t[5] : t[4], p3 = b0 ? t[7] : t[6];
        const float p4 = b0 ? t[9] : t[8], p5 = b0 ? t[11] : t[10], p6 = b0 ? t[13] : t[12], p7 = b0 ? t[15] : t[14];
        const float q0 = b1 ? p1 : p0, q1 = b1 ? p3 : p2, q2 = b1 ? p5 : p4, q3 = b1 ? p7 : p6;
        const float r0 = b2 ? q1 : q0, r1 = b2 ? q3 : q2;
        return b3 ? r1 : r0;
    }
    SLR_DEV float own(int i) const { return c[i]; }                       // inside make(): i is a compile-time constant after unrolling
    static constexpr int LANES = 1;
    SLR_DEV bool isZero() const {
        bool z = true;
#pragma unroll
        for (int i = 0; i < 16; ++i) z = z && (c[i] == 0.0f);
        return z;
    }
    SLR_DEV float sum() const {                                          // SpectrumTypes.h:516-518: running sum from 0
        float s = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) s += c[i];
        return s;
    }
};
#define SLR_SPEC16_OP(op)                                                                                   \
    SLR_DEV Spec16 operator op(const Spec16& a, const Spec16& b) {                                          \
        Spec16 r;                                                                                           \
        _Pragma("unroll") for (int i = 0; i < 16; ++i) r.c[i] = a.c[i] op b.c[i];                           \
        return r;                                                                                           \
    }
SLR_SPEC16_OP(+)
SLR_SPEC16_OP(-)
SLR_SPEC16_OP(*)
SLR_SPEC16_OP(/)
#undef SLR_SPEC16_OP
SLR_DEV Spec16 operator*(const Spec16& a, float s) { return Spec16::make([&](int i) { return a.c[i] * s; }); }
SLR_DEV Spec16 operator*(float s, const Spec16& a) { return Spec16::make([&](int i) { return a.c[i] * s; }); }   // SpectrumTypes.h:400-405: c.values[i] * s
SLR_DEV Spec16 operator/(const Spec16& a, float s) { float r = 1 / s; return Spec16::make([&](int i) { return a.c[i] * r; }); }   // :393-399
SLR_DEV Spec16 operator+(const Spec16& a, float s) { return a + Spec16(s); }
SLR_DEV Spec16 operator-(const Spec16& a, float s) { return a - Spec16(s); }

// ---- float libm on the device ------------------------------------------------------------------------------------------------
// The reference calls glibc's float functions; the device library's answers differ from glibc's in the last bit on a share of
// the arguments.  SLR_LIBM_DOUBLE (a bit mask, variant builds and the default below) evaluates a function through the device's
// DOUBLE routine and rounds once — the correctly rounded float, which is what glibc returns except where glibc itself is off
// by one (measured on 2 x 10^7 arguments per function against glibc 2.35, the image's: expf 0.06 %, powf 0.12 %, logf 0.3 %,
// sinf / cosf 1.3 %, tanf 3.7 %, atanf 4.3 %, acosf 7.8 %, atan2f 16 % — profiles/r03_h_libm_via_double.txt has the GPU side).
//   1 sin / cos   2 tan   4 acos   8 atan / atan2   16 exp / log   32 pow
#ifndef SLR_LIBM_DOUBLE
#define SLR_LIBM_DOUBLE 0
#endif
SLR_DEV float slrSin(float x) { return (SLR_LIBM_DOUBLE & 1) ? (float)sin((double)x) : sinf(x); }
SLR_DEV float slrCos(float x) { return (SLR_LIBM_DOUBLE & 1) ? (float)cos((double)x) : cosf(x); }
SLR_DEV float slrTan(float x) { return (SLR_LIBM_DOUBLE & 2) ? (float)tan((double)x) : tanf(x); }
SLR_DEV float slrAcos(float x) { return (SLR_LIBM_DOUBLE & 4) ? (float)acos((double)x) : acosf(x); }
SLR_DEV float slrAtan(float x) { return (SLR_LIBM_DOUBLE & 8) ? (float)atan((double)x) : atanf(x); }
SLR_DEV float slrAtan2(float y, float x) { return (SLR_LIBM_DOUBLE & 8) ? (float)atan2((double)y, (double)x) : atan2f(y, x); }
SLR_DEV float slrExp(float x) { return (SLR_LIBM_DOUBLE & 16) ? (float)exp((double)x) : expf(x); }
SLR_DEV float slrLog(float x) { return (SLR_LIBM_DOUBLE & 16) ? (float)log((double)x) : logf(x); }
SLR_DEV float slrPow(float x, float y) { return (SLR_LIBM_DOUBLE & 32) ? (float)pow((double)x, (double)y) : powf(x, y); }

// importance(): RGBTypes.h:103-108 / SpectrumTypes.h:512-526 (marginal = (1 - primary) / (N - 1); N = 3 gives / 2)
template <class S>
SLR_DEV float importance(const S& s, uint32_t selectedLambda) {
    float sum = s.sum();
    const float primary = 0.9f;
    const float marginal = (1 - primary) / (S::N - 1);
    return sum * marginal + s.comp(selectedLambda) * (primary - marginal);
}

SLR_DEV RGB selectSpectrum(bool pick, const RGB& a, const RGB& b) { return RGB(pick ? a.r : b.r, pick ? a.g : b.g, pick ? a.b : b.b); }
SLR_DEV Spec16 selectSpectrum(bool pick, const Spec16& a, const Spec16& b) {
    return Spec16::make([&](int i) { return pick ? a.c[i] : b.c[i]; });
}

// BasicTypes/CompensatedSum.h:24-30
template <class S>
SLR_DEV void kahanAdd(S& result, S& comp, const S& value) {
    S cInput = value - comp;
    S sumTemp = result + cInput;
    comp = (sumTemp - result) - cInput;
    result = sumTemp;
}

// ---- RNGs/XORShiftRNG.cpp:21-36, Core/RandomNumberGenerator.cpp:12-15 ----------------------------
struct Rng {
    uint32_t s0, s1, s2, s3;
    SLR_DEV uint32_t next() {
        uint32_t t = s0 ^ (s0 << 11);
        s0 = s1; s1 = s2; s2 = s3;
        return s3 = (s3 ^ (s3 >> 19)) ^ (t ^ (t >> 8));
    }
    SLR_DEV float nextFloat() { return __uint_as_float((next() >> 9) | 0x3f800000u) - 1.0f; }
    SLR_DEV void seed(int32_t seed) {
        // `seed` is signed in the reference: arithmetic shift, unsigned multiply
        uint32_t v;
        v = 1812433253U * ((uint32_t)seed ^ (uint32_t)(seed >> 30)) + 0u; s0 = v; seed = (int32_t)v;
        v = 1812433253U * ((uint32_t)seed ^ (uint32_t)(seed >> 30)) + 1u; s1 = v; seed = (int32_t)v;
        v = 1812433253U * ((uint32_t)seed ^ (uint32_t)(seed >> 30)) + 2u; s2 = v; seed = (int32_t)v;
        v = 1812433253U * ((uint32_t)seed ^ (uint32_t)(seed >> 30)) + 3u; s3 = v;
        for (int i = 0; i < 50; ++i) next();
    }
};

// The per-(pixel, sample) seeding contract (include/slrhip.h: slrhip_sample_seed).
SLR_DEV uint32_t fmix32(uint32_t h) { h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16; return h; }
SLR_DEV int32_t sampleSeed(int32_t rngSeed, uint32_t px, uint32_t py, uint32_t pass) {
    uint32_t h = (uint32_t)rngSeed;
    h = fmix32(h ^ (pass * 0x9E3779B1u));
    h = fmix32(h ^ (py * 0x85EBCA77u + 0x165667B1u));
    h = fmix32(h ^ (px * 0xC2B2AE3Du + 0x27D4EB2Fu));
    return (int32_t)h;
}

// Double-precision sin and cos for |x| <= 2 pi (the concentric map only produces theta in
// [-pi/4, 7 pi/4]).  The reference calls the host libm's double cos/sin and rounds the product
// r * cos(theta) to float, so any double result within a few 1e-16 of the true value rounds to the
// same float except when the exact product lies within ~1e-9 ulp_float of a rounding boundary.
// Two-term Cody-Waite reduction by pi/2 and the fdlibm kernel polynomials (< 1 ulp_double);
// about 30 f64 operations instead of the general-range library sincos.
SLR_DEV void sincosQuarterTurns(double x, double* sn, double* cs) {
    const double k = rint(x * 6.36619772367581382433e-01);                 // x * 2/pi
    double r = fma(-k, 1.57079632673412561417e+00, x);                      // pio2_1  (33 bits of pi/2)
    r = fma(-k, 6.07710050650619224932e-11, r);                             // pio2_1t (pi/2 - pio2_1)
    const double z = r * r;
    // __kernel_sin
    double ps = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = fma(z, ps, 2.75573137070700676789e-06);
    ps = fma(z, ps, -1.98412698298579493134e-04);
    ps = fma(z, ps, 8.33333333332248946124e-03);
    const double sinr = fma(z * r, fma(z, ps, -1.66666666666666324348e-01), r);
    // __kernel_cos
    double pc = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = fma(z, pc, -2.75573143513906633035e-07);
    pc = fma(z, pc, 2.48015872894767294178e-05);
    pc = fma(z, pc, -1.38888888888741095749e-03);
    pc = fma(z, pc, 4.16666666666666019037e-02);
    const double cosr = fma(z * z, pc, fma(z, -0.5, 1.0));
    const int q = (int)k & 3;
    const double s0 = (q & 1) ? cosr : sinr;
    const double c0 = (q & 1) ? sinr : cosr;
    *sn = (q & 2) ? -s0 : s0;
    *cs = (q == 1 || q == 2) ? -c0 : c0;
}

// ---- Core/distributions.cpp:37-70 (float instantiation) --------------------------------------------
// theta *= M_PI_4 is a double multiply rounded to float; cos/sin are the double functions and
// r * cos(theta) a double product rounded to float.
SLR_DEV void concentricSampleDisk(float u0, float u1, float* dx, float* dy) {
    float r, theta;
    float sx = 2 * u0 - 1;
    float sy = 2 * u1 - 1;
    if (sx == 0 && sy == 0) { *dx = 0; *dy = 0; return; }
    if (sx >= -sy) {
        if (sx > sy) { r = sx; theta = sy / sx; }
        else { r = sy; theta = 2 - sx / sy; }
    }
    else {
        if (sx > sy) { r = -sy; theta = 6 + sx / sy; }
        else { r = -sx; theta = 4 + sy / sx; }
    }
    theta = (float)((double)theta * kPi4);
    double s, c;
    sincosQuarterTurns((double)theta, &s, &c);
    *dx = (float)((double)r * c);
    *dy = (float)((double)r * s);
}
// Core/distributions.h:26-33
SLR_DEV V3 cosineSampleHemisphere(float u0, float u1) {
    float x, y;
    concentricSampleDisk(u0, u1, &x, &y);
    return V3(x, y, sqrtf(fmaxf(0.0f, 1.0f - x * x - y * y)));
}

// ---- Fresnel, Core/directional_distribution_functions.cpp:68-159 -------------------------------------
template <class S>
SLR_DEV S fresnelConductor(const S& eta, const S& k, float cosEnter) {   // :68-78
    cosEnter = fabsf(cosEnter);
    float cosEnter2 = cosEnter * cosEnter;
    S _2EtaCosEnter = (2.0f * eta) * cosEnter;
    S tmp_f = eta * eta + k * k;
    S tmp = tmp_f * cosEnter2;
    S Rparl2 = (tmp - _2EtaCosEnter + 1.0f) / (tmp + _2EtaCosEnter + 1.0f);
    S Rperp2 = (tmp_f - _2EtaCosEnter + cosEnter2) / (tmp_f + _2EtaCosEnter + cosEnter2);
    return (Rparl2 + Rperp2) / 2.0f;
}
SLR_DEV float fresnelEvalF(float etaEnter, float etaExit, float cosEnter, float cosExit) {   // :155-159
    float Rparl = ((etaExit * cosEnter) - (etaEnter * cosExit)) / ((etaExit * cosEnter) + (etaEnter * cosExit));
    float Rperp = ((etaEnter * cosEnter) - (etaExit * cosExit)) / ((etaEnter * cosEnter) + (etaExit * cosExit));
    return (Rparl * Rparl + Rperp * Rperp) / 2.0f;
}
SLR_DEV float fresnelDielectric1(float eEnter, float eExit, float sinTerm, float cosEnterAbs) {
    float sinExit = eEnter / eExit * sinTerm;
    if (sinExit >= 1.0f) return 1.0f;
    float cosExit = sqrtf(fmaxf(0.0f, 1.0f - sinExit * sinExit));
    return fresnelEvalF(eEnter, eExit, cosEnterAbs, cosExit);
}
template <class S>
SLR_DEV S fresnelDielectric(const S& etaExt, const S& etaInt, float cosEnter) {   // :90-111
    cosEnter = fminf(1.0f, fmaxf(-1.0f, cosEnter));
    bool entering = cosEnter > 0.0f;
    float sinTerm = sqrtf(fmaxf(0.0f, 1.0f - cosEnter * cosEnter));
    float c = fabsf(cosEnter);
    return S::make([&](int i) {
        float eEnter = entering ? etaExt.own(i) : etaInt.own(i);
        float eExit = entering ? etaInt.own(i) : etaExt.own(i);
        return fresnelDielectric1(eEnter, eExit, sinTerm, c);
    });
}

// ---- DirectionType flags, Core/directional_distribution_functions.h:18-91 ------------------------------
enum : uint32_t {
    DT_LowFreq = 1 << 0, DT_HighFreq = 1 << 1, DT_Delta0D = 1 << 2, DT_Delta1D = 1 << 3,
    DT_NonDelta = DT_LowFreq | DT_HighFreq, DT_Delta = DT_Delta0D | DT_Delta1D, DT_AllFreq = DT_NonDelta | DT_Delta,
    DT_Reflection = 1 << 4, DT_Transmission = 1 << 5, DT_WholeSphere = DT_Reflection | DT_Transmission,
    DT_All = DT_AllFreq | DT_WholeSphere, DT_Dispersive = 1 << 6
};
SLR_DEV bool dtMatches(uint32_t type, uint32_t t) { uint32_t res = type & t; return (res & DT_WholeSphere) && (res & DT_AllFreq); }
SLR_DEV bool dtIsDelta(uint32_t v) { return (v & DT_Delta) && !(v & DT_NonDelta); }

// ---- Matrix4x4.h:71-81, Transform.h:47-52 (column-major m[c*4+r]) ---------------------------------------
SLR_DEV V3 mulPoint(const float* m, V3 p) {
    float x = m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12] * 1.0f;
    float y = m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13] * 1.0f;
    float z = m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14] * 1.0f;
    float w = m[3] * p.x + m[7] * p.y + m[11] * p.z + m[15] * 1.0f;
    if (w != 1.0f) { float r = 1.0f / w; x *= r; y *= r; z *= r; }
    return V3(x, y, z);
}
SLR_DEV V3 mulVector(const float* m, V3 v) {
    return V3(m[0] * v.x + m[4] * v.y + m[8] * v.z, m[1] * v.x + m[5] * v.y + m[9] * v.z, m[2] * v.x + m[6] * v.y + m[10] * v.z);
}
SLR_DEV V3 mulNormal(const float* mi, V3 n) {
    return V3(mi[0] * n.x + mi[1] * n.y + mi[2] * n.z, mi[4] * n.x + mi[5] * n.y + mi[6] * n.z, mi[8] * n.x + mi[9] * n.y + mi[10] * n.z);
}

} // namespace slrhip
