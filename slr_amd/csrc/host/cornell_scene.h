// cornell_scene.h — the Cornell-box walls + light + camera of TestScenes/Cornell_Box_Spheres.txt:8-107,132-138 built in C++ on a
// SLRHip::Scene (shared by the example host programs cornell_main.cpp and reduce_main.cpp).
#pragma once
#include <cmath>
#include <string>

#include "SLRHip.h"

namespace cornell {
using namespace SLRHip;

static inline float degamma(float v) { return v <= 0.04045f ? (float)(v / 12.92) : (float)std::pow((v + 0.055) / 1.055, 2.4); }

static inline void quad(Scene& s, const float c[4][3], const float n[3], const float t[3], uint32_t mat) {
    const float uv[4][2] = {{0, 0}, {1, 0}, {1, 1}, {0, 1}};
    uint32_t v[4];
    for (int i = 0; i < 4; ++i) v[i] = s.addVertex(c[i], n, t, uv[i]);
    s.addTriangle(v[0], v[1], v[2], mat);
    s.addTriangle(v[0], v[2], v[3], mat);
}


// spectral: the scene language's spectra — Spectrum(r, g, b) = an upsampled reflectance, Spectrum("ID": "D65") * 4 = a regular
// spectrum — built in C++ (SLRHip::Scene::addUpsampledSpectrum / addD65Spectrum).  Both builds need scene.loadSpectralTables
// first: the RGB build converts the D65 table to its RGB value (slrhip_spectrum_to_rgb).
static inline void build(Scene& scene, int width, int height, bool spectral) {
    auto reflectance = [&](float r, float g, float b) {
        return spectral ? scene.addUpsampledSpectrum(SLRHIP_SPECTRUMTYPE_REFLECTANCE, SLRHIP_COLORSPACE_SRGB_NONLINEAR, r, g, b)
                        : scene.addSpectrumRGB(degamma(r), degamma(g), degamma(b));
    };
    auto matte = [&](float r, float g, float b) {
        return scene.addMaterial(SLRHIP_MATERIAL_MATTE, (int32_t)reflectance(r, g, b), -1, -1, -1.0f, -1);
    };
    const uint32_t red = matte(0.75f, 0.25f, 0.25f), blue = matte(0.25f, 0.25f, 0.75f), white = matte(0.75f, 0.75f, 0.75f);
    const uint32_t emit = scene.addD65Spectrum(4.0f);      // Spectrum("ID": "D65") * 4 in either build (needs loadSpectralTables)
    const uint32_t light = scene.addMaterial(SLRHIP_MATERIAL_MATTE, (int32_t)reflectance(0.9f, 0.9f, 0.9f), -1, -1, -1.0f, (int32_t)emit);
    const float L[4][3] = {{-1.5f, 0, 2.55f}, {-1.5f, 0, -2.55f}, {-1.5f, 2.5f, -2.55f}, {-1.5f, 2.5f, 2.55f}};
    const float R[4][3] = {{1.5f, 0, -2.55f}, {1.5f, 0, 2.55f}, {1.5f, 2.5f, 2.55f}, {1.5f, 2.5f, -2.55f}};
    const float F[4][3] = {{-1.5f, 0, 2.55f}, {1.5f, 0, 2.55f}, {1.5f, 0, -2.55f}, {-1.5f, 0, -2.55f}};
    const float I[4][3] = {{-1.5f, 0, -2.55f}, {1.5f, 0, -2.55f}, {1.5f, 2.5f, -2.55f}, {-1.5f, 2.5f, -2.55f}};
    const float C[4][3] = {{-1.5f, 2.5f, -2.55f}, {1.5f, 2.5f, -2.55f}, {1.5f, 2.5f, 2.55f}, {-1.5f, 2.5f, 2.55f}};
    const float E[4][3] = {{-0.5f, 2.499f, -0.5f}, {0.5f, 2.499f, -0.5f}, {0.5f, 2.499f, 0.5f}, {-0.5f, 2.499f, 0.5f}};
    const float px[3] = {1, 0, 0}, nx[3] = {-1, 0, 0}, py[3] = {0, 1, 0}, ny[3] = {0, -1, 0}, pz[3] = {0, 0, 1}, nz[3] = {0, 0, -1};
    quad(scene, L, px, nz, red); quad(scene, R, nx, pz, blue); quad(scene, F, py, px, white);
    quad(scene, I, pz, px, white); quad(scene, C, ny, px, white); quad(scene, E, ny, px, light);

    // camera of Cornell_Box_Spheres.txt:132-138: translate(0, 1.689714, 6.70284) * rotateY(pi) * rotateX(0.0563936)
    slrhip_camera cam;
    const double a = 0.0563936, ca = std::cos(a), sa = std::sin(a), cb = std::cos(3.1415926536), sb = std::sin(3.1415926536);
    const double m[3][3] = {{cb, sb * sa, sb * ca}, {0, ca, -sa}, {-sb, cb * sa, cb * ca}};       // Ry * Rx
    const double t[3] = {0.0, 1.689714, 6.70284};
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 4; ++r) {
            double v = (r < 3 && c < 3) ? m[r][c] : (r < 3 && c == 3 ? t[r] : (r == c ? 1.0 : 0.0));
            cam.local_to_world[c * 4 + r] = (float)v;
            double vi = 0.0;                                   // inverse of a rigid transform: [R^T | -R^T t]
            if (r < 3 && c < 3) vi = m[c][r];
            else if (r < 3 && c == 3) vi = -(m[0][r] * t[0] + m[1][r] * t[1] + m[2][r] * t[2]);
            else if (r == c) vi = 1.0;
            cam.world_to_local[c * 4 + r] = (float)vi;
        }
    cam.aspect = (float)width / (float)height; cam.fov_y = 0.4807705238f; cam.lens_radius = 0.025f;
    cam.img_plane_distance = 1.0f; cam.obj_plane_distance = 6.3f; cam.sensitivity = 0.0f;
    scene.setCamera(cam);

}

} // namespace cornell
