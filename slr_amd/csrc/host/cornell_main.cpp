// cornell_main.cpp — a libSLR-style host program (cf. HostProgram/main.cpp:20-62) on the HIP renderer:
// builds the Cornell-box walls + light of TestScenes/Cornell_Box_Spheres.txt:8-107 in C++, fills RenderSettings
// like main.cpp:51-57 and calls renderer->render(scene, settings).
//   usage: cornell_main [spp] [width] [height] [outdir] [rgb|spectral] [tables.bin]
// In "spectral" mode the same scene is built from the scene language's spectra — Spectrum(r, g, b) = an upsampled reflectance,
// Spectrum("ID": "D65") * 4 = a regular spectrum — entirely in C++ (SLRHip::Scene::addUpsampledSpectrum / addD65Spectrum).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <string>

#include "SLRHip.h"
#include "cornell_scene.h"

using namespace SLRHip;

int main(int argc, char** argv) {
    const uint32_t spp = argc > 1 ? (uint32_t)std::atoi(argv[1]) : 16;
    const int width = argc > 2 ? std::atoi(argv[2]) : 256, height = argc > 3 ? std::atoi(argv[3]) : 192;
    const char* outdir = argc > 4 ? argv[4] : ".";
    const bool spectral = argc > 5 && std::string(argv[5]) == "spectral";
    Scene scene;
    if (!scene.loadSpectralTables(argc > 6 ? argv[6] : "slr_amd/data/upsampling_tables.bin")) return -1;
    cornell::build(scene, width, height, spectral);

    RenderSettings settings;                                     // HostProgram/main.cpp:51-57, defaults of API.cpp:1071-1092
    settings.addItem(RenderSettingItem::ImageWidth, (int32_t)width);
    settings.addItem(RenderSettingItem::ImageHeight, (int32_t)height);
    settings.addItem(RenderSettingItem::TimeStart, 0.0f);
    settings.addItem(RenderSettingItem::TimeEnd, 0.0f);
    settings.addItem(RenderSettingItem::Brightness, 1.0f);
    settings.addItem(RenderSettingItem::RNGSeed, (int32_t)1509761209);

    std::unique_ptr<Renderer> renderer(new PathTracingRenderer(spp, 0, outdir, spectral ? SLRHIP_MODE_SPECTRAL : SLRHIP_MODE_RGB));
    renderer->render(scene, settings);
    return 0;
}
