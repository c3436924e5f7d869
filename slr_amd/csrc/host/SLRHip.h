// SLRHip.h — host-side C++ mirror of the reference's renderer interface for the path-tracing hot path.
//
// Same names, argument meaning and error behaviour as the reference, so a libSLR host program switches by
// changing one `new`:
//   SLR::Renderer            libSLR/Core/Renderer.h:15-19          -> SLRHip::Renderer
//   SLR::RenderSettings      libSLR/Core/RenderSettings.h:15-42    -> SLRHip::RenderSettings (same item enum, same getters)
//   SLR::PathTracingRenderer libSLR/Renderers/PathTracingRenderer.h:16-43 -> SLRHip::PathTracingRenderer(spp)
//   SLR::Scene               libSLR/Core/SurfaceObject.h:239-260   -> SLRHip::Scene, a FLAT scene: the reference's Scene is a
//                            private pointer graph (no accessor for triangles or materials), so the host layer owns the arrays.
// render() drives the C ABI of include/slrhip.h and reproduces what the reference's render() leaves behind:
// "%03u.bmp" after 1, 2, 4, ... passes with scale Brightness / (s + 1), and the stdout line
// "%u samples: %s, %g[s]" (PathTracingRenderer.cpp:83-94).  Failures print to stderr and exit(-1) as
// HostProgram/main.cpp:39-42 does.
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <vector>

#include "../../../include/slrhip.h"

namespace SLRHip {

enum class RenderSettingItem { ImageWidth, ImageHeight, TimeStart, TimeEnd, Brightness, RNGSeed };

class RenderSettings {
    std::map<RenderSettingItem, int32_t> m_int32Values;
    std::map<RenderSettingItem, float> m_floatValues;
public:
    void addItem(RenderSettingItem item, int32_t value) { m_int32Values[item] = value; }
    void addItem(RenderSettingItem item, float value) { m_floatValues[item] = value; }
    int32_t getInt(RenderSettingItem item) const { return m_int32Values.at(item); }
    float getFloat(RenderSettingItem item) const { return m_floatValues.at(item); }
};

// Flat scene owned by the host program (what libSLRSceneGraph would have flattened: TriangleMeshNode.cpp:68-112).
class Scene {
    std::vector<slrhip_vertex> m_vertices;
    std::vector<slrhip_triangle> m_triangles;
    std::vector<slrhip_material> m_materials;
    std::vector<slrhip_spectrum> m_spectra;
    std::vector<float> m_spectrumData;
    std::vector<slrhip_texture> m_textures;
    slrhip_camera m_camera;
    std::vector<float> m_envTexels, m_envImportance;
    slrhip_envmap m_env;
    bool m_hasEnv = false;
    // the Meng-15 upsampling grid and the D65 table (slr_amd/data/upsampling_tables.bin), needed to build spectral-mode spectra
    std::vector<uint8_t> m_gridCells;
    std::vector<float> m_pointUV, m_pointSpectrum, m_d65;
    slrhip_upsampling_tables m_tables;
    bool m_hasTables = false;
public:
    Scene();
    // Loads the tables the spectral build of the reference compiles in (BasicTypes/Spectrum.h:197-575, common_spectra.h:49-55).
    // Returns false (and prints to stderr) if the file is missing or malformed.
    bool loadSpectralTables(const std::string& path);
    // `Spectrum(spType, space, e0, e1, e2)` of the scene language (libSLRSceneGraph/API.cpp:286-441): in the spectral build an
    // UpsampledContinuousSpectrum (SpectrumTypes.h:180-237; its (u, v) cell look-up resolved here, slrhip_resolve_upsampled), in
    // the RGB build the (de-gamma'd) triple.  One record serves both modes.  Needs loadSpectralTables().
    uint32_t addUpsampledSpectrum(int32_t spectrumType, int32_t colorSpace, float e0, float e1, float e2);
    // `Spectrum("ID": "D65") * scale` (API.cpp:405-406,443-462): RegularContinuousSpectrum 300-830 nm, 531 samples, and its
    // RGB-build value through slrhip_spectrum_to_rgb (API.cpp:1149-1214,1326-1347).  One record serves both modes.
    // Needs loadSpectralTables().
    uint32_t addD65Spectrum(float scale);
    // RegularContinuousSpectrum / IrregularContinuousSpectrum from the caller's samples (refractive-index tables, API.cpp:420-441)
    uint32_t addRegularSpectrum(float lambdaMin, float lambdaMax, const float* values, uint32_t numSamples, const float rgb[3]);
    uint32_t addIrregularSpectrum(const float* lambdas, const float* values, uint32_t numSamples, const float rgb[3]);
    uint32_t addVertex(const float position[3], const float normal[3], const float tangent[3], const float texcoord[2]);
    uint32_t addTriangle(uint32_t v0, uint32_t v1, uint32_t v2, uint32_t material);
    uint32_t addSpectrumRGB(float r, float g, float b);
    // A spectrum with its spectral-mode descriptor (UPSAMPLED / REGULAR / IRREGULAR, include/slrhip.h) and sample payload;
    // data_offset is assigned here (16-byte aligned).  RGB mode reads only descriptor.rgb.
    uint32_t addSpectrum(slrhip_spectrum descriptor, const float* payload, uint32_t numPayloadFloats);
    // Image-based environment light (InfiniteSphereSurfaceObject + IBLEmission): lat-long texels [height][width][3], the
    // `scale` of IBLEmission, and the quarter-resolution luminance map ImageSpectrumTexture::createIBLImportanceMap would build.
    void setEnvironment(const float* texels, uint32_t width, uint32_t height, float scale, const float* importance, uint32_t mapWidth,
                        uint32_t mapHeight);
    // Checkerboard textures of libSLR (Textures/checker_board_textures.h) over an OffsetAndScale2DMapping (Core/textures.h:32-42).
    // addCheckerSpectrumTexture returns the VALUE for a material's spectrum slot (SLRHIP_TEXTURE_REF); the other two return the
    // texture index for addMaterial's normalMap / alphaMap (BumpSingleSurfaceObject, Triangle's alpha texture).
    int32_t addCheckerSpectrumTexture(uint32_t spectrum0, uint32_t spectrum1, float offsetX = 0, float offsetY = 0, float scaleX = 1, float scaleY = 1);
    uint32_t addCheckerFloatTexture(float v0, float v1, float offsetX = 0, float offsetY = 0, float scaleX = 1, float scaleY = 1);
    uint32_t addCheckerNormalTexture(float stepWidth, bool reverse, float offsetX = 0, float offsetY = 0, float scaleX = 1, float scaleY = 1);
    uint32_t addMaterial(uint32_t type, int32_t s0, int32_t s1, int32_t s2, float param, int32_t emittance, float param2 = 0.0f,
                         int32_t normalMap = -1, int32_t alphaMap = -1);
    // SurfaceMaterial::createSummedMaterial / createMixedMaterial (constant factor) over two single-lobe materials, either of
    // which may be wrapped as SurfaceMaterial::createInverseMaterial would (libSLR/Core/surface_material.h; API.cpp:583-636).
    uint32_t addSummedMaterial(uint32_t mat0, uint32_t mat1, bool inverse0 = false, bool inverse1 = false, int32_t emittance = -1);
    uint32_t addMixedMaterial(uint32_t mat0, uint32_t mat1, float factor, bool inverse0 = false, bool inverse1 = false, int32_t emittance = -1);
    void setCamera(const slrhip_camera& camera) { m_camera = camera; }
    const slrhip_camera& camera() const { return m_camera; }
    slrhip_scene_desc desc() const;
};

class Renderer {
public:
    virtual ~Renderer() {}
    virtual void render(const Scene& scene, const RenderSettings& settings) const = 0;
};

class PathTracingRenderer : public Renderer {
    uint32_t m_samplesPerPixel;
    int m_device;
    std::string m_outputDir;
    int m_mode;
public:
    // mode: SLRHIP_MODE_RGB or SLRHIP_MODE_SPECTRAL — the reference's compile-time Use_Spectral_Representation switch
    explicit PathTracingRenderer(uint32_t spp, int device = 0, const std::string& outputDir = ".", int mode = SLRHIP_MODE_RGB)
        : m_samplesPerPixel(spp), m_device(device), m_outputDir(outputDir), m_mode(mode) {}
    void render(const Scene& scene, const RenderSettings& settings) const override;
};

} // namespace SLRHip
