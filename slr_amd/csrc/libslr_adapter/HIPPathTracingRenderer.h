// HIPPathTracingRenderer.h — the file a libSLR maintainer adds under libSLR/Renderers/ to put the MI355X path tracer behind
// the reference's own renderer interface:
//
//     class Renderer { virtual void render(const Scene &scene, const RenderSettings &settings) const = 0; }   libSLR/Core/Renderer.h:15-19
//     context.renderer->render(*rawScene, settings)                                                            HostProgram/main.cpp:59
//     setRenderer("method": "PT", ("samples": N,)) -> new PathTracingRenderer(spp)                             libSLRSceneGraph/API.cpp:1015-1020
//
// It is compiled AGAINST THE REFERENCE'S HEADERS (this repository does so only in the container that has /root/reference:
// oracle/ref_build/Makefile adds it to the compiled reference used by the tests), includes no reference source text, and reaches
// the GPU only through the C ABI of include/slrhip.h, loaded with dlopen so that libSLR itself gains no link dependency.
//
// The reference keeps the scene private (SurfaceObject.h:187-204,239-260: no accessor for objects, triangles or materials).  The
// maintainer's version of this file would be a `friend` of the eight classes it reads; here the translation unit is compiled
// with -fno-access-control instead, which reads the same members without editing the reference.
#pragma once
#include <string>
#include <vector>

#include "Core/Renderer.h"
#include "Core/RenderSettings.h"
#include "Core/SurfaceObject.h"

#include "slrhip.h"

namespace SLR {

// The flat scene the C ABI takes (include/slrhip.h), produced from a live SLR::Scene.
struct FlatScene {
    std::vector<slrhip_vertex> vertices;
    std::vector<slrhip_triangle> triangles;
    std::vector<slrhip_material> materials;
    std::vector<slrhip_spectrum> spectra;
    std::vector<float> spectrumData;
    std::vector<slrhip_texture> textures;
    std::vector<slrhip_instance> instances;
    slrhip_camera camera;
    // spectral build: libSLR's own Meng-15 tables (BasicTypes/Spectrum.h:197-575) in the layout of slrhip_upsampling_tables
    std::vector<uint8_t> gridCells;
    std::vector<float> pointUV, pointSpectrum;
    slrhip_upsampling_tables tables;
    slrhip_scene_desc desc() const;
};

// Walks Scene -> SurfaceObjectAggregate -> (accelerator's object list) -> SingleSurfaceObject -> Triangle -> Vertex and the
// material / texture / spectrum objects behind them.  One slrhip_triangle per SingleSurfaceObject, in the order of the Triangle
// objects in memory (= the order libSLRSceneGraph/TriangleMeshNode.cpp:80-112 created them in), which keeps the light list
// (SurfaceObject.cpp:232-249) in the reference's order.  Returns false with a message for anything outside the hot path
// (animated or nested instance transforms, image textures, an environment sphere over an image texture).  A TransformedSurfaceObject
// with a StaticTransform over an aggregate of triangles becomes a slrhip_instance (its triangles form one range per distinct mesh).
// Checkerboard spectrum textures in material slots, a BumpSingleSurfaceObject's checkerboard normal map and a Triangle's
// checkerboard alpha texture become slrhip_texture records.  `resolve` = slrhip_resolve_upsampled of the HIP library (spectral build only; may be null in the RGB build).
typedef int (*slrhip_resolve_upsampled_fn)(const slrhip_upsampling_tables*, float, float, uint32_t*, float*);
bool flattenScene(const Scene &scene, FlatScene* out, std::string* error, slrhip_resolve_upsampled_fn resolve);
// Same, with the helper taken from libslrhip.so (`libraryPath` as in HIPPathTracingRenderer's constructor).
bool flattenSceneWithLibrary(const Scene &scene, FlatScene* out, std::string* error, const std::string &libraryPath);

class SLR_API HIPPathTracingRenderer : public Renderer {
    uint32_t m_samplesPerPixel;
    int m_device;
    std::string m_libraryPath;      // libslrhip.so; empty = $SLRHIP_LIBRARY, else the dynamic loader's search path
public:
    HIPPathTracingRenderer(uint32_t spp, int device = 0, const std::string &libraryPath = "") :
        m_samplesPerPixel(spp), m_device(device), m_libraryPath(libraryPath) { }
    // Same contract as PathTracingRenderer::render (Renderers/PathTracingRenderer.cpp:27-98): fills the camera's ImageSensor,
    // writes NNN.bmp after 1, 2, 4, ... passes through the reference's own ImageSensor::saveImage and prints
    // "%u samples: %s, %g[s]"; failure = message on stderr + exit(-1), like HostProgram/main.cpp:39-42.
    void render(const Scene &scene, const RenderSettings &settings) const override;
};

} // namespace SLR
