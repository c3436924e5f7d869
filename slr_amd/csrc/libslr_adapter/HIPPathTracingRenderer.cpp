// HIPPathTracingRenderer.cpp — see HIPPathTracingRenderer.h.  Compiled against the reference's headers with
// -fno-access-control (standing in for the friend declarations a maintainer would add); contains no reference source text.
#include "HIPPathTracingRenderer.h"

#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <tuple>
#include <typeinfo>

#include "Accelerator/QBVH.h"
#include "Accelerator/SBVH.h"
#include "Accelerator/StandardBVH.h"
#include "BasicTypes/Spectrum.h"
#include "Cameras/PerspectiveCamera.h"
#include "Core/ImageSensor.h"
#include "Core/Transform.h"
#include "Core/surface_material.h"
#include "Surface/TriangleMesh.h"
#include "SurfaceMaterials/AshikhminShirleyReflection.h"
#include "SurfaceMaterials/DiffuseEmission.h"
#include "SurfaceMaterials/MicrofacetSurfaceMaterial.h"
#include "SurfaceMaterials/MixedSurfaceMaterial.h"
#include "SurfaceMaterials/ModifiedWardDurReflection.h"
#include "SurfaceMaterials/SummedSurfaceMaterial.h"
#include "SurfaceMaterials/basic_SurfaceMaterials.h"
#include "Textures/checker_board_textures.h"
#include "Textures/constant_textures.h"

namespace SLR {

slrhip_scene_desc FlatScene::desc() const {
    slrhip_scene_desc d;
    std::memset(&d, 0, sizeof(d));
    d.vertices = vertices.data(); d.num_vertices = (uint32_t)vertices.size();
    d.triangles = triangles.data(); d.num_triangles = (uint32_t)triangles.size();
    d.materials = materials.data(); d.num_materials = (uint32_t)materials.size();
    d.spectra = spectra.data(); d.num_spectra = (uint32_t)spectra.size();
    d.spectrum_data = spectrumData.data(); d.num_spectrum_data = (uint32_t)spectrumData.size();
    d.camera = camera;
    d.env = nullptr;
    d.upsampling = gridCells.empty() ? nullptr : &tables;
    d.textures = textures.empty() ? nullptr : textures.data();
    d.num_textures = (uint32_t)textures.size();
    d.instances = instances.empty() ? nullptr : instances.data();
    d.num_instances = (uint32_t)instances.size();
    return d;
}

namespace {

struct Flattener {
    FlatScene* out;
    std::string* error;
    slrhip_resolve_upsampled_fn resolve;
    std::map<const Vertex*, uint32_t> vertexIndex;
    std::map<const InputSpectrum*, int32_t> spectrumIndex;
    std::map<const SurfaceMaterial*, uint32_t> materialIndex;
    std::map<const void*, uint32_t> textureIndex;
    // a material record is keyed by (material, normal map, alpha texture): the material group of TriangleMeshNode pairs them
    std::map<std::tuple<const SurfaceMaterial*, const Normal3DTexture*, const FloatTexture*>, uint32_t> groupIndex;

    bool fail(const std::string &what) { *error = "HIPPathTracingRenderer: " + what; return false; }

    bool mappingOf(const Texture2DMapping* mapping, slrhip_texture* rec) {
        rec->offset[0] = rec->offset[1] = 0.0f; rec->scale[0] = rec->scale[1] = 1.0f;
        if (const OffsetAndScale2DMapping* os = dynamic_cast<const OffsetAndScale2DMapping*>(mapping)) {
            rec->offset[0] = os->m_offsetX; rec->offset[1] = os->m_offsetY; rec->scale[0] = os->m_scaleX; rec->scale[1] = os->m_scaleY;
            return true;
        }
        if (mapping && typeid(*mapping) == typeid(Texture2DMapping)) return true;       // the default mapping: texCoord itself
        return fail("only the default and the offset-and-scale 2D texture mappings are supported");
    }
    uint32_t pushTexture(const void* key, const slrhip_texture& rec) {
        out->textures.push_back(rec);
        textureIndex[key] = (uint32_t)out->textures.size() - 1;
        return (uint32_t)out->textures.size() - 1;
    }

    // a SpectrumTexture in a material slot -> the slot's value: a spectrum index (ConstantSpectrumTexture) or SLRHIP_TEXTURE_REF
    // (CheckerBoardSpectrumTexture over two constant spectra)
    bool spectrumOf(const SpectrumTexture* tex, int32_t* index) {
        *index = -1;
        if (!tex) return true;
        if (const CheckerBoardSpectrumTexture* cb = dynamic_cast<const CheckerBoardSpectrumTexture*>(tex)) {
            auto it = textureIndex.find(cb);
            if (it == textureIndex.end()) {
                slrhip_texture rec;
                std::memset(&rec, 0, sizeof(rec));
                rec.kind = SLRHIP_TEXTURE_CHECKER_SPECTRUM;
                if (!mappingOf(cb->m_mapping, &rec)) return false;
                for (int k = 0; k < 2; ++k) if (!inputSpectrumOf(cb->m_values[k], &rec.spectrum[k])) return false;
                *index = SLRHIP_TEXTURE_REF(pushTexture(cb, rec));
            }
            else *index = SLRHIP_TEXTURE_REF(it->second);
            return true;
        }
        const ConstantSpectrumTexture* c = dynamic_cast<const ConstantSpectrumTexture*>(tex);
        if (!c) return fail("only constant and checkerboard spectrum textures are on the hot path (image textures are not)");
        return inputSpectrumOf(c->m_value, index);
    }

    bool inputSpectrumOf(const InputSpectrum* sp, int32_t* index) {
        auto it = spectrumIndex.find(sp);
        if (it != spectrumIndex.end()) { *index = it->second; return true; }
        slrhip_spectrum rec;
        std::memset(&rec, 0, sizeof(rec));
        std::vector<float> payload;
#ifdef Use_Spectral_Representation
        if (const UpsampledContinuousSpectrum* u = dynamic_cast<const UpsampledContinuousSpectrum*>(sp)) {
            rec.kind = SLRHIP_SPECTRUM_UPSAMPLED;
            rec.u = u->u; rec.v = u->v; rec.scale = u->scale;
            rec.num_samples = SLRHIP_UPSAMPLING_SAMPLES;
            payload.resize(4 + 4 * SLRHIP_UPSAMPLING_SAMPLES);
            uint32_t numPoints = 0;
            if (!resolve || resolve(&out->tables, u->u, u->v, &numPoints, payload.data()) != SLRHIP_OK)
                return fail("cannot resolve an upsampled spectrum (slrhip_resolve_upsampled)");
            rec.reserved = numPoints;
        }
        else if (const RegularContinuousSpectrum* r = dynamic_cast<const RegularContinuousSpectrum*>(sp)) {
            rec.kind = SLRHIP_SPECTRUM_REGULAR;
            rec.lambda_min = r->minLambda; rec.lambda_max = r->maxLambda; rec.num_samples = r->numSamples;
            payload.assign(r->values, r->values + r->numSamples);
        }
        else if (const IrregularContinuousSpectrum* ir = dynamic_cast<const IrregularContinuousSpectrum*>(sp)) {
            rec.kind = SLRHIP_SPECTRUM_IRREGULAR;
            rec.num_samples = ir->numSamples;
            payload.assign(ir->lambdas, ir->lambdas + ir->numSamples);
            payload.insert(payload.end(), ir->values, ir->values + ir->numSamples);
        }
        else return fail("unknown ContinuousSpectrum subclass");
#else
        rec.kind = SLRHIP_SPECTRUM_RGB_ONLY;
        rec.rgb[0] = sp->r; rec.rgb[1] = sp->g; rec.rgb[2] = sp->b;
#endif
        while (out->spectrumData.size() % 4) out->spectrumData.push_back(0.0f);
        rec.data_offset = (uint32_t)out->spectrumData.size();
        out->spectrumData.insert(out->spectrumData.end(), payload.begin(), payload.end());
        out->spectra.push_back(rec);
        *index = (int32_t)out->spectra.size() - 1;
        spectrumIndex[sp] = *index;
        return true;
    }

    bool normalMapOf(const Normal3DTexture* tex, uint32_t* index) {
        const CheckerBoardNormal3DTexture* cb = dynamic_cast<const CheckerBoardNormal3DTexture*>(tex);
        if (!cb) return fail("only the checkerboard normal map is supported");
        auto it = textureIndex.find(cb);
        if (it != textureIndex.end()) { *index = it->second; return true; }
        slrhip_texture rec;
        std::memset(&rec, 0, sizeof(rec));
        rec.kind = SLRHIP_TEXTURE_CHECKER_NORMAL;
        if (!mappingOf(cb->m_mapping, &rec)) return false;
        rec.value[0] = cb->m_stepWidth; rec.value[1] = cb->m_reverse ? 1.0f : 0.0f;
        *index = pushTexture(cb, rec);
        return true;
    }
    bool alphaMapOf(const FloatTexture* tex, uint32_t* index) {
        const CheckerBoardFloatTexture* cb = dynamic_cast<const CheckerBoardFloatTexture*>(tex);
        if (!cb) return fail("only the checkerboard float texture is supported as an alpha texture");
        auto it = textureIndex.find(cb);
        if (it != textureIndex.end()) { *index = it->second; return true; }
        slrhip_texture rec;
        std::memset(&rec, 0, sizeof(rec));
        rec.kind = SLRHIP_TEXTURE_CHECKER_FLOAT;
        if (!mappingOf(cb->m_mapping, &rec)) return false;
        rec.value[0] = cb->m_values[0]; rec.value[1] = cb->m_values[1];
        *index = pushTexture(cb, rec);
        return true;
    }
    // the material record of one object: its surface material + the object's normal map + its triangle's alpha texture
    bool groupOf(const SurfaceMaterial* mat, const Normal3DTexture* nmap, const FloatTexture* alpha, uint32_t* index) {
        if (!nmap && !alpha) return materialOf(mat, index);
        auto key = std::make_tuple(mat, nmap, alpha);
        auto it = groupIndex.find(key);
        if (it != groupIndex.end()) { *index = it->second; return true; }
        uint32_t base;
        if (!materialOf(mat, &base)) return false;
        slrhip_material m = out->materials[base];
        if (nmap) { uint32_t t; if (!normalMapOf(nmap, &t)) return false; m.reserved |= SLRHIP_MATERIAL_NORMAL_MAP(t); }
        if (alpha) { uint32_t t; if (!alphaMapOf(alpha, &t)) return false; m.reserved |= SLRHIP_MATERIAL_ALPHA_MAP(t); }
        out->materials.push_back(m);
        *index = (uint32_t)out->materials.size() - 1;
        groupIndex[key] = *index;
        return true;
    }

    bool floatOf(const FloatTexture* tex, float* value) {
        const ConstantFloatTexture* c = dynamic_cast<const ConstantFloatTexture*>(tex);
        if (!c) return fail("only constant float textures are on the hot path");
        *value = c->m_value;
        return true;
    }

    bool alphaOfGGX(const SVMicrofacetDistribution* D, float* alpha) {
        const SVGGX* g = dynamic_cast<const SVGGX*>(D);
        if (!g) return fail("only the GGX microfacet distribution is on the hot path");
        return floatOf(g->m_alpha_g, alpha);
    }

    // SurfaceMaterial -> index into out->materials (material -> BSDF factories of SURVEY row a16)
    // multiLevels: how many levels of summed / mixed materials may still open below this one (the C ABI takes one level of nesting)
    bool materialOf(const SurfaceMaterial* mat, uint32_t* index, int multiLevels = 2) {
        auto it = materialIndex.find(mat);
        if (it != materialIndex.end()) { *index = it->second; return true; }
        slrhip_material m;
        std::memset(&m, 0, sizeof(m));
        m.spectrum[0] = m.spectrum[1] = m.spectrum[2] = -1;
        m.param = -1.0f;
        m.emittance = -1;
        const SurfaceMaterial* base = mat;
        if (const EmitterSurfaceMaterial* e = dynamic_cast<const EmitterSurfaceMaterial*>(mat)) {
            const DiffuseEmission* de = dynamic_cast<const DiffuseEmission*>(e->m_emit);
            if (!de) return fail("only DiffuseEmission emitters are on the hot path");
            if (!spectrumOf(de->m_emittance, &m.emittance)) return false;
            base = e->m_mat;
            if (!base) return fail("an emitter without a surface material");
        }
        if (const DiffuseReflection* d = dynamic_cast<const DiffuseReflection*>(base)) {
            m.type = SLRHIP_MATERIAL_MATTE;
            if (!spectrumOf(d->m_reflectance, &m.spectrum[0])) return false;
            if (d->m_sigma && !floatOf(d->m_sigma, &m.param)) return false;      // no sigma texture -> Lambert (param < 0)
        }
        else if (const SpecularReflection* s = dynamic_cast<const SpecularReflection*>(base)) {
            m.type = SLRHIP_MATERIAL_METAL;
            if (!spectrumOf(s->m_coeffR, &m.spectrum[0]) || !spectrumOf(s->m_eta, &m.spectrum[1]) || !spectrumOf(s->m_k, &m.spectrum[2])) return false;
        }
        else if (const SpecularScattering* s = dynamic_cast<const SpecularScattering*>(base)) {
            m.type = SLRHIP_MATERIAL_GLASS;
            if (!spectrumOf(s->m_coeff, &m.spectrum[0]) || !spectrumOf(s->m_etaExt, &m.spectrum[1]) || !spectrumOf(s->m_etaInt, &m.spectrum[2])) return false;
        }
        else if (const MicrofacetReflection* r = dynamic_cast<const MicrofacetReflection*>(base)) {
            m.type = SLRHIP_MATERIAL_MICROFACET_METAL;
            if (!spectrumOf(r->m_eta, &m.spectrum[1]) || !spectrumOf(r->m_k, &m.spectrum[2]) || !alphaOfGGX(r->m_D, &m.param)) return false;
        }
        else if (const MicrofacetScattering* r = dynamic_cast<const MicrofacetScattering*>(base)) {
            m.type = SLRHIP_MATERIAL_MICROFACET_GLASS;
            if (!spectrumOf(r->m_etaExt, &m.spectrum[1]) || !spectrumOf(r->m_etaInt, &m.spectrum[2]) || !alphaOfGGX(r->m_D, &m.param)) return false;
        }
        else if (const ModifiedWardDurReflection* w = dynamic_cast<const ModifiedWardDurReflection*>(base)) {
            m.type = SLRHIP_MATERIAL_WARD;
            if (!spectrumOf(w->m_reflectance, &m.spectrum[0]) || !floatOf(w->m_anisoX, &m.param) || !floatOf(w->m_anisoY, &m.param2)) return false;
        }
        else if (const AshikhminShirleyReflection* a = dynamic_cast<const AshikhminShirleyReflection*>(base)) {
            m.type = SLRHIP_MATERIAL_ASHIKHMIN;
            if (!spectrumOf(a->m_Rs, &m.spectrum[0]) || !spectrumOf(a->m_Rd, &m.spectrum[1]) || !floatOf(a->m_nu, &m.param) || !floatOf(a->m_nv, &m.param2)) return false;
        }
        else {
            // SummedSurfaceMaterial / MixedSurfaceMaterial over two materials: single lobes (either possibly an InverseSurfaceMaterial) or,
            // one level down, summed / mixed materials of single lobes
            const SurfaceMaterial* c[2] = {nullptr, nullptr};
            float scale0 = 1.0f, scale1 = 1.0f;
            if (const SummedSurfaceMaterial* s = dynamic_cast<const SummedSurfaceMaterial*>(base)) { c[0] = s->m_mat0; c[1] = s->m_mat1; }
            else if (const MixedSurfaceMaterial* x = dynamic_cast<const MixedSurfaceMaterial*>(base)) {
                c[0] = x->m_mat0; c[1] = x->m_mat1;
                float f;
                if (!floatOf(x->m_factor, &f)) return false;
                scale0 = 1.0f - f; scale1 = f;                       // MixedSurfaceMaterial.cpp:16-17 with scale = 1
            }
            else return fail("a surface material outside the hot path");
            if (multiLevels <= 0) return fail("sum / mix materials nested more than one level deep are not supported");
            m.type = SLRHIP_MATERIAL_MULTI;
            int32_t bits = 0;
            for (int k = 0; k < 2; ++k) {
                const SurfaceMaterial* child = c[k];
                bool inverted = false;
                if (const InverseSurfaceMaterial* inv = dynamic_cast<const InverseSurfaceMaterial*>(child)) { child = inv->m_baseMat; bits |= 1 << k; inverted = true; }
                uint32_t ci;
                if (!materialOf(child, &ci, inverted ? 0 : multiLevels - 1)) return false;      // InverseBSDF only over a single lobe
                m.spectrum[k] = (int32_t)ci;
            }
            m.spectrum[2] = bits;
            m.param = scale0; m.param2 = scale1;
        }
        out->materials.push_back(m);
        *index = (uint32_t)out->materials.size() - 1;
        materialIndex[mat] = *index;
        return true;
    }
};

} // namespace

bool flattenScene(const Scene &scene, FlatScene* out, std::string* error, slrhip_resolve_upsampled_fn resolve) {
    Flattener f = {out, error, resolve, {}, {}, {}, {}, {}};
    *out = FlatScene();
    std::memset(&out->tables, 0, sizeof(out->tables));
#ifdef Use_Spectral_Representation
    {
        // libSLR's own upsampling tables, BasicTypes/Spectrum.h:197-575
        using namespace Upsampling;
        const uint32_t numCells = GridWidth * GridHeight;
        const uint32_t numPoints = sizeof(spectrum_data_points) / sizeof(spectrum_data_points[0]);
        out->gridCells.resize((size_t)numCells * 8);
        for (uint32_t c = 0; c < numCells; ++c) {
            uint8_t* dst = &out->gridCells[(size_t)c * 8];
            dst[0] = spectrum_grid[c].inside; dst[1] = spectrum_grid[c].num_points;
            for (int k = 0; k < 6; ++k) dst[2 + k] = spectrum_grid[c].idx[k];
        }
        for (uint32_t p = 0; p < numPoints; ++p) {
            out->pointUV.push_back(spectrum_data_points[p].uv[0]); out->pointUV.push_back(spectrum_data_points[p].uv[1]);
            out->pointSpectrum.insert(out->pointSpectrum.end(), spectrum_data_points[p].spectrum, spectrum_data_points[p].spectrum + NumWavelengthSamples);
        }
        out->tables.grid_width = GridWidth; out->tables.grid_height = GridHeight; out->tables.num_points = numPoints;
        out->tables.cells = out->gridCells.data(); out->tables.point_uv = out->pointUV.data(); out->tables.point_spectrum = out->pointSpectrum.data();
    }
#endif
    if (scene.m_envSphere) return f.fail("environment sphere: its image texture (Core/Image.h, OpenEXR half) is outside this build");
    const SurfaceObjectAggregate* agg = scene.m_aggregate;
    if (!agg) return f.fail("scene has no aggregate");
    const auto objectList = [](const SurfaceObjectAggregate* a) -> const std::vector<const SurfaceObject*>* {
        if (const SBVH* b = dynamic_cast<const SBVH*>(a->m_accelerator)) return &b->m_objLists;
        if (const QBVH* b = dynamic_cast<const QBVH*>(a->m_accelerator)) return &b->m_objLists;
        if (const StandardBVH* b = dynamic_cast<const StandardBVH*>(a->m_accelerator)) return &b->m_objLists;
        return nullptr;
    };
    const auto bySurface = [](const SingleSurfaceObject* a, const SingleSurfaceObject* b) { return a->m_surface < b->m_surface; };
    const std::vector<const SurfaceObject*>* list = objectList(agg);
    if (!list) return f.fail("unknown accelerator");

    // the distinct objects (spatial splits reference an object from several leaves), ordered by the address of their Surface.
    // A TransformedSurfaceObject over an aggregate of triangles with a StaticTransform (an instanced TriangleMeshNode,
    // Core/SurfaceObject.cpp:303-392) becomes a slrhip_instance: the mesh's triangles follow the loose ones as one range per
    // distinct aggregate, in their local space.
    std::vector<const SingleSurfaceObject*> objs;
    std::vector<const TransformedSurfaceObject*> placements;
    for (const SurfaceObject* o : *list) {
        if (const TransformedSurfaceObject* t = dynamic_cast<const TransformedSurfaceObject*>(o)) { placements.push_back(t); continue; }
        const SingleSurfaceObject* s = dynamic_cast<const SingleSurfaceObject*>(o);
        if (!s || dynamic_cast<const InfiniteSphereSurfaceObject*>(o))
            return f.fail("only SingleSurfaceObjects over Triangles and TransformedSurfaceObjects over aggregates of them are on the hot path");
        objs.push_back(s);
    }
    std::sort(objs.begin(), objs.end(), bySurface);
    objs.erase(std::unique(objs.begin(), objs.end()), objs.end());
    const size_t numLoose = objs.size();
    std::sort(placements.begin(), placements.end());
    placements.erase(std::unique(placements.begin(), placements.end()), placements.end());
    // distinct meshes, ordered like the loose triangles: by the address of their first Triangle
    std::map<const SurfaceObject*, std::vector<const SingleSurfaceObject*>> meshObjs;
    for (const TransformedSurfaceObject* t : placements) {
        const SurfaceObjectAggregate* mesh = dynamic_cast<const SurfaceObjectAggregate*>(t->m_surfObj);
        const StaticTransform* tf = dynamic_cast<const StaticTransform*>(t->m_transform);
        if (!mesh || !tf) return f.fail("an instance must be a StaticTransform over an aggregate of triangles (no animated or chained transforms)");
        if (mesh->isEmitting()) return f.fail("instanced triangles must not emit");
        if (meshObjs.count(mesh)) continue;
        const std::vector<const SurfaceObject*>* inner = objectList(mesh);
        if (!inner) return f.fail("unknown accelerator inside an instance");
        std::vector<const SingleSurfaceObject*> &mobjs = meshObjs[mesh];
        for (const SurfaceObject* o : *inner) {
            const SingleSurfaceObject* s = dynamic_cast<const SingleSurfaceObject*>(o);
            if (!s || dynamic_cast<const InfiniteSphereSurfaceObject*>(o)) return f.fail("an instanced aggregate may hold only SingleSurfaceObjects over Triangles (one level of instancing)");
            mobjs.push_back(s);
        }
        std::sort(mobjs.begin(), mobjs.end(), bySurface);
        mobjs.erase(std::unique(mobjs.begin(), mobjs.end()), mobjs.end());
        if (mobjs.empty()) return f.fail("empty instanced aggregate");
    }
    std::vector<const SurfaceObject*> meshOrder;
    for (const auto &kv : meshObjs) meshOrder.push_back(kv.first);
    std::sort(meshOrder.begin(), meshOrder.end(), [&](const SurfaceObject* a, const SurfaceObject* b) { return meshObjs[a].front()->m_surface < meshObjs[b].front()->m_surface; });
    std::map<const SurfaceObject*, std::pair<uint32_t, uint32_t>> meshRange;
    for (const SurfaceObject* mesh : meshOrder) {
        meshRange[mesh] = std::make_pair((uint32_t)objs.size(), (uint32_t)meshObjs[mesh].size());
        objs.insert(objs.end(), meshObjs[mesh].begin(), meshObjs[mesh].end());
    }
    out->instances.clear();
    for (const TransformedSurfaceObject* t : placements) {
        const StaticTransform* tf = static_cast<const StaticTransform*>(t->m_transform);
        slrhip_instance in;
        in.first_triangle = meshRange[t->m_surfObj].first; in.num_triangles = meshRange[t->m_surfObj].second;
        std::memcpy(in.local_to_world, &tf->mat, sizeof(float) * 16);          // Matrix4x4: four column vectors, m[c * 4 + r]
        std::memcpy(in.world_to_local, &tf->matInv, sizeof(float) * 16);
        out->instances.push_back(in);
    }

    // vertices: the distinct Vertex objects in address order (a mesh's vertices are one array: TriangleMeshNode.cpp:68-78)
    std::vector<const Vertex*> verts;
    for (const SingleSurfaceObject* o : objs) {
        const Triangle* t = dynamic_cast<const Triangle*>(o->m_surface);
        if (!t) return f.fail("only Triangle surfaces are on the hot path");
        for (int k = 0; k < 3; ++k) verts.push_back(t->m_v[k]);
    }
    std::sort(verts.begin(), verts.end());
    verts.erase(std::unique(verts.begin(), verts.end()), verts.end());
    out->vertices.resize(verts.size());
    for (size_t i = 0; i < verts.size(); ++i) {
        const Vertex &v = *verts[i];
        slrhip_vertex &d = out->vertices[i];
        d.position[0] = v.position.x; d.position[1] = v.position.y; d.position[2] = v.position.z;
        d.normal[0] = v.normal.x; d.normal[1] = v.normal.y; d.normal[2] = v.normal.z;
        d.tangent[0] = v.tangent.x; d.tangent[1] = v.tangent.y; d.tangent[2] = v.tangent.z;
        d.texcoord[0] = v.texCoord.u; d.texcoord[1] = v.texCoord.v;
        f.vertexIndex[verts[i]] = (uint32_t)i;
    }
    out->triangles.resize(objs.size());
    for (size_t i = 0; i < objs.size(); ++i) {
        const Triangle* t = static_cast<const Triangle*>(objs[i]->m_surface);
        slrhip_triangle &d = out->triangles[i];
        for (int k = 0; k < 3; ++k) d.v[k] = f.vertexIndex[t->m_v[k]];
        const BumpSingleSurfaceObject* bump = dynamic_cast<const BumpSingleSurfaceObject*>(objs[i]);
        if (!f.groupOf(objs[i]->m_material, bump ? bump->m_normalMap : nullptr, t->m_alphaTex, &d.material)) return false;
    }
    // the light list must come out in the reference's order (Scene::selectLight indexes it): emitting triangles appear in
    // out->triangles in the order of m_lightList (SurfaceObject.cpp:232-249)
    {
        size_t next = 0;
        for (size_t i = 0; i < numLoose; ++i) {
            if (!objs[i]->isEmitting()) continue;
            if (agg->m_lightList[next] != objs[i]) return f.fail("light list order differs from the order of the Triangle objects in memory");
            ++next;
        }
    }

    const PerspectiveCamera* cam = dynamic_cast<const PerspectiveCamera*>(scene.getCamera());
    if (!cam) return f.fail("only the PerspectiveCamera is on the hot path");
    const StaticTransform* tf = dynamic_cast<const StaticTransform*>(cam->m_transform);
    if (!tf) return f.fail("only a static camera transform is on the hot path");
    std::memset(&out->camera, 0, sizeof(out->camera));
    std::memcpy(out->camera.local_to_world, &tf->mat, sizeof(float) * 16);       // Matrix4x4: four column vectors, m[c * 4 + r]
    std::memcpy(out->camera.world_to_local, &tf->matInv, sizeof(float) * 16);
    out->camera.aspect = cam->m_aspect; out->camera.fov_y = cam->m_fovY; out->camera.lens_radius = cam->m_lensRadius;
    out->camera.img_plane_distance = cam->m_imgPlaneDistance; out->camera.obj_plane_distance = cam->m_objPlaneDistance;
    out->camera.sensitivity = cam->getSensor()->m_sensitivity;                   // already resolved by PerspectiveCamera.cpp:23
    return true;
}

namespace {

struct HipApi {
    void* handle = nullptr;
    decltype(&slrhip_create) create = nullptr;
    decltype(&slrhip_destroy) destroy = nullptr;
    decltype(&slrhip_upload_scene) upload_scene = nullptr;
    decltype(&slrhip_render_begin) render_begin = nullptr;
    decltype(&slrhip_render) render = nullptr;
    decltype(&slrhip_read_framebuffer) read_framebuffer = nullptr;
    decltype(&slrhip_last_error_string) last_error_string = nullptr;
    decltype(&slrhip_resolve_upsampled) resolve_upsampled = nullptr;
    bool load(const std::string &path, std::string* error) {
        std::string p = path;
        if (p.empty()) { const char* e = std::getenv("SLRHIP_LIBRARY"); p = e ? e : "libslrhip.so"; }
        handle = dlopen(p.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (!handle) { *error = std::string("cannot load ") + p + ": " + dlerror(); return false; }
#define SLRHIP_SYM(name) name = reinterpret_cast<decltype(name)>(dlsym(handle, "slrhip_" #name)); if (!name) { *error = "libslrhip.so lacks slrhip_" #name; return false; }
        SLRHIP_SYM(create) SLRHIP_SYM(destroy) SLRHIP_SYM(upload_scene) SLRHIP_SYM(render_begin) SLRHIP_SYM(render)
        SLRHIP_SYM(read_framebuffer) SLRHIP_SYM(last_error_string) SLRHIP_SYM(resolve_upsampled)
#undef SLRHIP_SYM
        return true;
    }
};

[[noreturn]] void die(const std::string &what) {
    std::fprintf(stderr, "%s\n", what.c_str());
    std::exit(-1);      // HostProgram/main.cpp:39-42
}

} // namespace

bool flattenSceneWithLibrary(const Scene &scene, FlatScene* out, std::string* error, const std::string &libraryPath) {
    HipApi api;
    if (!api.load(libraryPath, error)) return false;
    return flattenScene(scene, out, error, api.resolve_upsampled);
}

void HIPPathTracingRenderer::render(const Scene &scene, const RenderSettings &settings) const {
    HipApi api;
    std::string error;
    if (!api.load(m_libraryPath, &error)) die("HIPPathTracingRenderer: " + error);
    FlatScene flat;
    if (!flattenScene(scene, &flat, &error, api.resolve_upsampled)) die(error);
    auto check = [&](int rc, const char* what) { if (rc != SLRHIP_OK) die(std::string(what) + " failed: " + api.last_error_string()); };

#ifdef Use_Spectral_Representation
    const int mode = SLRHIP_MODE_SPECTRAL, components = 16;
#else
    const int mode = SLRHIP_MODE_RGB, components = 3;
#endif
    slrhip_config cfg = {m_device, mode, 0, 0};
    slrhip_ctx* ctx = nullptr;
    check(api.create(&cfg, &ctx), "slrhip_create");
    slrhip_scene_desc desc = flat.desc();
    check(api.upload_scene(ctx, &desc), "slrhip_upload_scene");
    slrhip_render_settings st = {settings.getInt(RenderSettingItem::ImageWidth), settings.getInt(RenderSettingItem::ImageHeight),
                                 settings.getFloat(RenderSettingItem::TimeStart), settings.getFloat(RenderSettingItem::TimeEnd),
                                 settings.getFloat(RenderSettingItem::Brightness), settings.getInt(RenderSettingItem::RNGSeed)};
    slrhip_shard whole = {0, 1};
    check(api.render_begin(ctx, &st, whole), "slrhip_render_begin");

    ImageSensor* sensor = scene.getCamera()->getSensor();
    sensor->init((uint32_t)st.image_width, (uint32_t)st.image_height);                     // PathTracingRenderer.cpp:67
    std::vector<float> fb((size_t)st.image_width * st.image_height * components);

    // the pass loop of PathTracingRenderer.cpp:63-94: images after 1, 2, 4, ... passes, at most 16 of them
    uint32_t exportPass = 1, imgIdx = 0, done = 0;
    const uint32_t endIdx = 16;
    auto start = std::chrono::system_clock::now();
    while (done < m_samplesPerPixel) {
        const uint32_t upTo = std::min(exportPass, m_samplesPerPixel);
        check(api.render(ctx, done, upTo - done, nullptr), "slrhip_render");
        done = upTo;
        // the sensor receives exactly what it would hold after these passes: the un-normalised sums (ImageSensor.cpp:124-129)
        check(api.read_framebuffer(ctx, fb.data(), fb.size()), "slrhip_read_framebuffer");
        for (int32_t y = 0; y < st.image_height; ++y)
            for (int32_t x = 0; x < st.image_width; ++x) {
                const float* p = &fb[((size_t)y * st.image_width + x) * components];
#ifdef Use_Spectral_Representation
                sensor->pixel((uint32_t)x, (uint32_t)y).value = CompensatedSum<DiscretizedSpectrum>(DiscretizedSpectrum(p));
#else
                sensor->pixel((uint32_t)x, (uint32_t)y).value = CompensatedSum<DiscretizedSpectrum>(DiscretizedSpectrum(p[0], p[1], p[2]));
#endif
            }
        if (done == exportPass) {
            char filename[256];
            std::snprintf(filename, sizeof(filename), "%03u.bmp", imgIdx);
            double elapsed = (double)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::system_clock::now() - start).count();
            sensor->saveImage(filename, settings.getFloat(RenderSettingItem::Brightness) / done);   // the reference's own tone map + BMP
            std::printf("%u samples: %s, %g[s]\n", exportPass, filename, elapsed * 0.001f);
            ++imgIdx;
            if (imgIdx == endIdx) break;
            exportPass += exportPass;
        }
    }
    api.destroy(ctx);
}

} // namespace SLR
