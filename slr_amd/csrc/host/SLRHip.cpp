// SLRHip.cpp — see SLRHip.h.  Host C++ only; everything on the GPU goes through the C ABI.
#include "SLRHip.h"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace SLRHip {

Scene::Scene() { std::memset(&m_camera, 0, sizeof(m_camera)); std::memset(&m_tables, 0, sizeof(m_tables)); }

bool Scene::loadSpectralTables(const std::string& path) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) { std::fprintf(stderr, "SLRHip: cannot open %s\n", path.c_str()); return false; }
    char magic[8];
    uint32_t hdr[4];
    bool ok = std::fread(magic, 1, 8, f) == 8 && std::memcmp(magic, "SLRUPS01", 8) == 0 && std::fread(hdr, 4, 4, f) == 4;
    if (ok) ok = hdr[0] > 0 && hdr[1] > 0 && hdr[0] * hdr[1] <= 65536 && hdr[2] > 0 && hdr[2] <= 255 && hdr[3] >= 2 && hdr[3] <= 65536;
    if (ok) {
        m_gridCells.resize((size_t)hdr[0] * hdr[1] * 8);
        m_pointUV.resize((size_t)hdr[2] * 2);
        m_pointSpectrum.resize((size_t)hdr[2] * SLRHIP_UPSAMPLING_SAMPLES);
        m_d65.resize(hdr[3]);
        ok = std::fread(m_gridCells.data(), 1, m_gridCells.size(), f) == m_gridCells.size() &&
             std::fread(m_pointUV.data(), 4, m_pointUV.size(), f) == m_pointUV.size() &&
             std::fread(m_pointSpectrum.data(), 4, m_pointSpectrum.size(), f) == m_pointSpectrum.size() &&
             std::fread(m_d65.data(), 4, m_d65.size(), f) == m_d65.size();
    }
    std::fclose(f);
    if (!ok) { std::fprintf(stderr, "SLRHip: %s is not a spectral table file\n", path.c_str()); return false; }
    m_tables.grid_width = hdr[0]; m_tables.grid_height = hdr[1]; m_tables.num_points = hdr[2];
    m_tables.cells = m_gridCells.data(); m_tables.point_uv = m_pointUV.data(); m_tables.point_spectrum = m_pointSpectrum.data();
    m_hasTables = true;
    return true;
}

static float degammaSRGB(float v) { return v <= 0.04045 ? (float)(v / 12.92) : (float)std::pow((v + 0.055) / 1.055, 2.4); }     // Spectrum.cpp:24-30

uint32_t Scene::addUpsampledSpectrum(int32_t spectrumType, int32_t colorSpace, float e0, float e1, float e2) {
    if (!m_hasTables) { std::fprintf(stderr, "SLRHip: addUpsampledSpectrum needs loadSpectralTables()\n"); std::exit(-1); }
    slrhip_spectrum s;
    std::memset(&s, 0, sizeof(s));
    s.kind = SLRHIP_SPECTRUM_UPSAMPLED;
    // RGB build: Spectrum::create keeps linear sRGB (API.cpp:1281-1369); a non-linear triple is de-gamma'd first
    const bool nonLinear = colorSpace == SLRHIP_COLORSPACE_SRGB_NONLINEAR;
    s.rgb[0] = nonLinear ? degammaSRGB(e0) : e0; s.rgb[1] = nonLinear ? degammaSRGB(e1) : e1; s.rgb[2] = nonLinear ? degammaSRGB(e2) : e2;
    float uvs[3];
    std::vector<float> payload(4 + 4 * SLRHIP_UPSAMPLING_SAMPLES);
    uint32_t numPoints = 0;
    if (slrhip_upsample(spectrumType, colorSpace, e0, e1, e2, uvs) != SLRHIP_OK ||
        slrhip_resolve_upsampled(&m_tables, uvs[0], uvs[1], &numPoints, payload.data()) != SLRHIP_OK) {
        std::fprintf(stderr, "SLRHip: cannot upsample (%g, %g, %g)\n", e0, e1, e2);
        std::exit(-1);
    }
    s.u = uvs[0]; s.v = uvs[1]; s.scale = uvs[2];
    s.num_samples = SLRHIP_UPSAMPLING_SAMPLES;
    s.reserved = numPoints;
    return addSpectrum(s, payload.data(), (uint32_t)payload.size());
}
uint32_t Scene::addRegularSpectrum(float lambdaMin, float lambdaMax, const float* values, uint32_t numSamples, const float rgb[3]) {
    slrhip_spectrum s;
    std::memset(&s, 0, sizeof(s));
    s.kind = SLRHIP_SPECTRUM_REGULAR;
    s.lambda_min = lambdaMin; s.lambda_max = lambdaMax; s.num_samples = numSamples;
    for (int i = 0; i < 3; ++i) s.rgb[i] = rgb[i];
    return addSpectrum(s, values, numSamples);
}
uint32_t Scene::addIrregularSpectrum(const float* lambdas, const float* values, uint32_t numSamples, const float rgb[3]) {
    slrhip_spectrum s;
    std::memset(&s, 0, sizeof(s));
    s.kind = SLRHIP_SPECTRUM_IRREGULAR;
    s.num_samples = numSamples;
    for (int i = 0; i < 3; ++i) s.rgb[i] = rgb[i];
    std::vector<float> payload(lambdas, lambdas + numSamples);
    payload.insert(payload.end(), values, values + numSamples);
    return addSpectrum(s, payload.data(), (uint32_t)payload.size());
}
uint32_t Scene::addD65Spectrum(float scale) {
    if (!m_hasTables) { std::fprintf(stderr, "SLRHip: addD65Spectrum needs loadSpectralTables()\n"); std::exit(-1); }
    // RGB build: Spectrum::create(Illuminant, 300, 830, D65, 531) (API.cpp:405-406,1326-1347), then `* scale` on the RGB triple
    float rgb[3];
    if (slrhip_spectrum_to_rgb(SLRHIP_SPECTRUMTYPE_ILLUMINANT, nullptr, 300.0f, 830.0f, m_d65.data(), (uint32_t)m_d65.size(), rgb) != SLRHIP_OK) std::exit(-1);
    std::vector<float> v(m_d65);
    for (float& x : v) x = scale * x;                  // spectral build: RegularContinuousSpectrum::createScaled, SpectrumTypes.h:112-118
    const float scaled[3] = {scale * rgb[0], scale * rgb[1], scale * rgb[2]};
    return addRegularSpectrum(300.0f, 830.0f, v.data(), (uint32_t)v.size(), scaled);
}

uint32_t Scene::addVertex(const float position[3], const float normal[3], const float tangent[3], const float texcoord[2]) {
    slrhip_vertex v;
    for (int i = 0; i < 3; ++i) { v.position[i] = position[i]; v.normal[i] = normal[i]; v.tangent[i] = tangent[i]; }
    v.texcoord[0] = texcoord[0]; v.texcoord[1] = texcoord[1];
    m_vertices.push_back(v);
    return (uint32_t)m_vertices.size() - 1;
}
uint32_t Scene::addTriangle(uint32_t v0, uint32_t v1, uint32_t v2, uint32_t material) {
    slrhip_triangle t = {{v0, v1, v2}, material};
    m_triangles.push_back(t);
    return (uint32_t)m_triangles.size() - 1;
}
uint32_t Scene::addSpectrumRGB(float r, float g, float b) {
    slrhip_spectrum s;
    std::memset(&s, 0, sizeof(s));
    s.kind = SLRHIP_SPECTRUM_RGB_ONLY;
    s.rgb[0] = r; s.rgb[1] = g; s.rgb[2] = b;
    m_spectra.push_back(s);
    return (uint32_t)m_spectra.size() - 1;
}
uint32_t Scene::addSpectrum(slrhip_spectrum descriptor, const float* payload, uint32_t numPayloadFloats) {
    while (m_spectrumData.size() % 4) m_spectrumData.push_back(0.0f);
    descriptor.data_offset = (uint32_t)m_spectrumData.size();
    if (payload) m_spectrumData.insert(m_spectrumData.end(), payload, payload + numPayloadFloats);
    m_spectra.push_back(descriptor);
    return (uint32_t)m_spectra.size() - 1;
}
void Scene::setEnvironment(const float* texels, uint32_t width, uint32_t height, float scale, const float* importance, uint32_t mapWidth,
                           uint32_t mapHeight) {
    m_envTexels.assign(texels, texels + (size_t)width * height * 3);
    m_envImportance.assign(importance, importance + (size_t)mapWidth * mapHeight);
    std::memset(&m_env, 0, sizeof(m_env));
    m_env.width = width; m_env.height = height; m_env.scale = scale;
    m_env.map_width = mapWidth; m_env.map_height = mapHeight;
    m_hasEnv = true;
}
static slrhip_texture makeTexture(uint32_t kind, float ox, float oy, float sx, float sy) {
    slrhip_texture t;
    std::memset(&t, 0, sizeof(t));
    t.kind = kind; t.offset[0] = ox; t.offset[1] = oy; t.scale[0] = sx; t.scale[1] = sy;
    t.spectrum[0] = t.spectrum[1] = -1;
    return t;
}
int32_t Scene::addCheckerSpectrumTexture(uint32_t spectrum0, uint32_t spectrum1, float ox, float oy, float sx, float sy) {
    slrhip_texture t = makeTexture(SLRHIP_TEXTURE_CHECKER_SPECTRUM, ox, oy, sx, sy);
    t.spectrum[0] = (int32_t)spectrum0; t.spectrum[1] = (int32_t)spectrum1;
    m_textures.push_back(t);
    return SLRHIP_TEXTURE_REF(m_textures.size() - 1);
}
uint32_t Scene::addCheckerFloatTexture(float v0, float v1, float ox, float oy, float sx, float sy) {
    slrhip_texture t = makeTexture(SLRHIP_TEXTURE_CHECKER_FLOAT, ox, oy, sx, sy);
    t.value[0] = v0; t.value[1] = v1;
    m_textures.push_back(t);
    return (uint32_t)m_textures.size() - 1;
}
uint32_t Scene::addCheckerNormalTexture(float stepWidth, bool reverse, float ox, float oy, float sx, float sy) {
    slrhip_texture t = makeTexture(SLRHIP_TEXTURE_CHECKER_NORMAL, ox, oy, sx, sy);
    t.value[0] = stepWidth; t.value[1] = reverse ? 1.0f : 0.0f;
    m_textures.push_back(t);
    return (uint32_t)m_textures.size() - 1;
}
uint32_t Scene::addMaterial(uint32_t type, int32_t s0, int32_t s1, int32_t s2, float param, int32_t emittance, float param2, int32_t normalMap,
                            int32_t alphaMap) {
    const uint32_t maps = (normalMap >= 0 ? SLRHIP_MATERIAL_NORMAL_MAP(normalMap) : 0u) | (alphaMap >= 0 ? SLRHIP_MATERIAL_ALPHA_MAP(alphaMap) : 0u);
    slrhip_material m = {type, {s0, s1, s2}, param, emittance, param2, maps};
    m_materials.push_back(m);
    return (uint32_t)m_materials.size() - 1;
}
uint32_t Scene::addSummedMaterial(uint32_t mat0, uint32_t mat1, bool inverse0, bool inverse1, int32_t emittance) {
    const int32_t bits = (inverse0 ? SLRHIP_MULTI_INVERSE_0 : 0) | (inverse1 ? SLRHIP_MULTI_INVERSE_1 : 0);
    return addMaterial(SLRHIP_MATERIAL_MULTI, (int32_t)mat0, (int32_t)mat1, bits, 1.0f, emittance, 1.0f);
}
uint32_t Scene::addMixedMaterial(uint32_t mat0, uint32_t mat1, float factor, bool inverse0, bool inverse1, int32_t emittance) {
    const int32_t bits = (inverse0 ? SLRHIP_MULTI_INVERSE_0 : 0) | (inverse1 ? SLRHIP_MULTI_INVERSE_1 : 0);
    // MixedSurfaceMaterial::getBSDF hands scale * (1.0f - factor) and scale * factor to the components (MixedSurfaceMaterial.cpp:16-17)
    return addMaterial(SLRHIP_MATERIAL_MULTI, (int32_t)mat0, (int32_t)mat1, bits, 1.0f - factor, emittance, factor);
}
slrhip_scene_desc Scene::desc() const {
    slrhip_scene_desc d;
    std::memset(&d, 0, sizeof(d));
    d.vertices = m_vertices.data(); d.num_vertices = (uint32_t)m_vertices.size();
    d.triangles = m_triangles.data(); d.num_triangles = (uint32_t)m_triangles.size();
    d.materials = m_materials.data(); d.num_materials = (uint32_t)m_materials.size();
    d.spectra = m_spectra.data(); d.num_spectra = (uint32_t)m_spectra.size();
    d.spectrum_data = m_spectrumData.data(); d.num_spectrum_data = (uint32_t)m_spectrumData.size();
    d.camera = m_camera;
    d.env = nullptr;
    d.upsampling = m_hasTables ? &m_tables : nullptr;
    d.textures = m_textures.empty() ? nullptr : m_textures.data();
    d.num_textures = (uint32_t)m_textures.size();
    if (m_hasEnv) {
        // the descriptor points into this object: valid as long as the Scene is
        const_cast<Scene*>(this)->m_env.texels = m_envTexels.data();
        const_cast<Scene*>(this)->m_env.importance = m_envImportance.data();
        d.env = &m_env;
    }
    return d;
}

static void die(const char* what, int rc) {
    std::fprintf(stderr, "%s failed (%d): %s\n", what, rc, slrhip_last_error_string());
    std::exit(-1);      // HostProgram/main.cpp:39-42
}

void PathTracingRenderer::render(const Scene& scene, const RenderSettings& settings) const {
    slrhip_config cfg = {m_device, m_mode, 0, 0};
    slrhip_ctx* ctx = nullptr;
    int rc = slrhip_create(&cfg, &ctx);
    if (rc) die("slrhip_create", rc);
    slrhip_scene_desc desc = scene.desc();
    if ((rc = slrhip_upload_scene(ctx, &desc))) die("slrhip_upload_scene", rc);

    slrhip_render_settings st;
    st.image_width = settings.getInt(RenderSettingItem::ImageWidth);
    st.image_height = settings.getInt(RenderSettingItem::ImageHeight);
    st.time_start = settings.getFloat(RenderSettingItem::TimeStart);
    st.time_end = settings.getFloat(RenderSettingItem::TimeEnd);
    st.brightness = settings.getFloat(RenderSettingItem::Brightness);
    st.rng_seed = settings.getInt(RenderSettingItem::RNGSeed);
    slrhip_shard whole = {0, 1};
    if ((rc = slrhip_render_begin(ctx, &st, whole))) die("slrhip_render_begin", rc);

    // sensitivity of PerspectiveCamera.cpp:23 (ImageSensor::saveImage multiplies the scale by it, ImageSensor.cpp:143-147)
    const slrhip_camera& cam = scene.camera();
    float sensitivity = cam.sensitivity > 0 ? cam.sensitivity : (float)(1.0f / (M_PI * (double)cam.lens_radius * (double)cam.lens_radius));
    if (std::isinf(sensitivity)) sensitivity = 1.0f;

    const int components = slrhip_components(ctx);          // 3, or the 16 storage bins of the spectral build
    const size_t numFloats = (size_t)st.image_width * st.image_height * components;
    std::vector<float> fb(numFloats);
    const uint32_t byteWidth = 3u * (uint32_t)st.image_width + (uint32_t)st.image_width % 4u;
    std::vector<uint8_t> bmp((size_t)byteWidth * st.image_height);

    // the pass loop of PathTracingRenderer.cpp:63-94: images after 1, 2, 4, ... passes, at most 16 of them
    uint32_t exportPass = 1, imgIdx = 0;
    const uint32_t endIdx = 16;
    auto start = std::chrono::system_clock::now();
    uint32_t done = 0;
    while (done < m_samplesPerPixel) {
        const uint32_t upTo = exportPass <= m_samplesPerPixel ? exportPass : m_samplesPerPixel;
        if ((rc = slrhip_render(ctx, done, upTo - done, nullptr))) die("slrhip_render", rc);
        done = upTo;
        if (done == exportPass) {
            if ((rc = slrhip_read_framebuffer(ctx, fb.data(), numFloats))) die("slrhip_read_framebuffer", rc);
            char filename[256];
            std::snprintf(filename, sizeof(filename), "%03u.bmp", imgIdx);
            double elapsed = (double)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::system_clock::now() - start).count();
            const float scale = st.brightness / (float)done * sensitivity;
            if ((rc = slrhip_tonemap_bgr8(fb.data(), st.image_width, st.image_height, components, scale, bmp.data(), bmp.size()))) die("slrhip_tonemap_bgr8", rc);
            const std::string path = m_outputDir + "/" + filename;
            if ((rc = slrhip_save_bmp(path.c_str(), bmp.data(), st.image_width, st.image_height))) die("slrhip_save_bmp", rc);
            std::printf("%u samples: %s, %g[s]\n", exportPass, filename, elapsed * 0.001f);
            ++imgIdx;
            if (imgIdx == endIdx) break;
            exportPass += exportPass;
        }
    }
    slrhip_destroy(ctx);
}

} // namespace SLRHip
