// device_types.h — data layouts shared by the host library and the HIP kernels.
//
// Everything the kernels read is a flat, index-based array in HBM; no pointers cross the
// host/device boundary.  Layouts are chosen for the access pattern of the wavefront kernels:
//   * QNode   128 B = one L2 line: the 4 child boxes SoA (6 x float4) so one lane tests four
//             slabs from six 16-byte loads, + 4 child references.  Same information as the
//             reference's QBVH::Node (libSLR/Accelerator/QBVH.h:42-54); the child reference
//             packs (leaf flag, count, index) like QBVH::Children (:27-40).
//   * LeafTri  48 B: v0, e1 = v1-v0, e2 = v2-v0 (the two subtractions Triangle::intersect does
//             first, TriangleMesh.cpp:136-137, hoisted to build time: same float operations,
//             same bits) + the triangle's index in scene order.
//   * ShadeTri 96 B: everything Triangle::getSurfacePoint needs (TriangleMesh.cpp:180-215)
//             inline, so shading a hit is six 16-byte loads and no vertex indirection.
#pragma once
#include <stdint.h>
#include <stdlib.h>

namespace slrhip {

// Measurement knobs (SLRHIP_WS_NC, SLRHIP_QUANT, SLRHIP_PAIRS, ...: the values the defaults were chosen against) are read
// only by libraries built with -DSLR_TUNING_KNOBS (tools/build_variant.sh).  The shipped library reads five environment variables:
// SLRHIP_BVH (host | device | sbvh), SLRHIP_BVH_TIMING, SLRHIP_TAIL_SLOTS, SLRHIP_GRAPH, SLRHIP_ITER_LOG.
inline const char* tuningEnv(const char* name) {
#ifdef SLR_TUNING_KNOBS
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

static const uint32_t kInvalidChild = 0xFFFFFFFFu;
static const uint32_t kLeafFlag = 0x80000000u;
static const uint32_t kLeafCountShift = 27;
static const uint32_t kLeafIndexMask = (1u << 27) - 1;
static const uint32_t kMaxLeafTris = 4;
// Instanced scenes (slrhip_instance): a child reference with the leaf flag and a COUNT OF ZERO names an instance (index in the low
// bits) instead of a packet of triangles; the traversal marks the place on its stack where it has to leave the instance's local
// space again with kPopInstance (leaf flag, count 15: no packet has that count).
static const uint32_t kPopInstance = 0xFFFFFFFEu;

struct alignas(16) QNode {
    float minx[4], miny[4], minz[4];
    float maxx[4], maxy[4], maxz[4];
    uint32_t child[4];      // kInvalidChild | inner node index | kLeafFlag | count << 27 | first LeafTri
    uint32_t pad[4];
};
static_assert(sizeof(QNode) == 128, "QNode must be one 128-byte line");

// QNodeQ 64 B: the same node with the child boxes quantized to 8 bits per plane inside the box of all children
// (origin + q * scale, per axis), for scenes whose 128-byte nodes do not fit the caches: half the bytes per node visit
// and a 10 M-triangle tree (3.3 M nodes) drops from 421 MB to 210 MB, inside the 256 MB Infinity Cache.  The host
// rounds q outwards and checks, with the device's own fma, that every dequantized box contains the float box, so the
// traversal visits a superset of the nodes and the hits (decided by the exact triangle test) are unchanged.
struct alignas(16) QNodeQ {
    float ox, oy, oz, sx;
    float sy, sz; uint32_t qlox, qloy;         // one byte per child
    uint32_t qloz, qhix, qhiy, qhiz;
    uint32_t child[4];
};
static_assert(sizeof(QNodeQ) == 64, "QNodeQ must be half a 128-byte line");

// QNode8 128 B = one line: an EIGHT-wide node with the child boxes quantized like QNodeQ's (8 bits per plane inside the box of all
// children).  Half the levels of the four-wide tree: fewer dependent node visits per ray, the same seven 16-byte loads per visit as
// a float QNode.  Measured against the four-wide trees in DESIGN.md (k_trace_ws<.., WIDE8>); float4 index inside the node:
//   0: ox oy oz sx | 1: sy sz lox[0-3] lox[4-7] | 2: loy[0-3] loy[4-7] loz[0-3] loz[4-7] | 3: hix hix hiy hiy | 4: hiz hiz - - |
//   5: child[0-3] | 6: child[4-7] | 7: unused
struct alignas(16) QNode8 {
    float ox, oy, oz, sx;
    float sy, sz; uint32_t qlox[2];
    uint32_t qloy[2], qloz[2];
    uint32_t qhix[2], qhiy[2];
    uint32_t qhiz[2], pad0[2];
    uint32_t child[8];
    uint32_t pad1[4];
};
static_assert(sizeof(QNode8) == 128, "QNode8 must be one 128-byte line");

struct alignas(16) LeafTri {
    float v0[3]; uint32_t tri;
    float e1[3]; uint32_t alpha;         // index into the alpha records (Triangle::m_alphaTex, pt_tex.h), kNoAlpha = none
    float e2[3]; uint32_t pad1;
};
static const uint32_t kNoAlpha = 0xFFFFFFFFu;
static_assert(sizeof(LeafTri) == 48, "LeafTri is three float4");

struct alignas(16) ShadeTri {
    float n0[3]; uint32_t material;
    float n1[3]; int32_t light;          // index into the light table, -1 if not emitting
    float n2[3]; float areaPDF;          // 1 / Triangle::area() (TriangleMesh.cpp:217-222,257-260)
    float t0[3]; float gnx;
    float t1[3]; float gny;
    float t2[3]; float gnz;              // gNormal = normalize(cross(e01, e02)) (TriangleMesh.cpp:171)
};
static_assert(sizeof(ShadeTri) == 96, "ShadeTri is six float4");

// One slrhip_instance (TransformedSurfaceObject over a mesh's aggregate, Core/SurfaceObject.cpp:303-392), 144 B = nine float4:
// both matrices column-major as the ABI gives them (float4 k = column k; the kernels require the bottom row 0 0 0 1, checked at
// upload), then the root of the mesh's tree in the scene's node array.
struct alignas(16) DevInstance {
    float localToWorld[16];
    float worldToLocal[16];
    uint32_t rootNode, firstTriangle, numTriangles, mesh;
};
static_assert(sizeof(DevInstance) == 144, "DevInstance is nine float4");

// One emitting triangle (a "light" of SurfaceObjectAggregate's light list, SurfaceObject.cpp:232-249):
// what Triangle::sample needs (TriangleMesh.cpp:224-255).
struct alignas(16) LightTri {
    float p0[3]; uint32_t tri;
    float p1[3]; uint32_t material;
    float p2[3]; float areaPDF;
    float n0[3]; float gnx;
    float n1[3]; float gny;
    float n2[3]; float gnz;
    float t0[3]; float pad0;
    float t1[3]; float pad1;
    float t2[3]; float pad2;
};
static_assert(sizeof(LightTri) == 144, "LightTri is nine float4");

// Per-material constants, RGB mode (BSDF factories of basic_SurfaceMaterials.cpp:15-43 with
// constant textures evaluated once: RGBTemplate::evaluate returns itself, RGBTypes.h:124-126).
struct alignas(16) DevMaterial {
    uint32_t type;          // SLRHIP_MATERIAL_*
    float param;            // matte: sigma (< 0: Lambert) | microfacet: alpha_g
    float onA, onB;         // Oren-Nayar m_A, m_B (OrenNayerBRDF.h:28-30), evaluated on the host in double like the reference
    float a[4];             // matte: R | metal: coeffR | glass: coeff
    float b[4];             // metal: eta | glass: etaExt
    float c[4];             // metal: k   | glass: etaInt
    float emittance[4];
};
static_assert(sizeof(DevMaterial) == 80, "DevMaterial layout");

// SLRHIP_MATERIAL_MULTI records (both modes): param / onA = the two components' scales, onB = these bits
// (component indices into the same table, InverseBSDF flags, the components' material types).
static const uint32_t kMultiMaxChildIndex = 1023;
inline uint32_t packMultiBits(uint32_t child0, uint32_t child1, uint32_t inverseBits, uint32_t type0, uint32_t type1) {
    return child0 | (child1 << 10) | (inverseBits << 20) | (type0 << 22) | (type1 << 25);
}

// Spectral mode: a material names its constant spectra; they are evaluated at the path's wavelengths per hit
// (ConstantSpectrumTexture::evaluate, Textures/constant_textures.h:16-31), like the reference does.
struct alignas(16) DevMaterialS {
    uint32_t type;
    float param, onA, onB;
    int32_t spec[4];        // a, b, c, emittance: indices into the spectrum table, -1 = unused
};
static_assert(sizeof(DevMaterialS) == 32, "DevMaterialS layout");

struct alignas(16) DevSpectrum {
    uint32_t kind;          // SLRHIP_SPECTRUM_*
    uint32_t numPoints;     // UPSAMPLED: 3 or 4 data points (0 = outside the grid)
    uint32_t numSamples;
    uint32_t dataOffset;    // into the float pool
    float scale, lambdaMin, lambdaMax;
    uint32_t cellOffset;    // IRREGULAR: pool offset of the 472-byte search table built at upload, 0xFFFFFFFF = none
};
static_assert(sizeof(DevSpectrum) == 32, "DevSpectrum layout");

// One slrhip_texture (checkerboard and image textures, pt_tex.h), 64 B = four float4; the RGB build's values of a CHECKER_SPECTRUM's two
// spectra are resolved at upload, the spectral build reads the spectrum indices.  IMAGE_SPECTRUM: spec0 / spec1 = width / height,
// pad = index of the image's first texel in DevScene::texTexels.
struct alignas(16) DevTexture {
    uint32_t kind; float ox, oy, sx;
    float sy, v0, v1; int32_t spec0;
    int32_t spec1; float rgb0[3];
    float rgb1[3]; uint32_t pad;
};
static_assert(sizeof(DevTexture) == 64, "DevTexture layout");

// Per material, for scenes with textures: the texture behind each spectrum slot (-1 = constant) and the normal map (-1 = none)
struct alignas(16) DevMatTex {
    int32_t slot[3];
    int32_t normalMap;
};
static const uint32_t kMatTexturedBit = 0x100u;      // ORed into DevMaterial(S)::type when any DevMatTex field is set

// PerspectiveCamera constants (PerspectiveCamera.cpp:15-24) computed on the host with the same
// libm the reference uses, so the device never evaluates tan/pow.
struct DevCamera {
    float mat[16];          // local -> world, column-major
    float matInv[16];
    float lensRadius, imgPlaneDistance, objPlaneDistance, opWidth;
    float opHeight, imgPlaneArea, areaPDF, sensitivity;
};

} // namespace slrhip
