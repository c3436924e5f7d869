"""Synthetic scene builders for the path-tracing hot path (numpy -> flat slrhip scene).

Every BASELINE scene of the reference needs assets that are not shipped
(models/*.assbin, images/*.exr: TestScenes/Cornell_Box_Spheres.txt:114,127), so the
scenes are synthesised here and the SAME arrays are handed to the HIP library, the
CPU oracle and the compiled reference (oracle/_ref).  This replaces, for the asset-free
subset, what libSLRSceneGraph does before render() (createMesh TriangleMeshNode.cpp:68-112,
static transforms baked into vertices, one SingleSurfaceObject per triangle).
"""
import math

import numpy as np

from . import abi, spectra


def _translate(x, y, z):
    m = np.eye(4)
    m[:3, 3] = (x, y, z)
    return m


def _scale(s):
    m = np.eye(4)
    m[0, 0] = m[1, 1] = m[2, 2] = s
    return m


def _rotate(angle, axis):
    # libSLR/BasicTypes/Matrix4x4.cpp:111-135 (right-handed axis-angle)
    a = np.asarray(axis, dtype=np.float64)
    a = a / np.linalg.norm(a)
    c, s = math.cos(angle), math.sin(angle)
    x, y, z = a
    m = np.eye(4)
    m[:3, :3] = [[x * x * (1 - c) + c, x * y * (1 - c) - z * s, z * x * (1 - c) + y * s],
                 [x * y * (1 - c) + z * s, y * y * (1 - c) + c, y * z * (1 - c) - x * s],
                 [z * x * (1 - c) - y * s, y * z * (1 - c) + x * s, z * z * (1 - c) + c]]
    return m


def make_camera(local_to_world, aspect, fov_y, lens_radius, img_dist, obj_dist, sensitivity=0.0):
    cam = abi.Camera()
    m = np.asarray(local_to_world, dtype=np.float64)
    mi = np.linalg.inv(m)
    # column-major storage like Matrix4x4Template (m[c*4+r])
    cam.local_to_world[:] = m.T.astype(np.float32).reshape(-1).tolist()
    cam.world_to_local[:] = mi.T.astype(np.float32).reshape(-1).tolist()
    cam.aspect, cam.fov_y, cam.lens_radius = aspect, fov_y, lens_radius
    cam.img_plane_distance, cam.obj_plane_distance, cam.sensitivity = img_dist, obj_dist, sensitivity
    return cam


class SceneBuilder:
    def __init__(self):
        self.vertices, self.triangles, self.materials = [], [], []     # vertices / triangles: lists of structured-array chunks
        self.num_vertices = 0
        self.sset = spectra.SpectrumSet()     # every constant carries its RGB value and its spectral descriptor
        self.textures = []
        self.texture_texels, self.texture_texels_uvs = [], []
        self.instances = []

    # --- spectra (the scene language's Spectrum(...) overloads, libSLRSceneGraph/API.cpp:286-441) -------
    def spectrum_rgb(self, r, g, b, uvs=None):
        """RGB-mode-only constant (no spectral descriptor: spectral contexts reject it)."""
        rec = np.zeros((), dtype=abi.spectrum_dtype)
        rec["kind"] = abi.SPEC_RGB_ONLY
        rec["rgb"] = (r, g, b)
        return self.sset._append(rec, [])

    def spectrum_srgb_nonlinear(self, r, g, b):
        """Spectrum(r, g, b): Reflectance in non-linear sRGB (API.cpp:62-63,295-296).  RGB mode: sRGB_degamma
        (API.cpp:1286-1291); spectral mode: UpsampledContinuousSpectrum (API.cpp:1139-1141)."""
        return self.sset.reflectance_srgb(r, g, b)

    def spectrum_grey(self, v):
        """Spectrum("Reflectance", v): linear grey (API.cpp:327)."""
        return self.sset.reflectance_grey(v)

    def spectrum_d65(self, scale, rgb):
        """Spectrum("ID": "D65") * scale (API.cpp:405-406,443-462); `rgb` is its RGB-mode value."""
        return self.sset.d65(scale, rgb)

    def spectrum_ior(self, name, which, rgb):
        """Spectrum("ID": name, which): eta (0) or k (1) of spectrum_library.cpp; `rgb` is its RGB-mode value."""
        return self.sset.ior(name, which, rgb)

    # --- textures (libSLR/Textures/checker_board_textures.h; mapping = OffsetAndScale2DMapping, Core/textures.h:32-42) -------
    def _texture(self, kind, offset, scale, spectrum=(-1, -1), value=(0.0, 0.0)):
        rec = np.zeros((), dtype=abi.texture_dtype)
        rec["kind"], rec["offset"], rec["scale"], rec["spectrum"], rec["value"] = kind, offset, scale, spectrum, value
        self.textures.append(rec)
        return len(self.textures) - 1

    def checker_spectrum(self, s0, s1, offset=(0.0, 0.0), scale=(1.0, 1.0)):
        """CheckerBoardSpectrumTexture over two constant spectra; returns the VALUE to put in a material's spectrum slot."""
        return abi.texture_ref(self._texture(abi.TEX_CHECKER_SPECTRUM, offset, scale, spectrum=(s0, s1)))

    def checker_float(self, v0, v1, offset=(0.0, 0.0), scale=(1.0, 1.0)):
        """CheckerBoardFloatTexture (texture index: use as alpha_map=)."""
        return self._texture(abi.TEX_CHECKER_FLOAT, offset, scale, value=(v0, v1))

    def checker_normal(self, step_width, reverse=False, offset=(0.0, 0.0), scale=(1.0, 1.0)):
        """CheckerBoardNormal3DTexture (texture index: use as normal_map=)."""
        return self._texture(abi.TEX_CHECKER_NORMAL, offset, scale, value=(step_width, 1.0 if reverse else 0.0))

    # --- instancing (TransformedSurfaceObject, Core/SurfaceObject.cpp:303-392) -----------------------------------------
    def num_triangles(self):
        return sum(len(t) for t in self.triangles)

    def add_instance(self, first_triangle, num_triangles, transform):
        """Places the mesh = triangles [first, first + num) (given in the mesh's local space) with a 4x4 local-to-world matrix;
        the inverse is computed here in double and rounded (the reference's StaticTransform(mat) inverts in float: both
        matrices are scene inputs, the same on every side)."""
        m = np.asarray(transform, dtype=np.float64)
        rec = np.zeros((), dtype=abi.instance_dtype)
        rec["first_triangle"], rec["num_triangles"] = first_triangle, num_triangles
        rec["local_to_world"] = m.T.astype(np.float32).reshape(16)              # column-major
        rec["world_to_local"] = np.linalg.inv(m).T.astype(np.float32).reshape(16)
        self.instances.append(rec)
        return len(self.instances) - 1

    def image_spectrum(self, texels_rgb, texels_uvs, offset=(0.0, 0.0), scale=(1.0, 1.0)):
        """ImageSpectrumTexture (Textures/image_textures.cpp:13-79) over an OffsetAndScale2DMapping: nearest texel, wrap by fmod.
        texels_rgb [h][w][3]: what the RGB build's look-up returns (8-bit images: byte / 255); texels_uvs [h][w][3]: the (u, v, s)
        the spectral build stores for the same image (halves).  Returns the VALUE to put in a material's spectrum slot."""
        rgb = np.asarray(texels_rgb, dtype=np.float32)
        uvs = np.asarray(texels_uvs, dtype=np.float32)
        assert rgb.shape == uvs.shape and rgb.ndim == 3 and rgb.shape[2] == 3
        first = sum(len(t) for t in self.texture_texels)
        self.texture_texels.append(rgb.reshape(-1, 3))
        self.texture_texels_uvs.append(uvs.reshape(-1, 3))
        t = self._texture(abi.TEX_IMAGE_SPECTRUM, offset, scale)
        self.textures[t]["reserved"] = (rgb.shape[1], rgb.shape[0], first)
        return abi.texture_ref(t)

    # --- materials -----------------------------------------------------------------
    def material(self, mtype, spectra=(-1, -1, -1), param=-1.0, emittance=-1, param2=0.0, normal_map=None, alpha_map=None):
        rec = np.zeros((), dtype=abi.material_dtype)
        rec["type"], rec["spectrum"], rec["param"], rec["emittance"], rec["param2"] = mtype, spectra, param, emittance, param2
        rec["reserved"] = (0 if normal_map is None else normal_map + 1) | (0 if alpha_map is None else (alpha_map + 1) << 16)
        self.materials.append(rec)
        return len(self.materials) - 1

    def with_maps(self, material, normal_map=None, alpha_map=None):
        """A copy of `material` with a normal map (BumpSingleSurfaceObject) and / or an alpha texture (Triangle::m_alphaTex): the
        material group of libSLRSceneGraph's TriangleMeshNode pairs a surface material with both."""
        rec = self.materials[material].copy()
        rec["reserved"] = (0 if normal_map is None else normal_map + 1) | (0 if alpha_map is None else (alpha_map + 1) << 16)
        self.materials.append(rec)
        return len(self.materials) - 1

    def matte(self, refl, sigma=-1.0, emittance=-1):
        return self.material(abi.MAT_MATTE, (refl, -1, -1), sigma, emittance)

    def metal(self, coeff, eta, k):
        return self.material(abi.MAT_METAL, (coeff, eta, k))

    def glass(self, coeff, eta_ext, eta_int):
        return self.material(abi.MAT_GLASS, (coeff, eta_ext, eta_int))

    def ward(self, refl, aniso_x, aniso_y):
        """createSurfaceMaterial("Ward", (R, anisoX, anisoY)) -> ModifiedWardDurBRDF (API.cpp:522-535)."""
        return self.material(abi.MAT_WARD, (refl, -1, -1), aniso_x, param2=aniso_y)

    def ashikhmin(self, rd, rs, nu, nv):
        """createSurfaceMaterial("Ashikhmin", (Rd, Rs, nx, ny)) -> AshikhminShirleyBRDF(Rs, Rd, nu, nv) (API.cpp:537-552)."""
        return self.material(abi.MAT_ASHIKHMIN, (rs, rd, -1), nu, param2=nv)

    def summed(self, m0, m1, inverse=(False, False), emittance=-1):
        """createSurfaceMaterial("sum", (m0, m1)) -> MultiBSDF of the two components' BSDFs (API.cpp "sum",
        SummedSurfaceMaterial.cpp:13-20); inverse[i] wraps component i as createSurfaceMaterial("inverse", (m_i,))."""
        return self.material(abi.MAT_MULTI, (m0, m1, int(bool(inverse[0])) | int(bool(inverse[1])) << 1), 1.0, emittance, 1.0)

    def mixed(self, m0, m1, factor, inverse=(False, False), emittance=-1):
        """createSurfaceMaterial("mix", (m0, m1, factor)) with a constant factor (MixedSurfaceMaterial.cpp:14-22):
        the components receive scale 1 - factor and factor."""
        f = np.float32(factor)
        return self.material(abi.MAT_MULTI, (m0, m1, int(bool(inverse[0])) | int(bool(inverse[1])) << 1), np.float32(1.0) - f, emittance, f)

    def microfacet_metal(self, eta, k, alpha):
        return self.material(abi.MAT_MF_METAL, (-1, eta, k), alpha)

    def microfacet_glass(self, eta_ext, eta_int, alpha):
        return self.material(abi.MAT_MF_GLASS, (-1, eta_ext, eta_int), alpha)

    # --- geometry ------------------------------------------------------------------
    def add_mesh(self, positions, normals, tangents, texcoords, faces, material, transform=None):
        """createMesh + static transform baked into the vertices (TriangleMeshNode.cpp:68-78)."""
        p = np.asarray(positions, dtype=np.float64)
        n = np.asarray(normals, dtype=np.float64)
        t = np.asarray(tangents, dtype=np.float64)
        if transform is not None:
            m = np.asarray(transform, dtype=np.float64)
            p = p @ m[:3, :3].T + m[:3, 3]
            n = n @ np.linalg.inv(m[:3, :3])          # normals: inverse transpose
            n = n / np.linalg.norm(n, axis=1, keepdims=True)
            t = t @ m[:3, :3].T
            t = t / np.linalg.norm(t, axis=1, keepdims=True)
        base = self.num_vertices
        v = np.zeros(len(p), dtype=abi.vertex_dtype)
        v["position"], v["normal"], v["tangent"], v["texcoord"] = p, n, t, np.asarray(texcoords, dtype=np.float64)
        f = np.asarray(faces, dtype=np.int64).reshape(-1, 3)
        tri = np.zeros(len(f), dtype=abi.triangle_dtype)
        tri["v"] = f + base
        tri["material"] = material
        self.vertices.append(v)
        self.triangles.append(tri)
        self.num_vertices += len(p)

    def add_quad(self, corners, normal, tangent, material, transform=None):
        uv = [(0, 0), (1, 0), (1, 1), (0, 1)]
        self.add_mesh(corners, [normal] * 4, [tangent] * 4, uv, [(0, 1, 2), (0, 2, 3)], material, transform)

    def add_uv_sphere(self, segments, rings, material, transform=None):
        """Unit sphere at the origin, y up; stands in for models/sphere.assbin."""
        pos, nrm, tan, uv, faces = [], [], [], [], []
        for r in range(rings + 1):
            theta = math.pi * r / rings
            for s in range(segments + 1):
                phi = 2 * math.pi * s / segments
                d = (math.sin(theta) * math.cos(phi), math.cos(theta), math.sin(theta) * math.sin(phi))
                pos.append(d)
                nrm.append(d)
                tan.append((-math.sin(phi), 0.0, math.cos(phi)))
                uv.append((s / segments, r / rings))
        w = segments + 1
        for r in range(rings):
            for s in range(segments):
                a, b, c, d = r * w + s, r * w + s + 1, (r + 1) * w + s + 1, (r + 1) * w + s
                if r != 0:
                    faces.append((a, b, c))
                if r != rings - 1:
                    faces.append((a, c, d))
        self.add_mesh(pos, nrm, tan, uv, faces, material, transform)

    def add_box(self, material, transform=None):
        """Unit cube [-0.5,0.5]^3 as 6 quads (stands in for models/box.assbin)."""
        faces = [((0, 0, 1), (1, 0, 0)), ((0, 0, -1), (-1, 0, 0)), ((1, 0, 0), (0, 0, -1)),
                 ((-1, 0, 0), (0, 0, 1)), ((0, 1, 0), (1, 0, 0)), ((0, -1, 0), (1, 0, 0))]
        for n, t in faces:
            n, t = np.array(n, float), np.array(t, float)
            b = np.cross(n, t)
            c = [0.5 * n - 0.5 * t - 0.5 * b, 0.5 * n + 0.5 * t - 0.5 * b,
                 0.5 * n + 0.5 * t + 0.5 * b, 0.5 * n - 0.5 * t + 0.5 * b]
            self.add_quad(c, n, t, material, transform)

    def build(self, camera, env=None, name="scene"):
        return abi.Scene(np.concatenate(self.vertices), np.concatenate(self.triangles),
                         np.array(self.materials, dtype=abi.material_dtype),
                         np.array(self.sset.records, dtype=abi.spectrum_dtype),
                         np.array(self.sset.data, dtype=np.float32), camera, env, name,
                         textures=np.array(self.textures, dtype=abi.texture_dtype) if self.textures else None,
                         texture_texels=np.concatenate(self.texture_texels) if self.texture_texels else None,
                         texture_texels_uvs=np.concatenate(self.texture_texels_uvs) if self.texture_texels_uvs else None,
                         instances=np.array(self.instances, dtype=abi.instance_dtype) if self.instances else None)


# RGB-mode constants of the scene's named spectra.  In the reference these come from
# Spectrum::create's RGB branch (libSLRSceneGraph/API.cpp:1326-1347: integrate the table
# against the CMFs, XYZ -> sRGB, clamp).  Values below are fixed scene INPUTS of the
# synthetic scene (the same numbers go to oracle, reference and GPU).
# RGB-build values of the named spectra: what Spectrum::create makes of the sampled tables in the reference's RGB build
# (libSLRSceneGraph/API.cpp:1149-1278,1326-1369: CMF integration, XYZ -> sRGB / sRGB_E, clamp), computed by the C++ host
# function slrhip_spectrum_to_rgb from the tables dumped from the compiled reference (spectra.named_rgb).
D65_RGB = spectra.named_rgb("D65")        # ~ (98.89, 98.89, 98.88); x 4 for the Cornell light
ALUMINIUM_ETA_RGB = spectra.named_rgb("Aluminium", 0)
ALUMINIUM_K_RGB = spectra.named_rgb("Aluminium", 1)
AIR_ETA_RGB = spectra.named_rgb("Air", 0)
BK7_ETA_RGB = spectra.named_rgb("Glass_BK7", 0)


def cornell_walls(b):
    """Walls, light of TestScenes/Cornell_Box_Spheres.txt:8-107 (verbatim coordinates)."""
    red = b.matte(b.spectrum_srgb_nonlinear(0.75, 0.25, 0.25))
    blue = b.matte(b.spectrum_srgb_nonlinear(0.25, 0.25, 0.75))
    white = b.matte(b.spectrum_srgb_nonlinear(0.75, 0.75, 0.75))
    b.add_quad([(-1.5, 0, 2.55), (-1.5, 0, -2.55), (-1.5, 2.5, -2.55), (-1.5, 2.5, 2.55)], (1, 0, 0), (0, 0, -1), red)
    b.add_quad([(1.5, 0, -2.55), (1.5, 0, 2.55), (1.5, 2.5, 2.55), (1.5, 2.5, -2.55)], (-1, 0, 0), (0, 0, 1), blue)
    b.add_quad([(-1.5, 0, 2.55), (1.5, 0, 2.55), (1.5, 0, -2.55), (-1.5, 0, -2.55)], (0, 1, 0), (1, 0, 0), white)
    b.add_quad([(-1.5, 0, -2.55), (1.5, 0, -2.55), (1.5, 2.5, -2.55), (-1.5, 2.5, -2.55)], (0, 0, 1), (1, 0, 0), white)
    b.add_quad([(-1.5, 2.5, -2.55), (1.5, 2.5, -2.55), (1.5, 2.5, 2.55), (-1.5, 2.5, 2.55)], (0, -1, 0), (1, 0, 0), white)
    light = b.matte(b.spectrum_srgb_nonlinear(0.9, 0.9, 0.9), emittance=b.spectrum_d65(4.0, D65_RGB))
    b.add_quad([(-0.5, 2.499, -0.5), (0.5, 2.499, -0.5), (0.5, 2.499, 0.5), (-0.5, 2.499, 0.5)], (0, -1, 0), (1, 0, 0), light)


def cornell_camera(aspect):
    # Cornell_Box_Spheres.txt:132-138
    m = _translate(0.0, 1.689714, 6.70284) @ _rotate(3.1415926536, (0, 1, 0)) @ _rotate(0.0563936, (1, 0, 0))
    return make_camera(m, aspect, 0.4807705238, 0.025, 1.0, 6.3)


def cornell_box_spheres(aspect=4.0 / 3.0, segments=48, rings=24, right="glass"):
    """Config 1/2 of BASELINE.json: Cornell box + aluminium mirror sphere + a second sphere
    (BK7 glass as in Cornell_Box_Spheres.txt:120-130, or Lambert for 'Lambert+specular only')."""
    b = SceneBuilder()
    cornell_walls(b)
    one = b.spectrum_grey(1.0)           # Spectrum("Reflectance", 1.0): linear grey, API.cpp:327
    left = b.metal(one, b.spectrum_ior("Aluminium", 0, ALUMINIUM_ETA_RGB), b.spectrum_ior("Aluminium", 1, ALUMINIUM_K_RGB))
    b.add_uv_sphere(segments, rings, left, _translate(-0.7, 0, -1.05) @ _scale(0.5) @ _translate(0, 1, 0))
    if right == "glass":
        mat = b.glass(b.spectrum_grey(0.999), b.spectrum_ior("Air", 0, AIR_ETA_RGB), b.spectrum_ior("Glass_BK7", 0, BK7_ETA_RGB))
    else:
        mat = b.matte(b.spectrum_srgb_nonlinear(0.75, 0.75, 0.25))
    b.add_uv_sphere(segments, rings, mat, _translate(0.7, 0, 0) @ _scale(0.5) @ _translate(0, 1, 0))
    return b.build(cornell_camera(aspect), name="cornell_box_spheres_" + right)


def cornell_textured(aspect=1.0, segments=16, rings=8):
    """SURVEY 8 row f3, the textured half: the Cornell walls with a CHECKERBOARD floor (two reflectances, 4 x 6 squares through an
    offset-and-scale mapping), a bump-mapped Oren-Nayar sphere (CheckerBoardNormal3DTexture through BumpSingleSurfaceObject), a
    quad in front of the back wall whose alpha texture (CheckerBoardFloatTexture 1 / 0) cuts half of its squares away, and a mirror
    sphere whose coefficient is a checker of two greys."""
    b = SceneBuilder()
    red = b.matte(b.spectrum_srgb_nonlinear(0.75, 0.25, 0.25))
    blue = b.matte(b.spectrum_srgb_nonlinear(0.25, 0.25, 0.75))
    white = b.matte(b.spectrum_srgb_nonlinear(0.75, 0.75, 0.75))
    floor = b.matte(b.checker_spectrum(b.spectrum_srgb_nonlinear(0.8, 0.8, 0.8), b.spectrum_srgb_nonlinear(0.2, 0.3, 0.2), (0.125, 0.0), (2.0, 3.0)))
    b.add_quad([(-1.5, 0, 2.55), (-1.5, 0, -2.55), (-1.5, 2.5, -2.55), (-1.5, 2.5, 2.55)], (1, 0, 0), (0, 0, -1), red)
    b.add_quad([(1.5, 0, -2.55), (1.5, 0, 2.55), (1.5, 2.5, 2.55), (1.5, 2.5, -2.55)], (-1, 0, 0), (0, 0, 1), blue)
    b.add_quad([(-1.5, 0, 2.55), (1.5, 0, 2.55), (1.5, 0, -2.55), (-1.5, 0, -2.55)], (0, 1, 0), (1, 0, 0), floor)
    b.add_quad([(-1.5, 0, -2.55), (1.5, 0, -2.55), (1.5, 2.5, -2.55), (-1.5, 2.5, -2.55)], (0, 0, 1), (1, 0, 0), white)
    b.add_quad([(-1.5, 2.5, -2.55), (1.5, 2.5, -2.55), (1.5, 2.5, 2.55), (-1.5, 2.5, 2.55)], (0, -1, 0), (1, 0, 0), white)
    light = b.matte(b.spectrum_srgb_nonlinear(0.9, 0.9, 0.9), emittance=b.spectrum_d65(4.0, D65_RGB))
    b.add_quad([(-0.5, 2.499, -0.5), (0.5, 2.499, -0.5), (0.5, 2.499, 0.5), (-0.5, 2.499, 0.5)], (0, -1, 0), (1, 0, 0), light)
    bumpy = b.with_maps(b.matte(b.spectrum_srgb_nonlinear(0.7, 0.6, 0.3), sigma=0.4), normal_map=b.checker_normal(0.3, False, (0.0, 0.0), (6.0, 3.0)))
    b.add_uv_sphere(segments, rings, bumpy, _translate(0.7, 0, 0) @ _scale(0.5) @ _translate(0, 1, 0))
    coeff = b.checker_spectrum(b.spectrum_grey(1.0), b.spectrum_grey(0.4), (0.0, 0.0), (4.0, 2.0))
    mirror = b.metal(coeff, b.spectrum_ior("Aluminium", 0, ALUMINIUM_ETA_RGB), b.spectrum_ior("Aluminium", 1, ALUMINIUM_K_RGB))
    b.add_uv_sphere(segments, rings, mirror, _translate(-0.7, 0, -1.05) @ _scale(0.5) @ _translate(0, 1, 0))
    lattice = b.with_maps(b.matte(b.spectrum_srgb_nonlinear(0.3, 0.5, 0.8)), alpha_map=b.checker_float(1.0, 0.0, (0.0, 0.0), (3.0, 3.0)))
    b.add_quad([(-1.0, 0.2, -1.9), (1.0, 0.2, -1.9), (1.0, 1.8, -1.9), (-1.0, 1.8, -1.9)], (0, 0, 1), (1, 0, 0), lattice)
    return b.build(cornell_camera(aspect), name="cornell_textured")


def cornell_instanced(aspect=1.0, segments=10, rings=5, copies=6):
    """Instanced geometry (TransformedSurfaceObject over a mesh aggregate, SurfaceObject.cpp:303-392): ONE unit UV sphere mesh and
    ONE small box mesh in their local spaces, placed several times in the Cornell box with rotations, NON-UNIFORM scales and
    translations (so that the inverse-transpose of the normals and the un-normalised local ray direction are on the path), next to
    the loose wall triangles and the area light.  Materials: matte, mirror (delta lobe through the instance transform)."""
    b = SceneBuilder()
    cornell_walls(b)
    mirror = b.metal(b.spectrum_grey(0.9), b.spectrum_ior("Aluminium", 0, ALUMINIUM_ETA_RGB), b.spectrum_ior("Aluminium", 1, ALUMINIUM_K_RGB))
    orange = b.matte(b.spectrum_srgb_nonlinear(0.8, 0.45, 0.15), sigma=0.3)
    first_s = b.num_triangles()
    b.add_uv_sphere(segments, rings, orange, np.eye(4))
    n_s = b.num_triangles() - first_s
    first_b = b.num_triangles()
    b.add_box(mirror, _scale(0.5))
    n_b = b.num_triangles() - first_b
    r = np.random.default_rng(11)
    for i in range(copies):
        ang = float(r.uniform(0, 2 * np.pi))
        c, s_ = np.cos(ang), np.sin(ang)
        rot = np.array([[c, 0, s_, 0], [0, 1, 0, 0], [-s_, 0, c, 0], [0, 0, 0, 1]])
        sc = np.diag([float(r.uniform(0.2, 0.45)), float(r.uniform(0.2, 0.5)), float(r.uniform(0.2, 0.45)), 1.0])
        pos = _translate(float(r.uniform(-1.0, 1.0)), float(r.uniform(0.35, 1.6)), float(r.uniform(-1.8, 0.8)))
        if i % 3 == 2:
            b.add_instance(first_b, n_b, pos @ rot @ sc)
        else:
            b.add_instance(first_s, n_s, pos @ rot @ sc)
    return b.build(cornell_camera(aspect), name="cornell_instanced")


def synthetic_image(width, height, seed):
    """A deterministic test image in both stored forms: (texels_rgb, texels_uvs).  RGB: 8-bit values / 255, what the RGB build's
    ImageSpectrumTexture::evaluate returns for an RGB8x3 image (image_textures.cpp:36-42).  uvs: UpsampledContinuousSpectrum's
    (u, v, s) of the same colour taken as a linear-sRGB reflectance, rounded to binary16 as the spectral build's image stores it
    (Core/Image.h:39-40) — the reference converts at image load, outside the boundary; here both forms are scene inputs."""
    r = np.random.default_rng(seed)
    bytes_ = r.integers(0, 256, size=(height, width, 3))
    yy, xx = np.mgrid[0:height, 0:width]
    bytes_ = (bytes_ // 2 + 64 * (((xx // 2) + (yy // 2)) % 2)[..., None]).clip(0, 255)      # coarse checks under the noise, so that the mapping shows
    rgb = (bytes_.astype(np.float32) / np.float32(255.0)).astype(np.float32)
    uvs = np.zeros_like(rgb)
    for y in range(height):
        for x in range(width):
            u, v, scale = spectra.upsample(spectra.REFLECTANCE, spectra.SRGB, *[float(c) for c in rgb[y, x]])
            uvs[y, x] = (u, v, np.float32(scale) * spectra.EQUAL_ENERGY_REFLECTANCE)      # the image stores s = X + Y + Z; evaluate divides it back
    return rgb, _f16(uvs)


def cornell_image_textured(aspect=1.0, segments=12, rings=6):
    """SURVEY 8 row f3, image textures in material slots (ImageSpectrumTexture, Textures/image_textures.cpp:13-79): the Cornell walls
    with an IMAGE on the floor (16 x 12 texels through an offset-and-scale mapping that repeats it, so the fmod wrap is on the
    path), a matte sphere wrapped in a second image and a mirror sphere whose coefficient is an image."""
    b = SceneBuilder()
    red = b.matte(b.spectrum_srgb_nonlinear(0.75, 0.25, 0.25))
    blue = b.matte(b.spectrum_srgb_nonlinear(0.25, 0.25, 0.75))
    white = b.matte(b.spectrum_srgb_nonlinear(0.75, 0.75, 0.75))
    floor = b.matte(b.image_spectrum(*synthetic_image(16, 12, 5), offset=(0.125, -0.25), scale=(2.0, 3.0)))
    b.add_quad([(-1.5, 0, 2.55), (-1.5, 0, -2.55), (-1.5, 2.5, -2.55), (-1.5, 2.5, 2.55)], (1, 0, 0), (0, 0, -1), red)
    b.add_quad([(1.5, 0, -2.55), (1.5, 0, 2.55), (1.5, 2.5, 2.55), (1.5, 2.5, -2.55)], (-1, 0, 0), (0, 0, 1), blue)
    b.add_quad([(-1.5, 0, 2.55), (1.5, 0, 2.55), (1.5, 0, -2.55), (-1.5, 0, -2.55)], (0, 1, 0), (1, 0, 0), floor)
    b.add_quad([(-1.5, 0, -2.55), (1.5, 0, -2.55), (1.5, 2.5, -2.55), (-1.5, 2.5, -2.55)], (0, 0, 1), (1, 0, 0), white)
    b.add_quad([(-1.5, 2.5, -2.55), (1.5, 2.5, -2.55), (1.5, 2.5, 2.55), (-1.5, 2.5, 2.55)], (0, -1, 0), (1, 0, 0), white)
    light = b.matte(b.spectrum_srgb_nonlinear(0.9, 0.9, 0.9), emittance=b.spectrum_d65(4.0, D65_RGB))
    b.add_quad([(-0.5, 2.499, -0.5), (0.5, 2.499, -0.5), (0.5, 2.499, 0.5), (-0.5, 2.499, 0.5)], (0, -1, 0), (1, 0, 0), light)
    wrapped = b.matte(b.image_spectrum(*synthetic_image(8, 8, 6)), sigma=0.3)
    b.add_uv_sphere(segments, rings, wrapped, _translate(0.7, 0, 0) @ _scale(0.5) @ _translate(0, 1, 0))
    mirror = b.metal(b.image_spectrum(*synthetic_image(6, 4, 7), scale=(3.0, 2.0)), b.spectrum_ior("Aluminium", 0, ALUMINIUM_ETA_RGB),
                     b.spectrum_ior("Aluminium", 1, ALUMINIUM_K_RGB))
    b.add_uv_sphere(segments, rings, mirror, _translate(-0.7, 0, -1.05) @ _scale(0.5) @ _translate(0, 1, 0))
    return b.build(cornell_camera(aspect), name="cornell_image_textured")


def tiny_box(aspect=1.0):
    """Walls + light only (12 triangles): the smallest closed scene, for fast tests."""
    b = SceneBuilder()
    cornell_walls(b)
    return b.build(cornell_camera(aspect), name="tiny_box")


# RGB-mode eta / k of titanium (Cornell_Box_Boxes.txt:32-33 uses the "Titanium" IOR table); fixed scene inputs.
TITANIUM_ETA_RGB = spectra.named_rgb("Titanium", 0)
TITANIUM_K_RGB = spectra.named_rgb("Titanium", 1)


def cornell_lobes(kind, aspect=1.0, segments=16, rings=8):
    """Cornell walls + one sphere carrying the lobe under test: 'oren_nayar', 'ggx_metal', 'ggx_glass', 'ward' or 'ashikhmin'."""
    b = SceneBuilder()
    cornell_walls(b)
    if kind == "oren_nayar":
        m = b.matte(b.spectrum_srgb_nonlinear(0.7, 0.6, 0.3), sigma=0.6)
    elif kind == "ggx_metal":
        m = b.microfacet_metal(b.spectrum_ior("Titanium", 0, TITANIUM_ETA_RGB), b.spectrum_ior("Titanium", 1, TITANIUM_K_RGB), 0.1)
    elif kind == "ggx_glass":
        m = b.microfacet_glass(b.spectrum_ior("Air", 0, AIR_ETA_RGB), b.spectrum_ior("Glass_BK7", 0, BK7_ETA_RGB), 0.2)
    elif kind == "ward":
        m = b.ward(b.spectrum_srgb_nonlinear(0.8, 0.7, 0.4), 0.15, 0.4)
    elif kind == "ashikhmin":
        m = b.ashikhmin(b.spectrum_srgb_nonlinear(0.6, 0.25, 0.2), b.spectrum_grey(0.05), 100.0, 20.0)
    else:
        raise ValueError(kind)
    b.add_uv_sphere(segments, rings, m, _translate(-0.3, 0, -0.5) @ _scale(0.6) @ _translate(0, 1, 0))
    return b.build(cornell_camera(aspect), name="cornell_" + kind)


MATERIAL_ZOO = ("lambert", "oren_nayar", "mirror", "glass", "ggx_metal_smooth", "ggx_metal_rough", "ggx_glass", "ward", "ashikhmin",
                "ashikhmin_isotropic", "emitter_over_lambert",
                # MultiBSDF (sum / mix / inverse): RTC3.txt:13-18 shape first, then lobes of different kinds side by side
                "sum_lambert_inverse_lambert", "sum_lambert_ward", "mix_ggx_metal_lambert", "sum_mirror_lambert", "mix_glass_ashikhmin",
                "sum_oren_nayar_inverse_ggx_metal", "mix_ggx_glass_inverse_ward", "emitter_over_sum")


def material_zoo():
    """One material per lobe (and parameter regime) the integrator supports, for the function-level BSDF known-answer
    tests: a closed Cornell box (so the scene is valid for every entry point) with a small quad per material.
    Returns (scene, {name: material index})."""
    b = SceneBuilder()
    cornell_walls(b)
    one = b.spectrum_grey(1.0)
    ti = (b.spectrum_ior("Titanium", 0, TITANIUM_ETA_RGB), b.spectrum_ior("Titanium", 1, TITANIUM_K_RGB))
    air, bk7 = b.spectrum_ior("Air", 0, AIR_ETA_RGB), b.spectrum_ior("Glass_BK7", 0, BK7_ETA_RGB)
    mats = {
        "lambert": b.matte(b.spectrum_srgb_nonlinear(0.75, 0.5, 0.25)),
        "oren_nayar": b.matte(b.spectrum_srgb_nonlinear(0.7, 0.6, 0.3), sigma=0.6),
        "mirror": b.metal(one, b.spectrum_ior("Aluminium", 0, ALUMINIUM_ETA_RGB), b.spectrum_ior("Aluminium", 1, ALUMINIUM_K_RGB)),
        "glass": b.glass(b.spectrum_grey(0.999), air, bk7),
        "ggx_metal_smooth": b.microfacet_metal(ti[0], ti[1], 0.05),
        "ggx_metal_rough": b.microfacet_metal(ti[0], ti[1], 0.5),
        "ggx_glass": b.microfacet_glass(air, bk7, 0.2),
        "ward": b.ward(b.spectrum_srgb_nonlinear(0.8, 0.7, 0.4), 0.15, 0.4),
        "ashikhmin": b.ashikhmin(b.spectrum_srgb_nonlinear(0.6, 0.25, 0.2), b.spectrum_grey(0.05), 100.0, 20.0),
        "ashikhmin_isotropic": b.ashikhmin(b.spectrum_srgb_nonlinear(0.3, 0.5, 0.2), b.spectrum_grey(0.2), 30.0, 30.0),
        "emitter_over_lambert": b.matte(b.spectrum_srgb_nonlinear(0.9, 0.9, 0.9), emittance=b.spectrum_d65(2.0, D65_RGB)),
    }
    thin = b.matte(b.spectrum_srgb_nonlinear(0.2, 0.6, 0.3))
    mats.update({
        "sum_lambert_inverse_lambert": b.summed(mats["lambert"], thin, inverse=(False, True)),
        "sum_lambert_ward": b.summed(mats["lambert"], mats["ward"]),
        "mix_ggx_metal_lambert": b.mixed(mats["ggx_metal_rough"], mats["lambert"], 0.3),
        "sum_mirror_lambert": b.summed(mats["mirror"], thin),
        "mix_glass_ashikhmin": b.mixed(mats["glass"], mats["ashikhmin_isotropic"], 0.625),
        "sum_oren_nayar_inverse_ggx_metal": b.summed(mats["oren_nayar"], mats["ggx_metal_rough"], inverse=(False, True)),
        "mix_ggx_glass_inverse_ward": b.mixed(mats["ggx_glass"], mats["ward"], 0.25, inverse=(False, True)),
        "emitter_over_sum": b.summed(thin, mats["ward"], emittance=b.spectrum_d65(2.0, D65_RGB)),
    })
    assert tuple(mats) == MATERIAL_ZOO
    for i, m in enumerate(mats.values()):
        x, z = -1.2 + 0.22 * (i % 11), -0.3 * (i // 11)
        b.add_quad([(x, 0.01, z), (x + 0.2, 0.01, z), (x + 0.2, 0.01, z - 0.2), (x, 0.01, z - 0.2)], (0, 1, 0), (1, 0, 0), m)
    return b.build(cornell_camera(1.0), name="material_zoo"), mats


def material_zoo_nested():
    """MultiBSDFs whose components are MultiBSDFs (SummedSurfaceMaterial / MixedSurfaceMaterial of summed / mixed materials:
    one level of nesting, up to four lobes — MultiBSDF.cpp:20-59 calling itself through the BSDF interface), for the
    function-level known-answer tests.  Returns (scene, {name: material index}); the libm-free entries are bit-exact on the GPU."""
    b = SceneBuilder()
    cornell_walls(b)
    one = b.spectrum_grey(1.0)
    lambert = b.matte(b.spectrum_srgb_nonlinear(0.75, 0.5, 0.25))
    thin = b.matte(b.spectrum_srgb_nonlinear(0.2, 0.6, 0.3))
    oren = b.matte(b.spectrum_srgb_nonlinear(0.7, 0.6, 0.3), sigma=0.6)
    mirror = b.metal(one, b.spectrum_ior("Aluminium", 0, ALUMINIUM_ETA_RGB), b.spectrum_ior("Aluminium", 1, ALUMINIUM_K_RGB))
    glass = b.glass(b.spectrum_grey(0.999), b.spectrum_ior("Air", 0, AIR_ETA_RGB), b.spectrum_ior("Glass_BK7", 0, BK7_ETA_RGB))
    ward = b.ward(b.spectrum_srgb_nonlinear(0.8, 0.7, 0.4), 0.15, 0.4)
    ggx = b.microfacet_metal(b.spectrum_ior("Titanium", 0, TITANIUM_ETA_RGB), b.spectrum_ior("Titanium", 1, TITANIUM_K_RGB), 0.5)
    mix_lo = b.mixed(lambert, oren, 0.4)
    sum_lm = b.summed(lambert, mirror)
    sum_oi = b.summed(oren, thin, inverse=(False, True))
    mix_gw = b.mixed(ggx, ward, 0.7)
    mats = {
        "sum_of_mix_and_mirror": b.summed(mix_lo, mirror),                    # nested on the left
        "mix_of_lambert_and_sum_with_inverse": b.mixed(lambert, sum_oi, 0.35),  # nested on the right, an InverseBSDF inside
        "sum_of_two_sums": b.summed(sum_lm, sum_oi),                          # four lobes, delta + diffuse + transmitted
        "mix_of_mix_and_glass": b.mixed(mix_lo, glass, 0.25),                 # a two-sided delta lobe next to a nested pair
        "sum_of_mix_ggx_ward_and_lambert": b.summed(mix_gw, lambert),         # float-libm lobes nested
        "mix_of_two_mixes": b.mixed(mix_lo, mix_gw, 0.6),
    }
    for i, m in enumerate(mats.values()):
        x, z = -1.2 + 0.22 * (i % 11), -0.3 * (i // 11)
        b.add_quad([(x, 0.01, z), (x + 0.2, 0.01, z), (x + 0.2, 0.01, z - 0.2), (x, 0.01, z - 0.2)], (0, 1, 0), (1, 0, 0), m)
    return b.build(cornell_camera(1.0), name="material_zoo_nested"), mats


NESTED_LIBM_FREE = ("sum_of_mix_and_mirror", "mix_of_lambert_and_sum_with_inverse", "sum_of_two_sums", "mix_of_mix_and_glass")


def cornell_multi_nested(aspect=1.0, segments=12, rings=6):
    """Nested MultiBSDF materials in a Cornell box, no float-libm lobe on the path (bit-exact on the GPU): a sphere of
    sum(mix(Lambert, Oren-Nayar; 0.4), specular aluminium), a sphere of mix(mix(Lambert, Oren-Nayar), glass; 0.25) and a
    free-standing sheet of sum(sum(Lambert, mirror), sum(Oren-Nayar, inverse(Lambert))) — four lobes."""
    b = SceneBuilder()
    cornell_walls(b)
    orange = b.matte(b.spectrum_srgb_nonlinear(0.8, 0.45, 0.15))
    green = b.matte(b.spectrum_srgb_nonlinear(0.2, 0.6, 0.3))
    rough = b.matte(b.spectrum_srgb_nonlinear(0.7, 0.7, 0.75), sigma=0.5)
    al = b.metal(b.spectrum_grey(0.6), b.spectrum_ior("Aluminium", 0, ALUMINIUM_ETA_RGB), b.spectrum_ior("Aluminium", 1, ALUMINIUM_K_RGB))
    glass = b.glass(b.spectrum_grey(0.999), b.spectrum_ior("Air", 0, AIR_ETA_RGB), b.spectrum_ior("Glass_BK7", 0, BK7_ETA_RGB))
    blend = b.mixed(orange, rough, 0.4)
    lacquer = b.summed(blend, al)
    frosted = b.mixed(blend, glass, 0.25)
    leaf = b.summed(b.summed(orange, al), b.summed(rough, green, inverse=(False, True)))
    b.add_uv_sphere(segments, rings, lacquer, _translate(-0.7, 0, -0.8) @ _scale(0.5) @ _translate(0, 1, 0))
    b.add_uv_sphere(segments, rings, frosted, _translate(0.75, 0, -0.2) @ _scale(0.45) @ _translate(0, 1, 0))
    b.add_quad([(-0.4, 0.0, 0.9), (0.5, 0.0, 0.6), (0.5, 1.3, 0.6), (-0.4, 1.3, 0.9)], (0.316, 0, 0.949), (0.949, 0, -0.316), leaf)
    return b.build(cornell_camera(aspect), name="cornell_multi_nested")


def bsdf_queries(n=256, seed=2024):
    """[n][12] query rows for slrhip_bsdf_queries / the oracle's bsdf_kat: outgoing direction, geometric normal (tilted up
    to ~30 degrees off the shading normal) and incoming direction in the shading frame, plus the three sample numbers.
    Both sides of the surface; the first rows are the special cases: exactly along +-z, grazing, dirIn = dirOut,
    dirIn = mirror of dirOut, and sample numbers at 0 and just below 1."""
    r = np.random.default_rng(seed)

    def dirs(k):
        v = r.normal(size=(k, 3))
        return v / np.linalg.norm(v, axis=1, keepdims=True)
    do, di = dirs(n), dirs(n)
    gn = np.tile([0.0, 0.0, 1.0], (n, 1)) + 0.25 * r.normal(size=(n, 3))
    gn /= np.linalg.norm(gn, axis=1, keepdims=True)
    u = r.random(size=(n, 3))
    do[0:4] = (0, 0, 1)
    do[4:8] = (0, 0, -1)
    di[8:12] = (0, 0, 1)
    do[12:20, 2] *= 1e-3                                           # grazing outgoing directions
    do[12:20] /= np.linalg.norm(do[12:20], axis=1, keepdims=True)
    di[20:24] = do[20:24]                                          # retro-reflection
    di[24:32] = do[24:32] * (-1, -1, 1)                            # the mirror direction
    di[32:36] = -do[32:36]                                         # straight through
    u[36:40] = 0.0
    u[40:44] = np.float32(1.0) - np.float32(2.0 ** -24)
    gn[44:48] = (0, 0, 1)
    return np.concatenate([do, gn, di, u], axis=1).astype(np.float32)


def cornell_multi(aspect=1.0, segments=16, rings=8, libm_free=False):
    """MultiBSDF materials in a Cornell box: a sphere of mix(GGX titanium, Lambert; 0.3), a sphere of
    sum(specular aluminium, Lambert) (a delta and a non-delta component side by side), and a free-standing sheet of
    sum(Lambert, inverse(Lambert)) — the reflect + diffuse-transmit material of TestScenes/RTC3.txt:13-18.
    libm_free: Oren-Nayar in place of the GGX lobe, so that no float libm call is on the path (bit-exact on the GPU)."""
    b = SceneBuilder()
    cornell_walls(b)
    if libm_free:
        ti = b.matte(b.spectrum_srgb_nonlinear(0.7, 0.7, 0.75), sigma=0.5)
    else:
        ti = b.microfacet_metal(b.spectrum_ior("Titanium", 0, TITANIUM_ETA_RGB), b.spectrum_ior("Titanium", 1, TITANIUM_K_RGB), 0.3)
    orange = b.matte(b.spectrum_srgb_nonlinear(0.8, 0.45, 0.15))
    green = b.matte(b.spectrum_srgb_nonlinear(0.2, 0.6, 0.3))
    grey = b.matte(b.spectrum_srgb_nonlinear(0.5, 0.5, 0.5))
    al = b.metal(b.spectrum_grey(0.6), b.spectrum_ior("Aluminium", 0, ALUMINIUM_ETA_RGB), b.spectrum_ior("Aluminium", 1, ALUMINIUM_K_RGB))
    coated = b.mixed(ti, orange, 0.3)
    lacquer = b.summed(al, grey)
    leaf = b.summed(grey, green, inverse=(False, True))
    b.add_uv_sphere(segments, rings, coated, _translate(-0.7, 0, -0.8) @ _scale(0.5) @ _translate(0, 1, 0))
    b.add_uv_sphere(segments, rings, lacquer, _translate(0.75, 0, -0.2) @ _scale(0.45) @ _translate(0, 1, 0))
    b.add_quad([(-0.4, 0.0, 0.9), (0.5, 0.0, 0.6), (0.5, 1.3, 0.6), (-0.4, 1.3, 0.9)], (0.316, 0, 0.949), (0.949, 0, -0.316), leaf)
    return b.build(cornell_camera(aspect), name="cornell_multi" + ("_libm_free" if libm_free else ""))


def cornell_box_boxes(aspect=1.0):
    """Config 3 of BASELINE.json, RGB variant: Cornell_Box_Boxes-shaped scene (TestScenes/Cornell_Box_Boxes.txt:7-53):
    white/red/blue walls, 0.5 x 0.5 light at y = 0.999 scaled to this box, two boxes: a GGX titanium conductor
    (alpha_g = 0.1) and a matte one."""
    b = SceneBuilder()
    cornell_walls(b)
    ti = b.microfacet_metal(b.spectrum_ior("Titanium", 0, TITANIUM_ETA_RGB), b.spectrum_ior("Titanium", 1, TITANIUM_K_RGB), 0.1)
    white = b.matte(b.spectrum_srgb_nonlinear(0.75, 0.75, 0.75))
    tall = _translate(-0.5, 0.0, -1.0) @ _rotate(0.3, (0, 1, 0)) @ _translate(0, 0.8, 0) @ np.diag([0.8, 1.6, 0.8, 1.0])
    short = _translate(0.55, 0.0, 0.2) @ _rotate(-0.35, (0, 1, 0)) @ _translate(0, 0.4, 0) @ np.diag([0.8, 0.8, 0.8, 1.0])
    b.add_box(ti, tall)
    b.add_box(white, short)
    return b.build(cornell_camera(aspect), name="cornell_box_boxes")


# ---- image-based environment light (BASELINE configs[3], IBL_Test-shaped) ------------------------------------------
def synthetic_sky(width=512, height=256):
    """A fixed-formula lat-long radiance map: horizon-to-zenith gradient, dim ground, and a small bright sun disc.
    Values are rounded to IEEE binary16 because the reference stores environment maps as RGBA16F
    (libSLR/Core/Image.h:38-40, image_textures.cpp:57-63).  Row 0 = theta 0 (+Y)."""
    v = (np.arange(height) + 0.5) / height
    u = (np.arange(width) + 0.5) / width
    theta, phi = np.pi * v[:, None], 2 * np.pi * u[None, :]
    d = np.stack([-np.sin(phi) * np.sin(theta), np.cos(theta) * np.ones_like(phi), np.cos(phi) * np.sin(theta)], axis=-1)
    up = np.clip(d[..., 1], 0, 1)
    sky = (0.25 + 0.75 * (1 - up) ** 3)[..., None] * np.array([0.35, 0.55, 1.0]) + (up ** 0.5)[..., None] * np.array([0.1, 0.15, 0.3])
    ground = np.where(d[..., 1:2] < 0, 1.0, 0.0) * np.array([0.12, 0.1, 0.08])
    img = np.where(d[..., 1:2] < 0, ground, sky)
    sun_dir = np.array([0.45, 0.6, 0.66]); sun_dir /= np.linalg.norm(sun_dir)
    cosang = d @ sun_dir
    img = img + (cosang > math.cos(math.radians(3.0)))[..., None] * np.array([60.0, 55.0, 45.0])
    return img.astype(np.float16).astype(np.float32)


def ibl_importance(texels):
    """ImageSpectrumTexture::createIBLImportanceMap before the sin(theta) factor (image_textures.cpp:81-108,131):
    the map is quarter resolution; each cell is Image2D::areaAverage over its 4x4 texel block (Core/Image.cpp:19-120:
    corner, edge and interior texels are Kahan-summed in that order with unit coefficients, divided by the area and
    stored back as binary16), then luminance = 0.222485 r + 0.716905 g + 0.060610 b in float."""
    h, w, _ = texels.shape
    mw, mh = w // 4, h // 4
    if w % 4 or h % 4:
        raise ValueError("environment map size must be a multiple of 4")
    out = np.zeros((mh, mw), np.float32)
    f = np.float32
    order = [(0, 0), (3, 0), (0, 3), (3, 3)]                                   # corners UL, UR, LL, LR (x, y)
    for x in (1, 2):
        order += [(x, 0), (x, 3)]                                               # top / bottom edges, interleaved
    for y in (1, 2):
        order += [(0, y), (3, y)]                                               # left / right edges
    order += [(x, y) for y in (1, 2) for x in (1, 2)]                           # interior
    for my in range(mh):
        for mx in range(mw):
            block = texels[4 * my:4 * my + 4, 4 * mx:4 * mx + 4]
            avg = []
            for c in range(3):
                s, comp = f(0), f(0)
                for (x, y) in order:                                            # CompensatedSum (FloatSum)
                    val = f(f(1.0) * block[y, x, c])
                    ci = f(val - comp); t = f(s + ci); comp = f(f(t - s) - ci); s = t
                avg.append(np.float32(np.float16(f(s / f(16.0)))))
            out[my, mx] = f(f(f(0.222485) * avg[0]) + f(f(0.716905) * avg[1])) + f(f(0.060610) * avg[2])
    return out


_SRGB_TO_XYZ = [(0.4124564, 0.3575761, 0.1804375), (0.2126729, 0.7151522, 0.0721750), (0.0193339, 0.1191920, 0.9503041)]


def _f16(a):
    return np.asarray(a, dtype=np.float32).astype(np.float16).astype(np.float32)


def sky_to_uvs(texels):
    """What the reference's spectral build stores for an RGB radiance image: Upsampling::sRGB_to_uvs(Illuminant) per texel
    (BasicTypes/Spectrum.h:148-171: sRGB -> XYZ with double literals, b = X + Y + Z, xy, xy_to_uv, s = b) kept as halves
    (Core/Image.h:39-40,265-310)."""
    f = np.float32
    rgb = np.asarray(texels, dtype=np.float32).astype(np.float64)
    xyz = [(_SRGB_TO_XYZ[i][0] * rgb[..., 0] + _SRGB_TO_XYZ[i][1] * rgb[..., 1] + _SRGB_TO_XYZ[i][2] * rgb[..., 2]).astype(f) for i in range(3)]
    b = (xyz[0] + xyz[1]).astype(f) + xyz[2]
    with np.errstate(divide="ignore", invalid="ignore"):
        x, y = (xyz[0] / b).astype(f), (xyz[1] / b).astype(f)
    third = f(1.0 / 3.0)
    x, y = np.where(b == 0, third, x), np.where(b == 0, third, y)
    u = (16.730260708356887 * x.astype(np.float64) + 7.7801960340706 * y.astype(np.float64) - 2.170152247475828).astype(f)
    v = (-7.530081094743006 * x.astype(np.float64) + 16.192422314095225 * y.astype(np.float64) + 1.1125529268825947).astype(f)
    return _f16(np.stack([u, v, b.astype(f)], axis=-1))


def _area_average_4x4(plane):
    """Image2D::areaAverage over each 4x4 block (Core/Image.cpp:19-120): corners, edges, interior Kahan-summed in that order with
    unit coefficients, divided by the area, stored back as binary16."""
    f = np.float32
    h, w = plane.shape
    order = [(0, 0), (3, 0), (0, 3), (3, 3)]
    for x in (1, 2):
        order += [(x, 0), (x, 3)]
    for y in (1, 2):
        order += [(0, y), (3, y)]
    order += [(x, y) for y in (1, 2) for x in (1, 2)]
    blocks = plane.reshape(h // 4, 4, w // 4, 4)
    s = np.zeros((h // 4, w // 4), f)
    comp = np.zeros_like(s)
    for (x, y) in order:
        val = blocks[:, y, :, x].astype(f)
        ci = (val - comp).astype(f)
        t = (s + ci).astype(f)
        comp = ((t - s).astype(f) - ci).astype(f)
        s = t
    return _f16(s / f(16.0))


def ibl_importance_uvs(uvs):
    """createIBLImportanceMap for a uvs image (image_textures.cpp:81-132): area-average u, v, s per 4x4 block, uvs_to_sRGB
    (Spectrum.h:174-195, Illuminant: uv_to_xy, XYZ = (x b, y b, b - X - Y), XYZ_to_sRGB), luminance in float."""
    f = np.float32
    au, av, asum = (_area_average_4x4(uvs[..., k]) for k in range(3))
    x = (0.0491440520940413 * au.astype(np.float64) - 0.02361291916573777 * av.astype(np.float64) + 0.13292069743203658).astype(f)
    y = (0.022853819546830627 * au.astype(np.float64) + 0.05077639329371236 * av.astype(np.float64) - 0.006895157122499944).astype(f)
    X, Y = (x * asum).astype(f), (y * asum).astype(f)
    Z = ((asum - X).astype(f) - Y).astype(f)
    Xd, Yd, Zd = X.astype(np.float64), Y.astype(np.float64), Z.astype(np.float64)
    r = (3.2404542 * Xd - 1.5371385 * Yd - 0.4985314 * Zd).astype(f)
    g = (-0.9692660 * Xd + 1.8760108 * Yd + 0.0415560 * Zd).astype(f)
    b = (0.0556434 * Xd - 0.2040259 * Yd + 1.0572252 * Zd).astype(f)
    return ((f(0.222485) * r).astype(f) + (f(0.716905) * g).astype(f)).astype(f) + (f(0.060610) * b).astype(f)


def ibl_test_scene(aspect=1.0, env_size=(256, 128), segments=24, rings=12, area_light=False):
    """Config 4 of BASELINE.json, IBL_Test-shaped (TestScenes/IBL_Test.txt:34-70): a 6x4 checker floor of matte patches,
    a mirror sphere of radius 0.4 standing on it, no area light: the only emitter is the environment sphere
    (synthetic sky, scale 4 like `brightness 4`).  `area_light` adds a small emitting quad above the sphere so that
    Scene::selectLight chooses between the triangle light list and the environment (SurfaceObject.cpp:432-450)."""
    b = SceneBuilder()
    rng = np.random.default_rng(24)
    colours = rng.uniform(0.15, 0.85, size=(24, 3))
    for j in range(4):
        for i in range(6):
            m = b.matte(b.spectrum_srgb_nonlinear(*colours[j * 6 + i]))
            x0, z0 = -1.5 + 0.5 * i, -1.0 + 0.5 * j
            b.add_quad([(x0, 0, z0 + 0.5), (x0 + 0.5, 0, z0 + 0.5), (x0 + 0.5, 0, z0), (x0, 0, z0)], (0, 1, 0), (1, 0, 0), m)
    mirror = b.metal(b.spectrum_grey(1.0), b.spectrum_ior("Aluminium", 0, ALUMINIUM_ETA_RGB), b.spectrum_ior("Aluminium", 1, ALUMINIUM_K_RGB))
    b.add_uv_sphere(segments, rings, mirror, _translate(0.0, 0.4, 0.0) @ _scale(0.4))
    if area_light:
        lm = b.matte(b.spectrum_grey(0.8), emittance=b.spectrum_d65(0.25, D65_RGB))
        b.add_quad([(-0.3, 1.6, -0.3), (0.3, 1.6, -0.3), (0.3, 1.6, 0.3), (-0.3, 1.6, 0.3)], (0, -1, 0), (1, 0, 0), lm)
    cam = make_camera(_translate(0.0, 0.9, 3.2) @ _rotate(3.1415926536, (0, 1, 0)) @ _rotate(0.2, (1, 0, 0)), aspect, 0.6, 0.02, 1.0, 3.2)
    sky = synthetic_sky(*env_size)
    uvs = sky_to_uvs(sky)
    return b.build(cam, env=(sky, 4.0, ibl_importance(sky), uvs, ibl_importance_uvs(uvs)), name="ibl_test")


def _hash_lattice(ix, iz, seed):
    """uint32 lattice hash -> [0, 1) (murmur3 finaliser over the two coordinates and the seed)."""
    h = (ix.astype(np.uint32) * np.uint32(0x85EBCA77)) ^ (iz.astype(np.uint32) * np.uint32(0xC2B2AE3D)) ^ np.uint32(seed)
    h ^= h >> np.uint32(16); h *= np.uint32(0x85EBCA6B); h ^= h >> np.uint32(13); h *= np.uint32(0xC2B2AE35); h ^= h >> np.uint32(16)
    return (h >> np.uint32(8)).astype(np.float64) * (1.0 / 16777216.0)


def hash_noise_heightfield(n, seed=20240611, octaves=6, base_cells=4):
    """(n+1) x (n+1) heights in [0, ~1): `octaves` of smooth value noise, octave k on a base_cells * 2^k lattice with
    amplitude 2^-k and its own seed (seed + k)."""
    t = np.arange(n + 1, dtype=np.float64) / n
    height = np.zeros((n + 1, n + 1))
    for k in range(octaves):
        cells = base_cells << k
        g = t * cells
        i0 = np.minimum(np.floor(g).astype(np.int64), cells - 1)
        f = g - i0
        w = f * f * (3 - 2 * f)
        ix0, iz0 = np.meshgrid(i0, i0, indexing="xy")
        wx, wz = np.meshgrid(w, w, indexing="xy")
        v00, v10 = _hash_lattice(ix0, iz0, seed + k), _hash_lattice(ix0 + 1, iz0, seed + k)
        v01, v11 = _hash_lattice(ix0, iz0 + 1, seed + k), _hash_lattice(ix0 + 1, iz0 + 1, seed + k)
        height += ((v00 * (1 - wx) + v10 * wx) * (1 - wz) + (v01 * (1 - wx) + v11 * wx) * wz) * 0.5 ** k
    return height / (2.0 - 0.5 ** (octaves - 1))


def instanced_grid(tiles_x=25, tiles_z=50, cells=64, aspect=16.0 / 9.0, seed=20240611, extent=4.0, relief=0.9):
    """Config 5 of BASELINE.json as it is WRITTEN — an instanced mesh: ONE patch of 2 cells^2 triangles (64 -> 8 192; a seeded
    hash-noise heightfield over [-1, 1]^2 in the patch's local space) placed tiles_x x tiles_z times (25 x 50 = 1 250 placements,
    10.24 M triangles seen by the rays) side by side over the extent of displaced_grid, each through its own StaticTransform:
    non-uniform scale (the tile is twice as long in x as in z, and every tile has its own relief), a half turn about y for every
    other tile, a translation.  Same material, light and thin-lens camera as displaced_grid."""
    n = cells
    h = hash_noise_heightfield(n, seed)
    t = np.linspace(-1.0, 1.0, n + 1)
    x, z = np.meshgrid(t, t, indexing="xy")
    step = 2.0 / n
    dhdx = np.gradient(h, step, axis=1)
    dhdz = np.gradient(h, step, axis=0)
    nrm = np.stack([-dhdx, np.ones_like(h), -dhdz], axis=-1)
    nrm /= np.linalg.norm(nrm, axis=-1, keepdims=True)
    tan = np.stack([np.ones_like(h), dhdx, np.zeros_like(h)], axis=-1)
    tan /= np.linalg.norm(tan, axis=-1, keepdims=True)
    pos = np.stack([x, h, z], axis=-1).reshape(-1, 3)
    uv = np.stack([(x + 1.0) / 2.0, (z + 1.0) / 2.0], axis=-1).reshape(-1, 2)
    idx = (np.arange(n)[:, None] * (n + 1) + np.arange(n)[None, :]).reshape(-1)
    a, b_, c, d = idx, idx + 1, idx + n + 2, idx + n + 1
    faces = np.concatenate([np.stack([a, d, c], axis=1), np.stack([a, c, b_], axis=1)], axis=1).reshape(-1, 3)
    b = SceneBuilder()
    lm = b.matte(b.spectrum_grey(0.8), emittance=b.spectrum_d65(1.0, D65_RGB))
    b.add_quad([(-1.0, 3.0, -1.0), (1.0, 3.0, -1.0), (1.0, 3.0, 1.0), (-1.0, 3.0, 1.0)], (0, -1, 0), (1, 0, 0), lm)
    ground = b.matte(b.spectrum_srgb_nonlinear(0.72, 0.64, 0.5))
    first = b.num_triangles()
    b.add_mesh(pos, nrm.reshape(-1, 3), tan.reshape(-1, 3), uv, faces, ground)
    count = b.num_triangles() - first
    r = np.random.default_rng(seed)
    sx, sz = extent / tiles_x, extent / tiles_z
    for iz in range(tiles_z):
        for ix in range(tiles_x):
            turn = np.eye(4) if (ix + iz) % 2 == 0 else np.diag([-1.0, 1.0, -1.0, 1.0])        # half turn about y
            scale = np.diag([sx, relief * float(r.uniform(0.05, 0.25)), sz, 1.0])
            place = _translate(-extent + (2 * ix + 1) * sx, 0.0, -extent + (2 * iz + 1) * sz)
            b.add_instance(first, count, place @ turn @ scale)
    cam = make_camera(_translate(0.0, 2.4, 5.2) @ _rotate(3.1415926536, (0, 1, 0)) @ _rotate(0.42, (1, 0, 0)), aspect, 0.6, 0.025, 1.0, 5.4)
    return b.build(cam, name="instanced_grid_%dx%d" % (tiles_x * tiles_z, count))


def displaced_grid(n=2236, aspect=16.0 / 9.0, seed=20240611, extent=4.0, relief=0.9):
    """Config 5 of BASELINE.json: ONE displaced grid of 2 n^2 triangles (n = 2236 -> 9 999 392), matte, one area light,
    thin-lens camera r = 0.025.  Vertices come from a seeded hash-noise heightfield (seed above); normals are the
    central-difference normals of the heightfield, tangents (1, dh/dx, 0) normalised (orthogonal to the normal)."""
    h = hash_noise_heightfield(n, seed) * relief
    t = np.linspace(-extent, extent, n + 1)
    x, z = np.meshgrid(t, t, indexing="xy")
    step = 2.0 * extent / n
    dhdx = np.gradient(h, step, axis=1)
    dhdz = np.gradient(h, step, axis=0)
    nrm = np.stack([-dhdx, np.ones_like(h), -dhdz], axis=-1)
    nrm /= np.linalg.norm(nrm, axis=-1, keepdims=True)
    tan = np.stack([np.ones_like(h), dhdx, np.zeros_like(h)], axis=-1)
    tan /= np.linalg.norm(tan, axis=-1, keepdims=True)
    pos = np.stack([x, h, z], axis=-1).reshape(-1, 3)
    uv = np.stack([(x + extent) / (2 * extent), (z + extent) / (2 * extent)], axis=-1).reshape(-1, 2)
    idx = (np.arange(n)[:, None] * (n + 1) + np.arange(n)[None, :]).reshape(-1)
    a, b_, c, d = idx, idx + 1, idx + n + 2, idx + n + 1
    # counter-clockwise seen from +y: (a, d, c), (a, c, b)
    faces = np.concatenate([np.stack([a, d, c], axis=1), np.stack([a, c, b_], axis=1)], axis=1).reshape(-1, 3)
    b = SceneBuilder()
    ground = b.matte(b.spectrum_srgb_nonlinear(0.72, 0.64, 0.5))
    b.add_mesh(pos, nrm.reshape(-1, 3), tan.reshape(-1, 3), uv, faces, ground)
    lm = b.matte(b.spectrum_grey(0.8), emittance=b.spectrum_d65(1.0, D65_RGB))
    b.add_quad([(-1.0, 3.0, -1.0), (1.0, 3.0, -1.0), (1.0, 3.0, 1.0), (-1.0, 3.0, 1.0)], (0, -1, 0), (1, 0, 0), lm)
    cam = make_camera(_translate(0.0, 2.4, 5.2) @ _rotate(3.1415926536, (0, 1, 0)) @ _rotate(0.42, (1, 0, 0)), aspect, 0.6, 0.025, 1.0, 5.4)
    return b.build(cam, name="displaced_grid_%d" % (2 * n * n))
