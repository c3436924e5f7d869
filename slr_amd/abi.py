"""ctypes mirror of include/slrhip.h (the C ABI of the path-tracing hot path).

Only plumbing lives here: struct layouts, library loading and a thin `Context`
wrapper.  The same scene structs are consumed by three implementations of one
interface: the HIP product library (slr_amd/csrc -> libslrhip.so), the CPU oracle
(oracle/ -> libslr_oracle.so, tests/bench only) and, when built in the container
that has /root/reference, the compiled reference itself (oracle/_ref).
"""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

MODE_RGB, MODE_SPECTRAL = 0, 1
MAT_MATTE, MAT_METAL, MAT_GLASS, MAT_MF_METAL, MAT_MF_GLASS, MAT_WARD, MAT_ASHIKHMIN, MAT_MULTI = 0, 1, 2, 3, 4, 5, 6, 7
MULTI_INVERSE_0, MULTI_INVERSE_1 = 1, 2
SPEC_RGB_ONLY, SPEC_UPSAMPLED, SPEC_REGULAR, SPEC_IRREGULAR = 0, 1, 2, 3

# numpy dtypes with the exact layout of the C structs (checked in tests/test_abi.py)
vertex_dtype = np.dtype([("position", "<f4", 3), ("normal", "<f4", 3), ("tangent", "<f4", 3),
                         ("texcoord", "<f4", 2)])
triangle_dtype = np.dtype([("v", "<u4", 3), ("material", "<u4")])
material_dtype = np.dtype([("type", "<u4"), ("spectrum", "<i4", 3), ("param", "<f4"), ("emittance", "<i4"), ("param2", "<f4"), ("reserved", "<u4")])
spectrum_dtype = np.dtype([("kind", "<u4"), ("rgb", "<f4", 3), ("u", "<f4"), ("v", "<f4"), ("scale", "<f4"),
                           ("lambda_min", "<f4"), ("lambda_max", "<f4"), ("num_samples", "<u4"),
                           ("data_offset", "<u4"), ("reserved", "<u4")])


texture_dtype = np.dtype([("kind", "<u4"), ("offset", "<f4", 2), ("scale", "<f4", 2), ("spectrum", "<i4", 2), ("value", "<f4", 2),
                          ("reserved", "<u4", 3)])
instance_dtype = np.dtype([("first_triangle", "<u4"), ("num_triangles", "<u4"), ("local_to_world", "<f4", 16), ("world_to_local", "<f4", 16)])
TEX_CHECKER_SPECTRUM, TEX_CHECKER_FLOAT, TEX_CHECKER_NORMAL, TEX_IMAGE_SPECTRUM = 0, 1, 2, 3


def texture_ref(t):
    """SLRHIP_TEXTURE_REF: the value of a material's spectrum slot that names texture t."""
    return -2 - int(t)


class Camera(C.Structure):
    _fields_ = [("local_to_world", C.c_float * 16), ("world_to_local", C.c_float * 16),
                ("aspect", C.c_float), ("fov_y", C.c_float), ("lens_radius", C.c_float),
                ("img_plane_distance", C.c_float), ("obj_plane_distance", C.c_float),
                ("sensitivity", C.c_float)]


class EnvMap(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("texels", C.c_void_p), ("scale", C.c_float),
                ("map_width", C.c_uint32), ("map_height", C.c_uint32), ("importance", C.c_void_p)]


class UpsamplingTables(C.Structure):
    _fields_ = [("grid_width", C.c_uint32), ("grid_height", C.c_uint32), ("cells", C.c_void_p), ("num_points", C.c_uint32),
                ("point_uv", C.c_void_p), ("point_spectrum", C.c_void_p)]


class SceneDesc(C.Structure):
    _fields_ = [("vertices", C.c_void_p), ("num_vertices", C.c_uint32),
                ("triangles", C.c_void_p), ("num_triangles", C.c_uint32),
                ("materials", C.c_void_p), ("num_materials", C.c_uint32),
                ("spectra", C.c_void_p), ("num_spectra", C.c_uint32),
                ("spectrum_data", C.c_void_p), ("num_spectrum_data", C.c_uint32),
                ("camera", Camera),
                ("env", C.POINTER(EnvMap)),
                ("upsampling", C.POINTER(UpsamplingTables)),
                ("textures", C.c_void_p), ("num_textures", C.c_uint32),
                ("texture_texels", C.c_void_p), ("num_texture_texels", C.c_uint32),
                ("instances", C.c_void_p), ("num_instances", C.c_uint32)]


class RenderSettings(C.Structure):
    _fields_ = [("image_width", C.c_int32), ("image_height", C.c_int32), ("time_start", C.c_float),
                ("time_end", C.c_float), ("brightness", C.c_float), ("rng_seed", C.c_int32)]


class Shard(C.Structure):
    _fields_ = [("shard_index", C.c_uint32), ("shard_count", C.c_uint32)]


class Config(C.Structure):
    _fields_ = [("device", C.c_int32), ("mode", C.c_int32), ("stripes", C.c_uint32), ("flags", C.c_uint32)]


class Counters(C.Structure):
    _fields_ = [("samples", C.c_uint64), ("extension_rays", C.c_uint64), ("shadow_rays", C.c_uint64),
                ("iterations", C.c_uint64), ("bvh_nodes", C.c_uint64), ("bvh_depth", C.c_uint64),
                ("build_seconds", C.c_double), ("bvh_leaf_references", C.c_uint64)]


class Profile(C.Structure):
    _fields_ = [("launches", C.c_uint64 * 3), ("milliseconds", C.c_double * 3), ("rays", C.c_uint64 * 2),
                ("nodes", C.c_uint64 * 2), ("triangles", C.c_uint64 * 2), ("slot_visits", C.c_uint64)]


FLAG_TIME_KERNELS, FLAG_COUNT_TRAVERSAL, FLAG_BVH_DEVICE_BUILD, FLAG_TEST_DEVICE_ERROR, FLAG_BVH_SPATIAL_SPLITS, FLAG_TAIL_KERNEL = 1, 2, 4, 16, 64, 128
MAX_STRIPES = 64
KERNEL_NAMES = ("trace", "shade", "tail")

DEFAULT_SEED = 1509761209  # libSLRSceneGraph/API.cpp:1080


class Scene:
    """Flat scene (numpy arrays) + the ctypes view handed across the ABI."""

    def __init__(self, vertices, triangles, materials, spectra, spectrum_data, camera, env=None, name="scene", textures=None,
                 texture_texels=None, texture_texels_uvs=None, instances=None):
        self.vertices = np.ascontiguousarray(vertices, dtype=vertex_dtype)
        self.triangles = np.ascontiguousarray(triangles, dtype=triangle_dtype)
        mats = np.asarray(materials)
        if mats.dtype != material_dtype:           # e.g. fixtures written before param2 existed: copy the common fields
            conv = np.zeros(len(mats), dtype=material_dtype)
            for name in mats.dtype.names:
                conv[name] = mats[name]
            mats = conv
        self.materials = np.ascontiguousarray(mats, dtype=material_dtype)
        self.spectra = np.ascontiguousarray(spectra, dtype=spectrum_dtype)
        self.spectrum_data = np.ascontiguousarray(spectrum_data, dtype=np.float32)
        self.camera = camera
        self.textures = np.ascontiguousarray(textures if textures is not None else np.zeros(0, texture_dtype), dtype=texture_dtype)
        self.instances = np.ascontiguousarray(instances if instances is not None else np.zeros(0, instance_dtype), dtype=instance_dtype)
        # texels of the image textures (TEX_IMAGE_SPECTRUM): [n][3] as the RGB build stores them, and as (u, v, s) for the spectral build
        self.texture_texels = np.ascontiguousarray(texture_texels if texture_texels is not None else np.zeros((0, 3)), dtype=np.float32).reshape(-1, 3)
        self.texture_texels_uvs = np.ascontiguousarray(texture_texels_uvs if texture_texels_uvs is not None else np.zeros((0, 3)), dtype=np.float32).reshape(-1, 3)
        self.env_texels = None
        self.env = None
        self.env_uvs = None
        self._tables = None
        if env is not None:
            # (texels rgb, scale, importance) or, for both modes, (texels rgb, scale, importance rgb, texels uvs, importance uvs)
            texels, scale, importance = env[:3]
            self.env_texels = np.ascontiguousarray(texels, dtype=np.float32)
            self.env_importance = np.ascontiguousarray(importance, dtype=np.float32)
            self.env_scale = float(scale)
            self.env = EnvMap(self.env_texels.shape[1], self.env_texels.shape[0], self.env_texels.ctypes.data, float(scale),
                              self.env_importance.shape[1], self.env_importance.shape[0], self.env_importance.ctypes.data)
            if len(env) == 5:
                self.env_texels_uvs = np.ascontiguousarray(env[3], dtype=np.float32)
                self.env_importance_uvs = np.ascontiguousarray(env[4], dtype=np.float32)
                self.env_uvs = EnvMap(self.env_texels_uvs.shape[1], self.env_texels_uvs.shape[0], self.env_texels_uvs.ctypes.data, float(scale),
                                      self.env_importance_uvs.shape[1], self.env_importance_uvs.shape[0], self.env_importance_uvs.ctypes.data)
        self.name = name
        self._validate()

    def _validate(self):
        nv, nm, ns = len(self.vertices), len(self.materials), len(self.spectra)
        if len(self.triangles) == 0:
            raise ValueError("scene has no triangles")
        if self.env is not None and (self.env_texels.ndim != 3 or self.env_texels.shape[2] != 3):
            raise ValueError("environment texels must be [height][width][3]")
        if self.triangles["v"].max() >= nv:
            raise ValueError("triangle references a vertex out of range")
        if self.triangles["material"].max() >= nm:
            raise ValueError("triangle references a material out of range")
        single = self.materials["type"] != MAT_MULTI      # a MULTI record's spectrum[] holds component material indices (checked by the library)
        if (single.any() and self.materials["spectrum"][single].max() >= ns) or self.materials["emittance"].max() >= ns:
            raise ValueError("material references a spectrum out of range")
        nt = len(self.textures)
        if single.any() and self.materials["spectrum"][single].min() < -1 - nt:
            raise ValueError("material references a texture out of range")
        if nt and (self.textures["spectrum"][self.textures["kind"] == TEX_CHECKER_SPECTRUM].max(initial=-1) >= ns):
            raise ValueError("texture references a spectrum out of range")

    def upsampling_tables(self):
        if self._tables is None:
            from . import spectra
            t = spectra.tables()
            cells = np.zeros((len(t["grid_inside"]), 8), np.uint8)
            cells[:, 0], cells[:, 1], cells[:, 2:8] = t["grid_inside"], t["grid_num_points"], t["grid_idx"]
            self._table_arrays = (np.ascontiguousarray(cells), np.ascontiguousarray(t["point_uv"], np.float32),
                                  np.ascontiguousarray(t["point_spectrum"], np.float32))
            self._tables = UpsamplingTables(spectra.GRID_WIDTH, spectra.GRID_HEIGHT, self._table_arrays[0].ctypes.data, len(self._table_arrays[1]),
                                            self._table_arrays[1].ctypes.data, self._table_arrays[2].ctypes.data)
        return self._tables

    def desc(self, mode=MODE_RGB):
        d = SceneDesc()
        d.vertices, d.num_vertices = self.vertices.ctypes.data, len(self.vertices)
        d.triangles, d.num_triangles = self.triangles.ctypes.data, len(self.triangles)
        d.materials, d.num_materials = self.materials.ctypes.data, len(self.materials)
        d.spectra, d.num_spectra = self.spectra.ctypes.data, len(self.spectra)
        d.spectrum_data, d.num_spectrum_data = self.spectrum_data.ctypes.data, len(self.spectrum_data)
        d.camera = self.camera
        env = self.env_uvs if (mode == MODE_SPECTRAL and self.env_uvs is not None) else self.env
        d.env = C.pointer(env) if env is not None else None
        texels = self.texture_texels_uvs if mode == MODE_SPECTRAL else self.texture_texels
        d.upsampling = C.pointer(self.upsampling_tables()) if ((env is not None or len(texels)) and mode == MODE_SPECTRAL) else None
        d.textures, d.num_textures = (self.textures.ctypes.data if len(self.textures) else None), len(self.textures)
        d.texture_texels, d.num_texture_texels = (texels.ctypes.data if len(texels) else None), len(texels)
        d.instances, d.num_instances = (self.instances.ctypes.data if len(self.instances) else None), len(self.instances)
        return d


def sample_seed(rng_seed, px, py, sample):
    """Python restatement of slrhip_sample_seed (the per-(pixel,sample) seeding contract)."""
    def fmix(h):
        h &= 0xFFFFFFFF
        h ^= h >> 16
        h = (h * 0x85EBCA6B) & 0xFFFFFFFF
        h ^= h >> 13
        h = (h * 0xC2B2AE35) & 0xFFFFFFFF
        h ^= h >> 16
        return h
    h = rng_seed & 0xFFFFFFFF
    h = fmix(h ^ ((sample * 0x9E3779B1) & 0xFFFFFFFF))
    h = fmix(h ^ ((py * 0x85EBCA77 + 0x165667B1) & 0xFFFFFFFF))
    h = fmix(h ^ ((px * 0xC2B2AE3D + 0x27D4EB2F) & 0xFFFFFFFF))
    return h - (1 << 32) if h & 0x80000000 else h
