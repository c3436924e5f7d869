"""ctypes loader for the oracle libraries (test infrastructure; see oracle/slr_oracle.h).

`load("oracle")` -> the restatement (oracle/libslr_oracle.so, prefix slr_oracle_).
`load("ref_rgb")` / `load("ref_spectral")` -> the compiled reference (oracle/_ref/, prefix
slr_ref_), or None when it has not been built (no /root/reference).
"""
import ctypes as C
import os
import subprocess

import numpy as np

from slr_amd import abi

HERE = os.path.dirname(os.path.abspath(__file__))


class OracleCounters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("samples", "extension_rays", "shadow_rays", "loop_iterations",
                                          "rng_draws", "nodes_visited", "tris_tested")]


ray_dtype = np.dtype([("org", "<f4", 3), ("dir", "<f4", 3), ("dist_min", "<f4"), ("dist_max", "<f4")])
hit_dtype = np.dtype([("triangle", "<u4"), ("dist", "<f4"), ("b0", "<f4"), ("b1", "<f4")])


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", HERE])


def build_ref():
    if os.path.isdir("/root/reference/libSLR"):
        subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(HERE, "ref_build")])


class OracleLib:
    def __init__(self, path, prefix, mode):
        self.lib = C.CDLL(path)
        self.prefix = prefix
        self.mode = mode
        f = self._f
        f("create").restype = C.c_void_p
        f("create").argtypes = [C.POINTER(abi.SceneDesc), C.c_int]
        f("destroy").argtypes = [C.c_void_p]
        f("destroy").restype = None
        f("render").argtypes = [C.c_void_p, C.POINTER(abi.RenderSettings), abi.Shard, C.c_uint32, C.c_uint32, C.c_int,
                                C.c_void_p, C.c_void_p, C.POINTER(OracleCounters)]
        f("sample").argtypes = [C.c_void_p, C.POINTER(abi.RenderSettings), C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
        f("trace").argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        f("rng").argtypes = [C.c_int32, C.c_uint32, C.c_void_p, C.c_void_p]
        f("rng").restype = None
        f("components").argtypes = [C.c_void_p]
        f("bsdf_kat").argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_float, C.c_float, C.c_void_p]
        f("light_kat").argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_float, C.c_float, C.c_void_p]
        f("render_serial").argtypes = [C.c_void_p, C.POINTER(abi.RenderSettings), C.c_uint32, C.c_void_p,
                                       C.POINTER(OracleCounters)]

        if prefix == "slr_ref_":
            f("render_native").argtypes = [C.c_void_p, C.POINTER(abi.RenderSettings), C.c_uint32, C.c_int, C.c_void_p,
                                           C.POINTER(C.c_double)]

    def _f(self, name):
        return getattr(self.lib, self.prefix + name)

    def scene(self, scene):
        return OracleScene(self, scene)

    def rng(self, seed, n):
        u = np.zeros(n, dtype=np.uint32)
        f = np.zeros(n, dtype=np.float32)
        self._f("rng")(seed, n, u.ctypes.data, f.ctypes.data)
        return u, f


class OracleScene:
    def __init__(self, lib, scene):
        self.lib = lib
        self.scene = scene  # keep the numpy arrays alive
        desc = scene.desc(lib.mode)
        self.handle = lib._f("create")(C.byref(desc), lib.mode)
        if not self.handle:
            raise RuntimeError("oracle: scene rejected (%s)" % lib.prefix)
        self.components = lib._f("components")(self.handle)

    def close(self):
        if self.handle:
            self.lib._f("destroy")(self.handle)
            self.handle = None

    def __del__(self):
        self.close()

    def render(self, settings, spp, spp_begin=0, shard=(0, 1), threads=0, state=None):
        """Returns (fb_sum[H,W,C], counters); `state` = (sum, comp) to continue a render."""
        h, w = settings.image_height, settings.image_width
        if state is None:
            state = (np.zeros((h, w, self.components), np.float32), np.zeros((h, w, self.components), np.float32))
        ctr = OracleCounters()
        rc = self.lib._f("render")(self.handle, C.byref(settings), abi.Shard(*shard), spp_begin, spp, threads,
                                   state[0].ctypes.data, state[1].ctypes.data, C.byref(ctr))
        if rc != 0:
            raise RuntimeError("oracle render failed: %d" % rc)
        return state[0], ctr

    def sample(self, settings, px, py, sample):
        out = np.zeros(self.components + 2, np.float32)
        rc = self.lib._f("sample")(self.handle, C.byref(settings), px, py, sample, out.ctypes.data)
        if rc != 0:
            raise RuntimeError("oracle sample failed: %d" % rc)
        return out

    def trace(self, rays):
        rays = np.ascontiguousarray(rays, dtype=ray_dtype)
        hits = np.zeros(len(rays), dtype=hit_dtype)
        rc = self.lib._f("trace")(self.handle, rays.ctypes.data, len(rays), hits.ctypes.data)
        if rc != 0:
            raise RuntimeError("oracle trace failed: %d" % rc)
        return hits

    def bsdf_kat(self, material, queries, wl_offset=0.5, u_lambda=0.5):
        """BSDF sample / evaluate / evaluatePDF of one scene material; queries [n][12], returns [n][6 + 2C]
        (layout in slr_oracle.h)."""
        q = np.ascontiguousarray(queries, dtype=np.float32).reshape(-1, 12)
        out = np.zeros((len(q), 6 + 2 * self.components), np.float32)
        rc = self.lib._f("bsdf_kat")(self.handle, material, len(q), q.ctypes.data, wl_offset, u_lambda, out.ctypes.data)
        if rc != 0:
            raise RuntimeError("oracle bsdf_kat failed: %d" % rc)
        return out

    def light_kat(self, u, wl_offset=0.5, u_lambda=0.5):
        """Scene::selectLight + Light::sample; u [n][3] = light selection sample, position samples -> [n][16 + C]
        (layout in slr_oracle.h)."""
        q = np.ascontiguousarray(u, dtype=np.float32).reshape(-1, 3)
        out = np.zeros((len(q), 16 + self.components), np.float32)
        rc = self.lib._f("light_kat")(self.handle, len(q), q.ctypes.data, wl_offset, u_lambda, out.ctypes.data)
        if rc != 0:
            raise RuntimeError("oracle light_kat failed: %d" % rc)
        return out

    def render_serial(self, settings, spp):
        h, w = settings.image_height, settings.image_width
        fb = np.zeros((h, w, self.components), np.float32)
        ctr = OracleCounters()
        rc = self.lib._f("render_serial")(self.handle, C.byref(settings), spp, fb.ctypes.data, C.byref(ctr))
        if rc != 0:
            raise RuntimeError("oracle render_serial failed: %d" % rc)
        return fb, ctr


def render_native(ref_scene, settings, spp, threads=0):
    """Times the reference's own unmodified multi-threaded render(); returns (fb, seconds)."""
    h, w = settings.image_height, settings.image_width
    fb = np.zeros((h, w, ref_scene.components), np.float32)
    sec = C.c_double(0.0)
    rc = ref_scene.lib._f("render_native")(ref_scene.handle, C.byref(settings), spp, threads, fb.ctypes.data, C.byref(sec))
    if rc != 0:
        raise RuntimeError("reference render failed: %d" % rc)
    return fb, sec.value


def ref_save_image(ref_lib, fb, sensitivity, scale, path):
    fb = np.ascontiguousarray(fb, np.float32)
    f = ref_lib.lib.slr_ref_save_image
    f.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_float, C.c_float, C.c_char_p]
    rc = f(fb.ctypes.data, fb.shape[1], fb.shape[0], sensitivity, scale, path.encode())
    if rc != 0:
        raise RuntimeError("reference saveImage failed: %d" % rc)


def load(which="oracle", mode=abi.MODE_RGB):
    if which == "oracle":
        path = os.path.join(HERE, "libslr_oracle.so")
        if not os.path.exists(path):
            build_oracle()
        return OracleLib(path, "slr_oracle_", mode)
    name = {"ref_rgb": "libslr_ref_rgb.so", "ref_spectral": "libslr_ref_spectral.so"}[which]
    path = os.path.join(HERE, "_ref", name)
    if not os.path.exists(path):
        return None
    return OracleLib(path, "slr_ref_", abi.MODE_RGB if which == "ref_rgb" else abi.MODE_SPECTRAL)


def settings(width, height, seed=abi.DEFAULT_SEED, brightness=1.0):
    return abi.RenderSettings(width, height, 0.0, 0.0, brightness, seed)
