"""ctypes binding of libslrhip.so (include/slrhip.h).  There is no fallback: if the HIP
library is missing or no GPU is present, creating a Context raises."""
import ctypes as C
import os
import subprocess

import numpy as np

from . import abi

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB_PATH = os.path.join(CSRC, "libslrhip.so")

EXPORTS = ["slrhip_create", "slrhip_destroy", "slrhip_upload_scene", "slrhip_render_begin", "slrhip_render",
           "slrhip_resolve_framebuffer", "slrhip_reduce_framebuffer", "slrhip_read_framebuffer", "slrhip_synchronize", "slrhip_get_counters",
           "slrhip_components", "slrhip_get_profile", "slrhip_trace_rays", "slrhip_bsdf_queries", "slrhip_debug_work_distribution", "slrhip_sample_seed", "slrhip_upsample", "slrhip_resolve_upsampled", "slrhip_spectrum_to_rgb", "slrhip_tonemap_bgr8", "slrhip_save_bmp",
           "slrhip_last_error_string", "slrhip_version"]


class HipLibraryMissing(RuntimeError):
    pass


class SlrHipError(RuntimeError):
    pass


_lib = None


def build_library():
    subprocess.check_call(["make", "-s", "-C", CSRC])


def load_library():
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("SLRHIP_LIBRARY", LIB_PATH)      # development: an alternate build of the same ABI (tools/build_variant.sh)
    if not os.path.exists(path):
        raise HipLibraryMissing("%s not built: run `make -C slr_amd/csrc` (or __graft_entry__.build())" % path)
    lib = C.CDLL(path)
    lib.slrhip_create.argtypes = [C.POINTER(abi.Config), C.POINTER(C.c_void_p)]
    lib.slrhip_destroy.argtypes = [C.c_void_p]
    lib.slrhip_upload_scene.argtypes = [C.c_void_p, C.POINTER(abi.SceneDesc)]
    lib.slrhip_render_begin.argtypes = [C.c_void_p, C.POINTER(abi.RenderSettings), abi.Shard]
    lib.slrhip_render.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
    lib.slrhip_resolve_framebuffer.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.slrhip_reduce_framebuffer.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.slrhip_read_framebuffer.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    lib.slrhip_synchronize.argtypes = [C.c_void_p]
    lib.slrhip_get_counters.argtypes = [C.c_void_p, C.POINTER(abi.Counters)]
    lib.slrhip_components.argtypes = [C.c_void_p]
    lib.slrhip_get_profile.argtypes = [C.c_void_p, C.POINTER(abi.Profile)]
    lib.slrhip_trace_rays.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    lib.slrhip_bsdf_queries.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_float, C.c_float, C.c_void_p]
    lib.slrhip_sample_seed.argtypes = [C.c_int32, C.c_uint32, C.c_uint32, C.c_uint32]
    lib.slrhip_sample_seed.restype = C.c_int32
    lib.slrhip_upsample.argtypes = [C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_void_p]
    lib.slrhip_resolve_upsampled.argtypes = [C.POINTER(abi.UpsamplingTables), C.c_float, C.c_float, C.POINTER(C.c_uint32), C.c_void_p]
    lib.slrhip_spectrum_to_rgb.argtypes = [C.c_int32, C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_uint32, C.c_void_p]
    lib.slrhip_tonemap_bgr8.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_void_p, C.c_size_t]
    lib.slrhip_save_bmp.argtypes = [C.c_char_p, C.c_void_p, C.c_int32, C.c_int32]
    lib.slrhip_last_error_string.restype = C.c_char_p
    _lib = lib
    return lib


def _check(lib, rc, what):
    if rc != 0:
        raise SlrHipError("%s failed (%d): %s" % (what, rc, lib.slrhip_last_error_string().decode()))


class Context:
    """One rendering context on one GPU (slrhip_ctx)."""

    def __init__(self, device=0, mode=abi.MODE_RGB, stripes=0, flags=0):
        self.lib = load_library()
        self.handle = C.c_void_p()
        cfg = abi.Config(device, mode, stripes, flags)
        _check(self.lib, self.lib.slrhip_create(C.byref(cfg), C.byref(self.handle)), "slrhip_create")
        self.components = self.lib.slrhip_components(self.handle)
        self.settings = None
        self._scene = None

    def close(self):
        if getattr(self, "handle", None):
            self.lib.slrhip_destroy(self.handle)
            self.handle = None

    def __del__(self):
        self.close()

    def upload_scene(self, scene):
        self._scene = scene
        desc = scene.desc(self.mode)
        _check(self.lib, self.lib.slrhip_upload_scene(self.handle, C.byref(desc)), "slrhip_upload_scene")

    def render_begin(self, settings, shard=(0, 1)):
        self.settings = settings
        _check(self.lib, self.lib.slrhip_render_begin(self.handle, C.byref(settings), abi.Shard(*shard)), "slrhip_render_begin")

    def render(self, spp_begin, spp_count, stream=None):
        _check(self.lib, self.lib.slrhip_render(self.handle, spp_begin, spp_count, stream), "slrhip_render")

    def resolve_into(self, device_ptr, num_floats, stream=None):
        _check(self.lib, self.lib.slrhip_resolve_framebuffer(self.handle, device_ptr, num_floats, stream),
               "slrhip_resolve_framebuffer")

    def read_framebuffer(self):
        h, w = self.settings.image_height, self.settings.image_width
        fb = np.zeros((h, w, self.components), np.float32)
        _check(self.lib, self.lib.slrhip_read_framebuffer(self.handle, fb.ctypes.data, fb.size), "slrhip_read_framebuffer")
        return fb

    def synchronize(self):
        _check(self.lib, self.lib.slrhip_synchronize(self.handle), "slrhip_synchronize")

    def counters(self):
        c = abi.Counters()
        _check(self.lib, self.lib.slrhip_get_counters(self.handle, C.byref(c)), "slrhip_get_counters")
        return c

    def profile(self):
        p = abi.Profile()
        _check(self.lib, self.lib.slrhip_get_profile(self.handle, C.byref(p)), "slrhip_get_profile")
        return p

    def trace_rays(self, org, direction, dist_min, dist_max):
        n = len(org)
        rays = np.zeros((n, 8), np.float32)
        rays[:, 0:3], rays[:, 3:6], rays[:, 6], rays[:, 7] = org, direction, dist_min, dist_max
        hits = np.zeros((n, 4), np.float32)
        _check(self.lib, self.lib.slrhip_trace_rays(self.handle, rays.ctypes.data, n, hits.ctypes.data), "slrhip_trace_rays")
        return hits[:, 0].copy().view(np.uint32), hits[:, 1], hits[:, 2], hits[:, 3]

    def bsdf_queries(self, material, queries, wl_offset=0.5, u_lambda=0.5):
        """Function-level BSDF queries (slrhip_bsdf_queries): queries [n][12] -> [n][6 + 2C]."""
        q = np.ascontiguousarray(queries, dtype=np.float32).reshape(-1, 12)
        out = np.zeros((len(q), 6 + 2 * self.components), np.float32)
        _check(self.lib, self.lib.slrhip_bsdf_queries(self.handle, material, len(q), q.ctypes.data, wl_offset, u_lambda, out.ctypes.data),
               "slrhip_bsdf_queries")
        return out

    @property
    def mode(self):
        return abi.MODE_SPECTRAL if self.components == 16 else abi.MODE_RGB

    def render_image(self, scene, settings, spp, shard=(0, 1)):
        """Convenience: upload, render all passes, read back the linear float framebuffer."""
        self.upload_scene(scene)
        self.render_begin(settings, shard)
        self.render(0, spp)
        return self.read_framebuffer()
