import os

import numpy as np

from slr_amd import abi

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


class _Prefixed:
    """View of an npz whose keys carry a prefix (several scenes in one fixture)."""
    def __init__(self, g, prefix):
        self.g, self.prefix = g, prefix
        self.files = [k[len(prefix):] for k in g.files if k.startswith(prefix)]

    def __getitem__(self, k):
        return self.g[self.prefix + k]


def scene_from_golden(g, name="golden", prefix=""):
    if prefix:
        g = _Prefixed(g, prefix)
    cam = abi.Camera()
    c = g["camera"]
    cam.local_to_world[:] = c[0:16].tolist()
    cam.world_to_local[:] = c[16:32].tolist()
    (cam.aspect, cam.fov_y, cam.lens_radius, cam.img_plane_distance, cam.obj_plane_distance,
     cam.sensitivity) = [float(v) for v in c[32:38]]
    env = None
    if "env_texels" in g.files:
        env = (g["env_texels"].astype(np.float32), float(g["env_scale"]), g["env_importance"])
        if "env_texels_uvs" in g.files:
            env = env + (g["env_texels_uvs"].astype(np.float32), g["env_importance_uvs"])
    textures = g["textures"] if "textures" in g.files else None
    return abi.Scene(g["vertices"], g["triangles"], g["materials"], g["spectra"], g["spectrum_data"], cam, env, name, textures=textures)


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def assert_bit_equal(a, b, what=""):
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    bad = bits(a) != bits(b)
    # +0 / -0 are the same radiance
    bad &= ~((a == 0) & (b == 0))
    assert not bad.any(), "%s: %d of %d floats differ, max abs diff %g" % (what, bad.sum(), bad.size, np.abs(a - b).max())
