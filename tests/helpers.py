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
    texels = g["texture_texels"] if "texture_texels" in g.files else None
    texels_uvs = g["texture_texels_uvs"] if "texture_texels_uvs" in g.files else None
    instances = g["instances"] if "instances" in g.files else None
    return abi.Scene(g["vertices"], g["triangles"], g["materials"], g["spectra"], g["spectrum_data"], cam, env, name, textures=textures,
                     texture_texels=texels, texture_texels_uvs=texels_uvs, instances=instances)


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def assert_bit_equal(a, b, what=""):
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    bad = bits(a) != bits(b)
    # +0 / -0 are the same radiance
    bad &= ~((a == 0) & (b == 0))
    assert not bad.any(), "%s: %d of %d floats differ, max abs diff %g" % (what, bad.sum(), bad.size, np.abs(a - b).max())


def libm_tolerance(got, want, what, within=0.999, rtol=2e-6, rmse_rel=2e-5, cap=None):
    """Frames whose paths call float libm (GGX / Ward / Ashikhmin lobes, the environment sphere): the device math library is not
    glibc, so a sample moves by an ulp and — rarely — a discrete decision flips.  Stated tolerance: at least `within` of the
    floats inside `rtol` relative (thresholds one notch under the figures measured on MI355X: profiles/r03_a_parity_stats.jsonl,
    written by this function on the GPU box), RMSE <= `rmse_rel` x the mean (measured <= 1.3e-6; north_star's bound is 1e-3; on frames clipped at `cap` x the mean where single
    texels are hundreds of times the mean), everything finite."""
    import json
    got, want = np.asarray(got), np.asarray(want)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    assert np.isfinite(got).all(), what
    close = float(np.isclose(got, want, rtol=rtol, atol=1e-9).mean())
    exact = float(((bits(got) == bits(want)) | ((got == 0) & (want == 0))).mean())
    a, b = got.astype(np.float64), want.astype(np.float64)
    if cap is not None:
        lim = cap * float(b.mean())
        a, b = np.minimum(a, lim), np.minimum(b, lim)
    rmse, mean = float(np.sqrt(np.mean((a - b) ** 2))), float(b.mean())
    out_dir = os.path.join(os.path.dirname(GOLDEN), os.pardir, "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "parity_stats.jsonl"), "a") as f:
            f.write(json.dumps({"what": what, "within_%g" % rtol: close, "bit_exact": exact, "rmse_over_mean": rmse / max(mean, 1e-30)}) + "\n")
    assert close >= within, (what, "fraction within %g" % rtol, close, "bit-exact", exact)
    assert rmse <= rmse_rel * mean, (what, rmse, mean)
