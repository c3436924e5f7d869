"""CPU, only where oracle/_ref is built (the container with /root/reference): the restatement
against the compiled reference itself on fresh inputs (not just the committed fixtures)."""
import numpy as np
import pytest

from helpers import assert_bit_equal
from oracle import binding as ob
from slr_amd import scenes


@pytest.mark.parametrize("right", ["glass", "matte"])
def test_full_cornell_frame_bit_exact(oracle_rgb, ref_rgb, right):
    sc = scenes.cornell_box_spheres(4.0 / 3.0, 48, 24, right)      # 4 428 triangles (config 1/2 geometry)
    so, sr = oracle_rgb.scene(sc), ref_rgb.scene(sc)
    st = ob.settings(64, 48, seed=20240229)
    fo, co = so.render(st, 8)
    fr, _ = sr.render(st, 8)
    assert_bit_equal(fo, fr, "cornell " + right)
    assert co.extension_rays / co.samples > 2.0


def test_serial_render_bit_exact(oracle_rgb, ref_rgb):
    sc = scenes.cornell_box_spheres(1.0, 24, 12, "glass")
    so, sr = oracle_rgb.scene(sc), ref_rgb.scene(sc)
    st = ob.settings(40, 40, seed=7)       # 40 is not a multiple of the 8x8 tile: exercises the clamp in ImageSensor::add
    fo, _ = so.render_serial(st, 3)
    fr, _ = sr.render_serial(st, 3)
    assert_bit_equal(fo, fr, "serial")


def test_random_rays_bit_exact(oracle_rgb, ref_rgb):
    sc = scenes.cornell_box_spheres(1.0, 32, 16, "glass")
    so, sr = oracle_rgb.scene(sc), ref_rgb.scene(sc)
    rng = np.random.default_rng(99)
    n = 4096
    rays = np.zeros(n, dtype=ob.ray_dtype)
    rays["org"] = rng.uniform([-1.4, 0.1, -2.4], [1.4, 2.4, 2.4], size=(n, 3))
    d = rng.normal(size=(n, 3))
    rays["dir"] = d / np.linalg.norm(d, axis=1, keepdims=True)
    rays["dist_min"], rays["dist_max"] = 1e-4, np.inf
    ho, hr = so.trace(rays), sr.trace(rays)
    same = ho["triangle"] == hr["triangle"]
    # equal-distance ties (shared edges) are resolved by traversal order in the reference
    tie = ~same & (ho["dist"] == hr["dist"])
    assert (same | tie).all()
    assert tie.sum() <= n // 500
    assert_bit_equal(ho["dist"], hr["dist"], "dist")


@pytest.mark.parametrize("kind", ["oren_nayar", "ggx_metal", "ggx_glass"])
def test_lobes_bit_exact(oracle_rgb, ref_rgb, kind):
    sc = scenes.cornell_lobes(kind, segments=24, rings=12)
    so, sr = oracle_rgb.scene(sc), ref_rgb.scene(sc)
    st = ob.settings(56, 56, seed=31337)
    fo, _ = so.render(st, 8)
    fr, _ = sr.render(st, 8)
    assert_bit_equal(fo, fr, kind)


def test_boxes_scene_bit_exact(oracle_rgb, ref_rgb):
    sc = scenes.cornell_box_boxes()
    so, sr = oracle_rgb.scene(sc), ref_rgb.scene(sc)
    st = ob.settings(64, 64, seed=5)
    fo, _ = so.render(st, 8)
    fr, _ = sr.render(st, 8)
    assert_bit_equal(fo, fr, "cornell_box_boxes")
