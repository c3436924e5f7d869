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


@pytest.mark.parametrize("mode", ["rgb", "spectral"])
def test_instanced_scene_bit_exact(request, mode):
    """TransformedSurfaceObject over mesh aggregates (Core/SurfaceObject.cpp:303-392) against the oracle's restatement on a fresh
    scene: more placements than the fixture, frames and random rays."""
    sc = scenes.cornell_instanced(4.0 / 3.0, 14, 7, copies=11)
    so, sr = request.getfixturevalue("oracle_" + mode).scene(sc), request.getfixturevalue("ref_" + mode).scene(sc)
    st = ob.settings(64, 48, seed=77)
    fo, _ = so.render(st, 8)
    fr, _ = sr.render(st, 8)
    assert_bit_equal(fo, fr, "instanced " + mode)
    rng = np.random.default_rng(5)
    n = 8192
    rays = np.zeros(n, dtype=ob.ray_dtype)
    rays["org"] = rng.uniform([-1.4, 0.1, -2.4], [1.4, 2.4, 2.4], size=(n, 3))
    d = rng.normal(size=(n, 3))
    rays["dir"] = d / np.linalg.norm(d, axis=1, keepdims=True)
    rays["dist_min"], rays["dist_max"] = 1e-4, np.inf
    ho, hr = so.trace(rays), sr.trace(rays)
    same = ho["triangle"] == hr["triangle"]
    tie = ~same & (ho["dist"] == hr["dist"])
    assert (same | tie).all() and tie.sum() <= n // 500
    assert_bit_equal(ho["dist"], hr["dist"], "dist")
    assert (ho["triangle"][ho["triangle"] != 0xFFFFFFFF] >= int(sc.instances["first_triangle"].min())).mean() > 0.05


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


def test_spectrum_construction_and_evaluation_bit_exact(oracle_spectral, ref_spectral):
    """a27 + a16: UpsampledContinuousSpectrum constructor (slr_amd/spectra.py) and the per-hit evaluation of all three kinds."""
    import ctypes as C
    from slr_amd import abi, spectra
    from slr_amd import binding
    hip_lib = binding.load_library()
    up = ref_spectral.lib.slr_ref_upsample
    up.argtypes = [C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_void_p]
    rng = np.random.default_rng(0)
    for _ in range(500):
        e = rng.random(3).astype(np.float32)
        spt, space = int(rng.integers(0, 2)), int(rng.integers(0, 2))
        out = np.zeros(3, np.float32)
        assert up(spt, space, float(e[0]), float(e[1]), float(e[2]), out.ctypes.data) == 0
        assert_bit_equal(np.array(spectra.upsample(spt, space, *e), np.float32), out, "upsample")
        mine = np.zeros(3, np.float32)            # the C++ host's constructor (include/slrhip.h)
        assert hip_lib.slrhip_upsample(spt, space, C.c_float(e[0]), C.c_float(e[1]), C.c_float(e[2]), mine.ctypes.data) == 0
        assert_bit_equal(mine, out, "slrhip_upsample")
    sc = scenes.cornell_box_spheres(1.0, 8, 4, "glass")
    d = sc.desc()
    fr, fo = ref_spectral.lib.slr_ref_eval_spectrum, oracle_spectral.lib.slr_oracle_eval_spectrum
    fr.argtypes = fo.argtypes = [C.POINTER(abi.SceneDesc), C.c_uint32, C.c_float, C.c_void_p]
    assert set(sc.spectra["kind"]) == {abi.SPEC_UPSAMPLED, abi.SPEC_REGULAR, abi.SPEC_IRREGULAR}
    for idx in range(len(sc.spectra)):
        for off in rng.random(40).astype(np.float32):
            a, b = np.zeros(16, np.float32), np.zeros(16, np.float32)
            assert fr(C.byref(d), idx, float(off), a.ctypes.data) == 0 and fo(C.byref(d), idx, float(off), b.ctypes.data) == 0
            assert_bit_equal(a, b, "spectrum %d" % idx)


@pytest.mark.parametrize("right", ["glass", "matte"])
def test_spectral_cornell_frame_bit_exact(oracle_spectral, ref_spectral, right):
    sc = scenes.cornell_box_spheres(4.0 / 3.0, 24, 12, right)
    so, sr = oracle_spectral.scene(sc), ref_spectral.scene(sc)
    st = ob.settings(48, 36, seed=4242)
    fo, _ = so.render(st, 8)
    fr, _ = sr.render(st, 8)
    assert_bit_equal(fo, fr, "spectral cornell " + right)


@pytest.mark.parametrize("mode", ["rgb", "spectral"])
def test_light_sampling_and_multibsdf_frames_bit_exact(request, oracle_rgb, oracle_spectral, mode):
    """Fresh inputs for the two newest pins: selectLight + Light::sample on 4096 queries, and whole frames of the
    MultiBSDF scene at a size the goldens do not hold."""
    orc = oracle_rgb if mode == "rgb" else oracle_spectral
    ref = request.getfixturevalue("ref_" + mode)
    u = np.random.default_rng(91).random((4096, 3)).astype(np.float32)
    for sc in (scenes.ibl_test_scene(1.0, (128, 64), 12, 6, area_light=True), scenes.cornell_box_boxes(1.0)):
        assert_bit_equal(orc.scene(sc).light_kat(u, 0.9, 0.1), ref.scene(sc).light_kat(u, 0.9, 0.1), "light_kat " + sc.name)
    sc = scenes.cornell_multi(1.0, 14, 7)
    st = ob.settings(56, 44, seed=17)
    a, _ = orc.scene(sc).render(st, 6)
    b, _ = ref.scene(sc).render(st, 6)
    assert_bit_equal(a, b, "cornell_multi " + mode)


@pytest.mark.parametrize("mode", ["rgb", "spectral"])
def test_textured_scene_bit_exact(request, oracle_rgb, oracle_spectral, mode):
    """texture coordinates from the original barycentrics, CheckerBoardSpectrumTexture in a reflectance and in a
    mirror coefficient, CheckerBoardNormal3DTexture through BumpSingleSurfaceObject, CheckerBoardFloatTexture as a triangle's
    alpha texture — the reference's own classes (ref_shim builds them) against the restatement on a fresh size and seed:
    frames, single samples and closest hits through the alpha-cut quad."""
    orc = oracle_rgb if mode == "rgb" else oracle_spectral
    ref = request.getfixturevalue("ref_" + mode)
    sc = scenes.cornell_textured(4.0 / 3.0, 14, 7)
    st = ob.settings(64, 48, seed=2718)
    so, sr = orc.scene(sc), ref.scene(sc)
    a, _ = so.render(st, 6)
    b, _ = sr.render(st, 6)
    assert_bit_equal(a, b, "cornell_textured " + mode)
    rng = np.random.default_rng(5)
    rays = np.zeros(1024, dtype=ob.ray_dtype)
    rays["org"] = rng.uniform([-1.2, 0.2, 0.5], [1.2, 2.0, 2.4], size=(1024, 3)).astype(np.float32)
    d = rng.normal(size=(1024, 3)) * [0.5, 0.4, 0.2] - [0, 0, 1.0]                 # towards the lattice in front of the back wall
    rays["dir"] = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    rays["dist_max"] = np.inf
    ho, hr = so.trace(rays), sr.trace(rays)
    assert (ho["triangle"] == hr["triangle"]).all()
    lattice = len(sc.triangles) - 2
    assert (ho["triangle"] >= lattice).sum() > 30 and (ho["triangle"] < lattice).sum() > 100    # some rays pass through the holes
    hit = hr["triangle"] != 0xFFFFFFFF
    for f in ("dist", "b0", "b1"):
        assert_bit_equal(ho[f][hit], hr[f][hit], f)
