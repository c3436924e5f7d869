"""GPU (MI355X): the HIP path, called through the C ABI, against the CPU oracle on the same
seeded inputs, and against the golden fixtures produced by the compiled reference."""
import os

import numpy as np
import pytest

from helpers import assert_bit_equal, libm_tolerance, load_golden, scene_from_golden
from oracle import binding as ob
from slr_amd import Context, abi, scenes

pytestmark = pytest.mark.gpu

SCENES = ["rgb_tiny_box", "rgb_cornell_glass", "rgb_cornell_matte"]

# Floors for helpers.libm_tolerance (fraction of floats within 2e-6 relative of the reference's), one notch under what was
# measured on MI355X for each scene (profiles/r03_a_parity_stats.jsonl).  Scenes whose paths never call float libm are not
# here: they are tested bit for bit.
TOL = {"rgb_ggx_metal": 0.998, "rgb_ggx_glass": 0.985, "rgb_ward": 0.9995, "rgb_ashikhmin": 0.9995, "rgb_ibl": 0.9995, "rgb_ibl_area": 0.9995,
       "rgb_boxes": 0.999, "rgb_multi": 0.999, "spectral_boxes": 0.998, "spectral_ibl": 0.9995, "spectral_ggx_metal": 0.998,
       "spectral_ggx_glass": 0.98, "spectral_ashikhmin": 0.998, "spectral_multi": 0.9995}


def frame_stats(a, b):
    a, b = a.astype(np.float64), b.astype(np.float64)
    diff = np.abs(a - b)
    exact = (a.astype(np.float32).view(np.uint32) == b.astype(np.float32).view(np.uint32)) | ((a == 0) & (b == 0))
    return dict(rmse=float(np.sqrt(np.mean((a - b) ** 2))), max_abs=float(diff.max()), exact_fraction=float(exact.mean()),
                mean=float(a.mean()))


@pytest.fixture(scope="module")
def ctx():
    c = Context(device=0, mode=abi.MODE_RGB, stripes=1)
    yield c
    c.close()


@pytest.mark.parametrize("name", SCENES)
def test_closest_hits_match_golden(ctx, name):
    g = load_golden(name)
    ctx.upload_scene(scene_from_golden(g))
    rays, want = g["rays"], g["hits"]
    tri, dist, b0, b1 = ctx.trace_rays(rays["org"], rays["dir"], rays["dist_min"], rays["dist_max"])
    assert (tri == want["triangle"]).all()
    hit = want["triangle"] != 0xFFFFFFFF
    assert_bit_equal(dist[hit], want["dist"][hit], "dist")
    assert_bit_equal(b0[hit], want["b0"][hit], "b0")
    assert_bit_equal(b1[hit], want["b1"][hit], "b1")


@pytest.mark.parametrize("name", SCENES)
def test_frame_matches_reference_golden(ctx, name):
    """stripes = 1 keeps the sensor's accumulation order, so the frame is expected bit-exact;
    the stated tolerance (libm last-bit differences in double cos/sin) is <= 0.1 % of floats."""
    g = load_golden(name)
    st = ob.settings(int(g["width"]), int(g["height"]), int(g["seed"]))
    fb = ctx.render_image(scene_from_golden(g), st, int(g["spp"]))
    s = frame_stats(fb, g["framebuffer"])
    assert s["exact_fraction"] >= 0.999, s
    assert s["rmse"] <= 1e-3 * max(s["mean"], 1e-6), s
    c = ctx.counters()
    assert c.samples == int(g["width"]) * int(g["height"]) * int(g["spp"])


def test_continued_render_matches_single_render(ctx):
    g = load_golden("rgb_cornell_glass")
    st = ob.settings(int(g["width"]), int(g["height"]), int(g["seed"]))
    ctx.upload_scene(scene_from_golden(g))
    ctx.render_begin(st)
    half = int(g["spp"]) // 2
    ctx.render(0, half)
    s = frame_stats(ctx.read_framebuffer(), g["framebuffer_half"])
    assert s["exact_fraction"] >= 0.999, s
    ctx.render(half, half)
    s = frame_stats(ctx.read_framebuffer(), g["framebuffer"])
    assert s["exact_fraction"] >= 0.999, s


@pytest.mark.parametrize("right", ["glass", "matte"])
def test_full_cornell_against_oracle(ctx, oracle_rgb, right):
    """Config-1/2 geometry (4 428 triangles) at a size the oracle finishes in seconds."""
    sc = scenes.cornell_box_spheres(4.0 / 3.0, 48, 24, right)
    st = ob.settings(160, 120, seed=777)
    want, ctr = oracle_rgb.scene(sc).render(st, 16)
    fb = ctx.render_image(sc, st, 16)
    s = frame_stats(fb, want)
    assert s["exact_fraction"] >= 0.999, s
    assert s["rmse"] <= 1e-3 * s["mean"], s
    c = ctx.counters()
    assert c.samples == ctr.samples
    assert abs(int(c.extension_rays) - int(ctr.extension_rays)) <= ctr.extension_rays * 1e-4
    assert abs(int(c.shadow_rays) - int(ctr.shadow_rays)) <= ctr.shadow_rays * 1e-4


@pytest.mark.parametrize("mode", ["rgb", "spectral"])
def test_stripes_do_not_change_the_image(request, mode):
    """More paths in flight (slrhip_config::stripes sizes the number of slots) change nothing: a sample is a function of
    (pixel, pass) and the sensor adds a pixel's samples in pass order (k_fold), whichever slot rendered them.  Also with the
    result window cut into pieces of a few passes (SLRHIP_RESULT_WINDOW_MB)."""
    sc = scenes.cornell_box_spheres(1.0, 24, 12, "glass")
    st = ob.settings(64, 64, seed=5)
    want, _ = request.getfixturevalue("oracle_" + mode).scene(sc).render(st, 32)
    amode = abi.MODE_RGB if mode == "rgb" else abi.MODE_SPECTRAL
    frames = {}
    for stripes in (1, 3, 8, 64, 0):
        c = Context(mode=amode, stripes=stripes)
        frames[stripes] = c.render_image(sc, st, 32)
        assert c.counters().samples == 64 * 64 * 32
        c.close()
    for stripes in (3, 8, 64, 0):
        assert_bit_equal(frames[stripes], frames[1], "stripes %d vs 1" % stripes)
    assert np.allclose(frames[1], want, rtol=2e-6, atol=1e-9)
    os.environ["SLRHIP_RESULT_WINDOW_MB"] = "1"           # 64 x 64 pixels x 16 B (64 B spectral) per pass: windows of 16 (4) passes
    try:
        c = Context(mode=amode, stripes=8)
        windows = c.render_image(sc, st, 32)
        c.close()
    finally:
        del os.environ["SLRHIP_RESULT_WINDOW_MB"]
    assert_bit_equal(windows, frames[1], "two result windows vs one")


def test_shards_sum_to_full_image(ctx, oracle_rgb):
    sc = scenes.cornell_box_spheres(4.0 / 3.0, 16, 8, "matte")
    st = ob.settings(100, 76, seed=11)      # not a multiple of the 8x8 tile
    want, _ = oracle_rgb.scene(sc).render(st, 4)
    total = np.zeros_like(want)
    ctx.upload_scene(sc)
    for i in range(3):
        ctx.render_begin(st, shard=(i, 3))
        ctx.render(0, 4)
        part = ctx.read_framebuffer()
        assert ((total != 0) & (part != 0)).sum() == 0
        total += part
    s = frame_stats(total, want)
    assert s["exact_fraction"] >= 0.999, s


@pytest.mark.parametrize("size,spp,stripes", [((1, 1), 7, 1), ((3, 5), 5, 3), ((9, 8), 4, 64), ((37, 21), 3, 0)])
def test_ragged_sizes_against_oracle(oracle_rgb, size, spp, stripes):
    """Images smaller than a tile, not a multiple of 8, fewer samples than stripes, stripes that do not divide spp, and the
    automatic stripe choice: every pixel still gets exactly its passes, added in pass order (bit-exact at any slot count)."""
    sc = scenes.tiny_box(size[0] / size[1])
    st = ob.settings(size[0], size[1], seed=21)
    want, ctr = oracle_rgb.scene(sc).render(st, spp)
    c = Context(stripes=stripes)
    fb = c.render_image(sc, st, spp)
    k = c.counters()
    c.close()
    assert k.samples == size[0] * size[1] * spp == ctr.samples
    assert int(k.extension_rays) == int(ctr.extension_rays) and int(k.shadow_rays) == int(ctr.shadow_rays)
    assert_bit_equal(fb, want, "stripes=%d" % stripes)


def test_large_image_against_oracle(oracle_rgb):
    """3.1 M pixels x 4 spp (12.6 M samples, automatic stripes = 2).  Oracle and GPU traverse different trees, so a ray that
    grazes a box boundary within float rounding can be resolved differently (the reference has the same dependence on its
    own tree): observed 2 of 3.5e7 rays.  Bound: ray counts within 1e-6, >= 99.999 % of floats close."""
    sc = scenes.cornell_box_spheres(4.0 / 3.0, 16, 8, "glass")
    st = ob.settings(2048, 1536, seed=31)
    want, ctr = oracle_rgb.scene(sc).render(st, 4)
    c = Context()
    fb = c.render_image(sc, st, 4)
    k = c.counters()
    c.close()
    assert int(k.samples) == int(ctr.samples)
    assert abs(int(k.extension_rays) - int(ctr.extension_rays)) <= 1e-6 * ctr.extension_rays
    assert abs(int(k.shadow_rays) - int(ctr.shadow_rays)) <= 1e-6 * ctr.shadow_rays
    assert np.isclose(fb, want, rtol=2e-6, atol=1e-9).mean() >= 0.99999


def test_empty_shards_and_empty_renders(ctx):
    """More shards than tiles: the surplus shards own no pixel and must render (nothing) without error; zero passes is a no-op."""
    sc = scenes.tiny_box(1.0)
    st = ob.settings(8, 8, seed=2)
    ctx.upload_scene(sc)
    ctx.render_begin(st, shard=(0, 5))
    ctx.render(0, 2)
    full = ctx.read_framebuffer()
    assert (full != 0).any()
    for i in range(1, 5):
        ctx.render_begin(st, shard=(i, 5))
        ctx.render(0, 2)
        assert not ctx.read_framebuffer().any()
        assert ctx.counters().samples == 0
    ctx.render_begin(st)
    ctx.render(0, 0)
    assert not ctx.read_framebuffer().any()


def test_spectral_continued_render_matches_single_render(sctx):
    g = load_golden("spectral_cornell_matte")
    st = ob.settings(int(g["width"]), int(g["height"]), int(g["seed"]))
    sctx.upload_scene(scene_from_golden(g))
    sctx.render_begin(st)
    half = int(g["spp"]) // 2
    sctx.render(0, half)
    assert_bit_equal(sctx.read_framebuffer(), g["framebuffer_half"], "first half")
    sctx.render(half, half)
    assert_bit_equal(sctx.read_framebuffer(), g["framebuffer"], "both halves")


def test_wave_specialised_traversal_is_reproducible_and_matches_the_oracle(oracle_rgb):
    """The wave-specialised traversal (producer wave, LDS ray ring, refilled consumer lanes) must give the same frame and the
    same ray counts run after run, and the oracle's: enough rays (4.7 M slots, ring wrapping thousands of times per
    workgroup) that a hand-off race in the ring shows up as a differing hash (round 1 found one exactly so)."""
    sc = scenes.cornell_box_spheres(1.0, 48, 24, "glass")
    st = ob.settings(768, 768, seed=5)
    want, ctr = oracle_rgb.scene(sc).render(st, 32)
    frames, counts = [], []
    for _ in range(2):
        c = Context(stripes=8)
        frames.append(c.render_image(sc, st, 32))
        k = c.counters()
        counts.append((int(k.extension_rays), int(k.shadow_rays), int(k.samples)))
        c.close()
    assert counts[0] == counts[1], counts
    assert_bit_equal(frames[1], frames[0], "wave-specialised, second run")
    # against the oracle: among 56 M rays one or two meet two triangles at the same distance where the trees differ in which
    # of them they test (the tie rule is tree-independent only among the triangles a tree TESTS)
    want_counts = (int(ctr.extension_rays), int(ctr.shadow_rays), int(ctr.samples))
    assert counts[0][2] == want_counts[2] and all(abs(a - b) <= 1e-6 * b for a, b in zip(counts[0][:2], want_counts[:2])), (counts[0], want_counts)
    assert np.isclose(frames[0], want, rtol=2e-6, atol=1e-9).mean() >= 0.99999


def test_quantized_nodes_give_the_same_hits(oracle_rgb):
    """Trees of >= 64 Ki nodes are traversed through 64-byte nodes with 8-bit child boxes (rounded outwards on the host):
    a superset of the float boxes, so frame and ray counts must equal the oracle's (its own binary tree, float boxes)."""
    sc = scenes.displaced_grid(400, 16.0 / 9.0)
    st = ob.settings(320, 180, seed=9)
    want, ctr = oracle_rgb.scene(sc).render(st, 16)
    c = Context(stripes=1)
    fb = c.render_image(sc, st, 16)
    k = c.counters()
    c.close()
    assert k.bvh_nodes >= 65536
    assert (int(k.extension_rays), int(k.shadow_rays), int(k.samples)) == (int(ctr.extension_rays), int(ctr.shadow_rays), int(ctr.samples))
    assert_bit_equal(fb, want, "quantized wave-specialised vs the oracle")


def test_quantized_grid_against_oracle_and_reference_golden(oracle_rgb):
    """BASELINE configs[4]'s shape: the displaced grid (320 002 triangles -> >= 64 Ki four-wide nodes, so the DEFAULT
    wave-specialised kernel traverses the 64-byte quantized nodes), thin lens r = 0.025.  Against the ORACLE at matched
    seeds: frame and both ray counts.  Against the compiled reference's golden (tests/golden/rgb_grid400.npz; SBVH.h:417-442,
    TriangleMesh.cpp:131-178): the 2 048 closest hits bit-equal, the frame equal except at equal-distance hits (see
    test_oracle_golden.test_displaced_grid_matches_reference)."""
    from test_oracle_golden import procedural_scene
    g = load_golden("rgb_grid400")
    sc = procedural_scene(g)
    assert sc.camera.lens_radius == np.float32(0.025)
    c = Context(stripes=1)
    try:
        st = ob.settings(int(g["width"]), int(g["height"]), int(g["seed"]))
        fb = c.render_image(sc, st, int(g["spp"]))
        k = c.counters()
        assert k.bvh_nodes >= 65536
        tri, dist, b0, b1 = c.trace_rays(g["rays"]["org"], g["rays"]["dir"], g["rays"]["dist_min"], g["rays"]["dist_max"])
        want = g["hits"]
        assert (tri == want["triangle"]).all()
        hit = want["triangle"] != 0xFFFFFFFF
        assert_bit_equal(dist[hit], want["dist"][hit], "dist")
        assert_bit_equal(b0[hit], want["b0"][hit], "b0")
        assert_bit_equal(b1[hit], want["b1"][hit], "b1")
        s = frame_stats(fb, g["framebuffer"])
        assert s["exact_fraction"] >= 0.9995 and s["rmse"] <= 1e-3 * s["mean"], s
        # a larger frame against the oracle, default stripes (the timed configuration), ray counts included
        st2 = ob.settings(320, 180, seed=9)
        worc, ctr = oracle_rgb.scene(sc).render(st2, 16)
        c.upload_scene(sc)
        c.render_begin(st2)
        c.render(0, 16)
        fb2, k2 = c.read_framebuffer(), c.counters()
        assert_bit_equal(fb2, worc, "grid, stripes = 1, vs oracle")
        assert int(k2.samples) == int(ctr.samples) == 320 * 180 * 16
        assert int(k2.extension_rays) == int(ctr.extension_rays) and int(k2.shadow_rays) == int(ctr.shadow_rays)
    finally:
        c.close()
    a = Context()          # automatic stripe count + sample pool: the summation order of a pixel's stripes changes, nothing else
    try:
        fb3 = a.render_image(sc, st2, 16)
        k3 = a.counters()
        assert int(k3.extension_rays) == int(ctr.extension_rays) and int(k3.shadow_rays) == int(ctr.shadow_rays)
        assert np.allclose(fb3, worc, rtol=2e-6, atol=1e-9)
    finally:
        a.close()


def test_spectral_boxes_against_oracle_and_reference_golden(sctx, oracle_spectral):
    """BASELINE configs[2] exactly: scenes.cornell_box_boxes() (GGX titanium box, alpha_g 0.1, + matte box) in SPECTRAL mode.
    Against the compiled spectral reference's golden (MicrofacetBSDF.cpp:11-110 on its own BSDF objects) and, at 128x128x16 spp,
    against the oracle: the float-libm tolerance of helpers.libm_tolerance (device acosf / tanf / atan2f differ from glibc in
    the last ulp); ray counts within 1e-4."""
    g = load_golden("spectral_boxes")
    st = ob.settings(int(g["width"]), int(g["height"]), int(g["seed"]))
    fb = sctx.render_image(scene_from_golden(g), st, int(g["spp"]))
    want = g["framebuffer"]
    assert fb.shape == want.shape and fb.shape[2] == 16
    libm_tolerance(fb, want, "spectral_boxes vs reference golden", within=TOL["spectral_boxes"])
    sc = scenes.cornell_box_boxes()
    st = ob.settings(128, 128, seed=3)
    want, ctr = oracle_spectral.scene(sc).render(st, 16)
    fb = sctx.render_image(sc, st, 16)
    libm_tolerance(fb, want, "spectral boxes 128x128x16 vs oracle", within=TOL["spectral_boxes"])
    k = sctx.counters()
    assert int(k.samples) == int(ctr.samples) == 128 * 128 * 16
    assert abs(int(k.extension_rays) - int(ctr.extension_rays)) <= ctr.extension_rays * 1e-4
    assert abs(int(k.shadow_rays) - int(ctr.shadow_rays)) <= ctr.shadow_rays * 1e-4


def test_first_render_call_may_start_at_a_later_pass(ctx, oracle_rgb):
    """slrhip_render(ctx, spp_begin, ...) is an ABI promise (include/slrhip.h): the FIRST call after render_begin may start at
    pass k > 0 (a second device rendering the later half of the passes).  Against oracle.render(sppBegin = k)."""
    sc = scenes.cornell_box_spheres(1.0, 16, 8, "glass")
    st = ob.settings(72, 48, seed=17)
    want, ctr = oracle_rgb.scene(sc).render(st, 9, spp_begin=23)
    ctx.upload_scene(sc)
    ctx.render_begin(st)
    ctx.render(23, 9)
    assert_bit_equal(ctx.read_framebuffer(), want, "passes 23..31, stripes = 1")
    k = ctx.counters()
    assert int(k.samples) == int(ctr.samples) and int(k.extension_rays) == int(ctr.extension_rays)
    a = Context(stripes=8)
    try:
        a.upload_scene(sc)
        a.render_begin(st)
        a.render(23, 9)
        assert np.allclose(a.read_framebuffer(), want, rtol=2e-6, atol=1e-9)
        assert int(a.counters().samples) == 72 * 48 * 9
    finally:
        a.close()


def test_two_contexts_on_non_blocking_streams_from_two_threads(oracle_rgb):
    """Regression for round 1's `iteration bound exceeded` (gpurun_out/overlap.log): two contexts on ONE device, each driven by
    its own host thread on its own NON-BLOCKING stream, the first render call of one of them starting at pass k > 0.  The
    cause was render_begin's null-stream memset of the queue counters racing k_reset_slots on the non-blocking stream
    (DESIGN.md); each half is checked against the oracle, and their Kahan-free sum against the full render."""
    import ctypes as C
    import threading
    # Non-blocking streams from the HIP runtime libslrhip.so itself is linked against — by its full path: torch ships a second
    # copy of the runtime (imported by other test modules at collection), and a bare "libamdhip64.so" may resolve to that copy,
    # which then reports hipErrorNoDevice
    import re
    from slr_amd import binding
    binding.load_library()
    paths = set(re.findall(r"(/\S*rocm\S*/libamdhip64\.so\S*)", open("/proc/self/maps").read()))
    assert len(paths) == 1, paths
    hip = C.CDLL(paths.pop())
    hip.hipStreamCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
    hip.hipStreamDestroy.argtypes = [C.c_void_p]

    def non_blocking_stream():
        s = C.c_void_p()
        assert hip.hipStreamCreateWithFlags(C.byref(s), 1) == 0       # hipStreamNonBlocking
        return s
    sc = scenes.cornell_box_spheres(1.0, 24, 12, "matte")
    st = ob.settings(256, 192, seed=5)
    SPP = 32
    o = oracle_rgb.scene(sc)
    halves = [o.render(st, SPP // 2, spp_begin=i * SPP // 2)[0].copy() for i in range(2)]
    for attempt in range(1):      # the fix is a stream-order change, deterministic by construction: one pass is the test
        ctxs = [Context(stripes=4), Context(stripes=4)]
        streams = [non_blocking_stream(), non_blocking_stream()]
        got, errors = [None, None], []

        def work(i):
            try:
                ctxs[i].upload_scene(sc)
                ctxs[i].render_begin(st)
                ctxs[i].render(i * SPP // 2, SPP // 2, streams[i])
                got[i] = ctxs[i].read_framebuffer()
            except Exception as e:
                errors.append(e)
        th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
        [t.start() for t in th]
        [t.join() for t in th]
        [c.close() for c in ctxs]
        [hip.hipStreamDestroy(s_) for s_ in streams]
        assert not errors, errors
        for i in range(2):
            assert np.allclose(got[i], halves[i], rtol=2e-6, atol=1e-9), "context %d, attempt %d" % (i, attempt)


def test_device_built_tree_gives_the_same_image(oracle_rgb):
    """SURVEY 8 row f2, second half: tree, quantized nodes, leaf packets and per-triangle shading records built on the GPU
    (bvh_device.hip: LBVH over Morton codes + the host build's 4-wide collapse; reference build path Accelerator/SBVH.h:379-407,
    QBVH.h:254-284).  Another tree over the same triangles: the reference golden's closest hits must come out bit-equal, the frame
    and the ray counts must equal the oracle's (up to the one-in-a-million equal-distance ray whose winner depends on which
    triangles a tree tests), on a float-node tree (Cornell, 4 428 triangles) and on a quantized one (320 002)."""
    from test_oracle_golden import procedural_scene
    g = load_golden("rgb_grid400")
    for name, sc, st, spp in (("cornell", scenes.cornell_box_spheres(4.0 / 3.0, 48, 24, "glass"), ob.settings(160, 120, seed=12), 8),
                              ("grid", procedural_scene(g), ob.settings(320, 180, seed=9), 16),
                              ("grid640", scenes.displaced_grid(640, 16.0 / 9.0), ob.settings(160, 90, seed=4), 4)):      # 819 200 triangles: quantized nodes on either tree
        want, ctr = oracle_rgb.scene(sc).render(st, spp)
        out = {}
        for label, flags in (("host", abi.FLAG_COUNT_TRAVERSAL), ("device", abi.FLAG_COUNT_TRAVERSAL | abi.FLAG_BVH_DEVICE_BUILD)):
            c = Context(stripes=1, flags=flags)
            try:
                fb = c.render_image(sc, st, spp)
                k, p = c.counters(), c.profile()
                out[label] = (fb, int(k.extension_rays), int(k.shadow_rays), int(k.samples), int(k.bvh_nodes), p.nodes[0] / p.rays[0], p.triangles[0] / p.rays[0], k.build_seconds)
                if name == "grid640":
                    assert k.bvh_nodes >= 65536
                if name == "grid":
                    tri, dist, b0, b1 = c.trace_rays(g["rays"]["org"], g["rays"]["dir"], g["rays"]["dist_min"], g["rays"]["dist_max"])
                    hit = g["hits"]["triangle"] != 0xFFFFFFFF
                    assert (tri == g["hits"]["triangle"]).all(), label
                    assert_bit_equal(dist[hit], g["hits"]["dist"][hit], label + " dist")
                    assert_bit_equal(b0[hit], g["hits"]["b0"][hit], label + " b0")
            finally:
                c.close()
        print(name, {k: v[4:] for k, v in out.items()})
        assert_bit_equal(out["host"][0], want, name + ": host-built tree vs oracle")
        s = frame_stats(out["device"][0], want)
        assert s["exact_fraction"] >= 0.9999, (name, s)
        assert out["device"][3] == out["host"][3] == int(ctr.samples)
        assert all(abs(a - b) <= 1e-5 * b for a, b in zip(out["device"][1:3], (int(ctr.extension_rays), int(ctr.shadow_rays)))), (name, out["device"][1:3])
        assert out["device"][5] < 2.0 * out["host"][5]              # nodes per extension ray: an LBVH is worse than the SAH tree, not absurdly so


def test_spatial_split_tree_gives_the_same_image(oracle_rgb):
    """the tree built with spatial splits (sbvh.cpp: the reference's SBVH, Accelerator/SBVH.h:57-348 — references
    duplicated across split planes with clipped boxes, Surface/TriangleMesh.cpp:19-125) is another tree over the same triangles:
    frame, ray counts and closest hits must equal the object-split tree's and the oracle's bit for bit, it must actually contain
    duplicated references on this scene (the walls' two-triangle quads span the whole box), and test fewer triangles per ray."""
    sc = scenes.cornell_box_spheres(4.0 / 3.0, 32, 16, "glass")
    st = ob.settings(160, 120, seed=12)
    want, ctr = oracle_rgb.scene(sc).render(st, 8)
    g = load_golden("rgb_cornell_glass")
    out = {}
    for name, flags in (("sah", abi.FLAG_COUNT_TRAVERSAL), ("sbvh", abi.FLAG_COUNT_TRAVERSAL | abi.FLAG_BVH_SPATIAL_SPLITS)):
        c = Context(stripes=1, flags=flags)
        try:
            fb = c.render_image(sc, st, 8)
            k, p = c.counters(), c.profile()
            out[name] = (fb, int(k.extension_rays), int(k.shadow_rays), int(k.bvh_leaf_references), p.triangles[0] / p.rays[0], p.nodes[0] / p.rays[0])
            c.upload_scene(scene_from_golden(g))
            r = g["rays"]
            tri, dist, b0, b1 = c.trace_rays(r["org"], r["dir"], r["dist_min"], r["dist_max"])
            assert (tri == g["hits"]["triangle"]).all(), name
            hit = g["hits"]["triangle"] != 0xFFFFFFFF
            assert_bit_equal(dist[hit], g["hits"]["dist"][hit], name + " dist")
        finally:
            c.close()
    assert_bit_equal(out["sah"][0], want, "object-split tree vs oracle")
    assert_bit_equal(out["sbvh"][0], want, "spatial-split tree vs oracle")
    assert out["sah"][1:3] == out["sbvh"][1:3] == (int(ctr.extension_rays), int(ctr.shadow_rays))
    assert out["sah"][3] == len(sc.triangles) and out["sbvh"][3] > len(sc.triangles)
    assert out["sbvh"][4] < out["sah"][4]            # triangles tested per extension ray


@pytest.mark.parametrize("name", ["rgb_textured", "spectral_textured"])
def test_textured_scene_matches_reference_golden(name):
    """SURVEY 8 row f3, textured half: a checkerboard reflectance on the floor and in a mirror's coefficient (CheckerBoardSpectrumTexture
    at the texture coordinate Triangle::intersect interpolates from the original barycentrics), a bump-mapped Oren-Nayar sphere
    (CheckerBoardNormal3DTexture through BumpSingleSurfaceObject) and an alpha-cut quad (CheckerBoardFloatTexture as
    Triangle::m_alphaTex, tested INSIDE the traversal kernels).  No float libm beyond fmod on this path: frame, closest hits
    through the lattice and ray counts are expected bit-exact against the compiled reference's golden and the oracle."""
    g = load_golden(name)
    mode = abi.MODE_SPECTRAL if name.startswith("spectral") else abi.MODE_RGB
    sc = scene_from_golden(g)
    st = ob.settings(int(g["width"]), int(g["height"]), int(g["seed"]))
    for flags in (0,):
        c = Context(mode=mode, stripes=1, flags=flags)
        try:
            fb = c.render_image(sc, st, int(g["spp"]))
            assert_bit_equal(fb, g["framebuffer"], name + " frame")
            r = g["rays"]
            tri, dist, b0, b1 = c.trace_rays(r["org"], r["dir"], r["dist_min"], r["dist_max"])
            want = g["hits"]
            assert (tri == want["triangle"]).all()
            hit = want["triangle"] != 0xFFFFFFFF
            assert_bit_equal(dist[hit], want["dist"][hit], "dist")
            assert_bit_equal(b0[hit], want["b0"][hit], "b0")
            assert_bit_equal(b1[hit], want["b1"][hit], "b1")
        finally:
            c.close()
    # a larger frame with the automatic stripe count against the oracle
    sc2 = scenes.cornell_textured(4.0 / 3.0, 20, 10)
    st2 = ob.settings(200, 150, seed=8)
    want, ctr = ob.load("oracle", mode).scene(sc2).render(st2, 8)
    c = Context(mode=mode)
    try:
        fb = c.render_image(sc2, st2, 8)
        k = c.counters()
    finally:
        c.close()
    assert_bit_equal(fb, want, "automatic slot count vs the oracle")
    assert int(k.extension_rays) == int(ctr.extension_rays) and int(k.shadow_rays) == int(ctr.shadow_rays)


@pytest.mark.parametrize("name", ["rgb_image_textured", "spectral_image_textured"])
def test_image_textured_scene_matches_reference_golden(name):
    """SURVEY 8 row f3, image textures in material slots: nearest texel of an image behind an offset-and-scale mapping
    (ImageSpectrumTexture, Textures/image_textures.cpp:13-79), RGB texels in the RGB build, (u, v, s) through the run-time Meng-15
    look-up in the spectral build.  No float libm beyond fmod on the path: the frame is expected bit-exact against the golden
    (compiled reference; its texel addressing is the shim's restatement — OpenEXR half is absent — so that step is unpinned) and,
    on a larger frame with the automatic stripe count, equal to the oracle's up to the stripes' summation order."""
    g = load_golden(name)
    mode = abi.MODE_SPECTRAL if name.startswith("spectral") else abi.MODE_RGB
    st = ob.settings(int(g["width"]), int(g["height"]), int(g["seed"]))
    c = Context(mode=mode, stripes=1)
    try:
        fb = c.render_image(scene_from_golden(g), st, int(g["spp"]))
        assert_bit_equal(fb, g["framebuffer"], name + " frame")
    finally:
        c.close()
    sc2 = scenes.cornell_image_textured(4.0 / 3.0, 16, 8)
    st2 = ob.settings(160, 120, seed=8)
    want, ctr = ob.load("oracle", mode).scene(sc2).render(st2, 8)
    c = Context(mode=mode)
    try:
        fb = c.render_image(sc2, st2, 8)
        k = c.counters()
    finally:
        c.close()
    assert_bit_equal(fb, want, "automatic slot count vs the oracle")
    assert int(k.extension_rays) == int(ctr.extension_rays) and int(k.shadow_rays) == int(ctr.shadow_rays)


@pytest.mark.parametrize("name", ["rgb_instanced", "spectral_instanced"])
def test_instanced_scene_matches_reference_golden(name):
    """Instanced meshes (slrhip_instance; TransformedSurfaceObject over a mesh aggregate, Core/SurfaceObject.cpp:303-392): two-level
    traversal in k_trace_ws (ray to local space and back, un-normalised direction), the surface point through the instance
    transform in the shade kernel.  No float libm on these paths: closest hits and the frame are expected bit-exact against the
    golden of the compiled reference, and a larger frame with the automatic stripe count equal to the oracle's up to the stripes'
    summation order, with the same ray counts."""
    g = load_golden(name)
    mode = abi.MODE_SPECTRAL if name.startswith("spectral") else abi.MODE_RGB
    st = ob.settings(int(g["width"]), int(g["height"]), int(g["seed"]))
    c = Context(mode=mode, stripes=1)
    try:
        sc = scene_from_golden(g)
        fb = c.render_image(sc, st, int(g["spp"]))
        assert_bit_equal(fb, g["framebuffer"], name + " frame")
        rays, want = g["rays"], g["hits"]
        tri, dist, b0, b1 = c.trace_rays(rays["org"], rays["dir"], rays["dist_min"], rays["dist_max"])
        assert (tri == want["triangle"]).all()
        hit = want["triangle"] != 0xFFFFFFFF
        for got, k in ((dist, "dist"), (b0, "b0"), (b1, "b1")):
            assert_bit_equal(got[hit], want[k][hit], k)
    finally:
        c.close()
    sc2 = scenes.cornell_instanced(4.0 / 3.0, 16, 8, copies=24)
    st2 = ob.settings(160, 120, seed=8)
    want, ctr = ob.load("oracle", mode).scene(sc2).render(st2, 8)
    c = Context(mode=mode)
    try:
        fb = c.render_image(sc2, st2, 8)
        k = c.counters()
    finally:
        c.close()
    assert_bit_equal(fb, want, "automatic slot count vs the oracle")
    assert int(k.extension_rays) == int(ctr.extension_rays) and int(k.shadow_rays) == int(ctr.shadow_rays)


def test_errors_are_loud(ctx):
    import ctypes as C
    from slr_amd.binding import SlrHipError
    c2 = Context()
    with pytest.raises(SlrHipError):
        c2.render_begin(ob.settings(8, 8))          # no scene uploaded
    c2.close()
    with pytest.raises(SlrHipError, match="64 sample stripes"):
        Context(stripes=65)                         # slrhip_config::stripes is bounded by SLRHIP_MAX_STRIPES
    # instances the traversal cannot take are refused at upload, with the reason: a transform that is not affine, triangle
    # ranges that overlap without being equal (nested instancing), an instanced triangle that emits
    good = scenes.cornell_instanced(1.0, 8, 4)
    c4 = Context()
    try:
        bad = scenes.cornell_instanced(1.0, 8, 4)
        bad.instances["local_to_world"][0][3] = 0.25                    # bottom row of the matrix (column-major: element 3 of column 0)
        with pytest.raises(SlrHipError, match="affine"):
            c4.upload_scene(bad)
        bad = scenes.cornell_instanced(1.0, 8, 4)
        bad.instances["first_triangle"][1] += 1
        bad.instances["num_triangles"][1] -= 1
        with pytest.raises(SlrHipError, match="equal or disjoint"):
            c4.upload_scene(bad)
        bad = scenes.cornell_instanced(1.0, 8, 4)
        light = int(np.nonzero(bad.materials["emittance"] >= 0)[0][0])
        bad.triangles["material"][int(bad.instances["first_triangle"][0])] = light
        with pytest.raises(SlrHipError, match="must not emit"):
            c4.upload_scene(bad)
        bad = scenes.cornell_instanced(1.0, 8, 4)
        bad.instances["num_triangles"][0] = len(bad.triangles) + 5
        with pytest.raises(SlrHipError, match="out of range"):
            c4.upload_scene(bad)
        c4.upload_scene(good)                                           # and the context is still usable
    finally:
        c4.close()
    # a kernel that gives up (bounded spin, dropped push) raises the device error word: the render must fail, not return rc 0
    c3 = Context(flags=abi.FLAG_TEST_DEVICE_ERROR)
    try:
        c3.upload_scene(scenes.tiny_box(1.0))
        c3.render_begin(ob.settings(16, 16))
        with pytest.raises(SlrHipError, match="device-side error word"):
            c3.render(0, 2)
    finally:
        c3.close()


@pytest.mark.parametrize("name", ["rgb_oren_nayar"])
def test_oren_nayar_matches_reference_golden(ctx, name):
    """Oren-Nayar uses no libm beyond the cosine sample: expected bit-exact like Lambert."""
    g = load_golden(name)
    st = ob.settings(int(g["width"]), int(g["height"]), int(g["seed"]))
    fb = ctx.render_image(scene_from_golden(g), st, int(g["spp"]))
    s = frame_stats(fb, g["framebuffer"])
    assert s["exact_fraction"] >= 0.999, s


@pytest.mark.parametrize("name", ["rgb_ggx_metal", "rgb_ggx_glass", "rgb_ward", "rgb_ashikhmin"])
def test_ggx_matches_reference_golden_within_tolerance(ctx, name):
    """(Also the Ward and Ashikhmin-Shirley lobes: expf, logf, atanf, powf, double pow.)  GGX calls float libm (acosf, atan2f, tanf, cosf, sinf): the device library can differ from glibc in the last ulp,
    which perturbs a sample by ~1e-7 relative and, rarely, flips a discrete decision.  Tolerance (north_star: per-pixel
    RMSE < 1e-3): helpers.libm_tolerance with the per-scene floor of TOL."""
    g = load_golden(name)
    st = ob.settings(int(g["width"]), int(g["height"]), int(g["seed"]))
    fb = ctx.render_image(scene_from_golden(g), st, int(g["spp"]))
    libm_tolerance(fb, g["framebuffer"], name + " vs reference golden", within=TOL[name])


@pytest.mark.parametrize("name", ["rgb_ibl", "rgb_ibl_area"])
def test_environment_light_matches_reference_golden_within_tolerance(ctx, name):
    """Image-based environment light (BASELINE configs[3]): miss -> InfiniteSphere hit, importance-sampled NEE, MIS.
    Float libm on the path (acosf, atan2f, fmodf, sinf, cosf) -> same tolerance as GGX; the sun texels are ~750x the
    mean radiance, so the RMSE bound is taken on frames clipped at 10x the mean (a flipped sun sample is one float)."""
    g = load_golden(name)
    st = ob.settings(int(g["width"]), int(g["height"]), int(g["seed"]))
    fb = ctx.render_image(scene_from_golden(g), st, int(g["spp"]))
    libm_tolerance(fb, g["framebuffer"], name + " vs reference golden", within=TOL[name], cap=10)
    assert ctx.counters().samples == int(g["width"]) * int(g["height"]) * int(g["spp"])


def test_environment_light_full_size_against_oracle(ctx, oracle_rgb):
    sc = scenes.ibl_test_scene(16.0 / 9.0, (256, 128), 24, 12)
    st = ob.settings(160, 90, seed=11)
    want, ctr = oracle_rgb.scene(sc).render(st, 16)
    fb = ctx.render_image(sc, st, 16)
    libm_tolerance(fb, want, "environment light 160x90x16 vs oracle", within=TOL["rgb_ibl"], cap=10)
    c = ctx.counters()
    assert abs(int(c.extension_rays) - int(ctr.extension_rays)) <= ctr.extension_rays * 1e-4
    assert abs(int(c.shadow_rays) - int(ctr.shadow_rays)) <= ctr.shadow_rays * 1e-4


def test_spectral_environment_light_matches_reference_golden_within_tolerance(sctx):
    """Spectral build: environment texels are (u, v, s) and every look-up runs UpsampledContinuousSpectrum::evaluate with its grid
    search (SpectrumTypes.h:239-339) on the device.  Same float-libm tolerance as the RGB environment test."""
    g = load_golden("spectral_ibl")
    st = ob.settings(int(g["width"]), int(g["height"]), int(g["seed"]))
    fb = sctx.render_image(scene_from_golden(g), st, int(g["spp"]))
    want = g["framebuffer"]
    assert fb.shape == want.shape and fb.shape[2] == 16
    libm_tolerance(fb, want, "spectral_ibl vs reference golden", within=TOL["spectral_ibl"], cap=10)


def test_spectral_environment_needs_the_upsampling_tables(sctx):
    import ctypes as C
    from slr_amd.binding import SlrHipError
    scn = scenes.ibl_test_scene(1.0, (64, 32), 8, 4)
    desc = scn.desc(abi.MODE_SPECTRAL)
    desc.upsampling = None
    with pytest.raises(SlrHipError):
        from slr_amd import binding
        binding._check(sctx.lib, sctx.lib.slrhip_upload_scene(sctx.handle, C.byref(desc)), "slrhip_upload_scene")


def test_boxes_scene_against_oracle(ctx, oracle_rgb):
    """BASELINE configs[2] geometry and lobes (GGX titanium box), RGB variant, at a size the oracle finishes in seconds."""
    sc = scenes.cornell_box_boxes()
    st = ob.settings(128, 128, seed=3)
    want, ctr = oracle_rgb.scene(sc).render(st, 16)
    fb = ctx.render_image(sc, st, 16)
    libm_tolerance(fb, want, "boxes RGB 128x128x16 vs oracle", within=TOL["rgb_boxes"])
    c = ctx.counters()
    assert abs(int(c.extension_rays) - int(ctr.extension_rays)) <= ctr.extension_rays * 1e-4


@pytest.fixture(scope="module")
def sctx():
    c = Context(device=0, mode=abi.MODE_SPECTRAL, stripes=1)
    yield c
    c.close()


@pytest.mark.parametrize("name", ["spectral_cornell_glass", "spectral_cornell_matte", "spectral_oren_nayar"])
def test_spectral_frame_matches_reference_golden(sctx, name):
    """Spectral mode (16 wavelengths, 16-bin framebuffer) without float-libm lobes: expected bit-exact."""
    g = load_golden(name)
    st = ob.settings(int(g["width"]), int(g["height"]), int(g["seed"]))
    fb = sctx.render_image(scene_from_golden(g), st, int(g["spp"]))
    assert fb.shape == g["framebuffer"].shape
    s = frame_stats(fb, g["framebuffer"])
    assert s["exact_fraction"] >= 0.999, s
    assert s["rmse"] <= 1e-3 * max(s["mean"], 1e-9), s


@pytest.mark.parametrize("name", ["spectral_ggx_metal", "spectral_ggx_glass", "spectral_ashikhmin"])
def test_spectral_ggx_within_tolerance(sctx, name):
    g = load_golden(name)
    st = ob.settings(int(g["width"]), int(g["height"]), int(g["seed"]))
    fb = sctx.render_image(scene_from_golden(g), st, int(g["spp"]))
    libm_tolerance(fb, g["framebuffer"], name + " vs reference golden", within=TOL[name])


def test_spectral_mode_rejects_rgb_only_spectra(sctx):
    from slr_amd.binding import SlrHipError
    b = scenes.SceneBuilder()
    scenes.cornell_walls(b)
    b.add_uv_sphere(8, 4, b.matte(b.spectrum_rgb(0.5, 0.5, 0.5)), scenes._translate(0, 0.5, 0) @ scenes._scale(0.4))
    with pytest.raises(SlrHipError):
        sctx.upload_scene(b.build(scenes.cornell_camera(1.0)))


# ---- function-level parity: the device BSDF functions against the reference's known answers --------------------------
LIBM_FREE_LOBES = ("lambert", "oren_nayar", "mirror", "glass", "emitter_over_lambert", "sum_lambert_inverse_lambert", "sum_mirror_lambert")


def _assert_lobe_close(got, want, what, exact_rows_floor, far_fraction, far_rel, worst_rel):
    """Lobes that call float libm (GGX: acosf, tanf, atan2f, cosf, sinf; Ward: logf, atanf, expf; Ashikhmin: powf, acosf):
    the device math library and glibc differ in the last ulp, and tan(acos(z)) amplifies that by 1 / z at grazing angles.
    Stated tolerance: the zero / non-zero pattern (every branch taken) and the sampled lobe type agree on all but
    `far_fraction` of the rows, at least `exact_rows_floor` of the floats are bit-equal, at most `far_fraction` of the
    floats are further than `far_rel` relative, none further than `worst_rel`.  Measured on MI355X (tools/bsdf_kat_stats.py):
    66-91 % bit-equal, 99th percentile 1e-5, worst 2e-3 (a direction 5e-5 off the tangent plane)."""
    got64, want64 = got.astype(np.float64), want.astype(np.float64)
    pattern_rows = ((got == 0) != (want == 0)).any(axis=-1)
    assert pattern_rows.mean() <= far_fraction, (what, "branch pattern differs on", int(pattern_rows.sum()), "rows")
    ok_rows = ~pattern_rows
    g, w = got64[ok_rows], want64[ok_rows]
    assert (g[..., 4] == w[..., 4]).all(), (what, "dirType")
    exact = (got[ok_rows].view(np.uint32) == want[ok_rows].view(np.uint32)) | ((g == 0) & (w == 0))
    rel = np.abs(g - w) / np.maximum(np.abs(w), 1e-30)
    rel[exact] = 0
    assert exact.mean() >= exact_rows_floor, (what, "bit-equal fraction", float(exact.mean()))
    assert (rel > far_rel).mean() <= far_fraction, (what, "fraction beyond", far_rel, float((rel > far_rel).mean()))
    # rows 36-43 of scenes.bsdf_queries put the sample numbers at 0 and 1 - 2^-24: microfacet slopes of +-4096, where
    # D(m) goes with tan^-4 and one ulp of acosf moves it by tens of percent; they stay in the checks above only
    regular = np.ones(got.shape[:-1], bool)
    regular[..., 36:44] = False
    rel_regular = rel[regular[ok_rows]]
    assert rel_regular.max() <= worst_rel, (what, "worst relative difference", float(rel_regular.max()))


@pytest.mark.parametrize("mode", ["rgb", "spectral"])
def test_bsdf_queries_match_golden_and_oracle(mode):
    """slrhip_bsdf_queries runs the same device functions k_shade calls.  Against the compiled reference's answers
    (tests/golden/bsdf_kat_*.npz) and, on 4096 fresh queries per lobe, against the oracle.  Lobes without float libm
    calls: every float bit-exact.  The others: the tolerance stated in _assert_lobe_close."""
    g = load_golden("bsdf_kat_" + mode)
    amode = abi.MODE_RGB if mode == "rgb" else abi.MODE_SPECTRAL
    scene = scene_from_golden(g)
    c = Context(device=0, mode=amode)
    try:
        c.upload_scene(scene)
        o = ob.load("oracle", amode).scene(scene)
        fresh = scenes.bsdf_queries(4096, 31)
        for name, m in zip(g["material_names"], g["material_indices"]):
            name, m = str(name), int(m)
            got = np.stack([c.bsdf_queries(m, g["queries"], float(off), float(ul)) for off, ul in g["wavelengths"]])
            got_fresh, want_fresh = c.bsdf_queries(m, fresh, 0.71, 0.33), o.bsdf_kat(m, fresh, 0.71, 0.33)
            if name in LIBM_FREE_LOBES:
                assert_bit_equal(got, g["out_" + name], "%s %s vs golden" % (mode, name))
                assert_bit_equal(got_fresh, want_fresh, "%s %s vs oracle" % (mode, name))
            else:
                _assert_lobe_close(got, g["out_" + name], "%s %s vs golden" % (mode, name), 0.5, 0.01, 1e-4, 1e-2)
                _assert_lobe_close(got_fresh, want_fresh, "%s %s vs oracle" % (mode, name), 0.5, 0.01, 1e-4, 5e-2)
    finally:
        c.close()


def test_bsdf_queries_reject_bad_arguments(ctx):
    ctx.upload_scene(scenes.tiny_box(1.0))
    q = scenes.bsdf_queries(4, 1)
    with pytest.raises(Exception, match="material index"):
        ctx.bsdf_queries(10_000, q)
    with pytest.raises(Exception, match="wl_offset"):
        ctx.bsdf_queries(0, q, 1.0, 0.5)


# ---- MultiBSDF: summed / mixed / inverted materials -----------------------------------------------------
def _render_golden(name):
    g = load_golden(name)
    mode = abi.MODE_SPECTRAL if name.startswith("spectral") else abi.MODE_RGB
    c = Context(device=0, mode=mode, stripes=1)
    try:
        st = ob.settings(int(g["width"]), int(g["height"]), int(g["seed"]))
        fb = c.render_image(scene_from_golden(g), st, int(g["spp"]))
        assert c.counters().samples == int(g["width"]) * int(g["height"]) * int(g["spp"])
    finally:
        c.close()
    return fb, g["framebuffer"]


@pytest.mark.parametrize("name", ["rgb_multi_libm_free", "spectral_multi_libm_free"])
def test_multibsdf_frame_matches_reference_golden(name):
    """sum(mirror, Lambert), mix(Oren-Nayar, Lambert), sum(Lambert, inverse(Lambert)): component selection by weight,
    the mixture PDF and the summed value of MultiBSDF.cpp:20-59, the InverseBSDF forwarding — no float libm on the
    path, so the frame is expected bit-exact against the compiled reference's."""
    fb, want = _render_golden(name)
    s = frame_stats(fb, want)
    assert s["exact_fraction"] >= 0.999, s
    assert s["rmse"] <= 1e-3 * max(s["mean"], 1e-9), s


@pytest.mark.parametrize("name", ["rgb_multi", "spectral_multi"])
def test_multibsdf_with_ggx_component_within_tolerance(name):
    """The same scene with a GGX component in the mix: the float-libm tolerance of the GGX tests."""
    fb, want = _render_golden(name)
    libm_tolerance(fb, want, name + " vs reference golden", within=TOL[name])


@pytest.mark.parametrize("mode", ["rgb", "spectral"])
def test_nested_multibsdf_matches_reference_golden(mode):
    """a summed / mixed material whose components are summed / mixed materials (MultiBSDF.cpp:20-59,125-212
    calling itself through the BSDF interface; up to four lobes).  Function level against the compiled reference's answers on
    six nested materials (tests/golden/bsdf_kat_nested_*.npz): every float bit-equal where no float-libm lobe is a component,
    the libm tolerance otherwise; and a whole frame with four-lobe materials, bit for bit."""
    amode = abi.MODE_RGB if mode == "rgb" else abi.MODE_SPECTRAL
    g = load_golden("bsdf_kat_nested_" + mode)
    c = Context(device=0, mode=amode, stripes=1)
    try:
        c.upload_scene(scene_from_golden(g))
        for name, m in zip(g["material_names"], g["material_indices"]):
            name, m = str(name), int(m)
            got = np.stack([c.bsdf_queries(m, g["queries"], float(off), float(ul)) for off, ul in g["wavelengths"]])
            if name in scenes.NESTED_LIBM_FREE:
                assert_bit_equal(got, g["out_" + name], "%s %s vs golden" % (mode, name))
            else:
                _assert_lobe_close(got, g["out_" + name], "%s %s vs golden" % (mode, name), 0.5, 0.01, 1e-4, 1e-2)
        f = load_golden(mode + "_multi_nested")
        st = ob.settings(int(f["width"]), int(f["height"]), int(f["seed"]))
        fb = c.render_image(scene_from_golden(f), st, int(f["spp"]))
        assert_bit_equal(fb, f["framebuffer"], mode + " nested MultiBSDF frame")
        assert c.counters().samples == int(f["width"]) * int(f["height"]) * int(f["spp"])
    finally:
        c.close()


def test_multibsdf_rejects_what_it_does_not_support(ctx):
    b = scenes.SceneBuilder()
    scenes.cornell_walls(b)
    glass = b.glass(b.spectrum_grey(0.999), b.spectrum_ior("Air", 0, scenes.AIR_ETA_RGB), b.spectrum_ior("Glass_BK7", 0, scenes.BK7_ETA_RGB))
    matte = b.matte(b.spectrum_grey(0.5))
    inner = b.summed(matte, matte)
    deeper = b.summed(inner, matte)                                          # one level of nesting: accepted
    for bad in (lambda: b.summed(matte, glass, inverse=(False, True)),       # inverse of a two-sided lobe
                lambda: b.summed(deeper, matte),                             # two levels of nesting
                lambda: b.summed(inner, matte, inverse=(True, False)),       # inverse of a MULTI component
                lambda: b.material(abi.MAT_MULTI, (matte, 10_000, 0), 1.0, -1, 1.0)):   # component index out of range
        n = len(b.materials)
        bad()
        with pytest.raises(Exception, match="MULTI|inverse"):
            ctx.upload_scene(b.build(scenes.cornell_camera(1.0)))
        del b.materials[n:]


# ---- table placement and size-independent properties at BASELINE size ------------------------------------------------
def _patchwork_scene(num_materials, num_lights):
    """Cornell walls + a floor patchwork in which every patch has its own matte material and some patches emit:
    more materials (> 32) and lights (> 16) than the shade kernel stages in LDS, so the tables stay in HBM."""
    b = scenes.SceneBuilder()
    scenes.cornell_walls(b)
    rng = np.random.default_rng(3)
    side = int(np.ceil(np.sqrt(num_materials)))
    for i in range(num_materials):
        r, g, bl = rng.uniform(0.1, 0.9, 3)
        emit = b.spectrum_d65(0.5 + 0.1 * i, scenes.D65_RGB) if i < num_lights else -1
        m = b.matte(b.spectrum_srgb_nonlinear(float(r), float(g), float(bl)), emittance=emit)
        x, z = -1.2 + 2.4 * (i % side) / side, -1.8 + 3.0 * (i // side) / side
        w, y = 2.2 / side, 0.02 + 0.3 * (i % 3)
        b.add_quad([(x, y, z + w), (x + w, y, z + w), (x + w, y, z), (x, y, z)], (0, 1, 0), (1, 0, 0), m)
    return b.build(scenes.cornell_camera(1.0), name="patchwork")


def test_many_materials_and_lights_use_the_tables_in_hbm(oracle_rgb, oracle_spectral):
    sc = _patchwork_scene(45, 20)
    assert len(sc.materials) > 32 and (sc.materials["emittance"] >= 0).sum() > 16
    st = ob.settings(72, 56, seed=21)
    for mode, orc in ((abi.MODE_RGB, oracle_rgb), (abi.MODE_SPECTRAL, oracle_spectral)):
        want, _ = orc.scene(sc).render(st, 8)
        c = Context(mode=mode, stripes=1)
        fb = c.render_image(sc, st, 8)
        c.close()
        assert_bit_equal(fb, want, "patchwork mode %d" % mode)
        assert want.sum() > 0


def test_radiance_is_linear_in_the_emitted_power_at_full_size():
    """A property that needs no oracle, at BASELINE's 1280x720: scaling every emitter by a power of two scales every float of
    the frame by exactly that factor (products by 2^k are exact, and no decision on a path — BSDF sampling, Russian
    roulette on throughput, light selection among equal importances — looks at the emitted radiance)."""
    frames = []
    for scale in (4.0, 16.0):
        b = scenes.SceneBuilder()
        red = b.matte(b.spectrum_srgb_nonlinear(0.75, 0.25, 0.25))
        white = b.matte(b.spectrum_srgb_nonlinear(0.75, 0.75, 0.75))
        b.add_quad([(-1.5, 0, 2.55), (-1.5, 0, -2.55), (-1.5, 2.5, -2.55), (-1.5, 2.5, 2.55)], (1, 0, 0), (0, 0, -1), red)
        b.add_quad([(-1.5, 0, 2.55), (1.5, 0, 2.55), (1.5, 0, -2.55), (-1.5, 0, -2.55)], (0, 1, 0), (1, 0, 0), white)
        b.add_quad([(-1.5, 0, -2.55), (1.5, 0, -2.55), (1.5, 2.5, -2.55), (-1.5, 2.5, -2.55)], (0, 0, 1), (1, 0, 0), white)
        light = b.matte(b.spectrum_srgb_nonlinear(0.9, 0.9, 0.9), emittance=b.spectrum_d65(scale, scenes.D65_RGB))
        b.add_quad([(-0.5, 2.499, -0.5), (0.5, 2.499, -0.5), (0.5, 2.499, 0.5), (-0.5, 2.499, 0.5)], (0, -1, 0), (1, 0, 0), light)
        mirror = b.metal(b.spectrum_grey(1.0), b.spectrum_ior("Aluminium", 0, scenes.ALUMINIUM_ETA_RGB), b.spectrum_ior("Aluminium", 1, scenes.ALUMINIUM_K_RGB))
        b.add_uv_sphere(32, 16, mirror, scenes._translate(0.2, 0, -0.6) @ scenes._scale(0.6) @ scenes._translate(0, 1, 0))
        sc = b.build(scenes.cornell_camera(1280 / 720))
        c = Context(mode=abi.MODE_RGB)
        frames.append(c.render_image(sc, ob.settings(1280, 720), 8))
        assert c.counters().samples == 1280 * 720 * 8
        c.close()
    assert frames[0].sum() > 0
    assert_bit_equal(frames[0] * np.float32(4.0), frames[1], "16x light vs 4 * (4x light)")


def test_eight_tile_shards_assemble_the_full_size_frame_exactly():
    """The multi-GPU decomposition at BASELINE's image size, rehearsed on one device: the eight shards a node's ranks would
    render (8x8 tiles, t % 8 == rank) have disjoint supports, and their sum — what the RCCL reduce computes — equals the
    unsharded frame bit for bit (per-(pixel, sample) seeding: a pixel's samples do not depend on which rank owns it)."""
    sc = scenes.cornell_box_spheres(1280 / 720, 24, 12, "matte")
    st = ob.settings(1280, 720)
    # (the sensor adds a pixel's samples in pass order whatever the slot count, so the automatic count — which grows with the
    # world size — gives the same bits: below)
    c = Context(mode=abi.MODE_RGB, stripes=4)
    try:
        c.upload_scene(sc)
        c.render_begin(st)
        c.render(0, 16)
        full = c.read_framebuffer()
        total = np.zeros_like(full)
        covered = np.zeros(full.shape[:2], bool)
        for rank in range(8):
            c.render_begin(st, shard=(rank, 8))
            c.render(0, 16)
            part = c.read_framebuffer()
            support = (part != 0).any(axis=2)
            assert not (covered & support).any(), "shards overlap"
            covered |= support
            total += part
            assert c.counters().samples == 16 * sum(1 for ty in range(90) for tx in range(160) if (ty * 160 + tx) % 8 == rank) * 64
    finally:
        c.close()
    assert_bit_equal(total, full, "sum of 8 shards vs full frame")
    # automatic stripe counts (8 for the full frame, 64 for an eighth of it): same samples, same sums
    a = Context(mode=abi.MODE_RGB)
    try:
        a.upload_scene(sc)
        a.render_begin(st, shard=(3, 8))
        a.render(0, 16)
        part_auto = a.read_framebuffer()
        a.render_begin(st)
        a.render(0, 16)
        full_auto = a.read_framebuffer()
    finally:
        a.close()
    mask = (part_auto != 0).any(axis=2)
    assert_bit_equal(part_auto[mask], full_auto[mask], "a shard with its automatic slot count vs the full frame with its own")
    assert_bit_equal(full_auto, full, "automatic vs fixed slot count")


def test_work_queues_are_reproducible_and_complete(oracle_rgb):
    """64 slots per pixel drawing samples from the wave queues: every (pixel, pass) is rendered exactly once (sample count; the
    image bit-identical to the one-slot-per-pixel frame and equal to the oracle's), and a render continued in a second call — with
    fewer passes than slots per pixel in the first — continues the sensor's sums exactly."""
    sc = scenes.cornell_box_spheres(1.0, 16, 8, "glass")
    st = ob.settings(48, 40, seed=9)
    want, _ = oracle_rgb.scene(sc).render(st, 96)
    frames = []
    for _ in range(2):
        c = Context(stripes=64)
        frames.append(c.render_image(sc, st, 96))
        assert c.counters().samples == 48 * 40 * 96
        c.close()
    assert_bit_equal(frames[0], frames[1], "two runs with 64 stripes")
    assert np.allclose(frames[0], want, rtol=2e-6, atol=1e-9)
    c = Context(stripes=1)
    one = c.render_image(sc, st, 96)
    c.close()
    assert_bit_equal(frames[0], one, "64 stripes vs 1")
    c = Context(stripes=64)
    c.upload_scene(sc)
    c.render_begin(st)
    c.render(0, 40)            # fewer passes than slots per pixel: some queues are empty
    c.render(40, 56)
    two = c.read_framebuffer()
    assert c.counters().samples == 48 * 40 * 96      # sample totals run from slrhip_render_begin
    c.close()
    assert_bit_equal(two, one, "a render continued in a second call vs one call")


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [abi.MODE_RGB, abi.MODE_SPECTRAL])
def test_tail_kernel_gives_the_frame_of_the_wavefront_iterations(mode):
    """Once few slots are live the rest of a render call is ONE launch (k_tail: each remaining slot taken to its end by one lane,
    through the same logicSlot / writeResult / startSample code as k_shade and the one-lane-per-ray traversal).
    On request with a fixed slot count (SLRHIP_FLAG_TAIL_KERNEL).  Against the pure wavefront schedule: the same samples (sample
    and ray counts equal) and the same frame bit for bit at any slot count.  Also for a render continued in a second call and
    more slots per pixel than passes; and reproducible."""
    if os.environ.get("SLRHIP_TAIL_SLOTS") == "0":
        pytest.skip("the tail kernel is switched off in this environment")
    sc = scenes.cornell_box_spheres(1.0, 16, 8, "glass")
    st = ob.settings(64, 48, seed=21)

    def run(flags, stripes, calls):
        c = Context(mode=mode, stripes=stripes, flags=flags | abi.FLAG_TIME_KERNELS)
        c.upload_scene(sc)
        c.render_begin(st)
        for begin, count in calls:
            c.render(begin, count)
        fb = c.read_framebuffer()
        ctr, prof = c.counters(), c.profile()
        c.close()
        return fb, (int(ctr.samples), int(ctr.extension_rays), int(ctr.shadow_rays)), int(prof.launches[2])

    for stripes, calls, extra in ((1, ((0, 24),), 0), (8, ((0, 40),), 0), (8, ((0, 24), (24, 16)), 0), (64, ((0, 16),), 0)):
        want, counts_w, tails_w = run(extra, stripes, calls)
        got, counts_g, tails_g = run(abi.FLAG_TAIL_KERNEL | extra, stripes, calls)
        again, counts_a, _ = run(abi.FLAG_TAIL_KERNEL | extra, stripes, calls)
        assert tails_w == 0 and tails_g == len(calls), (tails_w, tails_g)      # the tail ran once per render call, and only when allowed
        assert counts_g == counts_w == counts_a, (counts_g, counts_w)
        assert counts_g[0] == 64 * 48 * sum(n for _, n in calls)
        assert_bit_equal(got, again, "two runs with the tail kernel (stripes %d)" % stripes)
        assert_bit_equal(got, want, "tail kernel vs wavefront iterations (stripes %d)" % stripes)


@pytest.mark.gpu
def test_tail_kernel_variants_match_the_wavefront_iterations_bit_for_bit():
    """Every k_tail instantiation against the wavefront schedule with ONE stripe (where the tail changes nothing but the
    schedule, so the frames must be bit-identical): the glossy-lobe kernels (GGX boxes, both modes), the MultiBSDF and the
    texture variants (tables in HBM), Ward / Ashikhmin lobes, and an environment light (paths that end at infinity)."""
    if os.environ.get("SLRHIP_TAIL_SLOTS") == "0":
        pytest.skip("the tail kernel is switched off in this environment")
    cases = [("boxes ggx rgb", scenes.cornell_box_boxes(1.0), abi.MODE_RGB), ("boxes ggx spectral", scenes.cornell_box_boxes(1.0), abi.MODE_SPECTRAL),
             ("multi rgb", scenes.cornell_multi(1.0, 10, 5), abi.MODE_RGB), ("multi spectral", scenes.cornell_multi(1.0, 10, 5), abi.MODE_SPECTRAL),
             ("textured rgb", scenes.cornell_textured(1.0, 10, 5), abi.MODE_RGB), ("textured spectral", scenes.cornell_textured(1.0, 10, 5), abi.MODE_SPECTRAL),
             ("ward", scenes.cornell_lobes("ward", segments=8, rings=4), abi.MODE_RGB),
             ("instanced rgb", scenes.cornell_instanced(1.0, 10, 5), abi.MODE_RGB), ("instanced spectral", scenes.cornell_instanced(1.0, 10, 5), abi.MODE_SPECTRAL),
             ("environment light", scenes.ibl_test_scene(1.0, (64, 32), 8, 4), abi.MODE_RGB)]
    st = ob.settings(48, 36, seed=33)
    for name, sc, mode in cases:
        out = []
        for flags in (0, abi.FLAG_TAIL_KERNEL):
            c = Context(mode=mode, stripes=1, flags=flags | abi.FLAG_TIME_KERNELS)
            fb = c.render_image(sc, st, 6)
            ctr, prof = c.counters(), c.profile()
            out.append((fb, (int(ctr.samples), int(ctr.extension_rays), int(ctr.shadow_rays)), int(prof.launches[2])))
            c.close()
        assert out[0][2] == 0 and out[1][2] == 1, (name, out[0][2], out[1][2])
        assert out[0][1] == out[1][1], (name, out[0][1], out[1][1])
        assert_bit_equal(out[1][0], out[0][0], "tail kernel vs wavefront iterations, one stripe: " + name)
        assert out[0][0].sum() > 0, name


@pytest.mark.gpu
def test_graph_replay_and_event_timed_launch_loops_give_the_same_frame():
    """slrhip_render has two launch loops: blocks of 16 iterations captured once into a hipGraph and replayed (>= 2^18 slots, no
    kernel timing) and plain launches bracketed by HIP events (SLRHIP_FLAG_TIME_KERNELS, what bench.py times).  Same kernels, same
    order, the same tail-mode decision (it is taken on the device) => the same frame bit for bit, the same counters — with the
    automatic stripe count (tail kernel on) and with a fixed one."""
    sc = scenes.cornell_box_spheres(4.0 / 3.0, 16, 8, "matte")
    st = ob.settings(640, 480, seed=77)
    for stripes in (0, 2):
        out = []
        for flags in (0, abi.FLAG_TIME_KERNELS):
            c = Context(stripes=stripes, flags=flags)
            fb = c.render_image(sc, st, 48)
            ctr = c.counters()
            out.append((fb, (int(ctr.samples), int(ctr.extension_rays), int(ctr.shadow_rays), int(ctr.iterations))))
            c.close()
        assert out[0][1] == out[1][1], (stripes, out[0][1], out[1][1])
        assert out[0][1][0] == 640 * 480 * 48
        assert_bit_equal(out[0][0], out[1][0], "hipGraph replay vs event-timed launches (stripes %d)" % stripes)
