"""Generates tests/golden/*.npz from the COMPILED REFERENCE (oracle/_ref, built by
oracle/ref_build from /root/reference).  Run in the container that has the reference:

    python tests/golden/make_golden.py

A fixture is data only: the flat input scene (our own arrays), the render settings and the
reference's outputs (float framebuffers, single-sample radiance, closest hits, RNG draws).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import binding as ob  # noqa: E402
from slr_amd import abi, scenes  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def scene_arrays(sc):
    cam = sc.camera
    env = {}
    if sc.env is not None:
        # texels are binary16-exact (scenes.synthetic_sky): float16 storage is lossless
        assert np.array_equal(sc.env_texels.astype(np.float16).astype(np.float32), sc.env_texels)
        env = dict(env_texels=sc.env_texels.astype(np.float16), env_scale=np.float32(sc.env_scale), env_importance=sc.env_importance)
        if sc.env_uvs is not None:
            assert np.array_equal(sc.env_texels_uvs.astype(np.float16).astype(np.float32), sc.env_texels_uvs)
            env.update(env_texels_uvs=sc.env_texels_uvs.astype(np.float16), env_importance_uvs=sc.env_importance_uvs)
    if len(sc.textures):
        env["textures"] = sc.textures
    if len(sc.texture_texels):
        env["texture_texels"], env["texture_texels_uvs"] = sc.texture_texels, sc.texture_texels_uvs
    if len(sc.instances):
        env["instances"] = sc.instances
    return dict(env, vertices=sc.vertices, triangles=sc.triangles, materials=sc.materials, spectra=sc.spectra,
                spectrum_data=sc.spectrum_data,
                camera=np.array(list(cam.local_to_world) + list(cam.world_to_local) +
                                [cam.aspect, cam.fov_y, cam.lens_radius, cam.img_plane_distance,
                                 cam.obj_plane_distance, cam.sensitivity], dtype=np.float32))


def random_rays(rng, n):
    org = rng.uniform([-1.4, 0.1, -2.4], [1.4, 2.4, 2.4], size=(n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3))
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    rays = np.zeros(n, dtype=ob.ray_dtype)
    rays["org"], rays["dir"] = org, d
    rays["dist_min"] = 1e-4
    rays["dist_max"] = np.inf
    rays["dist_max"][: n // 4] = rng.uniform(0.2, 3.0, size=n // 4).astype(np.float32)
    return rays


def make(name, sc, lib, width, height, spp, serial_spp):
    if lib is None:
        raise SystemExit("reference library for %s not built" % name)
    ref = lib.scene(sc)
    st = ob.settings(width, height)
    fb, _ = ref.render(st, spp, threads=0)
    fb_half, _ = ref.render(st, spp // 2, threads=0)
    rng = np.random.default_rng(1234)
    picks = np.stack([rng.integers(0, width, 64), rng.integers(0, height, 64), rng.integers(0, 8, 64)], axis=1)
    samples = np.stack([ref.sample(st, int(x), int(y), int(p)) for x, y, p in picks])
    rays = random_rays(rng, 512)
    hits = ref.trace(rays)
    st_serial = ob.settings(width // 2, height // 2)
    fb_serial, _ = ref.render_serial(st_serial, serial_spp)
    out = dict(scene_arrays(sc))
    out.update(width=width, height=height, spp=spp, seed=st.rng_seed, framebuffer=fb, framebuffer_half=fb_half,
               sample_picks=picks, sample_values=samples, rays=rays, hits=hits,
               serial_width=width // 2, serial_height=height // 2, serial_spp=serial_spp, serial_framebuffer=fb_serial)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "framebuffer mean", fb.mean(), "hits", int((hits["triangle"] != 0xFFFFFFFF).sum()), "/", len(hits))


def make_procedural(name, generator, args, lib, width, height, spp):
    """A fixture for a scene too large to store (BASELINE configs[4]: the displaced grid): the fixture holds the
    GENERATOR's name and arguments (slr_amd.scenes, deterministic, seeded) plus a checksum of the arrays it produced,
    and the reference's outputs: float framebuffer and 2048 closest hits."""
    import zlib
    if lib is None:
        raise SystemExit("reference library for %s not built" % name)
    sc = getattr(scenes, generator)(*args)
    ref = lib.scene(sc)
    st = ob.settings(width, height)
    fb, _ = ref.render(st, spp, threads=0)
    rng = np.random.default_rng(4321)
    n = 2048
    # rays from above the terrain, pointing down into it and across it
    rays = np.zeros(n, dtype=ob.ray_dtype)
    rays["org"] = rng.uniform([-3.5, 1.2, -3.5], [3.5, 2.8, 3.5], size=(n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)) * [1.0, 0.6, 1.0] - [0.0, 0.7, 0.0]
    rays["dir"] = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    rays["dist_min"] = 0.0
    rays["dist_max"] = np.inf
    hits = ref.trace(rays)
    crc = zlib.crc32(sc.vertices.tobytes()) ^ zlib.crc32(sc.triangles.tobytes())
    np.savez_compressed(os.path.join(HERE, name + ".npz"), generator=generator, generator_args=np.array(args, np.float64),
                        scene_crc=np.uint32(crc), num_triangles=len(sc.triangles), materials=sc.materials, spectra=sc.spectra,
                        spectrum_data=sc.spectrum_data, width=width, height=height, spp=spp,
                        seed=st.rng_seed, framebuffer=fb, rays=rays, hits=hits)
    print(name, "framebuffer mean", fb.mean(), "hits", int((hits["triangle"] != 0xFFFFFFFF).sum()), "/", len(hits))


KAT_WAVELENGTHS = ((0.37, 0.61), (0.0, 0.0), (0.93, 0.9999))    # (offset, uLambda) of createWithEqualOffsets


def make_bsdf_kat(name, lib, zoo=None):
    """Function-level known answers: BSDF::sample / evaluate / evaluatePDF of the reference's own BSDF objects,
    one material per lobe, 192 queries (scenes.bsdf_queries) under three wavelength selections."""
    if lib is None:
        raise SystemExit("reference library for %s not built" % name)
    sc, mats = (zoo or scenes.material_zoo)()
    ref = lib.scene(sc)
    q = scenes.bsdf_queries(192, 2024)
    out = dict(scene_arrays(sc))
    out.update(queries=q, wavelengths=np.array(KAT_WAVELENGTHS, np.float32), material_names=np.array(list(mats)),
               material_indices=np.array(list(mats.values()), np.uint32))
    for mname, m in mats.items():
        out["out_" + mname] = np.stack([ref.bsdf_kat(m, q, off, ul) for off, ul in KAT_WAVELENGTHS])
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, {k: int((out["out_" + k][0][:, 3] != 0).sum()) for k in mats})


def light_queries(n, seed):
    """[n][3] (light selection sample, two position samples) with the corner cases first."""
    u = np.random.default_rng(seed).random((n, 3)).astype(np.float32)
    one = np.float32(1.0) - np.float32(2.0 ** -24)
    u[0] = 0.0
    u[1] = one
    u[2] = (0.5, 0.0, 0.0)
    u[3] = (0.5, one, one)
    u[4] = (one, 0.0, one)
    return u


def make_light_kat(name, lib):
    """Scene::selectLight + Light::sample of the reference on a scene with an environment sphere next to triangle lights,
    and on one with several triangle lights: 512 queries each."""
    if lib is None:
        raise SystemExit("reference library for %s not built" % name)
    out = {}
    q = light_queries(512, 77)
    for tag, sc in (("env", scenes.ibl_test_scene(1.0, (128, 64), 12, 6, area_light=True)), ("zoo", scenes.material_zoo()[0])):
        out.update({tag + "_" + k: v for k, v in scene_arrays(sc).items()})
        out[tag + "_out"] = lib.scene(sc).light_kat(q, 0.3, 0.7)
    out["queries"] = q
    out["wavelengths"] = np.array([0.3, 0.7], np.float32)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "environment picks", int((out["env_out"][:, 0] < 0).sum()), "of", len(q))


def main():
    only = sys.argv[1:]
    if only:
        global make
        _make = make
        make = lambda name, *a: _make(name, *a) if name in only else None
    lib = ob.load("ref_rgb")
    if lib is None:
        raise SystemExit("oracle/_ref is not built: run `make -C oracle ref` where /root/reference exists")
    u, f = lib.rng(abi.DEFAULT_SEED, 64)
    seeds = np.array([abi.DEFAULT_SEED, 0, 1, -1, 123456789, -2147483648], dtype=np.int32)
    kat = {"seeds": seeds}
    for i, s in enumerate(seeds):
        kat["uints_%d" % i], kat["floats_%d" % i] = lib.rng(int(s), 64)
    np.savez_compressed(os.path.join(HERE, "rng_kat.npz"), **kat)
    # ImageSensor::saveImage on a known framebuffer -> BMP bytes (pins slrhip_tonemap_bgr8 / slrhip_save_bmp)
    import tempfile
    rng = np.random.default_rng(7)
    w, h = 37, 21                      # width % 4 != 0 exercises the reference's row padding (ImageSensor.cpp:149)
    fb = (rng.random((h, w, 3)) ** 4 * 3.0).astype(np.float32)
    fb[0, 0] = 0.0
    fb[1, 1] = (1e-4, 2e-4, 3e-4)
    fb[2, 2] = (50.0, 60.0, 70.0)
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "t.bmp")
        ob.ref_save_image(lib, fb, 509.29581, 0.37, path)
        bmp = np.frombuffer(open(path, "rb").read(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "tonemap_bmp.npz"), framebuffer=fb, sensitivity=np.float32(509.29581),
                        scale=np.float32(0.37), bmp=bmp)
    if not only or "bsdf_kat_rgb" in only:
        make_bsdf_kat("bsdf_kat_rgb", lib)
    if not only or "light_kat_rgb" in only:
        make_light_kat("light_kat_rgb", lib)
    make("rgb_tiny_box", scenes.tiny_box(1.0), lib, 32, 32, 8, 2)
    make("rgb_cornell_glass", scenes.cornell_box_spheres(4.0 / 3.0, 16, 8, "glass"), lib, 48, 36, 8, 2)
    make("rgb_oren_nayar", scenes.cornell_lobes("oren_nayar"), lib, 40, 40, 8, 2)
    make("rgb_ggx_metal", scenes.cornell_lobes("ggx_metal"), lib, 40, 40, 8, 2)
    make("rgb_ggx_glass", scenes.cornell_lobes("ggx_glass"), lib, 40, 40, 8, 2)
    make("rgb_ward", scenes.cornell_lobes("ward"), lib, 40, 40, 8, 2)
    make("rgb_ashikhmin", scenes.cornell_lobes("ashikhmin"), lib, 40, 40, 8, 2)
    spec = ob.load("ref_spectral")
    if spec is not None:
        # the spectral build's saveImage (16 storage bins -> getRGB -> tone map -> BMP) on a known framebuffer
        rng = np.random.default_rng(11)
        w, h = 29, 17
        fb16 = (rng.random((h, w, 16)) ** 3 * 0.02).astype(np.float32)
        fb16[0, 0] = 0.0
        fb16[1, 1, :] = 0.0; fb16[1, 1, 2] = 0.5          # a saturated blue bin: negative sRGB components are clamped
        fb16[2, 2] = 5.0
        with tempfile.TemporaryDirectory() as d:
            path = os.path.join(d, "t.bmp")
            ob.ref_save_image(spec, fb16, 509.29581, 0.41, path)
            bmp16 = np.frombuffer(open(path, "rb").read(), dtype=np.uint8)
        np.savez_compressed(os.path.join(HERE, "tonemap_bmp_spectral.npz"), framebuffer=fb16, sensitivity=np.float32(509.29581),
                            scale=np.float32(0.41), bmp=bmp16)
    if not only or "bsdf_kat_spectral" in only:
        make_bsdf_kat("bsdf_kat_spectral", spec)
    if not only or "light_kat_spectral" in only:
        make_light_kat("light_kat_spectral", spec)
    make("spectral_cornell_glass", scenes.cornell_box_spheres(1.0, 12, 6, "glass"), spec, 32, 32, 8, 2)
    make("spectral_cornell_matte", scenes.cornell_box_spheres(1.0, 12, 6, "matte"), spec, 32, 32, 8, 2)
    make("spectral_oren_nayar", scenes.cornell_lobes("oren_nayar", segments=10, rings=5), spec, 32, 32, 8, 2)
    make("spectral_ggx_metal", scenes.cornell_lobes("ggx_metal", segments=10, rings=5), spec, 32, 32, 8, 2)
    make("spectral_ggx_glass", scenes.cornell_lobes("ggx_glass", segments=10, rings=5), spec, 32, 32, 8, 2)
    make("rgb_ibl", scenes.ibl_test_scene(1.0, (128, 64), 12, 6), lib, 40, 40, 8, 2)
    make("rgb_ibl_area", scenes.ibl_test_scene(1.0, (128, 64), 12, 6, area_light=True), lib, 40, 40, 8, 2)
    make("spectral_ashikhmin", scenes.cornell_lobes("ashikhmin", segments=10, rings=5), spec, 32, 32, 8, 2)
    make("spectral_ibl", scenes.ibl_test_scene(1.0, (128, 64), 12, 6, area_light=True), spec, 32, 32, 8, 2)
    make("rgb_cornell_matte", scenes.cornell_box_spheres(4.0 / 3.0, 16, 8, "matte"), lib, 48, 36, 8, 2)
    # MultiBSDF (sum / mix / inverse materials)
    make("rgb_multi", scenes.cornell_multi(1.0, 12, 6), lib, 40, 40, 8, 2)
    make("rgb_multi_libm_free", scenes.cornell_multi(1.0, 12, 6, libm_free=True), lib, 40, 40, 8, 2)
    make("spectral_multi", scenes.cornell_multi(1.0, 10, 5), spec, 32, 32, 8, 2)
    make("spectral_multi_libm_free", scenes.cornell_multi(1.0, 10, 5, libm_free=True), spec, 32, 32, 8, 2)


def main_round2(only):
    """Fixtures added in round 2: BASELINE configs[2] exactly (Cornell_Box_Boxes-shaped, GGX titanium, SPECTRAL build of the
    reference) and configs[4]'s shape (displaced grid, >= 64 Ki nodes on the HIP side, thin lens r = 0.025)."""
    spec = ob.load("ref_spectral")
    lib = ob.load("ref_rgb")
    if not only or "spectral_boxes" in only:
        make("spectral_boxes", scenes.cornell_box_boxes(1.0), spec, 48, 48, 8, 2)
    if not only or "rgb_textured" in only:
        make("rgb_textured", scenes.cornell_textured(1.0, 12, 6), lib, 48, 48, 8, 2)
    if not only or "spectral_textured" in only:
        make("spectral_textured", scenes.cornell_textured(1.0, 10, 5), spec, 40, 40, 8, 2)
    if not only or "upsample_kat" in only:
        # UpsampledContinuousSpectrum(spType, space, e0, e1, e2) of the compiled reference: (u, v, scale) for every colour space
        import ctypes as C
        up = spec.lib.slr_ref_upsample
        up.argtypes = [C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_void_p]
        rng = np.random.default_rng(2718)
        rows = []
        for i in range(768):
            spt, space = int(rng.integers(0, 2)), int(rng.integers(0, 4))
            e = rng.random(3).astype(np.float32)
            if space == 2:
                e[:2] = (0.2 + 0.4 * e[:2]).astype(np.float32)          # xyY: chromaticities inside the gamut
            if i < 8:
                e[:] = [(0, 0, 0), (1, 1, 1), (0.04045, 0.04045, 0.04045), (1, 0, 0), (0, 1, 0), (0, 0, 1), (0.5, 0.5, 0.5), (0.75, 0.25, 0.25)][i]
                space = 1 if i != 0 else 3
            out = np.zeros(3, np.float32)
            assert up(spt, space, float(e[0]), float(e[1]), float(e[2]), out.ctypes.data) == 0
            rows.append([spt, space, e[0], e[1], e[2], out[0], out[1], out[2]])
        np.savez_compressed(os.path.join(HERE, "upsample_kat.npz"), rows=np.array(rows, np.float32))
        print("upsample_kat", len(rows))
    if not only or "rgb_conversion_kat" in only:
        # the libSLR-held pieces of the RGB build's Spectrum::create: integralCMF and XYZ -> sRGB / sRGB_E on 256 XYZ triples
        import ctypes as C
        f = lib.lib.slr_ref_rgb_pieces
        f.argtypes = [C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        rng = np.random.default_rng(31415)
        xyz = (rng.random((256, 3)) * [1.2, 1.0, 1.4]).astype(np.float32)
        xyz[0] = (0.95047, 1.0, 1.08883)
        xyz[1] = 0.0
        out = {}
        integral = np.zeros(1, np.float32)
        for tag, tp in (("illuminant", 1), ("reflectance", 0)):
            rgb = np.zeros_like(xyz)
            assert f(tp, xyz.ctypes.data, len(xyz), rgb.ctypes.data, integral.ctypes.data) == 0
            out["rgb_" + tag] = rgb
        np.savez_compressed(os.path.join(HERE, "rgb_conversion_kat.npz"), xyz=xyz, integral_cmf=integral, **out)
        print("rgb_conversion_kat integralCMF", integral[0])
    if not only or "rgb_grid400" in only:
        make_procedural("rgb_grid400", "displaced_grid", (400, 16.0 / 9.0), lib, 160, 90, 8)


def main_round3(only):
    """Fixtures added in round 3: nested MultiBSDFs (a summed / mixed material whose components are summed / mixed materials,
    SummedSurfaceMaterial.cpp:13-20 over MultiBSDF.cpp:20-59), function level and whole frames, both builds of the reference."""
    spec = ob.load("ref_spectral")
    lib = ob.load("ref_rgb")
    if not only or "bsdf_kat_nested_rgb" in only:
        make_bsdf_kat("bsdf_kat_nested_rgb", lib, scenes.material_zoo_nested)
    if not only or "bsdf_kat_nested_spectral" in only:
        make_bsdf_kat("bsdf_kat_nested_spectral", spec, scenes.material_zoo_nested)
    if not only or "rgb_multi_nested" in only:
        make("rgb_multi_nested", scenes.cornell_multi_nested(1.0, 12, 6), lib, 40, 40, 8, 2)
    if not only or "spectral_multi_nested" in only:
        make("spectral_multi_nested", scenes.cornell_multi_nested(1.0, 10, 5), spec, 32, 32, 8, 2)
    # image textures in material slots: the texel addressing is the shim's restatement (ImageSpectrumTexture needs OpenEXR half),
    # everything after the texel is the compiled reference's
    if not only or "rgb_image_textured" in only:
        make("rgb_image_textured", scenes.cornell_image_textured(1.0, 12, 6), lib, 48, 48, 8, 2)
    if not only or "spectral_image_textured" in only:
        make("spectral_image_textured", scenes.cornell_image_textured(1.0, 10, 5), spec, 40, 40, 8, 2)
    # instanced meshes: TransformedSurfaceObjects over mesh aggregates (Core/SurfaceObject.cpp:303-392), the reference's own classes
    if not only or "rgb_instanced" in only:
        make("rgb_instanced", scenes.cornell_instanced(1.0, 12, 6), lib, 48, 48, 8, 2)
    if not only or "spectral_instanced" in only:
        make("spectral_instanced", scenes.cornell_instanced(1.0, 10, 5), spec, 40, 40, 8, 2)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "round3":
        main_round3(sys.argv[2:])
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "round2":
        main_round2(sys.argv[2:])
        sys.exit(0)
    main()
