"""The libSLR-side adapter (slr_amd/csrc/libslr_adapter/HIPPathTracingRenderer.{h,cpp}: the Renderer subclass a libSLR maintainer
would add, compiled against the reference's headers into oracle/_ref by oracle/ref_build).

CPU (where oracle/_ref exists): a libSLR Scene built from a flat description D through the reference's C++ API
(ref_shim.cpp: SurfaceObjectAggregate, SingleSurfaceObject, Triangle, materials, textures, spectra, PerspectiveCamera) is
flattened again by the adapter's flattenScene(const SLR::Scene&) — the result must be D: same vertices, same triangles in the
same order, the same material / spectrum CONTENT per triangle, the same camera.
GPU: HIPPathTracingRenderer(spp).render(scene, settings) called through the reference's Renderer vtable renders the reference's
own Scene object on the HIP path and leaves the frame in the camera's ImageSensor; it must equal the frame the flat scene
renders directly, and the oracle's."""
import ctypes as C
import os

import numpy as np
import pytest

from helpers import assert_bit_equal
from oracle import binding as ob
from slr_amd import abi, binding, scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ref(mode):
    lib = ob.load("ref_rgb" if mode == abi.MODE_RGB else "ref_spectral")
    if lib is None or not hasattr(lib.lib, "slr_ref_flatten"):
        pytest.skip("oracle/_ref with the adapter not built (no /root/reference on this machine)")
    lib.lib.slr_ref_flatten.restype = C.c_void_p
    lib.lib.slr_ref_flatten.argtypes = [C.c_void_p, C.c_char_p]
    lib.lib.slr_ref_flat_desc.argtypes = [C.c_void_p, C.POINTER(abi.SceneDesc)]
    lib.lib.slr_ref_flat_free.argtypes = [C.c_void_p]
    lib.lib.slr_ref_flat_error.restype = C.c_char_p
    lib.lib.slr_ref_render_hip.argtypes = [C.c_void_p, C.POINTER(abi.RenderSettings), C.c_uint32, C.c_int, C.c_char_p, C.c_void_p]
    return lib


def _array(ptr, count, dtype):
    if count == 0:
        return np.zeros(0, dtype)
    buf = (C.c_char * (count * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype, count=count).copy()


def _flatten(lib, ref_scene):
    h = lib.lib.slr_ref_flatten(ref_scene.handle, binding.LIB_PATH.encode())
    if not h:
        raise RuntimeError(lib.lib.slr_ref_flat_error().decode())
    d = abi.SceneDesc()
    lib.lib.slr_ref_flat_desc(h, C.byref(d))
    out = dict(vertices=_array(d.vertices, d.num_vertices, abi.vertex_dtype), triangles=_array(d.triangles, d.num_triangles, abi.triangle_dtype),
               materials=_array(d.materials, d.num_materials, abi.material_dtype), spectra=_array(d.spectra, d.num_spectra, abi.spectrum_dtype),
               spectrum_data=_array(d.spectrum_data, d.num_spectrum_data, np.float32), camera=bytes(d.camera), has_tables=bool(d.upsampling),
               textures=_array(d.textures, d.num_textures, abi.texture_dtype), instances=_array(d.instances, d.num_instances, abi.instance_dtype))
    lib.lib.slr_ref_flat_free(h)
    return out


def _spectrum_content(spectra, data, idx, mode):
    """What a spectrum IS, independent of its index and of where its payload sits."""
    if idx < 0:
        return None
    s = spectra[idx]
    if mode == abi.MODE_RGB:
        return ("rgb",) + tuple(np.float32(v).tobytes() for v in s["rgb"])
    kind = int(s["kind"])
    n = int(s["num_samples"])
    size = {abi.SPEC_UPSAMPLED: 4 + 4 * n, abi.SPEC_REGULAR: n, abi.SPEC_IRREGULAR: 2 * n}[kind]
    payload = data[int(s["data_offset"]):int(s["data_offset"]) + size].tobytes()
    head = (kind, n, int(s["reserved"])) + tuple(np.float32(s[k]).tobytes() for k in ("u", "v", "scale", "lambda_min", "lambda_max"))
    return head + (payload,)


def _texture_content(sc, t_idx, mode):
    t = sc["textures"][t_idx]
    head = (int(t["kind"]), t["offset"].tobytes(), t["scale"].tobytes())
    if int(t["kind"]) == abi.TEX_CHECKER_SPECTRUM:
        return head + tuple(_spectrum_content(sc["spectra"], sc["spectrum_data"], int(i), mode) for i in t["spectrum"])
    return head + (t["value"].tobytes(),)


def _slot_content(sc, slot, mode):
    """A material's spectrum slot: a constant spectrum or (SLRHIP_TEXTURE_REF) a checkerboard of two."""
    return _texture_content(sc, -2 - slot, mode) if slot < -1 else _spectrum_content(sc["spectra"], sc["spectrum_data"], slot, mode)


def _material_content(sc, m_idx, mode):
    m = sc["materials"][m_idx]
    t = int(m["type"])
    maps = tuple(_texture_content(sc, k - 1, mode) if k else None for k in (int(m["reserved"]) & 0xFFFF, int(m["reserved"]) >> 16))
    return maps + _lobe_content(sc, m_idx, mode)


def _lobe_content(sc, m_idx, mode):
    m = sc["materials"][m_idx]
    t = int(m["type"])
    emit = _spectrum_content(sc["spectra"], sc["spectrum_data"], int(m["emittance"]), mode)
    scalars = (np.float32(m["param"]).tobytes(), np.float32(m["param2"]).tobytes())
    if t == abi.MAT_MULTI:
        return (t, scalars, int(m["spectrum"][2]), _lobe_content(sc, int(m["spectrum"][0]), mode), _lobe_content(sc, int(m["spectrum"][1]), mode), emit)
    # the spectra a lobe reads (include/slrhip.h); unused slots are not compared
    used = {abi.MAT_MATTE: (0,), abi.MAT_METAL: (0, 1, 2), abi.MAT_GLASS: (0, 1, 2), abi.MAT_MF_METAL: (1, 2), abi.MAT_MF_GLASS: (1, 2),
            abi.MAT_WARD: (0,), abi.MAT_ASHIKHMIN: (0, 1)}[t]
    if t == abi.MAT_MATTE and float(m["param"]) < 0:
        scalars = (b"lambert",)
    elif t in (abi.MAT_METAL, abi.MAT_GLASS):
        scalars = ()
    elif t in (abi.MAT_MF_METAL, abi.MAT_MF_GLASS, abi.MAT_MATTE):
        scalars = scalars[:1]
    return (t, scalars, tuple(_slot_content(sc, int(m["spectrum"][k]), mode) for k in used), emit)


def _as_dict(scene):
    return dict(vertices=scene.vertices, triangles=scene.triangles, materials=scene.materials, spectra=scene.spectra, spectrum_data=scene.spectrum_data,
                textures=scene.textures)


SCENES = {
    "cornell_glass": lambda: scenes.cornell_box_spheres(4.0 / 3.0, 12, 6, "glass"),
    "boxes_ggx": lambda: scenes.cornell_box_boxes(1.0),
    "material_zoo": lambda: scenes.material_zoo()[0],
    "cornell_multi": lambda: scenes.cornell_multi(1.0, 10, 5),
    "lobes_ward": lambda: scenes.cornell_lobes("ward", segments=8, rings=4),
    "textured": lambda: scenes.cornell_textured(1.0, 10, 5),          # checkerboard slots, BumpSingleSurfaceObject, alpha texture
    "instanced": lambda: scenes.cornell_instanced(1.0, 8, 4),         # TransformedSurfaceObjects over two mesh aggregates
}


@pytest.mark.parametrize("mode", [abi.MODE_RGB, abi.MODE_SPECTRAL])
@pytest.mark.parametrize("name", sorted(SCENES))
def test_flatten_gives_back_the_scene_the_reference_objects_were_built_from(name, mode):
    lib = _ref(mode)
    sc = SCENES[name]()
    got = _flatten(lib, lib.scene(sc))
    want = _as_dict(sc)
    # same triangles in the same order, corner by corner (a vertex no triangle uses does not exist on the libSLR side, so the
    # vertex arrays are compared through the triangles' indices)
    assert len(got["triangles"]) == len(want["triangles"])
    assert got["vertices"][got["triangles"]["v"]].tobytes() == want["vertices"][want["triangles"]["v"]].tobytes()
    used = np.unique(want["triangles"]["v"])
    assert len(got["vertices"]) == len(used) and got["vertices"].tobytes() == want["vertices"][used].tobytes()
    # materials and spectra are re-indexed in order of first use: compare what each triangle's material IS
    cache_g, cache_w = {}, {}
    for tg, tw in zip(got["triangles"]["material"], want["triangles"]["material"]):
        a = cache_g.setdefault(int(tg), _material_content(got, int(tg), mode))
        b = cache_w.setdefault(int(tw), _material_content(want, int(tw), mode))
        assert a == b, (name, int(tg), int(tw))
    # every MULTI component precedes its user, as slrhip_upload_scene requires
    for i, m in enumerate(got["materials"]):
        if int(m["type"]) == abi.MAT_MULTI:
            assert 0 <= m["spectrum"][0] < i and 0 <= m["spectrum"][1] < i
    # instances: the same placements of the same triangle ranges (their order is the order of the objects in memory)
    assert sorted(r.tobytes() for r in got["instances"]) == sorted(r.tobytes() for r in sc.instances)
    cam = abi.Camera.from_buffer_copy(got["camera"])
    assert bytes(cam.local_to_world) == bytes(sc.camera.local_to_world) and bytes(cam.world_to_local) == bytes(sc.camera.world_to_local)
    for f in ("aspect", "fov_y", "lens_radius", "img_plane_distance", "obj_plane_distance"):
        assert getattr(cam, f) == getattr(sc.camera, f), f
    # the sensitivity comes back resolved (PerspectiveCamera.cpp:23: 1 / (pi r^2) when the scene passes 0)
    r = np.float64(np.float32(sc.camera.lens_radius))
    assert cam.sensitivity == np.float32(np.float32(1.0) / (np.pi * r * r))
    assert got["has_tables"] == (mode == abi.MODE_SPECTRAL)


def test_flatten_refuses_what_is_outside_the_hot_path_loudly():
    lib = _ref(abi.MODE_RGB)
    with pytest.raises(RuntimeError, match="environment sphere"):
        _flatten(lib, lib.scene(scenes.ibl_test_scene(1.0, (64, 32), 8, 4)))


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [abi.MODE_RGB, abi.MODE_SPECTRAL])
@pytest.mark.parametrize("which", ["cornell_glass", "instanced"])
def test_reference_scene_object_renders_on_the_hip_path_through_the_renderer_vtable(mode, which):
    """HostProgram/main.cpp:59 `renderer->render(*scene, settings)` with renderer = HIPPathTracingRenderer: the libSLR Scene
    object (pointer graph, SBVH and all) is flattened, rendered on the GPU and lands in the camera's ImageSensor."""
    from slr_amd import Context
    lib = _ref(mode)
    # "instanced": the libSLR Scene holds TransformedSurfaceObjects over mesh aggregates; the adapter turns them into slrhip_instances
    sc = scenes.cornell_box_spheres(4.0 / 3.0, 16, 8, "glass") if which == "cornell_glass" else scenes.cornell_instanced(4.0 / 3.0, 12, 6, copies=9)
    st = ob.settings(96, 72, seed=99)
    spp = 8
    rs = lib.scene(sc)
    fb = np.zeros((72, 96, rs.components), np.float32)
    assert lib.lib.slr_ref_render_hip(rs.handle, C.byref(st), spp, 0, binding.LIB_PATH.encode(), fb.ctypes.data) == 0
    c = Context(mode=mode)          # same automatic stripe count as the adapter's context ...
    c.upload_scene(sc)
    c.render_begin(st)
    for begin, count in ((0, 1), (1, 1), (2, 2), (4, 4)):      # ... and the same calls: one per export of PathTracingRenderer.cpp:83-94
        c.render(begin, count)
    direct = c.read_framebuffer()
    c.close()
    assert_bit_equal(fb, direct, "through libSLR's Renderer vtable vs the flat scene directly")
    want, _ = ob.load("oracle", mode).scene(sc).render(st, spp)
    assert np.allclose(fb, want, rtol=2e-6, atol=1e-9)
    assert fb.sum() > 0
