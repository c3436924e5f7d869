"""The scene-language loader (slr_amd/scene_language.py): grammar, argument binding, built-ins, flattening.
Scene scripts are generated here from this repository's own builders, so nothing is read from the reference at test time;
where /root/reference exists its own Cornell_Box_Spheres.txt is loaded as well and must give the same geometry."""
import os

import numpy as np
import pytest

from slr_amd import scene_language as sl
from slr_amd import scenes


def quad_script(name, corners, normal, tangent, material_expr):
    uv = [(0, 0), (1, 0), (1, 1), (0, 1)]
    verts = ",\n    ".join("((%r, %r, %r), (%r, %r, %r), (%r, %r, %r), (%r, %r))" % (tuple(c) + tuple(normal) + tuple(tangent) + uv[i])
                           for i, c in enumerate(corners))
    return "%s = createMesh(\n  (\n    %s\n  ),\n  (\n    (%s, ((0, 1, 2), (0, 2, 3))),\n  )\n);\naddChild(box, %s);\n" % (name, verts, material_expr, name)


def cornell_script(right="glass"):
    """The same numbers scenes.cornell_box_spheres() uses, written as a scene script."""
    s = 'setRenderer("method": "PT", ("samples": 64,));\nsetRenderSettings("width": 320, "height": 240, "rngSeed": 77);\n'
    s += "box = createNode();\nsetTransform(box, translate(0, 0, 0));\n"
    s += "function diffuse(r, g, b) { return createSurfaceMaterial(\"matte\", (SpectrumTexture(Spectrum(r, g, b)),)); }\n"
    s += "white = diffuse(0.75, 0.75, 0.75);\n"
    s += quad_script("leftWall", [(-1.5, 0, 2.55), (-1.5, 0, -2.55), (-1.5, 2.5, -2.55), (-1.5, 2.5, 2.55)], (1, 0, 0), (0, 0, -1), "diffuse(0.75, 0.25, 0.25)")
    s += quad_script("rightWall", [(1.5, 0, -2.55), (1.5, 0, 2.55), (1.5, 2.5, 2.55), (1.5, 2.5, -2.55)], (-1, 0, 0), (0, 0, 1), "diffuse(0.25, 0.25, 0.75)")
    s += quad_script("floor", [(-1.5, 0, 2.55), (1.5, 0, 2.55), (1.5, 0, -2.55), (-1.5, 0, -2.55)], (0, 1, 0), (1, 0, 0), "white")
    s += quad_script("back", [(-1.5, 0, -2.55), (1.5, 0, -2.55), (1.5, 2.5, -2.55), (-1.5, 2.5, -2.55)], (0, 0, 1), (1, 0, 0), "white")
    s += quad_script("ceiling", [(-1.5, 2.5, -2.55), (1.5, 2.5, -2.55), (1.5, 2.5, 2.55), (-1.5, 2.5, 2.55)], (0, -1, 0), (1, 0, 0), "white")
    s += ('lamp = createSurfaceMaterial("emitter", (diffuse(0.9, 0.9, 0.9), '
          'createEmitterSurfaceProperty("diffuse", (SpectrumTexture(Spectrum("ID": "D65") * 4),))));\n')
    s += quad_script("light", [(-0.5, 2.499, -0.5), (0.5, 2.499, -0.5), (0.5, 2.499, 0.5), (-0.5, 2.499, 0.5)], (0, -1, 0), (1, 0, 0), "lamp")
    s += "addChild(root, box);\n"
    s += ('function mirror(name, attrs) {\n  eta = SpectrumTexture(Spectrum("ID": "Aluminium", 0));\n  k = SpectrumTexture(Spectrum("ID": "Aluminium", 1));\n'
          '  return createSurfaceMaterial("metal", (SpectrumTexture(Spectrum("Reflectance", 1.0)), eta, k));\n}\n')
    s += 'ball = load3DModel("models/sphere.assbin", mirror);\nsetTransform(ball, translate(-0.7, 0, -1.05) * scale(0.5) * translate(0, 1, 0));\naddChild(box, ball);\n'
    if right == "glass":
        s += ('function crystal(name, attrs) {\n  return createSurfaceMaterial("glass", (SpectrumTexture(Spectrum("Reflectance", 0.999)), '
              'SpectrumTexture(Spectrum("ID": "Air", 0)), SpectrumTexture(Spectrum("ID": "Glass_BK7", 0))));\n}\n')
    else:
        s += 'function crystal(name, attrs) { return diffuse(0.75, 0.75, 0.25); }\n'
    s += 'ball2 = load3DModel("models/sphere.assbin", crystal);\nsetTransform(ball2, translate(0.7, 0, 0) * scale(0.5) * translate(0, 1, 0));\naddChild(box, ball2);\n'
    s += ('eye = createNode();\naddChild(eye, createPerspectiveCamera("aspect": 4.0 / 3.0, "fovY": 0.4807705238, "radius": 0.025, "imgDist": 1.0, "objDist": 6.3));\n'
          'setTransform(eye, translate(0.0, 1.689714, 6.70284) * rotateY(3.1415926536) * rotateX(0.0563936));\naddChild(root, eye);\n')
    return s


def resolved_materials(sc):
    """Per triangle: (type, param, rgb of its three spectra, rgb of its emittance) — independent of how materials are shared."""
    out = []
    for t in sc.triangles:
        m = sc.materials[t["material"]]
        rgbs = tuple(tuple(sc.spectra[i]["rgb"]) if i >= 0 else None for i in list(m["spectrum"]) + [m["emittance"]])
        out.append((int(m["type"]), float(m["param"])) + rgbs)
    return out


@pytest.mark.parametrize("right", ["glass", "matte"])
def test_script_builds_the_same_flat_scene_as_the_builders(right):
    got, settings, renderer = sl.load_scene(cornell_script(right), sphere_tessellation=(16, 8))
    want = scenes.cornell_box_spheres(4.0 / 3.0, 16, 8, right)
    assert renderer == {"method": "PT", "samples": 64}
    assert (settings["width"], settings["height"], settings["rngSeed"]) == (320, 240, 77)
    for field in ("position", "normal", "tangent", "texcoord"):
        assert np.array_equal(got.vertices[field], want.vertices[field]), field
    assert np.array_equal(got.triangles["v"], want.triangles["v"])
    assert resolved_materials(got) == resolved_materials(want)
    assert list(got.camera.local_to_world) == list(want.camera.local_to_world)
    for f in ("aspect", "fov_y", "lens_radius", "img_plane_distance", "obj_plane_distance", "sensitivity"):
        assert getattr(got.camera, f) == getattr(want.camera, f)


def test_language_features():
    it = sl.Interpreter()
    it.run('''
        function fact(n) { if (n <= 1) return 1; return n * fact(n - 1); }
        total = 0;
        for (i = 0; i < 5; ++i) { total += i * 2; }
        t = (1, "key": 7, 3.5);
        t = addItem(t, "extra", 9);
        a = fact(5);
        b = t["key"] + t[1];          // named element + second unnamed element
        c = 7 / 2;                     // integer division like the reference's C++
        d = 7.0 / 2;
        e = !(1 > 2) && (3 != 4);
        v = cross(Vector(1, 0, 0), Vector(0, 1, 0));
        z = getZ(v);
        n = numElements(t);
        m = translate(1, 2, 3) * scale(2);
        s = Spectrum("type": "Reflectance", 0.025);
        function withDefault(x, y = 10) { return x + y; }
        w = withDefault(1) + withDefault(1, "y": 2) + withDefault("x": 5, 5);
    ''')
    g = it.globals
    assert (g["a"], g["total"], g["b"], g["c"], g["d"], g["e"], g["z"], g["n"]) == (120, 20, 7 + 3.5, 3, 3.5, True, 1.0, 4)
    assert g["m"][:3, 3].tolist() == [1, 2, 3] and g["m"][0, 0] == 2
    assert g["s"].ctor == "grey" and g["s"].args == ("Reflectance", 0.025)
    assert g["w"] == 11 + 3 + 10


def test_argument_binding_follows_the_reference():
    """SceneParser.cpp:399-453: named first, then each positional to the first unassigned parameter it converts to."""
    it = sl.Interpreter()
    it.run('a = Spectrum(0.1, 0.2, 0.3); b = Spectrum("Reflectance", 0.5); c = Spectrum("Illuminant", "sRGB", 1, 2, 3);'
           'd = Spectrum("ID": "D65") * 4; e = Spectrum("ID": "Aluminium", 1);')
    g = it.globals
    assert g["a"].ctor == "tristimulus" and g["a"].args == ("Reflectance", "sRGB", 0.1, 0.2, 0.3)
    assert g["b"].ctor == "grey"
    assert g["c"].args[:2] == ("Illuminant", "sRGB")
    assert g["d"].ctor == "library" and g["d"].scale == 4
    assert g["e"].args == ("Aluminium", 1)
    with pytest.raises(sl.SceneLanguageError):
        it.run('Spectrum("nonsense": 1);')


def test_glossy_lobes_and_emitters_on_any_base():
    it = sl.Interpreter()
    it.run('''
        lamp = createSurfaceMaterial("emitter", (createSurfaceMaterial("metal", (SpectrumTexture(Spectrum("Reflectance", 1.0)),
               SpectrumTexture(Spectrum("ID": "Aluminium", 0)), SpectrumTexture(Spectrum("ID": "Aluminium", 1)))),
               createEmitterSurfaceProperty("diffuse", (SpectrumTexture(Spectrum("ID": "D65")),))));
        w = createSurfaceMaterial("Ward", (SpectrumTexture(Spectrum(0.5, 0.5, 0.5)), FloatTexture(0.1), FloatTexture(0.2)));
        a = createSurfaceMaterial("Ashikhmin", (SpectrumTexture(Spectrum(0.5, 0.2, 0.2)), SpectrumTexture(Spectrum("Reflectance", 0.05)),
                                                FloatTexture(100), FloatTexture(50)));
        q = createMesh((((0, 0, 0), (0, 1, 0), (1, 0, 0), (0, 0)), ((1, 0, 0), (0, 1, 0), (1, 0, 0), (1, 0)), ((1, 0, 1), (0, 1, 0), (1, 0, 0), (1, 1))),
                       ((lamp, ((0, 1, 2),)), (w, ((0, 2, 1),)), (a, ((1, 2, 0),))));
        addChild(root, q);
        c = createNode(); addChild(c, createPerspectiveCamera()); addChild(root, c);
    ''')
    sc = it.build()
    from slr_amd import abi
    assert list(sc.materials["type"]) == [abi.MAT_METAL, abi.MAT_WARD, abi.MAT_ASHIKHMIN]
    assert sc.materials["emittance"][0] >= 0 and (sc.materials["emittance"][1:] == -1).all()
    assert (float(sc.materials["param"][1]), float(sc.materials["param2"][1])) == (np.float32(0.1), np.float32(0.2))
    rs, rd = sc.materials["spectrum"][2][:2]                    # Ashikhmin: spectrum = {Rs, Rd}; the script passes (Rd, Rs, nx, ny)
    assert tuple(sc.spectra[rs]["rgb"]) == (np.float32(0.05),) * 3 and sc.spectra[rd]["rgb"][0] > sc.spectra[rd]["rgb"][1]
    assert (float(sc.materials["param"][2]), float(sc.materials["param2"][2])) == (100.0, 50.0)


@pytest.mark.parametrize("snippet", ['x = Image2D("images/a.exr");', 'setEnvironment("images/sky.exr", 4);',
                                     'n = load3DModel("models/teapot.assbin");'])
def test_missing_assets_and_lobes_are_refused_loudly(snippet):
    with pytest.raises(sl.UnsupportedFeature):
        sl.Interpreter().run(snippet)


MULTI_SCRIPT = '''
    function leaf(R, T) {
        r = createSurfaceMaterial("matte", (SpectrumTexture(R),));
        tBase = createSurfaceMaterial("matte", (SpectrumTexture(T),));
        t = createSurfaceMaterial("inverse", (tBase,));
        return createSurfaceMaterial("sum", (r, t));
    }
    green = leaf(Spectrum(0.5, 0.5, 0.5), Spectrum(0.2, 0.6, 0.3));
    ti = createSurfaceMaterial("microfacet metal", (SpectrumTexture(Spectrum("ID": "Titanium", 0)), SpectrumTexture(Spectrum("ID": "Titanium", 1)), FloatTexture(0.3)));
    coated = createSurfaceMaterial("mix", (ti, createSurfaceMaterial("matte", (SpectrumTexture(Spectrum(0.8, 0.45, 0.15)),)), FloatTexture(0.3)));
    %s
    q = createMesh((((0, 0, 0), (0, 1, 0), (1, 0, 0), (0, 0)), ((1, 0, 0), (0, 1, 0), (1, 0, 0), (1, 0)), ((1, 0, 1), (0, 1, 0), (1, 0, 0), (1, 1))),
                   ((green, ((0, 1, 2),)), (%s, ((0, 2, 1),))));
    addChild(root, q);
    c = createNode(); addChild(c, createPerspectiveCamera()); addChild(root, c);
'''


def test_summed_mixed_and_inverse_materials_become_multi_records():
    """TestScenes/RTC3.txt:13-18 builds its leaves as sum(matte, inverse(matte)): API.cpp:583-636."""
    from slr_amd import abi
    it = sl.Interpreter()
    it.run(MULTI_SCRIPT % ("", "coated"))
    sc = it.build()
    m = sc.materials
    assert list(m["type"]) == [abi.MAT_MATTE, abi.MAT_MATTE, abi.MAT_MULTI, abi.MAT_MF_METAL, abi.MAT_MATTE, abi.MAT_MULTI]
    assert tuple(m["spectrum"][2]) == (0, 1, abi.MULTI_INVERSE_1) and (float(m["param"][2]), float(m["param2"][2])) == (1.0, 1.0)
    assert tuple(m["spectrum"][5]) == (3, 4, 0)
    assert (m["param"][5], m["param2"][5]) == (np.float32(1.0) - np.float32(0.3), np.float32(0.3))      # scale * (1.0f - factor), scale * factor
    assert set(sc.triangles["material"]) == {2, 5}


def test_one_level_of_nested_sum_mix_materials_loads():
    """sum(sum(matte, inverse(matte)), microfacet metal): a summed material whose component is a summed material (MULTI record over a MULTI
    record), as SummedSurfaceMaterial.cpp:13-20 builds it."""
    from slr_amd import abi
    it = sl.Interpreter()
    it.run(MULTI_SCRIPT % ('nested = createSurfaceMaterial("sum", (green, ti));', "nested"))
    m = it.build().materials
    top = [i for i, t in enumerate(m["type"]) if t == abi.MAT_MULTI][-1]
    kinds = [m["type"][m["spectrum"][top][k]] for k in range(2)]
    assert sorted(kinds) == sorted([abi.MAT_MF_METAL, abi.MAT_MULTI])


@pytest.mark.parametrize("extra, used", [('mid = createSurfaceMaterial("sum", (green, ti)); bad = createSurfaceMaterial("sum", (mid, green));', "bad"),      # two levels of nesting
                                         ('bad = createSurfaceMaterial("inverse", (ti,));', "bad"),                       # inverse on its own
                                         ('g = createSurfaceMaterial("glass", (SpectrumTexture(Spectrum(0.9, 0.9, 0.9)), SpectrumTexture(Spectrum("ID": "Air", 0)), '
                                          'SpectrumTexture(Spectrum("ID": "Glass_BK7", 0)))); bad = createSurfaceMaterial("sum", (ti, createSurfaceMaterial("inverse", (g,))));', "bad")])
def test_multi_materials_outside_the_supported_subset_are_refused(extra, used):
    it = sl.Interpreter()
    it.run(MULTI_SCRIPT % (extra, used))
    with pytest.raises(sl.UnsupportedFeature):
        it.build()


def test_reference_nodes_become_instances():
    """createReferenceNode (TestScenes/RTC3.txt:21-30 places its grass this way; API.cpp:745-752, nodes.cpp:174-184): the referenced
    subtree is flattened ONCE in its own space — its static transforms baked into the vertices, as the reference does — and every
    reference is a slrhip_instance of that triangle range with the transform of the nodes above it."""
    script = cornell_script("matte") + '''
        tuft = createNode();
        setTransform(tuft, scale(0.5, 1.0, 0.5));
        blade = load3DModel("models/box.assbin", crystal);
        addChild(tuft, blade);
        ref = createReferenceNode(tuft);
        a = createNode(); setTransform(a, translate(0.4, 0.2, 1.0) * rotateY(0.3)); addChild(a, ref); addChild(box, a);
        b = createNode(); setTransform(b, translate(-0.6, 0.4, 1.2) * scale(0.3, 0.6, 0.3)); addChild(b, createReferenceNode(tuft)); addChild(box, b);
    '''
    sc, _, _ = sl.load_scene(script, sphere_tessellation=(8, 4))
    plain, _, _ = sl.load_scene(cornell_script("matte"), sphere_tessellation=(8, 4))
    assert len(sc.instances) == 2 and len(sc.triangles) == len(plain.triangles) + 12          # the box once, not twice
    first = len(plain.triangles)
    assert all(int(i["first_triangle"]) == first and int(i["num_triangles"]) == 12 for i in sc.instances)
    # the mesh sits in the referenced node's own space: the unit box under scale(0.5, 1, 0.5) x the model's own factor 2
    p = sc.vertices["position"][np.unique(sc.triangles["v"][first:])]
    assert np.allclose(np.abs(p).max(axis=0), [0.5, 1.0, 0.5])
    m0 = sc.instances[0]["local_to_world"].reshape(4, 4).T
    assert np.allclose(m0[:3, 3], [0.4, 0.2, 1.0]) and np.allclose(m0[:3, :3] @ m0[:3, :3].T, np.eye(3), atol=1e-6)
    m1 = sc.instances[1]["local_to_world"].reshape(4, 4).T
    assert np.allclose(np.diag(m1)[:3], [0.3, 0.6, 0.3])
    assert np.allclose(sc.instances[1]["world_to_local"].reshape(4, 4).T @ m1, np.eye(4), atol=1e-6)
    with pytest.raises(sl.UnsupportedFeature):
        sl.load_scene(script + "inner = createNode(); addChild(inner, createReferenceNode(tuft)); addChild(root, createReferenceNode(inner));")


def test_syntax_errors_are_reported():
    with pytest.raises(sl.SceneLanguageError):
        sl.Interpreter().run("a = (1, 2;")
    with pytest.raises(sl.SceneLanguageError):
        sl.Interpreter().run("a = undefinedName + 1;")


@pytest.mark.skipif(not os.path.exists("/root/reference/TestScenes/Cornell_Box_Spheres.txt"), reason="reference tree not present")
def test_reference_cornell_file_gives_the_builders_geometry():
    got, settings, renderer = sl.load_scene("/root/reference/TestScenes/Cornell_Box_Spheres.txt", sphere_tessellation=(48, 24))
    want = scenes.cornell_box_spheres(4.0 / 3.0, 48, 24, "glass")
    assert (settings["width"], settings["height"]) == (1024, 768) and renderer["samples"] == 16384
    assert np.array_equal(got.vertices["position"], want.vertices["position"])
    assert np.array_equal(got.triangles["v"], want.triangles["v"])
    assert resolved_materials(got) == resolved_materials(want)
    assert list(got.camera.local_to_world) == list(want.camera.local_to_world)


@pytest.mark.gpu
def test_loaded_scene_renders_like_the_built_one():
    from helpers import assert_bit_equal
    from slr_amd import Context, abi
    got, settings, _ = sl.load_scene(cornell_script("glass"), sphere_tessellation=(16, 8))
    want = scenes.cornell_box_spheres(4.0 / 3.0, 16, 8, "glass")
    st = abi.RenderSettings(64, 48, 0.0, 0.0, 1.0, int(settings["rngSeed"]))
    frames = []
    for sc in (got, want):
        c = Context(stripes=1)
        frames.append(c.render_image(sc, st, 8))
        c.close()
    assert_bit_equal(frames[0], frames[1], "scene script vs builders")


@pytest.mark.gpu
def test_host_program_on_a_scene_script(tmp_path, capsys):
    """python -m slr_amd.host: the reference's HostProgram flow on the HIP path — export cadence, file names, stdout lines."""
    from slr_amd import host
    script = tmp_path / "box.txt"
    script.write_text(cornell_script("matte").replace('"width": 320, "height": 240', '"width": 48, "height": 36'))
    assert host.main([str(script), "--samples", "5", "--out", str(tmp_path)]) == 0
    lines = [l for l in capsys.readouterr().out.splitlines() if "samples:" in l]
    assert [l.split()[0] for l in lines] == ["1", "2", "4"]                 # 5 samples: the image after 8 is never reached
    assert [l.split()[2].rstrip(",") for l in lines] == ["000.bmp", "001.bmp", "002.bmp"]
    data = (tmp_path / "002.bmp").read_bytes()
    assert data[:2] == b"BM" and len(data) == 54 + 36 * (48 * 3)
