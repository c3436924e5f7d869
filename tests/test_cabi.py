"""CPU: the C-ABI library loads without a GPU, exports every symbol include/slrhip.h declares, fails loudly
where a device is needed, and its host-only entry points (seeding contract, image export) match the reference."""
import ctypes as C
import os
import re
import tempfile

import numpy as np
import pytest

from helpers import load_golden
from slr_amd import abi, binding

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    return binding.load_library()


def declared_functions():
    text = open(os.path.join(ROOT, "include", "slrhip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(slrhip_[a-z_0-9]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(lib):
    names = declared_functions()
    assert len(names) >= 16
    for n in names:
        assert hasattr(lib, n), "libslrhip.so does not export " + n
    assert sorted(binding.EXPORTS) == names


def test_version(lib):
    text = open(os.path.join(ROOT, "include", "slrhip.h")).read()
    assert lib.slrhip_version() == int(re.search(r"#define SLRHIP_VERSION (\d+)", text).group(1)) == 4


def test_create_fails_loudly_without_gpu(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(binding.SlrHipError) as e:
        binding.Context(device=0)
    assert "no HIP device" in str(e.value)


def test_invalid_arguments_are_rejected(lib):
    assert lib.slrhip_create(None, None) == 1
    assert lib.slrhip_upload_scene(None, None) == 1
    assert lib.slrhip_render(None, 0, 1, None) == 1
    assert b"null" in lib.slrhip_last_error_string()


def test_sample_seed_matches_python_restatement(lib):
    rng = np.random.default_rng(3)
    for _ in range(200):
        seed = int(rng.integers(-2**31, 2**31))
        x, y, s = int(rng.integers(0, 65536)), int(rng.integers(0, 65536)), int(rng.integers(0, 1 << 20))
        assert lib.slrhip_sample_seed(seed, x, y, s) == abi.sample_seed(seed, x, y, s)


@pytest.mark.parametrize("name", ["tonemap_bmp", "tonemap_bmp_spectral"])
def test_tonemap_and_bmp_match_reference_saveimage(lib, name):
    """slrhip_tonemap_bgr8 + slrhip_save_bmp == ImageSensor::saveImage + saveBMP, byte for byte
    (golden BMP written by the compiled reference — RGB build, and spectral build: 16 storage bins through
    DiscretizedSpectrum::getRGB; pad bytes excluded: the reference leaves them uninitialised)."""
    g = load_golden(name)
    fb = np.ascontiguousarray(g["framebuffer"])
    h, w, comps = fb.shape
    byte_width = 3 * w + w % 4
    out = np.zeros(byte_width * h, np.uint8)
    scale = float(g["scale"]) * float(g["sensitivity"])          # saveImage: scale *= sensitivity (ImageSensor.cpp:147)
    rc = lib.slrhip_tonemap_bgr8(fb.ctypes.data, w, h, comps, C.c_float(scale), out.ctypes.data, out.size)
    assert rc == 0
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "o.bmp")
        assert lib.slrhip_save_bmp(path.encode(), out.ctypes.data, w, h) == 0
        mine = np.frombuffer(open(path, "rb").read(), np.uint8)
    ref = g["bmp"]
    assert mine.size == ref.size
    assert (mine[:54] == ref[:54]).all()                          # headers
    a, b = mine[54:].reshape(h, byte_width)[:, :3 * w], ref[54:].reshape(h, byte_width)[:, :3 * w]
    assert (a == b).all(), "%d of %d pixel bytes differ" % ((a != b).sum(), a.size)


def test_cpp_spectrum_construction_matches_the_reference_and_the_python_host(lib):
    """slrhip_upsample / slrhip_resolve_upsampled (the C++ host's construction of spectral-mode spectra) against the compiled
    reference's UpsampledContinuousSpectrum constructor (tests/golden/upsample_kat.npz: (u, v, scale) for 768 inputs over both
    spectrum types and all four colour spaces, SpectrumTypes.h:180-237), and against slr_amd/spectra.py, whose payloads the
    GPU parity tests render with: every float bit-equal."""
    from helpers import assert_bit_equal
    from slr_amd import spectra
    g = load_golden("upsample_kat")["rows"]
    for row in g:
        out = np.zeros(3, np.float32)
        assert lib.slrhip_upsample(int(row[0]), int(row[1]), C.c_float(row[2]), C.c_float(row[3]), C.c_float(row[4]), out.ctypes.data) == 0
        assert_bit_equal(out, row[5:8], "slrhip_upsample %s" % row[:5])
        assert_bit_equal(np.array(spectra.upsample(int(row[0]), int(row[1]), row[2], row[3], row[4]), np.float32), row[5:8], "spectra.upsample")
    assert lib.slrhip_upsample(2, 0, C.c_float(0.5), C.c_float(0.5), C.c_float(0.5), np.zeros(3, np.float32).ctypes.data) == 1   # IOR from sRGB: the reference asserts
    # the (u, v)-only half of evaluate(): cell, data points, weights and the interleaved payload
    holder = type("T", (), {"_tables": None})()          # keeps the table arrays alive
    sc_tables = abi.Scene.upsampling_tables(holder)
    rng = np.random.default_rng(5)
    uvs = np.concatenate([g[:, 5:7], rng.uniform(-1, 15, (512, 2)).astype(np.float32)])
    kinds = set()
    for u, v in uvs:
        n = C.c_uint32(0)
        payload = np.full(4 + 4 * 95, -1.0, np.float32)
        assert lib.slrhip_resolve_upsampled(C.byref(sc_tables), C.c_float(u), C.c_float(v), C.byref(n), payload.ctypes.data) == 0
        pn, pw, pidx = spectra.resolve_upsampled(u, v)
        assert n.value == pn
        kinds.add(pn)
        assert_bit_equal(payload[:4], pw, "weights")
        spec = spectra.tables()["point_spectrum"]
        want = np.stack([spec[pidx[k]] if k < pn else np.zeros(95, np.float32) for k in range(4)], axis=1)
        assert_bit_equal(payload[4:].reshape(95, 4), want, "payload")
    assert kinds == {0, 3, 4}
