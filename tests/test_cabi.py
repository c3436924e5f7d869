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
