"""GPU: the C++ Renderer-shaped adapter (slr_amd/csrc/host) run as a libSLR-style host program; its BMP after 8
passes must equal the tone-mapped framebuffer of the same scene rendered through the Python binding."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import binding as ob
from slr_amd import Context, binding, scenes

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cornell_host_program_writes_reference_style_output(tmp_path):
    exe = os.path.join(ROOT, "slr_amd", "csrc", "host", "cornell_main")
    if not os.path.exists(exe):
        pytest.skip("host program not built")
    w, h, spp = 64, 48, 8
    tables = os.path.join(ROOT, "slr_amd", "data", "upsampling_tables.bin")      # the D65 table, for the light's RGB value
    out = subprocess.check_output([exe, str(spp), str(w), str(h), str(tmp_path), "rgb", tables], text=True)
    lines = [l for l in out.splitlines() if "samples:" in l]
    assert [l.split(" ")[0] for l in lines] == ["1", "2", "4", "8"]              # export cadence of PathTracingRenderer.cpp:83-94
    assert lines[3].split(" ")[2].rstrip(",") == "003.bmp"
    bmp = np.frombuffer(open(tmp_path / "003.bmp", "rb").read(), np.uint8)

    sc = scenes.tiny_box(w / h)                                                  # same walls + light, built in Python
    st = ob.settings(w, h)
    ctx = Context(stripes=0)
    fb = ctx.render_image(sc, st, spp)
    ctx.close()
    lib = binding.load_library()
    byte_width = 3 * w + w % 4
    mine = np.zeros(byte_width * h, np.uint8)
    sens = float(np.float32(1.0 / (np.pi * np.float64(np.float32(0.025)) ** 2)))
    scale = np.float32(np.float32(1.0) / np.float32(spp)) * np.float32(sens)
    assert lib.slrhip_tonemap_bgr8(fb.ctypes.data, w, h, 3, C.c_float(float(scale)), mine.ctypes.data, mine.size) == 0
    assert (bmp[54:] == mine).all(), "%d bytes differ" % (bmp[54:] != mine).sum()


def test_cornell_host_program_builds_a_spectral_scene_in_cpp(tmp_path):
    """The same host program in the spectral build: every spectrum — Spectrum(r, g, b) upsampled reflectances, the D65 light —
    is constructed in C++ (SLRHip::Scene::addUpsampledSpectrum -> slrhip_upsample + slrhip_resolve_upsampled; addD65Spectrum)
    from slr_amd/data/upsampling_tables.bin; its BMP (16 storage bins -> getRGB -> tone map) must equal the one computed from
    the frame of the Python-built scene, whose payloads are pinned against the reference."""
    from slr_amd import abi
    exe = os.path.join(ROOT, "slr_amd", "csrc", "host", "cornell_main")
    if not os.path.exists(exe):
        pytest.skip("host program not built")
    w, h, spp = 48, 40, 4
    tables = os.path.join(ROOT, "slr_amd", "data", "upsampling_tables.bin")
    out = subprocess.check_output([exe, str(spp), str(w), str(h), str(tmp_path), "spectral", tables], text=True)
    lines = [l for l in out.splitlines() if "samples:" in l]
    assert [l.split(" ")[0] for l in lines] == ["1", "2", "4"]
    bmp = np.frombuffer(open(tmp_path / "002.bmp", "rb").read(), np.uint8)
    ctx = Context(mode=abi.MODE_SPECTRAL)
    fb = ctx.render_image(scenes.tiny_box(w / h), ob.settings(w, h), spp)
    ctx.close()
    lib = binding.load_library()
    byte_width = 3 * w + w % 4
    mine = np.zeros(byte_width * h, np.uint8)
    sens = float(np.float32(1.0 / (np.pi * np.float64(np.float32(0.025)) ** 2)))
    scale = np.float32(np.float32(1.0) / np.float32(spp)) * np.float32(sens)
    assert lib.slrhip_tonemap_bgr8(fb.ctypes.data, w, h, 16, C.c_float(float(scale)), mine.ctypes.data, mine.size) == 0
    assert (bmp[54:] == mine).all(), "%d bytes differ" % (bmp[54:] != mine).sum()


def test_cpp_host_reduces_tile_shards_over_rccl():
    """slrhip_reduce_framebuffer from a C++ host (slr_amd/csrc/host/reduce_main.cpp): tile shards rendered one after the other,
    each pushed through ONE ncclReduce on an RCCL communicator (size 1 on this box) and summed — must equal the unsharded
    frame bit for bit; the program exits non-zero otherwise."""
    exe = os.path.join(ROOT, "slr_amd", "csrc", "host", "reduce_main")
    if not os.path.exists(exe):
        pytest.skip("reduce_main not built")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([exe, "5", "200", "136", "8", os.path.join(ROOT, "slr_amd", "data", "upsampling_tables.bin")], text=True, capture_output=True, env=env, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "0 differ" in out.stdout
    # the N-process flow (children forked before any GPU call, ncclUniqueId over a pipe, one device per rank) with the one rank this
    # box has a device for; the same code path an 8-GPU node runs with --ranks 8
    import torch
    ranks = min(torch.cuda.device_count(), 2)
    out = subprocess.run([exe, "--ranks", str(ranks), "200", "136", "8", os.path.join(ROOT, "slr_amd", "data", "upsampling_tables.bin")], text=True, capture_output=True, env=env, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "%d ranks (communicator size %d)" % (ranks, ranks) in out.stdout and " 0 differ" in out.stdout
