"""CPU: the ctypes mirror matches include/slrhip.h (sizes/offsets checked by compiling a probe)."""
import ctypes as C
import os
import subprocess
import tempfile

from slr_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PROBE = r"""
#include <stdio.h>
#include <stddef.h>
#include "slrhip.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(slrhip_vertex), sizeof(slrhip_triangle),
         sizeof(slrhip_material), sizeof(slrhip_spectrum), sizeof(slrhip_camera), sizeof(slrhip_scene_desc),
         sizeof(slrhip_render_settings), sizeof(slrhip_config), sizeof(slrhip_counters), sizeof(slrhip_envmap), sizeof(slrhip_profile),
         offsetof(slrhip_scene_desc, camera), offsetof(slrhip_scene_desc, env));
  return 0;
}
"""


def test_struct_layouts_match_header():
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "probe.c")
        open(src, "w").write(PROBE)
        exe = os.path.join(d, "probe")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), src, "-o", exe])
        got = [int(v) for v in subprocess.check_output([exe]).split()]
    want = [abi.vertex_dtype.itemsize, abi.triangle_dtype.itemsize, abi.material_dtype.itemsize,
            abi.spectrum_dtype.itemsize, C.sizeof(abi.Camera), C.sizeof(abi.SceneDesc), C.sizeof(abi.RenderSettings),
            C.sizeof(abi.Config), C.sizeof(abi.Counters), C.sizeof(abi.EnvMap), C.sizeof(abi.Profile), abi.SceneDesc.camera.offset, abi.SceneDesc.env.offset]
    assert got == want
