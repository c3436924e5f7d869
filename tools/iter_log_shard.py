#!/usr/bin/env python
"""Per-iteration kernel times (SLRHIP_ITER_LOG) of the headline frame for the shard a rank owns at world size N.

    python tools/iter_log_shard.py WORLD OUT.txt [stripes]"""
import os
import sys
import time

world, out = int(sys.argv[1]), sys.argv[2]
os.environ["SLRHIP_ITER_LOG"] = out
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slr_amd import Context, abi, scenes  # noqa: E402

W, H, SPP = 1280, 720, 1024
scene = scenes.cornell_box_spheres(W / H, 48, 24, "matte")
st = abi.RenderSettings(W, H, 0.0, 0.0, 1.0, abi.DEFAULT_SEED)
c = Context(device=0, mode=abi.MODE_RGB, stripes=int(sys.argv[3]) if len(sys.argv) > 3 else 0, flags=abi.FLAG_TIME_KERNELS)
c.upload_scene(scene)
c.render_begin(st, (0, world)); c.render(0, 64); c.synchronize()
open(out, "w").close()
t = time.perf_counter()
c.render_begin(st, (0, world)); c.render(0, SPP); c.synchronize()
dt = time.perf_counter() - t
print("world %d: %.1f ms, %d iterations" % (world, dt * 1e3, c.counters().iterations))
c.close()
