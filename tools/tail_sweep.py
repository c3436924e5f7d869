#!/usr/bin/env python
"""Where to hand the last paths of a frame to the tail kernel: the headline frame (world 1) and the shard a rank owns at world
size 8, timed for several SLRHIP_TAIL_SLOTS bounds (0 = pure wavefront, the default).  One process per bound (the bound is read once).

    python tools/tail_sweep.py            # driver: runs itself once per bound
    python tools/tail_sweep.py BOUND      # worker"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) == 1:
    for bound in [int(b) for b in os.environ.get("BOUNDS", "0,65536,262144,1048576,4194304").split(",")]:
        subprocess.run([sys.executable, os.path.abspath(__file__), str(bound)], check=True)
    sys.exit(0)

os.environ["SLRHIP_TAIL_SLOTS"] = sys.argv[1]
sys.path.insert(0, ROOT)
from slr_amd import Context, abi, scenes  # noqa: E402

W, H, SPP = 1280, 720, 1024
scene = scenes.cornell_box_spheres(W / H, 48, 24, "matte")
st = abi.RenderSettings(W, H, 0.0, 0.0, 1.0, abi.DEFAULT_SEED)
line = "tail bound %8s:" % sys.argv[1]
for world in (1, 8):
    c = Context(device=0, mode=abi.MODE_RGB, flags=abi.FLAG_TIME_KERNELS)
    c.upload_scene(scene)
    c.render_begin(st, (0, world)); c.render(0, 64); c.synchronize()
    best = 1e9
    for _ in range(2):
        p0 = c.profile()
        t = time.perf_counter()
        c.render_begin(st, (0, world)); c.render(0, SPP); c.synchronize()
        best = min(best, time.perf_counter() - t)
        p1 = c.profile()
    ms = [p1.milliseconds[k] - p0.milliseconds[k] for k in range(5)]
    line += "   N=%d %.1f ms (%d iterations; trace %.1f shade %.1f regen %.1f tail %.2f ms)" % (world, best * 1e3, c.counters().iterations, ms[0], ms[2], ms[3], ms[4])
    c.close()
print(line, flush=True)
