#!/usr/bin/env python
"""The lane-mapping experiment of SURVEY §7 on REAL secondary rays: one lane = one ray (64 rays per wave, four slab tests per lane)
against four lanes = one ray (16 rays per wave, one child box / one leaf triangle per lane, DPP min + ballot ordering).
Rays = the last extension ray of every path slot after a short render (camera rays, bounce rays, mirror chains, in tile order),
traced by both kernels from device memory; hits must be identical.

    python tools/lane_mapping_experiment.py [cornell|grid] [million rays]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slr_amd import Context, abi, scenes  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "cornell"
want = int(float(sys.argv[2]) * 1e6) if len(sys.argv) > 2 else 4_000_000
W, H = 1280, 720
if which == "cornell":
    sc = scenes.cornell_box_spheres(W / H, 48, 24, "matte")
else:
    sc = scenes.displaced_grid(int(os.environ.get("GRID_N", "2236")), W / H)
st = abi.RenderSettings(W, H, 0.0, 0.0, 1.0, abi.DEFAULT_SEED)
c = Context(stripes=8, flags=abi.FLAG_QUAD_LAYOUT)
t = time.time()
c.upload_scene(sc)
print("%s: %d triangles, %d nodes, upload %.1f s" % (which, len(sc.triangles), c.counters().bvh_nodes, time.time() - t), flush=True)
c.render_begin(st)
c.render(0, 16)
n = min(want, W * H * 8)
rays = c.read_slot_rays(0, n)
ok = np.isfinite(rays[:, :7]).all(axis=1) & (np.abs(rays[:, 3:6]).sum(axis=1) > 0)
rays = rays[ok]
print("rays: %d (%.1f %% with a finite tmax)" % (len(rays), 100.0 * np.isfinite(rays[:, 7]).mean()), flush=True)
res = {}
for mapping, name in ((0, "one lane per ray"), (1, "four lanes per ray")):
    hits, ms = c.trace_rays_timed(rays, mapping, repeats=10)
    res[mapping] = hits
    print("%-20s %8.3f ms per launch   %7.2f Grays/s" % (name, ms, len(rays) / ms / 1e6), flush=True)
same = (res[0].view(np.uint32) == res[1].view(np.uint32)).all(axis=1)
miss = res[0][:, 0].view(np.uint32) == 0xFFFFFFFF
print("hits identical: %d of %d rays (%d hit something)" % (same.sum(), len(rays), (~miss).sum()))
assert same.all()
c.close()
