#!/usr/bin/env python
"""Where do the last 0.003 % of the 10 M-triangle grid's values come from?  Real path rays (the last extension ray of every
slot after a short render) traced by the HIP library (slrhip_trace_rays: the production node / leaf layouts) and by the oracle
(its own binary BVH); every ray whose hit differs is printed with both answers.

    python tools/grid_hit_parity.py [million rays] [grid n]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import binding as ob  # noqa: E402
from slr_amd import Context, abi, scenes  # noqa: E402

want = int(float(sys.argv[1]) * 1e6) if len(sys.argv) > 1 else 2_000_000
n_grid = int(sys.argv[2]) if len(sys.argv) > 2 else 2236
W, H = 1280, 720
sc = scenes.displaced_grid(n_grid, W / H)
st = abi.RenderSettings(W, H, 0.0, 0.0, 1.0, abi.DEFAULT_SEED)
c = Context(stripes=8)
t = time.time()
c.upload_scene(sc)
print("%d triangles, %d nodes, upload %.1f s" % (len(sc.triangles), c.counters().bvh_nodes, time.time() - t), flush=True)
c.render_begin(st)
c.render(0, 16)
rays = c.read_slot_rays(0, min(want, W * H * 8))
ok = np.isfinite(rays[:, :7]).all(axis=1) & (np.abs(rays[:, 3:6]).sum(axis=1) > 0)
rays = rays[ok]
print("rays: %d" % len(rays), flush=True)
tri, dist, b0, b1 = c.trace_rays(rays[:, 0:3], rays[:, 3:6], rays[:, 6], rays[:, 7])
c.close()
t = time.time()
osc = ob.load("oracle", abi.MODE_RGB).scene(sc)
print("oracle scene %.1f s" % (time.time() - t), flush=True)
t = time.time()
oh = osc.trace(rays.view(ob.ray_dtype).reshape(-1))
print("oracle trace %.1f s" % (time.time() - t), flush=True)
same_tri = tri == oh["triangle"]
same_all = same_tri & (dist.view(np.uint32) == oh["dist"].view(np.uint32)) & (b0.view(np.uint32) == oh["b0"].view(np.uint32)) & (b1.view(np.uint32) == oh["b1"].view(np.uint32))
print("identical hit records: %d of %d (%d hit something); same triangle: %d" % (same_all.sum(), len(rays), (tri != 0xFFFFFFFF).sum(), same_tri.sum()))
T = sc.triangles["v"]
for i in np.flatnonzero(~same_all)[:40]:
    g, o = int(tri[i]), int(oh["triangle"][i])
    shared = len(set(T[g].tolist()) & set(T[o].tolist())) if g != 0xFFFFFFFF and o != 0xFFFFFFFF else -1
    print("ray %8d: gpu tri %9d t %.9g (%08x) | oracle tri %9d t %.9g (%08x) | equal t: %s, shared vertices: %d, tmin %.3g tmax %.6g" % (
        i, g, dist[i], dist[i:i + 1].view(np.uint32)[0], o, oh["dist"][i], oh["dist"][i:i + 1].view(np.uint32)[0], dist[i] == oh["dist"][i], shared, rays[i, 6], rays[i, 7]))
