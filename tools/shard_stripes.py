"""Stripe count of the N = 4 and N = 8 shards of the headline workload (same box, alternating): how many paths per pixel the
sample pool should keep in flight when a rank owns a fraction of the image."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slr_amd import Context, abi, scenes  # noqa: E402

W, H, SPP = 1280, 720, 1024
scene = scenes.cornell_box_spheres(W / H, 48, 24, "matte")
st = abi.RenderSettings(W, H, 0.0, 0.0, 1.0, abi.DEFAULT_SEED)


def shard_time(world, stripes):
    c = Context(device=0, mode=abi.MODE_RGB, stripes=stripes, flags=abi.FLAG_TIME_KERNELS)
    c.upload_scene(scene)
    c.render_begin(st, (0, world)); c.render(0, 64); c.synchronize()
    best = 1e9
    for _ in range(2):
        t = time.perf_counter()
        c.render_begin(st, (0, world)); c.render(0, SPP); c.synchronize()
        best = min(best, time.perf_counter() - t)
    it = c.counters().iterations
    c.close()
    return best, it


for rnd in range(2):
    for world, ks in ((1, (8, 16, 24, 32)), (4, (16, 32, 48, 64)), (8, (16, 32, 48, 64))):
        for k in ks:
            t, it = shard_time(world, k)
            print("N=%d stripes=%3d: %.1f ms, %d iterations" % (world, k, t * 1e3, it), flush=True)
