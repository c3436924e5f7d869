"""Strong-scaling rehearsal on ONE GPU: time the shard a rank would own at world size N (tiles t % N == 0) for
N = 1, 2, 4, 8 of the headline workload, and print T(1) / (N * T(N)) — the efficiency the tile sharding can reach before
any communication (the reduce is 11 MB).  Also sweeps the stripe count per shard, since the drain tail of the last
paths is the part that does not shrink with N."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slr_amd import Context, abi, scenes  # noqa: E402

W, H, SPP = 1280, 720, 1024
scene = scenes.cornell_box_spheres(W / H, 48, 24, "matte")
st = abi.RenderSettings(W, H, 0.0, 0.0, 1.0, abi.DEFAULT_SEED)


TIMED = "--timed" in sys.argv      # the event-timed launch loop bench.py uses (no hipGraph replay)


def shard_time(world, stripes=0, reps=2):
    c = Context(device=0, mode=abi.MODE_RGB, stripes=stripes, flags=abi.FLAG_TIME_KERNELS if TIMED else 0)
    c.upload_scene(scene)
    c.render_begin(st, (0, world)); c.render(0, 64); c.synchronize()
    best = 1e9
    for _ in range(reps):
        t = time.perf_counter()
        c.render_begin(st, (0, world)); c.render(0, SPP); c.synchronize()
        best = min(best, time.perf_counter() - t)
    it = c.counters().iterations
    c.close()
    return best, it


t1, it1 = shard_time(1)
print("N=1: %.1f ms, %d iterations, %.0f Msamples/s" % (t1 * 1e3, it1, W * H * SPP / t1 / 1e6), flush=True)
for n in (2, 4, 8):
    for stripes in ((0,) if TIMED else (0, 8, 16, 32, 64)):
        t, it = shard_time(n, stripes)
        print("N=%d stripes=%2d: %.1f ms, %d iterations, efficiency %.3f (speed-up %.2fx)" % (n, stripes, t * 1e3, it, t1 / (n * t), t1 / t), flush=True)
