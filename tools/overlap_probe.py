"""Experiment: do two half-size path pools on two streams overlap (the traversal launch of one with the shade launch of the other)?
Two contexts, each with half the slots, each rendering half of the passes of the headline frame from its own thread on its own
stream, against one context with all the slots.  SLRHIP_WS_BLOCKS_PER_CU (variant builds) leaves LDS and wave slots for the other
context's kernels."""
import sys, time, threading
sys.path.insert(0, ".")
import torch
from slr_amd import Context, abi, scenes

W, H, SPP = 1280, 720, 1024
sc = scenes.cornell_box_spheres(W / H, 48, 24, "matte")
st = abi.RenderSettings(W, H, 0.0, 0.0, 1.0, abi.DEFAULT_SEED)


def make(stripes):
    c = Context(stripes=stripes, flags=abi.FLAG_TAIL_KERNEL)
    c.upload_scene(sc)
    c.render_begin(st)
    c.render(0, 64)
    c.synchronize()
    return c


for rnd in range(2):
    c = make(32)
    t = time.perf_counter(); c.render_begin(st); c.render(0, SPP); c.synchronize(); dt = time.perf_counter() - t
    print("one context, 32 slots per pixel: %.1f Msamples/s" % (W * H * SPP / dt / 1e6), flush=True)
    c.close()
    for stripes in (16, 32):
        ctxs = [make(stripes) for _ in range(2)]
        errors = []

        def work(i):
            try:
                ctxs[i].render_begin(st)
                ctxs[i].render(i * SPP // 2, SPP // 2)
            except Exception as e:      # a worker's failure must fail the probe, not vanish with the thread
                errors.append(e)
        t = time.perf_counter()
        th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
        [x.start() for x in th]; [x.join() for x in th]
        if errors:
            raise errors[0]
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        print("two contexts x %d slots per pixel, half the passes each, concurrently: %.1f Msamples/s" % (stripes, W * H * SPP / dt / 1e6), flush=True)
        [c.close() for c in ctxs]
