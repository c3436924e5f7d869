"""Experiment: do two half-size path pools on two streams overlap (trace of one with shade/regen of the other)?"""
import sys, time, threading
sys.path.insert(0, ".")
import torch
from slr_amd import Context, abi, scenes

W, H, SPP = 1280, 720, 256
sc = scenes.cornell_box_spheres(W / H, 48, 24, "matte")
st = abi.RenderSettings(W, H, 0.0, 0.0, 1.0, abi.DEFAULT_SEED)


def one(stripes, spp, stream=None, begin=0):
    c = Context(stripes=stripes)
    c.upload_scene(sc)
    c.render_begin(st)
    c.render(begin, 16, stream)
    c.synchronize()
    return c


for stripes in (8, 4):
    c = one(stripes, SPP)
    t = time.perf_counter(); c.render_begin(st); c.render(0, SPP); c.synchronize(); dt = time.perf_counter() - t
    print("single context stripes=%d: %.1f Msamples/s" % (stripes, W * H * SPP / dt / 1e6), flush=True)
    c.close()

for stripes in (4, 8):
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    ctxs = [one(stripes, SPP, s.cuda_stream) for s in streams]
    errors = []
    def work(i):
        try:
            ctxs[i].render_begin(st)
            ctxs[i].render(i * SPP // 2, SPP // 2, streams[i].cuda_stream)
        except Exception as e:      # a worker's failure must fail the probe, not vanish with the thread
            errors.append(e)
    t = time.perf_counter()
    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    [x.start() for x in th]; [x.join() for x in th]
    if errors:
        raise errors[0]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    print("two contexts x stripes=%d on two streams: %.1f Msamples/s" % (stripes, W * H * SPP / dt / 1e6), flush=True)
    [c.close() for c in ctxs]
