import sys, time
sys.path.insert(0, ".")
from slr_amd import Context, abi, scenes
st = abi.RenderSettings(1280, 720, 0.0, 0.0, 1.0, abi.DEFAULT_SEED)
for name, sc in (("boxes_ggx", scenes.cornell_box_boxes(1280/720)), ("spheres", scenes.cornell_box_spheres(1280/720, 48, 24, "matte"))):
    c = Context(mode=abi.MODE_RGB)
    c.upload_scene(sc)
    for rep in range(3):
        t = time.perf_counter(); c.render_begin(st); c.render(0, 512); c.synchronize(); dt = time.perf_counter() - t
        print(name, "render", rep, "%.1f ms -> %.1f Msamples/s" % (dt*1e3, 1280*720*512/dt/1e6), flush=True)
    # the export cadence of PathTracingRenderer: 1, 1, 2, 4, ... passes per call
    t = time.perf_counter(); c.render_begin(st); b = 0
    for n in (1, 1, 2, 4, 8, 16, 32, 64, 128, 256):
        c.render(b, n); b += n
    c.synchronize(); dt = time.perf_counter() - t
    print(name, "doubling cadence to 512: %.1f ms -> %.1f Msamples/s" % (dt*1e3, 1280*720*512/dt/1e6), flush=True)
    c.close()
