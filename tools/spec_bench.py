import sys, time
sys.path.insert(0, ".")
import numpy as np
from slr_amd import Context, abi, scenes
for mode, name in ((abi.MODE_SPECTRAL, "spectral"), (abi.MODE_RGB, "rgb")):
    for scn, sc in (("boxes_ggx", scenes.cornell_box_boxes(1280/720)), ("spheres_matte", scenes.cornell_box_spheres(1280/720, 48, 24, "matte"))):
        st = abi.RenderSettings(1280, 720, 0.0, 0.0, 1.0, abi.DEFAULT_SEED)
        c = Context(mode=mode, flags=abi.FLAG_TIME_KERNELS)
        c.upload_scene(sc)
        c.render_begin(st); c.render(0, 16); c.synchronize()
        p0 = c.profile()
        t = time.perf_counter(); c.render_begin(st); c.render(0, 512); c.synchronize(); dt = time.perf_counter() - t
        p1 = c.profile()
        k = {n: round((p1.milliseconds[i]-p0.milliseconds[i]) / max(1, p1.launches[i]-p0.launches[i]) * 1e3, 1) for i, n in enumerate(abi.KERNEL_NAMES)}
        print(name, scn, "Msamples/s %.1f" % (1280*720*512/dt/1e6), k, flush=True)
        c.close()
