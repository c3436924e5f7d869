"""Schedule diagnostics of the wave-specialised closest-hit kernel (counting build): SLRHIP_DEBUG_WS=1 python tools/ws_debug.py"""
import os, sys
sys.path.insert(0, ".")
os.environ.setdefault("SLRHIP_DEBUG_WS", "1")
from slr_amd import Context, abi, scenes
W, H = 1280, 720
for name, sc in (("cornell", scenes.cornell_box_spheres(W / H, 48, 24, "matte")),):
    c = Context(flags=abi.FLAG_COUNT_TRAVERSAL | abi.FLAG_TIME_KERNELS)
    c.upload_scene(sc)
    c.render_begin(abi.RenderSettings(W, H, 0.0, 0.0, 1.0, abi.DEFAULT_SEED))
    c.render(0, 64)
    p = c.profile()
    print(name, "launches", p.launches[0], "closest ms", p.milliseconds[0], flush=True)
    c.close()
