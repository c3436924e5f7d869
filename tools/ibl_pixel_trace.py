"""Where the environment-light frame's largest relative difference comes from (bench.py's parity leg reports max_rel_err; round 2's
line had 0.229 on one pixel under an RMSE of 4e-5).  Run on the GPU box:

    python tools/ibl_pixel_trace.py [spp]

Renders BASELINE configs[3]'s scene at 1280x720 with ONE stripe on the GPU and with the oracle, finds the pixel with the largest
relative difference, then renders every pass of that pixel on both sides on its own (slrhip_render(ctx, k, 1) / oracle
render(spp_begin = k)) to name the (pixel, pass) that differs, and prints the oracle's own value of that sample
(oracle.sample) next to the two frames' — a sample that sees the sun texel is hundreds of times the mean, so one float-libm
ulp that moves a look-up across a texel edge, or flips a discrete choice, shows up as a large relative difference on one pixel."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import binding as ob  # noqa: E402
from slr_amd import Context, abi, scenes  # noqa: E402

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 16
W, H = 1280, 720
sc = scenes.ibl_test_scene(W / H, (2048, 1024), 48, 24)
st = abi.RenderSettings(W, H, 0.0, 0.0, 1.0, abi.DEFAULT_SEED)
orc = ob.load("oracle").scene(sc)
want, _ = orc.render(st, spp, threads=16)
c = Context(stripes=1)
c.upload_scene(sc)
c.render_begin(st)
c.render(0, spp)
got = c.read_framebuffer()
nz = np.abs(want) > 1e-9
rel = np.zeros_like(want, dtype=np.float64)
rel[nz] = np.abs(got.astype(np.float64) - want)[nz] / np.abs(want[nz])
y, x, ch = np.unravel_index(np.argmax(rel), rel.shape)
exact = ((got.view(np.uint32) == want.view(np.uint32)) | ((got == 0) & (want == 0)))
print("frame %dx%d x %d spp: bit-exact floats %.5f, pixels with any difference %d, max rel %.4g at pixel (%d, %d) channel %d: gpu %.9g oracle %.9g, frame mean %.4g"
      % (W, H, spp, exact.mean(), int((~exact).any(axis=2).sum()), rel.max(), x, y, ch, got[y, x, ch], want[y, x, ch], want.mean()))
worst = None
for k in range(spp):
    c.render_begin(st)
    c.render(k, 1)
    g1 = c.read_framebuffer()[y, x].copy()
    o1 = orc.render(st, 1, spp_begin=k, threads=16)[0][y, x].copy()
    d = float(np.abs(g1.astype(np.float64) - o1).max())
    tag = ""
    if not np.array_equal(g1.view(np.uint32), o1.view(np.uint32)):
        tag = "  <-- differs"
        if worst is None or d > worst[1]:
            worst = (k, d, g1, o1)
    print("  pass %2d: gpu %s oracle %s%s" % (k, g1, o1, tag))
c.close()
if worst is None:
    print("every pass of that pixel is bit-equal on its own: the frame difference is the accumulation order only")
else:
    k, d, g1, o1 = worst
    s = orc.sample(st, int(x), int(y), int(k))
    print("pass %d of pixel (%d, %d): gpu %s, oracle %s; oracle.sample: %s" % (k, x, y, g1, o1, s))
    print("ratio of the pass's largest channel to the frame mean per sample: %.1f (a sample that reaches the sun disc)" % (float(max(g1.max(), o1.max())) / (want.mean() / spp)))
    n_diff_small = int(((rel > 0) & (rel < 1e-5)).sum())
    print("floats that differ by < 1e-5 relative elsewhere in the frame: %d; by >= 1e-3: %d" % (n_diff_small, int((rel >= 1e-3).sum())))
