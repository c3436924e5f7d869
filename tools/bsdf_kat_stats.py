"""Per-lobe agreement of slrhip_bsdf_queries with the reference's known answers (tests/golden/bsdf_kat_*.npz):
fraction of bit-equal floats and the distribution of relative differences.  Run on the GPU box."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from helpers import load_golden, scene_from_golden  # noqa: E402
from slr_amd import Context, abi  # noqa: E402

for mode, amode in (("rgb", abi.MODE_RGB), ("spectral", abi.MODE_SPECTRAL)):
    g = load_golden("bsdf_kat_" + mode)
    c = Context(device=0, mode=amode)
    c.upload_scene(scene_from_golden(g))
    for name, m in zip(g["material_names"], g["material_indices"]):
        got = np.stack([c.bsdf_queries(int(m), g["queries"], float(o), float(u)) for o, u in g["wavelengths"]])
        want = g["out_" + str(name)]
        eq = (got.view(np.uint32) == want.view(np.uint32)) | ((got == 0) & (want == 0))
        with np.errstate(divide="ignore", invalid="ignore"):
            rel = np.abs(got.astype(np.float64) - want) / np.maximum(np.abs(want.astype(np.float64)), 1e-30)
        rel[eq] = 0
        rows_bad = (~eq).any(axis=2)
        zero_mismatch = ((got == 0) != (want == 0))
        worst = np.unravel_index(np.argmax(rel), rel.shape)
        print("%-8s %-22s exact floats %.4f  rows exact %.4f  rel p99 %.2e max %.2e  zero/nonzero mismatches %d  worst (wl,row,col)=%s got %g want %g"
              % (mode, name, eq.mean(), 1 - rows_bad.mean(), np.percentile(rel, 99), rel.max(), int(zero_mismatch.sum()), worst,
                 got[worst], want[worst]), flush=True)
    c.close()
