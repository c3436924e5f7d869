"""Dumps the published numeric tables the spectral path needs into slr_amd/data/spectral_tables.npz.

The tables exist only inside the reference tree (no network): the Meng et al. 2015 RGB-upsampling grid
(libSLR/BasicTypes/Spectrum.h:199-575), the CIE 1931 2-degree colour matching functions, the D65 illuminant
(libSLR/BasicTypes/common_spectra.cpp) and the refractive-index tables of refractiveindex.info
(libSLR/BasicTypes/spectrum_library.cpp).  They are read through the COMPILED reference (oracle/_ref, function
slr_ref_dump_table) and stored as plain arrays: data, not source.  Run where /root/reference exists:

    python tools/extract_spectral_tables.py
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import binding as ob  # noqa: E402

IOR_NAMES = ["Air", "Water", "Glass_BK7", "Diamond", "Aluminium", "Copper", "Gold", "Iron", "Lead", "Mercury", "Platinum",
             "Silver", "Titanium"]


def main():
    ref = ob.load("ref_spectral")
    if ref is None:
        raise SystemExit("oracle/_ref is not built")
    f = ref.lib.slr_ref_dump_table
    f.restype = C.c_long
    f.argtypes = [C.c_int, C.c_char_p, C.c_void_p, C.c_long]

    def dump(what, name=None, dtype=np.float32):
        n = f(what, name.encode() if name else None, None, 0)
        if n < 0:
            raise RuntimeError("unknown table %s %s" % (what, name))
        buf = np.zeros(n, dtype=dtype)
        f(what, name.encode() if name else None, buf.ctypes.data, n)
        return buf

    out = {}
    cells = dump(0, dtype=np.uint8).reshape(-1, 8)
    out["grid_inside"] = cells[:, 0].copy()
    out["grid_num_points"] = cells[:, 1].copy()
    out["grid_idx"] = cells[:, 2:8].copy()
    pts = dump(1).reshape(-1, 2 + 2 + 95)
    out["point_xystar"] = pts[:, 0:2].copy()
    out["point_uv"] = pts[:, 2:4].copy()
    out["point_spectrum"] = pts[:, 4:].copy()
    out["cmf"] = dump(2).reshape(3, 471)
    out["d65"] = dump(3)
    for name in IOR_NAMES:
        t = dump(4, name)
        ns = int(t[0])
        out["ior_%s_meta" % name] = t[1:5].copy()           # min lambda, max lambda, regular?, has k?
        out["ior_%s_lambdas" % name] = t[5:5 + ns].copy()
        out["ior_%s_etas" % name] = t[5 + ns:5 + 2 * ns].copy()
        out["ior_%s_ks" % name] = t[5 + 2 * ns:5 + 3 * ns].copy()
    path = os.path.join(ROOT, "slr_amd", "data", "spectral_tables.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: v.shape for k, v in out.items() if not k.startswith("ior_")}, len(IOR_NAMES), "IOR tables")


if __name__ == "__main__":
    main()
