#!/usr/bin/env python
"""Writes slr_amd/data/upsampling_tables.bin for hosts without numpy (the C++ SLRHip::Scene): the Meng-15 upsampling grid and
the D65 table, from slr_amd/data/spectral_tables.npz (dumped from the compiled reference by tools/extract_spectral_tables.py).

Layout (little endian): "SLRUPS01", u32 grid_width, grid_height, num_points, num_d65; then grid cells (width * height * 8 bytes:
inside, num_points, idx[6]), point_uv (num_points * 2 f32), point_spectrum (num_points * 95 f32), d65 (num_d65 f32, 300-830 nm)."""
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from slr_amd import spectra  # noqa: E402

t = spectra.tables()
cells = np.zeros((len(t["grid_inside"]), 8), np.uint8)
cells[:, 0], cells[:, 1], cells[:, 2:8] = t["grid_inside"], t["grid_num_points"], t["grid_idx"]
path = os.path.join(ROOT, "slr_amd", "data", "upsampling_tables.bin")
with open(path, "wb") as f:
    f.write(b"SLRUPS01")
    f.write(struct.pack("<4I", spectra.GRID_WIDTH, spectra.GRID_HEIGHT, len(t["point_uv"]), len(t["d65"])))
    f.write(cells.tobytes())
    f.write(np.ascontiguousarray(t["point_uv"], "<f4").tobytes())
    f.write(np.ascontiguousarray(t["point_spectrum"], "<f4").tobytes())
    f.write(np.ascontiguousarray(t["d65"], "<f4").tobytes())
print(path, os.path.getsize(path), "bytes")
