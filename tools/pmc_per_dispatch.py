#!/usr/bin/env python
"""Per-DISPATCH HBM traffic of the hot-path kernels from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE): the launches of a
render in launch order, so that the steady state (every slot live) can be read apart from the drain at the end of a render.

    python tools/pmc_per_dispatch.py FETCH_DIR WRITE_DIR > table.txt

Same units and gfx950 correction as tools/pmc_traffic.py: bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024."""
import csv
import glob
import os
import sys

CLASSES = {"k_trace_ws": "trace", "k_shade": "shade", "k_tail<": "tail"}


def collect(d, counter):
    rows = {}
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if row["Counter_Name"] != counter:
                    continue
                cls = next((v for k, v in CLASSES.items() if k in row["Kernel_Name"]), None)
                if cls is None:
                    continue
                rows[int(row["Dispatch_Id"])] = (cls, rows.get(int(row["Dispatch_Id"]), (cls, 0.0))[1] + float(row["Counter_Value"]))
    return rows


def main():
    fetch, write = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
    order = {}
    for cls in ("shade", "trace", "tail"):
        f = [v for k, (c, v) in sorted(fetch.items()) if c == cls]
        w = [v for k, (c, v) in sorted(write.items()) if c == cls]
        order[cls] = list(zip(f, w))
    print("# launch  shade_read_GB shade_write_GB  trace_read_GB trace_write_GB   (read = 2 x FETCH_SIZE KiB, write = WRITE_SIZE KiB)")
    n = max(len(order["shade"]), len(order["trace"]))
    for i in range(n):
        s = order["shade"][i] if i < len(order["shade"]) else (0.0, 0.0)
        t = order["trace"][i] if i < len(order["trace"]) else (0.0, 0.0)
        print("%4d  %8.3f %8.3f   %8.3f %8.3f" % (i, 2 * s[0] * 1024 / 1e9, s[1] * 1024 / 1e9, 2 * t[0] * 1024 / 1e9, t[1] * 1024 / 1e9))


if __name__ == "__main__":
    main()
