#!/usr/bin/env python
"""HBM traffic per launch of each hot-path kernel from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE cannot
share a pass on gfx950: MI355X_MICROARCH.md, "rocprofv3 PMC slots").

    python tools/pmc_traffic.py WORKLOAD FETCH_DIR WRITE_DIR [existing.json] > profiles/pmc_traffic.json

Units and corrections (MI355X_MICROARCH.md, "HBM"): both counters are in KiB; on gfx950 FETCH_SIZE reports half the
bytes of wide coalesced reads, so bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.  Kernel template variants
(k_shade<...>) are merged, weighted by their dispatch counts; the traversal-counting k_trace_*<true> variants are skipped."""
import csv
import glob
import json
import os
import sys

CLASSES = {"k_trace_ws": "trace", "k_shade": "shade", "k_tail<": "tail"}


def collect(d, counter):
    acc = {}
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if row["Counter_Name"] != counter:
                    continue
                name = row["Kernel_Name"]
                if "k_trace" in name and ("<true>" in name or "<true," in name):
                    continue          # the traversal-counting variant runs only in bench.py's untimed statistics pass
                cls = next((v for k, v in CLASSES.items() if k in name), None)
                if cls is None:
                    continue
                e = acc.setdefault(cls, {"sum": 0.0, "ids": set()})
                e["sum"] += float(row["Counter_Value"])
                e["ids"].add(row["Dispatch_Id"])
    return {k: (v["sum"], len(v["ids"])) for k, v in acc.items()}


def main():
    workload, fetch_dir, write_dir = sys.argv[1:4]
    out = json.load(open(sys.argv[4])) if len(sys.argv) > 4 and os.path.exists(sys.argv[4]) else {}
    fetch, write = collect(fetch_dir, "FETCH_SIZE"), collect(write_dir, "WRITE_SIZE")
    entry = {}
    for cls in fetch:
        fs, fn = fetch[cls]
        ws, wn = write.get(cls, (0.0, 1))
        entry[cls] = {"launches": fn, "fetch_size_kib_per_launch": fs / fn, "write_size_kib_per_launch": ws / max(wn, 1),
                      "traffic_bytes_per_launch": (2.0 * fs / fn + ws / max(wn, 1)) * 1024.0}
    out[workload] = entry
    json.dump(out, sys.stdout, indent=1, sort_keys=True)
    print()


if __name__ == "__main__":
    main()
