#!/usr/bin/env python
"""Summarise rocprofv3 --pmc output (…_counter_collection.csv) per kernel: dispatches, counter totals and per-dispatch means.

    python tools/pmc_summary.py gpurun_out/pmc_fetch [more dirs…] > profiles/r01_c_pmc_summary.json
"""
import csv
import glob
import json
import os
import re
import sys


def short(name):
    m = re.match(r"(?:void )?(?:slrhip::)?([A-Za-z_0-9]+)(<[^(]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name


def main():
    out = {}
    for d in sys.argv[1:]:
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            seen = {}
            with open(path, newline="") as f:
                for row in csv.DictReader(f):
                    k = short(row["Kernel_Name"])
                    c = row["Counter_Name"]
                    e = out.setdefault(k, {}).setdefault(c, {"sum": 0.0, "dispatches": 0})
                    e["sum"] += float(row["Counter_Value"])
                    key = (k, c, row["Dispatch_Id"])
                    if key not in seen:
                        seen[key] = 1
                        e["dispatches"] += 1
    for k in out:
        for c, e in out[k].items():
            e["per_dispatch"] = e["sum"] / max(e["dispatches"], 1)
    json.dump(out, sys.stdout, indent=1, sort_keys=True)
    print()


if __name__ == "__main__":
    main()
