#!/usr/bin/env python3
"""Digest of tools/prof_single.sh's output directory: per mode the wall time per call and the kernels behind it."""
import csv
import os
import re
import sys

d = sys.argv[1]
for m in ("NFM", "AM", "WFM", "USB"):
    log = open(os.path.join(d, m + ".log")).read()
    wall = re.search(r"([0-9.]+) ms per call", log)
    rows = list(csv.DictReader(open(os.path.join(d, m + "_kernel_stats.csv"))))
    calls = max(int(r["Calls"]) for r in rows if "k_" in r["Name"]) if rows else 1
    print(f"{m}: {wall.group(1) if wall else '?'} ms per call (host buffer in, host buffer out, under the profiler)")
    tot = 0.0
    for r in rows:
        k = re.search(r"(k_[a-z_0-9]+|__amd_rocclr_[A-Za-z]+)", r["Name"])
        per_call = float(r["TotalDurationNs"]) / 1e3 / 11
        tot += per_call
        print(f"    {k.group(1) if k else r['Name'][:40]:32s} launches/call {int(r['Calls']) / 11:4.1f}  mean {float(r['AverageNs']) / 1e3:8.1f} us  per call {per_call:8.1f} us")
    print(f"    kernels per call: {tot:.0f} us")
