#!/usr/bin/env python3
"""Kernel timeline of the LAST pass of a rocprofv3 --kernel-trace CSV: start offset, duration, gap to the previous kernel's end (us).
    python tools/trace_timeline.py gpurun_out/x/kt_kernel_trace.csv [kernels of the last pass = 7]"""
import csv
import re
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if re.search(r"k_[a-z_0-9]+", r["Kernel_Name"]) and "at::" not in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 7
last = rows[-n:]
t0 = int(last[0]["Start_Timestamp"])
prev_end = None
for r in last:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = re.search(r"(k_[a-z_0-9]+)", r["Kernel_Name"]).group(1)
    gap = "" if prev_end is None else f"gap {(s - prev_end) / 1e3:8.1f} us"
    print(f"{name:22s} q{r.get('Queue_Id', '?'):>3s} start {(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:9.1f} us  {gap}")
    prev_end = e if prev_end is None else max(prev_end, e)
print(f"pass: {(prev_end - t0) / 1e3:.1f} us")
