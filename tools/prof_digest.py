#!/usr/bin/env python3
"""Digest a tools/prof_round.sh output directory into a small text summary (pss kernels only)."""
import collections
import csv
import json
import os
import re
import sys

out, tag = sys.argv[1], sys.argv[2]
lines = []
p = os.path.join(out, "kt_kernel_stats.csv")
if os.path.exists(p):
    lines.append("== rocprofv3 --kernel-trace --stats : python bench.py --steps 10 --warmup 2 --no-cpu-baseline")
    lines.append(f"{'kernel':28s} {'calls':>6s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>7s}")
    for r in csv.DictReader(open(p)):
        m = re.search(r"(k_[a-z_0-9]+)", r["Name"])
        if not m or "at::" in r["Name"]:
            continue
        lines.append(f"{m.group(1):28s} {r['Calls']:>6s} {float(r['AverageNs']) / 1e3:10.1f} {float(r['MinNs']) / 1e3:10.1f} "
                     f"{float(r['MaxNs']) / 1e3:10.1f} {float(r['Percentage']):7.2f}")
for name, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    p = os.path.join(out, f"{name}_counter_collection.csv")
    if not os.path.exists(p):
        continue
    agg = collections.defaultdict(float)
    n = collections.Counter()
    for r in csv.DictReader(open(p)):
        m = re.search(r"(k_[a-z_0-9]+)", r["Kernel_Name"])
        if not m or "at::" in r["Kernel_Name"] or r["Counter_Name"] != ctr:
            continue
        agg[m.group(1)] += float(r["Counter_Value"])
        n[m.group(1)] += 1
    lines.append(f"== rocprofv3 --pmc {ctr} (raw counter, KiB per dispatch; gfx950: FETCH_SIZE under-reports wide coalesced "
                 f"reads by 2x — MI355X_MICROARCH.md §HBM)")
    for k in agg:
        lines.append(f"{k:28s} dispatches={n[k]:4d}  {ctr}={agg[k] / n[k]:14.1f} KiB  ({agg[k] / n[k] * 1024 / 1e6:10.2f} MB)")
for lg in ("kt.log",):
    p = os.path.join(out, lg)
    if os.path.exists(p):
        for l in open(p):
            if l.startswith('{"metric"'):
                j = json.loads(l)
                lines.append("== bench line of the kernel-trace run")
                lines.append(json.dumps({k: j[k] for k in ("value", "unit", "ms_per_step", "roofline")}))
txt = "\n".join(lines)
open(os.path.join(out, f"summary_{tag}.txt"), "w").write(txt + "\n")
print(txt)
