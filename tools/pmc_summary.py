#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files per pss kernel (mean per dispatch, and per wave)."""
import collections
import csv
import re
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(lambda: collections.defaultdict(set))
meta = {}
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        m = re.search(r"(k_[a-z_0-9]+)", r["Kernel_Name"])
        if not m or "at::" in r["Kernel_Name"]:
            continue
        n = m.group(1)
        agg[n][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[n][r["Counter_Name"]].add((path, r["Dispatch_Id"]))
        meta[n] = (float(r["Grid_Size"]), float(r["Workgroup_Size"]), r["VGPR_Count"], r["Accum_VGPR_Count"], r["LDS_Block_Size"])
for n, d in agg.items():
    grid, wg, vg, ag, lds = meta[n]
    waves = grid / 64
    print(f"{n}: grid={grid:.0f} wg={wg:.0f} waves={waves:.0f} vgpr={vg} agpr={ag} lds={lds}")
    v = {k: d[k] / len(disp[n][k]) for k in d}
    wc = v.get("SQ_WAVE_CYCLES", 0)
    for k in sorted(v):
        extra = f"  ({v[k] / wc:.2f} of wave cycles)" if wc and k.startswith(("SQ_WAIT", "SQ_ACTIVE")) else ""
        print(f"    {k:24s} {v[k]:12.4e}  per wave {v[k] / waves:10.1f}{extra}")
