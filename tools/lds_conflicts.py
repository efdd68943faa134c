#!/usr/bin/env python3
"""LDS bank-conflict share per spectrum kernel from rocprofv3 counter runs:

    rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_X -o c -- \\
        python tools/ab_spec.py --sizes 512,1024,2048,4096,8192,16384 <lib>
    python tools/lds_conflicts.py gpurun_out/pmc_X [...]

SQ_LDS_IDX_ACTIVE = all LDS-array cycles, SQ_LDS_BANK_CONFLICT = the extra ones (MI355X_MICROARCH.md, LDS); mean per launch."""
import collections
import csv
import glob
import sys

for d in [a for a in sys.argv[1:]]:
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not f:
        print(d, "no counter_collection.csv")
        continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][-70:]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[(k, r["Counter_Name"])] += 1
    for k in sorted(acc):
        c = acc[k]
        if c["SQ_LDS_IDX_ACTIVE"] == 0:
            continue
        n = cnt[(k, "SQ_LDS_IDX_ACTIVE")]
        busy = f"  LDS cycles / wave cycles {c['SQ_LDS_IDX_ACTIVE'] / c['SQ_WAVE_CYCLES']:.3f}" if c.get("SQ_WAVE_CYCLES") else ""
        print(f"{d.rstrip('/').split('/')[-1]:14s} {k:72s} launches {n:3d}  LDS_IDX_ACTIVE {c['SQ_LDS_IDX_ACTIVE'] / n:.3e}  "
              f"BANK_CONFLICT {c['SQ_LDS_BANK_CONFLICT'] / n:.3e}  share {c['SQ_LDS_BANK_CONFLICT'] / max(c['SQ_LDS_IDX_ACTIVE'], 1):.3f}{busy}")
