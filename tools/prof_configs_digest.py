#!/usr/bin/env python3
"""Digest tools/prof_configs.sh: per config of tools/bench_configs.py, the HBM traffic of ONE pass from the FETCH_SIZE / WRITE_SIZE
counter passes (KiB per dispatch; gfx950: FETCH_SIZE under-reports wide coalesced reads by 2x — MI355X_MICROARCH.md §HBM — so
traffic = 2 x FETCH_SIZE + WRITE_SIZE, as tools/prof_digest.py computes it for the headline step) and the kernel-trace durations.

    python tools/prof_configs_digest.py gpurun_out/prof_cfgs_r04 r04
      -> gpurun_out/prof_r04/hbm_traffic.json gains {"configs": {cfg: {"traffic_bytes", "passes", "kernels": {k: {...}}}}}
         (bench.py / tools/bench_configs.py quote it as other_configs.<cfg>.traffic_bytes / traffic_ratio when src_hash matches)
      -> gpurun_out/prof_cfgs_r04/r04_configs_summary.txt (copied to profiles/<tag>_other_configs_kernel_stats_and_hbm.txt)
"""
import collections
import csv
import json
import os
import re
import sys

out, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
PASSES = 2  # bench_configs.py --launch-only runs the config's step twice


def kname(full):
    m = re.search(r"(k_[a-z_0-9]+)", full)
    return None if (not m or "at::" in full) else m.group(1)


def source_hash():
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "pyspecsdr_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h", ".cpp")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


lines = [f"== per-config profile {tag}: source hash {source_hash()}; each config = python tools/bench_configs.py <cfg> --launch-only "
         f"({PASSES} plain passes), rocprofv3 --kernel-trace --stats / --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate runs"]
configs = {}
for cfg in ("cfg2_f32_rows", "cfg2_exact_cells", "cfg3", "cfg4", "cfg5_resident", "wfm_step"):
    ctr = {}
    for name, c in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        p = os.path.join(out, f"{cfg}_{name}_counter_collection.csv")
        if not os.path.exists(p):
            continue
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(p)):
            n = kname(r["Kernel_Name"])
            if n and r["Counter_Name"] == c:
                acc[n].append(float(r["Counter_Value"]) * 1024)
        ctr[c] = acc
    if len(ctr) < 2:
        continue
    dur = collections.defaultdict(list)
    p = os.path.join(out, f"{cfg}_kt_kernel_trace.csv")
    if os.path.exists(p):
        for r in csv.DictReader(open(p)):
            n = kname(r["Kernel_Name"])
            if n:
                dur[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    ks = {}
    lines.append(f"== {cfg}: kernel, launches per pass, ms per pass (kernel trace), 2 x FETCH_SIZE + WRITE_SIZE per pass")
    for n in sorted(set(ctr["FETCH_SIZE"]) | set(ctr["WRITE_SIZE"])):
        f, w = sum(ctr["FETCH_SIZE"].get(n, [])) / PASSES, sum(ctr["WRITE_SIZE"].get(n, [])) / PASSES
        ks[n] = {"launches": len(ctr["FETCH_SIZE"].get(n, [])) // PASSES, "fetch_size_bytes_raw": f, "write_size_bytes": w,
                 "traffic_bytes": 2 * f + w}
        if dur.get(n):
            ks[n]["rocprof_ms"] = sum(dur[n]) / PASSES
        lines.append(f"{n:24s} launches={ks[n]['launches']:3d}  ms={ks[n].get('rocprof_ms', float('nan')):8.4f}  fetch(raw)={f / 1e6:10.2f} MB  "
                     f"write={w / 1e6:10.2f} MB  traffic={(2 * f + w) / 1e6:10.2f} MB")
    tot = sum(k["traffic_bytes"] for k in ks.values())
    configs[cfg] = {"traffic_bytes": tot, "passes": PASSES, "kernels": ks}
    lines.append(f"{cfg}: traffic per pass {tot / 1e6:.1f} MB, kernel time per pass {sum(k.get('rocprof_ms', 0) for k in ks.values()):.4f} ms")

dst = os.path.join(os.path.dirname(out.rstrip("/")), f"prof_{tag}", "hbm_traffic.json")
if not os.path.exists(dst):
    dst = os.path.join(ROOT, "profiles", "hbm_traffic.json")
d = json.load(open(dst))
if d.get("src_hash") != source_hash():
    lines.append(f"== {dst} is from source {d.get('src_hash')}, this tree is {source_hash()}: run tools/prof_round.sh {tag} first")
else:
    d["configs"] = configs
    d["configs_profile"] = f"profiles/{tag}_other_configs_kernel_stats_and_hbm.txt (tools/prof_configs.sh {tag})"
    json.dump(d, open(dst, "w"), indent=1)
    lines.append(f"== configs written into {dst}")
open(os.path.join(out, f"{tag}_configs_summary.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
