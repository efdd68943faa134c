#!/usr/bin/env python3
"""Digest a tools/prof_round.sh output directory into gpurun_out/prof_<tag>/{<tag>_summary.txt, hbm_traffic.json,
<tag>_kernel_stats.csv} — the files to copy into profiles/.  Everything is keyed by the bench step's LARGE launch of each
kernel (the small side launches of bench.py's untimed sections are told apart by grid size)."""
import collections
import csv
import hashlib
import json
import os
import re
import subprocess
import sys

out, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_FRAMES = 65536


def source_hash():
    h = hashlib.sha256()
    d = os.path.join(ROOT, "pyspecsdr_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h", ".cpp")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def kname(full):
    m = re.search(r"(k_[a-z_0-9]+)", full)
    return None if (not m or "at::" in full) else m.group(1)


def git_head():
    try:
        return subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip() or "n/a"
    except Exception:  # noqa: BLE001
        return "n/a (no .git on the GPU box)"


lines = [f"== profile {tag}: source hash {source_hash()} (sha256 over pyspecsdr_amd/csrc), git {git_head()}"]
p = os.path.join(out, "kt_kernel_stats.csv")
if os.path.exists(p):
    lines.append("== rocprofv3 --kernel-trace --stats : python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-side")
    lines.append(f"{'kernel':28s} {'calls':>6s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>7s}")
    rows = list(csv.DictReader(open(p)))
    with open(os.path.join(out, f"{tag}_kernel_stats.csv"), "w") as f:
        f.write(open(p).read())
    for r in rows:
        n = kname(r["Name"])
        if n:
            lines.append(f"{n:28s} {r['Calls']:>6s} {float(r['AverageNs']) / 1e3:10.1f} {float(r['MinNs']) / 1e3:10.1f} "
                         f"{float(r['MaxNs']) / 1e3:10.1f} {float(r['Percentage']):7.2f}")
# per-dispatch duration of the step's large launches (the stats above average a kernel's small side launches in)
p = os.path.join(out, "kt_kernel_trace.csv")
big_ms = {}
if os.path.exists(p):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(p)):
        n = kname(r["Kernel_Name"])
        if n:
            acc[(n, int(r["Grid_Size_X"]))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    lines.append("== per launch shape (kernel, grid size): launches, mean ms")
    for (n, g), v in sorted(acc.items()):
        lines.append(f"{n:28s} grid={g:9d} launches={len(v):4d} mean={sum(v) / len(v):8.4f} ms")
        if n not in big_ms or g > big_ms[n][0]:
            big_ms[n] = (g, sum(v) / len(v))

p = os.path.join(out, "alone_kernel_trace.csv")
if os.path.exists(p):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(p)):
        n = kname(r["Kernel_Name"])
        if n:
            acc[(n, int(r["Grid_Size_X"]))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    lines.append("== standalone launches (rocprofv3 --kernel-trace: python tools/bench_alone.py; its own lines below): kernel, grid, launches, mean ms")
    for (n, g), v in sorted(acc.items()):
        lines.append(f"{n:28s} grid={g:9d} launches={len(v):4d} mean={sum(v) / len(v):8.4f} ms  min={min(v):8.4f} ms")
    lg = os.path.join(out, "alone.log")
    if os.path.exists(lg):
        lines += [l.rstrip() for l in open(lg) if l.startswith("alone:")]

# register / LDS use from the code object (tools/kernel_resources.py -> profiles/kernel_resources.json), NOT from the profiler's dispatch
# record: rocprofv3's VGPR_Count / LDS_Block_Size columns were wrong for these kernels (round 2: 64 / 0 reported for k_nfm_fwd)
res_by_short = collections.defaultdict(list)
try:
    kr = json.load(open(os.path.join(ROOT, "profiles", "kernel_resources.json")))
    if kr.get("src_hash") == source_hash():
        for full, k in kr["kernels"].items():
            res_by_short[k["short"]].append(k)
    else:
        lines.append(f"== profiles/kernel_resources.json is from source {kr.get('src_hash')}: rerun tools/kernel_resources.py")
except Exception:  # noqa: BLE001
    pass


def code_object(n):
    ks = res_by_short.get(n)
    if not ks:
        return {}
    v = [k["vgpr"] for k in ks]
    return {"vgpr_code_object": max(v) if min(v) == max(v) else [min(v), max(v)], "vgpr_spill": max(k["vgpr_spill"] for k in ks),
            "sgpr_spill": max(k["sgpr_spill"] for k in ks), "lds_static_bytes": max(k["lds_static"] for k in ks),
            "scratch_bytes": max(k["scratch"] for k in ks), "instantiations": len(ks)}


digest = {"src_hash": source_hash(), "git": git_head(), "n_frames": N_FRAMES,
          "profile": f"profiles/{tag}_summary.txt (tools/prof_round.sh {tag})", "kernels": {}}
meta = {}
traffic = collections.defaultdict(dict)
for name, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    p = os.path.join(out, f"{name}_counter_collection.csv")
    if not os.path.exists(p):
        continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(p)):
        n = kname(r["Kernel_Name"])
        if n and r["Counter_Name"] == ctr:
            acc[(n, int(float(r["Grid_Size"])))].append(float(r["Counter_Value"]))
            meta[n] = {"workgroup": r.get("Workgroup_Size"), "lds_bytes_dispatch": r.get("LDS_Block_Size"), **code_object(n)}
    lines.append(f"== rocprofv3 --pmc {ctr} (KiB per dispatch, by launch shape; gfx950: FETCH_SIZE under-reports wide coalesced "
                 f"reads by 2x — MI355X_MICROARCH.md §HBM)")
    best = {}
    for (n, g), v in sorted(acc.items()):
        lines.append(f"{n:28s} grid={g:9d} dispatches={len(v):4d}  {ctr}={sum(v) / len(v):14.1f} KiB  ({sum(v) / len(v) * 1024 / 1e6:10.2f} MB)")
        if n not in best or g > best[n][0]:
            best[n] = (g, sum(v) / len(v) * 1024)
    for n, (g, b) in best.items():
        traffic[n][ctr] = b
for n, t in traffic.items():
    f, w = t.get("FETCH_SIZE"), t.get("WRITE_SIZE")
    if f is None or w is None:
        continue
    digest["kernels"][n] = {"fetch_size_bytes_raw": f, "fetch_bytes_corrected_2x": 2 * f, "write_size_bytes": w,
                            "traffic_bytes": 2 * f + w, **meta.get(n, {})}
    if n in big_ms:
        digest["kernels"][n]["rocprof_ms"] = big_ms[n][1]
# duration of the large launches inside the SQ counter pass itself (counters serialise dispatches: not the bench's timing)
sq_ms = {}
p = os.path.join(out, "sqa_kernel_trace.csv")
if os.path.exists(p):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(p)):
        n = kname(r["Kernel_Name"])
        if n:
            acc[(n, int(r["Grid_Size_X"]))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    for (n, g), v in acc.items():
        if n not in sq_ms or g > sq_ms[n][0]:
            sq_ms[n] = (g, sum(v) / len(v))
# SQ counters: totals per kernel over its large launches
sq = collections.defaultdict(lambda: collections.defaultdict(list))
for name in ("sqa", "sqb"):
    p = os.path.join(out, f"{name}_counter_collection.csv")
    if not os.path.exists(p):
        continue
    for r in csv.DictReader(open(p)):
        n = kname(r["Kernel_Name"])
        if n:
            sq[(n, int(float(r["Grid_Size"])))][r["Counter_Name"]].append(float(r["Counter_Value"]))
sq_lines = [f"== SQ counters per launch (mean over dispatches), python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-side; source hash {source_hash()}"]
seen = {}
for (n, g), d in sorted(sq.items()):
    v = {k: sum(x) / len(x) for k, x in d.items()}
    sq_lines.append(f"{n} grid={g} vgpr(code object)={meta.get(n, {}).get('vgpr_code_object')} static lds={meta.get(n, {}).get('lds_static_bytes')}")
    wc = v.get("SQ_WAVE_CYCLES", 0)
    for k in sorted(v):
        extra = f"  ({v[k] / wc:.3f} of wave cycles)" if wc and k.startswith(("SQ_WAIT", "SQ_ACTIVE")) else ""
        sq_lines.append(f"    {k:26s} {v[k]:14.5e}{extra}")
    if n not in seen or g > seen[n]:
        seen[n] = g
        if n in digest["kernels"] and wc:
            waves = v.get("SQ_WAVES", 0)
            digest["kernels"][n]["valu_active_frac_of_wave_cycles"] = v.get("SQ_ACTIVE_INST_VALU", 0) / wc
            digest["kernels"][n]["valu_insts"] = v.get("SQ_INSTS_VALU")
            digest["kernels"][n]["waves"] = waves
            digest["kernels"][n]["wave_cycles"] = wc                                   # SQ counts in units of 4 clocks
            digest["kernels"][n]["valu_active_cycles"] = v.get("SQ_ACTIVE_INST_VALU", 0)
            if n in sq_ms:
                digest["kernels"][n]["sq_pass_ms"] = sq_ms[n][1]
        if n in digest["kernels"]:
            for k in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_CVT"):
                if k in v:
                    digest["kernels"][n][k.lower()] = v[k]
for lg in ("kt.log",):
    p = os.path.join(out, lg)
    if os.path.exists(p):
        for l in open(p):
            if l.startswith('{"metric"'):
                j = json.loads(l)
                lines.append("== bench line of the kernel-trace run")
                lines.append(json.dumps({k: j[k] for k in ("value", "unit", "ms_per_step", "roofline")}))
open(os.path.join(out, f"{tag}_summary.txt"), "w").write("\n".join(lines) + "\n")
open(os.path.join(out, f"{tag}_sq_counters.txt"), "w").write("\n".join(sq_lines) + "\n")
json.dump(digest, open(os.path.join(out, "hbm_traffic.json"), "w"), indent=1)
print("\n".join(lines))
print(json.dumps(digest, indent=1)[:3000])
