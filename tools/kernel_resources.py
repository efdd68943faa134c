#!/usr/bin/env python3
"""Register / LDS / scratch use of every kernel of libpss.so, read from the CODE OBJECT's metadata (hipcc --offload-device-only -S of
the two kernel units with the product build's flags), not from a profiler's dispatch record (rocprofv3's VGPR / LDS columns were wrong
for these kernels in round 2: 64 VGPRs / 0 bytes reported for k_nfm_fwd, the compiler says 128 / static 0 + 39 KB dynamic).

    python tools/kernel_resources.py  ->  profiles/kernel_resources.json  {src_hash, kernels: {mangled name: {short, vgpr, agpr, sgpr,
                                          vgpr_spill, sgpr_spill, lds_static, scratch, max_wg, waves_per_simd}}}
Runs without a GPU (cross-compile)."""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pyspecsdr_amd import build as B  # noqa: E402
import bench  # noqa: E402


def main():
    out_dir = os.path.join(B.BUILD, "asm")
    os.makedirs(out_dir, exist_ok=True)
    kernels = {}
    procs = []
    for src, extra in B.UNITS:
        if not src.endswith(".hip"):
            continue
        s = os.path.join(out_dir, src.replace(".hip", ".s"))
        cmd = [B.hipcc()] + [f for f in B.COMMON if f != "-fPIC"] + extra + ["--offload-device-only", "-S", os.path.join(B.CSRC, src), "-o", s]
        procs.append((s, subprocess.Popen(cmd, stderr=subprocess.DEVNULL)))
    for s, p in procs:
        if p.wait() != 0:
            sys.exit(f"compile failed: {s}")
        txt = open(s).read()
        for blk in re.split(r"\n  - \.agpr_count:", txt)[1:]:
            blk = "    .agpr_count:" + blk
            g = lambda k: (re.search(rf"\.{k}:\s*(\S+)", blk) or [None, None])[1]
            name = g("name")
            if not name:
                continue
            m = re.search(r"(k_[a-z_0-9]+)", name)
            vg, ag = int(g("vgpr_count") or 0), int(g("agpr_count") or 0)
            tot = vg + ag if ag else vg
            kernels[name] = {"short": m.group(1) if m else name, "vgpr": vg, "agpr": ag, "sgpr": int(g("sgpr_count") or 0),
                             "vgpr_spill": int(g("vgpr_spill_count") or 0), "sgpr_spill": int(g("sgpr_spill_count") or 0),
                             "lds_static": int(g("group_segment_fixed_size") or 0), "scratch": int(g("private_segment_fixed_size") or 0),
                             "max_wg": int(g("max_flat_workgroup_size") or 0),
                             "waves_per_simd_by_vgpr": min(8, 512 // max(8 * ((tot + 7) // 8), 8))}
    res = {"src_hash": bench.source_hash(), "note": "code-object metadata (hipcc --offload-device-only -S, product flags); dynamic LDS is a launch "
           "parameter: k_nfm_fwd 39 040 B, k_spectrum_r16<N = 1024> 70 KB, see DESIGN.md §4", "kernels": kernels}
    p = os.path.join(ROOT, "profiles", "kernel_resources.json")
    json.dump(res, open(p, "w"), indent=1, sort_keys=True)
    hot = ("k_nfm_fwd", "k_nfm_bwd", "k_spectrum_r16", "k_post_sel", "k_disp_rows", "k_spectrum_xl", "k_wfm_fwd", "k_hilbert_xl")
    for n, k in sorted(kernels.items(), key=lambda kv: kv[1]["short"]):
        if k["short"] in hot:
            print(f"{k['short']:18s} vgpr {k['vgpr']:3d} agpr {k['agpr']:3d} sgpr {k['sgpr']:3d} spills v{k['vgpr_spill']}/s{k['sgpr_spill']} "
                  f"static LDS {k['lds_static']:6d} scratch {k['scratch']:5d}  {n[:70]}")
    print(p, len(kernels), "kernels")


if __name__ == "__main__":
    main()
