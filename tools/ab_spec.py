#!/usr/bin/env python3
"""A/B timing of the spectrum / scanner kernels across builds of the library in ONE process, interleaved round by round (tools/ab_fwd.py
explains why).   python tools/ab_spec.py [--scan] [--sizes 1024,2048,4096] lib[:opt=val,...] ...
Each size runs 2^26 samples per launch (65536 x 1024, 32768 x 2048, ...); prints mean / min ms and algorithmic TB/s (12 B per sample)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = [a for a in sys.argv]
import torch

from ab_fwd_common import engine_for

args = sys.argv[1:]
scan = "--scan" in args
args = [a for a in args if a != "--scan"]
sizes = [1024, 2048, 4096, 8192, 16384]
if "--sizes" in args:
    i = args.index("--sizes")
    sizes = [int(x) for x in args[i + 1].split(",")]
    del args[i:i + 2]
specs = args or ["product"]
dev = torch.device("cuda", 0)
engs = [engine_for(s) for s in specs]
tot = 1 << 26
g = torch.Generator(device=dev).manual_seed(3)
iq = torch.randn((tot, 2), generator=g, device=dev, dtype=torch.float32) * 0.3
iq[:, 0] += 0.4
db = torch.empty((tot,), dtype=torch.float32, device=dev)
pk = torch.empty((tot // 256,), dtype=torch.float32, device=dev)
bw = torch.empty((tot // 256,), dtype=torch.float64, device=dev)
cnt = torch.empty((tot // 256,), dtype=torch.int32, device=dev)
torch.cuda.synchronize()
for n in sizes:
    nf = tot // n
    times = {s: [] for s in specs}
    sums = {}

    def go(e):
        if scan:
            e.scan(iq, nf, n, 2.4e6, db, pk, bw, cnt)
        else:
            e.spectrum_db(iq, nf, n, db)

    for e in engs:
        for _ in range(3):
            go(e)
        e.sync()
    for rnd in range(8):
        for s, e in zip(specs, engs):
            e.enable_timing(True)
            for _ in range(6):
                go(e)
            e.sync()
            times[s] += e.kernel_times().get("k_spectrum", [])
            e.enable_timing(False)
            sums[s] = float(db.double().sum())
    for s in specs:
        v = times[s]
        m = sum(v) / len(v)
        print(f"n={n:6d} {'scan' if scan else 'spec'} {s:30s} mean {m:.4f}  min {min(v):.4f} ms  {tot * 12 / m / 1e9:5.2f} TB/s = {tot * 12 / m / 1e9 / 8:.3f} of 8 TB/s   checksum {sums[s]:.6e}")
