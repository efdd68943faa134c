#!/usr/bin/env python3
"""A/B timing of k_nfm_fwd across several builds of the library IN ONE PROCESS, interleaved round by round (run-to-run drift of the
box — clocks, temperature — is ~4 %, more than most kernel changes: only paired measurements resolve them).

    python tools/ab_fwd.py [--frames F] [--n N] [lib.so[:opt=val,...]] ...      (default: the product library; name 'X' = pyspecsdr_amd/libpss_X.so)
Prints per build the mean / min launch time of k_nfm_fwd at BASELINE cfg 2 over all rounds, and the PCM checksum."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench
from pyspecsdr_amd import _lib as L
from pyspecsdr_amd.engine import Engine


from ab_fwd_common import engine_for

args = sys.argv[1:]
nf, n = bench.N_FRAMES, bench.N_FFT
if "--frames" in args:                    # other batch sizes / frame lengths: --frames F --n N
    i = args.index("--frames"); nf = int(args[i + 1]); del args[i:i + 2]
if "--n" in args:
    i = args.index("--n"); n = int(args[i + 1]); del args[i:i + 2]
specs = args or ["product"]
dev = torch.device("cuda", 0)
iq = bench.synth_fm_iq(nf, n, bench.FS, dev, seed=20260930)
pcm = torch.empty((nf, 10, 2), dtype=torch.int16, device=dev)
engs = [engine_for(s) for s in specs]
times = {s: [] for s in specs}
sums = {}
for e in engs:
    for _ in range(5):
        e.demod(0, iq, nf, n, bench.FS, pcm, None)
    e.sync()
for rnd in range(12):
    for s, e in zip(specs, engs):
        e.enable_timing(True)
        for _ in range(8):
            e.demod(0, iq, nf, n, bench.FS, pcm, None)
        e.sync()
        times[s] += e.kernel_times().get("k_nfm_fwd", [])
        e.enable_timing(False)
        sums[s] = int(pcm.to(torch.int64).sum())
for s in specs:
    v = times[s]
    print(f"{s:28s} k_nfm_fwd mean {sum(v) / len(v):.4f}  min {min(v):.4f} ms over {len(v)} launches  checksum {sums[s]}")
