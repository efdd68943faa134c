#!/usr/bin/env python3
"""Mean launch time of k_nfm_fwd / k_nfm_bwd in a plain NFM demodulate call at BASELINE cfg-2 size (HIP events);
used with PSS_LIBRARY=<variant> to compare kernel variants on the same box."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench
from pyspecsdr_amd.engine import Engine

dev = torch.device("cuda", 0)
e = Engine(0)
nf, n = bench.N_FRAMES, bench.N_FFT
iq = bench.synth_fm_iq(nf, n, bench.FS, dev, seed=20260930)
n_out = e.demod_out_len(0, n, bench.FS)
pcm = torch.empty((nf, n_out, 2), dtype=torch.int16, device=dev)
for rnd in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    for _ in range(3):
        e.demod(0, iq, nf, n, bench.FS, pcm, None)
    e.sync()
    e.enable_timing(True)
    for _ in range(30):
        e.demod(0, iq, nf, n, bench.FS, pcm, None)
    e.sync()
    kt = e.kernel_times()
    e.enable_timing(False)
    print(os.environ.get("PSS_LIBRARY", "libpss.so"), " ".join(f"{k} {sum(v) / len(v):.4f} (min {min(v):.4f})" for k, v in kt.items()), "checksum", int(pcm.to(torch.int64).sum()))
