#!/usr/bin/env python3
"""The opt-in NFM forward kernel with the FIR on the matrix pipe (option "fir_mfma") against the exact default kernel:
int16 / float64 differences over many random FM frames, and the forward kernel's time.
    python tools/mfma_variant_stats.py [batches of 65536 frames]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from pyspecsdr_amd.engine import Engine

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda", 0)
e = Engine(0)
e.set_option("small_batch", 0)
nf, n, fs = 65536, 1024, 2.4e6
n_out = e.demod_out_len(0, n, fs)
tot = diff = 0
maxrel = 0.0
maxlsb = 0
times = {0: [], 1: []}
for b in range(nb):
    iq = bench.synth_fm_iq(nf, n, fs, dev, seed=777 + b)
    torch.cuda.synchronize()
    res = {}
    for mf in (0, 1):
        e.set_option("fir_mfma", mf)
        pcm = torch.empty((nf, n_out, 2), dtype=torch.int16, device=dev)
        au = torch.empty((nf, n_out), dtype=torch.float64, device=dev)
        e.demod(0, iq, nf, n, fs, pcm, au)
        e.sync()
        e.enable_timing(True)
        for _ in range(5):
            e.demod(0, iq, nf, n, fs, pcm, au)
        e.sync()
        times[mf] += e.kernel_times()["k_nfm_fwd"]
        e.enable_timing(False)
        res[mf] = (pcm, au)
    e.set_option("fir_mfma", 0)
    d = (res[0][0][..., 0].int() - res[1][0][..., 0].int()).abs()
    tot += d.numel()
    diff += int((d != 0).sum())
    maxlsb = max(maxlsb, int(d.max()))
    rel = (res[0][1] - res[1][1]).abs().max() / 0.95       # relative to the frame peak (every frame is normalised to 0.95)
    maxrel = max(maxrel, float(rel))
print(f"frames {nb * nf} x {n}: int16 samples differing {diff} of {tot} (max {maxlsb} LSB); float64 audio max difference relative to the frame peak {maxrel:.3e}")
print(f"k_nfm_fwd exact {sum(times[0]) / len(times[0]):.4f} ms   fir_mfma {sum(times[1]) / len(times[1]):.4f} ms   (65536 x 1024 frames)")
