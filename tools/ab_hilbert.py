#!/usr/bin/env python3
"""Timing of pss_hilbert with and without option "hilbert_exact" (pocketfft's butterfly order, pss_hilbert_pf.h), and of demodulate_ssb at
cfg 3 (8192 frames x 16384 samples) both ways.   python tools/ab_hilbert.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from pyspecsdr_amd.engine import Engine
from pyspecsdr_amd import _lib as L

dev = torch.device("cuda", 0)
e = Engine(0)
tot = 1 << 25
g = torch.Generator(device=dev).manual_seed(3)
x = torch.randn((tot,), generator=g, device=dev, dtype=torch.float64)
out = torch.empty((tot,), dtype=torch.complex128, device=dev)


def timed(fn, reps=6):
    for _ in range(2):
        fn()
    e.sync()
    e.enable_timing(True)
    for _ in range(reps):
        fn()
    e.sync()
    kt = e.kernel_times()
    e.enable_timing(False)
    return {k: sum(v) / reps for k, v in kt.items()}


for n in (256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 131072, 1048576):
    rows = tot // n
    res = []
    for ex in (0, 1):
        e.set_option("hilbert_exact", ex)
        t = timed(lambda: e.hilbert(x, rows, n, out))
        res.append(sum(t.values()))
    e.set_option("hilbert_exact", 0)
    gb = tot * 24 / 1e9
    print(f"hilbert n={n:6d} rows={rows:7d}  register transform {res[0]:.3f} ms ({gb / res[0]:.2f} TB/s)   pocketfft order {res[1]:.3f} ms ({gb / res[1]:.2f} TB/s)   x{res[1] / res[0]:.2f}")

nf, n, fs = 8192, 16384, 2.4e6
iq = torch.randn((nf, n, 2), generator=g, device=dev, dtype=torch.float32) * 0.3
pcm = torch.empty((nf, n, 2), dtype=torch.int16, device=dev)
for ex in (0, 1):
    e.set_option("hilbert_exact", ex)
    t = timed(lambda: e.demod(L.MODE_USB, iq, nf, n, fs, pcm, None), reps=4)
    print(f"cfg 3 USB hilbert_exact={ex}: " + "  ".join(f"{k} {v:.3f}" for k, v in t.items()) + f"   total {sum(t.values()):.3f} ms")
e.set_option("hilbert_exact", 0)
