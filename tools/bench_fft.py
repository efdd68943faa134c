#!/usr/bin/env python3
"""Kernel time of pss_spectrum_db per frame length (HIP events on the library's stream); options as key=value.
    python tools/bench_fft.py 16384 8192 [fft_big_scratch=1]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyspecsdr_amd.engine import Engine

sizes = [int(a) for a in sys.argv[1:] if "=" not in a] or [16384]
opts = [a.split("=") for a in sys.argv[1:] if "=" in a]
e = Engine(0)
for k, v in opts:
    e.set_option(k, int(v))
for n in sizes:
    frames = (1 << 27) // n
    g = torch.Generator(device="cuda").manual_seed(n)
    iq = torch.randn((frames, n, 2), generator=g, device="cuda") * 0.1
    db = torch.empty((frames, n), device="cuda")
    torch.cuda.synchronize()
    for _ in range(2):
        e.spectrum_db(iq, frames, n, db)
    e.sync()
    e.enable_timing(True)
    for _ in range(5):
        e.spectrum_db(iq, frames, n, db)
    e.sync()
    kt = e.kernel_times()
    e.enable_timing(False)
    tot = sum(sum(v) / len(v) for v in kt.values())
    print(f"n={n} frames={frames} {dict((k, round(sum(v) / len(v), 4)) for k, v in kt.items())}  total {tot:.4f} ms = "
          f"{frames * n * 12 / tot / 1e9:.2f} TB/s algorithmic", dict(opts))
