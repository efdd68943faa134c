#!/usr/bin/env python3
"""Mean / min launch times of the kernels of ONE demodulate call at a given batch shape (HIP events), for the library PSS_LIBRARY selects:

    PSS_LIBRARY=pyspecsdr_amd/libpss_x.so PSS_OPTS="am_lds_handoff=1" python tools/time_demod.py USB 8192 16384 [fs] [reps]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from pyspecsdr_amd import _lib as L
from pyspecsdr_amd.engine import Engine

mode = {"NFM": L.MODE_NFM, "AM": L.MODE_AM, "USB": L.MODE_USB, "LSB": L.MODE_LSB, "WFM": L.MODE_WFM}[sys.argv[1]]
nf, n = int(sys.argv[2]), int(sys.argv[3])
fs = float(sys.argv[4]) if len(sys.argv) > 4 else 2.4e6
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 10
e = Engine(0, order="none")
for kv in os.environ.get("PSS_OPTS", "").split(","):      # PSS_OPTS="am_lds_handoff=1,small_batch=0"
    if kv:
        e.set_option(kv.split("=")[0], int(kv.split("=")[1]))
g = torch.Generator(device="cuda").manual_seed(7)
iq = torch.randn((nf, n, 2), generator=g, device="cuda", dtype=torch.float32) * 0.3 + 0.2
n_out = e.demod_out_len(mode, n, fs)
pcm = torch.empty((nf, n_out, 2), dtype=torch.int16, device="cuda")
torch.cuda.synchronize()
for _ in range(2):
    e.demod(mode, iq, nf, n, fs, pcm, None)
e.sync()
e.enable_timing(True)
e.kernel_times()
for _ in range(reps):
    e.demod(mode, iq, nf, n, fs, pcm, None)
e.sync()
kt = e.kernel_times()
tot = sum(sum(v) for v in kt.values()) / reps
print(os.environ.get("PSS_LIBRARY", "product"), sys.argv[1], nf, n, " ".join(f"{k} mean {sum(v) / len(v):.4f} min {min(v):.4f}" for k, v in kt.items()),
      f"| sum {tot:.4f} ms | checksum {int(pcm.view(torch.int16).to(torch.int64).sum())}")
