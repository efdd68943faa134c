#!/usr/bin/env python3
"""Per-workgroup start / end times of k_nfm_fwd at cfg-2 size (variant build with -DPSS_UBENCH: the kernel stamps s_memrealtime):
how much of a launch is dispatch skew and tail, how long does a workgroup itself take?"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("PSS_LIBRARY", os.path.join(ROOT, "pyspecsdr_amd", "libpss_ubench.so"))
import numpy as np
import torch

import bench
from pyspecsdr_amd.engine import Engine

e = Engine(0)
lib = e.lib
lib.pss_ubench_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
dev = torch.device("cuda", 0)
nf, n = bench.N_FRAMES, bench.N_FFT
iq = bench.synth_fm_iq(nf, n, bench.FS, dev, seed=1)
pcm = torch.empty((nf, 10, 2), dtype=torch.int16, device=dev)
for _ in range(5):
    e.demod(0, iq, nf, n, bench.FS, pcm, None)
e.sync()
lib.pss_ubench_stamps(e.h, None, 0, 1)
e.demod(0, iq, nf, n, bench.FS, pcm, None)
e.sync()
nwg = int(sys.argv[1]) if len(sys.argv) > 1 else 256
st = np.zeros(4 * nwg, np.uint64)
lib.pss_ubench_stamps(e.h, st.ctypes.data, nwg, 0)
st = st.reshape(nwg, 4)
t0 = st[:, 0].min()
us = lambda c: (st[:, c] - t0).astype(np.float64) / 100.0
start, end, fill, head = us(0), us(1), us(2), us(3)
dur = end - start
print(f"{nwg} workgroups: first start 0, last start {start.max():.1f} us, first end {end.min():.1f} us, last end {end.max():.1f} us")
print(f"workgroup duration: min {dur.min():.1f}  median {np.median(dur):.1f}  max {dur.max():.1f} us")
print(f"window fill done (median) {np.median(fill - start):.1f} us after start; head dots done {np.median(head - start):.1f} us; chunk loop + tail {np.median(end - head):.1f} us")
step = max(nwg // 8, 1)
print("duration by dispatch order:", " ".join(f"{dur[i:i + step].mean():.1f}" for i in range(0, nwg, step)))
