#!/usr/bin/env python3
"""One read buffer through the drop-in module, REPS times — to be run under `rocprofv3 --kernel-trace --stats` for the per-kernel
durations behind tools/shim_latency.py's single-buffer lines.
    python tools/trace_single.py NFM|AM|USB|WFM [n] [reps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

import pyspecsdr_amd.signal_processing as sp

mode = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
rng = np.random.default_rng(1)
ph = np.cumsum(rng.standard_normal(n) * 0.1)
x = (0.5 * np.exp(1j * ph) + 0.02 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))).astype(np.complex64)
sp.demodulate_signal(x, 2.4e6, mode)
t0 = time.perf_counter()
for _ in range(reps):
    sp.demodulate_signal(x, 2.4e6, mode)
print(f"{mode} n={n}: {(time.perf_counter() - t0) / reps * 1e3:.3f} ms per call (under the profiler)")
