#!/usr/bin/env python3
"""Mean launch time of the spectrum / post-process kernels standalone at BASELINE cfg-2 size (HIP events); with
PSS_LIBRARY=<variant> for side-by-side comparisons of kernel variants on the same box."""
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
db = torch.empty((nf, n), dtype=torch.float32, device=dev)
post = torch.empty((nf, n - 4), dtype=torch.float32, device=dev)
lo, hi = torch.empty(nf, dtype=torch.float32, device=dev), torch.empty(nf, dtype=torch.float32, device=dev)
for rnd in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    out = []
    for name, fn in (("k_spectrum", lambda: e.spectrum_db(iq, nf, n, db)), ("k_post", lambda: e.spectrum_post_extremes(db, nf, n, post, lo, hi))):
        for _ in range(5):
            fn()
        e.sync()
        e.enable_timing(True)
        for _ in range(30):
            fn()
        e.sync()
        v = e.kernel_times()[name]
        e.enable_timing(False)
        out.append(f"{name} {sum(v) / len(v):.4f} (min {min(v):.4f})")
    print(os.environ.get("PSS_LIBRARY", "libpss.so"), " ".join(out), "checksum", float(db.double().sum()))
