#!/usr/bin/env python3
"""A few launches of the display half alone (pss_spectrum_cells: k_spectrum_post + the two line kernels) on the benchmark's FM IQ, or of the
two-kernel form (fuse_post=0) — the command the SQ-counter passes of tools/prof_select.sh wrap.
    python tools/run_cells_alone.py [frames] [fuse_post=0]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from pyspecsdr_amd.engine import Engine  # noqa: E402

nf = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 65536
n, fs, W = 1024, 2.4e6, 112
eng = Engine(0, order="none")
for a in sys.argv[1:]:
    if "=" in a:
        k, v = a.split("=")
        eng.set_option(k, int(v))
dev = torch.device("cuda", 0)
iq = bench.synth_fm_iq(nf, n, fs, dev, seed=5)
db32 = torch.zeros((nf, n), dtype=torch.float32, device=dev)
db64 = torch.zeros((nf, n), dtype=torch.float64, device=dev)
lo, hi = torch.zeros(nf, dtype=torch.float64, device=dev), torch.zeros(nf, dtype=torch.float64, device=dev)
g, c = torch.zeros((nf, W), dtype=torch.int8, device=dev), torch.zeros((nf, W), dtype=torch.int8, device=dev)
torch.cuda.synchronize()
for _ in range(4):
    eng.spectrum_cells(iq, nf, n, db32, db64 if "fuse_post=0" in sys.argv else None, lo, hi, W, g, c, window=30)
eng.sync()
