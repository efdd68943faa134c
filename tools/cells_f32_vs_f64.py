#!/usr/bin/env python3
"""How many waterfall cells of the float32-row step (pss_frame_pipeline_nfm) differ from the cell-exact float64-row step
(pss_frame_pipeline_nfm_f64 — whose cells equal the oracle's from IQ: tests) over the WHOLE bench batch (65 536 x 1024, 2 x 7.3 M cells):
    python tools/cells_f32_vs_f64.py [seed ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from pyspecsdr_amd.engine import Engine  # noqa: E402

nf, n, fs, W = 65536, 1024, 2.4e6, 112
dev = torch.device("cuda", 0)
eng = Engine(0)
for seed in [int(a) for a in sys.argv[1:]] or [20260930, 777, 1]:
    iq = bench.synth_fm_iq(nf, n, fs, dev, seed=seed)
    out = {}
    for name, dt in (("f32", torch.float32), ("f64", torch.float64)):
        db = torch.empty((nf, n), dtype=dt, device=dev)
        lo, hi = torch.empty(nf, dtype=dt, device=dev), torch.empty(nf, dtype=dt, device=dev)
        g, c = torch.empty((nf, W), dtype=torch.int8, device=dev), torch.empty((nf, W), dtype=torch.int8, device=dev)
        pcm = torch.empty((nf, 10, 2), dtype=torch.int16, device=dev)
        (eng.frame_pipeline_nfm if name == "f32" else eng.frame_pipeline_nfm_f64)(iq, nf, n, fs, db, None, lo, hi, W, g, c, pcm)
        eng.sync()
        out[name] = (g, c, pcm)
        del db
    dg = int((out["f32"][0] != out["f64"][0]).sum())
    dc = int((out["f32"][1] != out["f64"][1]).sum())
    print(f"seed {seed}: glyph cells differing {dg} of {nf * W}, colour cells differing {dc} of {nf * W} "
          f"({(dg + dc) / (2.0 * nf * W):.2e} of all cells); PCM equal: {bool(torch.equal(out['f32'][2], out['f64'][2]))}", flush=True)
