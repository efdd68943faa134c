#!/usr/bin/env python3
"""Standalone launches of the kernels of the bench step at BASELINE cfg-2 size (and the big-N spectrum at cfg 3),
each alone on the machine, on the dB rows / IQ the step itself would see: algorithmic bytes / mean launch time (HIP events).
Run under rocprofv3 by tools/prof_round.sh; prints one "alone:" line per kernel."""
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
torch.cuda.synchronize()
db = torch.empty((nf, n), dtype=torch.float32, device=dev)
post = torch.empty((nf, n - 4), dtype=torch.float32, device=dev)
lo, hi = torch.empty(nf, dtype=torch.float32, device=dev), torch.empty(nf, dtype=torch.float32, device=dev)
g, c = (torch.empty((nf, bench.DISP_W), dtype=torch.int8, device=dev) for _ in range(2))


def run(name, fn, kernel, nbytes, reps=8):
    for _ in range(2):
        fn()
    e.sync()
    e.enable_timing(True)
    for _ in range(reps):
        fn()
    e.sync()
    v = e.kernel_times()[kernel]
    e.enable_timing(False)
    ms = sum(v) / len(v)
    print(f"alone: {name:44s} {ms:8.4f} ms  {nbytes / ms / 1e9:6.2f} TB/s algorithmic = {nbytes / ms / 1e9 / 8.0:5.3f} of the 8 TB/s HBM peak")


# float32 rows (pss_spectrum_db and the separate post-process / display entry points: rows written)
run("k_spectrum_r16 65536 x 1024 (FM IQ), f32 rows", lambda: e.spectrum_db(iq, nf, n, db), "k_spectrum", nf * (n * 8 + n * 4))
run("k_post_sel 65536 x 1024 f32 rows in + out", lambda: e.spectrum_post_extremes(db, nf, n, post, lo, hi), "k_post", nf * (n * 4 + (n - 4) * 4 + 8))
run("k_disp_rows 65536 x 1020 f32 -> 112 cells", lambda: e.waterfall_rows(post, nf, n - 4, lo, hi, bench.DISP_W, g, c), "k_disp_rows",
    nf * ((n - 4) * 4 + 2 * bench.DISP_W))
# float64 rows (the bench step's row type: the reference's own)
db64 = torch.empty((nf, n), dtype=torch.float64, device=dev)
post64 = torch.empty((nf, n - 4), dtype=torch.float64, device=dev)
lo64, hi64 = torch.empty(nf, dtype=torch.float64, device=dev), torch.empty(nf, dtype=torch.float64, device=dev)
run("k_spectrum_r16 65536 x 1024 (FM IQ), f64 rows", lambda: e.spectrum_db_f64(iq, nf, n, db64), "k_spectrum", nf * (n * 8 + n * 8))
run("k_post_sel 65536 x 1024 f64 rows in + out", lambda: e.spectrum_post_f64(db64, nf, n, post64, lo64, hi64), "k_post", nf * (n * 8 + (n - 4) * 8 + 16))
# the fused transform + post-process kernel of the timed step (pss_spectrum_cells: float32 row, extremes, resampled row out), alone
run("k_spectrum_post 65536 x 1024 (FM IQ), f32 rows", lambda: e.spectrum_cells(iq, nf, n, db, None, lo64, hi64, bench.DISP_W, g, c, window=bench.WF_WINDOW),
    "k_spectrum_post", nf * bench.algo_bytes(4)["k_spectrum_post"])
run("k_spectrum_post 65536 x 1024 (FM IQ), f64 rows", lambda: e.spectrum_cells(iq, nf, n, None, db64, lo64, hi64, bench.DISP_W, g, c, window=bench.WF_WINDOW),
    "k_spectrum_post", nf * bench.algo_bytes(8)["k_spectrum_post"])
del db64, post64
n_out = e.demod_out_len(0, n, bench.FS)
pcm = torch.empty((nf, n_out, 2), dtype=torch.int16, device=dev)
run("k_nfm_fwd 65536 x 1024 (demodulator alone)", lambda: e.demod(0, iq, nf, n, bench.FS, pcm, None), "k_nfm_fwd", nf * bench.ALGO_BYTES["k_nfm_fwd"])
run("k_nfm_bwd 65536 x 1024 (demodulator alone)", lambda: e.demod(0, iq, nf, n, bench.FS, pcm, None), "k_nfm_bwd", nf * 8616)   # y_fwd read back
del iq, db, post, pcm
for n2, f2 in ((16384, 8192), (8192, 16384), (4096, 32768), (2048, 32768)):
    x = torch.randn((f2, n2, 2), device=dev) * 0.1
    d = torch.empty((f2, n2), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    run(f"k_spectrum {f2} x {n2}", lambda: e.spectrum_db(x, f2, n2, d), "k_spectrum", f2 * n2 * 12)
    del x, d
