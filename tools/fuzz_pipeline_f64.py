#!/usr/bin/env python3
"""Shape fuzz of the cell-exact step (GPU box): pss_frame_pipeline_f64 / pss_frame_pipeline_nfm_f64 on random frame lengths, batch sizes,
history lengths, display geometries, signals and a random cut into two calls joined by the extremes halo — display cells, row extremes
and NFM PCM against the oracle's own step FROM THE IQ (float64 rows).    FUZZ_SEED=7 python tools/fuzz_pipeline_f64.py [cases]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import gpu_util as G
import oracle_lib as O
from pyspecsdr_amd import _lib as L

rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", "7")))
e = G.engine()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
bad = 0


def signal(kind, nf, n, fs):
    t = np.arange(n)
    if kind == "fm":
        ph = np.cumsum(2 * np.pi * 5e3 / fs * np.sin(2 * np.pi * rng.uniform(300, 3000, (nf, 1)) * t / fs + rng.uniform(0, 6, (nf, 1))), axis=1)
        x = rng.uniform(0.05, 0.9, (nf, 1)) * np.exp(1j * ph)
    elif kind == "tones":
        x = sum(rng.uniform(0.01, 0.5) * np.exp(2j * np.pi * (rng.integers(0, n) / n) * t + 1j * rng.uniform(0, 6, (nf, 1))) for _ in range(3))
    elif kind == "noise":
        x = np.zeros((nf, n), complex)
    else:  # "steps": frames whose level jumps (rows whose extremes move the history's range)
        x = np.exp(2j * np.pi * 0.11 * t)[None, :] * (10.0 ** rng.integers(-4, 1, (nf, 1)))
    x = x + rng.uniform(1e-4, 0.05) * (rng.standard_normal((nf, n)) + 1j * rng.standard_normal((nf, n)))
    return x.astype(np.complex64)


for it in range(cases):
    n = int(rng.choice([256, 512, 1024, 1024, 2048, 4096]))
    nf = int(rng.integers(1, 260))
    fs = float(rng.choice([2.4e6, 10e6, 1.024e6]))
    display = str(rng.choice(["waterfall", "persistence"]))
    window = int(rng.integers(1, 41))
    W, H = int(rng.integers(8, 300)), int(rng.integers(4, 100))
    mode = int(rng.choice([L.MODE_NFM, L.MODE_NFM, L.MODE_AM, L.MODE_USB, L.MODE_WFM]))
    kind = str(rng.choice(["fm", "tones", "noise", "steps"]))
    iq = signal(kind, nf, n, fs)
    taps, sos, zi = e.nfm_filters(fs)
    o = O.headline_f64(iq, fs, taps, sos, zi, window, W, O.threads_available(), pcm=(mode == L.MODE_NFM), display=display, disp_h=H)
    want = (o["glyph"], o["colour"]) if display == "waterfall" else (o["glyph"],)
    d_iq = G.dev(iq)
    n_out = e.demod_out_len(mode, n, fs)
    d_db = G.empty((nf, n), torch.float64)
    lo, hi = G.empty((nf,), torch.float64), G.empty((nf,), torch.float64)
    a, b = G.empty((nf, W), torch.int8), torch.zeros((nf, W), dtype=torch.int8, device="cuda")
    pcm = G.empty((nf, max(n_out, 1), 2), torch.int16)
    cut = int(rng.integers(1, nf)) if nf > 1 and rng.random() < 0.6 else nf
    tag = (it, n, nf, fs, display, window, W, H, mode, kind, cut)
    try:
        e.frame_pipeline_f64(mode, d_iq[:cut], cut, n, fs, d_db[:cut], None, lo, hi, W, a[:cut], b[:cut], pcm[:cut], window=window, display=display, disp_h=H)
        if cut < nf:
            e.frame_pipeline_f64(mode, d_iq[cut:], nf - cut, n, fs, d_db[cut:], None, lo, hi, W, a[cut:], b[cut:], pcm[cut:], n_halo=cut,
                                 window=window, display=display, disp_h=H)
        e.sync()
    except Exception as ex:  # noqa: BLE001
        bad += 1
        print("RAISED", tag, ex)
        continue
    got = (G.host(a), G.host(b)) if display == "waterfall" else (G.host(a),)
    ok = all(np.array_equal(g, w) for g, w in zip(got, want))
    ok = ok and np.allclose(G.host(lo), o["lo"], rtol=0, atol=1e-9) and np.allclose(G.host(hi), o["hi"], rtol=0, atol=1e-9)
    if mode == L.MODE_NFM:
        ok = ok and np.array_equal(G.host(pcm), o["pcm"])
    if not ok:
        bad += 1
        print("MISMATCH", tag, [int(np.count_nonzero(g != w)) for g, w in zip(got, want)])
print("cases", cases, "bad", bad)
