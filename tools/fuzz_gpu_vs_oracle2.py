#!/usr/bin/env python3
"""Second GPU fuzz set (GPU box): display quantisers, post-process, scanner slice, band-pass rows and the AFSK bit slicer on
random inputs, HIP path vs the C oracle.    python tools/fuzz_gpu_vs_oracle2.py [cases]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import oracle_lib as O
import gpu_util as G

rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", "5")))
e = G.engine()
bad = 0
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
for it in range(cases):
    L = int(rng.choice([60, 252, 1020, 2044, 8188])); H = int(rng.integers(12, 60)); W = int(rng.integers(30, 220))
    nr = int(rng.integers(1, 31))
    rows = rng.standard_normal((nr, L)) * rng.uniform(0.5, 8) - rng.uniform(5, 70)
    rows[:, L // 4:L // 4 + 9] += rng.uniform(3, 45)
    d_rows = G.dev(np.ascontiguousarray(rows))
    d_gl, d_co = G.empty((H, W), torch.int8), G.empty((H, W), torch.int8)
    e.waterfall_cells(d_rows, nr, L, H, W, d_gl, d_co, f64=True); e.sync()
    og, oc = O.waterfall_cells(rows, H, W)
    if not (np.array_equal(G.host(d_gl), og) and np.array_equal(G.host(d_co), oc)): bad += 1; print("WATERFALL", nr, L, H, W)
    e.gradient_cells(d_rows, nr, L, H, W, d_gl, d_co, f64=True); e.sync()
    og, oc = O.gradient_cells(rows, H, W)
    if not (np.array_equal(G.host(d_gl), og) and np.array_equal(G.host(d_co), oc)): bad += 1; print("GRADIENT", nr, L, H, W)
    np_ = min(nr, 10)
    e.persistence_cells(G.dev(np.ascontiguousarray(rows[:np_])), np_, L, H, W, d_co, f64=True); e.sync()
    if not np.array_equal(G.host(d_co), O.persistence_cells(rows[:np_], H, W)): bad += 1; print("PERSISTENCE", np_, L, H, W)
    d_rg = G.empty((1, 2), torch.float64)
    e.spectrogram_cells(G.dev(np.ascontiguousarray(rows[0])), 1, L, H, W, d_gl, d_co, d_rg, f64=True); e.sync()
    og, oc, _, _ = O.spectrogram_cells(rows[0], H, W)
    if not (np.array_equal(G.host(d_gl), og) and np.array_equal(G.host(d_co), oc)): bad += 1; print("SPECTROGRAM", L, H, W, int((G.host(d_gl) != og).sum()))
    if H >= 4 and W >= 10:
        e.surface_cells(G.dev(np.ascontiguousarray(rows[0])), L, H, W, d_co, f64=True); e.sync()
        if not np.array_equal(G.host(d_co), O.surface_cells(rows[0], H, W)): bad += 1; print("SURFACE", L, H, W)
    vs = ((rng.standard_normal(700) + 1j * rng.standard_normal(700)) * rng.uniform(0.2, 2)).astype(np.complex64)
    e.vector_cells(G.dev(vs), len(vs), H, W, d_co); e.sync()
    if not np.array_equal(G.host(d_co), O.vector_cells(vs, H, W)): bad += 1; print("VECTOR", H, W)
    # band-pass rows + AFSK
    n = int(rng.integers(40, 5000)); nrw = int(rng.choice([1, 3, 64, 70]))
    a = rng.standard_normal((nrw, n)); fs = float(rng.choice([22050.0, 48000.0]))
    import scipy.signal as ss  # coefficient tables only
    nyq = fs / 2
    s1 = ss.butter(5, [1100 / nyq, 1300 / nyq], btype="band", output="sos"); s2 = ss.butter(5, [2100 / nyq, 2300 / nyq], btype="band", output="sos")
    d_y = G.empty((nrw, n), torch.float64)
    e.sosfilt(G.dev(a), nrw, n, s1, d_y); e.sync()
    y = G.host(d_y)
    for r in (0, nrw - 1):
        if not np.array_equal(y[r], O.sosfilt(s1, a[r])): bad += 1; print("SOSFILT", nrw, n, r)
    nb = e.afsk_n_bits(n, fs)
    if nb:
        d_b = G.empty((nrw, nb), torch.uint8)
        e.afsk_bits(G.dev(a), nrw, n, fs, d_b, s1, s2); e.sync()
        b = G.host(d_b)
        for r in (0, nrw - 1):
            if not np.array_equal(b[r], O.afsk_bits(a[r], fs, s1, s2)): bad += 1; print("AFSK", nrw, n, fs, r)
    # classify_signal: a few reads of a random length (below 1024: one Welch segment of that length), every output against the oracle
    n = int(rng.choice([1024, 1025, 1536, 2048, 3001, 4096, 9000, 20000, 1023, 700, 512, 257, 100, 31, 6, 2])); nf = int(rng.integers(1, 9)); fs = float(rng.choice([2.4e6, 1.024e6, 250e3]))
    t = np.arange(n) / fs
    iq = np.empty((nf, n), np.complex64)
    for f in range(nf):
        k = int(rng.integers(0, 4)); off = rng.uniform(-0.45, 0.45) * fs; nz = 10.0 ** rng.uniform(-4, -0.7)
        ph = 2 * np.pi * (0.0, 3e3, 6e4, 2e5)[k] * np.cumsum(np.sin(2 * np.pi * rng.uniform(200, 9e3) * t)) / fs + 2 * np.pi * off * t
        iq[f] = ((0.0 if k == 0 and f % 2 else 0.5) * np.exp(1j * ph) + nz * (rng.standard_normal(n) + 1j * rng.standard_normal(n))).astype(np.complex64)
    d_lab, d_bw = G.empty((nf,), torch.int32), G.empty((nf,), torch.float64)
    d_mi, d_fl, d_psd = G.empty((nf,), torch.float32), G.empty((nf,), torch.float32), G.empty((nf, 1024), torch.float32)
    e.classify(G.dev(iq.view(np.float32).reshape(nf, n, 2)), nf, n, fs, d_lab, d_bw, d_mi, d_fl, d_psd); e.sync()
    lab, bw, mi, fl, psd = (G.host(a) for a in (d_lab, d_bw, d_mi, d_fl, d_psd))
    for f in range(nf):
        ol, ob, om, of, op = O.classify(iq[f], fs)
        if not (O.CLASS_LABELS[lab[f]] == ol and bw[f] == ob and mi[f].tobytes() == om.tobytes()
                and (abs(float(fl[f]) - float(of)) <= 1e-5 * abs(float(of)) or float(fl[f]) == float(of)) and np.all(np.abs(psd[f, :len(op)] - op) <= 1e-6 * (op + 1e-10))):
            bad += 1; print("CLASSIFY", nf, n, fs, f, (O.CLASS_LABELS[lab[f]], ol), (bw[f], ob), (mi[f], om), (fl[f], of))
print("cases", cases, "bad", bad)
