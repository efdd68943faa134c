#!/usr/bin/env python3
"""Time the library on the other BASELINE.json configs (device-resident inputs, HIP-event kernel times).

    python tools/bench_configs.py [cfg3] [cfg4] [cfg5] [cfg5stream] [big] [wfm]
"""
import json
import sys
import os
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from pyspecsdr_amd import _lib as L
from pyspecsdr_amd.engine import Engine

dev = torch.device("cuda", 0)
eng = Engine(0)


def rand_iq(nf, n, scale=0.3):
    g = torch.Generator(device=dev).manual_seed(7)
    return (torch.randn((nf, n, 2), generator=g, device=dev, dtype=torch.float32) * scale + 0.2).contiguous()


def timed(name, fn, reps=5):
    fn(); eng.sync(); torch.cuda.synchronize()
    eng.enable_timing(True)
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    eng.sync(); torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / reps
    kt = {k: round(sum(v) / len(v), 4) for k, v in eng.kernel_times().items()}
    eng.enable_timing(False)
    return {"call": name, "wall_ms": round(wall * 1e3, 4), "kernel_ms": kt}


def cfg3():
    nf, n, fs = 8192, 16384, 2.4e6
    iq = rand_iq(nf, n)
    db = torch.empty((nf, n), dtype=torch.float32, device=dev)
    pcm = torch.empty((nf, n, 2), dtype=torch.int16, device=dev)
    pw = torch.empty((nf,), dtype=torch.float32, device=dev)
    out = [timed("spectrum_db 16384", lambda: eng.spectrum_db(iq, nf, n, db)),
           timed("demod AM", lambda: eng.demod(L.MODE_AM, iq, nf, n, fs, pcm, None)),
           timed("demod USB", lambda: eng.demod(L.MODE_USB, iq, nf, n, fs, pcm, None)),
           timed("power_db", lambda: eng.power_db(iq, nf, n, pw))]
    tot = sum(o["wall_ms"] for o in out)
    return {"config": "cfg3 8192 x 16384 AM+SSB+power+spectrum", "calls": out, "samples_per_s": nf * n / (tot * 1e-3)}


def cfg4():
    ns, n, fs = 8192, 4096, 2.4e6
    iq = rand_iq(ns, n)
    db = torch.empty((ns, n), dtype=torch.float32, device=dev)
    pk = torch.empty((ns,), dtype=torch.float32, device=dev)
    bw = torch.empty((ns,), dtype=torch.float64, device=dev)
    cnt = torch.empty((ns,), dtype=torch.int32, device=dev)
    o = timed("scan 4096", lambda: eng.scan(iq, ns, n, fs, db, pk, bw, cnt), reps=20)
    ms = list(o["kernel_ms"].values())[0]
    return {"config": "cfg4 8192 slices x 4096 scanner", "calls": [o], "samples_per_s": ns * n / (o["wall_ms"] * 1e-3),
            "hbm_GBs": ns * (n * 8 + n * 4 + 16) / (ms * 1e-3) / 1e9}


def cfg5():
    nf, n, fs = 48828, 2048, 10e6
    iq = rand_iq(nf, n)
    db = torch.empty((nf, n), dtype=torch.float32, device=dev)
    post = torch.empty((nf, n - 4), dtype=torch.float32, device=dev)
    n_out = eng.demod_out_len(L.MODE_NFM, n, fs)
    pcm = torch.empty((nf, n_out, 2), dtype=torch.int16, device=dev)
    out = [timed("spectrum_nfm 2048 @10MS/s", lambda: eng.spectrum_nfm(iq, nf, n, fs, db, pcm)),
           timed("spectrum_post 2048", lambda: eng.spectrum_post(db, nf, n, post))]
    return {"config": "cfg5 48828 x 2048 @10 MS/s (device-resident)", "calls": out,
            "samples_per_s": nf * n / (out[0]["wall_ms"] * 1e-3)}


def cfg5stream():
    """10 s @ 10 MS/s capture in pinned host memory -> chunked H2D / compute / D2H (PCIe-inclusive rate)."""
    nf, n, fs = 48828, 2048, 10e6
    h_iq = eng.pinned_empty((nf, n), np.complex64)
    rng = np.random.default_rng(3)
    h_iq.view(np.float32)[:] = rng.standard_normal((nf, 2 * n), dtype=np.float32) * 0.3
    n_out = eng.demod_out_len(L.MODE_NFM, n, fs)
    h_db = eng.pinned_empty((nf, n), np.float32)
    h_pcm = eng.pinned_empty((nf, n_out, 2), np.int16)
    res = []
    for chunk, with_db in ((4096, True), (4096, False), (1024, True), (16384, True)):
        eng.stream_spectrum_nfm(h_iq, fs, chunk, h_db if with_db else None, h_pcm)
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            eng.stream_spectrum_nfm(h_iq, fs, chunk, h_db if with_db else None, h_pcm)
        dt = (time.perf_counter() - t0) / reps
        res.append({"chunk_frames": chunk, "dB_rows_downloaded": with_db, "wall_ms": round(dt * 1e3, 2),
                    "samples_per_s": nf * n / dt, "h2d_GBs": nf * n * 8 / dt / 1e9})
    # the same capture through the display pipeline: per frame a waterfall line + PCM come back (BASELINE configs[4])
    outp = {"lines": (eng.pinned_empty((nf, 112), np.int8), eng.pinned_empty((nf, 112), np.int8)), "pcm": h_pcm,
            "row_lo": eng.pinned_empty((nf,), np.float32), "row_hi": eng.pinned_empty((nf,), np.float32)}
    for chunk in (4096, 8192, 16384):
        eng.stream_display_nfm(h_iq, fs, chunk, out=outp)
        t0 = time.perf_counter()
        for _ in range(3):
            eng.stream_display_nfm(h_iq, fs, chunk, out=outp)
        dt = (time.perf_counter() - t0) / 3
        res.append({"call": "stream_display_nfm", "chunk_frames": chunk, "wall_ms": round(dt * 1e3, 2), "samples_per_s": nf * n / dt,
                    "h2d_GBs": nf * n * 8 / dt / 1e9})
    return {"config": "cfg5 streamed: 48828 x 2048 @10 MS/s from pinned host memory", "runs": res}


def big():
    out = []
    for n, nf in ((32768, 2048), (65536, 256), (1 << 20, 8)):
        iq = rand_iq(nf, n)
        db = torch.empty((nf, n), dtype=torch.float32, device=dev)
        out.append(timed(f"spectrum_db {n} x{nf}", lambda: eng.spectrum_db(iq, nf, n, db), reps=2))
    n, nf, fs = 32768, 4096, 2.4e6
    iq = rand_iq(nf, n)
    n_out = eng.demod_out_len(L.MODE_NFM, n, fs)
    pcm = torch.empty((nf, n_out, 2), dtype=torch.int16, device=dev)
    out.append(timed("demod NFM 32768 x4096", lambda: eng.demod(L.MODE_NFM, iq, nf, n, fs, pcm, None), reps=2))
    return {"config": "reference-default buffer sizes", "calls": out}


def wfm():
    """The reference's default mode (pyspecsdr.py --demod WFM): iq_correction + demodulate_wfm, cfg2-sized batch."""
    nf, n, fs = 65536, 1024, 2.4e6
    iq = rand_iq(nf, n)
    n_out = eng.demod_out_len(L.MODE_WFM, n, fs)
    pcm = torch.empty((nf, n_out, 2), dtype=torch.int16, device=dev)
    corr = torch.empty((nf, n, 2), dtype=torch.float32, device=dev)
    out = [timed("iq_correction 1024", lambda: eng.iq_correction(iq, nf, n, corr, None)),
           timed("demod_signal WFM", lambda: eng.demod_signal(L.MODE_WFM, iq, nf, n, fs, pcm, None), reps=3)]
    return {"config": "WFM 65536 x 1024 @2.4 MS/s (device-resident)", "calls": out,
            "samples_per_s": nf * n / (out[1]["wall_ms"] * 1e-3)}


def classify():
    """classify_signal over a scanner sweep: cfg-4-sized slices (8192 x 4096) and the reference's own dwell (64 x 240 000)."""
    out = []
    for nf, n in ((8192, 4096), (64, 240000)):
        iq = rand_iq(nf, n)
        lab = torch.empty((nf,), dtype=torch.int32, device=dev)
        bw = torch.empty((nf,), dtype=torch.float64, device=dev)
        mi = torch.empty((nf,), dtype=torch.float32, device=dev)
        fl = torch.empty((nf,), dtype=torch.float32, device=dev)
        out.append(timed(f"classify {n} x{nf}", lambda: eng.classify(iq, nf, n, 2.4e6, lab, bw, mi, fl, None), reps=3))
    return {"config": "classify_signal (Welch PSD + modulation index + flatness + label)", "calls": out}


def sweep():
    """The sweep driver's reads (pyspecsdr.py:1022-1093): 0.1 s dwells of 240 000 samples — not a power of two, i.e. the Bluestein
    path — spectrum rows + max power + bins above a threshold, and classify_signal on the same reads."""
    nf, n = 64, 240000
    iq = rand_iq(nf, n)
    db = torch.empty((nf, n), dtype=torch.float32, device=dev)
    pk = torch.empty((nf,), dtype=torch.float32, device=dev)
    bw = torch.empty((nf,), dtype=torch.float64, device=dev)
    cnt = torch.empty((nf,), dtype=torch.int32, device=dev)
    out = [timed(f"scan_threshold {n} x{nf}", lambda: eng.scan_threshold(iq, nf, n, 2.4e6, -10.0, db, pk, bw, cnt), reps=3),
           timed(f"spectrum_db (Hamming) {n} x{nf}", lambda: eng.spectrum_db(iq, nf, n, db), reps=3)]
    return {"config": "sweep driver reads, 64 x 240000 (Bluestein, M = 2^19)", "calls": out,
            "samples_per_s": nf * n / (out[0]["wall_ms"] * 1e-3)}


if __name__ == "__main__":
    which = sys.argv[1:] or ["cfg3", "cfg4", "cfg5"]
    for w in which:
        print(json.dumps(globals()[w]()), flush=True)
