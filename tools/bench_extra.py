#!/usr/bin/env python3
"""Side benchmarks outside the BASELINE.json configs (device-resident inputs, HIP-event kernel times): the reference's largest read
buffers, classify_signal over a sweep, the sweep driver's 240 000-sample reads.

    python tools/bench_extra.py [big] [classify] [sweep]
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: F401
import torch

from pyspecsdr_amd import _lib as L
from pyspecsdr_amd.engine import Engine

dev = torch.device("cuda", 0)
eng = Engine(0)


def rand_iq(nf, n, scale=0.3):
    g = torch.Generator(device=dev).manual_seed(7)
    x = (torch.randn((nf, n, 2), generator=g, device=dev, dtype=torch.float32) * scale + 0.2).contiguous()
    torch.cuda.synchronize()
    return x


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
    for w in sys.argv[1:] or ["big", "classify", "sweep"]:
        print(json.dumps(globals()[w]()), flush=True)
