#!/usr/bin/env python3
"""The cfg 2 step on float32 rows (pss_frame_pipeline_nfm) and on float64 rows (pss_frame_pipeline_nfm_f64, the cell-exact step), A/B in one
process: ms per step (wall, fenced) and the mean launch time of every kernel (HIP events, separate pass).
    python tools/time_pipeline.py [frames] [opt=val ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from pyspecsdr_amd.engine import Engine  # noqa: E402


def main():
    nf = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 65536
    n, fs, W = 1024, 2.4e6, 112
    eng = Engine(0, order="none")
    for a in sys.argv[1:]:
        if "=" in a:
            k, v = a.split("=")
            eng.set_option(k, int(v))
    dev = torch.device("cuda", 0)
    iq = bench.synth_fm_iq(nf, n, fs, dev, seed=5)
    torch.cuda.synchronize()
    out = {}
    for name, dt in (("f32", torch.float32), ("f64", torch.float64)):
        db = torch.empty((nf, n), dtype=dt, device=dev)
        lo, hi = torch.empty(nf, dtype=dt, device=dev), torch.empty(nf, dtype=dt, device=dev)
        g, c = torch.empty((nf, W), dtype=torch.int8, device=dev), torch.empty((nf, W), dtype=torch.int8, device=dev)
        pcm = torch.empty((nf, 10, 2), dtype=torch.int16, device=dev)
        fn = eng.frame_pipeline_nfm if name == "f32" else eng.frame_pipeline_nfm_f64
        call = (lambda fn=fn, db=db, lo=lo, hi=hi, g=g, c=c, pcm=pcm: fn(iq, nf, n, fs, db, None, lo, hi, W, g, c, pcm))
        out[name] = (call, (db, lo, hi, g, c, pcm))
    for rep in range(3):
        for name in ("f32", "f64"):
            call = out[name][0]
            for _ in range(3):
                call()
            eng.sync()
            t = time.perf_counter()
            for _ in range(20):
                call()
            eng.sync()
            ms = (time.perf_counter() - t) / 20 * 1e3
            eng.enable_timing(True)
            for _ in range(5):
                call()
            eng.sync()
            kt = {k: round(sum(v) / len(v), 4) for k, v in eng.kernel_times().items()}
            eng.enable_timing(False)
            print(f"{name}: {ms:.4f} ms/step  {kt}", flush=True)
    # each chain kernel alone
    eng.enable_timing(True)
    for name in ("f32", "f64"):
        db, lo, hi, g, c, pcm = out[name][1]
        for _ in range(6):
            (eng.spectrum_db if name == "f32" else eng.spectrum_db_f64)(iq, nf, n, db)
        eng.sync()
        kt = {k: round(sum(v[1:]) / len(v[1:]), 4) for k, v in eng.kernel_times().items()}
        print(f"{name} alone: {kt}", flush=True)
    eng.enable_timing(False)


if __name__ == "__main__":
    main()
