#!/usr/bin/env python3
"""The cell-exact cfg 2 step with the transform and the post-process as two kernels (option fuse_post = 0) against the fused kernel
k_spectrum_post (pss_spec_post.h), A/B in one process.  Variants:
    two64    pss_frame_pipeline_nfm_f64, fuse_post = 0   (round 5's step: k_spectrum_r16<D64> -> k_post_sel<double>)
    fus64    pss_frame_pipeline_nfm_f64, fuse_post = 1   (fused, float64 rows written)
    fus32    pss_frame_pipeline_cells,   fuse_post = 1   (fused, float32 rows written: the step bench.py times)
    fus3264  pss_frame_pipeline_cells with both row types written
    two32    pss_frame_pipeline_cells,   fuse_post = 0   (two kernels + conversion pass)
Outputs compared bit for bit with two64's; ms per step (wall, fenced; three interleaved rounds), the mean launch time of every kernel (HIP
events, separate pass), and the display half alone (pss_spectrum_cells: no demodulator beside it).
    python tools/ab_fuse.py [frames] [opt=val ...]      FUSE_VARIANTS=two64,fus32 to choose"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from pyspecsdr_amd import _lib as L  # noqa: E402
from pyspecsdr_amd.engine import Engine  # noqa: E402

VARIANTS = {"two64": (0, False, True), "fus64": (1, False, True), "fus32": (1, True, False), "fus3264": (1, True, True), "two32": (0, True, False)}


def main():
    nf = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 65536
    n, fs, W = 1024, 2.4e6, 112
    eng = Engine(0, order="none")
    for a in sys.argv[1:]:
        if "=" in a:
            k, v = a.split("=")
            eng.set_option(k, int(v))
    names = os.environ.get("FUSE_VARIANTS", "two64,fus64,fus32,fus3264,two32").split(",")
    dev = torch.device("cuda", 0)
    iq = bench.synth_fm_iq(nf, n, fs, dev, seed=5)
    torch.cuda.synchronize()
    bufs = {}
    for v in names:
        fuse, r32, r64 = VARIANTS[v]
        bufs[v] = dict(db64=torch.zeros((nf, n), dtype=torch.float64, device=dev) if r64 else None,
                       db32=torch.zeros((nf, n), dtype=torch.float32, device=dev) if r32 else None,
                       lo=torch.zeros(nf, dtype=torch.float64, device=dev), hi=torch.zeros(nf, dtype=torch.float64, device=dev),
                       g=torch.zeros((nf, W), dtype=torch.int8, device=dev), c=torch.zeros((nf, W), dtype=torch.int8, device=dev),
                       pcm=torch.zeros((nf, 10, 2), dtype=torch.int16, device=dev))
    torch.cuda.synchronize()

    def call(v, demod=True):
        fuse, r32, r64 = VARIANTS[v]
        b = bufs[v]
        eng.set_option("fuse_post", fuse)
        if not demod:
            eng.spectrum_cells(iq, nf, n, b["db32"], b["db64"], b["lo"], b["hi"], W, b["g"], b["c"], window=30)
        elif r32:
            eng.frame_pipeline_cells(L.MODE_NFM, iq, nf, n, fs, b["db32"], b["db64"], b["lo"], b["hi"], W, b["g"], b["c"], b["pcm"], window=30)
        else:
            eng.frame_pipeline_nfm_f64(iq, nf, n, fs, b["db64"], None, b["lo"], b["hi"], W, b["g"], b["c"], b["pcm"])

    for v in names:
        call(v)
    eng.sync()
    ref = {k: (x.cpu().numpy() if x is not None else None) for k, x in bufs[names[0]].items()}
    import hashlib
    print("sha256[:16] of", names[0], {k: hashlib.sha256(np.ascontiguousarray(x)).hexdigest()[:16] for k, x in ref.items() if x is not None and k != "pcm"},
          "(compare across libraries: PSS_LIBRARY=...)", flush=True)
    for v in names[1:]:
        got = {k: (x.cpu().numpy() if x is not None else None) for k, x in bufs[v].items()}
        res = {}
        for k in ("lo", "hi", "g", "c", "pcm"):
            res[k] = bool(np.array_equal(ref[k], got[k]))
        if got["db64"] is not None:
            res["db64"] = bool(np.array_equal(ref["db64"], got["db64"]))
        if got["db32"] is not None:
            res["db32"] = bool(np.array_equal(ref["db64"].astype(np.float32), got["db32"]))
        print(f"{v} vs {names[0]}: {res}", flush=True)
    for rep in range(3):
        for v in names:
            for _ in range(3):
                call(v)
            eng.sync()
            t = time.perf_counter()
            for _ in range(20):
                call(v)
            eng.sync()
            ms = (time.perf_counter() - t) / 20 * 1e3
            eng.enable_timing(True)
            for _ in range(5):
                call(v)
            eng.sync()
            kt = {k: round(sum(x) / len(x), 4) for k, x in eng.kernel_times().items()}
            eng.enable_timing(False)
            print(f"{v:8s}: {ms:.4f} ms/step  {kt}", flush=True)
    for v in names:
        for _ in range(3):
            call(v, False)
        eng.sync()
        t = time.perf_counter()
        for _ in range(20):
            call(v, False)
        eng.sync()
        ms = (time.perf_counter() - t) / 20 * 1e3
        eng.enable_timing(True)
        for _ in range(5):
            call(v, False)
        eng.sync()
        kt = {k: round(sum(x) / len(x), 4) for k, x in eng.kernel_times().items()}
        eng.enable_timing(False)
        print(f"{v:8s} display half alone: {ms:.4f} ms  {kt}", flush=True)


if __name__ == "__main__":
    main()
