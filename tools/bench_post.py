#!/usr/bin/env python3
"""Kernel times of the post-process / batched accumulator kernels (HIP events on the library's stream).
    python tools/bench_post.py [n_frames] [n_fft ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyspecsdr_amd.engine import Engine

nf = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
sizes = [int(a) for a in sys.argv[2:]] or [1024]
e = Engine(0)
for n in sizes:
    frames = max(1, nf * 1024 // n)
    g = torch.Generator(device="cuda").manual_seed(n)
    db = torch.randn((frames, n), generator=g, device="cuda") * 6.0 - 40.0
    post = torch.empty((frames, n - 4), device="cuda")
    lo, hi = torch.empty(frames, device="cuda"), torch.empty(frames, device="cuda")
    gl, co = (torch.empty((frames, 112), dtype=torch.int8, device="cuda") for _ in range(2))
    torch.cuda.synchronize()
    for legacy in (0, 1):
        if legacy and frames * n > 1 << 27:
            continue
        try:
            e.set_option("post_legacy", legacy)     # the forcing switch exists in -DPSS_VARIANTS builds only (PSS_LIBRARY=...)
        except Exception:
            if legacy:
                continue
        for _ in range(2):
            e.spectrum_post_extremes(db, frames, n, post, lo, hi)
        e.sync()
        e.enable_timing(True)
        for _ in range(5):
            e.spectrum_post_extremes(db, frames, n, post, lo, hi)
            e.waterfall_rows(post, frames, n - 4, lo, hi, 112, gl, co)
        e.sync()
        kt = e.kernel_times()
        e.enable_timing(False)
        byt = frames * (n * 4 + (n - 4) * 4)
        out = {k: sum(v) / len(v) for k, v in kt.items()}
        print(f"n={n} frames={frames} legacy={legacy}: " + "  ".join(f"{k}={v:.4f} ms" for k, v in out.items())
              + f"   k_post: {byt / out['k_post'] / 1e9:.2f} TB/s" if "k_post" in out else "")
    try:
        e.set_option("post_legacy", 0)
    except Exception:
        pass
