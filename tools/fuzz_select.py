#!/usr/bin/env python3
"""GPU post-process (5-tap smoothing, np.median, clamp, finite extremes) against NumPy, bit for bit, on random dB rows of many shapes — the
campaign behind the compaction select of one-wavefront rows (pss_post.h select_kth64_compact / select_kth32_compact, NOTEBOOK R6-14).
Row families per case: noise of random spread, values on a random grid (ties), runs (piecewise constant), two modes, a cluster of values
inside one float64 high word, skewed rows, rows with NaN / +-inf, constant rows; row lengths over everything the register select serves.
    FUZZ_SEED=1 python tools/fuzz_select.py [cases]     -> "cases N rows R bad B" """
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import gpu_util as G  # noqa: E402


def make_rows(rng, nf, n):
    rows = np.empty((nf, n))
    for k in range(nf):
        kind = rng.integers(0, 10)
        base = rng.uniform(-120, 20)
        if kind == 0:
            r = base + rng.uniform(0.01, 12) * rng.standard_normal(n)
        elif kind == 1:
            step = rng.choice([0.05, 0.25, 0.6, 1.0, 3.0])
            r = base + step * rng.integers(-6, 7, n)
        elif kind == 2:
            r = np.empty(n)
            pos = 0
            while pos < n:
                ln = min(int(rng.integers(1, 60)), n - pos)
                r[pos:pos + ln] = base + rng.choice([0.3, 0.6, 5.0]) * rng.integers(-4, 5)
                pos += ln
        elif kind == 3:
            r = np.where(rng.random(n) < rng.uniform(0.3, 0.7), base + rng.standard_normal(n), base - rng.uniform(5, 40) + 0.2 * rng.standard_normal(n))
        elif kind == 4:
            r = base + 5.6 * rng.standard_normal(n)
            idx = rng.permutation(n)[: int(rng.integers(2, n))]
            r[idx] = base + 10.0 ** rng.uniform(-13, -6) * rng.standard_normal(idx.size)
        elif kind == 5:
            r = base - rng.uniform(0.5, 6) * np.log(rng.random(n))
        elif kind == 6:
            r = base + 5.6 * rng.standard_normal(n)
            for _ in range(int(rng.integers(1, 4))):
                r[rng.integers(0, n)] = rng.choice([np.nan, np.inf, -np.inf])
        elif kind == 7:
            r = np.full(n, base if rng.random() < 0.5 else np.round(base))
        elif kind == 8:
            r = base + 0.2 * rng.standard_normal(n)            # everything inside the first bracket
        else:
            r = base + 5.6 * rng.standard_normal(n)
            r[: n // 2] = np.round(r[: n // 2] * 4e4) / 4e4 + rng.integers(0, 3, n // 2) * 1e-12
        rows[k] = r
    return rows


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", "1")))
    e = G.engine()
    bad = total = 0
    for case in range(cases):
        n = int(rng.choice([8, 12, 64, 128, 256, 260, 512, 516, 1000, 1024, 1028, 2048, 2052, 4096]))
        nf = int(rng.integers(1, 200))
        rows = make_rows(rng, nf, n)
        m = n - 4
        with np.errstate(invalid="ignore", over="ignore"):
            sm = rows[:, 0:m] * 0.2
            for j in range(1, 5):
                sm = sm + rows[:, j:j + m] * 0.2
            med = np.median(sm, axis=1)
            want = sm.copy()
            for k in range(nf):
                want[k][want[k] < med[k] - 10] = med[k] - 10
        d_p, d_lo, d_hi = G.empty((nf, m), torch.float64), G.empty((nf,), torch.float64), G.empty((nf,), torch.float64)
        e.spectrum_post_f64(G.dev(rows), nf, n, d_p, d_lo, d_hi)
        e.sync()
        got, lo, hi = G.host(d_p), G.host(d_lo), G.host(d_hi)
        for k in range(nf):
            ok = np.array_equal(got[k], want[k], equal_nan=True)
            fin = want[k][np.isfinite(want[k])]
            ok = ok and ((lo[k] == fin.min() and hi[k] == fin.max()) if fin.size else (lo[k] == np.inf and hi[k] == -np.inf))
            if not ok:
                bad += 1
                print("MISMATCH float64", case, n, k, flush=True)
        # float32 rows: the float32 rounding of the float64 sums, the median of those, the threshold rounded to float32
        rows32 = rows.astype(np.float32)
        with np.errstate(invalid="ignore", over="ignore"):
            s32 = rows32[:, 0:m].astype(np.float64) * 0.2
            for j in range(1, 5):
                s32 = s32 + rows32[:, j:j + m].astype(np.float64) * 0.2
            s32 = s32.astype(np.float32)
            want32 = s32.copy()
            for k in range(nf):
                md = np.median(s32[k].astype(np.float64))
                thr = np.float32(md - 10.0)
                want32[k][want32[k] < thr] = thr
        d_p, d_lo, d_hi = G.empty((nf, m), torch.float32), G.empty((nf,), torch.float32), G.empty((nf,), torch.float32)
        e.spectrum_post_extremes(G.dev(rows32), nf, n, d_p, d_lo, d_hi)
        e.sync()
        got, lo, hi = G.host(d_p), G.host(d_lo), G.host(d_hi)
        for k in range(nf):
            if np.isnan(s32[k]).any():
                continue          # (float32 rows holding a NaN: not pinned — the float32 path is the 1e-4 spectrum contract's, the reference's rows are float64)
            ok = np.array_equal(got[k], want32[k], equal_nan=True)
            fin = want32[k][np.isfinite(want32[k])]
            if fin.size:
                ok = ok and lo[k] == fin.min() and hi[k] == fin.max()
            if not ok:
                bad += 1
                print("MISMATCH float32", case, n, k, flush=True)
        total += 2 * nf
    print(f"cases {cases} rows {total} bad {bad}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
