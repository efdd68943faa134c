#!/usr/bin/env python3
"""Single-frame latency of the drop-in module (host buffer in, host buffer out, synchronous) — what the reference's
interactive loop would see per call — next to the C oracle on one host core.
    python tools/shim_latency.py [n ...]      (default: 1024 32768)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import pyspecsdr_amd.signal_processing as sp
sp.CLASSIFY_RAISES_NAMEERROR = False  # time the function as written
import oracle_lib as O


def timeit(fn, reps):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [1024, 32768]
    fs = 2.4e6
    rng = np.random.default_rng(1)
    e = sp.get_engine()
    for n in sizes:
        ph = np.cumsum(rng.standard_normal(n) * 0.1)
        x = (0.5 * np.exp(1j * ph) + 0.02 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))).astype(np.complex64)
        taps, sos, zi = e.nfm_filters(fs)
        am = np.empty((5, 6)); e.lib.pss_am_bandpass_sos(am.ctypes.data)
        reps = 20 if n <= 4096 else 5
        rows = [
            ("compute_fft", lambda: sp.compute_fft(x), lambda: O.compute_fft(x)),
            ("measure_signal_power", lambda: sp.measure_signal_power(x), lambda: O.power_db(x)),
            ("demodulate_signal NFM", lambda: sp.demodulate_signal(x, fs, "NFM"), lambda: O.demod_nfm(x, fs, taps, sos, zi)),
            ("demodulate_signal AM", lambda: sp.demodulate_signal(x, fs, "AM"), lambda: O.demod_am(x, am)),
            ("demodulate_signal USB", lambda: sp.demodulate_signal(x, fs, "USB"), lambda: O.demod_ssb(x, e.ssb_taps(fs))),
            ("demodulate_signal WFM", lambda: sp.demodulate_signal(x, fs, "WFM"), None),
            ("demodulate_signal RAW", lambda: sp.demodulate_signal(x, fs, "RAW"), lambda: O.iq_correction(x)),
            ("classify_signal", lambda: sp.classify_signal(x, fs, 0.0), lambda: O.classify(x, fs)),
            ("decode_morse (edges)", lambda: e.h_morse_edges(x), lambda: O.morse_edges(x)),
        ]
        for name, g, c in rows:
            try:
                tg = timeit(g, reps)
            except Exception as ex:                      # e.g. compute_fft on a buffer that is not a power of two
                print(f"n={n:6d}  {name:24s} not applicable ({type(ex).__name__})", flush=True)
                continue
            tc = timeit(c, reps) if c else float("nan")
            print(f"n={n:6d}  {name:24s} GPU shim {tg:8.3f} ms   C oracle (1 core, filters given) {tc:8.3f} ms", flush=True)


if __name__ == "__main__":
    main()
