#!/usr/bin/env python3
"""Fuzz (build container only): classify_signal — the reference with scipy.signal.welch bound to the name it forgets to
import — against the oracle on random reads: labels, bandwidth, modulation index (bit-exact), flatness, PSD."""
import os, sys, warnings
sys.path.insert(0, '/root/reference'); sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np, scipy.signal as ss
import signal_processing as sp
import oracle_lib as O
sp.welch = ss.welch
warnings.simplefilter('ignore')
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 777)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
bad = labels = 0
seen = {}
for it in range(N):
    n = int(rng.choice([1024, 1025, 1535, 1536, 2048, 3000, 4096, 8191, 16384, 33000, 70000,
                        1023, 1000, 640, 513, 512, 300, 129, 64, 33, 17, 9, 4, 3, 2]))   # below 1024: welch's nperseg = n fallback
    fs = float(rng.choice([2.4e6, 1.024e6, 250e3, 10e6, 48e3]))
    t = np.arange(n) / fs
    kind = int(rng.integers(0, 7))
    noise = 10.0 ** rng.uniform(-4, -0.5)
    off = rng.uniform(-0.45, 0.45) * fs
    if kind == 0:   x = 0.5 * np.exp(2j * np.pi * off * t)
    elif kind == 1: x = 0.5 * np.exp(1j * (2 * np.pi * rng.uniform(1e3, 0.2 * fs) * np.cumsum(np.sin(2 * np.pi * rng.uniform(100, 0.01 * fs) * t)) / fs + 2 * np.pi * off * t))
    elif kind == 2: x = np.zeros(n)
    elif kind == 3: x = (1 + 0.7 * np.sin(2 * np.pi * rng.uniform(100, 5e3) * t)) * 0.4 * np.exp(2j * np.pi * off * t)
    elif kind == 4: x = 0.5 * np.exp(2j * np.pi * (off + rng.uniform(300, 3000)) * t) + 0.3 * np.exp(2j * np.pi * off * t)
    elif kind == 5: x = rng.choice([-1, 1], n) * 0.5 + 1j * rng.choice([-1, 1], n) * 0.5      # QPSK-like chips
    else:           x = np.round((rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 20) / 127.0   # 8-bit style, exact zeros
    x = (x + noise * (rng.standard_normal(n) + 1j * rng.standard_normal(n)) * (kind != 6)).astype(np.complex64)
    f, psd = sp.welch(x, fs=fs, nperseg=1024)
    rbw = float(sp.estimate_bandwidth(psd, f)); rmi = sp.estimate_modulation_index(x)
    rfl = np.exp(np.mean(np.log(psd + 1e-10))) / np.mean(psd); rl = sp.classify_signal(x, fs, 0.0)
    lab, bw, mi, fl, opsd = O.classify(x, fs)
    seen[rl] = seen.get(rl, 0) + 1
    ok_mi = mi.tobytes() == np.float32(rmi).tobytes() or (np.isnan(mi) and np.isnan(rmi))
    ok_fl = float(fl) == float(rfl) or abs(float(fl) - float(rfl)) <= 1e-5 * abs(float(rfl)) or (np.isnan(fl) and np.isnan(rfl))
    # the reference's segment FFT is single precision: its error on a bin of power p next to a peak P is ~1e-7 sqrt(p P)
    ok_psd = np.all(np.abs(opsd - psd) <= 1e-5 * (psd + 1e-10) + 1e-6 * np.sqrt(psd * np.max(psd)))
    if lab != rl: labels += 1
    if not (lab == rl and bw == rbw and ok_mi and ok_fl and ok_psd):
        bad += 1
        print("MISMATCH", it, n, fs, kind, (rl, lab), (rbw, bw), (rmi, mi), (rfl, fl), float(np.max(np.abs(opsd - psd) / (psd + 1e-10))))
print("classify fuzz:", N, "cases,", bad, "mismatches,", labels, "label differences; labels seen:", seen)
