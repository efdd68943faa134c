#!/usr/bin/env python3
"""Build container only: the library's decoder back halves (pss_h_morse_decode, pss_h_ax25_frame — host code, no GPU) against the
reference's decode_morse / decode_ax25_frame on random inputs.

Morse: keyed envelopes built from random texts, speeds, jitter and dropouts (the envelope is 1 / 1e-4, so the reference's -20 dB mask IS
the keying and its edge arrays are known); the reference is run under several seeds of NumPy's global generator (scipy's kmeans draws
its starting points from it): a case counts only when all seeds agree — then text and timing must equal the library's on every bit.
AX.25: random bit streams, half of them with a stuffed frame planted.

    PYTHONPATH=/root/reference python tools/fuzz_decoders_vs_reference.py [cases]
"""
import os
import sys
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True
warnings.simplefilter("ignore")
import decoders as R                       # noqa: E402  the reference
import pyspecconst                         # noqa: E402
from pyspecsdr_amd import decoders as D    # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 400
rng = np.random.default_rng(2026)
code = {v: k for k, v in pyspecconst.MORSE_CODE.items() if len(v) == 1}
letters = sorted(code)


def keyed(text, unit, jitter, fs):
    key = [0] * int(rng.integers(0, 4))
    for wi, word in enumerate(text.split(" ")):
        if wi:
            key += [0] * 4
        for ch in word:
            for sym in code[ch]:
                key += [1] * (1 if sym == "." else 3) + [0]
            key += [0] * 2
    out = []
    for k in key:
        out += [k] * max(1, int(round(unit * (1 + jitter * rng.standard_normal()))))
    env = np.where(np.array(out) > 0, 1.0, 1e-4)
    if rng.random() < 0.3:     # start or end in the middle of a pulse
        env = env[int(rng.integers(0, unit * 2)):]
    return env.astype(np.complex64)


agree = skipped = 0
for it in range(cases):
    nw = int(rng.integers(1, 4))
    text = " ".join("".join(rng.choice(letters, int(rng.integers(1, 6)))) for _ in range(nw))
    fs = float(rng.choice([8000.0, 24000.0, 48000.0]))
    # dots of 10 .. 120 ms (120 .. 10 words per minute).  scipy's kmeans stops when its mean distance improves by less than 1e-5 SECONDS:
    # with millisecond pulses it stops before it has converged and returns whatever its random start led to first
    x = keyed(text, int(rng.integers(int(0.010 * fs), int(0.120 * fs))), float(rng.choice([0.0, 0.02, 0.1, 0.25])), fs)
    if len(x) < 4:
        continue
    refs = set()
    for seed in range(5):
        np.random.seed(seed)
        t, tm = R.decode_morse(x, fs)
        refs.add((t, float(tm["dot"]), float(tm["dash"]), float(tm["gap"])))
    env = np.abs(x)
    sig = 20 * np.log10(env / env.max() + 1e-10) > -20
    tr = np.diff(sig.astype(int))
    rise, fall = np.where(tr == 1)[0], np.where(tr == -1)[0]
    t2, m2 = D.morse_from_edges(rise, fall, fs)
    mine = (t2, float(m2["dot"]), float(m2["dash"]), float(m2["gap"]))
    if len(refs) > 1:
        skipped += 1
        continue
    ref = next(iter(refs))
    if ref != mine:
        print("MORSE MISMATCH", it, repr(text), ref, mine)
        sys.exit(1)
    agree += 1
print(f"morse: {agree} cases equal (text and dot / dash / gap on every bit), {skipped} skipped (the reference's own answer depends on its seed)")


def ax25_bits(dest, src, info):
    by = [(ord(c) << 1) for c in dest.ljust(6)] + [0x60] + [(ord(c) << 1) for c in src.ljust(6)] + [0x61, 0x03, 0xF0] + [ord(c) & 0xFF for c in info]
    bits = [(b >> j) & 1 for b in by for j in range(8)]
    st, ones = [], 0
    for b in bits:
        st.append(b)
        ones = ones + 1 if b else 0
        if ones == 5:
            st.append(0)
            ones = 0
    flag = [0, 1, 1, 1, 1, 1, 1, 0]
    return flag + st + flag


n_ok = 0
for it in range(cases * 3):
    bits = [int(b) for b in rng.integers(0, 2, int(rng.integers(0, 400)))]
    if it % 2 == 0:
        info = "".join(chr(int(c)) for c in rng.integers(0, 256, int(rng.integers(0, 30))))
        dest = "".join(chr(int(c)) for c in rng.choice([32, 9, 12, 28, 65, 66, 48, 49, 45], 6))
        pos = int(rng.integers(0, len(bits) + 1))
        bits = bits[:pos] + ax25_bits(dest, "SRC%d" % (it % 1000), info) + bits[pos:pos + int(rng.integers(0, 20))]
    a, b = R.decode_ax25_frame(bits), D.decode_ax25_frame(bits)
    if a != b:
        print("AX25 MISMATCH", it, repr(a), repr(b))
        sys.exit(1)
    n_ok += 1
print(f"ax25: {n_ok} bit streams equal (packet strings, None for no frame)")
