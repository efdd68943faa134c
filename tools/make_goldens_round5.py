#!/usr/bin/env python3
"""Golden vectors for the arguments round 5 opened up (tests/golden/args.npz) — made like tools/make_goldens.py: the build container imports
the reference's own modules from /root/reference, feeds them seeded synthetic inputs and stores DATA only (inputs, outputs, SciPy's filter
tables).  Kept in its own script so that the fixtures of rounds 1-4 stay byte-identical when regenerated.

    demodulate_nfm / demodulate_wfm with target_rate != 22050   (signal_processing.py:91, :111-112, :119: int(sample_rate / target_rate))
    decode_morse with threshold != -20                           (decoders.py:136, :156)
    bandpass_filter on complex and on 2-D input                  (signal_processing.py:34-42: sosfilt along the last axis)

    python tools/make_goldens_round5.py
"""
import os
import sys
import warnings

sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import numpy as np
import scipy.signal as ss

import make_goldens as mg            # the generators' helpers (fm_iq, wfm_iq, stamp, save); importing it stubs nothing and writes nothing
import signal_processing as sp      # the reference hot path


def main():
    d = {}
    # ---- NFM / WFM at other target rates
    cases = [("n1", "nfm", 1024, 2.4e6, 11025, 3, 41), ("n2", "nfm", 4096, 2.4e6, 48000, 2, 42), ("n3", "nfm", 2048, 1.024e6, 8000, 2, 43),
             ("n4", "nfm", 1000, 250e3, 44100, 2, 44), ("w1", "wfm", 4096, 2.4e6, 44100, 2, 45), ("w2", "wfm", 2048, 1.024e6, 48000, 2, 46)]
    tags = []
    for tag, kind, n, fs, tr, nf, seed in cases:
        iq = mg.fm_iq(nf, n, fs, seed) if kind == "nfm" else mg.wfm_iq(nf, n, fs, seed)
        fn = sp.demodulate_nfm if kind == "nfm" else sp.demodulate_wfm
        aud = np.stack([fn(f, fs, tr) for f in iq])                      # (nf, n_out, 2)
        q = int(fs / tr)
        sos = ss.cheby1(8, 0.05, 0.8 / q, output="sos")                   # decimate()'s filter at this factor
        d[f"iq_{tag}"], d[f"fs_{tag}"], d[f"tr_{tag}"], d[f"audio_{tag}"] = iq, np.array(fs), np.array(tr), aud
        d[f"pcm_{tag}"] = np.int16(aud * 32767)
        d[f"taps_{tag}"] = ss.firwin(numtaps=65, cutoff=15000 / (fs / 2))
        d[f"sos_{tag}"], d[f"zi_{tag}"] = sos, ss.sosfilt_zi(sos)
        if kind == "wfm":
            nyq = fs / 2
            d[f"lp_{tag}"] = ss.butter(5, 15000 / nyq, btype="low", output="sos")
            d[f"pil_{tag}"] = ss.butter(5, [18800 / nyq, 19200 / nyq], btype="band", output="sos")
            d[f"lmr_{tag}"] = ss.butter(5, [23000 / nyq, 53000 / nyq], btype="band", output="sos")
            d[f"alpha_{tag}"] = np.array(np.exp(-1 / (75e-6 * fs)))
        tags.append(tag)
    d["rate_tags"] = np.array(tags)
    # ---- decode_morse at other thresholds (np.random seeded: scipy's kmeans draws its starting centroids from the global state)
    import decoders
    rng = np.random.default_rng(505)
    code = {v: k for k, v in __import__("pyspecconst").MORSE_CODE.items() if len(v) == 1}

    def cw(text, fs, unit, n, noise, soft):
        key = [0] * 3
        for wi, word in enumerate(text.split(" ")):
            if wi: key += [0] * 4
            for ch in word:
                for sym in code[ch]:
                    key += [1] * (1 if sym == "." else 3) + [0]
                key += [0] * 2
        k = np.repeat(np.array(key, float), unit)[:n]
        k = np.concatenate([k, np.zeros(n - len(k))])
        if soft: k = np.convolve(k, np.ones(soft) / soft, mode="same")
        t = np.arange(n)
        return (0.6 * k * np.exp(2j * np.pi * 700.0 / fs * t) + noise * (rng.standard_normal(n) + 1j * rng.standard_normal(n))).astype(np.complex64)

    mt = []
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for tag, x, fs, thr in (("t15", cw("SOS K", 24000.0, 900, 40000, 0.01, 60), 24000.0, -15),
                                ("t30", cw("TEST", 24000.0, 700, 30000, 0.004, 200), 24000.0, -30.5),
                                ("t6", cw("CQ", 48000.0, 1000, 50000, 0.02, 400), 48000.0, -6.25),
                                ("t3", cw("E E", 24000.0, 800, 12000, 0.05, 300), 24000.0, -3)):
            np.random.seed(4321)
            text, timing = decoders.decode_morse(x, fs, thr)
            env = np.abs(x); env = env / np.max(env); sig = 20 * np.log10(env + 1e-10) > thr       # decoders.py:149-156, the same expressions
            tr_ = np.diff(sig.astype(int))
            d[f"m_iq_{tag}"], d[f"m_fs_{tag}"], d[f"m_thr_{tag}"], d[f"m_text_{tag}"] = x, np.array(fs), np.array(float(thr)), np.array(text)
            d[f"m_timing_{tag}"] = np.array([float(timing["dot"]), float(timing["dash"]), float(timing["gap"])])
            d[f"m_rise_{tag}"] = np.where(tr_ == 1)[0].astype(np.int32); d[f"m_fall_{tag}"] = np.where(tr_ == -1)[0].astype(np.int32)
            mt.append(tag)
    d["morse_tags"] = np.array(mt)
    # ---- bandpass_filter on complex / 2-D input
    bt = []
    for tag, shape, lo, hi, fs, dt in (("c64", (3000,), 300, 3000, 22050.0, np.complex64), ("c128", (700,), 0, 15000, 250e3, np.complex128),
                                       ("rows", (5, 1200), 1100, 1300, 22050.0, np.float64), ("crows", (3, 900), 300, 3000, 22050.0, np.complex64)):
        x = rng.standard_normal(shape) + (1j * rng.standard_normal(shape) if np.issubdtype(dt, np.complexfloating) else 0)
        x = x.astype(dt)
        y = sp.bandpass_filter(x, lo, hi, fs)
        nyq = fs / 2
        d[f"b_x_{tag}"], d[f"b_y_{tag}"], d[f"b_args_{tag}"] = x, y, np.array([lo, hi, fs])
        d[f"b_sos_{tag}"] = (ss.butter(5, hi / nyq, btype="low", output="sos") if lo <= 0 else ss.butter(5, [lo / nyq, hi / nyq], btype="band", output="sos"))
        bt.append(tag)
    d["bandpass_tags"] = np.array(bt)
    mg.save("args", **d)


if __name__ == "__main__":
    main()
