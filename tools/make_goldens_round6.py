#!/usr/bin/env python3
"""Golden vectors for complex128 read buffers (tests/golden/c128.npz) — made like tools/make_goldens.py: the build container imports the
reference's own module from /root/reference, feeds it seeded synthetic inputs and stores DATA only (inputs, outputs, SciPy's filter table).

The reference's SDR read buffer is complex64 (pyspecsdr.py:1887), but its functions accept any array; handed complex128 they compute in float64
from the first statement on:
    compute_fft    (signal_processing.py:243-264): `samples * window` is a float64 product of float64 samples
    demodulate_am  (:179-195): np.abs / np.mean / the subtraction in float64 (complex64 input: float32)
    measure_signal_power (:325-328): np.abs ** 2 / np.mean in float64, the scalar log10 in float64 (keys mp_* / pw_*)
    demodulate_ssb (:198-217): the complex128 convolution on the samples as they are (keys ssb_* / ssbpcm_* / ssb_taps)
Rounds 1-5 narrowed such input to complex64 with a warning; round 6 serves these two functions in float64.
The same file carries round 6's other new vectors: demodulate_nfm and demodulate_wfm at a decimation factor int(sample_rate / target_rate) of ONE
(keys n_* / w_*: NFM runs decimate(x, 1), WFM skips the stage, signal_processing.py:111-112 / :152-155).

    python tools/make_goldens_round6.py
"""
import os
import sys

sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import numpy as np
import scipy.signal as ss

import make_goldens as mg            # the generators' helpers (stamp, save); importing it stubs nothing and writes nothing
import signal_processing as sp      # the reference hot path


def main():
    d = {}
    rng = np.random.default_rng(606)
    tags = []
    for tag, nf, n in (("a", 3, 1024), ("b", 1, 4096), ("c", 1, 16384), ("d", 2, 1000), ("e", 1, 32768), ("f", 2, 37)):
        t = np.arange(n) / 2.4e6
        iq = np.empty((nf, n), np.complex128)
        for f in range(nf):
            m = 0.5 * np.sin(2 * np.pi * 40e3 * t + 0.3 * f) + 0.3 * np.sin(2 * np.pi * 90e3 * t + 0.1 * f)
            iq[f] = (1 + 0.5 * m) * 0.5 * np.exp(1j * (0.3 + 2 * np.pi * 1234.5 * t)) + 0.01 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
        iq *= 1.0 + 1e-9 * rng.standard_normal((nf, n))        # values that are NOT representable in complex64: narrowing would change every result
        d[f"iq_{tag}"] = iq
        d[f"db_{tag}"] = np.stack([sp.compute_fft(x) for x in iq])                       # float64 [nf][n]
        aud = np.stack([sp.demodulate_am(x) for x in iq])                                # float64 [nf][n][2]
        assert np.array_equal(aud[..., 0], aud[..., 1])                                  # mono_to_stereo: the columns are equal, one is stored
        d[f"audio_{tag}"] = aud[..., 0]
        d[f"pcm_{tag}"] = np.int16(aud[..., 0] * 32767)
        d[f"abs_{tag}"] = np.abs(iq[0])
        d[f"mean_{tag}"] = np.array(np.mean(np.abs(iq[0])))
        # measure_signal_power (:325-328) of the same buffers: float64 np.abs ** 2 / np.mean, then the scalar 10 * np.log10(power + 1e-10);
        # mp_*: the array part (the mean power), pw_*: what the function returns
        # demodulate_ssb (:198-217; both branches are the same statements) at 48 kHz: the complex128 FIR the reference also runs for complex64 input
        # (lfilter widens it), the hilbert() round trip, normalisation
        ssb = np.stack([sp.demodulate_ssb(x, 48000.0, lower=bool(k & 1)) for k, x in enumerate(iq)])
        assert np.array_equal(ssb[..., 0], ssb[..., 1])
        d[f"ssb_{tag}"] = ssb[..., 0]
        d[f"ssbpcm_{tag}"] = np.int16(ssb[..., 0] * 32767)
        d[f"mp_{tag}"] = np.array([np.mean(np.abs(x) ** 2) for x in iq])
        d[f"pw_{tag}"] = np.array([sp.measure_signal_power(x) for x in iq])
        tags.append(tag)
    # a frame of zeros (np.abs = 0, silence -> NaN audio -> int16 0), one with an infinity
    z = np.zeros(512, np.complex128)
    with np.errstate(all="ignore"):
        d["iq_z"], d["db_z"], d["audio_z"] = z[None], sp.compute_fft(z)[None], sp.demodulate_am(z)[None][..., 0]
        d["pcm_z"] = np.int16(np.nan_to_num(d["audio_z"], nan=0.0) * 32767)      # np.int16(NaN) = 0 on x86 (App. C); stored explicitly
    # demodulate_nfm at a decimation factor of ONE (target_rate above half the sample rate: decimate(x, 1) = cheby1(8, 0.05, 0.8) both ways, no
    # samples dropped) — the reference's `int(sample_rate / target_rate)` (:111) reaches it; ADVICE r5 asked for a vector
    qt = []
    for tag, n, fs, tr, seed in (("q1a", 2048, 52920.0, 44100, 71), ("q1b", 1500, 96000.0, 50000, 72)):
        iq = mg.fm_iq(2, n, fs, seed, dev=3e3)
        aud = np.stack([sp.demodulate_nfm(f, fs, tr) for f in iq])
        assert int(fs / tr) == 1 and aud.shape == (2, n - 1, 2)
        sos = ss.cheby1(8, 0.05, 0.8, output="sos")
        d[f"n_iq_{tag}"], d[f"n_fs_{tag}"], d[f"n_tr_{tag}"] = iq, np.array(fs), np.array(tr)
        d[f"n_audio_{tag}"], d[f"n_pcm_{tag}"] = aud[..., 0], np.int16(aud[..., 0] * 32767)
        d[f"n_taps_{tag}"] = ss.firwin(numtaps=65, cutoff=15000 / (fs / 2))
        d[f"n_sos_{tag}"], d[f"n_zi_{tag}"] = sos, ss.sosfilt_zi(sos)
        qt.append(tag)
    d["q1_tags"] = np.array(qt)
    # demodulate_wfm at a decimation factor of one: the reference SKIPS decimate() (`if decimation_factor > 1`, :152-155) and normalises the
    # de-emphasised channels themselves
    wt = []
    for tag, n, fs, tr, seed in (("w1a", 4096, 240e3, 192000, 81), ("w1b", 3000, 250e3, 130000, 82)):
        iq = mg.wfm_iq(2, n, fs, seed)
        aud = np.stack([sp.demodulate_wfm(f, fs, tr) for f in iq])
        assert int(fs / tr) == 1 and aud.shape == (2, n - 1, 2)
        nyq = fs / 2
        d[f"w_iq_{tag}"], d[f"w_fs_{tag}"], d[f"w_tr_{tag}"], d[f"w_audio_{tag}"] = iq, np.array(fs), np.array(tr), aud
        d[f"w_pcm_{tag}"] = np.int16(aud * 32767)
        d[f"w_lp_{tag}"] = ss.butter(5, 15000 / nyq, btype="low", output="sos")
        d[f"w_pil_{tag}"] = ss.butter(5, [18800 / nyq, 19200 / nyq], btype="band", output="sos")
        d[f"w_lmr_{tag}"] = ss.butter(5, [23000 / nyq, 53000 / nyq], btype="band", output="sos")
        d[f"w_alpha_{tag}"] = np.array(np.exp(-1 / (75e-6 * fs)))
        wt.append(tag)
    d["wq1_tags"] = np.array(wt)
    d["tags"] = np.array(tags)
    d["ssb_taps"] = ss.firwin(65, 3000 / 48000.0, window="hamming")                          # demodulate_ssb's taps at 48 kHz (:203 / :208)
    d["am_sos"] = ss.butter(5, [300 / 11025, 3000 / 11025], btype="band", output="sos")   # demodulate_am's filter (:188-191, fs fixed at 22 050)
    mg.save("c128", **d)


if __name__ == "__main__":
    main()
