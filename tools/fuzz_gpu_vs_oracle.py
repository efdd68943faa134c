#!/usr/bin/env python3
"""Fuzz the HIP path against the C oracle on the GPU box: random batch sizes, frame lengths, sample rates and signal kinds
through NFM / AM / SSB / WFM (fused and small-batch kernels), iq_correction and power.  Bit for bit.
    python tools/fuzz_gpu_vs_oracle.py [cases]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import oracle_lib as O
import gpu_util as G
from pyspecsdr_amd import _lib as L

rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", "99")))
e = G.engine()

EDGE = os.environ.get("FUZZ_EDGE", "0") == "1"   # extreme scales, constants, sprinkled zeros, impulses, lattices, exact tones


def rnd_iq(nf, n):
    if EDGE:
        kind = rng.integers(0, 6)
        scale = 10.0 ** rng.uniform(-8, 4)
        if kind == 0: x = (rng.standard_normal((nf, n)) + 1j * rng.standard_normal((nf, n))) * scale
        elif kind == 1: x = np.repeat((rng.standard_normal((nf, 1)) + 1j * rng.standard_normal((nf, 1))) * scale, n, axis=1)
        elif kind == 2:
            x = (rng.standard_normal((nf, n)) + 1j * rng.standard_normal((nf, n))) * scale
            x[rng.random((nf, n)) < 0.25] = 0
        elif kind == 3:
            x = np.zeros((nf, n), complex); x[np.arange(nf), rng.integers(0, n, size=nf)] = scale
        elif kind == 4: x = (rng.integers(-1, 2, size=(nf, n)) + 1j * rng.integers(-1, 2, size=(nf, n))) * scale
        else: x = np.repeat(np.exp(2j * np.pi * 0.25 * np.arange(n))[None, :], nf, axis=0) * scale
        return x.astype(np.complex64)
    kind = rng.integers(0, 4)
    if kind == 0:
        x = rng.uniform(0.05, 2.0) * np.exp(1j * np.cumsum(rng.standard_normal((nf, n)) * rng.uniform(0.01, 0.5), axis=1))
    elif kind == 1:
        x = rng.standard_normal((nf, n)) + 1j * rng.standard_normal((nf, n))
    elif kind == 2:
        u = rng.integers(0, 256, size=(nf, n, 2)); x = ((u[..., 0] - 127.5) / 127.5) + 1j * ((u[..., 1] - 127.5) / 127.5)
    else:
        x = 0.3 * np.exp(2j * np.pi * rng.uniform(-0.4, 0.4) * np.arange(n))[None, :] + 0.01 * (rng.standard_normal((nf, n)) + 1j * rng.standard_normal((nf, n)))
    return (x + rng.uniform(0, 0.05) * (rng.standard_normal((nf, n)) + 1j * rng.standard_normal((nf, n)))).astype(np.complex64)

def wfm(iq, fs):
    nf, n = iq.shape
    n_out = e.demod_out_len(L.MODE_WFM, n, fs)
    d_pcm, d_au = G.empty((nf, n_out, 2), torch.int16), G.empty((nf, n_out, 2), torch.float64)
    e.demod_signal(L.MODE_WFM, G.dev(iq), nf, n, fs, d_pcm, d_au); e.sync()
    return G.host(d_pcm), G.host(d_au)

bad = 0
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
for it in range(cases):
    nf = int(rng.choice([1, 2, 15, 16, 17, 63, 64, 65, 130, 300]))
    n = int(rng.choice([29, 64, 129, 256, 257, 512, 1000, 1024, 2048, 2049, 4096, 8192, 9000, 16384, 33000]))
    fs = float(rng.choice([2.4e6, 1.024e6, 2.048e6, 250e3, 10e6]))
    iq = rnd_iq(nf, n)
    pick = sorted(set([0, nf - 1, int(rng.integers(0, nf))]))
    taps, sos, zi = e.nfm_filters(fs)
    with np.errstate(all="ignore"):
        for sb in (1, 0):
            e.set_option("small_batch", sb)
            pcm, au = G.demod(L.MODE_NFM, iq, fs)
            for f in pick:
                ref = O.demod_nfm(iq[f], fs, taps, sos, zi)
                if not np.array_equal(au[f], ref, equal_nan=True): bad += 1; print("NFM", sb, nf, n, fs, f)
            if fs > 106e3 + 1:
                lp, pil, lmr, alpha = e.wfm_filters(fs)
                filt = dict(lp_sos=lp, pilot_sos=pil, lmr_sos=lmr, alpha=alpha, dec_sos=sos, dec_zi=zi)
                pcm, au = wfm(iq, fs)
                for f in pick:
                    ref = O.demod_wfm(O.iq_correction(iq[f]), fs, filt)
                    if not np.array_equal(au[f], ref, equal_nan=True): bad += 1; print("WFM", sb, nf, n, fs, f)
        e.set_option("small_batch", 1)
        am = np.empty((5, 6)); e.lib.pss_am_bandpass_sos(am.ctypes.data)
        pcm, au = G.demod(L.MODE_AM, iq, fs)
        stp = e.ssb_taps(fs)
        e.set_option("ssb_hilbert", 0)          # without the Hilbert round trip: bit for bit; with it (the default, power-of-two
        pcm2, au2 = G.demod(L.MODE_USB, iq, fs)  # frames of 256..16384 samples): the same int16, float64 within the round trip's rounding
        e.set_option("ssb_hilbert", 1)
        pcm3, au3 = G.demod(L.MODE_USB, iq, fs)
        au4 = None
        if n & (n - 1) == 0 and 256 <= n <= 16384:   # option "hilbert_exact": SciPy's hilbert() replayed — the reference's float64 audio, every bit
            e.set_option("hilbert_exact", 1)
            pcm4, au4 = G.demod(L.MODE_USB, iq, fs)
            e.set_option("hilbert_exact", 0)
        d_p = G.empty((nf,), torch.float32); e.power_db(G.dev(iq), nf, n, d_p)
        d_c = G.empty((nf, n, 2), torch.float32); e.iq_correction(G.dev(iq), nf, n, d_c, None); e.sync()
        p = G.host(d_p); corr = G.host(d_c).reshape(nf, -1).view(np.complex64)
        for f in pick:
            if not np.array_equal(au[f], O.demod_am(iq[f], am), equal_nan=True): bad += 1; print("AM", nf, n, f)
            if not np.array_equal(au2[f], O.demod_ssb(iq[f], stp, hilbert=False), equal_nan=True): bad += 1; print("SSB", nf, n, f)
            if au4 is not None and not np.array_equal(au4[f], O.demod_ssb(iq[f], stp), equal_nan=True): bad += 1; print("SSB hilbert_exact", nf, n, f)
            if not np.array_equal(pcm3[f], pcm2[f]): bad += 1; print("SSB-hilbert int16", nf, n, f, int((pcm3[f] != pcm2[f]).sum()))
            if not np.allclose(au3[f], au2[f], rtol=0, atol=1e-13, equal_nan=True): bad += 1; print("SSB-hilbert f64", nf, n, f, np.nanmax(np.abs(au3[f] - au2[f])))
            ref = O.iq_correction(iq[f])   # NaN payloads / signs differ between x86 and the GPU: compare values, NaN == NaN
            if not (np.array_equal(corr[f].real, ref.real, equal_nan=True) and np.array_equal(corr[f].imag, ref.imag, equal_nan=True)
                    and np.array_equal(np.signbit(corr[f].real[np.isfinite(ref.real)]), np.signbit(ref.real[np.isfinite(ref.real)]))):
                bad += 1; print("IQC", nf, n, f)
            r = np.float32(O.power_db(iq[f]))
            if not (p[f].tobytes() == r.tobytes() or (np.isnan(p[f]) and np.isnan(r))): bad += 1; print("POW", nf, n, f, p[f], r)
print("cases", cases, "bad", bad)
