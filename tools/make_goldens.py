#!/usr/bin/env python3
"""Generate the golden input/output vectors under tests/golden/.

Runs ONLY in the build container, where the Python reference is mounted read-only at
/root/reference.  It imports the reference's own hot-path module (signal_processing.py) and,
with sounddevice/SoapySDR/curses stubbed, its caller (pyspecsdr.py), feeds them seeded
synthetic IQ, and stores *data only* (inputs, outputs, designed filter coefficients) as
compressed .npz fixtures.  No reference source text or bytecode is written anywhere.

The numbers are produced by NumPy/SciPy native code (pocketfft, SVML, _sosfilt ...), so the
fixtures are stamped with the library versions and the enabled NumPy CPU-dispatch features.

    python tools/make_goldens.py            # rewrites tests/golden/*.npz
"""
import json
import os
import sys
import types
import warnings

sys.dont_write_bytecode = True
REF = "/root/reference"
sys.path.insert(0, REF)

import numpy as np
import scipy
import scipy.signal as ss

import signal_processing as sp  # the reference hot path (signal_processing.py)

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
os.makedirs(OUT, exist_ok=True)


# --------------------------------------------------------------------------------------
# synthetic IQ (SURVEY.md §8d shapes; plain seeded NumPy — inputs are stored in the fixture)
# --------------------------------------------------------------------------------------
def fm_iq(n_frames, n, fs, seed, amp=0.5, sigma=0.02, dev=5e3):
    rng = np.random.default_rng(seed)
    t = np.arange(n) / fs
    out = np.empty((n_frames, n), np.complex64)
    for f in range(n_frames):
        ph0 = 0.1 * f
        m = (0.5 * np.sin(2 * np.pi * 400 * t + ph0) + 0.3 * np.sin(2 * np.pi * 1000 * t + 2 * ph0)
             + 0.2 * np.sin(2 * np.pi * 2500 * t + 3 * ph0))
        phase = 2 * np.pi * dev * np.cumsum(m) / fs + ph0
        x = amp * np.exp(1j * phase) + sigma * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
        out[f] = x.astype(np.complex64)
    return out


def am_iq(n_frames, n, fs, seed, sigma=0.01):
    rng = np.random.default_rng(seed)
    t = np.arange(n) / fs
    out = np.empty((n_frames, n), np.complex64)
    for f in range(n_frames):
        m = 0.5 * np.sin(2 * np.pi * 40e3 * t + 0.3 * f) + 0.3 * np.sin(2 * np.pi * 90e3 * t + 0.1 * f)
        x = (1 + 0.5 * m) * 0.5 * np.exp(1j * 0.3) + sigma * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
        out[f] = x.astype(np.complex64)
    return out


def ssb_iq(n_frames, n, fs, seed, sigma=0.01):
    rng = np.random.default_rng(seed)
    t = np.arange(n) / fs
    out = np.empty((n_frames, n), np.complex64)
    for f in range(n_frames):
        x = (0.4 * np.exp(1j * 2 * np.pi * (1500 + 37 * f) * t) + 0.2 * np.exp(1j * 2 * np.pi * 2400 * t)
             + sigma * (rng.standard_normal(n) + 1j * rng.standard_normal(n)))
        out[f] = x.astype(np.complex64)
    return out


def scan_iq(n_slices, n, fs, seed):
    rng = np.random.default_rng(seed)
    t = np.arange(n)
    out = np.empty((n_slices, n), np.complex64)
    for s in range(n_slices):
        x = 0.01 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
        b = int(rng.integers(-n // 2 + 8, n // 2 - 8))
        x = x + 0.3 * np.exp(2j * np.pi * b * t / n)
        if s % 2 == 0:  # wide FM-ish carrier: many bins within 20 dB of the peak
            m = np.cumsum(rng.standard_normal(n)) * 0.4
            x = x + 0.5 * np.exp(1j * m)
        out[s] = x.astype(np.complex64)
    return out


def stamp():
    from numpy._core._multiarray_umath import __cpu_features__ as feats
    return json.dumps({
        "numpy": np.__version__, "scipy": scipy.__version__,
        "python": sys.version.split()[0],
        "cpu_features": sorted(k for k, v in feats.items() if v),
        "reference": "xqtr/PySpecSDR v1.0.6 tree (signal_processing.py, pyspecsdr.py)",
    })


def save(name, **arrs):
    arrs["stamp"] = np.array(stamp())
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"{name}.npz  {os.path.getsize(path) / 1024:.1f} KiB")


# --------------------------------------------------------------------------------------
def gen_atan2():
    """np.angle on complex64 == the float32 arctan2 loop (SVML __svml_atan2f16 on AVX512_SKX)."""
    rng = np.random.default_rng(101)
    n = 60000
    re = rng.standard_normal(n).astype(np.float32) * np.float32(0.3)
    im = rng.standard_normal(n).astype(np.float32) * np.float32(0.3)
    # wide dynamic range + axis-hugging cases
    k = n // 4
    re[:k] *= np.exp2(rng.integers(-60, 60, k)).astype(np.float32)
    im[:k] *= np.exp2(rng.integers(-60, 60, k)).astype(np.float32)
    im[k:k + 2000] *= np.float32(1e-6)
    re[k + 2000:k + 4000] *= np.float32(1e-6)
    special = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 1e-40, -1e-40, 3e38, -3e38,
                        1e-38, 2.5e-38, 1e37, 2e36], np.float32)
    sy, sx = np.meshgrid(special, special)
    y = np.concatenate([im, sy.ravel()])
    x = np.concatenate([re, sx.ravel()])
    z = (x + 1j * y).astype(np.complex64)
    # make sure inf/nan/-0 survive the complex construction
    z.real[:] = x
    z.imag[:] = y
    with np.errstate(all="ignore"):
        th = np.angle(z)
        th2 = np.arctan2(y, x)
    assert th.dtype == np.float32
    assert np.array_equal(th.view(np.uint32), th2.view(np.uint32))
    save("atan2f", y=y, x=x, theta=th, n_random=np.array(n))


def gen_atan2f_bits():
    """np.arctan2 float32 on operands drawn as RANDOM BIT PATTERNS (every exponent, denormals, infinities, NaNs): pins the
    special-value path of the library routine, which gen_atan2's mostly well-scaled operands reach only through its 15 x 15 grid."""
    rng = np.random.default_rng(505)
    n = 20000
    y = rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32).view(np.float32)
    x = rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32).view(np.float32)
    sp_ = np.float32([0.0, -0.0, np.inf, -np.inf, 1e-45, -1e-45, 1e-38, 3e38, -3e38, 1.0, -1.0, 1e-20, 1e20])
    y = np.concatenate([y, rng.choice(sp_, 4000), rng.integers(0, 2 ** 32, 4000, dtype=np.uint64).astype(np.uint32).view(np.float32)])
    x = np.concatenate([x, rng.integers(0, 2 ** 32, 4000, dtype=np.uint64).astype(np.uint32).view(np.float32), rng.choice(sp_, 4000)])
    with np.errstate(all="ignore"):
        th = np.arctan2(y, x)
    save("atan2f_bits", y=y, x=x, theta=th)


def gen_spectrum():
    d = {}
    for n, fs, nf, seed in [(1024, 2.4e6, 4, 11), (2048, 10e6, 2, 12), (4096, 2.4e6, 2, 13),
                            (16384, 2.4e6, 1, 14), (256, 1.024e6, 2, 15), (8192, 1.024e6, 1, 16)]:
        iq = fm_iq(nf, n, fs, seed)
        if n == 4096:  # one high-dynamic-range frame: strong tone over a -100 dB floor
            t = np.arange(n)
            iq[1] = (0.9 * np.exp(2j * np.pi * 300.25 * t / n) + 1e-5 * iq[1]).astype(np.complex64)
        db = np.stack([sp.compute_fft(f) for f in iq])
        post = []
        for row in db:
            # caller-side post-process, exactly the three statements at pyspecsdr.py:2278-2283
            fd = np.convolve(row, np.ones(5) / 5, mode="valid")
            thr = np.median(fd) - 10
            fd[fd < thr] = thr
            post.append(fd)
        d[f"iq_{n}"] = iq
        d[f"db_{n}"] = db
        d[f"post_{n}"] = np.stack(post)
    z = np.zeros(1024, np.complex64)
    d["db_zero"] = sp.compute_fft(z)
    # lengths that are not a power of two (np.fft.fft takes any length; the sweep driver reads int(0.1 * fs) samples)
    for n, seed in [(1000, 17), (3001, 18), (24000, 19), (17, 20)]:
        iq = fm_iq(1, n, 2.4e6, seed)
        d[f"iq_np2_{n}"] = iq
        d[f"db_np2_{n}"] = np.stack([sp.compute_fft(f) for f in iq])
    d["np2_sizes"] = np.array([1000, 3001, 24000, 17])
    save("spectrum", **d)


def gen_nfm():
    d = {}
    for tag, n, fs, nf, seed in [("a", 1024, 2.4e6, 6, 21), ("b", 2048, 10e6, 3, 22), ("c", 4096, 1.024e6, 2, 23),
                                 ("d", 29, 2.4e6, 1, 24), ("e", 32768, 2.4e6, 1, 25), ("f", 1000, 2.4e6, 2, 26),
                                 ("g", 40000, 2.4e6, 2, 27)]:   # g: past NumPy's 256 KiB temporary-elision threshold (N - 1 >= 32768)
        iq = fm_iq(nf, n, fs, seed)
        aud = np.stack([sp.demodulate_signal(f, fs, "NFM") for f in iq])  # (nf, n_out, 2)
        assert np.array_equal(aud[..., 0], aud[..., 1])
        pcm = np.int16(aud * 32767)
        q = int(fs / 22050)
        d[f"iq_{tag}"] = iq
        d[f"fs_{tag}"] = np.array(fs)
        d[f"audio_{tag}"] = aud[..., 0].copy()
        d[f"pcm_{tag}"] = pcm
        # intermediate stages for the first frame (helps localise a mismatch)
        x = iq[0]
        dem = np.angle(x[1:] * np.conj(x[:-1]))
        dem = dem * (fs / (2 * np.pi))
        taps = ss.firwin(numtaps=65, cutoff=15000 / (fs / 2))
        fil = ss.lfilter(taps, 1.0, dem)
        d[f"disc_{tag}"] = dem
        d[f"fir_{tag}"] = fil
        d[f"taps_{tag}"] = taps
        sos = ss.cheby1(8, 0.05, 0.8 / q, output="sos")
        d[f"sos_{tag}"] = sos
        d[f"zi_{tag}"] = ss.sosfilt_zi(sos)
    # coefficient sets for the config sample rates
    for fs in (1.024e6, 2.4e6, 10e6, 2.048e6, 250e3):
        q = int(fs / 22050)
        key = f"{int(fs)}"
        d["design_taps_" + key] = ss.firwin(numtaps=65, cutoff=15000 / (fs / 2))
        sos = ss.cheby1(8, 0.05, 0.8 / q, output="sos")
        d["design_sos_" + key] = sos
        d["design_zi_" + key] = ss.sosfilt_zi(sos)
    # silence -> NaN audio -> int16 0
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        z = np.zeros(1024, np.complex64)
        a = sp.demodulate_nfm(z, 2.4e6)
        d["audio_silence"] = a[:, 0].copy()
        d["pcm_silence"] = np.int16(a * 32767)
    save("nfm", **d)


def gen_am_ssb():
    d = {}
    for tag, n, nf, seed in [("a", 1024, 3, 31), ("b", 16384, 1, 32), ("c", 300, 2, 33), ("d", 20000, 1, 34)]:
        iq = am_iq(nf, n, 2.4e6, seed)
        aud = np.stack([sp.demodulate_signal(f, 2.4e6, "AM") for f in iq])
        d[f"am_iq_{tag}"] = iq
        d[f"am_audio_{tag}"] = aud[..., 0].copy()
        d[f"am_pcm_{tag}"] = np.int16(aud * 32767)
        d[f"am_env_{tag}"] = np.abs(iq[0])
        d[f"am_mean_{tag}"] = np.array(np.mean(np.abs(iq[0])))
    d["am_sos"] = ss.butter(5, [300.0 / (22050 / 2), 3000.0 / (22050 / 2)], btype="band", output="sos")
    for tag, n, fs, nf, seed in [("a", 1024, 2.4e6, 3, 41), ("b", 16384, 2.4e6, 1, 42), ("c", 2048, 1.024e6, 1, 43)]:
        iq = ssb_iq(nf, n, fs, seed)
        usb = np.stack([sp.demodulate_signal(f, fs, "USB") for f in iq])
        lsb = np.stack([sp.demodulate_signal(f, fs, "LSB") for f in iq])
        assert np.array_equal(usb, lsb)
        d[f"ssb_iq_{tag}"] = iq
        d[f"ssb_fs_{tag}"] = np.array(fs)
        d[f"ssb_audio_{tag}"] = usb[..., 0].copy()
        d[f"ssb_pcm_{tag}"] = np.int16(usb * 32767)
        d[f"ssb_taps_{tag}"] = ss.firwin(65, 3000 / fs, window="hamming")
    # dispatcher edge cases
    x = fm_iq(1, 64, 2.4e6, 44)[0]
    d["raw_iq"] = x
    d["raw_out"] = sp.demodulate_signal(x, 2.4e6, "RAW")
    d["unknown_out"] = sp.demodulate_signal(x, 2.4e6, "DIGITAL")
    save("am_ssb", **d)


def gen_iqcorr():
    """iq_correction (signal_processing.py:46-80) + demodulate_signal(..., 'RAW') on frames of assorted lengths."""
    d = {}
    rng = np.random.default_rng(71)
    sizes = [64, 1000, 1024, 4096, 10000, 16384, 32768, 40001]
    for n in sizes:
        # unbalanced IQ: DC offset, gain mismatch, phase skew — what the routine is meant to repair
        i = rng.standard_normal(n) * 0.3 + 0.05
        q = 0.8 * (rng.standard_normal(n) * 0.3 + 0.2 * i) - 0.02
        x = (i + 1j * q).astype(np.complex64)
        d[f"iq_{n}"] = x
        d[f"corr_{n}"] = sp.iq_correction(x)
        d[f"raw_{n}"] = sp.demodulate_signal(x, 2.4e6, "RAW")
        assert d[f"corr_{n}"].dtype == np.complex64 and d[f"raw_{n}"].dtype == np.float32
    # rtl-sdr style 8-bit samples (pyrtlsdr: (u8 - 127.5)/127.5), a tone plus noise, with exact zeros spliced in
    n = 2048
    u = rng.integers(0, 256, size=(n, 2)).astype(np.float64)
    x = ((u[:, 0] - 127.5) / 127.5 + 1j * (u[:, 1] - 127.5) / 127.5).astype(np.complex64)
    x[5] = 0
    x[17] = complex(0.0, 0.25)
    x[33] = complex(-0.5, 0.0)
    x[40] = complex(-0.0, -0.0)
    d["iq_u8"] = x
    d["corr_u8"] = sp.iq_correction(x)
    # a batch of assorted random frames (fuzzing against the reference showed that single cases hide 1-ulp differences in
    # np.var's |x|^2: ~8 % of frames are sensitive to its exact form)
    fz = []
    for k in range(48):
        m = 257
        kind = k % 4
        if kind == 0:
            v = rng.uniform(0.05, 2.0) * np.exp(1j * np.cumsum(rng.standard_normal(m) * rng.uniform(0.01, 0.5)))
        elif kind == 1:
            v = rng.standard_normal(m) + 1j * rng.standard_normal(m)
        elif kind == 2:
            u = rng.integers(0, 256, size=(m, 2))
            v = ((u[:, 0] - 127.5) / 127.5) + 1j * ((u[:, 1] - 127.5) / 127.5)
        else:
            v = 0.3 * np.exp(2j * np.pi * rng.uniform(-0.4, 0.4) * np.arange(m)) + 0.01 * (rng.standard_normal(m) + 1j * rng.standard_normal(m))
        fz.append(v.astype(np.complex64))
    fz = np.stack(fz)
    d["iq_fuzz"] = fz
    d["corr_fuzz"] = np.stack([sp.iq_correction(f) for f in fz])
    d["sizes"] = np.array(sizes)
    save("iqcorr", **d)


def wfm_iq(n_frames, n, fs, seed, sigma=0.01):
    """Stereo-multiplexed broadcast FM: L+R, 19 kHz pilot, (L-R) on 38 kHz, 75 kHz deviation, mild IQ imbalance."""
    rng = np.random.default_rng(seed)
    t = np.arange(n) / fs
    out = np.empty((n_frames, n), np.complex64)
    for f in range(n_frames):
        l = np.sin(2 * np.pi * (1000 + 50 * f) * t)
        r = np.sin(2 * np.pi * 3000 * t + 0.3)
        mpx = 0.45 * (l + r) + 0.1 * np.sin(2 * np.pi * 19000 * t) + 0.45 * (l - r) * np.sin(2 * np.pi * 38000 * t)
        ph = 2 * np.pi * 75000 * np.cumsum(mpx) / fs + rng.uniform(0, 6.28)
        x = 0.5 * np.exp(1j * ph) + sigma * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
        x = x.real * 1.05 + 0.01 + 1j * (x.imag * 0.97 + 0.03 * x.real - 0.02)
        out[f] = x.astype(np.complex64)
    return out


def gen_wfm():
    """demodulate_signal(..., 'WFM') = iq_correction (:46-80) + demodulate_wfm (:119-176), plus its filter sets."""
    d = {}
    tags = []
    for tag, n, fs, nf, seed in [("a", 1024, 2.4e6, 4, 81), ("b", 16384, 2.4e6, 1, 82), ("c", 4096, 1.024e6, 2, 83),
                                 ("d", 8192, 250e3, 1, 84), ("e", 2048, 2.048e6, 2, 85), ("f", 29, 2.4e6, 1, 86),
                                 ("g", 40000, 2.4e6, 1, 87)]:
        iq = wfm_iq(nf, n, fs, seed)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            aud = np.stack([sp.demodulate_signal(f, fs, "WFM") for f in iq])  # (nf, n_out, 2)
        assert aud.dtype == np.float64
        d[f"iq_{tag}"] = iq
        d[f"fs_{tag}"] = np.array(fs)
        d[f"audio_{tag}"] = aud
        d[f"pcm_{tag}"] = np.int16(aud * 32767)
        tags.append(tag)
    d["tags"] = np.array(tags)
    for fs in (1.024e6, 2.4e6, 2.048e6, 250e3, 10e6):
        key = f"{int(fs)}"
        q = int(fs / 22050)
        d["lp_sos_" + key] = ss.butter(5, 15000 / (fs / 2), btype="low", output="sos")
        d["pilot_sos_" + key] = ss.butter(5, [18800 / (fs / 2), 19200 / (fs / 2)], btype="band", output="sos")
        d["lmr_sos_" + key] = ss.butter(5, [23000 / (fs / 2), 53000 / (fs / 2)], btype="band", output="sos")
        d["alpha_" + key] = np.array(np.exp(-1 / (75e-6 * fs)))
        sos = ss.cheby1(8, 0.05, 0.8 / q, output="sos")
        d["dec_sos_" + key] = sos
        d["dec_zi_" + key] = ss.sosfilt_zi(sos)
    d["sin_pi"] = np.array(np.sin(np.pi))
    save("wfm", **d)


def gen_bandpass():
    """bandpass_filter (signal_processing.py:34-42) the way decoders.py:100-101 and the demodulators call it."""
    d = {}
    rng = np.random.default_rng(91)
    cases = [("afsk1200", 4000, 1100, 1300, 22050.0, np.float64), ("afsk2200", 4000, 2100, 2300, 22050.0, np.float64),
             ("low", 3000, 0, 15000, 2.4e6, np.float32), ("voice", 5000, 300, 3000, 22050.0, np.float32),
             ("short", 3, 300, 3000, 22050.0, np.float64)]
    for tag, n, lo, hi, fs, dt in cases:
        t = np.arange(n) / fs
        x = (np.sin(2 * np.pi * 1200 * t) + 0.5 * np.sin(2 * np.pi * 2200 * t) + 0.1 * rng.standard_normal(n)).astype(dt)
        y = sp.bandpass_filter(x, lo, hi, fs)
        assert y.dtype == np.float64
        d[f"x_{tag}"], d[f"y_{tag}"] = x, y
        d[f"args_{tag}"] = np.array([lo, hi, fs])
        nyq = fs / 2
        d[f"sos_{tag}"] = (ss.butter(5, hi / nyq, btype="low", output="sos") if lo <= 0 else
                           ss.butter(5, [lo / nyq, hi / nyq], btype="band", output="sos"))
    d["tags"] = np.array([c[0] for c in cases])
    save("bandpass", **d)


def gen_afsk():
    """decode_afsk (decoders.py:94-112): Bell-202 tone energies per bit period -> bit list, on AFSK-like audio."""
    import decoders
    d = {}
    rng = np.random.default_rng(101)
    cases = [("a", 6000, 22050.0), ("b", 9000, 48000.0), ("c", 30000, 250000.0), ("d", 17, 22050.0)]
    for tag, n, fs in cases:
        t = np.arange(n) / fs
        w = int(fs / 1200)
        sym = rng.integers(0, 2, size=n // w + 1)[np.arange(n) // w]
        x = np.where(sym == 1, np.sin(2 * np.pi * 2200 * t), np.sin(2 * np.pi * 1200 * t)) + 0.2 * rng.standard_normal(n)
        x = x / np.max(np.abs(x))                              # decode_aprs normalises first (decoders.py:126)
        bits = decoders.decode_afsk(x, fs)
        d[f"x_{tag}"] = x
        d[f"fs_{tag}"] = np.array(fs)
        d[f"bits_{tag}"] = np.array(bits, np.uint8)
        nyq = fs / 2
        d[f"sos1200_{tag}"] = ss.butter(5, [1100 / nyq, 1300 / nyq], btype="band", output="sos")
        d[f"sos2200_{tag}"] = ss.butter(5, [2100 / nyq, 2300 / nyq], btype="band", output="sos")
    d["tags"] = np.array([c[0] for c in cases])
    save("afsk", **d)


def gen_classify():
    """classify_signal (signal_processing.py:296-322) with its missing import supplied: the module calls `welch` without
    importing it (NameError on every call in the reference); binding scipy.signal.welch into the module's namespace is the
    one-line fix SURVEY §8(f) #3 names.  Stored: inputs, the Welch PSD, the three features and the label."""
    sp.welch = ss.welch
    fs = 2.4e6
    rng = np.random.default_rng(202)

    def sig(n, dev, ftone, noise, off=0.0, amp=0.5):
        t = np.arange(n) / fs
        ph = 2 * np.pi * dev * np.cumsum(np.sin(2 * np.pi * ftone * t)) / fs + 2 * np.pi * off * t
        return (amp * np.exp(1j * ph) + noise * (rng.standard_normal(n) + 1j * rng.standard_normal(n))).astype(np.complex64)

    d, tags = {}, []
    for n in (1024, 1500, 2048, 4096, 10000):
        t = np.arange(n) / fs
        cases = [("fm75k", sig(n, 75e3, 1e3, 0.01)), ("nfm5k", sig(n, 5e3, 1e3, 0.01)), ("noise", sig(n, 0, 1, 0.1, amp=0)),
                 ("tone300k", sig(n, 0, 1, 0.001, off=300e3)), ("fmoff", sig(n, 75e3, 5e3, 0.02, off=-450e3)),
                 ("am", ((1 + 0.5 * np.sin(2 * np.pi * 1e3 * t)) * 0.5 * np.exp(0.3j)).astype(np.complex64) + sig(n, 0, 1, 0.005, amp=0)),
                 ("wide", sig(n, 400e3, 20e3, 0.01))]
        for name, x in cases:
            tags.append(f"{name}_{n}")
            d[f"iq_{name}_{n}"] = x
    # a long scanner dwell (pyspecsdr.py:1026: SCAN_DWELL_TIME * fs samples): the float32 cumsum inside np.unwrap grows large
    tags.append("tone300k_60000"); d["iq_tone300k_60000"] = sig(60000, 0, 1, 0.001, off=300e3)
    tags.append("silence_2048"); d["iq_silence_2048"] = np.zeros(2048, np.complex64)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for tag in tags:
            x = d[f"iq_{tag}"]
            freqs, psd = sp.welch(x, fs=fs, nperseg=1024)
            assert psd.dtype == np.float32 and freqs.dtype == np.float64
            d[f"psd_{tag}"] = psd
            d[f"bw_{tag}"] = np.array(float(sp.estimate_bandwidth(psd, freqs)))
            d[f"mi_{tag}"] = np.array(sp.estimate_modulation_index(x))
            d[f"flat_{tag}"] = np.array(np.exp(np.mean(np.log(psd + 1e-10))) / np.mean(psd))
            d[f"label_{tag}"] = np.array(sp.classify_signal(x, fs, 0.0))
            assert d[f"mi_{tag}"].dtype == np.float32 and d[f"flat_{tag}"].dtype == np.float32
    d["win"] = ss.get_window("hann", 1024).astype(np.complex64).real.copy()
    d["fs"] = np.array(fs)
    d["tags"] = np.array(tags)
    del sp.welch
    save("classify", **d)


def gen_classify_short():
    """classify_signal on reads SHORTER than Welch's segment (signal_processing.py:299: nperseg=1024; SciPy then takes
    nperseg = len(x): ONE segment, a Hann window of that length, an FFT of that — arbitrary — length).  Same stored items
    as gen_classify; the PSD has len(x) bins."""
    sp.welch = ss.welch
    rng = np.random.default_rng(303)
    d, tags = {}, []
    for fs in (2.4e6, 250e3):
        for n in (1023, 1000, 777, 512, 257, 100, 31, 8, 5, 3, 2, 1):
            t = np.arange(n) / fs
            cases = [("nfm", 5e3, 0.0, 0.01), ("fm", 75e3, 0.0, 0.01), ("tone", 0.0, 0.11 * fs, 0.002), ("noise", 0.0, 0.0, 0.2)]
            for name, dev, off, noise in cases:
                ph = 2 * np.pi * dev * np.cumsum(np.sin(2 * np.pi * 3e3 * t)) / fs + 2 * np.pi * off * t
                amp = 0.0 if name == "noise" else 0.5
                x = (amp * np.exp(1j * ph) + noise * (rng.standard_normal(n) + 1j * rng.standard_normal(n))).astype(np.complex64)
                tag = f"{name}_{n}_{int(fs)}"
                tags.append(tag)
                d[f"iq_{tag}"] = x
                d[f"fs_{tag}"] = np.array(fs)
    tags.append("silence_600_2400000"); d["iq_silence_600_2400000"] = np.zeros(600, np.complex64); d["fs_silence_600_2400000"] = np.array(2.4e6)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for tag in tags:
            x, fs = d[f"iq_{tag}"], float(d[f"fs_{tag}"])
            freqs, psd = sp.welch(x, fs=fs, nperseg=1024)
            assert psd.dtype == np.float32 and freqs.dtype == np.float64 and len(psd) == len(x)
            d[f"psd_{tag}"] = psd
            d[f"bw_{tag}"] = np.array(float(sp.estimate_bandwidth(psd, freqs)))
            d[f"mi_{tag}"] = np.array(sp.estimate_modulation_index(x))
            d[f"flat_{tag}"] = np.array(np.exp(np.mean(np.log(psd + 1e-10))) / np.mean(psd))
            d[f"label_{tag}"] = np.array(sp.classify_signal(x, fs, 0.0))
    for n in (1000, 257, 8, 3, 1):
        d[f"win_{n}"] = ss.get_window("hann", n).astype(np.complex64).real.copy()
    d["tags"] = np.array(tags)
    del sp.welch
    save("classify_short", **d)


def gen_log10f():
    """np.log10 on float32 (measure_signal_power :328, estimate_bandwidth :270, the scanner's dB rows): inputs and NumPy's
    outputs, bit patterns.  Under the AVX512_SKX dispatch this is SVML's __svml_log10f16, not a correctly rounded log10f."""
    rng = np.random.default_rng(404)
    u = np.concatenate([
        rng.integers(1, 0x7f800000, 60000, dtype=np.uint32),                        # all over the positive finite range
        np.arange(1, 4097, dtype=np.uint32), np.arange(0x00800000 - 2048, 0x00800000 + 2048, dtype=np.uint32),   # denormals, the boundary
        np.arange(0x3f800000 - 4096, 0x3f800000 + 4096, dtype=np.uint32),           # around 1
        np.arange(0x3fc00000 - 2048, 0x3fc00000 + 2048, dtype=np.uint32),           # around 1.5 (the mantissa fold)
        np.float32([1e-10, 0.1, 0.01, 10.0, 100.0, 1e10]).view(np.uint32),
        np.float32(1e-10).view(np.uint32) + np.arange(-64, 64).astype(np.uint32),
        np.uint32([0x7f7fffff, 0x7f800000, 0x00000000, 0x80000000, 0xbf800000, 0x7fc00000, 0xff800000])])
    x = u.astype(np.uint32).view(np.float32)
    with np.errstate(all="ignore"):
        y = np.log10(x)
        ys = np.array([np.log10(v) for v in x[:2000]])                               # scalars take the same loop
    assert y.dtype == np.float32 and np.array_equal(ys.view(np.uint32), y[:2000].view(np.uint32))
    save("log10f", x=x, y=y)


def gen_decoders():
    """decoders.py end to end: decode_morse (text + timing; np.random seeded because scipy's kmeans draws its initial centroids from
    the global state) and decode_aprs (AFSK audio -> packet list).  rise_ / fall_ are NOT reference outputs (decode_morse keeps
    them local): they are the same NumPy expressions (decoders.py:149-161) evaluated here, stored to pin the edge kernel."""
    import decoders
    d, mtags, atags = {}, [], []
    rng = np.random.default_rng(303)
    code = {v: k for k, v in __import__("pyspecconst").MORSE_CODE.items() if len(v) == 1}

    def cw(text, fs, unit, n, noise, lead=3, amp=0.6, soft=0):
        key = [0] * lead
        for wi, word in enumerate(text.split(" ")):
            if wi: key += [0] * 4                      # 7 units between words (3 already follow the last letter)
            for ch in word:
                for sym in code[ch]:
                    key += [1] * (1 if sym == "." else 3) + [0]
                key += [0] * 2
        k = np.repeat(np.array(key, float), unit)[:n]
        k = np.concatenate([k, np.zeros(n - len(k))])
        if soft: k = np.convolve(k, np.ones(soft) / soft, mode="same")
        t = np.arange(n)
        return (amp * k * np.exp(2j * np.pi * 700.0 / fs * t) + noise * (rng.standard_normal(n) + 1j * rng.standard_normal(n))).astype(np.complex64)

    cases = [("sos", cw("SOS", 24000.0, 1100, 32768, 0.002), 24000.0),
             ("cq", cw("CQ DE K", 48000.0, 900, 60000, 0.01, soft=40), 48000.0),
             ("noisy", cw("TEST", 24000.0, 700, 30000, 0.04, soft=25), 24000.0),
             ("one", cw("E", 24000.0, 1000, 5000, 0.001), 24000.0),
             ("keydown", cw("AN", 24000.0, 800, 20000, 0.002, lead=0), 24000.0),
             ("silence", np.zeros(4096, np.complex64), 24000.0),
             ("noise", (0.1 * (rng.standard_normal(8000) + 1j * rng.standard_normal(8000))).astype(np.complex64), 24000.0)]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for tag, x, fs in cases:
            np.random.seed(1234)
            text, timing = decoders.decode_morse(x, fs)
            env = np.abs(x); env = env / np.max(env); sig = 20 * np.log10(env + 1e-10) > -20
            tr = np.diff(sig.astype(int))
            mtags.append(tag)
            d[f"m_iq_{tag}"] = x; d[f"m_fs_{tag}"] = np.array(fs); d[f"m_text_{tag}"] = np.array(text)
            d[f"m_timing_{tag}"] = np.array([float(timing["dot"]), float(timing["dash"]), float(timing["gap"])])
            d[f"m_rise_{tag}"] = np.where(tr == 1)[0].astype(np.int32); d[f"m_fall_{tag}"] = np.where(tr == -1)[0].astype(np.int32)

    def ax25_bits(dest, src, info, pre):
        by = [(ord(c) << 1) for c in dest.ljust(6)] + [0x60] + [(ord(c) << 1) for c in src.ljust(6)] + [0x61, 0x03, 0xF0] + [ord(c) for c in info]
        bits = [(b >> j) & 1 for b in by for j in range(8)]
        st, ones = [], 0
        for b in bits:
            st.append(b); ones = ones + 1 if b else 0
            if ones == 5: st.append(0); ones = 0
        flag = [0, 1, 1, 1, 1, 1, 1, 0]
        return [0] * pre + flag + st + flag + [0] * 9

    for tag, fs, delay, cplx in (("a", 22050.0, 9, False), ("b", 48000.0, 30, False), ("c", 9600.0, 7, True), ("d", 22050.0, 0, False)):
        bits = ax25_bits("APRS", "N0CALL", "HELLO " + tag, 5)
        w = int(fs / 1200)
        sym = np.repeat(bits, w)
        n = len(sym) + 12 * w
        f = np.where(np.concatenate([np.zeros(delay), sym, np.zeros(n - len(sym) - delay)])[:n] > 0, 2200.0, 1200.0)
        x = 0.8 * np.sin(2 * np.pi * np.cumsum(f) / fs) + 0.01 * rng.standard_normal(n)
        if cplx: x = x + 0.3j * rng.standard_normal(n)
        out = decoders.decode_aprs(x, fs)
        atags.append(tag)
        d[f"a_x_{tag}"] = x; d[f"a_fs_{tag}"] = np.array(fs)
        d[f"a_packets_{tag}"] = np.array(json.dumps(out))          # JSON text: NumPy's str dtype drops trailing NULs
        xr = np.real(x) if np.iscomplexobj(x) else x               # the bit stream decode_aprs hands to its framing code:
        d[f"a_bits_{tag}"] = np.array(decoders.decode_afsk(xr / np.max(np.abs(xr)), fs), np.uint8)   # decoders.py:122-129
        nyq = fs / 2
        d[f"a_sos1200_{tag}"] = ss.butter(5, [1100 / nyq, 1300 / nyq], btype="band", output="sos")
        d[f"a_sos2200_{tag}"] = ss.butter(5, [2100 / nyq, 2300 / nyq], btype="band", output="sos")
    # the bookkeeping either side: random and planted bit streams through decode_ax25_frame
    streams, outs = [], []
    for it in range(40):
        bits = [int(b) for b in rng.integers(0, 2, int(rng.integers(0, 300)))]
        if it % 2 == 0: bits = bits[:7] + ax25_bits("DST" + str(it), "SRC" + str(it), "msg %d" % it, 0) + bits[7:15]
        streams.append(np.array(bits, np.uint8)); r = decoders.decode_ax25_frame(bits); outs.append("<None>" if r is None else r)
    d["ax_len"] = np.array([len(b) for b in streams]); d["ax_bits"] = np.concatenate(streams) if streams else np.zeros(0, np.uint8)
    d["ax_out"] = np.array(json.dumps(outs))
    d["mtags"] = np.array(mtags); d["atags"] = np.array(atags)
    save("decoders", **d)


def gen_power():
    d = {}
    frames, pw = [], []
    for n, seed in [(1024, 51), (16384, 52), (32768, 53), (100, 54), (7, 55), (20000, 56), (40001, 57)]:
        iq = fm_iq(1, n, 2.4e6, seed, amp=0.05 * (1 + seed % 3))[0]
        p = sp.measure_signal_power(iq)
        assert p.dtype == np.float32
        d[f"iq_{n}"] = iq
        d[f"p_{n}"] = np.array(p)
        # numpy float32 mean (pairwise tree) of |x| and of |x|^2
        d[f"mean_abs_{n}"] = np.array(np.mean(np.abs(iq)))
        d[f"mean_abs2_{n}"] = np.array(np.mean(np.abs(iq) ** 2))
    d["p_zero"] = np.array(sp.measure_signal_power(np.zeros(1024, np.complex64)))
    save("power", **d)


def gen_scanner():
    d = {}
    for n, ns, seed in [(2048, 6, 61), (4096, 4, 62)]:
        fs = 2.4e6
        iq = scan_iq(ns, n, fs, seed)
        dbs, peaks, bws = [], [], []
        for s in iq:
            # the inline scanner's five statements, pyspecsdr.py:2542-2552
            spectrum = np.fft.fftshift(np.fft.fft(s))
            power_db = 10 * np.log10(np.abs(spectrum) ** 2 + 1e-10)
            peak = np.max(power_db)
            mask = power_db > (peak - 20)
            bw = np.sum(mask) * (fs / len(power_db))
            dbs.append(power_db); peaks.append(peak); bws.append(bw)
        d[f"iq_{n}"] = iq
        d[f"db_{n}"] = np.stack(dbs)
        d[f"peak_{n}"] = np.array(peaks)
        d[f"bw_{n}"] = np.array(bws)
        d[f"count_{n}"] = np.array([int(round(b / (fs / n))) for b in bws])
        assert d[f"db_{n}"].dtype == np.float32
    # the sweep driver's per-read statements, pyspecsdr.py:1049-1057 (scan_frequencies): a read of int(SCAN_DWELL_TIME * fs)
    # samples (not a power of two) against an absolute threshold
    for n, fs, thr, seed in [(24000, 240e3, -10.0, 63), (5000, 50e3, 5.0, 64), (2048, 2.4e6, 0.0, 65)]:
        iq = scan_iq(3, n, fs, seed)
        dbs, peaks, bws, cnts = [], [], [], []
        for s_ in iq:
            spectrum = np.fft.fftshift(np.fft.fft(s_))
            power_db = 10 * np.log10(np.abs(spectrum) ** 2 + 1e-10)
            max_power = np.max(power_db)
            mask = power_db > thr
            bandwidth = np.sum(mask) * (fs / len(power_db))
            dbs.append(power_db); peaks.append(max_power); bws.append(bandwidth); cnts.append(int(np.sum(mask)))
        d[f"sw_iq_{n}"] = iq; d[f"sw_db_{n}"] = np.stack(dbs); d[f"sw_peak_{n}"] = np.array(peaks)
        d[f"sw_bw_{n}"] = np.array(bws); d[f"sw_count_{n}"] = np.array(cnts); d[f"sw_args_{n}"] = np.array([fs, thr])
        assert d[f"sw_db_{n}"].dtype == np.float32
    d["sw_sizes"] = np.array([24000, 5000, 2048])
    save("scanner", **d)


def gen_caller():
    """Caller-side state machines: AGC stepper and the waterfall / persistence quantisers."""
    import curses
    sd = types.ModuleType("sounddevice")
    sd.PortAudioError = type("PortAudioError", (Exception,), {})
    sd.OutputStream = object
    so = types.ModuleType("SoapySDR")
    so.SOAPY_SDR_RX = 1
    so.SOAPY_SDR_CF32 = "CF32"
    so.Device = object
    sys.modules["sounddevice"] = sd
    sys.modules["SoapySDR"] = so
    curses.color_pair = lambda n: n << 8
    import signal as _signal
    old = (_signal.getsignal(_signal.SIGINT), _signal.getsignal(_signal.SIGTERM))
    import pyspecsdr as P
    _signal.signal(_signal.SIGINT, old[0]); _signal.signal(_signal.SIGTERM, old[1])

    class Sdr:
        valid_gains_db = list(np.arange(0, 50, 1.7))
        gain = 0.0

    d = {}
    powers = np.array([-50, -45, -31, -29, -10, -10, -33, -28.0001, -32.0, -31.99, 5, 5, 5], np.float32)
    for start in (20, 0, len(Sdr.valid_gains_db) - 1):
        idx, traj = start, []
        for p in powers:
            idx = P.adjust_gain(Sdr, p, idx)
            traj.append(idx)
        d[f"agc_traj_{start}"] = np.array(traj)
    d["agc_powers"] = powers
    d["agc_ngains"] = np.array(len(Sdr.valid_gains_db))

    class Scr:
        def __init__(s, h, w): s.h, s.w, s.calls = h, w, []
        def getmaxyx(s): return s.h, s.w
        def addstr(s, *a): s.calls.append(a)
        def refresh(s): pass

    H, W = 40, 120
    iq = fm_iq(34, 1024, 2.4e6, 71)
    iq[5] *= 3.0
    iq[20] *= 0.1
    rows = []
    for f in iq:
        fd = sp.compute_fft(f)
        fd = np.convolve(fd, np.ones(5) / 5, mode="valid")
        thr = np.median(fd) - 10
        fd[fd < thr] = thr
        rows.append(fd)
    rows = np.stack(rows)
    d["rows"] = rows
    d["hw"] = np.array([H, W])
    # waterfall: after each push, grid[y, x] = glyph code (0 '.',1 '-',2 '=',3 '#'), colour idx; -1 = not drawn
    P.WATERFALL_HISTORY.clear()
    wf_glyph, wf_col = [], []
    glyphs = {".": 0, "-": 1, "=": 2, "#": 3}
    for r in rows:
        scr = Scr(H, W)
        P.draw_waterfall(scr, r, None, 100e6, 2.4e6, 0, 0, None)
        g = -np.ones((H - 4, W - 8), np.int8); c = -np.ones((H - 4, W - 8), np.int8)
        for call in scr.calls:
            y, x, s, attr = call
            if s in glyphs and x >= 9 and y >= 3 and len(s) == 1 and (attr >> 8) >= 10:
                g[y - 3, x - 9] = glyphs[s]; c[y - 3, x - 9] = (attr >> 8) - 10
        wf_glyph.append(g); wf_col.append(c)
    d["wf_glyph"] = np.stack(wf_glyph)
    d["wf_colour"] = np.stack(wf_col)
    # persistence: list of hits (trace i, x, y, colour) in draw order -> final grid of colour (last writer wins)
    P.PERSISTENCE_HISTORY.clear()
    ps = []
    for r in rows[:14]:
        scr = Scr(H, W)
        P.draw_persistence(scr, r, None, 100e6, 2.4e6, 0, 0, None)
        g = np.zeros((H - 4, W - 8), np.int8)  # 0 = empty, else colour pair
        for call in scr.calls:
            y, x, s, attr = call
            if s == "*":
                g[y - 2, x - 8] = attr >> 8
        ps.append(g)
    d["ps_colour"] = np.stack(ps)
    # spectrum display (draw_spectrogram, pyspecsdr.py:398-498): final cell grid, glyph code (0 '.',1 '-',2 '=',3 '#',
    # 4 ' '), colour pair; -1 = cell never written.  Short rows and one full default read buffer (32768 samples).
    glyphs5 = {".": 0, "-": 1, "=": 2, "#": 3, " ": 4}
    big = fm_iq(1, 32768, 2.4e6, 72)[0]
    fd = sp.compute_fft(big)
    fd = np.convolve(fd, np.ones(5) / 5, mode="valid")
    thr = np.median(fd) - 10
    fd[fd < thr] = thr
    cases = [("r0", rows[0], 40, 120), ("r5", rows[5], 40, 120), ("r20", rows[20], 30, 87), ("big", fd, 50, 200)]
    for tag, r, hh, ww in cases:
        scr = Scr(hh, ww)
        P.draw_spectrogram(scr, r.copy(), None, 100e6, 2.4e6, 0, 0, None)
        dh, dw = hh - 4, ww - 7
        g = -np.ones((dh, dw), np.int8); c = -np.ones((dh, dw), np.int8)
        for call in scr.calls:
            if len(call) != 4:
                continue
            y, x, st, attr = call
            if len(st) == 1 and st in glyphs5 and x >= 7 and 2 <= y < 2 + dh and x - 7 < dw:
                g[y - 2, x - 7] = glyphs5[st]; c[y - 2, x - 7] = (attr >> 8) & 0xFF
        d[f"sg_row_{tag}"] = r
        d[f"sg_hw_{tag}"] = np.array([hh, ww])
        d[f"sg_glyph_{tag}"] = g
        d[f"sg_colour_{tag}"] = c
    d["sg_tags"] = np.array([c_[0] for c_ in cases])
    # gradient waterfall (draw_gradient_waterfall, pyspecsdr.py:1640-1716): 9 intensity characters ' ._-=+*#@' (code =
    # index), colour index 0..5; same ring as the plain waterfall.  -1 = not drawn.
    P.WATERFALL_HISTORY.clear()
    chars9 = ' ._-=+*#@'
    gw_g, gw_c = [], []
    for r in rows[:33]:
        scr = Scr(H, W)
        P.draw_gradient_waterfall(scr, r, None, 100e6, 2.4e6, 0, 0, None)
        g = -np.ones((H - 4, W - 10), np.int8); c = -np.ones((H - 4, W - 10), np.int8)
        for call in scr.calls:
            y, x, st, attr = call
            if len(st) == 1 and st in chars9 and 9 <= x < 9 + (W - 10) and 2 <= y < 2 + (H - 4) and (attr >> 8) >= 10:
                g[y - 2, x - 9] = chars9.index(st); c[y - 2, x - 9] = (attr >> 8) - 10
        gw_g.append(g); gw_c.append(c)
    d["gw_glyph"] = np.stack(gw_g)
    d["gw_colour"] = np.stack(gw_c)
    P.WATERFALL_HISTORY.clear()
    # surface plot (draw_surface_plot, pyspecsdr.py:1567-1616): '#' cells on the whole screen, colour pair 1..5; 0 = empty
    sf = []
    for r, hh, ww in ((rows[0], 40, 120), (rows[5], 40, 120), (fd, 50, 200)):
        scr = Scr(hh, ww)
        P.draw_surface_plot(scr, r.copy(), None, 100e6, 2.4e6, 0, 0, None)
        g = np.zeros((hh, ww), np.int8)
        for call in scr.calls:
            y, x, st, attr = call
            if st == "#":
                g[y, x] = attr >> 8
        sf.append(g)
    d["sf_colour_0"], d["sf_colour_1"], d["sf_colour_2"] = sf
    d["sf_cos_sin"] = np.array([np.cos(np.radians(P.SURFACE_ANGLE)), np.sin(np.radians(P.SURFACE_ANGLE))])
    # vector (constellation) display (draw_vector_display, pyspecsdr.py:1718-1752): 1 where a sample's '.' lands
    vs = (iq[3][:600] * np.float32(2.5)).astype(np.complex64)
    vs[7] = complex(-1.999, 1.999)
    vs[8] = complex(0.0, -0.0)
    d["vec_iq"] = vs
    for tag, hh, ww in (("a", 40, 120), ("b", 25, 81)):
        scr = Scr(hh, ww)
        P.draw_vector_display(scr, vs, 100e6, 2.4e6, 0, 0, None)
        g = np.zeros((hh, ww), np.int8)
        for call in scr.calls:
            if len(call) == 4 and call[2] == ".":
                g[call[0], call[1]] = 1
        d[f"vec_grid_{tag}"] = g
    save("caller", **d)


def gen_caller_iq():
    """The read buffers behind caller.npz's `rows` (the same seeded generator and scalings as gen_caller), stored so that a test can go
    from IQ to the reference's waterfall cells; checked here against the stored rows through the reference's compute_fft."""
    iq = fm_iq(34, 1024, 2.4e6, 71)
    iq[5] *= 3.0
    iq[20] *= 0.1
    rows = []
    for f in iq:
        fd = sp.compute_fft(f)
        fd = np.convolve(fd, np.ones(5) / 5, mode="valid")
        thr = np.median(fd) - 10
        fd[fd < thr] = thr
        rows.append(fd)
    have = np.load(os.path.join(OUT, "caller.npz"))["rows"]
    assert np.array_equal(np.stack(rows), have), "caller.npz was generated from different read buffers"
    save("caller_iq", iq=iq, db=np.stack([sp.compute_fft(f) for f in iq]))


if __name__ == "__main__":
    gens = [gen_atan2, gen_atan2f_bits, gen_spectrum, gen_nfm, gen_am_ssb, gen_power, gen_iqcorr, gen_wfm, gen_bandpass, gen_afsk, gen_classify, gen_classify_short, gen_log10f,
            gen_decoders, gen_scanner, gen_caller, gen_caller_iq]
    want = sys.argv[1:]                      # e.g. `python tools/make_goldens.py decoders` regenerates one fixture
    for g in gens:
        if not want or g.__name__[4:] in want:
            g()
