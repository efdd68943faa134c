"""Drop-in for the reference's hot-path module `signal_processing.py` (xqtr/PySpecSDR), backed by libpss.so.

Same callables, positional arguments, defaults, return dtypes/shapes and error behaviour as the functions
the reference's main loop imports with `from signal_processing import *` (pyspecsdr.py:98):

    compute_fft(samples)                                   signal_processing.py:243-264  -> float64 (N,), writable
    demodulate_signal(samples, sample_rate, mode='NFM')    :220-240
    demodulate_nfm(samples, sample_rate, target_rate=...)  :91-116   -> float64 (n_out, 2)
    demodulate_wfm(samples, sample_rate, target_rate=...)  :119-176  -> float64 (n_out, 2), left/right
    demodulate_am(samples)                                 :179-195  -> float64 (N, 2)
    demodulate_ssb(samples, sample_rate, lower=True)       :198-217  -> float64 (N, 2)
    measure_signal_power(samples)                          :325-328  -> np.float32
    mono_to_stereo(mono_audio)                             :83-88

    iq_correction(samples)                                 :46-80    -> complex64 (N,)  ('RAW' mode = its real part)
    bandpass_filter(data, lowcut, highcut, sample_rate)    :34-42    -> float64 (N,)   (decoders.py:100-101)
    classify_signal(samples, sample_rate, bandwidth)       :296-322  -> label str (the reference raises NameError: see the function)
    estimate_modulation_index(samples)                     :283-293  -> np.float32

Every call runs on the GPU through the C ABI; there is no NumPy/SciPy compute path here.
"""
import numpy as np

from . import _lib as L
from .engine import Engine

# what `from signal_processing import *` hands to the caller (pyspecsdr.py:98): the reference's public callables
__all__ = ["bandpass_filter", "iq_correction", "mono_to_stereo", "demodulate_nfm", "demodulate_wfm", "demodulate_am",
           "demodulate_ssb", "demodulate_signal", "compute_fft", "estimate_modulation_index", "classify_signal",
           "measure_signal_power", "DEFAULT_SAMPLE_RATE", "BUTTER_ORDER"]

DEFAULT_SAMPLE_RATE = 22050  # pyspecconst.py:3
BUTTER_ORDER = 5             # pyspecconst.py:5

_engine = None


def get_engine(device=0):
    global _engine
    if _engine is None:
        _engine = Engine(device)
    return _engine


# ---- coefficient tables --------------------------------------------------------------------------------------------
# The reference designs its filters with SciPy on every call; libpss.so designs them natively once per sample rate
# (pss_design_*): SciPy 1.15's own bits on NumPy's AVX512_SKX dispatch (DESIGN.md par. 2).  A host that runs the reference has SciPy by
# definition, so this shim asks THAT SciPy — whatever its version and NumPy's dispatch — for the tables once
# per sample rate and injects them (pss_set_*_filters): the GPU path then reproduces the reference's float64 bits
# whatever SciPy version the host carries.  Only the coefficient design happens here — never sample data.
USE_SCIPY_DESIGNS = True
_designed = set()


_target_rate = float(DEFAULT_SAMPLE_RATE)     # what the engine's decimator is currently designed for


def _set_target_rate(target_rate):
    """demodulate_nfm / _wfm(target_rate=...): decimation factor int(sample_rate / target_rate) (signal_processing.py:111)."""
    global _target_rate
    tr = float(target_rate)
    if tr != _target_rate:
        get_engine().set_target_rate(tr)       # drops the engine's cached decimator designs
        for k in [k for k in _designed if k[0] in ('nfm', 'wfm')]:
            _designed.discard(k)
        _target_rate = tr


class _rate:
    """`with _rate(target_rate):` — the engine's decimator is designed for target_rate inside the block and for the default 22 050 Hz after it,
    whatever happens inside: a non-default rate never outlives the call that asked for it (demodulate_pcm, formats.demodulate_batch and every
    direct get_engine() batch / pipeline call see the reference's default, as demodulate_signal + write_to_pipe would)."""

    def __init__(self, target_rate):
        self.tr = target_rate

    def __enter__(self):
        _set_target_rate(self.tr)

    def __exit__(self, *exc):
        _set_target_rate(DEFAULT_SAMPLE_RATE)
        return False


def _inject_designs(kind, fs):
    key = (kind, float(fs))
    if not USE_SCIPY_DESIGNS or key in _designed:
        return
    try:
        import scipy.signal as ss
    except ImportError:
        _designed.add(key)
        return
    e = get_engine()
    fs = float(fs)
    q = int(fs / _target_rate)
    if kind == 'wfm':
        nyq = fs / 2
        lp = ss.butter(BUTTER_ORDER, 15000 / nyq, btype='low', output='sos')                     # :126 -> :39
        pil = ss.butter(BUTTER_ORDER, [18800 / nyq, 19200 / nyq], btype='band', output='sos')    # :129 -> :41
        lmr = ss.butter(BUTTER_ORDER, [23000 / nyq, 53000 / nyq], btype='band', output='sos')    # :133
        e.set_wfm_filters(fs, lp, pil, lmr, float(np.exp(-1 / (75e-6 * fs))))                    # :145
    if kind in ('nfm', 'wfm') and q > 1:
        taps = ss.firwin(numtaps=65, cutoff=15000 / (fs / 2))                 # :107 (ValueError as in the reference)
        sos = ss.cheby1(8, 0.05, 0.8 / q, output='sos')                       # decimate(), scipy _signaltools.py
        e.set_nfm_filters(fs, taps, sos, ss.sosfilt_zi(sos))
    if kind == 'ssb':
        e.set_ssb_taps(fs, ss.firwin(65, 3000 / fs, window='hamming'))                           # :203/:208
    _designed.add(key)


_warned_narrowing = False


def _samples(samples):
    """The reference's read buffer is complex64 (pyspecsdr.py:1885-1891) and the kernels replay NumPy's complex64
    arithmetic; wider input (complex128) makes the reference compute in float64.  compute_fft (power-of-two lengths),
    demodulate_am, demodulate_ssb and measure_signal_power serve such buffers in float64 (pss_h_*_c128); everywhere else it is narrowed, with a
    one-time warning."""
    global _warned_narrowing
    s = np.asarray(samples)
    if s.ndim != 1:
        raise ValueError("samples must be a 1-D complex array")
    if s.dtype == np.complex128 and not _warned_narrowing:
        import warnings
        warnings.warn("pyspecsdr_amd.signal_processing: complex128 samples are narrowed to complex64 (the reference's SDR read "
                      "buffer type); the reference would have computed this call in float64", RuntimeWarning, stacklevel=3)
        _warned_narrowing = True
    return np.ascontiguousarray(s, dtype=np.complex64)


def mono_to_stereo(mono_audio):
    stereo_audio = np.zeros((len(mono_audio), 2))
    stereo_audio[:, 0] = mono_audio
    stereo_audio[:, 1] = mono_audio
    return stereo_audio


def bandpass_filter(data, lowcut, highcut, sample_rate):
    """signal_processing.py:34-42 (imported by decoders.py:3): butter(5) low-/band-pass SOS + sosfilt -> float64 (N,)."""
    d = np.asarray(data)
    sos = None
    if USE_SCIPY_DESIGNS:
        try:
            import scipy.signal as ss
            nyq = sample_rate / 2
            sos = (ss.butter(BUTTER_ORDER, highcut / nyq, btype='low', output='sos') if lowcut <= 0 else
                   ss.butter(BUTTER_ORDER, [lowcut / nyq, highcut / nyq], btype='band', output='sos'))
        except ImportError:
            pass
    e = get_engine()
    one = lambda row: e.h_bandpass_filter(row, lowcut, highcut, sample_rate, sos)
    if d.ndim == 0:
        raise ValueError("bandpass_filter: sosfilt needs at least one axis")           # (sosfilt raises on a 0-d input as well)
    # sosfilt filters along the last axis; complex input: real coefficients act on the two components separately (result complex128)
    rows = d.reshape(-1, d.shape[-1])
    if np.iscomplexobj(d):
        out = np.stack([one(np.ascontiguousarray(r.real)) + 1j * one(np.ascontiguousarray(r.imag)) for r in rows]) if len(rows) else \
            np.zeros(rows.shape, np.complex128)
    else:
        out = np.stack([one(np.ascontiguousarray(r)) for r in rows]) if len(rows) else np.zeros(rows.shape, np.float64)
    return out.reshape(d.shape)


# signal_processing.py:296-322 calls `welch`, which the module never imports: in the reference the function
# raises NameError on every call (swallowed by the scanner's try/except, pyspecsdr.py:2571).  The drop-in default is the
# reference's PRESENT behaviour (a drop-in must not change what the application does); set CLASSIFY_RAISES_NAMEERROR = False
# — or start the launcher with --fix-classify — to run the function as it is written, i.e. as it behaves once
# `from scipy.signal import welch` is added.  classify_signal_features() is always available.
CLASSIFY_RAISES_NAMEERROR = True


def classify_signal_features(samples, sample_rate):
    """-> (label, signal_bw, modulation_index, spectral_flatness): the label of classify_signal and the three numbers it
    is decided on (estimate_bandwidth :267-280, estimate_modulation_index :283-293, flatness :304)."""
    x = _samples(samples)
    if len(x) < 1:
        raise ValueError("classify_signal: empty read (scipy.signal.welch raises on it)")
    return get_engine().h_classify_signal(x, sample_rate)


def classify_signal(samples, sample_rate, bandwidth):
    """signal_processing.py:296-322 -> 'FM_BROADCAST' | 'NARROW_FM' | 'AM_BROADCAST' | 'SSB' | 'DIGITAL' | 'UNKNOWN'
    (`bandwidth` is unused, as in the reference)."""
    if CLASSIFY_RAISES_NAMEERROR:
        raise NameError("name 'welch' is not defined")
    return classify_signal_features(samples, sample_rate)[0]


def estimate_modulation_index(samples):
    """signal_processing.py:283-293 -> np.float32, bit-exact with NumPy's float32 evaluation."""
    x = _samples(samples)
    if len(x) < 1:
        return np.float32("nan")              # np.var of empty arrays (NumPy warns and returns nan)
    return get_engine().h_classify_signal(x, 1.0)[2]


def iq_correction(samples):
    """signal_processing.py:46-80 -> complex64 (N,)."""
    return get_engine().h_iq_correction(_samples(samples))


def _is_c128(samples):
    s = np.asarray(samples)
    return s.ndim == 1 and s.dtype == np.complex128


def compute_fft(samples):
    # a complex128 buffer of a power-of-two length: float64 from the window product on, as the reference computes it (no narrowing)
    if _is_c128(samples) and 16 <= len(samples) <= 65536 and (len(samples) & (len(samples) - 1)) == 0:
        return get_engine().h_compute_fft_c128(samples)
    return get_engine().h_compute_fft(_samples(samples))


def demodulate_nfm(samples, sample_rate, target_rate=DEFAULT_SAMPLE_RATE):
    with _rate(target_rate):
        _inject_designs('nfm', sample_rate)
        audio, _ = get_engine().h_demodulate(L.MODE_NFM, _samples(samples), sample_rate)
    return audio


def demodulate_wfm(samples, sample_rate, target_rate=DEFAULT_SAMPLE_RATE):
    """signal_processing.py:119-176 -> float64 (n_out, 2) = column_stack((left, right)).  (The reference's RDS hooks at
    :165-174 call undefined names inside a try/except and never produce anything; nothing to mirror.)"""
    with _rate(target_rate):
        _inject_designs('wfm', sample_rate)
        audio, _ = get_engine().h_demodulate(L.MODE_WFM, _samples(samples), sample_rate)
    return audio


def demodulate_am(samples):
    if _is_c128(samples) and len(samples) >= 1:        # float64 np.abs / np.mean, as the reference computes it for such a buffer (no narrowing)
        return get_engine().h_demodulate_am_c128(samples)[0]
    audio, _ = get_engine().h_demodulate(L.MODE_AM, _samples(samples), float(DEFAULT_SAMPLE_RATE))
    return audio


def demodulate_ssb(samples, sample_rate, lower=True):
    _inject_designs('ssb', sample_rate)
    if _is_c128(samples) and len(samples) >= 1:        # lfilter's complex128 convolution on the samples as they are (no narrowing)
        return get_engine().h_demodulate_ssb_c128(samples, sample_rate, lower)[0]
    audio, _ = get_engine().h_demodulate(L.MODE_LSB if lower else L.MODE_USB, _samples(samples), sample_rate)
    return audio


def demodulate_signal(samples, sample_rate, mode='NFM'):
    if mode == 'NFM':
        return demodulate_nfm(samples, sample_rate)
    elif mode == 'AM':
        return demodulate_am(samples)
    elif mode == 'USB':
        return demodulate_ssb(samples, sample_rate, lower=False)
    elif mode == 'LSB':
        return demodulate_ssb(samples, sample_rate, lower=True)
    elif mode == 'RAW':
        # :222-225 + :238 — every non-voice mode is IQ-corrected first; RAW then returns the I samples (float32)
        return get_engine().h_raw(_samples(samples))
    elif mode == 'WFM':
        _inject_designs('wfm', sample_rate)
        audio, _ = get_engine().h_demodulate_signal(L.MODE_WFM, _samples(samples), sample_rate)  # :222-228
        return audio
    return np.zeros((len(samples), 2))  # unknown mode -> silence of shape (n, 2), as the reference


def demodulate_pcm(samples, sample_rate, mode='NFM'):
    """int16 (n_out, 2) exactly as io_manager.write_to_pipe would produce from demodulate_signal()'s output."""
    m = {'NFM': L.MODE_NFM, 'AM': L.MODE_AM, 'USB': L.MODE_USB, 'LSB': L.MODE_LSB, 'WFM': L.MODE_WFM}[mode]
    fs = float(DEFAULT_SAMPLE_RATE) if mode == 'AM' else sample_rate
    if mode != 'AM':
        _inject_designs({'NFM': 'nfm', 'WFM': 'wfm'}.get(mode, 'ssb'), fs)
    _, pcm = get_engine().h_demodulate_signal(m, _samples(samples), fs)
    return pcm


def measure_signal_power(samples):
    if _is_c128(samples) and len(samples) >= 1:
        # float64 as the reference computes it for such a buffer: the mean power on the device, the scalar log10 with NumPy's own float64 log10
        power = get_engine().h_mean_power_c128(samples)
        return 10 * np.log10(power + 1e-10)
    return get_engine().h_measure_power(_samples(samples))
