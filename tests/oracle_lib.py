"""ctypes binding of the CPU oracle (oracle/pss_oracle.c) — test infrastructure only.

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never from the
product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ODIR = os.path.join(ROOT, "oracle")
SO = os.path.join(ODIR, "_build", "libpss_oracle.so")

_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_i16p = np.ctypeslib.ndpointer(np.int16, flags="C_CONTIGUOUS")
_i8p = np.ctypeslib.ndpointer(np.int8, flags="C_CONTIGUOUS")


def build(force=False):
    src = [os.path.join(ODIR, f) for f in ("pss_oracle.c", "pss_pocketfft.c", "pss_oracle.h", "Makefile")]
    if force or not os.path.exists(SO) or any(os.path.getmtime(s) > os.path.getmtime(SO) for s in src):
        subprocess.run(["make", "-C", ODIR], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        L.pss_o_rcp14f.restype = C.c_float
        L.pss_o_rcp14f.argtypes = [C.c_float]
        L.pss_o_atan2f.restype = C.c_float
        L.pss_o_atan2f.argtypes = [C.c_float, C.c_float]
        L.pss_o_cabsf.restype = C.c_float
        L.pss_o_cabsf.argtypes = [C.c_float, C.c_float]
        L.pss_o_pairwise_sum_f32.restype = C.c_float
        L.pss_o_pairwise_sum_f32.argtypes = [_f32p, C.c_long]
        L.pss_o_cabs.restype, L.pss_o_cabs.argtypes = C.c_double, [C.c_double, C.c_double]
        L.pss_o_compute_fft.argtypes = [_f32p, C.c_int, _f64p]
        L.pss_o_postprocess.argtypes = [_f64p, C.c_int, _f64p]
        L.pss_o_iq_correction.argtypes = [_f32p, C.c_int, _f32p]
        L.pss_o_demod_wfm.argtypes = [_f32p, C.c_int, C.c_int, _f64p, _f64p, _f64p, C.c_double, _f64p, _f64p, _f64p, _f64p]
        L.pss_o_sosfilt.argtypes = [_f64p, C.c_int, _f64p, C.c_long, _f64p]
        L.pss_o_spectrogram_cells.argtypes = [_f64p, C.c_int, C.c_int, C.c_int, _i8p, _i8p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.pss_o_gradient_cells.argtypes = [_f64p, C.c_int, C.c_int, C.c_int, C.c_int, _i8p, _i8p]
        L.pss_o_surface_cells.argtypes = [_f64p, C.c_int, C.c_int, C.c_int, _i8p]
        L.pss_o_vector_cells.argtypes = [_f32p, C.c_int, C.c_int, C.c_int, _i8p]
        L.pss_o_afsk_bits.argtypes = [_f64p, C.c_int, C.c_double, _f64p, _f64p, C.c_int, np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")]
        L.pss_o_classify.restype = C.c_int
        L.pss_o_classify.argtypes = [_f32p, C.c_long, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_float), C.POINTER(C.c_float), _f32p]
        L.pss_o_hann1024_f32.argtypes = [_f32p]
        L.pss_o_log10f_np_many.argtypes = [_f32p, _f32p, C.c_long]
        L.pss_o_atan2f_many.argtypes = [_f32p, _f32p, _f32p, C.c_long]
        L.pss_o_cabsf_many.argtypes = [_f32p, _f32p, _f32p, C.c_long]
        L.pss_o_hann_f32.argtypes = [_f32p, C.c_int]
        _i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
        L.pss_o_morse_edges.restype = None
        L.pss_o_morse_edges.argtypes = [_f32p, C.c_long, _i32p, _i32p, C.c_long, C.POINTER(C.c_long), C.POINTER(C.c_long)]
        L.pss_o_power_db.restype = C.c_float
        L.pss_o_power_db.argtypes = [_f32p, C.c_int]
        L.pss_o_scan_threshold.restype = C.c_int
        L.pss_o_scan_threshold.argtypes = [_f32p, C.c_int, C.c_double, C.c_double, _f32p, C.POINTER(C.c_float), C.POINTER(C.c_double)]
        L.pss_o_scan_slice.restype = C.c_int
        L.pss_o_scan_slice.argtypes = [_f32p, C.c_int, C.c_double, _f32p, C.POINTER(C.c_float),
                                       C.POINTER(C.c_double)]
        L.pss_o_demod_nfm.restype = C.c_int
        L.pss_o_demod_nfm.argtypes = [_f32p, C.c_int, C.c_double, C.c_int, _f64p, _f64p, _f64p, _f64p,
                                      C.c_void_p, C.c_void_p]
        L.pss_o_demod_am.argtypes = [_f32p, C.c_int, _f64p, C.c_int, _f64p]
        L.pss_o_demod_ssb.argtypes = [_f32p, C.c_int, _f64p, _f64p]
        L.pss_o_demod_ssb_ex.argtypes = [_f32p, C.c_int, _f64p, _f64p, C.c_int]
        for f in ("pss_o_hilbert", "pss_o_rfft_full", "pss_o_cifft"):
            getattr(L, f).restype = None
            getattr(L, f).argtypes = [_f64p, C.c_int, _f64p]
        L.pss_o_pocketfft_twiddles.restype = None
        L.pss_o_pocketfft_twiddles.argtypes = [C.c_int, _f64p]
        L.pss_o_pcm16_stereo.argtypes = [_f64p, C.c_int, _i16p]
        L.pss_o_agc_step.restype = C.c_int
        L.pss_o_agc_step.argtypes = [C.c_float, C.c_int, C.c_int]
        L.pss_o_waterfall_cells.argtypes = [_f64p, C.c_int, C.c_int, C.c_int, C.c_int, _i8p, _i8p]
        L.pss_o_persistence_cells.argtypes = [_f64p, C.c_int, C.c_int, C.c_int, C.c_int, _i8p]
        L.pss_o_batch_spectrum_nfm.argtypes = [_f32p, C.c_long, C.c_int, C.c_double, C.c_int, _f64p, _f64p,
                                               _f64p, _f32p, _i16p, C.c_int]
        L.pss_o_batch_spectrum_post_nfm.argtypes = [_f32p, C.c_long, C.c_int, C.c_double, C.c_int, _f64p, _f64p, _f64p, _f32p,
                                                    C.c_void_p, C.c_void_p, C.c_void_p, _i16p, C.c_int]
        L.pss_o_waterfall_rows.argtypes = [_f32p, C.c_long, C.c_int, C.c_int, C.c_int, _i8p, _i8p, C.c_int]
        L.pss_o_batch_headline_f64.argtypes = [_f32p, C.c_long, C.c_int, C.c_double, C.c_int, _f64p, _f64p, _f64p, C.c_void_p, C.c_void_p,
                                               _f64p, _f64p, C.c_void_p, C.c_int]
        L.pss_o_waterfall_rows_f64.argtypes = [_f64p, _f64p, _f64p, C.c_long, C.c_int, C.c_int, C.c_int, _i8p, _i8p, C.c_int]
        L.pss_o_persistence_rows_f64.argtypes = [_f64p, _f64p, _f64p, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, _i8p, C.c_int]
        L.pss_o_persistence_rows.argtypes = [_f32p, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, _i8p, C.c_int]
        _lib = L
    return _lib


def _iq(x):
    x = np.ascontiguousarray(x, dtype=np.complex64)
    return x.view(np.float32)


def atan2f(y, x):
    y = np.ascontiguousarray(y, np.float32)
    x = np.ascontiguousarray(x, np.float32)
    out = np.empty_like(y)
    lib().pss_o_atan2f_many(y, x, out, y.size)
    return out


def cabsf(re, im):
    """np.abs(re + 1j im) on complex64, element by element."""
    re = np.ascontiguousarray(re, np.float32)
    im = np.ascontiguousarray(im, np.float32)
    out = np.empty_like(re)
    lib().pss_o_cabsf_many(re, im, out, re.size)
    return out


def compute_fft(iq):
    out = np.empty(len(iq), np.float64)
    lib().pss_o_compute_fft(_iq(iq), len(iq), out)
    return out


def compute_fft_c128(iq):
    """compute_fft of a complex128 buffer (float64 window product)."""
    x = np.ascontiguousarray(iq, np.complex128)
    out = np.empty(len(x), np.float64)
    f = lib().pss_o_compute_fft_c128
    f.argtypes, f.restype = [_f64p, C.c_int, _f64p], None
    f(x.view(np.float64), len(x), out)
    return out


def mean_power_c128(iq):
    """np.mean(np.abs(x) ** 2) of a complex128 buffer in float64 (the array part of measure_signal_power)."""
    x = np.ascontiguousarray(iq, np.complex128)
    f = lib().pss_o_mean_power_c128
    f.argtypes, f.restype = [_f64p, C.c_int, _f64p], C.c_double
    return np.float64(f(x.view(np.float64), len(x), np.empty(len(x), np.float64)))


def demod_am_c128(iq, sos):
    """demodulate_am of a complex128 buffer (float64 np.abs / np.mean)."""
    x = np.ascontiguousarray(iq, np.complex128)
    sos = np.ascontiguousarray(sos, np.float64)
    out = np.empty(len(x), np.float64)
    f = lib().pss_o_demod_am_c128
    f.argtypes, f.restype = [_f64p, C.c_int, _f64p, C.c_int, _f64p], None
    f(x.view(np.float64), len(x), sos, sos.shape[0], out)
    return out


def postprocess(db):
    db = np.ascontiguousarray(db, np.float64)
    out = np.empty(len(db) - 4, np.float64)
    lib().pss_o_postprocess(db, len(db), out)
    return out


def iq_correction(iq):
    out = np.empty(len(iq), np.complex64)
    lib().pss_o_iq_correction(_iq(iq), len(iq), out.view(np.float32))
    return out


def demod_wfm(iq, fs, filt, target_rate=22050):
    """filt: dict with lp_sos, pilot_sos, lmr_sos, alpha, dec_sos, dec_zi.  Returns (n_out, 2) float64 or None (ValueError)."""
    n = len(iq)
    q = int(fs / target_rate)
    cap = max(n, 1)
    left, right = np.empty(cap), np.empty(cap)
    c = lambda a: np.ascontiguousarray(a, np.float64)
    r = lib().pss_o_demod_wfm(_iq(iq), n, q, c(filt["lp_sos"]), c(filt["pilot_sos"]), c(filt["lmr_sos"]),
                              float(filt["alpha"]), c(filt["dec_sos"]), c(filt["dec_zi"]), left, right)
    if r < 0:
        return None
    return np.stack([left[:r], right[:r]], axis=1)


def sosfilt(sos, x):
    sos = np.ascontiguousarray(sos, np.float64)
    x = np.ascontiguousarray(x, np.float64)
    y = np.empty_like(x)
    lib().pss_o_sosfilt(sos, sos.shape[0], x, len(x), y)
    return y


def spectrogram_cells(row, disp_h, disp_w):
    row = np.ascontiguousarray(row, np.float64)
    gl, co = np.empty((disp_h, disp_w), np.int8), np.empty((disp_h, disp_w), np.int8)
    dmin, dmax = C.c_double(), C.c_double()
    lib().pss_o_spectrogram_cells(row, len(row), disp_h, disp_w, gl, co, C.byref(dmin), C.byref(dmax))
    return gl, co, dmin.value, dmax.value


def gradient_cells(rows, disp_h, disp_w):
    rows = np.ascontiguousarray(rows, np.float64)
    gl, co = np.empty((disp_h, disp_w), np.int8), np.empty((disp_h, disp_w), np.int8)
    lib().pss_o_gradient_cells(rows, rows.shape[0], rows.shape[1], disp_h, disp_w, gl, co)
    return gl, co


def surface_cells(row, max_h, max_w):
    row = np.ascontiguousarray(row, np.float64)
    co = np.empty((max_h, max_w), np.int8)
    lib().pss_o_surface_cells(row, len(row), max_h, max_w, co)
    return co


def vector_cells(iq, max_h, max_w):
    g = np.empty((max_h, max_w), np.int8)
    lib().pss_o_vector_cells(_iq(iq), len(iq), max_h, max_w, g)
    return g


def afsk_bits(x, fs, sos1200, sos2200):
    x = np.ascontiguousarray(x, np.float64)
    bits = np.empty(max(len(x), 1), np.uint8)
    c = lambda a: np.ascontiguousarray(a, np.float64)
    nb = lib().pss_o_afsk_bits(x, len(x), fs, c(sos1200), c(sos2200), 5, bits)
    return bits[:nb].copy()


CLASS_LABELS = ("UNKNOWN", "FM_BROADCAST", "NARROW_FM", "AM_BROADCAST", "SSB", "DIGITAL")


def classify(iq, fs):
    """-> (label, signal_bw float64, modulation_index float32, spectral_flatness float32, psd float32[min(n, 1024)])."""
    bw, mi, fl = C.c_double(), C.c_float(), C.c_float()
    psd = np.empty(1024, np.float32)
    lab = lib().pss_o_classify(_iq(iq), len(iq), fs, C.byref(bw), C.byref(mi), C.byref(fl), psd)
    if lab < 0:
        raise ValueError("classify: empty read")
    return CLASS_LABELS[lab], bw.value, np.float32(mi.value), np.float32(fl.value), psd[:min(len(iq), 1024)].copy()


def log10f(x):
    """np.log10 of a float32 array as NumPy's AVX512_SKX dispatch (SVML) evaluates it."""
    x = np.ascontiguousarray(x, np.float32)
    y = np.empty_like(x)
    lib().pss_o_log10f_np_many(x, y, x.size)
    return y


def hann1024():
    w = np.empty(1024, np.float32)
    lib().pss_o_hann1024_f32(w)
    return w


def hann(n):
    w = np.empty(n, np.float32)
    lib().pss_o_hann_f32(w, n)
    return w


def morse_edges(iq, threshold=-20):
    """-> (rise_times, fall_times) int32 arrays of decode_morse (decoders.py:159-161); threshold in dB (the reference's own: -20)."""
    n = len(iq)
    rise, fall = np.empty(max(n, 1), np.int32), np.empty(max(n, 1), np.int32)
    nr, nf = C.c_long(), C.c_long()
    if float(threshold) != -20.0:
        L = lib()
        L.pss_o_morse_edges_thr.restype = None
        L.pss_o_morse_edges_thr.argtypes = [_f32p, C.c_long, C.c_double, np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS"),
                                            np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS"), C.c_long, C.POINTER(C.c_long), C.POINTER(C.c_long)]
        L.pss_o_morse_edges_thr(_iq(iq), n, float(threshold), rise, fall, n, C.byref(nr), C.byref(nf))
        return rise[:nr.value].copy(), fall[:nf.value].copy()
    lib().pss_o_morse_edges(_iq(iq), n, rise, fall, n, C.byref(nr), C.byref(nf))
    return rise[:nr.value].copy(), fall[:nf.value].copy()


def power_db(iq):
    return np.float32(lib().pss_o_power_db(_iq(iq), len(iq)))


def scan_threshold(iq, fs, threshold_db):
    """-> (db float32[n], max_power float32, bandwidth float64, count) of pyspecsdr.py:1049-1057."""
    db = np.empty(len(iq), np.float32)
    pk, bw = C.c_float(), C.c_double()
    cnt = lib().pss_o_scan_threshold(_iq(iq), len(iq), fs, threshold_db, db, C.byref(pk), C.byref(bw))
    return db, np.float32(pk.value), bw.value, cnt


def scan_slice(iq, fs):
    db = np.empty(len(iq), np.float32)
    pk = C.c_float()
    bw = C.c_double()
    cnt = lib().pss_o_scan_slice(_iq(iq), len(iq), fs, db, C.byref(pk), C.byref(bw))
    return db, np.float32(pk.value), bw.value, cnt


def demod_nfm(iq, fs, taps, sos, zi, stages=False, target_rate=22050):
    n = len(iq)
    q = int(fs / target_rate)
    n_out = max((n - 1 + q - 1) // q, 0)
    audio = np.empty(max(n_out, 1), np.float64)
    disc = np.empty(max(n - 1, 1), np.float32)
    fir = np.empty(max(n - 1, 1), np.float64)
    r = lib().pss_o_demod_nfm(_iq(iq), n, fs, q, np.ascontiguousarray(taps, np.float64),
                              np.ascontiguousarray(sos, np.float64), np.ascontiguousarray(zi, np.float64),
                              audio, disc.ctypes.data, fir.ctypes.data)
    if r < 0:
        raise ValueError("The length of the input vector x must be greater than padlen, which is 27.")
    if stages:
        return audio[:r], disc[:n - 1], fir[:n - 1]
    return audio[:r]


def demod_am(iq, sos):
    sos = np.ascontiguousarray(sos, np.float64)
    out = np.empty(len(iq), np.float64)
    lib().pss_o_demod_am(_iq(iq), len(iq), sos, sos.shape[0], out)
    return out


def demod_ssb(iq, taps, hilbert=True):
    """hilbert=False: without the hilbert() round trip (the library's option "ssb_hilbert" = 0)."""
    out = np.empty(len(iq), np.float64)
    lib().pss_o_demod_ssb_ex(_iq(iq), len(iq), np.ascontiguousarray(taps, np.float64), out, 1 if hilbert else 0)
    return out


def demod_ssb_c128(iq, taps, hilbert=True):
    """demodulate_ssb of a complex128 buffer."""
    x = np.ascontiguousarray(iq, np.complex128)
    out = np.empty(len(x), np.float64)
    f = lib().pss_o_demod_ssb_c128
    f.argtypes, f.restype = [_f64p, C.c_int, _f64p, _f64p, C.c_int], None
    f(x.view(np.float64), len(x), np.ascontiguousarray(taps, np.float64), out, 1 if hilbert else 0)
    return out


def hilbert(x):
    """scipy.signal.hilbert of a real float64 row of 2^k samples (pss_pocketfft.c)."""
    x = np.ascontiguousarray(x, np.float64)
    out = np.empty(2 * len(x), np.float64)
    lib().pss_o_hilbert(x, len(x), out)
    return out.view(np.complex128)


def pocketfft_twiddles(n):
    out = np.empty(2 * n, np.float64)
    lib().pss_o_pocketfft_twiddles(n, out)
    return out.view(np.complex128)


def pcm16_stereo(audio):
    audio = np.ascontiguousarray(audio, np.float64)
    out = np.empty((len(audio), 2), np.int16)
    lib().pss_o_pcm16_stereo(audio, len(audio), out.reshape(-1))
    return out


def agc_step(p, idx, n):
    return lib().pss_o_agc_step(float(np.float32(p)), idx, n)


def waterfall_cells(rows, disp_h, disp_w):
    rows = np.ascontiguousarray(rows, np.float64)
    g = np.empty((disp_h, disp_w), np.int8)
    c = np.empty((disp_h, disp_w), np.int8)
    lib().pss_o_waterfall_cells(rows, rows.shape[0], rows.shape[1], disp_h, disp_w, g, c)
    return g, c


def persistence_cells(rows, disp_h, disp_w):
    rows = np.ascontiguousarray(rows, np.float64)
    c = np.empty((disp_h, disp_w), np.int8)
    lib().pss_o_persistence_cells(rows, rows.shape[0], rows.shape[1], disp_h, disp_w, c)
    return c


def batch_spectrum_nfm(iq2d, fs, taps, sos, zi, n_threads=1):
    iq2d = np.ascontiguousarray(iq2d, np.complex64)
    nf, n = iq2d.shape
    q = int(fs / 22050)
    n_out = (n - 1 + q - 1) // q
    db = np.empty((nf, n), np.float32)
    pcm = np.empty((nf, n_out, 2), np.int16)
    lib().pss_o_batch_spectrum_nfm(iq2d.view(np.float32).reshape(-1), nf, n, fs, q,
                                   np.ascontiguousarray(taps, np.float64), np.ascontiguousarray(sos, np.float64),
                                   np.ascontiguousarray(zi, np.float64), db.reshape(-1), pcm.reshape(-1), n_threads)
    return db, pcm


class HeadlineBuffers:
    """Preallocated outputs of the full BASELINE cfg-2 step on the CPU (bench.py's cpu_baseline leg re-runs the step
    several times; allocating 0.5 GB of outputs per call would be timed as well)."""

    def __init__(self, nf, n, fs, disp_w=112):
        self.q = int(fs / 22050)
        self.n_out = (n - 1 + self.q - 1) // self.q
        self.db = np.empty((nf, n), np.float32)
        self.post = np.empty((nf, n - 4), np.float32)
        self.lo = np.empty(nf, np.float32)
        self.hi = np.empty(nf, np.float32)
        self.pcm = np.empty((nf, self.n_out, 2), np.int16)
        self.glyph = np.empty((nf, disp_w), np.int8)
        self.colour = np.empty((nf, disp_w), np.int8)


def batch_headline(iq2d, fs, taps, sos, zi, buf, n_threads=1, window=30):
    """compute_fft + post-process + waterfall line + NFM -> int16 for every frame (what one bench.py step does on the GPU)."""
    iq2d = np.ascontiguousarray(iq2d, np.complex64)
    nf, n = iq2d.shape
    L = lib()
    L.pss_o_batch_spectrum_post_nfm(iq2d.view(np.float32).reshape(-1), nf, n, fs, buf.q, np.ascontiguousarray(taps, np.float64),
                                    np.ascontiguousarray(sos, np.float64), np.ascontiguousarray(zi, np.float64),
                                    buf.db[:nf].reshape(-1), buf.post[:nf].ctypes.data, buf.lo[:nf].ctypes.data,
                                    buf.hi[:nf].ctypes.data, buf.pcm[:nf].reshape(-1), n_threads)
    L.pss_o_waterfall_rows(buf.post[:nf].reshape(-1), nf, n - 4, window, buf.glyph.shape[1], buf.glyph[:nf].reshape(-1),
                           buf.colour[:nf].reshape(-1), n_threads)
    return buf


def threads_available():
    """Host threads this process may actually run on: the affinity mask, capped by the cgroup's CPU quota (a container on a 256-thread box
    may own far fewer)."""
    import os
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = max(1, min(n, int(float(q) / float(per) + 0.5)))
    except Exception:  # noqa: BLE001
        pass
    return n


def headline_f64_buffers(nf, n, fs, disp_w=112, pcm=True, keep_db=False):
    q = int(fs / 22050)
    n_out = (n - 1 + q - 1) // q
    out = {"post": np.empty((nf, n - 4), np.float64), "lo": np.empty(nf, np.float64), "hi": np.empty(nf, np.float64),
           "glyph": np.empty((nf, disp_w), np.int8), "colour": np.empty((nf, disp_w), np.int8)}
    if pcm:
        out["pcm"] = np.empty((nf, n_out, 2), np.int16)
    if keep_db:
        out["db"] = np.empty((nf, n), np.float64)
    return out


def headline_f64(iq2d, fs, taps, sos, zi, window=30, disp_w=112, n_threads=1, pcm=True, keep_db=False, out=None, display="waterfall", disp_h=36):
    """The reference's own step from IQ in its own row type (float64 rows from compute_fft to the cells): per frame the post-processed
    row, its extremes, the display line with a history of `window` rows — waterfall (glyph, colour) or persistence (row index per column,
    in "glyph") —, the NFM int16 PCM.  -> dict (`out`: buffers of a previous call / headline_f64_buffers, reused)."""
    iq2d = np.ascontiguousarray(iq2d, np.complex64)
    nf, n = iq2d.shape
    q = int(fs / 22050)
    if out is None:
        out = headline_f64_buffers(nf, n, fs, disp_w, pcm, keep_db)
    pcm, keep_db = "pcm" in out, "db" in out
    L = lib()
    L.pss_o_batch_headline_f64(iq2d.view(np.float32).reshape(-1), nf, n, fs, q, np.ascontiguousarray(taps, np.float64),
                               np.ascontiguousarray(sos, np.float64), np.ascontiguousarray(zi, np.float64),
                               out["db"].ctypes.data if keep_db else None, out["post"].ctypes.data, out["lo"], out["hi"],
                               out["pcm"].ctypes.data if pcm else None, n_threads)
    if display == "persistence":
        L.pss_o_persistence_rows_f64(out["post"].reshape(-1), out["lo"], out["hi"], nf, n - 4, window, disp_h, disp_w,
                                     out["glyph"].reshape(-1), n_threads)
    else:
        L.pss_o_waterfall_rows_f64(out["post"].reshape(-1), out["lo"], out["hi"], nf, n - 4, window, disp_w, out["glyph"].reshape(-1),
                                   out["colour"].reshape(-1), n_threads)
    return out


def persistence_rows(rows, window, disp_h, disp_w, n_threads=1):
    """Newest persistence trace's row index per column for every frame (float32 post-processed rows [n_frames][len])."""
    rows = np.ascontiguousarray(rows, np.float32)
    y = np.empty((rows.shape[0], disp_w), np.int8)
    lib().pss_o_persistence_rows(rows.reshape(-1), rows.shape[0], rows.shape[1], window, disp_h, disp_w, y.reshape(-1), n_threads)
    return y


def waterfall_rows(rows, window, disp_w, n_threads=1):
    rows = np.ascontiguousarray(rows, np.float32)
    g = np.empty((rows.shape[0], disp_w), np.int8)
    c = np.empty((rows.shape[0], disp_w), np.int8)
    lib().pss_o_waterfall_rows(rows.reshape(-1), rows.shape[0], rows.shape[1], window, disp_w, g.reshape(-1), c.reshape(-1), n_threads)
    return g, c
