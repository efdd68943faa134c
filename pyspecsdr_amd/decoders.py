"""The reference's decoders (xqtr/PySpecSDR decoders.py) on libpss.so: same names, arguments and return values.

    decode_morse(samples, sample_rate, threshold=-20) -> (text, {"dot", "dash", "gap"})     decoders.py:136-231
    decode_aprs(samples, sample_rate)                 -> [packet] or []                       :115-133
    decode_afsk(samples, sample_rate)                 -> list of 0 / 1                        :94-112
    decode_ax25_frame(bit_stream)                     -> "SRC>DEST:info" or None             :6-63
    decode_aprs_payload(frame_bytes)                  -> "SRC>DEST:info" or None             :66-91

The sample-rate halves run on the GPU (envelope mask + edge compaction: pss_morse_edges; normalisation, the Bell-202 band-pass pair and
the per-bit energy compare: pss_afsk_bits), the per-message halves in the library's host code (pss_h_morse_decode, pss_h_ax25_frame).
decode_morse's two pulse classes come from a deterministic search where the reference draws random starting points for scipy's kmeans:
identical centroids for keyed signals (include/pss.h), a reproducible answer for noise.  Batched building blocks for callers that want
only the front halves: afsk_bits(), morse_edges().

The zero-edit launcher (python -m pyspecsdr_amd.run) leaves the application's own `import decoders` alone by default — its decoders.py
then runs its bookkeeping on top of the GPU band-pass (decoders.py:3 imports bandpass_filter from signal_processing); `--gpu-decoders`
registers this module under the name `decoders` as well.
"""
import ctypes as C

import numpy as np

from . import signal_processing as _sp


def _bandpass_tables(sample_rate):
    if _sp.USE_SCIPY_DESIGNS:
        try:
            import scipy.signal as ss
            nyq = sample_rate / 2
            return tuple(np.ascontiguousarray(ss.butter(_sp.BUTTER_ORDER, [lo / nyq, hi / nyq], btype="band", output="sos"))
                         for lo, hi in ((1100, 1300), (2100, 2300)))
        except ImportError:
            pass
    return (None, None)


def afsk_bits(samples, sample_rate, normalise=False):
    """-> uint8 array, one 0/1 per bit period of int(sample_rate / 1200) samples."""
    x = np.asarray(samples)
    if np.iscomplexobj(x):
        x = np.real(x)
    x = np.ascontiguousarray(x, dtype=np.float64)
    return _sp.get_engine().h_afsk_bits(x, sample_rate, *_bandpass_tables(sample_rate), normalise=normalise)


def morse_edges(samples, threshold=-20):
    """-> (rise_times, fall_times): int32 sample indices where the -20 dB envelope mask switches on / off."""
    return _sp.get_engine().h_morse_edges(_sp._samples(samples), threshold)


def decode_afsk(samples, sample_rate):
    return [int(b) for b in afsk_bits(samples, sample_rate)]


def _lib():
    return _sp.get_engine().lib


def morse_from_edges(rise_times, fall_times, sample_rate):
    """The back half of decode_morse on its own: (text, timing dict) from the rise / fall indices."""
    from . import _lib as L
    lib = L.load()
    rise = np.ascontiguousarray(rise_times, np.int32)
    fall = np.ascontiguousarray(fall_times, np.int32)
    cap = 4 * (len(rise) + 2) + 16
    text = C.create_string_buffer(cap)
    tm = np.zeros(3)
    r = lib.pss_h_morse_decode(rise.ctypes.data, len(rise), fall.ctypes.data, len(fall), float(sample_rate), text, cap, tm.ctypes.data)
    if r < 0:
        raise ValueError("operands could not be broadcast together: rise / fall edges do not alternate")
    n_gaps = _n_gaps(rise, fall)
    if n_gaps < 0:                     # no complete pulse: the reference's early returns
        return "", {"dot": 0, "dash": 0, "gap": 0}
    return text.raw[:r].decode("ascii"), {"dot": np.float64(tm[0]), "dash": np.float64(tm[1]), "gap": np.float64(tm[2]) if n_gaps > 0 else 0}


def _n_gaps(rise, fall):
    """Number of gaps between the complete pulses of an edge list (-1: no complete pulse)."""
    nr, nf = len(rise), len(fall)
    if nr == 0 or nf == 0:
        return -1
    if fall[0] < rise[0]:
        nf -= 1
    if nr > nf:
        nr -= 1
    return nr - 1


def decode_morse(samples, sample_rate, threshold=-20):
    rise, fall = morse_edges(samples, threshold)
    return morse_from_edges(rise, fall, sample_rate)


def decode_ax25_frame(bit_stream):
    from . import _lib as L
    lib = L.load()
    bits = np.ascontiguousarray(np.asarray(bit_stream, dtype=np.uint8))
    cap = len(bits) // 8 + 64
    out = C.create_string_buffer(cap)
    n = C.c_long(0)
    r = lib.pss_h_ax25_frame(bits.ctypes.data, len(bits), out, cap, C.byref(n))
    if r != 1:
        return None
    return out.raw[:n.value].decode("latin-1")


def decode_aprs_payload(frame_bytes):
    """The address / information fields of a list of frame bytes (the reference's helper; here a thin pure-Python statement of the format:
    it never sees sample data)."""
    fb = [int(b) & 0xFF for b in frame_bytes]
    if len(fb) < 14:
        return None
    dest = "".join(chr((b >> 1) & 0x7F) for b in fb[0:6]).strip()
    source = "".join(chr((b >> 1) & 0x7F) for b in fb[7:13]).strip()
    info = "".join(chr(b) for b in fb[15:]) if len(fb) > 15 else ""
    return f"{source}>{dest}:{info}"


def decode_aprs(samples, sample_rate):
    x = np.asarray(samples)
    if np.iscomplexobj(x):
        x = np.real(x)
    x = np.ascontiguousarray(x, dtype=np.float64)
    eng = _sp.get_engine()
    s1, s2 = _bandpass_tables(sample_rate)
    cap = len(x) // 8 + 64
    out = C.create_string_buffer(cap)
    n = C.c_long(0)
    r = eng.lib.pss_h_decode_aprs(eng.h, x.ctypes.data, len(x), float(sample_rate), None if s1 is None else s1.ctypes.data,
                                  None if s2 is None else s2.ctypes.data, 0 if s1 is None else s1.shape[0], out, cap, C.byref(n))
    if r < 0:
        eng._ck(r)
    return [out.raw[:n.value].decode("latin-1")] if r == 1 else []
