"""GPU front halves of the reference's decoders (xqtr/PySpecSDR decoders.py) — the sample-rate work only.

    afsk_bits(samples, sample_rate, normalise=False)   the bit stream of decode_afsk (decoders.py:94-112): Bell-202 band-pass
                                                       pair + per-bit energy compare (pss_afsk_bits); normalise=True divides
                                                       by max|x| on the device first, as decode_aprs does (decoders.py:126)
    morse_edges(samples, threshold=-20)                rise / fall sample indices of decode_morse's envelope mask
                                                       (decoders.py:149-161) (pss_morse_edges)

This module is NOT a replacement for `decoders`: the per-message bookkeeping on the resulting few dozen numbers (AX.25 flag
search / bit de-stuffing, Morse timing and table lookup) is the reference's own code and stays there.  Under the zero-edit
launcher (python -m pyspecsdr_amd.run) the reference's decoders.py imports `bandpass_filter` from `signal_processing`
(decoders.py:3), which resolves to the GPU band-pass, so its decode_afsk already filters on the GPU; the two functions
here are the batched building blocks for callers that want the whole front half in one call.
"""
import numpy as np

from . import signal_processing as _sp


def afsk_bits(samples, sample_rate, normalise=False):
    """-> uint8 array, one 0/1 per bit period of int(sample_rate / 1200) samples."""
    x = np.asarray(samples)
    if np.iscomplexobj(x):
        x = np.real(x)
    x = np.ascontiguousarray(x, dtype=np.float64)
    sos = (None, None)
    if _sp.USE_SCIPY_DESIGNS:
        try:
            import scipy.signal as ss
            nyq = sample_rate / 2
            sos = tuple(ss.butter(_sp.BUTTER_ORDER, [lo / nyq, hi / nyq], btype="band", output="sos")
                        for lo, hi in ((1100, 1300), (2100, 2300)))
        except ImportError:
            pass
    return _sp.get_engine().h_afsk_bits(x, sample_rate, *sos, normalise=normalise)


def morse_edges(samples, threshold=-20):
    """-> (rise_times, fall_times): int32 sample indices where the -20 dB envelope mask switches on / off."""
    return _sp.get_engine().h_morse_edges(_sp._samples(samples), threshold)
