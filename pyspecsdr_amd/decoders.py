"""Drop-in for the reference's `decoders.py` (xqtr/PySpecSDR): same callables, arguments and return values.

The sample-rate work runs on the GPU through libpss.so:
    decode_afsk(samples, sample_rate)        decoders.py:94-112   Bell-202 band-pass pair + per-bit energy compare (pss_afsk_bits)
    decode_morse(samples, sample_rate, thr)  decoders.py:136-231  envelope, -20 dB mask, rise / fall indices (pss_morse_edges)
What is left on the host is the reference's own per-message bookkeeping on a few dozen numbers: AX.25 flag search and bit
de-stuffing (decode_ax25_frame :6-64, decode_aprs_payload :67-91, decode_aprs :115-133) and the dot/dash timing logic of
decode_morse (:167-231), which — like the reference — asks scipy.cluster.vq.kmeans for the two duration centroids.
"""
import numpy as np

from . import signal_processing as _sp

_FLAG = [0, 1, 1, 1, 1, 1, 1, 0]

# International Morse code plus the handful of procedural signs the reference's table (pyspecconst.py:345-358) carries
_MORSE = dict(zip(
    ".- -... -.-. -.. . ..-. --. .... .. .--- -.- .-.. -- -. --- .--. --.- .-. ... - ..- ...- .-- -..- -.-- --..".split(),
    "ABCDEFGHIJKLMNOPQRSTUVWXYZ"))
_MORSE.update(zip(".---- ..--- ...-- ....- ..... -.... --... ---.. ----. -----".split(), "1234567890"))
_MORSE.update({"--..--": ",", ".-.-.-": ".", "..--..": "?", "-..-.": "/", "-....-": "-", "-.--.": "(", "-.--.-": ")",
               ".-...": "&", "---...": ":", "-.-.-.": ";", "-...-": "=", ".-.-.": "+", ".-..-.": '"', "...-..-": "$",
               ".--.-.": "@", "..--.-": "_", "...---...": "SOS"})


def decode_afsk(samples, sample_rate):
    """decoders.py:94-112 -> list of 0/1 ints, one per bit period of int(sample_rate / 1200) samples."""
    x = np.ascontiguousarray(np.asarray(samples), dtype=np.float64)
    sos = (None, None)
    if _sp.USE_SCIPY_DESIGNS:
        try:
            import scipy.signal as ss
            nyq = sample_rate / 2
            sos = tuple(ss.butter(_sp.BUTTER_ORDER, [lo / nyq, hi / nyq], btype="band", output="sos")
                        for lo, hi in ((1100, 1300), (2100, 2300)))
        except ImportError:
            pass
    return [int(b) for b in _sp.get_engine().h_afsk_bits(x, sample_rate, *sos)]


def decode_ax25_frame(bit_stream):
    """decoders.py:6-64: first 01111110 flag, de-stuffed bits up to the next flag, LSB-first bytes -> decode_aprs_payload."""
    try:
        nb = len(bit_stream)
        start = next((i + 8 for i in range(nb - 7) if bit_stream[i:i + 8] == _FLAG), -1)
        if start == -1:
            return None
        bits, ones, i = [], 0, start
        while i < nb - 7:
            b = bit_stream[i]
            bits.append(b)
            ones = ones + 1 if b == 1 else 0
            if ones == 5 and i + 1 < nb and bit_stream[i + 1] == 0:   # a stuffed zero follows: drop it
                i += 2
                ones = 0
                continue
            i += 1
            if bits[-8:] == _FLAG:
                bits = bits[:-8]
                break
        octets = [sum(bits[k + j] << j for j in range(8)) for k in range(0, len(bits) - 7, 8)]
        return decode_aprs_payload(octets)
    except Exception:
        return None


def decode_aprs_payload(frame_bytes):
    """decoders.py:67-91 -> 'SOURCE>DEST:info' or None."""
    try:
        if len(frame_bytes) < 14:
            return None
        call = lambda bs: "".join(chr((b >> 1) & 0x7F) for b in bs).strip()
        dest, source = call(frame_bytes[0:6]), call(frame_bytes[7:13])
        info = "".join(chr(b) for b in frame_bytes[15:]) if len(frame_bytes) > 15 else ""
        return f"{source}>{dest}:{info}"
    except Exception:
        return None


def decode_aprs(samples, sample_rate):
    """decoders.py:115-133 -> [packet] or []."""
    if np.iscomplexobj(samples):
        samples = np.real(samples)
    samples = samples / np.max(np.abs(samples))
    packet = decode_ax25_frame(decode_afsk(samples, sample_rate))
    return [packet] if packet else []


def _two_centroids(durations):
    try:
        from scipy.cluster import vq
        centroids, _ = vq.kmeans(durations.reshape(-1, 1), 2)      # what the reference calls (decoders.py:182-183)
        return centroids
    except ImportError:                                            # no SciPy on this host: plain 1-D Lloyd from the extremes
        c = np.array([durations.min(), durations.max()])
        for _ in range(20):
            near = np.abs(durations[:, None] - c[None, :]).argmin(axis=1)
            c = np.array([durations[near == k].mean() if np.any(near == k) else c[k] for k in range(2)])
        return c


def decode_morse(samples, sample_rate, threshold=-20):
    """decoders.py:136-231 -> (decoded_text, {"dot", "dash", "gap"})."""
    rise_times, fall_times = _sp.get_engine().h_morse_edges(_sp._samples(samples), threshold)
    rise_times, fall_times = rise_times.astype(np.int64), fall_times.astype(np.int64)
    if len(rise_times) == 0 or len(fall_times) == 0:
        return "", {"dot": 0, "dash": 0, "gap": 0}
    if fall_times[0] < rise_times[0]:
        fall_times = fall_times[1:]
    if len(rise_times) > len(fall_times):
        rise_times = rise_times[:-1]
    durations = (fall_times - rise_times) / sample_rate
    gaps = (rise_times[1:] - fall_times[:-1]) / sample_rate
    if len(durations) == 0:
        return "", {"dot": 0, "dash": 0, "gap": 0}
    if len(durations) > 1:
        centroids = _two_centroids(durations)
        dot_duration, dash_duration = np.min(centroids), np.max(centroids)
    else:
        dot_duration = np.min(durations)
        dash_duration = dot_duration * 3
    symbols, letter = [], []
    for i, d in enumerate(durations):
        letter.append("." if d < (dot_duration + dash_duration) / 2 else "-")
        if i < len(gaps) and gaps[i] > dot_duration * 3:            # letter gap; a word gap on top of it beyond 7 dots
            symbols.append("".join(letter))
            letter = []
            if gaps[i] > dot_duration * 7:
                symbols.append(" ")
    if letter:
        symbols.append("".join(letter))
    text = "".join(" " if s == " " else _MORSE.get(s, "?") for s in symbols)
    return text, {"dot": dot_duration, "dash": dash_duration, "gap": np.mean(gaps) if len(gaps) > 0 else 0}
