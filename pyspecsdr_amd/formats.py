"""The data formats either side of the hot path, so that real captures can replace synthetic IQ:

  * IQ recordings     — `record_signal` / `play_recorded_signal` (pyspecsdr.py:814-824): `np.save` of the 1-D complex64 read
                        buffer (SoapySDR CF32, interleaved I/Q float32 — pyspecsdr.py:1870-1891);
  * audio recordings  — `start_audio_recording` / `write_audio_samples` (audio_processing.py:25-43): RIFF/WAVE, 2 channels,
                        16-bit, `np.int16(samples * 32767)`;
  * the audio FIFO    — `/tmp/sdrpipe` (io_manager.py:1-37): the same int16 frames, raw s16le `L R L R ...`.

Nothing here computes: the int16 conversion happens on the GPU (`pss_demod*`), these helpers move bytes.
"""
import wave

import numpy as np

from . import _lib as L
from .signal_processing import DEFAULT_SAMPLE_RATE, _inject_designs, get_engine

_MODES = {'NFM': L.MODE_NFM, 'AM': L.MODE_AM, 'USB': L.MODE_USB, 'LSB': L.MODE_LSB, 'WFM': L.MODE_WFM}


def load_iq_recording(path):
    """play_recorded_signal (pyspecsdr.py:821-824): the saved read buffer as 1-D complex64."""
    s = np.load(path)
    if s.ndim != 1 or not np.iscomplexobj(s):
        raise ValueError("not an IQ recording: expected a 1-D complex array")
    return np.ascontiguousarray(s, np.complex64)


def cut_frames(samples, frame_len):
    """The recording as the read buffers the main loop would have seen (pyspecsdr.py:2236): [n_frames][frame_len];
    an incomplete tail buffer is dropped."""
    nf = len(samples) // frame_len
    return samples[:nf * frame_len].reshape(nf, frame_len)


def write_wav(path, pcm, sample_rate=DEFAULT_SAMPLE_RATE):
    """start_audio_recording + write_audio_samples + stop_audio_recording (audio_processing.py:25-43) for int16 frames
    that are already converted: pcm int16 [..., 2]."""
    pcm = np.ascontiguousarray(pcm, np.int16).reshape(-1, 2)
    with wave.open(path, 'wb') as w:
        w.setnchannels(2)
        w.setsampwidth(2)
        w.setframerate(sample_rate)
        w.writeframes(pcm.tobytes())


def pipe_bytes(pcm):
    """What write_to_pipe (io_manager.py:23-27) hands to the FIFO for these frames."""
    return np.ascontiguousarray(pcm, np.int16).tobytes()


def demodulate_recording(samples, sample_rate, mode='NFM', frame_len=32768, chunk_frames=4096):
    """Every read buffer of a recording through demodulate_signal on the GPU -> int16 [n_frames][n_out][2], i.e. the
    audio the reference would have written had it played the recording buffer by buffer."""
    frames = cut_frames(np.ascontiguousarray(samples, np.complex64), frame_len)
    fs = float(DEFAULT_SAMPLE_RATE) if mode == 'AM' else float(sample_rate)
    if mode != 'AM':
        _inject_designs({'NFM': 'nfm', 'WFM': 'wfm'}.get(mode, 'ssb'), fs)
    return get_engine().h_demodulate_batch(_MODES[mode], frames, fs, chunk_frames)


def recording_to_wav(npy_path, wav_path, sample_rate, mode='NFM', frame_len=32768):
    pcm = demodulate_recording(load_iq_recording(npy_path), sample_rate, mode, frame_len)
    write_wav(wav_path, pcm)
    return pcm
