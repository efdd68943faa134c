"""Helpers for the -m gpu parity tests: device buffers via torch (plumbing only), engine fixture."""
import numpy as np
import torch

from pyspecsdr_amd import _lib as L
from pyspecsdr_amd.engine import Engine

_ENG = None


def engine():
    global _ENG
    if _ENG is None:
        assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
        _ENG = Engine(0)
    return _ENG


def dev(a):
    """numpy -> device tensor (complex64 is shipped as interleaved float32)."""
    a = np.ascontiguousarray(a)
    if a.dtype == np.complex64:
        a = a.view(np.float32)
    return torch.from_numpy(a).cuda()


def empty(shape, dtype):
    return torch.empty(shape, dtype=dtype, device="cuda")


def host(t):
    return t.cpu().numpy()


def spectrum(iq2d):
    e = engine()
    nf, n = iq2d.shape
    d_iq, d_db = dev(iq2d), empty((nf, n), torch.float32)
    e.spectrum_db(d_iq, nf, n, d_db)
    e.sync()
    return host(d_db)


def demod(mode, iq2d, fs, want_audio=True):
    e = engine()
    nf, n = iq2d.shape
    n_out = e.demod_out_len(mode, n, fs)
    d_iq = dev(iq2d)
    d_pcm = empty((nf, n_out, 2), torch.int16)
    d_au = empty((nf, n_out), torch.float64) if want_audio else None
    e.demod(mode, d_iq, nf, n, fs, d_pcm, d_au)
    e.sync()
    return host(d_pcm), (host(d_au) if want_audio else None)


def has_option(key, default):
    """True if this build of the library knows the (kernel-selection, -DPSS_VARIANTS) option `key`; sets it to `default`."""
    from pyspecsdr_amd.engine import PssError
    try:
        engine().set_option(key, default)
        return True
    except PssError:
        return False
