import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
import gpu_util as G
from pyspecsdr_amd import _lib as L
e = G.engine()
rng = np.random.default_rng(1)
for nf, n in ((1, 32768), (16, 32768), (64, 8192)):
    iq = (rng.standard_normal((nf, n)) + 1j * rng.standard_normal((nf, n))).astype(np.complex64)
    fs = 250e3
    d = G.dev(iq); n_out = e.demod_out_len(L.MODE_WFM, n, fs)
    pcm = G.empty((nf, n_out, 2), torch.int16)
    for _ in range(3): e.demod(L.MODE_WFM, d, nf, n, fs, pcm, None)
    e.sync(); e.enable_timing(True)
    for _ in range(5): e.demod(L.MODE_WFM, d, nf, n, fs, pcm, None)
    e.sync()
    kt = e.kernel_times(); e.enable_timing(False)
    print(nf, n, {k: [round(x, 3) for x in v[-4:]] for k, v in kt.items()})
