#!/usr/bin/env python3
"""Second fuzz set (build container only): SSB int16, compute_fft / post-process tolerances, scanner slice, bandpass_filter,
decode_afsk, and the display quantisers through a fake curses screen — oracle vs the reference on random inputs."""
import os, sys, types, warnings
sys.path.insert(0, '/root/reference'); sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np, scipy.signal as ss
import curses
sd = types.ModuleType("sounddevice"); sd.PortAudioError = type("PortAudioError", (Exception,), {}); sd.OutputStream = object
so = types.ModuleType("SoapySDR"); so.SOAPY_SDR_RX = 1; so.SOAPY_SDR_CF32 = "CF32"; so.Device = object
sys.modules["sounddevice"] = sd; sys.modules["SoapySDR"] = so
curses.color_pair = lambda n: n << 8
import signal as _signal
old = (_signal.getsignal(_signal.SIGINT), _signal.getsignal(_signal.SIGTERM))
import signal_processing as sp, decoders, pyspecsdr as P
_signal.signal(_signal.SIGINT, old[0]); _signal.signal(_signal.SIGTERM, old[1])
import oracle_lib as O
warnings.simplefilter('ignore')
rng = np.random.default_rng(4242)

class Scr:
    def __init__(s, h, w): s.h, s.w, s.calls = h, w, []
    def getmaxyx(s): return s.h, s.w
    def addstr(s, *a): s.calls.append(a)
    def refresh(s): pass

def rnd_iq(n):
    kind = rng.integers(0, 3)
    if kind == 0:
        x = rng.uniform(0.05, 1.5) * np.exp(1j * np.cumsum(rng.standard_normal(n) * rng.uniform(0.01, 0.5)))
    elif kind == 1:
        x = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    else:
        t = np.arange(n); x = 0.4 * np.exp(2j * np.pi * rng.uniform(-0.4, 0.4) * t)
    return (x + rng.uniform(0.001, 0.05) * (rng.standard_normal(n) + 1j * rng.standard_normal(n))).astype(np.complex64)

bad = 0
cnt = dict(ssb=0, fft=0, scan=0, bp=0, afsk=0, sg=0, wf=0, gw=0, sf=0, vec=0, ps=0)
scan_bits = scan_vals = 0
for it in range(150):
    n = int(rng.choice([64, 300, 1024, 2048, 4096, 16384]))
    fs = float(rng.choice([2.4e6, 1.024e6, 250e3]))
    x = rnd_iq(n)
    # SSB: int16 exact, float64 within 2e-14
    taps = ss.firwin(65, 3000 / fs, window='hamming')
    ref = sp.demodulate_signal(x, fs, 'USB')[:, 0]; got = O.demod_ssb(x, taps); cnt['ssb'] += 1
    # frames of 2^k samples: every bit of the float64 audio (the oracle replays SciPy's hilbert()); others: int16 exact, float64 within 2e-14
    if not (np.array_equal(np.int16(ref * 32767), np.int16(got * 32767)) and (np.array_equal(ref, got) if n & (n - 1) == 0 else np.max(np.abs(ref - got)) < 2e-14)):
        bad += 1; print('SSB mismatch', n, fs, np.max(np.abs(ref - got)))
    # compute_fft (power of two only) and the caller's post-process
    if n & (n - 1) == 0:
        ref = sp.compute_fft(x); got = O.compute_fft(x); cnt['fft'] += 1
        if not np.all(np.abs(got - ref) <= 1e-9 * np.maximum(np.abs(ref), 1.0)):
            bad += 1; print('FFT mismatch', n, np.max(np.abs(got - ref)))
        fd = np.convolve(ref, np.ones(5) / 5, mode='valid'); thr = np.median(fd) - 10; fd[fd < thr] = thr
        gp = O.postprocess(ref)
        if not np.allclose(gp, fd, rtol=1e-12, atol=1e-12): bad += 1; print('POST mismatch', n)
        # scanner slice formulas (pyspecsdr.py:2542-2552)
        spec = np.fft.fftshift(np.fft.fft(x)); pdb = 10 * np.log10(np.abs(spec) ** 2 + 1e-10); pk = np.max(pdb)
        mask = pdb > pk - 20; bw = np.sum(mask) * (fs / len(pdb))
        db, opk, obw, ocnt = O.scan_slice(x, fs); cnt['scan'] += 1
        nd = int((db.view(np.uint32) != pdb.astype(np.float32).view(np.uint32)).sum())   # rows: every bit (NaNs aside)
        scan_bits += nd; scan_vals += len(db)
        if not (pdb.dtype == np.float32 and nd == 0 and np.float32(opk).tobytes() == np.float32(pk).tobytes() and ocnt == int(np.sum(mask)) and obw == bw):
            bad += 1; print('SCAN mismatch', n, pk, opk, np.sum(mask), ocnt, nd)
    # bandpass_filter + decode_afsk on audio-rate rows
    afs = float(rng.choice([22050.0, 48000.0]))
    m = int(rng.integers(50, 6000)); a = rng.standard_normal(m); a = a / np.max(np.abs(a))
    lo, hi = (0, 3000) if it % 3 == 0 else (300, 3000)
    nyq = afs / 2
    sos = ss.butter(5, hi / nyq, btype='low', output='sos') if lo <= 0 else ss.butter(5, [lo / nyq, hi / nyq], btype='band', output='sos')
    ref = sp.bandpass_filter(a, lo, hi, afs); got = O.sosfilt(sos, a); cnt['bp'] += 1
    if not np.array_equal(ref, got): bad += 1; print('BP mismatch', m)
    bits = np.array(decoders.decode_afsk(a, afs), np.uint8)
    s1 = ss.butter(5, [1100 / nyq, 1300 / nyq], btype='band', output='sos'); s2 = ss.butter(5, [2100 / nyq, 2300 / nyq], btype='band', output='sos')
    ob = O.afsk_bits(a, afs, s1, s2); cnt['afsk'] += 1
    if not np.array_equal(bits, ob): bad += 1; print('AFSK mismatch', m, afs)
    # display quantisers on random post-processed rows
    L = int(rng.choice([252, 1020, 4092])); H = int(rng.integers(20, 60)); W = int(rng.integers(60, 200))
    rows = rng.standard_normal((12, L)) * rng.uniform(1, 8) - rng.uniform(10, 60)
    rows[:, L // 3:L // 3 + 20] += rng.uniform(5, 40)
    g5 = {".": 0, "-": 1, "=": 2, "#": 3, " ": 4}
    scr = Scr(H, W); P.draw_spectrogram(scr, rows[0].copy(), None, 100e6, 2.4e6, 0, 0, None)
    dh, dw = H - 4, W - 7
    g = -np.ones((dh, dw), np.int8); c = -np.ones((dh, dw), np.int8)
    for call in scr.calls:
        if len(call) != 4: continue
        y, xx, st, attr = call
        if len(st) == 1 and st in g5 and xx >= 7 and 2 <= y < 2 + dh and xx - 7 < dw:
            g[y - 2, xx - 7] = g5[st]; c[y - 2, xx - 7] = (attr >> 8) & 0xFF
    og, oc, _, _ = O.spectrogram_cells(rows[0], dh, dw); cnt['sg'] += 1
    if not (np.array_equal(g, og) and np.array_equal(c, oc)):
        bad += 1; print('SPECTROGRAM mismatch', L, H, W, int(np.sum(g != og)), int(np.sum(c != oc)))
    P.WATERFALL_HISTORY.clear(); gl = {".": 0, "-": 1, "=": 2, "#": 3}
    for r in rows: 
        scr = Scr(H, W); P.draw_waterfall(scr, r, None, 100e6, 2.4e6, 0, 0, None)
    g = -np.ones((H - 4, W - 8), np.int8); c = -np.ones((H - 4, W - 8), np.int8)
    for call in scr.calls:
        y, xx, st, attr = call
        if st in gl and xx >= 9 and y >= 3 and len(st) == 1 and (attr >> 8) >= 10 and y - 3 < H - 4 and xx - 9 < W - 8:
            g[y - 3, xx - 9] = gl[st]; c[y - 3, xx - 9] = (attr >> 8) - 10
    og, oc = O.waterfall_cells(rows, H - 4, W - 8); cnt['wf'] += 1
    if not (np.array_equal(g, og) and np.array_equal(c, oc)): bad += 1; print('WATERFALL mismatch', L, H, W, int(np.sum(g != og)))
    P.WATERFALL_HISTORY.clear(); ch9 = ' ._-=+*#@'
    for r in rows:
        scr = Scr(H, W); P.draw_gradient_waterfall(scr, r, None, 100e6, 2.4e6, 0, 0, None)
    g = -np.ones((H - 4, W - 10), np.int8); c = -np.ones((H - 4, W - 10), np.int8)
    for call in scr.calls:
        y, xx, st, attr = call
        if len(st) == 1 and st in ch9 and 9 <= xx < 9 + (W - 10) and 2 <= y < 2 + (H - 4) and (attr >> 8) >= 10:
            g[y - 2, xx - 9] = ch9.index(st); c[y - 2, xx - 9] = (attr >> 8) - 10
    og, oc = O.gradient_cells(rows, H - 4, W - 10); cnt['gw'] += 1
    if not (np.array_equal(g, og) and np.array_equal(c, oc)): bad += 1; print('GRADIENT mismatch', L, H, W, int(np.sum(g != og)))
    P.WATERFALL_HISTORY.clear(); P.PERSISTENCE_HISTORY.clear()
    for r in rows[:10]:
        scr = Scr(H, W); P.draw_persistence(scr, r, None, 100e6, 2.4e6, 0, 0, None)
    g = np.zeros((H - 4, W - 8), np.int8)
    for call in scr.calls:
        y, xx, st, attr = call
        if st == "*" and 0 <= y - 2 < H - 4 and 0 <= xx - 8 < W - 8: g[y - 2, xx - 8] = attr >> 8
    og = O.persistence_cells(rows[:10], H - 4, W - 8); cnt['ps'] += 1
    if not np.array_equal(g, og): bad += 1; print('PERSISTENCE mismatch', L, H, W, int(np.sum(g != og)))
    P.PERSISTENCE_HISTORY.clear()
    scr = Scr(H, W); P.draw_surface_plot(scr, rows[1].copy(), None, 100e6, 2.4e6, 0, 0, None)
    g = np.zeros((H, W), np.int8)
    for call in scr.calls:
        y, xx, st, attr = call
        if st == "#": g[y, xx] = attr >> 8
    og = O.surface_cells(rows[1], H, W); cnt['sf'] += 1
    if not np.array_equal(g, og): bad += 1; print('SURFACE mismatch', L, H, W, int(np.sum(g != og)))
    vs = (rnd_iq(500) * np.float32(rng.uniform(0.5, 3))).astype(np.complex64)
    scr = Scr(H, W); P.draw_vector_display(scr, vs, 100e6, 2.4e6, 0, 0, None)
    g = np.zeros((H, W), np.int8)
    for call in scr.calls:
        if len(call) == 4 and call[2] == ".": g[call[0], call[1]] = 1
    og = O.vector_cells(vs, H, W); cnt['vec'] += 1
    if not np.array_equal(g, og): bad += 1; print('VECTOR mismatch', H, W, int(np.sum(g != og)))
print('cases', cnt, 'bad', bad)
print('scanner rows: %d of %d float32 dB values differ' % (scan_bits, scan_vals))
