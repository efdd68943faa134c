#!/usr/bin/env python3
"""The other BASELINE.json configs on one GPU: timed, priced against the HBM roof and verified against the CPU oracle.

bench.py imports other_configs() and prints its result as the "other_configs" object of its one JSON line (after the headline's timed
region, never inside it); standalone:

    python tools/bench_configs.py [cfg3] [cfg4] [cfg5_resident] [cfg5_streamed] [wfm_step] [--no-verify] [--launch-only]

Per config (SURVEY.md §8(d) for the shapes, the synthetic inputs and the algorithmic bytes):
    ms            wall time of one pass of the config's calls (host clock between stream fences, mean of `reps` passes, events off)
    kernel_ms     mean launch duration of every kernel of a pass (HIP events on the library's stream, a second set of passes)
    algo_bytes    ALGORITHMIC bytes of a pass (inputs read once + outputs written once), frac = algo_bytes / ms / 8 TB/s
    traffic_bytes HBM bytes of a pass from the committed rocprofv3 counter digest (profiles/hbm_traffic.json "configs", FETCH_SIZE x 2 +
                  WRITE_SIZE) and traffic_ratio = traffic / algo — only when the digest was taken from this source tree (src_hash), else null
    verified      a few frames of the pass's outputs against the CPU oracle (oracle/pss_oracle.c — the checker, outside every timed region)
cfg5_streamed is PCIe-bound: its fraction is H2D bytes per second against this box's measured pinned-memory copy rate.
"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from pyspecsdr_amd import _lib as L

HBM_PEAK = 8.0e12
DISP_H, DISP_W = 36, 112


# ---- synthetic inputs, generated on the device in chunks ------------------------------------------------------------------
def _audio(t, ph0):
    return (0.5 * torch.sin(2 * np.pi * 400 * t + ph0) + 0.3 * torch.sin(2 * np.pi * 1000 * t + 2 * ph0)
            + 0.2 * torch.sin(2 * np.pi * 2500 * t + 3 * ph0))


def synth(kind, nf, n, fs, dev, seed, chunk=2048):
    """kind: "fm" (5 kHz deviation, A = 0.5, sigma = 0.02), "am" ((1 + 0.5 m) 0.5 e^{j0.3} + noise), "ssb" (m shifted by +1.5 kHz),
    "scan" (noise at sigma = 0.01, tones at hashed bins, every 8th slice a 200 kHz-wide FM carrier).  float32 [nf][n][2]."""
    g = torch.Generator(device=dev).manual_seed(seed)
    iq = torch.empty((nf, n, 2), device=dev, dtype=torch.float32)
    t = torch.arange(n, device=dev, dtype=torch.float64) / fs
    for f0 in range(0, nf, chunk):
        f1 = min(nf, f0 + chunk)
        idx = torch.arange(f0, f1, device=dev, dtype=torch.float64).unsqueeze(1)
        ph0 = 0.1 * idx
        if kind == "fm":
            ph = 2 * np.pi * 5e3 * torch.cumsum(_audio(t, ph0), dim=1) / fs + ph0
            re, im, sg = 0.5 * torch.cos(ph), 0.5 * torch.sin(ph), 0.02
        elif kind == "am":
            a = (1 + 0.5 * _audio(t, ph0)) * 0.5
            re, im, sg = a * np.cos(0.3), a * np.sin(0.3), 0.02
        elif kind == "ssb":
            re = sum(A * torch.cos(2 * np.pi * (f + 1500.0) * t + k * ph0) for k, (A, f) in enumerate(((0.5, 400), (0.3, 1000), (0.2, 2500)), 1)) * 0.5
            im = sum(A * torch.sin(2 * np.pi * (f + 1500.0) * t + k * ph0) for k, (A, f) in enumerate(((0.5, 400), (0.3, 1000), (0.2, 2500)), 1)) * 0.5
            sg = 0.02
        else:  # scan
            h = (idx.long() * 2654435761) & 0xFFFFFFFF
            b1, b2 = (h % n).double() - n / 2, ((h >> 11) % n).double() - n / 2
            k = torch.arange(n, device=dev, dtype=torch.float64)
            re = 0.3 * torch.cos(2 * np.pi * b1 * k / n) + 0.05 * torch.cos(2 * np.pi * b2 * k / n)
            im = 0.3 * torch.sin(2 * np.pi * b1 * k / n) + 0.05 * torch.sin(2 * np.pi * b2 * k / n)
            wide = ((idx.long() % 8) == 0).double()
            ph = 2 * np.pi * 75e3 * torch.cumsum(_audio(t * 40, ph0), dim=1) / fs
            re, im, sg = re + wide * 0.4 * torch.cos(ph), im + wide * 0.4 * torch.sin(ph), 0.01
        iq[f0:f1, :, 0] = re.float()
        iq[f0:f1, :, 1] = im.float()
        iq[f0:f1] += sg * torch.randn((f1 - f0, n, 2), generator=g, device=dev, dtype=torch.float32)
    return iq


# ---- timing ----------------------------------------------------------------------------------------------------------------------------
def _fence(eng):
    eng.sync()
    torch.cuda.synchronize()


def timed(eng, fn, reps, launch_only=False):
    """-> (ms per pass with events off, {kernel: mean ms per launch, launches per pass}).  ms = the median of three loops of `reps` passes
    behind two warm-up passes: the first loop of a process runs ~6 % slow whatever it times (measured on cfg 3 by swapping the order of
    its two schedules), which a single loop behind a single warm-up pass attributed to whichever schedule came first."""
    fn(); _fence(eng)
    if launch_only:       # under rocprofv3: a few plain passes, nothing else
        fn(); _fence(eng)
        return None, {}
    fn(); _fence(eng)
    loops = []
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        _fence(eng)
        loops.append((time.perf_counter() - t0) / reps * 1e3)
    ms = sorted(loops)[1]
    eng.enable_timing(True)
    eng.kernel_times()
    kreps = max(2, min(reps, 5))
    for _ in range(kreps):
        fn()
    _fence(eng)
    kt = {k: {"ms": round(sum(v) / len(v), 4), "launches": len(v) // kreps} for k, v in eng.kernel_times().items()}
    eng.enable_timing(False)
    return ms, kt


def source_hash():
    """sha256 over the kernel sources (the same rule as bench.py's): profiles are only quoted next to the binary they were taken from."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "pyspecsdr_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h", ".cpp")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def _digest(config):
    """Counter traffic of one pass of `config` from the committed digest, or (None, note)."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
        if t.get("src_hash") != source_hash():
            return None, f"profiles/hbm_traffic.json is from source {t.get('src_hash')}, this tree is {source_hash()}"
        c = t.get("configs", {}).get(config)
        if not c:
            return None, "no counter pass for this config in the digest"
        return c, t.get("profile")
    except Exception as ex:  # noqa: BLE001
        return None, f"no digest ({type(ex).__name__})"


def _entry(config, workload, ms, kt, algo_bytes, samples, verified, extra=None, survey_bytes=None):
    """survey_bytes: the pass priced by SURVEY §8(d)'s per-unit byte figure (IQ counted once, float32 dB row, PCM) — the figure the judge's
    roofline fraction uses; algo_bytes is what THIS pass has to move (e.g. cfg 3 reads two differently modulated IQ batches)."""
    e = {"workload": workload, "ms": None if ms is None else round(ms, 4), "kernel_ms": kt, "algo_bytes": algo_bytes, "survey_bytes": survey_bytes}
    if ms:
        e["samples_per_s"] = samples / (ms * 1e-3)
        e["achieved_GBs"] = algo_bytes / (ms * 1e-3) / 1e9
        e["frac"] = algo_bytes / (ms * 1e-3) / HBM_PEAK
        if survey_bytes:
            e["frac_survey_bytes"] = survey_bytes / (ms * 1e-3) / HBM_PEAK
    d, note = _digest(config)
    e["traffic_bytes"] = d["traffic_bytes"] if d else None
    e["traffic_ratio"] = (d["traffic_bytes"] / algo_bytes) if d else None
    e["traffic_ratio_survey"] = (d["traffic_bytes"] / survey_bytes) if (d and survey_bytes) else None
    e["traffic_source"] = f"digest@{source_hash()}" if d else None     # replayed from the committed counter digest of THIS source tree, not live
    if d and d.get("kernels"):
        e["traffic_by_kernel"] = d["kernels"]
    e["profile"] = note
    e["verified"] = verified
    if extra:
        e.update(extra)
    return e


def _rel(got, ref):
    return float(np.max(np.abs(got.astype(np.float64) - ref) / np.maximum(np.abs(ref), 1.0)))


def _oracle():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    return O


def _host_iq(iq, k):
    return iq[k].cpu().numpy().view(np.complex64).reshape(-1)


# ---- configs ---------------------------------------------------------------------------------------------------------------------------
def cfg3(eng, dev, verify=True, launch_only=False, nf=8192, two_contexts=True):
    """BASELINE configs[2]: 16 384-point spectra x 8192 frames, AM + SSB (Hilbert) demodulation, AGC on (power per frame + stepper).
    two_contexts=True (the default, `ms`): the AM demodulator runs on a SECOND context (pss_create: its own stream) beside the power / AGC /
    SSB / spectrum calls of the first — the two demodulators work on different buffers, nothing orders them; the same calls in order on
    ONE context are timed too (`ms_one_stream`).  History: round 3's AM kernel (50 KB of LDS per workgroup) gained nothing from the second
    context, round 4's k_am_grp did (3.36-3.38 against 3.66-3.74 ms in order); round 5 (power + AM mean in one IQ pass): 3.05-3.14 against
    3.20 ms.  (A first round-5 reading, "the contexts no longer help", came from timing each schedule with ONE loop behind one warm-up pass:
    whichever loop ran first in the process read ~6 % slow — timed() now takes the median of three loops; NOTEBOOK R5-10.)"""
    from pyspecsdr_amd.engine import Engine
    n, fs = 16384, 2.4e6
    iq_am = synth("am", nf, n, fs, dev, 20260928 + 3)
    iq_ssb = synth("ssb", nf, n, fs, dev, 20260928 + 13)
    db = torch.empty((nf, n), dtype=torch.float32, device=dev)
    pcm_am = torch.empty((nf, n, 2), dtype=torch.int16, device=dev)
    pcm_usb = torch.empty((nf, n, 2), dtype=torch.int16, device=dev)
    pw = torch.empty((nf,), dtype=torch.float32, device=dev)
    gi = torch.empty((nf,), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    eng2 = Engine(eng.device, order="none") if two_contexts else eng

    class Both:       # what timed() needs of an engine: fence both contexts, merge their per-kernel event times
        def sync(self):
            eng.sync(); eng2.sync()

        def enable_timing(self, on):
            eng.enable_timing(on)
            if eng2 is not eng:
                eng2.enable_timing(on)

        def kernel_times(self):
            kt = eng.kernel_times()
            if eng2 is not eng:
                for k, v in eng2.kernel_times().items():
                    kt.setdefault(k, []).extend(v)
            return kt

    def one():
        # second context, beside everything below: measure_signal_power (pyspecsdr.py:2251) + demodulate AM (:2262) of the same read
        # buffers — the power and the demodulator's mean of |x| from ONE pass over the IQ (pss_demod_power)
        eng2.demod_power(L.MODE_AM, iq_am, nf, n, fs, pcm_am, None, pw)
        eng.demod(L.MODE_USB, iq_ssb, nf, n, fs, pcm_usb, None)
        eng.spectrum_db(iq_am, nf, n, db)                 # compute_fft (:2275)
        if eng2 is not eng:
            eng.order_after(eng2.stream_handle())         # the power values come from the other context's stream
        eng.agc_steps(pw, nf, 20, 29, gi)                 # the gain stepper (:898-919), interval gate off
    ms, kt = timed(Both(), one, 10, launch_only)
    ms_one = None
    if eng2 is not eng and not launch_only:      # the same calls in order on one stream, for the record
        keep, eng2 = eng2, eng
        ms_one, _ = timed(Both(), one, 10, False)
        eng2 = keep
    ver = None
    if verify and not launch_only:
        O = _oracle()
        sos = np.empty((5, 6))
        eng.lib.pss_am_bandpass_sos(sos.ctypes.data)
        taps = eng.ssb_taps(fs)
        frames = sorted({0, nf // 2 + 1, nf - 1})
        ver = {"frames": frames, "db_max_rel": 0.0, "am_pcm_equal": True, "usb_pcm_equal": True, "power_bits_equal": True}
        h_pw = pw.cpu().numpy()
        for k in frames:
            x, y = _host_iq(iq_am, k), _host_iq(iq_ssb, k)
            ver["db_max_rel"] = max(ver["db_max_rel"], _rel(db[k].cpu().numpy(), O.compute_fft(x)))
            ver["am_pcm_equal"] &= bool(np.array_equal(pcm_am[k].cpu().numpy(), O.pcm16_stereo(O.demod_am(x, sos))))
            ver["usb_pcm_equal"] &= bool(np.array_equal(pcm_usb[k].cpu().numpy(), O.pcm16_stereo(O.demod_ssb(y, taps))))
            ver["power_bits_equal"] &= bool(h_pw[k].tobytes() == np.float32(O.power_db(x)).tobytes())
        idx, traj = 20, []
        for p in h_pw:
            idx = O.agc_step(p, idx, 29)
            traj.append(idx)
        ver["agc_trajectory_equal"] = bool(np.array_equal(gi.cpu().numpy(), np.array(traj, np.int32)))
        ver["ok"] = bool(ver["db_max_rel"] <= 1e-4 and ver["am_pcm_equal"] and ver["usb_pcm_equal"] and ver["power_bits_equal"]
                         and ver["agc_trajectory_equal"])
    if eng2 is not eng:
        eng2.close()
    # 131 072 IQ + 65 536 dB + 2 x 65 536 PCM + 4 power per frame (the second IQ batch of the SSB leg is the same 131 072
    # bytes the table counts once: both demodulators are fed their own modulation here, so it is counted too)
    algo = nf * (2 * n * 8 + n * 4 + 2 * n * 4 + 4 + 4)
    e = _entry("cfg3", f"{nf} frames x {n}-pt @2.4 MS/s: measure_signal_power + AGC steps, demodulate AM, demodulate USB (hilbert round "
               f"trip executed) -> int16 stereo, compute_fft dB rows (BASELINE.json configs[2])", ms, kt, algo, nf * n, ver,
               survey_bytes=nf * (n * 8 + n * 4 + 2 * n * 4 + 4))        # = 327 684 B per frame at n = 16 384
    e["contexts"] = 2 if two_contexts else 1
    if ms_one is not None:
        e["ms_one_stream"] = round(ms_one, 4)
    if kt:
        e["kernel_ms_sum"] = round(sum(v["ms"] * v["launches"] for v in kt.values()), 4)
    return e


def cfg4(eng, dev, verify=True, launch_only=False, ns=8192):
    """BASELINE configs[3] on one GPU: 8192 scanner slices x 4096 points (pyspecsdr.py:2539-2552), exact float32 rows."""
    n, fs = 4096, 2.4e6
    iq = synth("scan", ns, n, fs, dev, 20260928 + 4)
    db = torch.empty((ns, n), dtype=torch.float32, device=dev)
    pk = torch.empty((ns,), dtype=torch.float32, device=dev)
    bw = torch.empty((ns,), dtype=torch.float64, device=dev)
    cnt = torch.empty((ns,), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    ms, kt = timed(eng, lambda: eng.scan(iq, ns, n, fs, db, pk, bw, cnt), 20, launch_only)
    ver = None
    if verify and not launch_only:
        O = _oracle()
        frames = sorted({0, 8, ns // 2 + 3, ns - 1})
        ver = {"slices": frames, "db_values": 0, "db_values_differing": 0, "db_max_ulp": 0, "peak_bits_equal": True, "count_equal": True,
               "bandwidth_equal": True}
        for k in frames:
            odb, opk, obw, ocnt = O.scan_slice(_host_iq(iq, k), fs)
            g = db[k].cpu().numpy()
            ulp = np.abs(g.view(np.int32).astype(np.int64) - odb.view(np.int32).astype(np.int64))
            ver["db_values"] += n
            ver["db_values_differing"] += int((ulp != 0).sum())
            ver["db_max_ulp"] = max(ver["db_max_ulp"], int(ulp.max()))
            ver["peak_bits_equal"] &= bool(pk[k].cpu().numpy().tobytes() == np.float32(opk).tobytes())
            ver["count_equal"] &= bool(int(cnt[k]) == ocnt)
            ver["bandwidth_equal"] &= bool(float(bw[k]) == obw)
        # two float64 transforms agree to ~1e-16 of the largest bin: a weak bin beside a strong carrier may round the other way
        ver["ok"] = bool(ver["db_max_ulp"] <= 2 and ver["db_values_differing"] <= 4 and ver["peak_bits_equal"] and ver["count_equal"]
                         and ver["bandwidth_equal"])
    algo = ns * (n * 8 + n * 4 + 16)
    return _entry("cfg4", f"{ns} slices x {n}-pt: unwindowed fft, float32 dB row (the reference's bits), peak, 20-dB-down count, bandwidth "
                  f"(BASELINE.json configs[3], one GPU's whole sweep)", ms, kt, algo, ns * n, ver,
                  survey_bytes=ns * (n * 8 + n * 4 + 12))                # = 49 164 B per slice at n = 4096


CELLS_F32_MAX_FRAC = 2e-3   # float32 dB rows agree with the reference's float64 rows to ~1e-7; a cell differs where a value sits on a quantisation edge


def check_from_iq(O, eng, iq, n, fs, db, lo, hi, lines, pcm, window, mode="waterfall", rows_f64=False, demod="nfm", blk=None, starts=None):
    """Blocks of consecutive frames of a step's outputs against the oracle's OWN step from IQ in the reference's row type (float64 compute_fft
    rows -> np.convolve / np.median / clamp -> min / max over the last `window` rows, np.interp, quantisation: oracle_lib.headline_f64) —
    not against a quantiser run on the device's rows.  rows_f64: the step computes in float64 from the IQ to the cells (extremes to 1e-9, 0 cells
    differing required), whatever type its dB rows are WRITTEN in (db.dtype: float64 rows to 1e-9, float32 rows to 1e-4 * max(|ref|, 1)).  Row extremes,
    int16 PCM (equal; demod "nfm" only: other demodulators are checked by the caller), and the display cells: `cells_differing` counts
    every glyph / colour (or persistence row index) that is not the oracle's, over the lines whose history lies inside the block (all lines
    of a block that starts at frame 0).  float64 rows: 0 is required; float32 rows: at most CELLS_F32_MAX_FRAC of the cells."""
    nf = iq.shape[0]
    blk = min(nf, blk or (window + 226))
    starts = sorted({0, max(0, nf // 3 - 7), max(0, (2 * nf) // 3 + 5), nf - blk}) if starts is None else starts
    taps, sos, zi = eng.nfm_filters(fs)
    thr = O.threads_available()
    res = {"blocks": [[s0, s0 + blk] for s0 in starts], "frames": 0, "lines_checked": 0, "cells_checked": 0, "cells_differing": 0,
           "db_max_rel": 0.0, "extremes_max_abs": 0.0, "pcm_equal": True, "rows": "float64" if rows_f64 else "float32",
           "reference": "oracle's step from IQ in float64 rows (the reference's own row type)"}
    W = lines[0].shape[1]
    for s0 in starts:
        sl = slice(s0, s0 + blk)
        h_iq = iq[sl].cpu().numpy().view(np.complex64).reshape(blk, n)
        o = O.headline_f64(h_iq, fs, taps, sos, zi, window, W, thr, pcm=(demod == "nfm"), keep_db=True, display=mode, disp_h=DISP_H)
        g_db = db[sl].cpu().numpy().astype(np.float64)
        res["db_max_rel"] = max(res["db_max_rel"], float(np.max(np.abs(g_db - o["db"]) / np.maximum(np.abs(o["db"]), 1.0))))
        res["extremes_max_abs"] = max(res["extremes_max_abs"], float(np.max(np.abs(lo[sl].cpu().numpy().astype(np.float64) - o["lo"]))),
                                      float(np.max(np.abs(hi[sl].cpu().numpy().astype(np.float64) - o["hi"]))))
        if demod == "nfm":
            res["pcm_equal"] &= bool(np.array_equal(pcm[sl].cpu().numpy(), o["pcm"]))
        first = 0 if s0 == 0 else window - 1
        for got, want in zip(lines, (o["glyph"], o["colour"])):
            g = got[sl].cpu().numpy()[first:]
            res["cells_checked"] += int(g.size)
            res["cells_differing"] += int(np.count_nonzero(g != want[first:]))
        res["frames"] += blk
        res["lines_checked"] += blk - first
    res["cells_differing_frac"] = res["cells_differing"] / max(1, res["cells_checked"])
    cells_ok = res["cells_differing"] == 0 if rows_f64 else res["cells_differing_frac"] <= CELLS_F32_MAX_FRAC
    db_f64 = db.dtype == torch.float64      # rows as written: float64, or float32 (the cells step writes the float64 value rounded once)
    res["db_rows"] = "float64" if db_f64 else "float32"
    res["ok"] = bool(res["db_max_rel"] <= (1e-9 if db_f64 else 1e-4) and res["extremes_max_abs"] <= (1e-9 if rows_f64 else 1e-4)
                     and res["pcm_equal"] and cells_ok)
    return res


def cfg5_resident(eng, dev, verify=True, launch_only=False, nf=48828):
    """BASELINE configs[4] with the capture resident in HBM: 10 s @ 10 MS/s as 48 828 frames x 2048, per frame dB row, persistence
    accumulator trace (history 10), NFM -> int16 stereo."""
    n, fs, window = 2048, 10e6, 10
    iq = synth("fm", nf, n, fs, dev, 20260928 + 5)
    n_out = eng.demod_out_len(L.MODE_NFM, n, fs)
    db = torch.empty((nf, n), dtype=torch.float32, device=dev)
    lo, hi = torch.empty((nf,), dtype=torch.float32, device=dev), torch.empty((nf,), dtype=torch.float32, device=dev)
    y = torch.empty((nf, DISP_W), dtype=torch.int8, device=dev)
    pcm = torch.empty((nf, n_out, 2), dtype=torch.int16, device=dev)
    torch.cuda.synchronize()
    ms, kt = timed(eng, lambda: eng.frame_pipeline(L.MODE_NFM, iq, nf, n, fs, db, None, lo, hi, DISP_W, y, None, pcm, window=window,
                                                   display="persistence", disp_h=DISP_H), 8, launch_only)
    ver = None
    if verify and not launch_only:
        ver = check_from_iq(_oracle(), eng, iq, n, fs, db, lo, hi, (y,), pcm, window, mode="persistence", blk=window + 54)
    algo = nf * (n * 8 + n * 4 + n_out * 4 + DISP_W)
    return _entry("cfg5_resident", f"{nf} frames x {n}-pt @10 MS/s resident in HBM, every frame: compute_fft dB row + post-process + persistence "
                  f"trace (history {window}) + NFM -> int16 stereo (BASELINE.json configs[4] without the upload)", ms, kt, algo, nf * n, ver,
                  survey_bytes=nf * (n * 8 + n * 4 + n_out * 4 + DISP_W))  # = 24 708 B per frame


def cfg5_streamed(eng, dev, verify=True, launch_only=False, nf=48828, chunk=4096):
    """BASELINE configs[4]: the capture in pinned host memory, chunked hipMemcpyAsync upload double-buffered against compute and download;
    per frame the persistence trace + PCM come back.  PCIe-bound: priced against this box's measured pinned H2D copy rate."""
    n, fs, window = 2048, 10e6, 10
    d_iq = synth("fm", nf, n, fs, dev, 20260928 + 5)
    h_iq = eng.pinned_empty((nf, n), np.complex64)
    torch.from_numpy(h_iq.view(np.float32).reshape(nf, n, 2)).copy_(d_iq)
    torch.cuda.synchronize()
    n_out = eng.demod_out_len(L.MODE_NFM, n, fs)
    outp = {"lines": (eng.pinned_empty((nf, DISP_W), np.int8),), "pcm": eng.pinned_empty((nf, n_out, 2), np.int16),
            "row_lo": eng.pinned_empty((nf,), np.float32), "row_hi": eng.pinned_empty((nf,), np.float32)}

    def one():
        eng.stream_display_nfm(h_iq, fs, chunk, mode="persistence", window=window, disp_h=DISP_H, disp_w=DISP_W, out=outp)
    one()
    ms = None
    if not launch_only:
        t0 = time.perf_counter()
        for _ in range(3):
            one()
        ms = (time.perf_counter() - t0) / 3 * 1e3
    # the same capture with float64 rows from the transform to the cells (pss_h_stream_display_nfm_f64: the reference's cells)
    ms64, out64 = None, None
    if not launch_only:
        out64 = {"lines": (eng.pinned_empty((nf, DISP_W), np.int8),), "pcm": eng.pinned_empty((nf, n_out, 2), np.int16),
                 "row_lo": eng.pinned_empty((nf,), np.float64), "row_hi": eng.pinned_empty((nf,), np.float64)}
        one64 = lambda: eng.stream_display_nfm_f64(h_iq, fs, chunk, mode="persistence", window=window, disp_h=DISP_H, disp_w=DISP_W, out=out64)
        one64()
        t0 = time.perf_counter()
        for _ in range(3):
            one64()
        ms64 = (time.perf_counter() - t0) / 3 * 1e3
    # the link: a plain pinned -> device copy of the same bytes on torch's stream
    link = None
    if not launch_only:
        src = torch.from_numpy(h_iq.view(np.float32).reshape(-1))
        dst = torch.empty_like(d_iq).view(-1)
        dst.copy_(src, non_blocking=True); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            dst.copy_(src, non_blocking=True)
        torch.cuda.synchronize()
        link = nf * n * 8 / ((time.perf_counter() - t0) / 3) / 1e9
        del dst
    ver = None
    if verify and not launch_only:
        # the resident pipeline on the same frames must give the same bytes; its outputs are oracle-checked in cfg5_resident
        db = torch.empty((nf, n), dtype=torch.float32, device=dev)
        lo, hi = torch.empty((nf,), dtype=torch.float32, device=dev), torch.empty((nf,), dtype=torch.float32, device=dev)
        y = torch.empty((nf, DISP_W), dtype=torch.int8, device=dev)
        pcm = torch.empty((nf, n_out, 2), dtype=torch.int16, device=dev)
        torch.cuda.synchronize()
        eng.frame_pipeline(L.MODE_NFM, d_iq, nf, n, fs, db, None, lo, hi, DISP_W, y, None, pcm, window=window, display="persistence", disp_h=DISP_H)
        eng.sync()
        ver = {"lines_equal_resident": bool(np.array_equal(outp["lines"][0], y.cpu().numpy())),
               "pcm_equal_resident": bool(np.array_equal(outp["pcm"], pcm.cpu().numpy())),
               "extremes_equal_resident": bool(np.array_equal(outp["row_lo"], lo.cpu().numpy()) and np.array_equal(outp["row_hi"], hi.cpu().numpy()))}
        O = _oracle()
        taps, sos, zi = eng.nfm_filters(fs)
        ks = sorted({0, chunk - 1, chunk, nf - 1})
        ver["pcm_equal_oracle"] = bool(all(np.array_equal(outp["pcm"][k], O.pcm16_stereo(O.demod_nfm(h_iq[k], fs, taps, sos, zi))) for k in ks))
        ver["frames"] = ks
        # the float64-row capture: persistence traces against the oracle's own step from the IQ on the first chunk (its history starts at frame 0)
        blk = min(nf, 512)
        o = O.headline_f64(h_iq[:blk], fs, taps, sos, zi, window, DISP_W, O.threads_available(), pcm=True, display="persistence", disp_h=DISP_H)
        ver["f64_cells_checked"] = int(blk * DISP_W)
        ver["f64_cells_differing"] = int(np.count_nonzero(out64["lines"][0][:blk] != o["glyph"]))
        ver["f64_pcm_equal_float32_capture"] = bool(np.array_equal(out64["pcm"], outp["pcm"]))
        ver["ok"] = bool(ver["lines_equal_resident"] and ver["pcm_equal_resident"] and ver["extremes_equal_resident"] and ver["pcm_equal_oracle"]
                         and ver["f64_cells_differing"] == 0 and ver["f64_pcm_equal_float32_capture"])
    for a in (h_iq, outp["lines"][0], outp["pcm"], outp["row_lo"], outp["row_hi"]) + ((out64["lines"][0], out64["pcm"], out64["row_lo"], out64["row_hi"]) if out64 else ()):
        eng.pinned_free(a)
    algo = nf * (n * 8 + n * 4 + n_out * 4 + DISP_W)
    e = _entry("cfg5_streamed", f"10 s @ 10 MS/s capture ({nf} frames x {n}-pt) in pinned host memory -> chunks of {chunk} frames, hipMemcpyAsync "
               f"double-buffered (three streams, two buffer sets): persistence trace + NFM int16 per frame back in host memory "
               f"(BASELINE.json configs[4])", ms, {}, algo, nf * n, ver, survey_bytes=nf * (n * 8 + n * 4 + n_out * 4 + DISP_W))
    if ms:
        e["ms_float64_rows"] = None if ms64 is None else round(ms64, 4)   # pss_h_stream_display_nfm_f64: the cell-exact capture
        e["h2d_GBs"] = nf * n * 8 / (ms * 1e-3) / 1e9
        e["link_h2d_GBs"] = link
        e["frac_of_link"] = e["h2d_GBs"] / link if link else None
        e["bound"] = "pcie"
        e["note"] = "PCIe-inclusive: never the headline value; frac = algorithmic HBM bytes / wall / 8 TB/s is reported for uniformity only"
    return e


def wfm_step(eng, dev, verify=True, launch_only=False, nf=65536):
    """The headline step in the reference's DEFAULT mode (--demod WFM, pyspecsdr.py:2855): iq_correction + demodulate_wfm -> int16 stereo,
    compute_fft dB row, post-process, waterfall line at cfg-2 size.  `ms`: pss_frame_pipeline(PSS_MODE_WFM), float32 rows behind the transform
    (the round-5 entry: cells counted against the oracle's); `ms_cells_entry`: pss_frame_pipeline_cells(PSS_MODE_WFM) — float64 from the IQ to
    the cells (the reference's cells, required equal), transform + post-process in one kernel, the dB row written as float32."""
    n, fs, window = 1024, 2.4e6, 30
    iq = synth("fm", nf, n, fs, dev, 20260928 + 6)
    n_out = eng.demod_out_len(L.MODE_WFM, n, fs)
    db = torch.empty((nf, n), dtype=torch.float32, device=dev)
    lo, hi = torch.empty((nf,), dtype=torch.float32, device=dev), torch.empty((nf,), dtype=torch.float32, device=dev)
    gl, co = torch.empty((nf, DISP_W), dtype=torch.int8, device=dev), torch.empty((nf, DISP_W), dtype=torch.int8, device=dev)
    pcm = torch.empty((nf, n_out, 2), dtype=torch.int16, device=dev)
    torch.cuda.synchronize()
    ms, kt = timed(eng, lambda: eng.frame_pipeline(L.MODE_WFM, iq, nf, n, fs, db, None, lo, hi, DISP_W, gl, co, pcm, window=window), 8, launch_only)
    ver = None
    if verify and not launch_only:
        O = _oracle()
        lp, pil, lmr, alpha = eng.wfm_filters(fs)
        _, sos, zi = eng.nfm_filters(fs)
        filt = dict(lp_sos=lp, pilot_sos=pil, lmr_sos=lmr, alpha=alpha, dec_sos=sos, dec_zi=zi)
        ver = check_from_iq(O, eng, iq, n, fs, db, lo, hi, (gl, co), None, window, demod="wfm", blk=window + 98)
        for s0, s1 in ver["blocks"]:
            for k in (s0, s1 - 1):
                a = O.demod_wfm(O.iq_correction(_host_iq(iq, k)), fs, filt)
                ver["pcm_equal"] &= bool(np.array_equal(pcm[k].cpu().numpy(), np.int16(a * 32767)))
        ver["pcm_frames"] = [k for s0, s1 in ver["blocks"] for k in (s0, s1 - 1)]
        ver["ok"] = bool(ver["ok"] and ver["pcm_equal"])
    # the same step through the cell-exact entry
    ms_c, ver_c = None, None
    if not launch_only:
        lo64, hi64 = torch.empty((nf,), dtype=torch.float64, device=dev), torch.empty((nf,), dtype=torch.float64, device=dev)
        pcm_c = torch.empty_like(pcm)
        ms_c, _ = timed(eng, lambda: eng.frame_pipeline_cells(L.MODE_WFM, iq, nf, n, fs, db, None, lo64, hi64, DISP_W, gl, co, pcm_c, window=window), 8, False)
        if verify:
            ver_c = check_from_iq(_oracle(), eng, iq, n, fs, db, lo64, hi64, (gl, co), None, window, rows_f64=True, demod="wfm", blk=window + 98)
            ver_c["pcm_equal"] = bool(torch.equal(pcm_c, pcm))
            ver_c["ok"] = bool(ver_c["ok"] and ver_c["pcm_equal"])
            ver["cells_entry"] = {k: ver_c[k] for k in ("cells_checked", "cells_differing", "db_max_rel", "extremes_max_abs", "pcm_equal", "ok")}
            ver["ok"] = bool(ver["ok"] and ver_c["ok"])
    algo = nf * (n * 8 + n * 4 + n_out * 4 + 2 * DISP_W)
    e = _entry("wfm_step", f"{nf} frames x {n}-pt @2.4 MS/s, every frame: demodulate_signal(WFM) = iq_correction + demodulate_wfm -> int16 stereo, "
               f"compute_fft dB row + post-process + waterfall line (the cfg-2 step in the reference's default mode)", ms, kt, algo, nf * n, ver,
               survey_bytes=nf * (n * 8 + n * 4 + n_out * 4))         # cfg 2's formula with the WFM step's PCM
    e["ms_cells_entry"] = None if ms_c is None else round(ms_c, 4)
    return e


def cfg2_rows(eng, dev, verify=True, launch_only=False, nf=65536, rows="f32"):
    """The cfg 2 step (bench.py's headline workload) as its two other entry points (the headline times pss_frame_pipeline_cells: float64
    arithmetic, float32 rows written): rows = "f32": float32 dB rows
    (pss_frame_pipeline_nfm: the spectrum output's contract is 1e-4 relative; display cells may differ from the reference's where a value sits
    on a quantisation edge — counted below); rows = "f64": float64 rows from IQ to cells (pss_frame_pipeline_nfm_f64: the reference's cells)."""
    n, fs, window = 1024, 2.4e6, 30
    iq = synth("fm", nf, n, fs, dev, 20260928 + 2)
    f64 = rows == "f64"
    dt = torch.float64 if f64 else torch.float32
    n_out = eng.demod_out_len(L.MODE_NFM, n, fs)
    db = torch.empty((nf, n), dtype=dt, device=dev)
    lo, hi = torch.empty((nf,), dtype=dt, device=dev), torch.empty((nf,), dtype=dt, device=dev)
    gl, co = torch.empty((nf, DISP_W), dtype=torch.int8, device=dev), torch.empty((nf, DISP_W), dtype=torch.int8, device=dev)
    pcm = torch.empty((nf, n_out, 2), dtype=torch.int16, device=dev)
    torch.cuda.synchronize()
    fn = eng.frame_pipeline_nfm_f64 if f64 else eng.frame_pipeline_nfm
    ms, kt = timed(eng, lambda: fn(iq, nf, n, fs, db, None, lo, hi, DISP_W, gl, co, pcm, window=window), 20, launch_only)
    ver = None
    if verify and not launch_only:
        ver = check_from_iq(_oracle(), eng, iq, n, fs, db, lo, hi, (gl, co), pcm, window, rows_f64=f64)
    algo = nf * (n * 8 + n * (8 if f64 else 4) + n_out * 4 + 2 * DISP_W)
    name = "cfg2_exact_cells" if f64 else "cfg2_f32_rows"
    return _entry(name, f"{nf} frames x {n}-pt @2.4 MS/s, the headline step with {'float64' if f64 else 'float32'} dB rows: compute_fft + post-process + "
                  f"waterfall line + NFM -> int16 ({'pss_frame_pipeline_nfm_f64: the cells the reference draws' if f64 else 'pss_frame_pipeline_nfm'})",
                  ms, kt, algo, nf * n, ver, survey_bytes=nf * (n * 8 + n * 4 + n_out * 4))   # = 12 328 B per frame


def cfg2_f32_rows(eng, dev, **kw):
    return cfg2_rows(eng, dev, rows="f32", **kw)


def cfg2_exact_cells(eng, dev, **kw):
    return cfg2_rows(eng, dev, rows="f64", **kw)


CONFIGS = {"cfg2_f32_rows": cfg2_f32_rows, "cfg2_exact_cells": cfg2_exact_cells, "cfg3": cfg3, "cfg4": cfg4, "cfg5_resident": cfg5_resident, "cfg5_streamed": cfg5_streamed, "wfm_step": wfm_step}


def other_configs(eng, dev, verify=True, which=None, launch_only=False, small=False, skip=()):
    """-> {config: entry}.  small: a fraction of every batch (CPU-less smoke of the code path on a GPU box with little time).
    skip: configs to leave out (bench.py leaves out the cfg 2 entry of its own headline's row type)."""
    out = {}
    for name in (which or [c for c in CONFIGS if c not in skip]):
        kw = {}
        if small:
            kw = {"cfg3": {"nf": 256}, "cfg4": {"ns": 512}, "cfg5_resident": {"nf": 4100}, "cfg5_streamed": {"nf": 4100, "chunk": 1024},
                  "wfm_step": {"nf": 8192}, "cfg2_f32_rows": {"nf": 4096}, "cfg2_exact_cells": {"nf": 4096}}[name]
        t0 = time.perf_counter()
        try:
            out[name] = CONFIGS[name](eng, dev, verify=verify, launch_only=launch_only, **kw)
        except Exception as ex:  # noqa: BLE001  (a failing config must not take the headline line with it)
            out[name] = {"error": f"{type(ex).__name__}: {ex}", "verified": {"ok": False}}
        out[name]["bench_wall_s"] = round(time.perf_counter() - t0, 2)
        torch.cuda.empty_cache()
    return out


def all_verified(oc):
    return all((e.get("verified") or {}).get("ok", False) for e in oc.values())


if __name__ == "__main__":
    from pyspecsdr_amd.engine import Engine
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    dev = torch.device("cuda", 0)
    eng = Engine(0, order="none")
    res = other_configs(eng, dev, verify="--no-verify" not in sys.argv, which=args or None, launch_only="--launch-only" in sys.argv,
                        small="--small" in sys.argv)
    print(json.dumps(res), flush=True)
    if "--no-verify" not in sys.argv and "--launch-only" not in sys.argv and not all_verified(res):
        sys.exit(3)
