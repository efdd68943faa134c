#!/usr/bin/env python3
"""bench.py — BASELINE.json headline metric on MI355X.

    python bench.py --gpus N --steps K --warmup W

--gpus N means N ranks, one per GPU, whoever starts them: run without a launcher, the script re-executes itself under
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...`; run under one, it checks that the launcher's
WORLD_SIZE is N.  Fewer than N visible GPUs, or a world size other than N, end the run with a message and exit status 2 — never a
one-GPU number labelled N.  The result line carries the world size RCCL reported and the library's version.

One "step" = one pass of the hot path over one batch of synthetic FM-modulated IQ already resident in HBM
(BASELINE.json configs[1] = 65 536 frames x 1024 points @ 2.4 MS/s per GPU), for EVERY frame of the batch:
    compute_fft dB spectrum (signal_processing.py:243-264)            -> computed in float64 (the reference's own arithmetic), the row
                                                                          WRITTEN as float32 [frames][1024]: the float64 value rounded once
                                                                          (north_star: "float32 dB spectra", SURVEY §8(d): 4 bytes per bin)
    the caller's smoothing + median clamp (pyspecsdr.py:2278-2283)     -> float64, from the transform's registers: row extremes + the row
                                                                          resampled to the display width (the rows themselves only with
                                                                          --rows f64 --materialise-post)
    the waterfall accumulator's newest display line (:1342-1406)       -> int8 glyph + colour [frames][112]: the reference's cells
    demodulate_nfm -> int16 stereo (signal_processing.py:91-116)       -> int16 [frames][10][2]
That is --rows cells (pss_frame_pipeline_cells; since round 6 the transform and the post-process of a 1024-point frame are ONE kernel,
k_spectrum_post, so the float64 rows never travel through HBM).  --rows f64: the same arithmetic with the float64 rows written
(pss_frame_pipeline_nfm_f64, round 5's timed step); --rows f32: float32 arithmetic behind the transform (pss_frame_pipeline_nfm: a display
cell may differ from the reference's where a value sits on a quantisation edge).  The default line carries those two steps as
other_configs.cfg2_exact_cells / cfg2_f32_rows.
The timed region (K steps between fences) is run R times (--regions, default 5): ms_per_step / value are the MEDIAN region's, with the
minimum, maximum and the shader clock before / after beside them.
Frames are independent, so N GPUs each process their own batch (weak scaling).  The one exchange step of the path
(BASELINE.json north_star: "a trivial RCCL gather over xGMI") is INSIDE the timed region when N > 1: after every step
each rank's display lines + PCM (one packed buffer, 17 MB) are gathered to rank 0 over RCCL, on a side stream,
overlapped with the next step's compute (two output buffer sets).  --exchange db gathers the float32 dB rows instead
(256 MiB per rank and step: link-bound, reported separately as exchange_db by default).
value = complex64 IQ samples per second over all ranks, max-over-ranks time, barrier + synchronize on both sides.

Extra objects on the JSON line:
  roofline     — the dominant kernel of the step: its algorithmic bytes / its mean launch duration (HIP events
                 on the library's stream, measured inside the timed region) against the 8 TB/s HBM peak; plus the float64
                 issue roof of that kernel when it is the NFM forward kernel, and every HBM-bound kernel's own fraction.
  cpu_baseline — the CPU oracle (oracle/pss_oracle.c, a port of the reference's NumPy/SciPy path with the
                 filters designed once) timed on this box's host cores on a bounded sample (rank 0, N=1 only).
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

N_FRAMES = 65536
N_FFT = 1024
FS = 2.4e6
DISP_W = 112          # max_width 120 - 8 (pyspecsdr.py:1347)
WF_WINDOW = 30        # WATERFALL_MAX_LINES (pyspecsdr.py:131)
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec
F64_PEAK_LANEOPS = 256 * 4 * 16 * 2.4e9  # float64 VALU: 16 lanes / clk / SIMD x 1024 SIMDs x 2.4 GHz = 39.3e12 (= 78.6 TFLOP/s FMA)

SURVEY_BYTES_PER_FRAME = N_FFT * 8 + N_FFT * 4 + 40   # IQ read once, a float32 dB row, the PCM = 12 328 B


def algo_bytes(row_bytes):
    """algorithmic bytes per frame: what each kernel must move if nothing is re-read or spilled; row_bytes = 4 / 8 per dB bin"""
    return {
        "k_spectrum": N_FFT * 8 + N_FFT * row_bytes,    # IQ in + dB row out
        # the fused transform + post-process (pss_spec_post.h): IQ in; dB row, extremes and the row resampled to the display width out
        "k_spectrum_post": N_FFT * 8 + N_FFT * row_bytes + 16 + DISP_W * 8,
        "k_nfm_fwd": N_FFT * 8,                         # IQ in (y_fwd is an internal hand-off, not algorithmic)
        "k_nfm_bwd": 40,                                # 10 x 2 x int16 out
        "k_post": N_FFT * row_bytes + 2 * row_bytes + DISP_W * 8,   # dB row in; extremes + the row resampled to the display width out
        "k_disp_rows": DISP_W * 8 + 2 * DISP_W,         # resampled row in, glyph + colour line out
        "k_slide_extremes": 2 * row_bytes + 16,
        "k_nfm_front": N_FFT * 8, "k_nfm_edge": 0, "k_nfm_iir": 40,   # three-kernel fallback path (PSS_NO_FUSED=1)
        # what the step has to move: IQ read once, the dB row, the PCM, the waterfall line (glyph and colour)
        "path": N_FFT * 8 + N_FFT * row_bytes + 40 + 2 * DISP_W,
    }


ALGO_BYTES = algo_bytes(8)   # (main() replaces it with the timed row type's)
HBM_BOUND = ("k_spectrum", "k_spectrum_post", "k_post", "k_disp_rows")
# float64 VALU operations per input sample and lane of k_nfm_fwd, fixed by the reference's accumulation order (DESIGN.md §4;
# measured with SQ_INSTS_VALU_{FMA,ADD,MUL}_F64: profiles/r02_valu_instruction_mix.txt): 63 fma + 51 add + 12 mul
NFM_FWD_F64_OPS_PER_SAMPLE = 126


def source_hash():
    """sha256 over the kernel sources: profiles are only quoted next to the binary they were taken from."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "pyspecsdr_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h", ".cpp")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def profiled(kernel, n_frames):
    """(traffic bytes per launch, VALU-busy fraction) of `kernel` from the committed rocprofv3 PMC digest
    (profiles/hbm_traffic.json: separate --pmc FETCH_SIZE / WRITE_SIZE passes, gfx950 2x fetch correction), scaled to this
    run's frame count — and only if the digest was taken from THIS source tree (its src_hash matches); else (None, None).
    PMC counters cannot be collected from inside this process, so the number is the profiled one, not a live one."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
        if t.get("src_hash") != source_hash():
            return None, None, f"profiles/hbm_traffic.json is from source {t.get('src_hash')}, this tree is {source_hash()}"
        k = t["kernels"][kernel]
        busy = None
        if k.get("valu_insts") and k.get("rocprof_ms") and n_frames == t["n_frames"]:
            # pure VALU issue time of the launch (class-weighted, see valu_issue_ms) over its profiled duration
            busy = valu_issue_ms(k) / k["rocprof_ms"]
        return k["traffic_bytes"] * n_frames / float(t["n_frames"]), busy, t.get("profile")
    except Exception as ex:  # noqa: BLE001
        return None, None, f"no digest ({type(ex).__name__})"


# Machine-wide VALU issue rates measured with tools/ubench/valu_rate.hip on this part (>= 2 wavefronts per SIMD, hipEvent timing):
# float64 add / mul / fma and conversions to float64 37 T lane-ops/s (16 lanes per clock and SIMD at ~2.3 GHz: 4 clocks per
# wavefront-instruction), float32 / integer instructions 61 T lane-ops/s (32 lanes per clock: 2 clocks per wavefront-instruction).
VALU_RATE_F64 = 37.0e12 / 64     # wavefront-instructions per second, whole chip
VALU_RATE_B32 = 61.0e12 / 64


def valu_issue_ms(d):
    """Pure VALU issue time of one launch from its counter digest entry: float64-class and 32-bit-class instruction counts over the
    measured machine-wide rates of each class."""
    f64 = sum(d.get(k, 0.0) or 0.0 for k in ("sq_insts_valu_add_f64", "sq_insts_valu_mul_f64", "sq_insts_valu_fma_f64", "sq_insts_valu_cvt"))
    b32 = max(d["valu_insts"] - f64, 0.0)
    return (f64 / VALU_RATE_F64 + b32 / VALU_RATE_B32) * 1e3


def step_traffic(kernels, n_frames):
    """HBM bytes of one step from the committed counter digest (FETCH_SIZE x 2 + WRITE_SIZE per kernel, same source-hash rule as
    profiled()): the sum over the step's kernels, and its ratio to SURVEY §8(d)'s algorithmic bytes of the path (IQ read once, dB
    row, PCM: 12 328 B per frame) — how many bytes the step moves per byte it has to."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
        if t.get("src_hash") != source_hash():
            return None, None
        ks = t["kernels"]
        # timer label -> kernel name in the digest (k_spectrum_post: its own name; the lines: one launch since round 6, k_disp_vals_win)
        names = {"k_spectrum": "k_spectrum_r16", "k_post": "k_post_sel", "k_disp_rows": "k_disp_vals_win" if "k_disp_vals_win" in ks else "k_disp_vals"}
        tot = 0.0
        for k in kernels:
            d = ks.get(names.get(k, k))
            if not d or d.get("traffic_bytes") is None:
                return None, None
            tot += d["traffic_bytes"] * n_frames / float(t["n_frames"])
        return tot, tot / (SURVEY_BYTES_PER_FRAME * float(n_frames))
    except Exception:  # noqa: BLE001
        return None, None


def step_valu(kernels, n_frames, ms_per_step):
    """VALU-issue view of the WHOLE step from the committed counter digest (same source-hash rule as profiled()): per kernel, the
    float64-class and 32-bit-class instruction counts over the measured issue rates (above) = the time the kernel would take if nothing
    but VALU issue limited it; their sum over the measured step."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
        if t.get("src_hash") != source_hash() or n_frames != t["n_frames"]:
            return None
        ks = t["kernels"]
        names = {"k_spectrum": "k_spectrum_r16", "k_post": "k_post_sel",
                 "k_disp_rows": "k_disp_vals_win" if "k_disp_vals_win" in t["kernels"] else "k_disp_vals"}
        per = {}
        for k in kernels:
            d = ks.get(names.get(k, k))
            if d and d.get("valu_insts"):
                per[k] = round(valu_issue_ms(d), 4)
        issue_ms = sum(per.values())
        return {"issue_ms": issue_ms, "by_kernel_ms": per, "frac": issue_ms / ms_per_step,
                "note": "VALU wave-instructions of the step's kernels by class (float64 + conversions: 4 clk, everything else priced at the "
                        "float32-FMA / integer-add rate of 2.45 clk per wavefront-instruction; rates measured by tools/ubench/valu_rate.hip) = pure "
                        "issue time, over the measured step.  A LOWER bound: integer compare / min / select / cross-lane instructions issue at the "
                        "float64 rate (profiles/r06_valu_rate.txt), and the select kernels are made of them"}
    except Exception:  # noqa: BLE001
        return None


def synth_fm_iq(n_frames, n, fs, device, seed):
    """FM-modulated carrier + noise, SURVEY §8(d): three audio tones, 5 kHz deviation, A=0.5, sigma=0.02."""
    g = torch.Generator(device=device).manual_seed(seed)
    t = torch.arange(n, device=device, dtype=torch.float64) / fs
    ph0 = 0.1 * torch.arange(n_frames, device=device, dtype=torch.float64).unsqueeze(1)
    m = (0.5 * torch.sin(2 * np.pi * 400 * t + ph0) + 0.3 * torch.sin(2 * np.pi * 1000 * t + 2 * ph0)
         + 0.2 * torch.sin(2 * np.pi * 2500 * t + 3 * ph0))
    phase = 2 * np.pi * 5e3 * torch.cumsum(m, dim=1) / fs + ph0
    iq = torch.empty((n_frames, n, 2), device=device, dtype=torch.float32)
    iq[..., 0] = (0.5 * torch.cos(phase)).float()
    iq[..., 1] = (0.5 * torch.sin(phase)).float()
    iq += 0.02 * torch.randn((n_frames, n, 2), generator=g, device=device, dtype=torch.float32)
    return iq.contiguous()


def host_cpu():
    """CPU model string, hardware threads, and the threads this process may use (affinity capped by the cgroup quota)."""
    model = None
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    quota = None
    try:
        quota = open("/sys/fs/cgroup/cpu.max").read().strip()
    except OSError:
        pass
    return {"model": model, "hw_threads": os.cpu_count(), "affinity": len(os.sched_getaffinity(0)), "cgroup_cpu_max": quota,
            "threads_usable": O.threads_available()}


def cpu_baseline(iq_host, fs, taps, sos, zi, min_wall_s=1.0, max_wall_s=25.0):
    """The oracle's port of the same step in the reference's own row type (float64 compute_fft rows, post-process, waterfall line, NFM +
    int16: oracle_lib.headline_f64), OpenMP over contiguous blocks of frames (scratch and tables once per thread), on as many threads as
    this process may run on, repeated until every thread has had >= min_wall_s of work (the all-core number is not a 6 ms burst)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    host = host_cpu()
    cores = host["threads_usable"]
    nf, n = iq_host.shape
    b1 = O.headline_f64_buffers(512, n, fs, DISP_W)
    O.headline_f64(iq_host[:512], fs, taps, sos, zi, WF_WINDOW, DISP_W, 1, out=b1)      # tables, page faults
    t0 = time.perf_counter()
    O.headline_f64(iq_host[:512], fs, taps, sos, zi, WF_WINDOW, DISP_W, 1, out=b1)
    per_frame_1t = (time.perf_counter() - t0) / 512
    # one pass ~ 2 s of wall time on this host (the whole batch on a 256-thread box, fewer frames on a small one)
    nf = int(min(nf, max(cores * 64, 2.0 / per_frame_1t * cores)))
    iq_host = iq_host[:nf]
    buf = O.headline_f64_buffers(nf, n, fs, DISP_W)
    O.headline_f64(iq_host, fs, taps, sos, zi, WF_WINDOW, DISP_W, cores, out=buf)        # warm-up: thread pool, page faults of the outputs
    reps, t0 = 0, time.perf_counter()
    while True:
        O.headline_f64(iq_host, fs, taps, sos, zi, WF_WINDOW, DISP_W, cores, out=buf)
        reps += 1
        wall = time.perf_counter() - t0
        if wall >= min_wall_s and (wall >= max_wall_s or reps >= 3):
            break
    return {
        "value": reps * nf * n / wall, "unit": "IQ samples/s", "cores": cores, "kind": "port",
        "sample": f"{nf} frames x {n} pts x {reps} passes in {wall:.2f} s wall (float64 spectrum rows + post-process + waterfall line + NFM + "
                  f"int16, filters and tables designed once), OpenMP over contiguous blocks of frames on {cores} threads",
        "single_thread_value": n / per_frame_1t,
        "scaling": (reps * nf * n / wall) / (n / per_frame_1t) / cores,
        "host": host,
    }


def shader_clock_mhz(device_index=0):
    """Current shader clock (MHz) of the GPU this process computes on, from sysfs: the DRM card whose PCI address is the HIP device's (a box
    shows every GPU of the host under /sys/class/drm, the container computes on one of them), hwmon freq1_input (Hz; 'sclk') or, failing
    that, the pp_dpm_sclk level marked '*'.  None when the card cannot be identified."""
    import glob
    import re
    try:
        p = torch.cuda.get_device_properties(device_index)
        bdf = f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}."
    except Exception:  # noqa: BLE001
        return None
    for card in sorted(glob.glob("/sys/class/drm/card*/device")):
        try:
            if bdf not in os.path.realpath(card):
                continue
            for f in sorted(glob.glob(os.path.join(card, "hwmon", "hwmon*", "freq1_input"))):
                return int(round(int(open(f).read().strip()) / 1e6))
            for line in open(os.path.join(card, "pp_dpm_sclk")):
                if "*" in line:
                    m = re.search(r"(\d+)\s*Mhz", line, re.I)
                    if m:
                        return int(m.group(1))
        except (OSError, ValueError):
            continue
    return None


def verify_step(eng, iq, fs, d_db, d_lo, d_hi, pk, o_col, o_pcm, n_out, window, exact_cells):
    """Outside the timed region: the outputs the LAST timed step left in HBM against the CPU oracle (oracle/pss_oracle.c, the checker,
    never the thing measured) on blocks of 256 consecutive frames spread over the batch: the oracle runs its OWN step from the IQ in the
    reference's row type (float64 rows from compute_fft to the cells) — dB rows, row extremes, NFM int16 PCM (equal) and every display cell
    of the lines whose 30-row history lies inside the block (tools/bench_configs.py check_from_iq: cells_differing is a count)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_configs
    import oracle_lib as O
    nf = iq.shape[0]
    glyph = pk[:o_col].view(torch.int8).view(nf, DISP_W)
    colour = pk[o_col:o_pcm].view(torch.int8).view(nf, DISP_W)
    pcm = pk[o_pcm:].view(torch.int16).view(nf, n_out, 2)
    res = bench_configs.check_from_iq(O, eng, iq, iq.shape[1], fs, d_db, d_lo, d_hi, (glyph, colour), pcm, window, rows_f64=exact_cells)
    res["note"] = ("outputs of the last timed step vs the CPU oracle's own step from the IQ (float64 rows), outside the timed region, spot-checked "
                   "on the listed blocks: dB rows and extremes within tolerance, int16 PCM equal, display cells counted")
    return res


SUMMARY_MAX_CHARS = 1500
SUMMARY_CONFIGS = ("cfg2_f32_rows", "cfg2_exact_cells", "cfg3", "cfg4", "cfg5_resident", "cfg5_streamed", "wfm_step")


def _r(x, nd=4):
    return None if x is None else round(float(x), nd)


def summary_of(out):
    """The compact object that ENDS the JSON line (so that a record keeping only the line's tail still holds every config's result):
    per config {ms, fs = fraction of the 8 TB/s HBM peak by SURVEY §8(d) bytes, tr = counter traffic / §8(d) bytes (replayed from the
    committed digest of this source tree, see traffic_source), ok = oracle verification, cd = display cells differing from the oracle's},
    the headline's region spread and shader clocks, and the source hash.  Never longer than SUMMARY_MAX_CHARS characters."""
    s = {}
    for name in SUMMARY_CONFIGS:
        e = (out.get("other_configs") or {}).get(name)
        if e is None:
            continue
        v = e.get("verified") or {}
        cd = v.get("cells_differing", v.get("f64_cells_differing", v.get("db_values_differing")))   # cfg 4: dB values whose float32 bits differ
        c = {"ms": _r(e.get("ms")), "fs": _r(e.get("frac_survey_bytes")), "tr": _r(e.get("traffic_ratio_survey"), 2), "ok": v.get("ok"), "cd": cd}
        if name == "cfg5_streamed":
            c["link"] = _r(e.get("frac_of_link"), 3)
            c["ms64"] = _r(e.get("ms_float64_rows"))
        if name == "cfg3":
            c["ms1"] = _r(e.get("ms_one_stream"))
        if name == "wfm_step":
            c["msx"] = _r(e.get("ms_cells_entry"))      # the same step through the cell-exact entry (pss_frame_pipeline_cells)
        if e.get("error"):
            c["err"] = str(e["error"])[:60]
        s[name] = c
    roof, reg, ver = out.get("roofline") or {}, out.get("regions") or {}, out.get("verified") or {}
    s["headline"] = {"ms": _r(out.get("ms_per_step")), "min": _r(out.get("ms_per_step_min")), "max": _r(out.get("ms_per_step_max")),
                     "fs": _r(roof.get("frac_survey_bytes")), "tr": _r(roof.get("traffic_ratio"), 2), "dom": roof.get("kernel"),
                     "dom_ms": _r((roof.get("kernel_ms") or {}).get(roof.get("kernel"))), "dom_frac": _r(roof.get("frac")),
                     "ok": ver.get("ok"), "cd": ver.get("cells_differing"), "rows": (out.get("config") or {}).get("rows")}
    s["clk"] = [reg.get("shader_clock_mhz_before"), reg.get("shader_clock_mhz_after")]
    s["traffic_source"] = roof.get("traffic_source")
    s["src_hash"] = out.get("src_hash")
    s["keys"] = "ms=ms/pass fs=frac_of_8TB/s_by_SURVEY_8d_bytes tr=counter_traffic/8d_bytes ok=oracle_verified cd=cells_differing"
    txt = json.dumps(s, separators=(",", ":"))       # (the form the line is printed in)
    assert len(txt) <= SUMMARY_MAX_CHARS, len(txt)
    return s


def dry_run(args, world, rank):
    """The launch path without a GPU (tests): N ranks rendezvous over gloo and rank 0 prints one line that says who was there."""
    import torch.distributed as dist
    pids = [os.getpid()]
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        pids = [None] * world
        dist.all_gather_object(pids, os.getpid())
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        out = {"dry_run": True, "n_gpus": world, "gpus_arg": args.gpus, "pids": pids, "value": None}
        out["summary"] = summary_of(out)      # the real line ends with the same object (filled)
        print(json.dumps(out, separators=(",", ":")), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=N_FRAMES, help="frames per GPU per step (default: BASELINE cfg 2)")
    ap.add_argument("--exchange", choices=["display", "db", "none"], default="display",
                    help="what is gathered to rank 0 inside the timed region when N > 1")
    ap.add_argument("--rows", choices=["cells", "f64", "f32"], default="cells",
                    help="the timed step: cells = float64 arithmetic from the IQ to the display cells (the reference's cells), the dB row written as float32 "
                         "(the float64 value rounded once); f64 = the same with the float64 rows written; f32 = float32 rows and float32-row "
                         "post-process (the other two are timed and verified as other_configs.cfg2_exact_cells / cfg2_f32_rows)")
    ap.add_argument("--regions", type=int, default=5, help="how many times the K-step timed region is run (median reported, min / max beside it)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-side", action="store_true",
                    help="skip the untimed side measurements (standalone kernels, the 30-row reading, exchange alone): under a profiler "
                         "every launch then belongs to a step of the default schedule")
    ap.add_argument("--materialise-post", action="store_true",
                    help="also write the post-processed rows [frames][1020] float32 to HBM (default: the display lines are built without them)")
    ap.add_argument("--no-verify", action="store_true", help="skip the oracle check of the last timed step's outputs")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the other BASELINE.json configs (cfg 3, cfg 4, cfg 5 resident + streamed, the WFM step) that the default one-GPU "
                         "run times and verifies after the headline (tools/bench_configs.py)")
    ap.add_argument("--dry-run", action="store_true", help="launch path only (gloo rendezvous, no GPU work): what the CPU test runs")
    args = ap.parse_args()

    # N ranks for --gpus N: re-executes under torch.distributed.run when started bare; exits 2 on any mismatch (pyspecsdr_amd/launch.py)
    from pyspecsdr_amd.launch import ensure_ranks
    # PSS_BENCH_BACKEND=gloo: a FUNCTIONAL test hook for the N-rank code path on a box with fewer GPUs than ranks (tests/test_multi_gpu.py: two
    # ranks sharing one GPU; the collectives go through host memory).  Never a measurement: the line says "backend": "gloo".
    test_backend = os.environ.get("PSS_BENCH_BACKEND", "nccl")
    world, rank, local_rank = ensure_ranks(args.gpus, __file__, sys.argv[1:], need_gpus=not args.dry_run and test_backend != "gloo")
    if args.dry_run:
        return dry_run(args, world, rank)
    # stdout carries ONE line: the result.  Everything else this process or its libraries print there (RCCL's version banner is a
    # C-level printf that libc flushes at exit, i.e. AFTER a Python print) is sent to stderr by pointing fd 1 at fd 2 for the run.
    result_fd = os.dup(1)
    os.dup2(2, 1)
    dist = None
    if world > 1 or os.environ.get("PSS_BENCH_DIST") == "1":   # PSS_BENCH_DIST=1: exercise the RCCL path on one GPU
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if test_backend == "gloo":
            local_rank = local_rank % max(1, torch.cuda.device_count())      # ranks share the visible GPU(s)
            torch.cuda.set_device(local_rank)
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    staged = dist is not None and test_backend == "gloo"     # collectives through host copies (gloo moves no device memory)

    def c_gather(src, recv_rows, dst=0):
        """dist.gather of a device byte buffer to rank dst (recv_rows: its [world, bytes] device tensor there, None elsewhere)."""
        if not staged:
            dist.gather(src, list(recv_rows.unbind(0)) if rank == dst else None, dst=dst)
            return
        h = src.cpu()
        parts = [torch.empty_like(h) for _ in range(world)] if rank == dst else None
        dist.gather(h, parts, dst=dst)
        if rank == dst:
            recv_rows.copy_(torch.stack(parts))

    def c_all_reduce(t, op):
        if not staged:
            dist.all_reduce(t, op=op)
            return
        h = t.cpu()
        dist.all_reduce(h, op=op)
        t.copy_(h)

    def c_all_gather(mine):
        if not staged:
            allr = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(allr, mine)
            return allr
        h = mine.cpu()
        allr = [torch.empty_like(h) for _ in range(world)]
        dist.all_gather(allr, h)
        return allr

    from pyspecsdr_amd.engine import Engine
    eng = Engine(local_rank, order="none")   # this script orders its streams by hand (fences, events)
    comp = torch.cuda.ExternalStream(eng.stream_handle(), device=dev)   # the library's stream, for event ordering
    nf, n = args.frames, N_FFT
    global ALGO_BYTES
    rows_f64 = args.rows == "f64"             # float64 rows in memory
    exact = args.rows in ("cells", "f64")     # float64 arithmetic from the IQ to the cells: the reference's cells
    row_dt = torch.float64 if rows_f64 else torch.float32     # dB rows as written
    ext_dt = torch.float64 if exact else torch.float32        # row extremes (and the side measurements' post-processed rows)
    ALGO_BYTES = algo_bytes(8 if rows_f64 else 4)
    iq = synth_fm_iq(nf, n, FS, dev, seed=20260928 + 2 + rank)
    torch.cuda.synchronize(dev)            # the IQ batch is produced on torch's stream, consumed on the library's
    n_out = eng.demod_out_len(0, n, FS)
    m = n - 4
    d_post = torch.empty((nf, m), dtype=ext_dt, device=dev)      # side measurements; the timed step writes it only with --materialise-post
    if args.materialise_post and args.rows == "cells":
        sys.exit("bench.py: --materialise-post goes with --rows f64 or f32 (the cells step never writes the post-processed rows)")
    step_post = d_post if args.materialise_post else None
    d_lo = torch.empty((nf,), dtype=ext_dt, device=dev)
    d_hi = torch.empty((nf,), dtype=ext_dt, device=dev)
    # two output sets: step k+1 computes into one while step k's is in flight to rank 0.  A set is ONE packed buffer
    # [glyph | colour | pcm] (one message per rank and step) + the dB rows.
    o_col, o_pcm, set_bytes = nf * DISP_W, 2 * nf * DISP_W, 2 * nf * DISP_W + nf * n_out * 4
    packed = [torch.empty((set_bytes,), dtype=torch.uint8, device=dev) for _ in range(2)]
    d_db = [torch.empty((nf, n), dtype=row_dt, device=dev) for _ in range(2 if (dist and args.exchange == "db") else 1)]
    pipeline = eng.frame_pipeline_nfm_f64 if rows_f64 else eng.frame_pipeline_nfm
    from pyspecsdr_amd import _lib as L
    exch = args.exchange if dist is not None else "none"
    comm = torch.cuda.Stream(device=dev) if exch != "none" else None
    recv = None
    if exch != "none" and rank == 0:
        per = set_bytes if exch == "display" else nf * n * d_db[0].element_size()
        recv = torch.empty((world, per), dtype=torch.uint8, device=dev)
    sent = [None, None]     # event: set b's gather has finished (the set may be overwritten)

    def compute(b):
        db = d_db[b % len(d_db)]
        base = packed[b].data_ptr()
        if args.rows == "cells":
            eng.frame_pipeline_cells(L.MODE_NFM, iq, nf, n, FS, db, None, d_lo, d_hi, DISP_W, base, base + o_col, base + o_pcm, window=WF_WINDOW)
        else:
            pipeline(iq, nf, n, FS, db, step_post, d_lo, d_hi, DISP_W, base, base + o_col, base + o_pcm, window=WF_WINDOW)

    def exchange(b):
        src = packed[b] if exch == "display" else d_db[b % len(d_db)].view(torch.uint8).view(-1)
        comm.wait_stream(comp)
        with torch.cuda.stream(comm):
            c_gather(src, recv)
            sent[b] = comm.record_event()

    def step(k):
        b = k & 1
        if exch != "none" and sent[b] is not None:
            comp.wait_event(sent[b])
        compute(b)
        if exch != "none":
            exchange(b)

    def fence():
        eng.sync()
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for k in range(args.warmup):
        step(k)
    fence()
    # untimed survey pass: every launch bracketed by HIP events on the library's stream -> which kernel dominates, and the
    # table of mean launch durations.  An event is a barrier packet in the queue (~4 us each), so the timed region below
    # keeps only the dominant kernel's pair.
    eng.enable_timing(True)
    for k in range(max(3, min(args.steps, 10))):
        compute(k & 1)
    fence()
    ktimes = {k: sum(v) / len(v) for k, v in eng.kernel_times().items()}
    dom = max(ktimes, key=ktimes.get)
    eng.timing_filter(dom)
    fence()
    # timed region: exactly K steps (compute + exchange), barrier + synchronize on both sides — run R times back to back in this process;
    # the line reports the MEDIAN region (ms_per_step, value) with the fastest and slowest beside it and the shader clock around them
    def clock_under_load():
        # the shader clock is sampled WHILE untimed steps are in the queue: read on an idle GPU, sysfs reports the idle level (159 MHz on
        # this part), which says nothing about the clock the timed steps ran at
        for k in range(max(8, args.steps)):
            compute(k & 1)
        c = shader_clock_mhz(dev.index or 0) if rank == 0 else None
        fence()
        eng.kernel_times()       # (drop the event records of these untimed steps)
        return c
    clk0 = clock_under_load()
    regions, live_all = [], []
    for _ in range(max(1, args.regions)):
        t0 = time.perf_counter()
        for k in range(args.steps):
            step(k)   # the dominant kernel's launches are bracketed by HIP events; read back after the fence
        fence()
        regions.append(time.perf_counter() - t0)
        live = eng.kernel_times().get(dom, [])
        assert len(live) >= args.steps and len(live) % args.steps == 0, (dom, len(live))   # some kernels launch twice a step
        live_all.append(sum(live) / len(live))
    eng.timing_filter(None)
    eng.enable_timing(False)
    clk1 = clock_under_load()
    order = sorted(range(len(regions)), key=lambda i: regions[i])
    mid = order[len(order) // 2]
    elapsed = regions[mid]
    ktimes[dom] = live_all[mid]   # mean launch duration over the K steps of the median region
    eng.timing_filter(None)
    eng.enable_timing(False)

    # ---- outside the timed region -------------------------------------------------------------------------------------
    verified = None
    if not args.no_verify:
        last = (args.steps - 1) & 1 if args.steps else 0
        verified = verify_step(eng, iq, FS, d_db[last % len(d_db)], d_lo, d_hi, packed[last], o_col, o_pcm, n_out, WF_WINDOW, exact)
        if dist is not None:      # every rank checks its own outputs; the line reports the conjunction
            okt = torch.tensor([1 if verified["ok"] else 0], dtype=torch.int32, device=dev)
            c_all_reduce(okt, dist.ReduceOp.MIN)
            verified["ok_all_ranks"] = bool(okt.item())

    def timed(fn, reps=5):
        fn(); fence()
        t = time.perf_counter()
        for _ in range(reps):
            fn()
        fence()
        return (time.perf_counter() - t) / reps * 1e3

    side, spec_alone, post_alone, dom_alone = {}, [], [], []
    if not args.no_side:
        # each HBM-bound kernel alone (inside a step the spectrum and the post-process run beside the backward IIR pass)
        eng.enable_timing(True)
        if args.rows == "cells":
            # the display half of the step alone (pss_spectrum_cells: the fused transform + post-process kernel, then the lines)
            base = packed[0].data_ptr()
            for _ in range(2):
                eng.spectrum_cells(iq, nf, n, d_db[0], None, d_lo, d_hi, DISP_W, base, base + o_col, window=WF_WINDOW)
            eng.sync(); eng.kernel_times()
            for _ in range(5):
                eng.spectrum_cells(iq, nf, n, d_db[0], None, d_lo, d_hi, DISP_W, base, base + o_col, window=WF_WINDOW)
            eng.sync()
            spec_alone = eng.kernel_times().get("k_spectrum_post", [])
        else:
            spectrum = eng.spectrum_db_f64 if rows_f64 else eng.spectrum_db
            for _ in range(2):
                spectrum(iq, nf, n, d_db[0])
            eng.sync(); eng.kernel_times()
            for _ in range(5):
                spectrum(iq, nf, n, d_db[0])
            eng.sync()
            spec_alone = eng.kernel_times().get("k_spectrum", [])
            for _ in range(5):      # likewise the post-process kernel (here with the rows written: the separate entry points have no resampled-row output)
                if rows_f64:
                    eng.spectrum_post_f64(d_db[0], nf, n, d_post, d_lo, d_hi)
                else:
                    eng.spectrum_post_extremes(d_db[0], nf, n, d_post, d_lo, d_hi)
            eng.sync()
            post_alone = eng.kernel_times().get("k_post", [])
        for _ in range(5):      # and the demodulator with nothing beside it
            eng.demod(0, iq, nf, n, FS, packed[0].data_ptr() + o_pcm, None)
        eng.sync()
        dom_alone = eng.kernel_times().get(dom, [])
        eng.enable_timing(False)
        # round-1 reading of "waterfall": post-process + cell grid of the newest 30 rows only (a display's last state)
        WF = min(30, nf)
        d_g30 = torch.empty((36, DISP_W), dtype=torch.int8, device=dev)
        d_c30 = torch.empty((36, DISP_W), dtype=torch.int8, device=dev)

        if args.rows == "f32":
            def step30():
                eng.spectrum_nfm(iq, nf, n, FS, d_db[0], packed[0].data_ptr() + o_pcm)
                eng.spectrum_post(d_db[0][nf - WF:], WF, n, d_post)
                eng.waterfall_cells(d_post, WF, m, 36, DISP_W, d_g30, d_c30)
            side["ms_per_step_newest_30_rows_only"] = timed(step30)
        if exch != "none":
            # compute alone, and the exchange alone (both variants), so that overlap can be read off
            side["compute_ms"] = timed(lambda: compute(0))
            comm_db = torch.cuda.Stream(device=dev)

            def xfer(src, per):
                buf = torch.empty((world, per), dtype=torch.uint8, device=dev) if rank == 0 else None
                def go():
                    with torch.cuda.stream(comm_db):
                        c_gather(src, buf)
                return timed(go, 3)
            side["exchange_display_ms"] = xfer(packed[0], set_bytes)
            side["exchange_display_bytes_per_rank"] = set_bytes
            side["exchange_db_ms"] = xfer(d_db[0].view(torch.uint8).view(-1), nf * n * d_db[0].element_size())
            side["exchange_db_bytes_per_rank"] = nf * n * d_db[0].element_size()
            # how much of the in-region exchange the next step's compute hid: 1 = all of it, 0 = none (step = compute + exchange)
            xms = side["exchange_display_ms"] if exch == "display" else side["exchange_db_ms"]
            side["overlap_frac"] = 1.0 - (elapsed / args.steps * 1e3 - side["compute_ms"]) / xms if xms > 0 else None
            side["overlap_note"] = ("overlap_frac = 1 - (ms_per_step - compute_ms) / exchange_ms of the exchange inside the timed region "
                                    f"({exch}); all three measured on this rank (rank 0, the gather's root)")

    other = None
    if world == 1 and dist is None and not args.no_other_configs:
        # the other BASELINE.json configs and the WFM step: timed, priced and oracle-verified one after the other, outside the headline's
        # timed region (which ended above); a failing verification fails the run like the headline's
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_configs
        other = bench_configs.other_configs(eng, dev, verify=not args.no_verify, skip={"cells": (), "f64": ("cfg2_exact_cells",), "f32": ("cfg2_f32_rows",)}[args.rows])

    # per-rank readings gathered over the process group, so that ONE line shows whether a slow N-GPU step is rank 0's gather sharing its own
    # step's HBM (rank 0's compute_ms against the others'), the exchange itself, or a slower GPU (clocks): BASELINE.md §10's suspects (i) - (iii)
    per_rank = None
    if dist is not None:
        mine = torch.tensor([elapsed / args.steps * 1e3, side.get("compute_ms") or float("nan"), float(clk0 or 0), float(clk1 or 0),
                             ktimes.get(dom, float("nan"))], dtype=torch.float64, device=dev)
        if rank != 0:      # (rank 0 sampled its clock above; the other ranks read theirs here, under a short load)
            for k in range(8):
                compute(k & 1)
            mine[2] = mine[3] = float(shader_clock_mhz(dev.index or 0) or 0)
            eng.sync()
        cols = torch.stack(c_all_gather(mine)).cpu().tolist()
        per_rank = {"ms_per_step": [round(c[0], 4) for c in cols], "compute_ms": [None if c[1] != c[1] else round(c[1], 4) for c in cols],
                    "shader_clock_mhz": [int(c[2]) or None for c in cols], "dominant_kernel": dom,
                    "dominant_kernel_ms": [None if c[4] != c[4] else round(c[4], 4) for c in cols]}
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if dist is not None:
        c_all_reduce(t, dist.ReduceOp.MAX)
    elapsed = float(t.item())
    total_samples = float(world) * nf * n * args.steps
    value = total_samples / elapsed

    if rank == 0:
        ms = ktimes[dom]
        achieved = ALGO_BYTES.get(dom, ALGO_BYTES["path"]) * nf / (ms * 1e-3) / 1e9
        traffic, valu_busy, prof_note = profiled(dom, nf)
        # "bound" is the contract's label for the roof that achieved / peak / frac are priced against ("hbm" | "mfma"): HBM for this byte /
        # float64 path.  "limiter" names what actually binds the dominant kernel: k_nfm_fwd's float64 operation count per sample is fixed by
        # the reference's summation order and sits below the HBM ceiling — that roof's own numbers are in "f64_issue"
        roof = {"bound": "hbm", "limiter": "f64_issue" if dom == "k_nfm_fwd" else "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "hbm_frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "valu_busy_frac": valu_busy,
                # traffic, traffic_step, traffic_ratio, valu_busy_frac, step_valu are REPLAYED from the committed rocprofv3 counter digest
                # (PMC passes cannot run inside this process); the digest is used only when its src_hash equals this tree's
                "traffic_source": None if traffic is None else f"digest@{source_hash()}",
                "profile": prof_note,
                "kernel_ms": {k: round(v, 4) for k, v in ktimes.items()},
                "kernel_ms_note": f"{dom}: HIP events inside the timed region; the others: untimed survey pass before it",
                "path_achieved": ALGO_BYTES["path"] * nf / (elapsed / args.steps) / 1e9,
                "path_frac": ALGO_BYTES["path"] * nf / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS,
                "path_bytes_per_frame": ALGO_BYTES["path"],
                # the same with SURVEY §8(d)'s per-frame bytes (IQ once + a float32 dB row + PCM = 12 328 B), whatever row type is timed
                "frac_survey_bytes": SURVEY_BYTES_PER_FRAME * nf / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS}
        if dom_alone:
            ams = sum(dom_alone) / len(dom_alone)
            aa = ALGO_BYTES.get(dom, ALGO_BYTES["path"]) * nf / (ams * 1e-3) / 1e9
            roof["standalone"] = {"ms": round(ams, 4), "achieved": aa, "frac": aa / HBM_PEAK_GBS,
                                  "note": "the same kernel in a plain demodulate call (no spectrum / display work queued around it)"}
            if dom == "k_nfm_fwd":
                roof["standalone"]["f64_issue_frac"] = NFM_FWD_F64_OPS_PER_SAMPLE * float(nf) * n / (ams * 1e-3) / F64_PEAK_LANEOPS
        if dom == "k_nfm_fwd":
            ops = NFM_FWD_F64_OPS_PER_SAMPLE * float(nf) * n / (ms * 1e-3)
            roof["f64_issue"] = {"achieved": ops, "peak": F64_PEAK_LANEOPS, "unit": "float64 lane-ops/s", "frac": ops / F64_PEAK_LANEOPS,
                                 "ops_per_sample": NFM_FWD_F64_OPS_PER_SAMPLE,
                                 "note": "the kernel's binding roof: 63 fma + 51 add + 12 mul float64 per sample are fixed by the "
                                         "reference's accumulation order; peak = 16 lanes/clk/SIMD x 1024 SIMDs x 2.4 GHz"}
        elif "k_nfm_fwd" in ktimes:
            # the step's other long kernel (the forward demodulator and the fused transform + post-process kernel are within a few per cent of each
            # other; which of the two the untimed survey pass ranks first varies from run to run): its own numbers, from the survey pass
            fms = ktimes["k_nfm_fwd"]
            ops = NFM_FWD_F64_OPS_PER_SAMPLE * float(nf) * n / (fms * 1e-3)
            roof["k_nfm_fwd"] = {"ms": round(fms, 4), "hbm_frac": ALGO_BYTES["k_nfm_fwd"] * nf / (fms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                 "f64_issue_frac": ops / F64_PEAK_LANEOPS, "limiter": "f64_issue",
                                 "note": "untimed survey pass (every launch bracketed by events, which stretches it); 126 float64 operations per "
                                         "sample fixed by the reference's accumulation order against 16 lanes/clk/SIMD x 1024 SIMDs x 2.4 GHz"}
        if dom == "k_spectrum_post":
            roof["limiter"] = "valu_issue"      # ~2500 VALU wave-instructions per row, most of the 4-clk class (profiles/r06_valu_rate.txt, NOTEBOOK R6-04)
        roof["traffic_step"], roof["traffic_ratio"] = step_traffic(list(ktimes), nf)
        sv = step_valu(list(ktimes), nf, elapsed / args.steps * 1e3)
        if sv:
            roof["step_valu"] = sv
        hb = {}
        for k in HBM_BOUND:
            if k in ktimes:
                a = ALGO_BYTES[k] * nf / (ktimes[k] * 1e-3) / 1e9
                hb[k] = {"ms": round(ktimes[k], 4), "achieved": a, "frac": a / HBM_PEAK_GBS}
        if spec_alone:
            sk = "k_spectrum_post" if args.rows == "cells" else "k_spectrum"
            sms = sum(spec_alone) / len(spec_alone)
            sa = ALGO_BYTES[sk] * nf / (sms * 1e-3) / 1e9
            hb[sk + "_standalone"] = {"ms": round(sms, 4), "achieved": sa, "frac": sa / HBM_PEAK_GBS}
        if post_alone:
            pms = sum(post_alone) / len(post_alone)
            rb = 8 if rows_f64 else 4
            pa = (N_FFT * rb + (N_FFT - 4) * rb + 2 * rb) * nf / (pms * 1e-3) / 1e9     # standalone: rows in, post-processed rows + extremes out
            hb["k_post_standalone"] = {"ms": round(pms, 4), "achieved": pa, "frac": pa / HBM_PEAK_GBS}
        roof["hbm_bound_kernels"] = hb
        row_desc = {"cells": "computed in float64, the row written as float32 = the float64 value rounded once", "f64": "float64 rows, the reference's own type",
                    "f32": "float32 rows"}[args.rows]
        out = {
            "metric": "IQ MSamples/sec end-to-end (FFT+dB+FM demod)",
            "value": value / 1e6, "unit": "MSamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "ms_per_step_min": min(regions) / args.steps * 1e3, "ms_per_step_max": max(regions) / args.steps * 1e3,
            "regions": {"n": len(regions), "ms_per_step": [round(r / args.steps * 1e3, 5) for r in regions],
                        "reported": "median region (value, ms_per_step, roofline)", "shader_clock_mhz_before": clk0, "shader_clock_mhz_after": clk1},
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"cells": "complex64 IQ / f64 FFT + dB + post-process + FIR + IIR / dB rows stored as f32 / int16 PCM",
                      "f64": "complex64 IQ / f64 FFT + dB rows + post-process + FIR + IIR / int16 PCM",
                      "f32": "f32 front end / f64 FFT+FIR+IIR / f32 dB rows / int16 PCM"}[args.rows],
            "data": "synthetic",
            "config": {"workload": f"{nf} frames x {n}-pt complex64 @2.4 MS/s per GPU, every frame: compute_fft dB spectrum "
                                   f"({row_desc}) + "
                                   f"post-process (smoothing, median clamp) + waterfall display line + NFM demod -> int16 stereo "
                                   f"(BASELINE.json configs[1])",
                       "rows": args.rows,
                       "materialised": {"db_rows": {"cells": "float32", "f64": "float64", "f32": "float32"}[args.rows], "pcm": True, "waterfall_lines": True, "row_extremes": True,
                                        "post_processed_rows": bool(args.materialise_post)},
                       "frames_per_gpu": nf, "n_fft": n, "sample_rate": FS, "parallelism": f"frames sharded x{world}",
                       "exchange": {"display": "RCCL gather to rank 0 of every rank's waterfall lines + PCM (one packed buffer per step), "
                                               "overlapped with the next step, inside the timed region",
                                    "db": "RCCL gather to rank 0 of every rank's float32 dB rows, inside the timed region",
                                    "none": "none (one rank)"}[exch]},
            "roofline": roof,
            "verified": verified,
            "ranks": {"world": world if dist is None else dist.get_world_size(), "gpus_arg": args.gpus,
                      "backend": None if dist is None else dist.get_backend(),
                      "rccl_version": None if dist is None else ".".join(map(str, torch.cuda.nccl.version())),
                      "devices_visible": torch.cuda.device_count()},
            "src_hash": source_hash(),
        }
        if per_rank is not None:
            out["per_rank"] = per_rank
            # BASELINE.md §10, from the measured single-GPU parts: every rank computes its own batch (no data-path collective), the 17 MB
            # gather of lines + PCM rides behind the next step -> ms_per_step = the 1-GPU step + launch / event overhead; "1 GPU" = 0.98-1.05 ms
            one = (0.98, 1.05)
            out["predicted"] = {"ms_per_step": [one[0] + 0.03, one[1] + 0.10], "scaling_vs_1gpu": [round(world * one[0] / (one[1] + 0.10), 2), float(world)],
                                "exchange_display_ms": [0.15, 0.25], "basis": "BASELINE.md §10 (default --exchange display, weak scaling)",
                                "if_slower": "(i) rank 0's compute_ms above the others' = its gather shares its step's HBM / copy engines; (ii) "
                                             "exchange_display_ms >> 0.25 = RCCL serialises the peers; (iii) per_rank.shader_clock_mhz / "
                                             "dominant_kernel_ms differ = a slower GPU"}
        out.update(side)
        if other is not None:
            out["other_configs"] = other
        if world == 1 and not args.no_cpu_baseline:
            taps, sos, zi = eng.nfm_filters(FS)
            out["cpu_baseline"] = cpu_baseline(iq.cpu().numpy().view(np.complex64).reshape(-1, n), FS, taps, sos, zi)
            out["cpu_baseline"]["value"] /= 1e6
            out["cpu_baseline"]["single_thread_value"] /= 1e6
            out["cpu_baseline"]["unit"] = "MSamples/s"
            # the reference ITSELF (per-frame NumPy / SciPy calls, filters redesigned on every call) cannot travel to this box; its only
            # measurement is the build container's: the C port above is ~75x faster than what it restates
            out["cpu_baseline"]["reference_numpy"] = {"value": 0.635, "unit": "MSamples/s", "where": "build container only (8-core Xeon, 1 thread, "
                                                      "NumPy 2.2.6 / SciPy 1.15.3); not measured on this box"}
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        out["summary"] = summary_of(out)     # LAST key: a record that keeps only the line's tail still holds every config's result
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(out, separators=(",", ":")) + "\n").encode())
    if verified is not None and not verified.get("ok_all_ranks", verified["ok"]):
        sys.stderr.write(f"bench.py: the timed steps' outputs do NOT match the oracle: {verified}\n")
        sys.exit(3)
    if other is not None and not args.no_verify:
        bad = {k: v.get("verified") or v.get("error") for k, v in other.items() if not (v.get("verified") or {}).get("ok", False)}
        if bad:
            sys.stderr.write(f"bench.py: other_configs outputs do NOT match the oracle: {bad}\n")
            sys.exit(3)


if __name__ == "__main__":
    main()
