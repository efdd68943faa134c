#!/usr/bin/env python3
"""bench.py — BASELINE.json headline metric on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

One "step" = one pass of the hot path (compute_fft spectrum + NFM demod -> int16) over one batch of
synthetic FM-modulated IQ already resident in HBM: BASELINE.json configs[1] = 65 536 frames x 1024 points
@ 2.4 MS/s per GPU.  Frames are independent, so N GPUs each process their own 65 536-frame batch with no
data-path collective (weak scaling); the timed region is bracketed by barrier + device synchronize and
the max over ranks is reported.  value = complex64 IQ samples per second over all ranks.

Extra objects on the JSON line:
  roofline     — the dominant kernel of the step: its algorithmic bytes / its mean launch duration (HIP events
                 on the library's stream, measured inside the timed region) against the 8 TB/s HBM peak.
  cpu_baseline — the CPU oracle (oracle/pss_oracle.c, a port of the reference's NumPy/SciPy path with the
                 filters designed once) timed on this box's host cores on a bounded sample (rank 0, N=1 only).
"""
import argparse
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
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec

# algorithmic bytes per frame: what each kernel must move if nothing is re-read or spilled
ALGO_BYTES = {
    "k_spectrum": N_FFT * 8 + N_FFT * 4,            # IQ in + float32 dB out
    "k_nfm_fwd": N_FFT * 8,                         # IQ in (y_fwd is an internal hand-off, not algorithmic)
    "k_nfm_bwd": 40,                                # 10 x 2 x int16 out
    "k_nfm_front": N_FFT * 8, "k_nfm_edge": 0, "k_nfm_iir": 40,   # three-kernel fallback path (PSS_NO_FUSED=1)
    "path": N_FFT * 8 + N_FFT * 4 + 40,             # 12 328 B/frame, IQ read once
}


def traffic_of(kernel, n_frames):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC profile (profiles/hbm_traffic_r01.json: separate
    --pmc FETCH_SIZE / WRITE_SIZE passes, gfx950 2x fetch correction), scaled to this run's frame count; None if absent.
    PMC counters cannot be collected from inside this process, so the number is the profiled one, not a live one."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic_r01.json")))[kernel]["traffic_bytes"]
        return t * n_frames / 65536.0
    except Exception:
        return None


def valu_busy_of(kernel):
    """VALU busy fraction of `kernel` from the committed SQ-counter profile (profiles/r01g_sq_counters_bench.txt, digested
    into profiles/hbm_traffic_r01.json); None if absent.  Says how close an issue-bound kernel is to ITS roof."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic_r01.json")))[kernel].get("valu_busy_frac")
    except Exception:
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


def cpu_baseline(iq_host, fs, taps, sos, zi, budget_s=12.0):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    cores = os.cpu_count() or 1
    n = iq_host.shape[1]
    # calibrate on one thread, then size the all-core sample for ~budget_s of wall time
    t0 = time.perf_counter()
    O.batch_spectrum_nfm(iq_host[:256], fs, taps, sos, zi, 1)
    t1 = time.perf_counter() - t0
    per_frame_1t = t1 / 256
    nf = int(min(iq_host.shape[0], max(512, budget_s / per_frame_1t * cores * 0.7)))
    t0 = time.perf_counter()
    O.batch_spectrum_nfm(iq_host[:nf], fs, taps, sos, zi, cores)
    tall = time.perf_counter() - t0
    return {
        "value": nf * n / tall, "unit": "IQ samples/s", "cores": cores, "kind": "port",
        "sample": f"{nf} frames x {n} pts (spectrum + NFM + int16, filters designed once), OpenMP over frames",
        "single_thread_value": n / per_frame_1t,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=N_FRAMES, help="frames per GPU per step (default: BASELINE cfg 2)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from pyspecsdr_amd.engine import Engine
    eng = Engine(local_rank)
    nf, n = args.frames, N_FFT
    iq = synth_fm_iq(nf, n, FS, dev, seed=20260928 + 2 + rank)
    d_db = torch.empty((nf, n), dtype=torch.float32, device=dev)
    n_out = eng.demod_out_len(0, n, FS)
    d_pcm = torch.empty((nf, n_out, 2), dtype=torch.int16, device=dev)

    # waterfall state after the batch (configs[1] names it): the caller's smoothing + median clamp
    # (pyspecsdr.py:2278-2283) on the newest WATERFALL_MAX_LINES = 30 rows and the 36 x 112 cell quantiser
    # (pyspecsdr.py:1342-1406) — a display only ever shows the last 30 rows of a batch.
    WF = min(30, nf)
    d_post = torch.empty((WF, n - 4), dtype=torch.float32, device=dev)
    d_glyph = torch.empty((36, 112), dtype=torch.int8, device=dev)
    d_col = torch.empty((36, 112), dtype=torch.int8, device=dev)

    def step():
        eng.spectrum_nfm(iq, nf, n, FS, d_db, d_pcm)
        eng.spectrum_post(d_db[nf - WF:], WF, n, d_post)
        eng.waterfall_cells(d_post, WF, n - 4, 36, 112, d_glyph, d_col)

    def fence():
        eng.sync()
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    fence()
    # untimed survey pass: every launch bracketed by HIP events on the library's stream -> which kernel dominates, and the
    # table of mean launch durations.  An event is a barrier packet in the queue (~4 us each, 16 per step with every
    # kernel and call bracketed = 6 % of a step), so the timed region below keeps only the dominant kernel's pair.
    eng.enable_timing(True)
    for _ in range(max(3, min(args.steps, 10))):
        step()
    fence()
    ktimes = {k: sum(v) / len(v) for k, v in eng.kernel_times().items()}
    dom = max(ktimes, key=ktimes.get)
    eng.timing_filter(dom)
    fence()
    # timed region: exactly K steps, barrier + synchronize on both sides
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()   # the dominant kernel's launches are bracketed by HIP events; read back after the fence
    fence()
    elapsed = time.perf_counter() - t0
    live = eng.kernel_times().get(dom, [])
    assert len(live) >= args.steps and len(live) % args.steps == 0, (dom, len(live))   # some kernels launch twice a step
    ktimes[dom] = sum(live) / len(live)   # mean launch duration over the K timed steps
    eng.timing_filter(None)
    # outside the timed region: the spectrum kernel alone (inside a step it overlaps the backward IIR pass on a side
    # stream, which stretches its own duration) — this is the HBM-bound kernel of the path
    for _ in range(2):
        eng.spectrum_db(iq, nf, n, d_db)
    eng.sync()
    eng.kernel_times()
    for _ in range(5):
        eng.spectrum_db(iq, nf, n, d_db)
    eng.sync()
    spec_alone = eng.kernel_times().get("k_spectrum", [])
    eng.enable_timing(False)

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    total_samples = float(world) * nf * n * args.steps
    value = total_samples / elapsed

    if rank == 0:
        roof = None
        if dom:
            ms = ktimes[dom]
            achieved = ALGO_BYTES.get(dom, ALGO_BYTES["path"]) * nf / (ms * 1e-3) / 1e9
            roof = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": traffic_of(dom, nf), "valu_busy_frac": valu_busy_of(dom),
                    "kernel_ms": {k: round(v, 4) for k, v in ktimes.items()},
                    "kernel_ms_note": f"{dom}: HIP events inside the timed region; the others: untimed survey pass before it",
                    "path_achieved": ALGO_BYTES["path"] * nf / (elapsed / args.steps) / 1e9}
            if spec_alone:
                sms = sum(spec_alone) / len(spec_alone)
                sa = ALGO_BYTES["k_spectrum"] * nf / (sms * 1e-3) / 1e9
                roof["spectrum_kernel_standalone"] = {"ms": round(sms, 4), "achieved": sa, "frac": sa / HBM_PEAK_GBS}
        out = {
            "metric": "IQ MSamples/sec end-to-end (FFT+dB+FM demod)",
            "value": value / 1e6, "unit": "MSamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 front end / f64 FFT+FIR+IIR / int16 PCM", "data": "synthetic",
            "config": {"workload": f"{nf} frames x {n}-pt complex64 @2.4 MS/s per GPU: compute_fft dB spectrum + "
                                   f"NFM demod -> int16 stereo + waterfall cells of the newest 30 rows (BASELINE.json configs[1])",
                       "frames_per_gpu": nf, "n_fft": n, "sample_rate": FS, "parallelism": f"frames sharded x{world}"},
            "roofline": roof,
        }
        if world == 1 and not args.no_cpu_baseline:
            taps, sos, zi = eng.nfm_filters(FS)
            out["cpu_baseline"] = cpu_baseline(iq[:16384].cpu().numpy().view(np.complex64).reshape(-1, n), FS, taps, sos, zi)
            out["cpu_baseline"]["value"] /= 1e6
            out["cpu_baseline"]["single_thread_value"] /= 1e6
            out["cpu_baseline"]["unit"] = "MSamples/s"
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
