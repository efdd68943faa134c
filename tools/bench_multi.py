#!/usr/bin/env python3
"""Multi-rank benchmark of the sharded configs (BASELINE.json configs[3] / configs[4] halves that need > 1 GPU).

    python tools/bench_multi.py --gpus N --config cfg4 [--steps K]
    (starts N ranks itself — pyspecsdr_amd/launch.py: re-executes under python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1 — or checks the world size of the launcher it was started under; fewer than N visible GPUs: exit 2.
    N = 1 works too: PSS_BENCH_DIST=1 python tools/bench_multi.py --config cfg4 initialises RCCL with one rank.)

cfg5: 10 s @ 10 MS/s capture in 2048-pt frames, every rank streams its block from its own pinned host memory through
spectrum + post-process + display accumulator + NFM and gets display lines + int16 PCM back (PCIe-inclusive rate).
cfg4: scanner sweep of 8192 centre-frequency slices x 4096-pt FFT (pyspecsdr.py:2514-2590), slices sharded over the
ranks (strong scaling: the sweep is fixed), results gathered to rank 0 over RCCL.  Reported separately, as SURVEY §7.2 #5
asks: compute alone, the gather alone (full float32 dB rows: 16 KiB per slice; and the 16 B per slice of peak /
bandwidth / count), and back-to-back sweeps with the gather of sweep k overlapping the compute of sweep k+1.
cfg4host: the same sweep with the IQ where a scanner actually has it — in HOST memory, every rank holding its block of slices in its own
pinned buffer and uploading it over its own PCIe link inside the timed region (chunked, double-buffered against the scan), and only the
16 bytes per slice of (peak, bandwidth, count) gathered.  This is the cfg-4 quantity that CAN scale with the GPU count: with the IQ
already resident, one GPU does the whole sweep in 0.16 ms, less than the 0.22 ms a 128 MiB dB gather into one GPU takes over seven xGMI
links, so "resident IQ -> gathered dB rows" is below 1x at eight GPUs by construction (DESIGN.md par. 6).
One JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist


def cfg5(args, eng, dev, world, rank, use_dist, fence, maxr):
    """10 s @ 10 MS/s capture cut into 2048-pt frames (48 828 frames), contiguous blocks per rank, every rank streaming its
    block from its own pinned host memory: persistence display lines (history 10) + FM -> int16 PCM back to the host.
    PCIe-bound by construction (16 KB in, ~150 B out per frame); the halo exchange is 8 bytes per row."""
    import numpy as np
    from pyspecsdr_amd.multi import sharded_stream_display
    from pyspecsdr_amd.shard import shard_range
    fs, n = 10e6, 2048
    n_frames = int(args.seconds * fs) // n
    start, count = shard_range(n_frames, rank, world)
    h = eng.pinned_empty((count, n), np.complex64)
    rng = np.random.default_rng(5 + rank)
    base = (0.4 * np.exp(2j * np.pi * np.cumsum(rng.standard_normal((64, n)) * 0.03, axis=1))
            + 0.03 * (rng.standard_normal((64, n)) + 1j * rng.standard_normal((64, n)))).astype(np.complex64)
    for f0 in range(0, count, 64):
        h[f0:f0 + 64] = base[:min(64, count - f0)]
    res = {"config": "cfg5", "n_gpus": world, "frames": n_frames, "frames_per_rank": count, "n_fft": n, "chunk_frames": 4096}
    n_out = eng.demod_out_len(0, n, fs)
    pin = lambda shape, dt: eng.pinned_empty(shape, dt)      # results land in pinned host memory: the downloads stay asynchronous
    outs = {"persistence": {"lines": (pin((count, 112), np.int8),), "pcm": pin((count, n_out, 2), np.int16),
                            "row_lo": pin((count,), np.float32), "row_hi": pin((count,), np.float32)}}
    outs["waterfall"] = dict(outs["persistence"], lines=(outs["persistence"]["lines"][0], pin((count, 112), np.int8)))
    for mode in ("persistence", "waterfall"):
        sharded_stream_display(eng, h, fs, 4096, mode=mode, out=outs[mode])            # warm-up (buffers, plans)
        fence()
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            sharded_stream_display(eng, h, fs, 4096, mode=mode, out=outs[mode])
        fence()
        el = maxr(time.perf_counter() - t0) / reps
        res[f"{mode}_s_per_capture"] = el
        res[f"{mode}_samples_per_s"] = n_frames * n / el
        res[f"{mode}_h2d_GBps_per_rank"] = count * n * 8 / el / 1e9
    eng.pinned_free(h)
    if rank == 0:
        print(json.dumps(res), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def cfg4host(args, eng, dev, world, rank, use_dist, fence, maxr):
    """Host IQ sharded over the ranks' PCIe links -> scan -> gathered (peak, bandwidth, count)."""
    from pyspecsdr_amd.shard import scan_buffer, shard_counts, shard_range
    fs, n, ns = 2.4e6, args.n_fft, args.slices
    start, count = shard_range(ns, rank, world)
    g = torch.Generator().manual_seed(4 + rank)
    h = torch.empty((max(count, 1), n, 2), dtype=torch.float32, pin_memory=True)
    h.normal_(0.0, 0.01, generator=g)
    tt = torch.arange(n, dtype=torch.float32)
    for k in range(0, count, 8):      # 1 slice in 8 carries a carrier (SURVEY par. 8d)
        ph = 2 * torch.pi * (0.05 + 0.4 * float(torch.rand(1, generator=g))) * tt
        h[k, :, 0] += 0.3 * torch.cos(ph)
        h[k, :, 1] += 0.3 * torch.sin(ph)
    nchunk = max(1, min(args.chunks, count))
    per = (count + nchunk - 1) // nchunk
    d = [torch.empty((per, n, 2), dtype=torch.float32, device=dev) for _ in range(2)]
    buf = scan_buffer(ns, n, False, dev, None)
    up, comm = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    comp = torch.cuda.ExternalStream(eng.stream_handle(), device=dev)
    recv = torch.empty((world, buf.nbytes), dtype=torch.uint8, device=dev) if (use_dist and rank == 0) else None
    free = [None, None]
    pk, bw, cn = buf.view("peak"), buf.view("bw"), buf.view("cnt")

    def sweep():
        for c in range(nchunk):
            b, c0, c1 = c & 1, c * per, min(count, (c + 1) * per)
            if c1 <= c0:
                break
            if free[b] is not None:
                up.wait_event(free[b])            # the scan of the chunk that used this buffer has finished
            with torch.cuda.stream(up):
                d[b][:c1 - c0].copy_(h[c0:c1], non_blocking=True)
                ev = up.record_event()
            comp.wait_event(ev)
            eng.scan(d[b], c1 - c0, n, fs, None, pk[c0:], bw[c0:], cn[c0:])
            free[b] = comp.record_event()
        if use_dist:
            comm.wait_stream(comp)
            with torch.cuda.stream(comm):
                dist.gather(buf.raw, list(recv.unbind(0)) if rank == 0 else None, dst=0)

    for _ in range(args.warmup):
        sweep()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        sweep()
    fence()
    el = maxr(time.perf_counter() - t0) / args.steps
    # the upload alone (this rank's block, no scan, no gather): what the link gives
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        for c in range(nchunk):
            c0, c1 = c * per, min(count, (c + 1) * per)
            if c1 > c0:
                with torch.cuda.stream(up):
                    d[c & 1][:c1 - c0].copy_(h[c0:c1], non_blocking=True)
    fence()
    el_up = maxr(time.perf_counter() - t0) / args.steps
    res = {"config": "cfg4host", "n_gpus": world, "slices": ns, "n_fft": n, "steps": args.steps, "slices_per_rank": max(shard_counts(ns, world)),
           "chunks_per_rank": nchunk, "sweep_ms_host_iq_gather_peaks": el * 1e3, "samples_per_s": ns * n / el,
           "h2d_GBps_per_rank": count * n * 8 / el / 1e9, "upload_alone_ms": el_up * 1e3, "upload_alone_GBps_per_rank": count * n * 8 / el_up / 1e9,
           "message_bytes_per_rank": buf.nbytes,
           "note": "IQ in pinned host memory, one block of slices per rank, uploaded inside the timed region; 16 B per slice gathered"}
    if rank == 0:
        print(json.dumps(res), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="ranks = GPUs (default: the launcher's WORLD_SIZE, else 1)")
    ap.add_argument("--config", choices=["cfg4", "cfg4host", "cfg5"], default="cfg4")
    ap.add_argument("--chunks", type=int, default=4, help="cfg4host: upload / scan chunks per rank and sweep")
    ap.add_argument("--seconds", type=float, default=10.0, help="cfg5: length of the capture at 10 MS/s")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--slices", type=int, default=8192)
    ap.add_argument("--n-fft", type=int, default=4096)
    args = ap.parse_args()
    from pyspecsdr_amd.launch import ensure_ranks
    gpus = args.gpus if args.gpus is not None else int(os.environ.get("WORLD_SIZE", "1"))
    world, rank, local_rank = ensure_ranks(gpus, __file__, sys.argv[1:])
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29534")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or os.environ.get("PSS_BENCH_DIST") == "1"
    if use_dist:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from pyspecsdr_amd.engine import Engine
    from pyspecsdr_amd.multi import ShardedScanner
    eng = Engine(local_rank, order="none")   # this script orders its streams by hand (fences, events)
    fs, n = 2.4e6, args.n_fft

    def fence():
        eng.sync()
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize(dev)

    def maxr(x):
        t = torch.tensor([x], dtype=torch.float64, device=dev)
        if use_dist:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    if args.config == "cfg5":
        return cfg5(args, eng, dev, world, rank, use_dist, fence, maxr)
    if args.config == "cfg4host":
        return cfg4host(args, eng, dev, world, rank, use_dist, fence, maxr)
    res = {"config": args.config, "n_gpus": world, "slices": args.slices, "n_fft": n, "steps": args.steps}
    for gather_db in (True, False):
        sc = ShardedScanner(eng, args.slices, n, fs, gather_db=gather_db, dst=0)
        g = torch.Generator(device=dev).manual_seed(4 + rank)
        iq = 0.01 * torch.randn((max(sc.count, 1), n, 2), generator=g, device=dev)
        t = torch.arange(n, device=dev, dtype=torch.float32)
        car = torch.arange(sc.count, device=dev) % 8 == 0                  # 1 slice in 8 carries a carrier
        ph = 2 * torch.pi * (0.05 + 0.4 * torch.rand(sc.count, generator=g, device=dev)).unsqueeze(1) * t
        iq[:sc.count, :, 0] += car.unsqueeze(1) * 0.3 * torch.cos(ph)
        iq[:sc.count, :, 1] += car.unsqueeze(1) * 0.3 * torch.sin(ph)
        torch.cuda.synchronize(dev)
        tag = "db" if gather_db else "peaks"
        for _ in range(args.warmup):
            sc.sweep(iq)
        fence()
        # (1) back-to-back sweeps, gather inside the timed region and overlapped with the next sweep's compute
        t0 = time.perf_counter()
        for _ in range(args.steps):
            sc.sweep(iq)
        fence()
        el = maxr(time.perf_counter() - t0)
        res[f"sweep_ms_gather_{tag}"] = el / args.steps * 1e3
        res[f"samples_per_s_gather_{tag}"] = args.slices * n * args.steps / el
        res[f"message_bytes_per_rank_{tag}"] = sc.bufs[0].nbytes
        # (2) the gather alone (the message of the last sweep, no compute)
        if use_dist:
            fence()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                with torch.cuda.stream(sc.comm):
                    dist.gather(sc.bufs[0].raw, list(sc.recv[0].unbind(0)) if rank == 0 else None, dst=0)
            fence()
            res[f"gather_alone_ms_{tag}"] = maxr(time.perf_counter() - t0) / args.steps * 1e3
        if gather_db:
            # (3) compute alone: this rank's block, no exchange
            db = sc.bufs[0].view("db")
            for _ in range(args.warmup):
                eng.scan(iq, sc.count, n, fs, db, sc.bufs[0].view("peak"), sc.bufs[0].view("bw"), sc.bufs[0].view("cnt"))
            fence()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                eng.scan(iq, sc.count, n, fs, db, sc.bufs[0].view("peak"), sc.bufs[0].view("bw"), sc.bufs[0].view("cnt"))
            fence()
            res["compute_alone_ms"] = maxr(time.perf_counter() - t0) / args.steps * 1e3
            res["slices_per_rank"] = sc.count
        del sc
    if rank == 0:
        print(json.dumps(res), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
