"""Multi-rank GPU tests (-m gpu): the sharded scanner driver (pyspecsdr_amd.multi.ShardedScanner) with the real
Engine.scan on every rank must reproduce the single-rank sweep byte for byte.

With >= 2 visible GPUs: one rank per GPU, backend nccl (RCCL over xGMI).  On a one-GPU box the two ranks share
cuda:0 and exchange through gloo (host staging) — the kernels, the packed buffers, the double-buffered pipeline
and the sharding arithmetic are the same code; only the transport differs.
"""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _sweep_iq(k, n_slices, n_fft):
    rng = np.random.default_rng(1000 + k)
    t = np.arange(n_fft)
    iq = 0.02 * (rng.standard_normal((n_slices, n_fft)) + 1j * rng.standard_normal((n_slices, n_fft)))
    for s in range(0, n_slices, 3):                      # every third slice carries a carrier
        iq[s] += 0.4 * np.exp(2j * np.pi * (0.01 + 0.37 * rng.random()) * t)
    return iq.astype(np.complex64)


def _single_rank(n_sweeps, n_slices, n_fft, fs, gather_db):
    from pyspecsdr_amd.engine import Engine
    e = Engine(0)
    out = []
    for k in range(n_sweeps):
        iq = _sweep_iq(k, n_slices, n_fft)
        d_iq = torch.from_numpy(iq.view(np.float32)).cuda()
        db = torch.empty((n_slices, n_fft), dtype=torch.float32, device="cuda")
        pk = torch.empty(n_slices, dtype=torch.float32, device="cuda")
        bw = torch.empty(n_slices, dtype=torch.float64, device="cuda")
        cnt = torch.empty(n_slices, dtype=torch.int32, device="cuda")
        e.scan(d_iq, n_slices, n_fft, fs, db, pk, bw, cnt)
        e.sync()
        out.append((db.cpu().numpy() if gather_db else None, pk.cpu().numpy(), bw.cpu().numpy(), cnt.cpu().numpy()))
    e.close()
    return out


def _worker(rank, world, port, backend, n_sweeps, n_slices, n_fft, fs, gather_db, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = rank if backend == "nccl" else 0
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pyspecsdr_amd.engine import Engine
        from pyspecsdr_amd.multi import ShardedScanner
        e = Engine(dev)
        sc = ShardedScanner(e, n_slices, n_fft, fs, gather_db=gather_db, dst=0)
        handles, keep = [], []
        for k in range(n_sweeps):                         # back-to-back sweeps: the gather of k overlaps the compute of k+1
            iq = _sweep_iq(k, n_slices, n_fft)[sc.start:sc.start + sc.count]
            d_iq = torch.from_numpy(np.ascontiguousarray(iq).view(np.float32)).to(f"cuda:{dev}")
            keep.append(d_iq)
            handles.append(sc.sweep(d_iq))
            if k >= 1:                                    # results of sweep k-1 are read while sweep k is in flight
                res = sc.result(handles[k - 1])
                if rank == 0:
                    q.put((k - 1,) + tuple(None if a is None else a.cpu().numpy() for a in res))
        res = sc.result(handles[-1])
        if rank == 0:
            q.put((n_sweeps - 1,) + tuple(None if a is None else a.cpu().numpy() for a in res))
        if not gather_db:                                 # the dB rows stayed sharded: check this rank's block
            q.put(("local", rank, sc.start, sc.local_db(handles[-1]).cpu().numpy()))
        e.close()
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,gather_db", [(2, True), (2, False), (3, True)])
def test_sharded_scanner_equals_single_rank(world, gather_db):
    ngpu = torch.cuda.device_count()
    assert ngpu >= 1
    backend = "nccl" if ngpu >= world else "gloo"
    n_sweeps, n_slices, n_fft, fs = 3, 37, 4096, 2.4e6     # 37 slices: uneven blocks
    want = _single_rank(n_sweeps, n_slices, n_fft, fs, True)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, backend, n_sweeps, n_slices, n_fft, fs, gather_db, q))
             for r in range(world)]
    for p in procs:
        p.start()
    got, local = {}, {}
    n_msgs = n_sweeps + (world if not gather_db else 0)
    for _ in range(n_msgs):
        m = q.get(timeout=300)
        if m[0] == "local":
            local[m[1]] = (m[2], m[3])
        else:
            got[m[0]] = m[1:]
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    for k in range(n_sweeps):
        db, pk, bw, cnt = got[k]
        wdb, wpk, wbw, wcnt = want[k]
        if gather_db:
            assert np.array_equal(db.view(np.uint32), wdb.view(np.uint32)), k
        else:
            assert db is None
        assert np.array_equal(pk.view(np.uint32), wpk.view(np.uint32)) and np.array_equal(bw, wbw) and np.array_equal(cnt, wcnt), k
    for r, (start, rows) in local.items():
        assert np.array_equal(rows.view(np.uint32), want[-1][0][start:start + rows.shape[0]].view(np.uint32)), r


def _stream_iq(n_frames, n):
    rng = np.random.default_rng(77)
    t = np.arange(n) / 10e6
    iq = (0.4 * np.exp(2j * np.pi * (3e5 * t[None, :] + rng.random((n_frames, 1)))) * (1 + rng.random((n_frames, 1)))
          + 0.03 * (rng.standard_normal((n_frames, n)) + 1j * rng.standard_normal((n_frames, n)))).astype(np.complex64)
    iq[n_frames // 2 - 5:n_frames // 2 + 5] *= 20.0        # a burst across the block boundary of two ranks
    return iq


def _stream_worker(rank, world, port, backend, n_frames, n, fs, mode, q, rows="f32"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = rank if backend == "nccl" else 0
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pyspecsdr_amd.engine import Engine
        from pyspecsdr_amd.multi import sharded_stream_display
        from pyspecsdr_amd.shard import shard_range
        e = Engine(dev)
        start, count = shard_range(n_frames, rank, world)
        h = e.pinned_empty((count, n), np.complex64)       # every rank its own pinned buffer / PCIe link
        h[:] = _stream_iq(n_frames, n)[start:start + count]
        res = sharded_stream_display(e, h, fs, 64, mode=mode, gather_dst=0, rows=rows)
        if rank == 0:
            q.put(res)
        e.pinned_free(h)
        e.close()
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _stream_worker_engine_group(rank, world, idfile, dbl, box, n_frames, n, fs, mode, q, rows):
    """The same capture sharded over `world` processes whose exchange steps run BEHIND THE C ABI (shard.EngineGroup: pss_gather_packed /
    pss_halo_from_left) — no torch.distributed at all.  All ranks share this box's GPU, so libpss.so is pointed at the transport double."""
    os.environ["PSS_RCCL_LIB"], os.environ["PSS_RCCL_DOUBLE_DIR"] = dbl, box
    import time
    from pyspecsdr_amd.engine import Engine
    from pyspecsdr_amd.multi import sharded_stream_display
    from pyspecsdr_amd.shard import EngineGroup, shard_range
    e = Engine(0)
    if rank == 0:
        ident = e.comm_id()
        with open(idfile + ".tmp", "wb") as f:
            f.write(bytes(ident))
        os.replace(idfile + ".tmp", idfile)
    else:
        for _ in range(3000):
            if os.path.exists(idfile):
                break
            time.sleep(0.02)
        ident = open(idfile, "rb").read()
    e.comm_init(ident, rank, world)
    grp = EngineGroup(e)
    assert (grp.rank, grp.world) == (rank, world)
    start, count = shard_range(n_frames, rank, world)
    h = e.pinned_empty((count, n), np.complex64)
    h[:] = _stream_iq(n_frames, n)[start:start + count]
    res = sharded_stream_display(e, h, fs, 64, mode=mode, gather_dst=0, rows=rows, group=grp)
    if rank == 0:
        q.put(res)
    e.pinned_free(h)
    e.comm_free()
    e.close()


@pytest.mark.parametrize("world,mode,rows", [(2, "waterfall", "f64"), (3, "persistence", "f32")])
def test_sharded_stream_over_the_c_abi_exchange_with_real_peers(world, mode, rows, tmp_path):
    """shard.EngineGroup's multi-rank branches (the exchange steps behind the C ABI instead of torch.distributed) with real peer processes:
    sharded_stream_display on 2 / 3 ranks == one rank streaming the whole capture.  The ranks share this box's one GPU, which RCCL refuses, so
    libpss.so loads tests/rccl_double/rccl_double.c in RCCL's place (PSS_RCCL_LIB: test infrastructure — RCCL's signatures and group semantics
    over files + hipMemcpy).  Covers block counts by all-gather, the halo of row extremes from the left neighbour(s), the uneven packed gather."""
    import subprocess
    from pyspecsdr_amd.engine import Engine
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dbl = str(tmp_path / "librccl_double.so")
    subprocess.run(["gcc", "-O2", "-shared", "-fPIC", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", os.path.join(root, "tests", "rccl_double", "rccl_double.c"),
                    "-L/opt/rocm/lib", "-lamdhip64", "-o", dbl], check=True)
    box = tmp_path / "mail"
    box.mkdir()
    n_frames, n, fs = 301, 2048, 10e6
    e = Engine(0)
    h = e.pinned_empty((n_frames, n), np.complex64)
    h[:] = _stream_iq(n_frames, n)
    want = (e.stream_display_nfm_f64 if rows == "f64" else e.stream_display_nfm)(h, fs, 64, mode=mode)
    e.pinned_free(h)
    e.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_stream_worker_engine_group, args=(r, world, str(tmp_path / "id"), dbl, str(box), n_frames, n, fs, mode, q, rows)) for r in range(world)]
    for p in procs:
        p.start()
    lines, pcm = q.get(timeout=300)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    assert len(lines) == len(want["lines"]) and all(np.array_equal(a, b) for a, b in zip(lines, want["lines"]))
    assert np.array_equal(pcm, want["pcm"])


@pytest.mark.parametrize("world,mode,rows", [(2, "waterfall", "f32"), (3, "persistence", "f32"), (2, "persistence", "f64"), (3, "waterfall", "f64")])
def test_sharded_stream_equals_single_rank(world, mode, rows):
    """BASELINE configs[4] on N ranks: each rank streams its block of the capture from its own pinned memory; display
    lines (history continued across the block boundaries through the halo of row extremes) and PCM gathered to rank 0 must
    equal one rank streaming the whole capture — with float32 rows and with the cell-exact float64 rows."""
    from pyspecsdr_amd.engine import Engine
    ngpu = torch.cuda.device_count()
    backend = "nccl" if ngpu >= world else "gloo"
    n_frames, n, fs = 301, 2048, 10e6
    e = Engine(0)
    h = e.pinned_empty((n_frames, n), np.complex64)
    h[:] = _stream_iq(n_frames, n)
    want = (e.stream_display_nfm_f64 if rows == "f64" else e.stream_display_nfm)(h, fs, 64, mode=mode)
    e.pinned_free(h)
    e.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_stream_worker, args=(r, world, port, backend, n_frames, n, fs, mode, q, rows)) for r in range(world)]
    for p in procs:
        p.start()
    lines, pcm = q.get(timeout=300)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    assert len(lines) == len(want["lines"]) and all(np.array_equal(a, b) for a, b in zip(lines, want["lines"]))
    assert np.array_equal(pcm, want["pcm"])


def test_two_contexts_in_one_process_interleaved():
    """Two library contexts used alternately from one thread: every entry point runs on ITS context's device and
    scratch, whatever the calling thread's current device is.  With two visible GPUs the second context lives on cuda:1 while
    the thread's current device stays cuda:0; on a one-GPU box both live on cuda:0 (separate streams, scratch, tables)."""
    from pyspecsdr_amd.engine import Engine
    ngpu = torch.cuda.device_count()
    dev_b = 1 if ngpu >= 2 else 0
    torch.cuda.set_device(0)
    a, b = Engine(0), Engine(dev_b)
    rng = np.random.default_rng(31)
    nf, n, fs = 300, 1024, 2.4e6
    iq = (0.4 * np.exp(2j * np.pi * np.cumsum(rng.standard_normal((nf, n)) * 0.05, axis=1)) +
          0.03 * (rng.standard_normal((nf, n)) + 1j * rng.standard_normal((nf, n)))).astype(np.complex64)
    h = torch.from_numpy(iq.view(np.float32).reshape(nf, n, 2))
    xa, xb = h.to("cuda:0"), h.to(f"cuda:{dev_b}")
    torch.cuda.synchronize(0); torch.cuda.synchronize(dev_b)
    n_out = a.demod_out_len(0, n, fs)
    out = {}
    for name, e, x, dev in (("a", a, xa, 0), ("b", b, xb, dev_b)):
        d = f"cuda:{dev}"
        out[name] = dict(db=torch.empty((nf, n), dtype=torch.float32, device=d), pcm=torch.empty((nf, n_out, 2), dtype=torch.int16, device=d),
                         pw=torch.empty((nf,), dtype=torch.float32, device=d), post=torch.empty((nf, n - 4), dtype=torch.float32, device=d))
    assert torch.cuda.current_device() == 0
    for _ in range(2):                       # interleave: each call would trample the other's scratch if it were shared
        a.spectrum_nfm(xa, nf, n, fs, out["a"]["db"], out["a"]["pcm"])
        b.power_db(xb, nf, n, out["b"]["pw"])
        b.spectrum_nfm(xb, nf, n, fs, out["b"]["db"], out["b"]["pcm"])
        a.spectrum_post(out["a"]["db"], nf, n, out["a"]["post"])
        a.power_db(xa, nf, n, out["a"]["pw"])
        b.spectrum_post(out["b"]["db"], nf, n, out["b"]["post"])
    a.sync(); b.sync()
    assert torch.cuda.current_device() == 0          # the calls restored the caller's device
    for k in out["a"]:
        assert torch.equal(out["a"][k].cpu(), out["b"][k].cpu()), k
    b.close(); a.close()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("script,extra", [("bench.py", ["--steps", "4", "--warmup", "1", "--no-cpu-baseline", "--no-other-configs", "--frames", "4096"]),
                                          ("tools/bench_multi.py", ["--config", "cfg4"])])
def test_rccl_path_runs_with_one_rank(script, extra):
    """The nccl (= RCCL) side of the multi-GPU drivers with a world of ONE rank (PSS_BENCH_DIST=1): process-group initialisation on the GPU,
    the packed gather to rank 0 on the side stream, the barrier fences and the all-reduce of the verification flag — everything an N-rank
    run executes except the transport between GPUs, which a one-GPU box cannot show.  bench.py's line must carry the exchange fields."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PSS_BENCH_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, script)] + extra, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    if script == "bench.py":
        line = json.loads(r.stdout.strip().splitlines()[-1])
        assert line["ranks"]["backend"] == "nccl" and line["n_gpus"] == 1
        assert line["verified"]["ok"] and line["verified"]["ok_all_ranks"]
        assert line["exchange_display_ms"] is not None and line["compute_ms"] is not None and line["overlap_frac"] is not None
        # the fields the first N > 1 record is to be read against: per-rank readings gathered over the process group, the prediction
        pr = line["per_rank"]
        assert len(pr["ms_per_step"]) == 1 and pr["compute_ms"][0] is not None and pr["dominant_kernel_ms"][0] is not None
        assert line["predicted"]["ms_per_step"][0] < line["predicted"]["ms_per_step"][1] and "if_slower" in line["predicted"]
        assert list(line)[-1] == "summary" and line["summary"]["headline"]["rows"] == "cells" and line["summary"]["headline"]["cd"] == 0


def test_bench_two_ranks_functional_on_one_gpu():
    """bench.py --gpus 2 END TO END with two real ranks: the self-launch under torch.distributed.run, double-buffered output sets, the in-region
    gather to rank 0 with its `sent` events, the barrier fences, max-over-ranks timing, the all-reduced verification flag, the per-rank
    readings, rank 0's one line.  RCCL refuses two ranks on one GPU, so PSS_BENCH_BACKEND=gloo routes the collectives through host copies: a
    functional test of the N-rank CODE (which no round has been able to run on N GPUs), never a measurement — the line says backend gloo."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(PSS_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    for exchange in ("display", "db"):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--regions", "2", "--frames", "4096",
                            "--exchange", exchange, "--no-cpu-baseline"], cwd=root, env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        lines = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, r.stdout[-2000:]
        line = lines[0]
        assert line["n_gpus"] == 2 and line["ranks"]["world"] == 2 and line["ranks"]["backend"] == "gloo" and line["scaling"] == "weak"
        assert line["verified"]["ok"] and line["verified"]["ok_all_ranks"] and line["verified"]["cells_differing"] == 0
        pr = line["per_rank"]
        assert len(pr["ms_per_step"]) == 2 and all(v is not None and v > 0 for v in pr["ms_per_step"] + pr["compute_ms"] + pr["dominant_kernel_ms"])
        assert abs(line["value"] * 1e6 - 2 * 4096 * 1024 / (line["ms_per_step"] * 1e-3)) < 1e-3 * line["value"] * 1e6      # whole-job samples / max-over-ranks time
        assert line["ms_per_step"] >= max(pr["ms_per_step"]) - 1e-3      # (the per-rank values are rounded to 4 decimals)
        assert line["exchange_display_ms"] is not None and line["exchange_db_ms"] is not None and "other_configs" not in line
        assert list(line)[-1] == "summary" and line["predicted"]["scaling_vs_1gpu"][1] == 2.0


@pytest.mark.parametrize("rows", ["f64", "f32"])
def test_bench_other_row_types_still_run(rows):
    """bench.py --rows f64 / f32 (round 5's timed step and the float32-row step; the default since round 6 is --rows cells): a short run at a
    fraction of the batch must verify against the oracle and print the line."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--rows", rows, "--steps", "3", "--warmup", "1", "--regions", "2", "--frames", "4096",
                        "--no-cpu-baseline", "--no-other-configs"], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["config"]["rows"] == rows and line["verified"]["ok"] and line["config"]["materialised"]["db_rows"] == {"f64": "float64", "f32": "float32"}[rows]
    if rows == "f64":
        assert line["verified"]["cells_differing"] == 0
