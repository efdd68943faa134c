"""Multi-process (world_size 2, 3 and 8, gloo, CPU) tests of the frame-sharding layer used for N > 1 GPUs."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pyspecsdr_amd.shard import (ShardBuffer, gather_packed, gather_rows, halo_from_left, shard_counts, shard_range,
                                 sharded_scan)

import oracle_lib as O


def test_c_abi_shard_range_is_the_python_one():
    """pss_shard_range (the partitioning a host without Python uses, include/pss.h "Multi-GPU") == shard.shard_range."""
    import ctypes as C
    from pyspecsdr_amd import _lib as L
    lib = L.load()
    for n in (0, 1, 7, 8, 37, 8192, 65536, 1000003):
        for world in (1, 2, 3, 8, 13):
            for rank in range(world):
                st, cnt = C.c_long(-1), C.c_long(-1)
                assert lib.pss_shard_range(n, rank, world, C.byref(st), C.byref(cnt)) == 0
                assert (st.value, cnt.value) == shard_range(n, rank, world)
    assert lib.pss_shard_range(8, 2, 2, None, None) == L.PSS_E_ARG and lib.pss_shard_range(-1, 0, 1, None, None) == L.PSS_E_ARG
    assert lib.pss_shard_range(8, 0, 0, None, None) == L.PSS_E_ARG


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 8, 8192, 48828, 65536):
        for w in (1, 2, 3, 4, 8):
            blocks = [shard_range(n, r, w) for r in range(w)]
            assert sum(c for _, c in blocks) == n
            pos = 0
            for s, c in blocks:
                assert s == pos
                pos += c
            cs = shard_counts(n, w)
            assert max(cs) - min(cs) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_slices, n_fft, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = np.load(os.path.join(os.path.dirname(__file__), "golden", "scanner.npz"))
        iq = np.tile(g[f"iq_{n_fft}"], (4, 1))[:n_slices]   # the same sweep on every rank; each takes its block

        def scan_fn(start, count, views):  # stand-in for Engine.scan on the local GPU: the CPU oracle, returning tensors
            db, pk, bw, cnt = [], [], [], []
            for s in range(start, start + count):
                d, p, b, c = O.scan_slice(iq[s], 2.4e6)
                db.append(d); pk.append(p); bw.append(b); cnt.append(c)
            return (torch.from_numpy(np.stack(db)) if count else torch.empty((0, n_fft)),
                    torch.tensor(pk, dtype=torch.float32), torch.tensor(bw, dtype=torch.float64),
                    torch.tensor(cnt, dtype=torch.int32))

        res = sharded_scan(scan_fn, n_slices, n_fft, gather_db=True, dst=0)
        # all_gather variant: every rank ends up with the identical full table
        start, count = shard_range(n_slices, rank, world)
        mine = torch.arange(start, start + count, dtype=torch.float32).unsqueeze(1).repeat(1, 3)
        full = gather_rows(mine, n_slices)
        assert torch.equal(full[:, 0], torch.arange(n_slices, dtype=torch.float32))
        # several fields, written in place into the packed buffer, one collective
        buf = ShardBuffer([("a", (5,), torch.float32), ("b", (), torch.float64), ("c", (2,), torch.int8)], max(shard_counts(n_slices, world)), "cpu")
        idx = torch.arange(start, start + count)
        buf.view("a")[:count] = idx.float().unsqueeze(1) + torch.arange(5).float() / 8
        buf.view("b")[:count] = idx.double() * 1.5
        buf.view("c")[:count] = torch.stack([idx % 7, idx % 5], dim=1).to(torch.int8)
        got = gather_packed(buf, n_slices, dst=None)
        alli = torch.arange(n_slices)
        assert torch.equal(got["a"], alli.float().unsqueeze(1) + torch.arange(5).float() / 8)
        assert torch.equal(got["b"], alli.double() * 1.5)
        assert torch.equal(got["c"], torch.stack([alli % 7, alli % 5], dim=1).to(torch.int8))
        if rank == 0:
            db, pk, bw, cnt = res
            q.put((db.numpy(), pk.numpy(), bw.numpy(), cnt.numpy()))
        else:
            assert res is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_sharded_scan_equals_single_rank(world):
    n_slices, n_fft = 7, 2048          # 7 slices over 2/3 ranks: uneven blocks exercise the padding
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_slices, n_fft, q)) for r in range(world)]
    for p in procs:
        p.start()
    db, pk, bw, cnt = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "scanner.npz"))
    iq = np.tile(g[f"iq_{n_fft}"], (4, 1))[:n_slices]
    for s in range(n_slices):          # byte-for-byte what one rank computes alone
        d, p, b, c = O.scan_slice(iq[s], 2.4e6)
        assert np.array_equal(db[s], d) and pk[s] == p and bw[s] == b and cnt[s] == c


def _halo_worker(rank, world, port, n_rows, halo, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = np.load(os.path.join(os.path.dirname(__file__), "golden", "caller.npz"))
        rows = np.tile(g["rows"], (3, 1))[:n_rows]               # post-processed dB rows in frame order
        start, count = shard_range(n_rows, rank, world)
        mine = torch.from_numpy(rows[start:start + count].copy())
        left = halo_from_left(mine, halo)
        assert left.shape[0] == min(halo, start)
        assert np.array_equal(left.numpy(), rows[start - left.shape[0]:start])
        # the ring a display would hold after this rank's LAST frame, built from halo + own rows, quantised by the oracle
        ring = torch.cat([left, mine], dim=0)[-30:].numpy()
        H, W = [int(v) for v in g["hw"]]
        gl, co = O.waterfall_cells(ring, H - 4, W - 8)
        last = start + count - 1
        want_ring = rows[max(0, last + 1 - 30):last + 1]
        gl2, co2 = O.waterfall_cells(want_ring, H - 4, W - 8)
        assert np.array_equal(gl, gl2) and np.array_equal(co, co2)
        q.put(rank)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_rows", [(2, 40), (3, 40), (3, 8), (8, 100), (8, 9)])
def test_halo_exchange_feeds_ring_accumulators(world, n_rows):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_halo_worker, args=(r, world, port, n_rows, 29, q)) for r in range(world)]
    for p in procs:
        p.start()
    done = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
    assert done == list(range(world)) and all(p.exitcode == 0 for p in procs)


def _subgroup_worker(rank, world, port, q):
    """A sub-group that does not contain global rank 0: group ranks and global ranks differ, so a gather `dst` or a halo
    peer taken for a global rank would address the wrong process (or hang)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        members = list(range(1, world))
        grp = dist.new_group(ranks=members)        # collective over ALL ranks
        if rank in members:
            gw, gr = dist.get_world_size(grp), dist.get_rank(grp)
            assert (gw, gr) == (world - 1, rank - 1)
            n_items = 11
            start, count = shard_range(n_items, gr, gw)
            mine = torch.arange(start, start + count, dtype=torch.float64).unsqueeze(1) * torch.tensor([1.0, -2.0])
            for dst in (0, gw - 1):                # a rank OF THE GROUP (global rank dst + 1)
                full = gather_rows(mine, n_items, dst=dst, group=grp)
                if gr == dst:
                    assert torch.equal(full, torch.arange(n_items, dtype=torch.float64).unsqueeze(1) * torch.tensor([1.0, -2.0]))
                else:
                    assert full is None
            assert torch.equal(gather_rows(mine, n_items, dst=None, group=grp)[:, 0], torch.arange(n_items, dtype=torch.float64))
            left = halo_from_left(mine, 5, group=grp)
            assert torch.equal(left[:, 0], torch.arange(max(0, start - 5), start, dtype=torch.float64))
        dist.barrier()
        q.put(rank)
    finally:
        dist.destroy_process_group()


def test_sub_group_ranks_are_group_ranks():
    world = 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_subgroup_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    done = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
    assert done == list(range(world)) and all(p.exitcode == 0 for p in procs)


def test_bench_gpus_flag_starts_that_many_ranks():
    """`python bench.py --gpus 2` — no launcher — must become two processes that rendezvous, and print ONE result line saying so; a
    launcher whose world size is not --gpus, or fewer visible GPUs than --gpus, must end the run with exit status 2 (no GPU here)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run"], capture_output=True, text=True,
                       env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2 and len(set(lines[0]["pids"])) == 2, r.stdout
    # the line ENDS with the compact per-config summary (a record that keeps only the tail of the line still certifies every config)
    assert list(lines[0])[-1] == "summary" and r.stdout.rstrip().endswith("}}"), r.stdout
    # the shape the driver's scaling run has: eight ranks on one node
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--dry-run"], capture_output=True, text=True,
                       env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and lines[0]["n_gpus"] == 8 and len(set(lines[0]["pids"])) == 8, r.stdout
    # under a launcher with another world size
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run"], capture_output=True, text=True,
                       env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), timeout=120)
    assert r.returncode == 2 and "launcher started 1 rank" in r.stderr
    # WORLD_SIZE = 1 / RANK = 0 alone (what a batch environment leaves behind, not torchrun) do not count as a launcher: the script starts its own ranks
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run"], capture_output=True, text=True,
                       env=dict(env, WORLD_SIZE="1", RANK="0"), timeout=300)
    assert r.returncode == 0 and json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])["n_gpus"] == 2, r.stderr[-500:]
    # ... but a rank of a MULTI-rank job started by a scheduler wrapper (srun / mpirun export WORLD_SIZE and RANK, no LOCAL_RANK) never
    # launches ranks of its own: another world size than --gpus is an error, the local rank comes from the wrapper's per-node index
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run"], capture_output=True, text=True,
                       env=dict(env, WORLD_SIZE="4", RANK="1", SLURM_LOCALID="1"), timeout=120)
    assert r.returncode == 2 and "launcher started 4 rank" in r.stderr, r.stderr[-500:]
    from pyspecsdr_amd import launch
    keep = dict(os.environ)
    try:
        for k in ("LOCAL_RANK", "TORCHELASTIC_RUN_ID"):
            os.environ.pop(k, None)
        os.environ.update(WORLD_SIZE="4", RANK="3", SLURM_LOCALID="1")
        assert launch.under_launcher() and launch.ensure_ranks(4, "bench.py", [], need_gpus=False) == (4, 3, 1)
    finally:
        os.environ.clear()
        os.environ.update(keep)
    if not torch.cuda.is_available():   # the real run refuses to label a 1-GPU (here: 0-GPU) box as 2 GPUs
        for script in ("bench.py", os.path.join("tools", "bench_multi.py")):
            r = subprocess.run([sys.executable, os.path.join(root, script), "--gpus", "2"], capture_output=True, text=True, env=env,
                               timeout=120)
            assert r.returncode == 2 and "GPU(s) are visible" in r.stderr, (script, r.stderr[-500:])


def test_bench_summary_is_last_and_small():
    """bench.py's `summary` (the LAST key of the JSON line): every BASELINE config's time, SURVEY-bytes fraction, traffic ratio, oracle
    verdict and differing-cell count in <= 1500 characters — built here from the committed line of a real run and from a worst-case line."""
    import json
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    lines = sorted(f for f in os.listdir(os.path.join(root, "profiles")) if f.endswith("_bench_line.json"))
    out = json.load(open(os.path.join(root, "profiles", lines[-1])))
    s = bench.summary_of(out)
    txt = json.dumps(s, separators=(",", ":"))      # the separators bench.py prints its line with
    assert len(txt) <= bench.SUMMARY_MAX_CHARS
    for name in ("cfg3", "cfg4", "cfg5_resident", "cfg5_streamed", "wfm_step"):
        assert name in s and {"ms", "fs", "tr", "ok", "cd"} <= set(s[name]), (name, s.get(name))
        assert s[name]["ms"] == round(out["other_configs"][name]["ms"], 4)
    assert s["headline"]["ms"] == round(out["ms_per_step"], 4) and s["src_hash"] == out["src_hash"]
    # worst case: every config present, every number long, every config failing with an error text
    worst = {"ms_per_step": 1234.56789, "ms_per_step_min": 1234.56789, "ms_per_step_max": 1234.56789, "src_hash": "f" * 16,
             "regions": {"shader_clock_mhz_before": 2400, "shader_clock_mhz_after": 2400}, "config": {"rows": "f64"},
             "verified": {"ok": False, "cells_differing": 14680064},
             "roofline": {"frac_survey_bytes": 0.123456, "traffic_ratio": 12.3456, "kernel": "k_spectrum_post_fused", "frac": 0.123456,
                          "kernel_ms": {"k_spectrum_post_fused": 1234.56789}, "traffic_source": "digest@" + "f" * 16},
             "other_configs": {n: {"ms": 12345.6789, "frac_survey_bytes": 0.123456, "traffic_ratio_survey": 12.3456, "ms_one_stream": 12345.6789,
                                   "ms_float64_rows": 12345.6789, "frac_of_link": 0.98765, "error": "RuntimeError: " + "x" * 200,
                                   "verified": {"ok": False, "cells_differing": 14680064}} for n in bench.SUMMARY_CONFIGS}}
    assert len(json.dumps(bench.summary_of(worst), separators=(",", ":"))) <= bench.SUMMARY_MAX_CHARS
