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
