"""Multi-process (world_size 2 and 3, gloo, CPU) tests of the frame-sharding layer used for N > 1 GPUs."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pyspecsdr_amd.shard import gather_rows, shard_counts, shard_range, sharded_scan

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

        def scan_fn(start, count):  # stand-in for Engine.scan on the local GPU: the CPU oracle
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
        if rank == 0:
            db, pk, bw, cnt = res
            q.put((db.numpy(), pk.numpy(), bw.numpy(), cnt.numpy()))
        else:
            assert res is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
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
