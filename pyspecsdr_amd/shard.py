"""Multi-GPU sharding of independent frames / scanner slices (one process per GPU, torch.distributed).

Frames share nothing (every read buffer is demodulated and transformed on its own —
signal_processing.py:91-116, pyspecsdr.py:2523-2578), so the path shards by contiguous blocks of the frame
index with NO data-path collective.  The only exchange step is the optional gather of results to one
rank — per-slice (peak, bandwidth) pairs, or the float32 dB rows of a scanner sweep (BASELINE.json configs[3]).
On ROCm the "nccl" backend is RCCL; the xGMI mesh gives every pair of GPUs its own link, so a flat
all_gather / gather (each peer sends its block directly) is the right collective — no ring staging.
The same code runs on CPU tensors with the gloo backend (tests/test_shard.py).
"""
import torch
import torch.distributed as dist


def shard_range(n_items, rank, world):
    """Contiguous block of rank `rank`: items [start, start+count), sizes differ by at most one."""
    base, rem = divmod(int(n_items), int(world))
    start = rank * base + min(rank, rem)
    return start, base + (1 if rank < rem else 0)


def shard_counts(n_items, world):
    return [shard_range(n_items, r, world)[1] for r in range(world)]


def gather_rows(local, n_items, dst=None, group=None):
    """Collect per-item result rows from every rank.

    local : tensor [count_r, ...] holding this rank's block (count_r = shard_range(...)[1]).
    dst   : None -> all_gather (every rank gets the full [n_items, ...] tensor);
            int  -> gather to that rank only (others get None).
    Blocks are padded to the largest block so one fixed-size collective moves everything.
    """
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    counts = shard_counts(n_items, world)
    assert local.shape[0] == counts[rank], (local.shape, counts, rank)
    mx = max(counts)
    tail = tuple(local.shape[1:])
    padded = local
    if counts[rank] < mx:
        padded = torch.zeros((mx,) + tail, dtype=local.dtype, device=local.device)
        padded[:counts[rank]] = local
    padded = padded.contiguous()
    if dst is None:
        buf = torch.empty((world * mx,) + tail, dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(buf, padded, group=group)
        parts = [buf[r * mx:r * mx + counts[r]] for r in range(world)]
        return torch.cat(parts, dim=0)
    bufs = [torch.empty_like(padded) for _ in range(world)] if rank == dst else None
    dist.gather(padded, bufs, dst=dst, group=group)
    if rank != dst:
        return None
    return torch.cat([bufs[r][:counts[r]] for r in range(world)], dim=0)


def halo_from_left(local, halo, group=None):
    """Rows a rank needs from its LEFT neighbour so that the ring accumulators (waterfall: last 30 post-processed rows,
    persistence: last 10 — pyspecsdr.py:130-132,151-154) of its first frames see the frames just before its block.

    local : tensor [count_r, ...], this rank's block of rows in frame order.
    Returns a tensor [h, ...] (h <= halo): the last h rows that precede this block in global order; rank 0 gets an empty
    tensor.  One point-to-point message per neighbour pair (xGMI: a direct peer link, no collective).  A neighbour whose
    own block is shorter than `halo` forwards what it received, so short blocks still deliver a full halo.
    """
    tail = tuple(local.shape[1:])
    empty = torch.empty((0,) + tail, dtype=local.dtype, device=local.device)
    if not dist.is_initialized() or dist.get_world_size(group) == 1 or halo <= 0:
        return empty
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    got = empty
    if rank > 0:
        n = torch.zeros(1, dtype=torch.int64, device=local.device)
        dist.recv(n, src=rank - 1, group=group)
        got = torch.empty((int(n.item()),) + tail, dtype=local.dtype, device=local.device)
        if got.shape[0]:
            dist.recv(got, src=rank - 1, group=group)
    if rank < world - 1:
        rows = torch.cat([got, local], dim=0)[-halo:].contiguous()   # my last rows, topped up with what I was handed
        n = torch.tensor([rows.shape[0]], dtype=torch.int64, device=local.device)
        dist.send(n, dst=rank + 1, group=group)
        if rows.shape[0]:
            dist.send(rows, dst=rank + 1, group=group)
    return got


def sharded_scan(scan_fn, n_slices, n_fft, gather_db=False, dst=0, group=None):
    """Scanner sweep over n_slices centre frequencies, sharded over the ranks.

    scan_fn(start, count) -> (db [count, n_fft] float32 or None, peak [count] float32, bw [count] float64,
                              cnt [count] int32) for this rank's block (e.g. Engine.scan on the local GPU).
    Returns on rank dst: (db or None, peak, bw, cnt) for all slices in sweep order; None elsewhere.
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    start, count = shard_range(n_slices, rank, world)
    db, peak, bw, cnt = scan_fn(start, count)
    out_db = gather_rows(db, n_slices, dst=dst, group=group) if (gather_db and db is not None) else None
    out_peak = gather_rows(peak, n_slices, dst=dst, group=group)
    out_bw = gather_rows(bw, n_slices, dst=dst, group=group)
    out_cnt = gather_rows(cnt, n_slices, dst=dst, group=group)
    if rank != dst and world > 1:
        return None
    return out_db, out_peak, out_bw, out_cnt
