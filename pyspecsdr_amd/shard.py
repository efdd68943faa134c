"""Multi-GPU sharding of independent frames / scanner slices (one process per GPU, torch.distributed).

Frames share nothing (every read buffer is demodulated and transformed on its own —
signal_processing.py:91-116, pyspecsdr.py:2523-2578), so the path shards by contiguous blocks of the frame
index with NO data-path collective.  The only exchange step is the gather of results to one rank — per-slice
(peak, bandwidth, count) and, optionally, the float32 dB rows of a scanner sweep (BASELINE.json configs[3]) — and, for
the display accumulators, a halo of the rows that precede a rank's block.

One sweep = ONE collective: every rank's results live in one packed buffer (`ShardBuffer`: a section per field, each
section sized for the largest block, so the kernels write their outputs in place and nothing is copied or padded
afterwards) and `gather_packed` moves it with a single gather / all_gather.  On ROCm the "nccl" backend is RCCL; a
gather is grouped point-to-point sends, which on the xGMI mesh go over each peer's own link to the root — no ring
staging.  The same code runs on CPU tensors with the gloo backend (tests/test_shard.py).
"""
import numpy as np
import torch
import torch.distributed as dist


def _torch_version_ok():
    try:
        return tuple(int(x) for x in torch.__version__.split("+")[0].split(".")[:2]) >= (2, 6)
    except ValueError:
        return True            # an unparsable development version string: assume recent


if not _torch_version_ok():
    raise ImportError(f"pyspecsdr_amd.shard / .multi address sub-group peers by their rank IN THE GROUP (dist.gather(group_dst=), "
                      f"dist.P2POp(group_peer=)), which torch.distributed has since 2.6; this is torch {torch.__version__}")


def shard_range(n_items, rank, world):
    """Contiguous block of rank `rank`: items [start, start+count), sizes differ by at most one."""
    base, rem = divmod(int(n_items), int(world))
    start = rank * base + min(rank, rem)
    return start, base + (1 if rank < rem else 0)


def shard_counts(n_items, world):
    return [shard_range(n_items, r, world)[1] for r in range(world)]


class EngineGroup:
    """A `group` whose exchange steps run BEHIND THE C ABI (pss_gather_packed / pss_halo_from_left: libpss.so opens librccl itself) instead
    of over torch.distributed — the route a host without torch takes (include/pss.h "Multi-GPU", examples/pss_sweep_ranks.c).
    engine: an Engine that has joined a communicator (Engine.comm_init).
    Ordering: the C-ABI collectives run on the ENGINE's stream.  before() orders that stream behind torch's current stream (the producer
    of the buffers), after() waits for the collective on the host — the helpers below read the results straight away (counts -> Python
    ints, gathered views), so a blocking wait is the right one whatever `order` the Engine was created with.
    Exercised with one rank only (tests/test_gpu_parity.py; the boxes this was built on expose one GPU): the multi-rank branches follow
    pss_comm.cpp's grouped send / receive, which a second rank has never executed."""

    def __init__(self, engine):
        self.engine = engine
        self.rank, self.world = engine.comm_size()

    def before(self, device):
        self.engine.order_after(torch.cuda.current_stream(device).cuda_stream)

    def after(self):
        self.engine.sync()


def _world_rank(group=None):
    if isinstance(group, EngineGroup):
        return group.world, group.rank
    if not dist.is_initialized():
        return 1, 0
    return dist.get_world_size(group), dist.get_rank(group)


class ShardBuffer:
    """One rank's results of a sharded pass, as ONE contiguous byte buffer with a section per field.

    fields: [(name, per-item shape tuple, torch dtype), ...].  Section f holds `capacity` items (the largest block of
    any rank), 256-byte aligned; `view(name)` is the typed [capacity, *shape] tensor over it — hand its data_ptr() to
    the engine and the kernel writes straight into the message.
    """

    def __init__(self, fields, capacity, device):
        self.fields = [(n, tuple(s), d) for n, s, d in fields]
        self.capacity = int(capacity)
        self.offsets, off = {}, 0
        for name, shape, dtype in self.fields:
            self.offsets[name] = off
            nbytes = self.capacity * int(np.prod(shape, dtype=np.int64)) * torch.empty((), dtype=dtype).element_size()
            off += (nbytes + 255) & ~255
        self.nbytes = max(off, 256)
        self.raw = torch.zeros(self.nbytes, dtype=torch.uint8, device=device)

    def view(self, name, raw=None):
        raw = self.raw if raw is None else raw
        for n, shape, dtype in self.fields:
            if n == name:
                cnt = self.capacity * int(np.prod(shape, dtype=np.int64))
                es = torch.empty((), dtype=dtype).element_size()
                o = self.offsets[n]
                return raw[o:o + cnt * es].view(dtype).view((self.capacity,) + shape)
        raise KeyError(name)


def gather_packed(buf, n_items, dst=0, group=None, out=None):
    """ONE collective for everything a rank produced: gather (dst = rank) or all_gather (dst = None) of `buf.raw`.

    `dst` (here and everywhere in this package) is a rank of `group`, not a global rank: with a sub-group the two differ, and
    torch.distributed's plain dst= / peer= arguments mean global ranks — hence group_dst= / group_peer= below.
    Returns on the receiving rank(s) a dict name -> [n_items, *shape] tensor in item order (blocks of shard_range(),
    the per-rank padding dropped); None on the others.  `out`: optional preallocated [world, buf.nbytes] uint8 tensor
    on the receiving rank (reused across sweeps).  With one rank nothing is communicated.
    """
    world, rank = _world_rank(group)
    counts = shard_counts(n_items, world)
    if world == 1:
        return {n: buf.view(n)[:counts[0]] for n, _, _ in buf.fields}
    if isinstance(group, EngineGroup):
        if (dst is None or rank == dst) and out is None:
            out = torch.empty((world, buf.nbytes), dtype=torch.uint8, device=buf.raw.device)
        group.before(buf.raw.device)
        group.engine.gather_packed(buf.raw, buf.nbytes, out, dst)
        group.after()
        if dst is not None and rank != dst:
            return None
    elif dst is None:
        if out is None:
            out = torch.empty((world, buf.nbytes), dtype=torch.uint8, device=buf.raw.device)
        dist.all_gather_into_tensor(out.view(-1), buf.raw, group=group)
    else:
        if rank == dst and out is None:
            out = torch.empty((world, buf.nbytes), dtype=torch.uint8, device=buf.raw.device)
        dist.gather(buf.raw, list(out.unbind(0)) if rank == dst else None, group=group, group_dst=dst)   # dst is a rank OF THE GROUP
        if rank != dst:
            return None
    return unpack_gathered(buf, out, counts)


def unpack_gathered(buf, out, counts):
    """Typed item-order tensors over a gathered [world, buf.nbytes] byte matrix (per-rank padding dropped)."""
    res = {}
    for name, _, _ in buf.fields:
        res[name] = torch.cat([buf.view(name, out[r])[:counts[r]] for r in range(len(counts))], dim=0)
    return res


def gather_rows(local, n_items, dst=None, group=None):
    """Collect per-item result rows of ONE field from every rank (kept for single-field callers).

    local : tensor [count_r, ...] holding this rank's block (count_r = shard_range(...)[1]).
    dst   : None -> all_gather (every rank gets the full [n_items, ...] tensor);
            int  -> gather to that rank only (others get None).
    """
    world, rank = _world_rank(group)
    if world == 1:
        return local
    counts = shard_counts(n_items, world)
    assert local.shape[0] == counts[rank], (local.shape, counts, rank)
    buf = ShardBuffer([("x", tuple(local.shape[1:]), local.dtype)], max(counts), local.device)
    buf.view("x")[:counts[rank]] = local
    res = gather_packed(buf, n_items, dst=dst, group=group)
    return None if res is None else res["x"]


def halo_from_left(local, halo, group=None):
    """Rows a rank needs from its LEFT neighbours so that the display accumulators (waterfall: history of 30 post-processed
    rows, persistence: 10 — pyspecsdr.py:130-132,151-154) of its first frames see the frames just before its block.
    `local` may be the rows themselves or, with the batched accumulators (pss_waterfall_rows), just their (lo, hi)
    extremes — 8 bytes per row.

    local : tensor [count_r, ...], this rank's block in frame order.
    Returns a tensor [h, ...] (h = min(halo, rows before this block)): the rows that precede this block in global order.
    Every rank works out from the block sizes which of its rows which rank needs (a block shorter than `halo` means the
    halo spans several left neighbours) and all messages are posted at once (batch_isend_irecv): no rank waits for a
    neighbour's receive before it can send, as a recv -> forward chain would.
    """
    tail = tuple(local.shape[1:])
    empty = torch.empty((0,) + tail, dtype=local.dtype, device=local.device)
    world, rank = _world_rank(group)
    if world == 1 or halo <= 0:
        return empty
    cnt = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    allc = torch.empty(world, dtype=torch.int64, device=local.device)
    if isinstance(group, EngineGroup):
        group.before(local.device)                      # cnt / local come from torch's stream
        group.engine.gather_packed(cnt, 8, allc, None)
        group.after()                                   # the counts are read on the host next
        counts = [int(c) for c in allc.tolist()]
        have = min(int(halo), sum(counts[:rank]))
        got = torch.empty((have,) + tail, dtype=local.dtype, device=local.device)
        rows = local.contiguous()
        row_bytes = rows.element_size() * int(np.prod(tail, dtype=np.int64))
        n_got = group.engine.halo_from_left(rows, counts, row_bytes, int(halo), got)
        group.after()
        assert n_got == have, (n_got, have)
        return got
    dist.all_gather_into_tensor(allc, cnt, group=group)
    counts = [int(c) for c in allc.tolist()]
    starts = [sum(counts[:r]) for r in range(world)]

    def need(r):   # global rows rank r wants: [lo, hi)
        return max(0, starts[r] - halo), starts[r]

    ops, pieces = [], []
    lo, hi = need(rank)
    for s in range(rank):                          # what I receive, left to right
        a, b = max(lo, starts[s]), min(hi, starts[s] + counts[s])
        if b > a:
            t = torch.empty((b - a,) + tail, dtype=local.dtype, device=local.device)
            pieces.append(t)
            ops.append(dist.P2POp(dist.irecv, t, group=group, group_peer=s))   # peers are ranks of the group
    keep = []
    for r in range(rank + 1, world):               # what the ranks to my right need from me
        a, b = need(r)
        a, b = max(a, starts[rank]), min(b, starts[rank] + counts[rank])
        if b > a:
            t = local[a - starts[rank]:b - starts[rank]].contiguous()
            keep.append(t)
            ops.append(dist.P2POp(dist.isend, t, group=group, group_peer=r))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    return torch.cat(pieces, dim=0) if pieces else empty


SCAN_FIELDS = (("peak", (), torch.float32), ("bw", (), torch.float64), ("cnt", (), torch.int32))


def scan_buffer(n_slices, n_fft, gather_db, device, group=None):
    """The packed result buffer of one rank's share of a scanner sweep (sections: [db rows,] peak, bandwidth, count)."""
    world, _ = _world_rank(group)
    fields = ([("db", (n_fft,), torch.float32)] if gather_db else []) + list(SCAN_FIELDS)
    return ShardBuffer(fields, max(shard_counts(n_slices, world)), device)


def sharded_scan(scan_fn, n_slices, n_fft, gather_db=False, dst=0, group=None, device="cpu", buf=None, out=None):
    """Scanner sweep over n_slices centre frequencies (pyspecsdr.py:2514-2590), sharded over the ranks, one collective.

    scan_fn(start, count, views) handles this rank's block: `views` maps "db" (only with gather_db), "peak", "bw",
    "cnt" to the [capacity, ...] sections of the packed buffer; the function either fills views[name][:count] in place
    (Engine.scan writing through data_ptr()) and returns None, or returns (db or None, peak, bw, cnt) tensors, which
    are then copied in.  Returns on rank dst: (db or None, peak, bw, cnt) for all slices in sweep order; None elsewhere.
    """
    world, rank = _world_rank(group)
    start, count = shard_range(n_slices, rank, world)
    if buf is None:
        buf = scan_buffer(n_slices, n_fft, gather_db, device, group)
    views = {n: buf.view(n) for n, _, _ in buf.fields}
    ret = scan_fn(start, count, views)
    if ret is not None:
        db, peak, bw, cnt = ret
        if gather_db and db is not None:
            views["db"][:count] = db
        views["peak"][:count] = peak
        views["bw"][:count] = bw
        views["cnt"][:count] = cnt
    res = gather_packed(buf, n_slices, dst=dst, group=group, out=out)
    if res is None:
        return None
    return res.get("db"), res["peak"], res["bw"], res["cnt"]
