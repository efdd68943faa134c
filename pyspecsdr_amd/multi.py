"""GPU-side multi-rank drivers: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI), each rank an
Engine on its own device, the exchange step overlapped with the next batch's compute.

ShardedScanner — BASELINE.json configs[3]: the scanner sweep (pyspecsdr.py:2514-2590: every centre frequency is an
independent slice) split into contiguous blocks of slices per rank; Engine.scan writes its results straight into the
rank's packed message (shard.ShardBuffer), ONE gather per sweep moves it to the root on a side stream while the next
sweep computes into the other buffer set.

The library itself knows nothing about torch or RCCL (plain C ABI); this module is the host-side glue, in the
reference's own language.

sharded_stream_display — BASELINE.json configs[4]: a long capture cut into frames, contiguous blocks of frames per rank,
every rank streaming ITS block from its own pinned host memory over its own PCIe link (Engine.stream_display_nfm: copy /
compute / copy streams, two buffer sets) and returning the display lines + PCM of its frames.  The display history of a
block's first frames reaches into the left neighbour's block: the per-row extremes (8 bytes per row) are exchanged after
the pass and the first window-1 frames of every block are redone with that halo (a 29-frame fix-up instead of making
every rank wait for its neighbour's last chunk).
"""
import torch
import torch.distributed as dist

import numpy as np

from .shard import EngineGroup, ShardBuffer, gather_packed, halo_from_left, scan_buffer, shard_counts, shard_range, unpack_gathered


def _world_rank(group=None):
    if isinstance(group, EngineGroup):       # exchange steps behind the C ABI: the communicator the Engine joined (no torch.distributed involved)
        return group.world, group.rank
    if not dist.is_initialized():
        return 1, 0
    return dist.get_world_size(group), dist.get_rank(group)


class ShardedScanner:
    """Sweep of n_slices scanner slices of n_fft points, sharded over the ranks of `group`.

    gather_db=True moves every rank's float32 dB rows (n_fft * 4 bytes per slice) + (peak, bandwidth, count) to rank
    `dst`; gather_db=False only the 16 bytes per slice of (peak, bandwidth, count) — the dB rows stay sharded on the
    ranks that computed them (`local_db(handle)`).
    """

    def __init__(self, engine, n_slices, n_fft, fs, gather_db=True, dst=0, group=None, device=None):
        self.eng, self.n_slices, self.n_fft, self.fs = engine, int(n_slices), int(n_fft), float(fs)
        self.gather_db, self.dst, self.group = gather_db, dst, group
        self.world, self.rank = _world_rank(group)
        self.device = torch.device("cuda", engine.device) if device is None else torch.device(device)
        self.start, self.count = shard_range(self.n_slices, self.rank, self.world)
        self.counts = shard_counts(self.n_slices, self.world)
        # two buffer sets: sweep k+1 computes into one while sweep k's is in flight
        self.bufs = [scan_buffer(self.n_slices, self.n_fft, gather_db, self.device, group) for _ in range(2)]
        self.db_local = None if gather_db else [torch.empty((max(self.counts), self.n_fft), dtype=torch.float32, device=self.device)
                                                for _ in range(2)]
        # gloo (tests on a box with fewer GPUs than ranks) cannot gather device tensors: stage through the host
        # an initialised process group means "exchange", also with a single rank (exercises the RCCL path on one GPU)
        if isinstance(group, EngineGroup):
            raise TypeError("ShardedScanner overlaps its gather on a torch.distributed side stream; an EngineGroup (C-ABI exchange) serves the "
                            "blocking helpers: gather_packed, halo_from_left, sharded_stream_display (or call pss_gather_packed directly, "
                            "examples/pss_sweep_ranks.c)")
        self.exchange = dist.is_initialized()
        self.host_stage = self.exchange and dist.get_backend(group) == "gloo"
        self.comp = torch.cuda.ExternalStream(engine.stream_handle(), device=self.device)
        self.comm = torch.cuda.Stream(device=self.device)
        self.recv = [None, None]
        if self.rank == dst and self.exchange:
            dev = "cpu" if self.host_stage else self.device
            self.recv = [torch.empty((self.world, self.bufs[0].nbytes), dtype=torch.uint8, device=dev) for _ in range(2)]
        self.sent = [None, None]     # event on the comm stream: set b's message has left (the set may be rewritten)
        self.k = 0

    def sweep(self, d_iq_local):
        """Queue one sweep over this rank's block (d_iq_local: interleaved complex64 [count][n_fft] on this rank's GPU)
        and its gather.  Asynchronous; returns a handle for result().

        Ordering against the caller: everything queued so far on torch's CURRENT stream — the producer of d_iq_local, and the
        consumers of an earlier result() (its tensors are views / concatenations of the buffer set this sweep is about to reuse)
        — finishes before this sweep's scan and gather start."""
        b = self.k & 1
        self.k += 1
        buf = self.bufs[b]
        cur = torch.cuda.current_stream(self.device)
        self.comp.wait_stream(cur)
        self.comm.wait_stream(cur)
        if self.sent[b] is not None:
            self.comp.wait_event(self.sent[b])
        db = buf.view("db") if self.gather_db else self.db_local[b]
        if self.count:
            self.eng.scan(d_iq_local, self.count, self.n_fft, self.fs, db, buf.view("peak"), buf.view("bw"), buf.view("cnt"))
        if self.exchange:
            self.comm.wait_stream(self.comp)
            with torch.cuda.stream(self.comm):
                if self.host_stage:
                    h = buf.raw.cpu()            # synchronous on the comm stream
                    dist.gather(h, list(self.recv[b].unbind(0)) if self.rank == self.dst else None, group=self.group, group_dst=self.dst)
                else:
                    dist.gather(buf.raw, list(self.recv[b].unbind(0)) if self.rank == self.dst else None, group=self.group,
                                group_dst=self.dst)
                self.sent[b] = self.comm.record_event()
        return b

    def wait(self, handle):
        """Block the host until sweep `handle` (compute + gather) has finished."""
        self.eng.sync()
        if self.sent[handle] is not None:
            self.sent[handle].synchronize()

    def result(self, handle):
        """(db or None, peak, bw, cnt) of the whole sweep in slice order on rank dst; None on the other ranks."""
        self.wait(handle)
        buf = self.bufs[handle]
        if not self.exchange:
            res = gather_packed(buf, self.n_slices, dst=self.dst, group=self.group)
        elif self.rank != self.dst:
            return None
        else:
            res = unpack_gathered(buf, self.recv[handle], self.counts)
        return res.get("db"), res["peak"], res["bw"], res["cnt"]

    def local_db(self, handle):
        """This rank's dB rows [count][n_fft] of sweep `handle` (valid until the set is reused two sweeps later)."""
        db = self.bufs[handle].view("db") if self.gather_db else self.db_local[handle]
        return db[:self.count]


def sharded_stream_display(engine, h_iq_local, fs, chunk_frames, mode="waterfall", window=None, disp_h=36, disp_w=112, group=None,
                           gather_dst=None, out=None, rows="f32"):
    """Every rank streams its block of a capture (h_iq_local: complex64 [count_r][n], pinned) and ends up with the display
    lines + PCM of its frames, exactly as one rank streaming the whole capture would produce them.

    out: a result dict of Engine.stream_display_nfm to write into (arrays from Engine.pinned_empty keep the downloads asynchronous).
    rows: "f32" (Engine.stream_display_nfm) or "f64" (Engine.stream_display_nfm_f64: compute_fft's own float64 rows from the transform to the
    cells — the reference's cells; the halo of row extremes is float64 then: 16 bytes per row).
    gather_dst=None: results stay on the ranks (each writes its part of the FIFO / screen log); an int gathers
    (lines..., pcm) to that rank over the process group (one packed message per rank) and returns them in frame order
    there, None elsewhere.  Returns (lines tuple, pcm) of this rank's block otherwise.
    """
    world, rank = _world_rank(group)
    window = (30 if mode == "waterfall" else 10) if window is None else int(window)
    stream = engine.stream_display_nfm_f64 if rows == "f64" else engine.stream_display_nfm
    res = stream(h_iq_local, fs, chunk_frames, mode=mode, window=window, disp_h=disp_h, disp_w=disp_w, out=out)
    lines, pcm = res["lines"], res["pcm"]
    if world > 1:
        # extremes of the rows preceding this block: 8 bytes per row from the left neighbour(s), all messages posted at once
        dev = "cpu" if _backend(group) == "gloo" else torch.device("cuda", engine.device)
        ext = torch.from_numpy(np.stack([res["row_lo"], res["row_hi"]], axis=1)).to(dev)
        halo = halo_from_left(ext, window - 1, group=group).cpu().numpy()
        fix = min(window - 1, h_iq_local.shape[0])
        if len(halo) and fix:
            # only the first window-1 frames of the block see the halo; redo them with it (their history is complete now)
            r2 = stream(np.ascontiguousarray(h_iq_local[:fix]), fs, fix, mode=mode, window=window, disp_h=disp_h,
                        disp_w=disp_w, halo=(halo[:, 0], halo[:, 1]))
            for dst_a, src_a in zip(lines, r2["lines"]):
                dst_a[:fix] = src_a
    if gather_dst is None or world == 1:
        return lines, pcm
    # one packed message per rank: [lines... | pcm]
    dev = "cpu" if _backend(group) == "gloo" else torch.device("cuda", engine.device)
    counts = shard_counts_from(torch.tensor([h_iq_local.shape[0]], dtype=torch.int64), group, device=dev)
    fields = [(f"l{i}", (disp_w,), torch.int8) for i in range(len(lines))] + [("pcm", tuple(pcm.shape[1:]), torch.int16)]
    buf = ShardBuffer(fields, max(counts), dev)
    for i, a in enumerate(lines):
        buf.view(f"l{i}")[:a.shape[0]] = torch.from_numpy(a).to(dev)
    buf.view("pcm")[:pcm.shape[0]] = torch.from_numpy(pcm).to(dev)
    got = _gather_uneven(buf, counts, gather_dst, group)
    if got is None:
        return None
    return tuple(got[f"l{i}"].cpu().numpy() for i in range(len(lines))), got["pcm"].cpu().numpy()


def _backend(group):
    """"engine" for an EngineGroup (exchange steps behind the C ABI: device buffers, the engine's stream), else the process group's backend."""
    return "engine" if isinstance(group, EngineGroup) else dist.get_backend(group)


def shard_counts_from(count_tensor, group=None, device=None):
    """All ranks' block sizes (blocks need not come from shard_range: a capture is cut where the caller cut it).
    device: where the collective's buffers live (the engine's GPU for RCCL; default: the tensor's own device, or the host under gloo)."""
    world, _ = _world_rank(group)
    if world == 1:
        return [int(count_tensor.item())]
    dev = "cpu" if _backend(group) == "gloo" else (device if device is not None else count_tensor.device)
    allc = torch.empty(world, dtype=torch.int64, device=dev)
    if isinstance(group, EngineGroup):
        cnt = count_tensor.to(dev)
        group.before(dev)
        group.engine.gather_packed(cnt, 8, allc, None)
        group.after()
    else:
        dist.all_gather_into_tensor(allc, count_tensor.to(dev), group=group)
    return [int(c) for c in allc.tolist()]


def _gather_uneven(buf, counts, dst, group):
    world, rank = _world_rank(group)
    out = torch.empty((world, buf.nbytes), dtype=torch.uint8, device=buf.raw.device) if rank == dst else None
    if isinstance(group, EngineGroup):
        group.before(buf.raw.device)
        group.engine.gather_packed(buf.raw, buf.nbytes, out, dst)
        group.after()
    else:
        dist.gather(buf.raw, list(out.unbind(0)) if rank == dst else None, group=group, group_dst=dst)
    if rank != dst:
        return None
    return unpack_gathered(buf, out, counts)
