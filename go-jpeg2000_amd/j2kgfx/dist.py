"""Multi-GPU layer: one process per GPU (torch.distributed, backend "nccl" == RCCL over xGMI
on ROCm; "gloo" on CPU for tests).

The hot path shards with no halo: tile t (or frame f in throughput mode) is an independent
unit.  The only exchange step is the gather of the variable-length compressed
block streams to rank 0 for codestream assembly (encoder.generateTiles / createTileHeader stay
on the Go side, encoder.go:568-579, 746-760):

    all_gather(per-rank byte totals)            -- tiny, fixed size
    peers: isend(stream[:total]) -> rank 0      -- direct peer->root, each over its own xGMI link
    root : irecv into the output buffer at the exclusive-scan offsets

No ring: on an 8-GPU MI355X node every peer has its own link to rank 0.
"""
import numpy as np


def shard_range(n_units, rank, world):
    """Contiguous balanced partition of n_units over `world` ranks: (first, count) for `rank`.
    The first n_units % world ranks get one extra unit; rank 0's share comes first."""
    base, extra = divmod(int(n_units), int(world))
    first = rank * base + min(rank, extra)
    return first, base + (1 if rank < extra else 0)


def num_tiles(width, height, tile_w, tile_h, frame_rows=0):
    """Tiles of a plan with these j2k_params, by build_plan's arithmetic (csrc/j2k_planbuild.cpp): with frame_rows > 0 the plan is a
    BATCH of height / frame_rows frames and the tile grid starts again at every frame (a partial last tile row per frame;
    tile_h = 0: one tile row per frame, not per batch)."""
    fh = frame_rows if frame_rows > 0 else height
    if fh <= 0 or height % fh:
        raise ValueError("height is not a whole number of frames (frame_rows)")
    tw = tile_w if tile_w > 0 else width
    th = tile_h if tile_h > 0 else fh
    return ((width + tw - 1) // tw) * ((fh + th - 1) // th) * (height // fh)


def gather_streams(stream, nbytes, group=None, out=None):
    """Gather every rank's first `nbytes` bytes of its uint8 tensor `stream` to rank 0.

    Returns (buffer, offsets) on rank 0 -- offsets has world+1 entries, rank r's bytes are
    buffer[offsets[r]:offsets[r+1]] -- and (None, offsets) elsewhere.  Works on any backend
    (tensors must live where the backend wants them: cuda for nccl, cpu for gloo)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = stream.device
    mine = torch.tensor([int(nbytes)], dtype=torch.int64, device=dev)
    totals = torch.empty(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(totals, mine, group=group)
    totals_h = totals.cpu().numpy()
    offsets = np.concatenate(([0], np.cumsum(totals_h))).astype(np.int64)
    if world == 1:
        return stream[:int(nbytes)], offsets
    if rank == 0:
        total = int(offsets[-1])
        if out is None or out.numel() < total:
            out = torch.empty(max(total, 1), dtype=torch.uint8, device=dev)
        out[:int(nbytes)].copy_(stream[:int(nbytes)])
        ops = []
        for r in range(1, world):
            n = int(totals_h[r])
            if n:
                ops.append(dist.P2POp(dist.irecv, out[int(offsets[r]):int(offsets[r]) + n], r, group))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        return out, offsets
    if int(nbytes):
        for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, stream[:int(nbytes)], 0, group)]):
            w.wait()
    return None, offsets


class StreamGather:
    """Handle of gather_streams_start: the transfers are in flight until wait()."""

    def __init__(self, works, results):
        self._works, self._results = works, results

    def wait(self):
        """Completes the transfers (nccl: the current stream waits for them; gloo: blocks) and returns, per item,
        (buffer, offsets) on rank 0 and (None, offsets) elsewhere -- as gather_streams does."""
        for w in self._works:
            w.wait()
        self._works = []
        return self._results


def gather_streams_start(items, group=None, outs=None, size_group=None, self_loop=False):
    """Gather SEVERAL streams to rank 0 with one size exchange and one batch of point-to-point transfers, without
    waiting for the bytes: `items` = [(uint8 tensor, nbytes), ...] (e.g. one per frame in flight), `outs` = optional
    list of reusable receive buffers on rank 0.  The caller overlaps other work and then calls .wait().
    The source tensors (and `outs`) must not be rewritten before .wait() has returned and, on nccl, before the
    streams that rewrite them have been made to wait for the current stream.
    size_group: a gloo group of the same ranks for the size exchange.  The sizes are host integers on both ends; sent
    through RCCL they would cost a kernel that queues behind everything the GPU is busy with while the host waits.
    self_loop (dev, one rank): the lone rank sends every stream to itself, to exercise the transfer calls on one GPU."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    F = len(items)
    dev = items[0][0].device
    if size_group is not None:
        mine = torch.tensor([int(n) for _, n in items], dtype=torch.int64)
        totals = torch.empty(world * F, dtype=torch.int64)
        dist.all_gather_into_tensor(totals, mine, group=size_group)
        totals_h = totals.numpy().reshape(world, F)
    else:
        mine = torch.tensor([int(n) for _, n in items], dtype=torch.int64, device=dev)
        totals = torch.empty(world * F, dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(totals, mine, group=group)
        totals_h = totals.cpu().numpy().reshape(world, F)
    results, ops = [], []
    outs = list(outs) if outs is not None else [None] * F
    for f, (stream, nbytes) in enumerate(items):
        offsets = np.concatenate(([0], np.cumsum(totals_h[:, f]))).astype(np.int64)
        if world == 1 and self_loop:
            out = outs[f]
            if out is None or out.numel() < int(nbytes):
                out = torch.empty(max(int(nbytes), 1), dtype=torch.uint8, device=dev)
            if int(nbytes):
                ops.append(dist.P2POp(dist.irecv, out[:int(nbytes)], 0, group))
                ops.append(dist.P2POp(dist.isend, stream[:int(nbytes)], 0, group))
            results.append((out, offsets))
            continue
        if world == 1:
            results.append((stream[:int(nbytes)], offsets))
            continue
        if rank == 0:
            total = int(offsets[-1])
            out = outs[f]
            if out is None or out.numel() < total:
                out = torch.empty(max(total, 1), dtype=torch.uint8, device=dev)
            out[:int(nbytes)].copy_(stream[:int(nbytes)])
            for r in range(1, world):
                n = int(totals_h[r, f])
                if n:
                    ops.append(dist.P2POp(dist.irecv, out[int(offsets[r]):int(offsets[r]) + n], r, group))
            results.append((out, offsets))
        else:
            if int(nbytes):
                ops.append(dist.P2POp(dist.isend, stream[:int(nbytes)], 0, group))
            results.append((None, offsets))
    works = dist.batch_isend_irecv(ops) if ops else []
    return StreamGather(works, results)


# ---- the same exchange at the C ABI (include/j2kgfx.h: j2k_comm_*, j2k_gather_streams) -----------------------------
class Comm:
    """j2k_comm: an RCCL communicator owned by libj2kgfx.so (its own HIP stream for the transfers).  A thin caller: the
    128-byte id is made on rank 0 (j2k_comm_get_unique_id) and carried to the other ranks by whatever the host has -- here
    torch.distributed's object broadcast on `group` (any backend), or nothing for a one-rank communicator."""

    def __init__(self, ctx, rank=0, world=1, group=None, id_bytes=None):
        import ctypes as C
        from . import _lib
        self.ctx, self.rank, self.world = ctx, int(rank), int(world)
        L = ctx.L
        if id_bytes is None:
            buf = (C.c_uint8 * 128)()
            if self.rank == 0:
                st = L.j2k_comm_get_unique_id(buf)
                if st != _lib.OK:
                    raise _lib.J2KError(st, "j2k_comm_get_unique_id: %s: %s" % (L.j2k_status_string(st).decode(),
                                                                              L.j2k_comm_load_error().decode()))
            id_bytes = bytes(buf)
            if self.world > 1:
                import torch.distributed as dist
                box = [id_bytes]
                dist.broadcast_object_list(box, src=0, group=group)
                id_bytes = box[0]
        h = C.c_void_p()
        idb = (C.c_uint8 * 128).from_buffer_copy(id_bytes)
        ctx.check(L.j2k_comm_create(ctx.h, idb, self.rank, self.world, C.byref(h)))
        self.h = h
        # a j2k_comm keeps a pointer to its j2k_ctx (error text): it is closed with -- before -- its context, and first of all
        # by the atexit hook, exactly like a Graph (context.py)
        from . import context as _context
        ctx._comms.add(self)
        _context._live_comms.add(self)

    def _alive(self):
        from . import _lib
        if self.h is None or self.ctx.h is None:
            raise _lib.J2KError(_lib.ERR_INVALID_ARG, "communicator used after it or its context was closed")

    def gather(self, sends, nbytes, recv=None, producers=(), all_bytes=None, self_loop=False):
        """j2k_gather_streams: sends = list of device uint8 tensors, nbytes = their byte counts (host ints), recv = rank 0's
        device uint8 buffer, producers = the contexts whose streams are producing `sends`, all_bytes = every rank's counts
        (world x count) if the host already has them.  Returns the offsets array (world * count + 1): stream f of rank r is
        recv[offs[r * count + f] : ... + its byte count] on rank 0.  The transfers are queued, not finished: wait()."""
        import ctypes as C
        import numpy as np
        self._alive()
        k = len(sends)
        VP = C.c_void_p * k
        ptrs = VP(*[int(t.data_ptr()) for t in sends])
        nb = (C.c_uint64 * k)(*[int(n) for n in nbytes])
        offs = (C.c_uint64 * (self.world * k + 1))()
        ab = None
        if all_bytes is not None:
            flat = [int(v) for v in np.asarray(all_bytes).reshape(-1)]
            assert len(flat) == self.world * k
            ab = (C.c_uint64 * len(flat))(*flat)
        prod = (C.c_void_p * max(len(producers), 1))(*[p.h.value for p in producers]) if producers else None
        self.ctx.check(self.ctx.L.j2k_gather_streams(self.h, k, ptrs, nb, ab, prod, len(producers),
                                                     C.c_void_p(int(recv.data_ptr())) if recv is not None else None,
                                                     C.c_size_t(int(recv.numel()) if recv is not None else 0), offs,
                                                     1 if self_loop else 0))
        self._keep = (sends, recv)          # alive until the next gather / wait
        return np.array(offs[:], dtype=np.uint64)

    def wait(self, consumer=None):
        """consumer = a Context whose stream must see the gathered bytes (device-side wait), or None: the host waits."""
        self._alive()
        if consumer is not None and consumer.h is None:
            from . import _lib
            raise _lib.J2KError(_lib.ERR_INVALID_ARG, "consumer context is closed")
        self.ctx.check(self.ctx.L.j2k_comm_wait(self.h, consumer.h if consumer is not None else None))

    def close(self):
        # j2k_comm_destroy does not dereference the context (it keeps the device number), so a communicator that outlived
        # its context -- it should not: Context.close() closes its communicators first -- is still destroyed, not leaked
        if self.h:
            self.ctx.L.j2k_comm_destroy(self.h)
        self.h = None
        self._keep = None

    def __del__(self):
        import sys
        try:
            if not sys.is_finalizing():
                self.close()
        except Exception:
            pass
