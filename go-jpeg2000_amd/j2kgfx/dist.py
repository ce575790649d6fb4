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


def num_tiles(width, height, tile_w, tile_h):
    tw = tile_w if tile_w > 0 else width
    th = tile_h if tile_h > 0 else height
    return ((width + tw - 1) // tw) * ((height + th - 1) // th)


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
