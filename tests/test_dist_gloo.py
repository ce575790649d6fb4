"""CPU: the N>1 path -- tile/frame sharding and the variable-length gather of compressed streams to
rank 0 -- with two gloo ranks (the same code runs over RCCL/xGMI on the GPU node)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "go-jpeg2000_amd"))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, lens, q):
    sys.path.insert(0, os.path.join(ROOT, "go-jpeg2000_amd"))
    from j2kgfx import dist as jd
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        for trial, n in enumerate(lens[rank]):
            rng = np.random.default_rng(1000 * rank + trial)
            payload = rng.integers(0, 256, n + 37).astype(np.uint8)       # capacity > nbytes, like the slot stream
            buf, offs = jd.gather_streams(torch.from_numpy(payload), n)
            if rank == 0:
                got = buf[:int(offs[-1])].numpy().copy()
                q.put((trial, got, offs.copy()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_gather_streams_two_ranks():
    world = 2
    lens = [[1000, 0, 5, 123457], [17, 2048, 0, 99]]                     # includes empty contributions
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, lens, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(len(lens[0]))]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for trial, got, offs in results:
        want = []
        for r in range(world):
            rng = np.random.default_rng(1000 * r + trial)
            want.append(rng.integers(0, 256, lens[r][trial] + 37).astype(np.uint8)[:lens[r][trial]])
        assert offs.tolist() == [0, lens[0][trial], lens[0][trial] + lens[1][trial]]
        assert np.array_equal(got, np.concatenate(want))


def _worker_batched(rank, world, port, lens, q):
    sys.path.insert(0, os.path.join(ROOT, "go-jpeg2000_amd"))
    from j2kgfx import dist as jd
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        outs = None
        side = dist.new_group(backend="gloo")                             # bench.py: the sizes travel through a side group,
        for step in range(2):                                             # second step reuses the receive buffers
            items = []
            for f, n in enumerate(lens[rank]):
                rng = np.random.default_rng(1000 * rank + 10 * step + f)
                items.append((torch.from_numpy(rng.integers(0, 256, n + 5).astype(np.uint8)), n))
            if step == 0:
                g = jd.gather_streams_start(items, outs=outs)
            else:                                                         # ... and the exchange runs on a helper thread
                import threading
                box = []
                th = threading.Thread(target=lambda: box.append(jd.gather_streams_start(items, outs=outs, size_group=side)))
                th.start(); th.join()
                g = box[0]
            res = g.wait()
            if rank == 0:
                outs = [b for b, _ in res]
                q.put((step, [(b[:int(o[-1])].numpy().copy(), o.copy()) for b, o in res]))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_gather_streams_start_batches_frames_in_flight():
    """bench.py's N>1 step: the streams of all frames in flight go to rank 0 with one size exchange and one batch of
    point-to-point transfers (gather_streams_start / wait)."""
    world = 2
    lens = [[1000, 0, 70001], [17, 4096, 3]]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_batched, args=(r, world, port, lens, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for step, per_frame in results:
        for f, (got, offs) in enumerate(per_frame):
            want = []
            for r in range(world):
                rng = np.random.default_rng(1000 * r + 10 * step + f)
                want.append(rng.integers(0, 256, lens[r][f] + 5).astype(np.uint8)[:lens[r][f]])
            assert offs.tolist() == [0, lens[0][f], lens[0][f] + lens[1][f]]
            assert np.array_equal(got, np.concatenate(want))


def test_shard_range_partitions_exactly():
    from j2kgfx import dist as jd
    for n in (0, 1, 5, 40, 135, 256):
        for world in (1, 2, 3, 4, 8):
            spans = [jd.shard_range(n, r, world) for r in range(world)]
            covered = [i for (f, c) in spans for i in range(f, f + c)]
            assert covered == list(range(n))
            counts = [c for _, c in spans]
            assert max(counts) - min(counts) <= 1
    assert jd.num_tiles(3840, 2160, 512, 512) == 40                      # C2: 8 x 5
    assert jd.num_tiles(7680, 4320, 512, 512) == 135                     # C4: 15 x 9
    assert jd.num_tiles(512, 512, 0, 0) == 1
    # batches (j2k_params.frame_rows): the tile grid starts again at every frame -- a partial last tile row PER FRAME, and
    # tile_h = 0 is one tile per frame (the plan's arithmetic, not ceil(H_total / tile_h))
    assert jd.num_tiles(1024, 4 * 600, 512, 512, frame_rows=600) == 4 * 2 * 2
    assert jd.num_tiles(1024, 4 * 600, 0, 0, frame_rows=600) == 4
    assert jd.num_tiles(2048, 4 * 2048, 0, 0, frame_rows=2048) == 4      # C5 batches in bench.py
    with pytest.raises(ValueError):
        jd.num_tiles(64, 100, 0, 0, frame_rows=30)
