"""GPU (one device): tile sharding is exact -- the concatenation of every rank's shard output equals
the unsharded frame's output, for any world size; and full-size properties at BASELINE sizes."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def run(plan, frame):
    import torch
    d = torch.from_numpy(frame).to(plan.device)
    torch.cuda.synchronize()
    coeff = plan.forward(d)
    slots, lens, nb = plan.encode_blocks(coeff)
    offs, stream = plan.compact(slots, lens)
    plan.ctx.sync()
    n = int(plan.info.blocks)
    total = int(offs.cpu().numpy()[n])
    return stream.cpu().numpy()[:total].copy(), lens.cpu().numpy()[:n].copy(), nb.cpu().numpy()[:n].copy()


@pytest.mark.parametrize("coder", [0, 1])
@pytest.mark.parametrize("world", [2, 3, 8])
def test_shards_concatenate_to_full(coder, world):
    from j2kgfx import dist as jd
    from j2kgfx.codec import FramePlan
    W, H, Cn, tile = 640, 368, 3, 128
    rng = np.random.default_rng(world)
    frame = rng.integers(0, 256, size=(Cn, H, W)).astype(np.int32)
    kw = dict(precision=8, lossless=True, num_resolutions=4, cb=(32, 32), tile=(tile, tile), coder=coder)
    full = run(FramePlan(W, H, Cn, **kw), frame)
    ntiles = jd.num_tiles(W, H, tile, tile)
    parts = []
    for r in range(world):
        first, count = jd.shard_range(ntiles, r, world)
        if count == 0:
            continue
        parts.append(run(FramePlan(W, H, Cn, tile_first=first, tile_count=count, **kw), frame))
    assert np.array_equal(np.concatenate([p[0] for p in parts]), full[0])
    assert np.array_equal(np.concatenate([p[1] for p in parts]), full[1])
    assert np.array_equal(np.concatenate([p[2] for p in parts]), full[2])


@pytest.mark.parametrize("W,H,Cn,tile,prec", [(3840, 2160, 3, 512, 8), (7680, 4320, 3, 512, 10), (2048, 2048, 1, 0, 16)])
def test_full_size_lossless_roundtrip(W, H, Cn, tile, prec):
    """BASELINE configs C2 / C4 / C5 geometry: size-independent property -- forward then inverse
    returns every pixel (5-3 + RCT are exactly invertible), and HT coding is deterministic."""
    import torch
    from j2kgfx.codec import FramePlan
    rng = np.random.default_rng(W)
    frame = torch.from_numpy(rng.integers(0, 1 << prec, size=(Cn, H, W)).astype(np.int32))
    plan = FramePlan(W, H, Cn, precision=prec, lossless=True, num_resolutions=6, cb=(64, 64), tile=(tile, tile), coder=1)
    d = frame.to(plan.device)
    torch.cuda.synchronize()
    coeff = plan.forward(d)
    back = plan.inverse(coeff)
    s1 = plan.encode_blocks(coeff)
    o1, st1 = plan.compact(s1[0], s1[1])
    plan.ctx.sync()
    assert torch.equal(back.cpu(), frame)
    n = int(plan.info.blocks)
    tot = int(o1[n].item())
    a = st1[:tot].clone()
    s2 = plan.encode_blocks(coeff)
    o2, st2 = plan.compact(s2[0], s2[1])
    plan.ctx.sync()
    assert int(o2[n].item()) == tot and torch.equal(st2[:tot], a)


def _sample_tile_parity(W, H, tile, prec, lossless, quality, coder, sample_tiles, frame, ncomp=3, ctx=None):
    """GPU frame pipeline at full size vs the oracle's per-tile pipeline (encoder.preprocess + encodeTile body,
    encoder.go:198-281, 616-688) on a few sampled tiles: coefficients, block bytes, lengths, numBPS, and the block
    decoders' output for those blocks."""
    import torch
    import oracle as orc
    from j2kgfx.codec import FramePlan
    plan = FramePlan(W, H, ncomp, precision=prec, lossless=lossless, quality=quality, num_resolutions=6, cb=(64, 64),
                     tile=(tile, tile), coder=coder, ctx=ctx)
    tile = tile or max(W, H)
    d = torch.from_numpy(frame).to(plan.device)
    torch.cuda.synchronize()
    coeff = plan.forward(d)
    slots, lens, nb = plan.encode_blocks(coeff)
    offs, stream = plan.compact(slots, lens)
    dec = plan.decode_blocks(stream, offs, lens, nb)
    plan.ctx.sync()
    n = int(plan.info.blocks)
    hco, hl, hn, ho = coeff.cpu().numpy(), lens.cpu().numpy()[:n], nb.cpu().numpy()[:n], offs.cpu().numpy()
    hs, hd = stream.cpu().numpy(), dec.cpu().numpy()
    planes, blocks, doffs = plan.planes(), plan.blocks(), plan.decoded_offsets()
    tiles_x = (W + tile - 1) // tile
    first_block_of_tile = {}
    for j in range(n):
        first_block_of_tile.setdefault(int(blocks[j]["plane"]) // ncomp, j)
    for tl in sample_tiles:
        x0, y0 = (tl % tiles_x) * tile, (tl // tiles_x) * tile
        w, h = min(tile, W - x0), min(tile, H - y0)
        crop = [np.ascontiguousarray(frame[c, y0:y0 + h, x0:x0 + w]) for c in range(ncomp)]
        want_c = orc.preprocess(crop, w, h, prec, lossless, 6, quality)
        for c in range(ncomp):
            row = planes[tl * ncomp + c]
            assert (int(row[0]), int(row[1]), int(row[4]), int(row[5])) == (tl, c, w, h)
            got = hco[int(row[6]):int(row[6]) + w * h].reshape(h, w)
            assert np.array_equal(got, want_c[c]), ("coefficients", tl, c)
        data, wl, wn = orc.encode_tile_blocks(want_c, w, h, 6, 64, 64, coder)
        j0 = first_block_of_tile[tl]
        nj = len(wl)
        assert np.array_equal(hl[j0:j0 + nj], wl.astype(hl.dtype)), ("lens", tl)
        assert np.array_equal(hn[j0:j0 + nj], wn), ("numbps", tl)
        a, b = int(ho[j0]), int(ho[j0 + nj])
        assert np.array_equal(hs[a:b], data), ("bytes", tl)
        pos = 0
        for k in range(0, nj, 7):                       # every 7th block of the tile through the decoders
            j = j0 + k
            bw, bh, band = int(blocks[j]["w"]), int(blocks[j]["h"]), int(blocks[j]["band"])
            p0 = int(ho[j]) - a
            chunk = data[p0:p0 + int(wl[k])]
            want_d = orc.ht_decode(chunk, bw, bh) if coder == 1 else orc.t1_decode(chunk, int(wn[k]), band, bw, bh)
            got_d = hd[int(doffs[j]):int(doffs[j]) + bw * bh].reshape(bh, bw)
            assert np.array_equal(got_d, want_d), ("decoded", tl, k)


def test_c2_full_size_sampled_tiles_match_oracle():
    """BASELINE configs[1] exactly as bench.py runs it (3840x2160 RGB8, 512^2 tiles, 5-3 + HT, 64^2 blocks): the first
    tile, an interior tile, a 256-wide edge tile, a 112-high edge tile and the corner tile against the oracle."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
    import bench
    frame = bench.synth_frame(np, 0)
    _sample_tile_parity(3840, 2160, 512, 8, True, 0, 1, [0, 11, 7, 33, 39], frame)


def test_c3_full_size_sampled_tiles_match_oracle():
    """BASELINE C3 geometry: 3840x2160 RGB rescaled to 12 bit (encoder.go:198-210, v*4095/255), 9-7 + ICT lossy,
    Quality 75, 64^2 blocks, MQ coder (T1).  Quantised coefficients bit-identical, block bytes identical."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
    import bench
    frame = (bench.synth_frame(np, 1).astype(np.int64) * 4095 // 255).astype(np.int32)
    _sample_tile_parity(3840, 2160, 512, 12, False, 75, 0, [0, 39], frame)


def test_c3_full_size_sampled_tiles_match_oracle_throughput_kernels(monkeypatch):
    """The same C3 frame through the kernels `bench.py --config c3` TIMES: with a second MQ-coder context alive
    the library is in its throughput setting -- the encoder's MQ lanes kernel with 64 sorted chains per wavefront -- and
    J2K_T1_DEC_SPLIT=1 sends every block to the one-launch-per-frame lanes decoder (t1_dec_sig_lanes_kernel<true>, a block per lane).
    Coefficients, block bytes, lengths, bit-plane counts and sampled decoded blocks of tiles 0 and 39 against the oracle."""
    import os, sys
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
    import bench
    from j2kgfx import Context
    from j2kgfx.codec import FramePlan
    monkeypatch.setenv("J2K_T1_DEC_SPLIT", "1")
    monkeypatch.setenv("J2K_T1_DEC_LANES", "2")
    other = Context(0)                                                   # a second context that codes with the MQ coder: throughput mode
    op = FramePlan(128, 128, 3, precision=12, lossless=False, quality=75, num_resolutions=3, cb=(64, 64), coder=0, ctx=other)
    x = torch.zeros((3, 128, 128), dtype=torch.int32, device=op.device)
    op.encode_stream(op.forward(x)); other.sync()
    main = Context(0)
    try:
        frame = (bench.synth_frame(np, 1).astype(np.int64) * 4095 // 255).astype(np.int32)
        _sample_tile_parity(3840, 2160, 512, 12, False, 75, 0, [0, 39], frame, ctx=main)
    finally:
        op.close(); other.close(); main.close()


def test_bench_two_rank_control_flow_rehearsal():
    """bench.py's N > 1 step (encode all frames in flight -> one batched gather to rank 0 -> decode while the bytes
    travel; two sets of stream buffers, a set is rewritten only after its gather has finished) with two ranks sharing this box's GPU over gloo (RCCL refuses two ranks
    on one device): control flow only, the JSON line must come out and the round trips inside bench.py must hold."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, J2K_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2",
           "--no-cpu-baseline", "--inflight", "2"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["config"]["frames_in_flight"] == 2 and d["value"] > 0
    assert len([ln for ln in out.stdout.splitlines() if ln.startswith("{")]) == 1          # ONE line, whatever it carries
    # ... and the same command carries the tile-sharded C4 record and the frames-per-rank C5 record
    sr = d["scale_records"]
    assert sr["c4_tiles_sharded"]["value"] > 0 and sr["c4_tiles_sharded"]["scaling"] == "strong" and sr["c4_tiles_sharded"]["tiles"] == 135
    assert sr["c4_tiles_sharded"]["n_gpus"] == 2 and "byte for byte" in sr["c4_tiles_sharded"]["assembly_check"]
    assert sr["c5_frames_per_rank"]["value"] > 0 and sr["c5_frames_per_rank"]["scaling"] == "weak" and sr["c5_frames_per_rank"]["n_gpus"] == 2
    assert d["config"]["ranks_seen"] == 2 and "byte for byte" in d["config"]["gather_check"]
    # the same through bench.py's own launcher: `python bench.py --gpus 2` with no WORLD_SIZE starts the two ranks as a child
    # (before touching the GPU), relays rank 0's line and the status; n_gpus == --gpus
    env2 = dict(env)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env2.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2", "--no-cpu-baseline",
                          "--inflight", "2"], env=env2, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["value"] > 0
    # under a launcher, a world size that is not --gpus is refused
    out = subprocess.run(cmd[:cmd.index("--gpus") + 1] + ["3"] + cmd[cmd.index("--gpus") + 2:], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "WORLD_SIZE=2 but --gpus 3" in out.stderr


def _bench_two_ranks(extra):
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, J2K_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"] + extra
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    return json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])


def test_bench_tile_sharded_two_rank_rehearsal():
    """bench.py --shard tiles with two ranks sharing this box's GPU over gloo: rank r codes its contiguous range of one frame's
    tiles, rank 0 rebuilds rank 1's pack and assembles every tile into a tile-part; bench.py itself asserts that the tile-parts
    name tiles 0..n-1 in order and carry, byte for byte, the stream of an unsharded plan."""
    d = _bench_two_ranks(["--shard", "tiles", "--config", "c2"])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["tiles"] == 40 and d["value"] > 0


def test_bench_frames_per_rank_c5_two_rank_rehearsal():
    """bench.py --config c5 (frames sharded over the ranks, no exchange) with two ranks over gloo."""
    d = _bench_two_ranks(["--config", "c5", "--inflight", "2"])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0


def test_bench_rccl_self_loop_rehearsal():
    """The transfer calls of bench.py's N > 1 step on this box's one GPU: a one-rank RCCL group, the rank sends its packs
    to itself (batched isend / irecv issued from the helper thread), rebuilds them and bench.py asserts the rebuilt
    stream, offsets, lengths and bit-plane counts equal the originals."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, J2K_BENCH_PEER_REHEARSAL="5", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "6", "--warmup", "2", "--no-cpu-baseline"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["value"] > 0
    assert "j2k_gather_streams" in d["config"]["gather"], d["config"]["gather"]     # the probe exchange passed: the C-ABI path ran
    # the probe's fall-back: the same run with the probe made to fail goes through torch.distributed's transfers
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env2 = dict(env, J2K_BENCH_PROBE_FAIL="1", MASTER_PORT=str(port))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "4", "--warmup", "1", "--no-cpu-baseline"],
                         env=env2, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "falling back to torch.distributed" in out.stderr
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["value"] > 0 and "torch.distributed" in d["config"]["gather"], d["config"]["gather"]


def test_c4_full_size_sampled_tiles_match_oracle():
    """BASELINE C4 geometry: 7680x4320 RGB rescaled to 10 bit (v*1023/255), 512^2 tiles (135; last row 224 high),
    5-3 lossless + HT: first, interior, right-edge, bottom-row and corner tiles against the oracle."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
    import bench
    small = bench.synth_frame(np, 4)                                             # 3840x2160 pattern, tiled 2x2 into 8K
    frame = np.tile(small, (1, 2, 2)).astype(np.int64) * 1023 // 255
    frame = frame.astype(np.int32)
    _sample_tile_parity(7680, 4320, 512, 10, True, 0, 1, [0, 52, 14, 125, 134], frame)


def test_c5_full_size_frame_matches_oracle():
    """BASELINE C5 unit: one 2048x2048 16-bit gray frame, untiled, 5-3 lossless (HT coder): the whole frame is one
    tile-component -- coefficients, all 1024 block streams and sampled decoded blocks against the oracle."""
    rng = np.random.default_rng(55)
    yy, xx = np.mgrid[0:2048, 0:2048]
    frame = np.clip((xx * 65535 // 2048 + yy * 65535 // 2048) // 2 + rng.integers(-2000, 2001, (2048, 2048)), 0, 65535).astype(np.int32)[None]
    _sample_tile_parity(2048, 2048, 0, 16, True, 0, 1, [0], frame, ncomp=1)


@pytest.mark.parametrize("W,H,tile,first,count", [(1280, 624, 512, 0, 0), (1280, 624, 512, 2, 3), (200, 96, 64, 1, 4), (512, 512, 0, 0, 0), (3840, 2160, 512, 17, 9)])
def test_device_tile_part_assembly_equals_host(W, H, tile, first, count, oracle):
    """j2k_plan_assemble_tiles_device: the tile-parts of a shard built on the device are byte for byte (a) the ORACLE's
    encoder.createTileHeader (orc_create_tile_header, encoder.go:746-760) applied tile by tile to the ORACLE's block bytes of that
    tile, and (b) what the host call (j2k_assemble_tiles) builds from the same stream; the reference's tile-part parser reads
    them back"""
    import torch
    from j2kgfx import codestream
    from j2kgfx.codec import FramePlan
    rng = np.random.default_rng(W + first)
    frame_h = rng.integers(0, 256, (3, H, W)).astype(np.int32)
    plan = FramePlan(W, H, 3, precision=8, lossless=True, num_resolutions=4, cb=(32, 32), tile=(tile, tile), coder=1,
                     tile_first=first, tile_count=count)
    frame = torch.from_numpy(frame_h).to(plan.device)
    stream, offs, lens, nb = plan.encode_stream(plan.forward(frame))
    out, out_len = plan.assemble_tiles(stream, offs)
    plan.ctx.sync()
    n = int(plan.info.blocks)
    o_h = offs.cpu().numpy()
    tiles_of = plan.planes()[:, 0][plan.blocks()["plane"]]
    starts = np.flatnonzero(np.diff(np.concatenate([[-1], tiles_of]))).tolist() + [n]
    t_offs = np.array([int(o_h[j]) for j in starts], dtype=np.uint64)
    want = codestream.assemble_tiles(stream[:int(o_h[n])].cpu().numpy(), t_offs, tile_first=first)
    got = out[:int(out_len[0].item())].cpu().numpy().tobytes()
    assert got == want
    # straight against the oracle, no product code on the expected side: per tile of the shard, preprocess -> block bytes ->
    # createTileHeader
    tw = tile or W
    tx_n = (W + tw - 1) // tw
    th = tile or H
    pos = 0
    for t in range(first, first + len(starts) - 1):
        tx, ty = t % tx_n, t // tx_n
        x0, y0 = tx * tw, ty * th
        w_t, h_t = min(tw, W - x0), min(th, H - y0)
        comps = [np.ascontiguousarray(frame_h[c, y0:y0 + h_t, x0:x0 + w_t]) for c in range(3)]
        coeff = oracle.preprocess(comps, w_t, h_t, 8, True, 4)
        tile_bytes, _, _ = oracle.encode_tile_blocks(coeff, w_t, h_t, 4, 32, 32, 1)
        part = oracle.create_tile_header(t, bytes(tile_bytes))
        assert got[pos:pos + len(part)] == part, "tile-part %d differs from the oracle" % t
        pos += len(part)
    assert pos == len(got)
    parts = codestream.parse_tile_parts(got)
    assert [p.TileIndex for p, _ in parts] == list(range(first, first + len(starts) - 1))


def test_c_abi_gather_streams_self_loop():
    """j2k_comm_* / j2k_gather_streams on this box's one GPU: a one-rank RCCL communicator made through the C ABI, the packs of
    three frame slots (j2k_plan_pack_stream) sent to the rank itself with ncclSend / ncclRecv in one group on the
    communicator's stream -- once with the byte counts given by the host, once gathered by ncclAllGather inside the call --
    then rebuilt with j2k_plan_unpack_streams on a context that waits for the transfer on the device (j2k_comm_wait): stream,
    offsets, lengths and bit-plane counts byte for byte those of the encoder."""
    import torch
    from j2kgfx import Context
    from j2kgfx.codec import FramePlan
    from j2kgfx import dist as jdist
    W, H = 1280, 624
    rng = np.random.default_rng(77)
    ctxs = [Context(0) for _ in range(3)]
    kw = dict(precision=8, lossless=True, num_resolutions=6, cb=(64, 64), tile=(512, 512), coder=1, tile_first=1, tile_count=4)
    plans = [FramePlan(W, H, 3, ctx=c, **kw) for c in ctxs]
    root_ctx = Context(0)
    root_plan = FramePlan(W, H, 3, ctx=root_ctx, **kw)
    comm = jdist.Comm(root_ctx, 0, 1)
    for given in (True, False, True):
        enc, packs = [], []
        for p in plans:
            frame = torch.from_numpy(rng.integers(0, 256, (3, H, W)).astype(np.int32)).to(p.device)
            st = p.encode_stream(p.forward(frame))
            enc.append(st)
            packs.append(p.pack_stream(*st))
        for c in ctxs:
            c.sync()                                            # (the host needs the pack sizes: first int64 of each pack)
        nbytes = [int(pk[:8].view(torch.int64).item()) for pk in packs]
        assert all(0 < n <= pk.numel() for n, pk in zip(nbytes, packs))
        recv = torch.full((sum((n + 15) & ~15 for n in nbytes) + 64,), 0xEE, dtype=torch.uint8, device=root_plan.device)
        offs = comm.gather(packs, nbytes, recv=recv, producers=ctxs, all_bytes=[nbytes] if given else None, self_loop=True)
        assert len(offs) == 4 and int(offs[0]) == 0 and all(int(o) % 16 == 0 for o in offs)
        comm.wait(root_ctx)
        n = int(root_plan.info.blocks)
        outs = [(root_plan.empty(root_plan.info.bytes_cap, torch.uint8), root_plan.empty(n + 1, torch.int64), root_plan.empty(n, torch.int32),
                 root_plan.empty(n, torch.uint8)) for _ in packs]
        root_plan.unpack_streams([recv[int(offs[f]):int(offs[f]) + nbytes[f]] for f in range(3)], outs)
        root_ctx.sync()
        assert int(recv[int(offs[3]):].min().item()) == 0xEE     # nothing written past the gathered bytes
        for (s0, o0, l0, b0), (s1, o1, l1, b1) in zip(enc, outs):
            tot = int(o0[n].item())
            assert torch.equal(o1[:n + 1], o0[:n + 1]) and torch.equal(l1[:n], l0[:n]) and torch.equal(b1[:n], b0[:n])
            assert torch.equal(s1[:tot], s0[:tot])
    # a receive buffer that is too small is refused -- collectively: the sends are still posted and rank 0 drains them into a
    # scratch buffer (here the lone rank's own), so nobody is left waiting and the communicator stays usable
    from j2kgfx import J2KError, _lib as jl
    for given in (True, False):
        with pytest.raises(J2KError) as ei:
            comm.gather(packs, nbytes, recv=recv[:1024], producers=ctxs, all_bytes=[nbytes] if given else None, self_loop=True)
        assert ei.value.status == jl.ERR_CAPACITY
        comm.wait()                                              # the host-side wait returns: nothing is stuck on the stream
    recv.fill_(0xEE)
    offs = comm.gather(packs, nbytes, recv=recv, producers=ctxs, all_bytes=None, self_loop=True)
    comm.wait()
    for f in range(3):
        assert torch.equal(recv[int(offs[f]):int(offs[f]) + nbytes[f]], packs[f][:nbytes[f]])
    # lifetime: a communicator is closed with (before) its context and refuses use afterwards
    root_plan.close()
    root_ctx.close()
    assert comm.h is None
    with pytest.raises(J2KError):
        comm.gather(packs, nbytes, recv=recv, producers=ctxs, self_loop=True)
    with pytest.raises(J2KError):
        comm.wait()
    comm.close()
