"""GPU: the decode body.

(i)   PacketDecoder.DecodePacket on a device buffer (csrc/t2dec.hip, j2k_t2_decode_packets_device) == the host call
      (csrc/t2.cpp, j2k_t2_decode_packet) == the restatement of internal/tcd/t2.go:463-652 (oracle/t2ref.py), on the random runs
      tests/test_gpu_t2.py draws for the encoder -- the reference's decoder AS IT IS (its header reader runs on its own,
      Position() only moves over markers and bodies), so most of these runs end in its error return: packets done, every
      decoded field, every body and Position() are compared up to that point.
(ii)  closed-loop mode (j2k_params.closed_loop; this library's, not the reference's): 3840 x 2160 RGB8, MQ coder,
      pixels -> tile-parts of packets -> pixels, bit-exact.
(iii) every stage of (ii) against the oracle's composition of the reference's own functions (job list with windows that
      partition the plane, T1.EncodeFast5, PacketEncoder / createTileHeader, PacketDecoder, T1.Decode, placement,
      ReconstructMultiLevel53 + inverse RCT + DC shift) on a ragged multi-tile frame and on sampled tiles of the 4K frame."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "go-jpeg2000_amd"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

from test_gpu_t2 import _rand_bands  # noqa: E402


@pytest.fixture(scope="module")
def env():
    import torch
    import t2ref
    from j2kgfx import t2
    from j2kgfx.context import Context
    ctx = Context(0)
    yield torch, t2ref, t2, ctx
    ctx.close()


def _dev(torch, a):
    a = np.ascontiguousarray(a)
    return torch.from_numpy(a.view(np.uint8).reshape(-1).copy()).cuda() if a.size else None


def _decode_three_ways(env, data, shapes, layers, trees, sop, eph, flags=0, len_bits=3, seated=False, fresh_at=()):
    """shapes[p] = code-blocks per band of packet p.  Returns (per decoder: packets done, fields, bodies, Position)."""
    torch, t2ref, t2, ctx = env
    from j2kgfx import J2KError
    # restatement
    dec = t2ref.PacketDecoder(data, len_bits=len_bits, seated=seated)
    ref_done, ref_cbs, ref_err = 0, [], None
    for p, shape in enumerate(shapes):
        if p in fresh_at:
            dec.fresh()
        pr = t2ref.Precinct([[t2ref.CodeBlock(None, 0, 0, 0) for _ in range(n)] for n in shape], trees[p][0], trees[p][1])
        try:
            dec.decode_packet(pr, layers[p], sop, eph)
        except (t2ref.EOF, t2ref.GoPanic) as e:
            ref_err = type(e).__name__
            break
        ref_done += 1
        ref_cbs.append([(cb.included_in_layers, cb.zero_bit_planes, cb.num_passes, bytes(cb.data) if cb.data is not None else b"")
                        for band in pr.code_blocks for cb in band])
    ref_pos = dec.pos
    # device
    pk = np.zeros(len(shapes), t2.DEV_PACKET_DTYPE)
    k = 0
    for p, shape in enumerate(shapes):
        f = flags | (1 if p in fresh_at else 0)
        pk[p] = (layers[p], trees[p][0], trees[p][1], f, k, sum(shape))
        k += sum(shape)
    cbs = torch.zeros(max(k, 1) * 24, dtype=torch.uint8, device="cuda")
    d_data = _dev(torch, np.frombuffer(bytes(data), np.uint8)) if len(data) else torch.zeros(8, dtype=torch.uint8, device="cuda")[:0]
    dd = t2.DevicePacketDecoder(ctx)
    dev_err = None
    try:
        dd.decode(_dev(torch, pk), len(shapes), cbs, d_data if len(data) else None, sop, eph)
    except J2KError as e:
        dev_err = e.status
    tab = cbs.cpu().numpy().view(t2.DEV_CB_DTYPE)
    dev_cbs = []
    k = 0
    for p, shape in enumerate(shapes):
        if p >= dd.done:
            break
        row = []
        for _ in range(sum(shape)):
            c = tab[k]
            k += 1
            body = bytes(data[int(c["data_off"]):int(c["data_off"]) + int(c["data_len"])]) if (int(c["included_in_layers"]) == layers[p] and int(c["data_len"])) else None
            row.append((int(c["included_in_layers"]), int(c["zero_bit_planes"]), int(c["num_passes"]), int(c["data_len"]), body))
        dev_cbs.append(row)
    return (ref_done, ref_cbs, ref_err, ref_pos), (dd.done, dev_cbs, dev_err, dd.Position())


def _same(ref, dev, layers):
    ref_done, ref_cbs, ref_err, ref_pos = ref
    dev_done, dev_cbs, dev_err, dev_pos = dev
    assert dev_done == ref_done, (dev_done, ref_done, ref_err, dev_err)
    assert (ref_err is None) == (dev_err is None), (ref_err, dev_err)
    if ref_err == "GoPanic":
        assert dev_err == -5
    elif ref_err == "EOF":
        assert dev_err == -1
    for p in range(ref_done):
        assert len(ref_cbs[p]) == len(dev_cbs[p])
        for (ri, rz, rn, rd), (di, dz, dn, dl, body) in zip(ref_cbs[p], dev_cbs[p]):
            assert (ri, rz, rn, len(rd)) == (di, dz, dn, dl), p
            if ri == layers[p] and len(rd):
                assert body == rd, p                      # the body the reference copies == the bytes at data_off
    if ref_err is None:
        assert dev_pos == ref_pos


def test_device_packet_decoder_is_the_references_decoder_on_encoder_runs(env):
    """(i): 300 random encoder runs (the ones test_gpu_t2 draws), decoded by the reference's own decoder semantics three ways"""
    torch, t2ref, t2, ctx = env
    from j2kgfx import J2KError
    rng = np.random.default_rng(7)
    full = 0
    for it in range(300):
        ff_heavy = it % 2 == 0
        sop, eph = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        enc = t2ref.PacketEncoder()
        shapes, layers, trees = [], [], []
        for _ in range(int(rng.integers(1, 41))):
            bands = _rand_bands(rng, ff_heavy)
            if it % 5 == 0:                                     # runs the decoder can get through: one small block per packet, no SOP
                bands = [[(bytes(rng.integers(0, 256, int(rng.integers(1, 100))).astype(np.uint8)), 0, int(rng.integers(0, 9)), int(rng.choice([1, 2, 3, 7, 40])))]]
            layer = int(rng.integers(0, 3)) if it % 5 else 0
            enc.encode_packet(t2ref.Precinct([[t2ref.CodeBlock(*cb) for cb in b] for b in bands], 1, 1), layer, sop, eph)
            shapes.append([len(b) for b in bands]); layers.append(layer); trees.append((1, 1))
        data = bytes(enc.out)
        ref, dev = _decode_three_ways(env, data, shapes, layers, trees, sop, eph)
        _same(ref, dev, layers)
        full += ref[2] is None
        # the host call, packet by packet, against the same restatement
        hd = t2.PacketDecoder(data)
        done = 0
        for p, shape in enumerate(shapes):
            pr = t2.Precinct([[t2.CodeBlock(None, 0, 0, 0) for _ in range(n)] for n in shape])
            try:
                hd.DecodePacket(pr, layers[p], sop, eph)
            except J2KError:
                break
            done += 1
            got = [(cb.IncludedInLayers, cb.ZeroBitPlanes, cb.Passes, cb.Data or b"") for band in pr.CodeBlocks for cb in band]
            assert got == ref[1][p], (it, p)
        assert done == ref[0]
        if ref[2] is None:
            assert hd.Position() == ref[3]
    assert full >= 20                       # some runs do go through (a single small block per packet: body read from the header's own bytes)


def test_device_packet_decoder_arbitrary_bytes_tree_widths_and_prefilled_tables(env):
    """foreign input: random bytes as packets (long unary runs, lengths past the end), zero-width trees, layers > 0 on a
    table an earlier run filled"""
    torch, t2ref, t2, ctx = env
    rng = np.random.default_rng(11)
    for it in range(200):
        n = int(rng.integers(0, 400))
        kind = it % 4
        raw = rng.integers(0, 256, n).astype(np.uint8)
        if kind == 1:
            raw[rng.random(n) < 0.5] = 0                         # zero runs: long unary values
        elif kind == 2:
            raw[rng.random(n) < 0.5] = 0xFF                      # stuffing everywhere
        shapes = [[int(rng.integers(0, 5)) for _ in range(int(rng.integers(1, 4)))] for _ in range(int(rng.integers(1, 12)))]
        layers = [0 for _ in shapes]
        trees = [(int(rng.integers(0, 3)), int(rng.integers(0, 3))) for _ in shapes]
        sop, eph = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        ref, dev = _decode_three_ways(env, bytes(raw), shapes, layers, trees, sop, eph)
        _same(ref, dev, layers)


def _cl_runs(rng, ntiles):
    tiles = []
    for _ in range(ntiles):
        pk = []
        for _ in range(int(rng.integers(1, 7))):
            bands = []
            for _ in range(int(rng.integers(1, 4))):
                b = []
                for _ in range(int(rng.integers(0, 80 if rng.random() < 0.1 else 9))):
                    n = int(rng.choice([0, 1, 5, 127, 128, 300, 5000, 70000]))
                    d = bytes(rng.integers(0, 256, n).astype(np.uint8)) if rng.random() < 0.7 else b"\xff" * n
                    b.append((d, 1 if n == 0 else 0, int(rng.integers(0, 32)), int(rng.choice([1, 2, 4, 10, 40, 91]))))
                bands.append(b)
            pk.append(bands)
        tiles.append(pk)
    return tiles


def test_closed_loop_packets_device_encoder_and_decoder_against_the_restatement(env):
    """closed-loop flags (J2K_T2_FRESH / WIDE_LEN / SEATED): device encoder == t2ref.PacketEncoder(len_bits=5) with a new
    encoder per tile; device decoder on those bytes == t2ref.PacketDecoder(len_bits=5, seated=True) == what went in"""
    torch, t2ref, t2, ctx = env
    from j2kgfx import _lib
    rng = np.random.default_rng(21)
    for it in range(60):
        tiles = _cl_runs(rng, int(rng.integers(1, 5)))
        sop, eph = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        enc = t2ref.PacketEncoder(len_bits=5)
        run, flags, fresh_at, shapes = [], [], [], []
        for pk in tiles:
            enc.fresh()
            fresh_at.append(len(run))
            for bands in pk:
                enc.encode_packet(t2ref.Precinct([[t2ref.CodeBlock(*cb) for cb in b] for b in bands]), 0, sop, eph)
                flags.append(_lib.T2_WIDE_LEN | _lib.T2_SEATED | (_lib.T2_FRESH if len(run) == fresh_at[-1] else 0))
                run.append((t2.Precinct([[t2.CodeBlock(*cb) for cb in b] for b in bands]), 0))
                shapes.append([len(b) for b in bands])
        want = bytes(enc.out)
        dev_enc = t2.DevicePacketEncoder(ctx)
        packets, cbs, data = dev_enc.tables(run, flags)
        out = torch.zeros(len(want) + 64, dtype=torch.uint8, device="cuda")
        offs = torch.zeros(len(run) + 1, dtype=torch.int64, device="cuda")
        total = dev_enc.encode(_dev(torch, packets), len(run), _dev(torch, cbs), _dev(torch, data), sop, eph, out, offs)
        assert total == len(want) and out.cpu().numpy()[:total].tobytes() == want, it
        layers = [0] * len(run)
        ref, dev = _decode_three_ways(env, want, shapes, layers, [(1, 1)] * len(run), sop, eph,
                                      flags=_lib.T2_WIDE_LEN | _lib.T2_SEATED, len_bits=5, seated=True, fresh_at=set(fresh_at))
        _same(ref, dev, layers)
        assert ref[2] is None and ref[0] == len(run) and ref[3] == len(want)
        flat = [cb for pk in tiles for bands in pk for b in bands for cb in b]
        got = [c for row in dev[1] for c in row]
        assert len(flat) == len(got)
        for (d, incl, zbp, npass), (gi, gz, gn, gl, body) in zip(flat, got):
            if len(d):
                assert (gi, gz, gn, gl, body) == (0, zbp, npass, len(d), d)
            else:
                assert gl == 0 and body is None


from closed_loop_ref import frame as _frame, oracle_frame as _oracle_frame  # noqa: E402  (shared with the golden digests)


def _rgba(frame):
    C, H, W = frame.shape
    pix = np.full((H, W, 4), 255, np.uint8)
    pix[..., :3] = frame.transpose(1, 2, 0)
    return pix.reshape(H, W * 4)


def test_closed_loop_4k_rgb8_mq_pixels_to_tile_parts_to_pixels_bit_exact(env):
    """(ii): 3840 x 2160 RGB8, 512 x 512 tiles, 64 x 64 blocks, MQ coder: pixels -> SOT | SOD | packets -> pixels"""
    torch, t2ref, t2, ctx = env
    from j2kgfx import _lib
    from j2kgfx.codec import FramePlan
    W, H = 3840, 2160
    pix = _rgba(_frame(W, H, 5))
    plan = FramePlan(W, H, 3, precision=8, lossless=True, num_resolutions=6, cb=(64, 64), tile=(512, 512), coder=_lib.CODER_MQ, ctx=ctx, closed_loop=True)
    d_pix = torch.from_numpy(pix).to(plan.device)
    cs, toffs = plan.encode_frame_pixels(_lib.PIX_RGBA8, d_pix, sop=True, eph=True)
    plan.frame_status()
    total = int(toffs[-1].item())
    assert 0 < total <= cs.numel()
    h = cs[:14].cpu().numpy()
    assert bytes(h[:4]) == b"\xff\x90\x00\x0a" and bytes(h[12:14]) == b"\xff\x93"
    for given in (True, False):                                  # tile-part positions from the caller / found by walking the SOT segments
        back = torch.zeros_like(d_pix)
        plan.decode_frame_pixels(cs, total, back, tile_offs=toffs if given else None, sop=True, eph=True)
        plan.frame_status()
        got = back.cpu().numpy().reshape(H, W, 4)
        assert np.array_equal(got[..., :3], pix.reshape(H, W, 4)[..., :3]), given
        assert (got[..., 3] == 255).all()
    # a smaller buffer: nothing written, the need reported, the status says so
    from j2kgfx import J2KError
    small = torch.full((total - 1,), 0xA5, dtype=torch.uint8, device=plan.device)
    _, toffs2 = plan.encode_frame_pixels(_lib.PIX_RGBA8, d_pix, sop=True, eph=True, out=small)
    with pytest.raises(J2KError) as e:
        plan.frame_status()
    assert e.value.status == _lib.ERR_CAPACITY and int(toffs2[-1].item()) == total and (small.cpu().numpy() == 0xA5).all()
    # a truncated stream: malformed, reported, no fault
    with pytest.raises(J2KError) as e:
        plan.decode_frame_pixels(cs, total - 1000, torch.zeros_like(d_pix), tile_offs=None, sop=True, eph=True)
        plan.frame_status()
    assert e.value.status == _lib.ERR_INVALID_ARG
    plan.close()


@pytest.mark.parametrize("coder", [0, 1])
def test_closed_loop_every_stage_against_the_oracle_ragged_tiles(env, coder):
    """(iii) on a frame with ragged edge tiles (odd sizes: bands of unequal widths, one-sample bands): job windows, block bytes,
    tile-parts, parsed block tables, decoded + placed planes, pixels -- each against the oracle's composition"""
    torch, t2ref, t2, ctx = env
    import oracle as orc
    from j2kgfx import _lib
    from j2kgfx.codec import FramePlan
    W, H, tw, th, nres, cb = 301, 211, 128, 96, 4, 32
    frame = _frame(W, H, 3 + coder, noise=30 if coder == 0 else 3)
    plan = FramePlan(W, H, 3, precision=8, lossless=True, num_resolutions=nres, cb=(cb, cb), tile=(tw, th), coder=coder, ctx=ctx, closed_loop=True)
    want = _oracle_frame(frame, W, H, tw, th, nres, cb, coder, True, False, orc, t2ref)
    # job windows
    blocks = plan.blocks()
    planes = plan.planes()
    j = 0
    for t in sorted(want):
        jobs = orc.enumerate_blocks(3, want[t]["w"], want[t]["h"], nres, cb, cb, 1)
        for b in jobs:
            g = blocks[j]
            assert (int(planes[g["plane"]][0]), int(planes[g["plane"]][1]), g["band"], g["x0"], g["y0"], g["w"], g["h"]) == \
                (t, b["comp"], b["band"], b["x0"], b["y0"], b["w"], b["h"])
            j += 1
    assert j == len(blocks)
    # forward + block coder
    d_frame = torch.from_numpy(frame.astype(np.int32)).to(plan.device)
    coeff = plan.forward(d_frame)
    stream, offs, lens, numbps = plan.encode_stream(coeff)
    cs, toffs = plan.encode_tile_parts(stream, offs, lens, numbps, sop=True, eph=False)
    plan.frame_status()
    h_lens, h_nb = lens.cpu().numpy(), numbps.cpu().numpy()
    h_stream = stream.cpu().numpy()[:int(offs[-1].item())]
    assert bytes(h_stream) == b"".join(bytes(want[t]["bytes"]) for t in sorted(want))
    assert np.array_equal(h_lens[:len(blocks)].astype(np.uint32), np.concatenate([want[t]["lens"] for t in sorted(want)]))
    h_toffs = toffs.cpu().numpy()
    h_cs = cs.cpu().numpy()
    for i, t in enumerate(sorted(want)):
        assert bytes(h_cs[int(h_toffs[i]):int(h_toffs[i + 1])]) == want[t]["part"], t
    total = int(h_toffs[-1])
    # parse: the block tables point into the tile-parts
    offs2, lens2, nb2 = plan.decode_tile_parts(cs, total, tile_offs=None, sop=True, eph=False)
    plan.frame_status()
    o2, l2, n2 = offs2.cpu().numpy(), lens2.cpu().numpy(), nb2.cpu().numpy()
    assert np.array_equal(l2[:len(blocks)], h_lens[:len(blocks)])
    pos = 0
    for k in range(len(blocks)):
        ln = int(l2[k])
        if ln:
            assert bytes(h_cs[int(o2[k]):int(o2[k]) + ln]) == bytes(h_stream[pos:pos + ln]), k
            assert int(n2[k]) == int(h_nb[k])
        else:
            assert int(n2[k]) == 0
        pos += ln
    # block decode + placement against DecodeCodeBlock for every job of the oracle's list
    decoded = plan.decode_blocks(cs, offs2, lens2, nb2)
    placed = plan.place_blocks(decoded)
    back = plan.inverse(placed)
    ctx.sync()
    hp = placed.cpu().numpy()
    for t in sorted(want):
        wt = want[t]
        ref_planes = orc.decode_tile_blocks(wt["bytes"], wt["lens"], wt["numbps"], 3, wt["w"], wt["h"], nres, cb, cb, coder, 1)
        for c in range(3):
            row = [r for r in planes if int(r[0]) == t and int(r[1]) == c][0]
            got = hp[int(row[6]):int(row[6]) + wt["w"] * wt["h"]].reshape(wt["h"], wt["w"])
            assert np.array_equal(got, ref_planes[c]), (t, c)
            if coder == 0:
                assert np.array_equal(got, wt["coeff"][c])        # the MQ coder is lossless: the coefficients come back
        sub = [orc.reconstruct53(ref_planes[c], wt["w"], wt["h"], nres - 1) for c in range(3)]
        px = orc.postprocess(sub, 8, True)
        for c in range(3):
            assert np.array_equal(back.cpu().numpy()[c, wt["y0"]:wt["y0"] + wt["h"], wt["x0"]:wt["x0"] + wt["w"]], px[c]), (t, c)
    if coder == 0:
        assert np.array_equal(back.cpu().numpy(), frame.astype(np.int32))
    plan.close()


def test_closed_loop_4k_sampled_tiles_against_the_oracle(env):
    """(iii) at full size: tile-parts 0 (full 512 x 512), 7 (256 wide) and 39 (256 x 112) of the 4K MQ frame == the oracle's"""
    torch, t2ref, t2, ctx = env
    import oracle as orc
    from j2kgfx import _lib
    from j2kgfx.codec import FramePlan
    W, H = 3840, 2160
    frame = _frame(W, H, 5)
    plan = FramePlan(W, H, 3, precision=8, lossless=True, num_resolutions=6, cb=(64, 64), tile=(512, 512), coder=_lib.CODER_MQ, ctx=ctx, closed_loop=True)
    d_pix = torch.from_numpy(_rgba(frame)).to(plan.device)
    cs, toffs = plan.encode_frame_pixels(_lib.PIX_RGBA8, d_pix, sop=False, eph=True)
    plan.frame_status()
    h_cs, h_t = cs.cpu().numpy(), toffs.cpu().numpy()
    want = _oracle_frame(frame, W, H, 512, 512, 6, 64, 0, False, True, orc, t2ref, tiles={0, 7, 39})
    for t in (0, 7, 39):
        assert bytes(h_cs[int(h_t[t]):int(h_t[t + 1])]) == want[t]["part"], t
    plan.close()


@pytest.mark.parametrize("tw,th,W,H,nres,cb", [(1, 64, 3, 64, 4, 16), (2, 33, 5, 40, 3, 8), (3, 5, 7, 9, 6, 4), (64, 1, 130, 2, 5, 32), (5, 7, 5, 7, 1, 64),
                                                 (128, 64, 383, 64, 5, 8)])      # (8 x 8 blocks: packet headers longer than the parser's 256-byte chunk -- fuzz finding, round 5)
def test_closed_loop_degenerate_tiles_round_trip_and_packets_per_resolution(env, tw, th, W, H, nres, cb):
    """tiles one to three samples wide / one row high: bands without samples have no jobs, consecutive resolutions can hold
    the SAME single band -- a packet is still the jobs of one resolution (j2k_plan_t2_packets used to merge them).
    Closed loop: bit-exact round trip; both modes: every packet's jobs share one (tile-component, resolution)."""
    torch, t2ref, t2, ctx = env
    import oracle as orc
    from j2kgfx.codec import FramePlan
    rng = np.random.default_rng(W * 131 + H)
    frame = rng.integers(0, 256, (3, H, W)).astype(np.int32)
    for closed in (True, False):
        plan = FramePlan(W, H, 3, precision=8, lossless=True, num_resolutions=nres, cb=(cb, cb), tile=(tw, th), ctx=ctx, closed_loop=closed)
        blocks, planes, packets = plan.blocks(), plan.planes(), plan.t2_packets(0)
        res = []
        for t in range(int(plan.info.tiles)):
            w, h = int(planes[t * 3][4]), int(planes[t * 3][5])
            res += [int(b["res"]) for b in orc.enumerate_blocks(3, w, h, nres, cb, cb, 1 if closed else 0)]
        assert len(res) == len(blocks) and int(packets["ncb"].sum()) == len(blocks)
        for pk in packets:
            j0, j1 = int(pk["cb0"]), int(pk["cb0"] + pk["ncb"])
            assert len({(int(blocks[j]["plane"]), res[j]) for j in range(j0, j1)}) == 1
        keys = [(int(blocks[int(pk["cb0"])]["plane"]), res[int(pk["cb0"])]) for pk in packets]
        assert len(set(keys)) == len(keys)                       # and no (tile-component, resolution) is split over two packets
        if closed:
            coeff = plan.forward(torch.from_numpy(frame).to(plan.device))
            stream, offs, lens, numbps = plan.encode_stream(coeff)
            cs, toffs = plan.encode_tile_parts(stream, offs, lens, numbps, sop=True, eph=True)
            plan.frame_status()
            total = int(toffs[-1].item())
            o2, l2, n2 = plan.decode_tile_parts(cs, total, tile_offs=None, sop=True, eph=True)
            back = plan.inverse(plan.place_blocks(plan.decode_blocks(cs, o2, l2, n2)))
            plan.frame_status()
            assert np.array_equal(back.cpu().numpy(), frame)
        plan.close()


@pytest.mark.parametrize("W,H,tile,cb,nres", [(512, 384, (256, 256), 64, 5), (512, 512, (0, 0), 256, 6)])
def test_closed_loop_lossy_default_options_round_trip_of_the_quantised_coefficients(env, W, H, tile, cb, nres):
    """the lossy path (9-7 + quantisation, Quality 75 -- the reference's DefaultOptions, second case with its default 256 x 256
    code-blocks) through the closed loop: the MQ coder is lossless on the quantised coefficients, so what the decode body hands
    the inverse transform is exactly what the forward transform made, and the pixels are the library's own inverse of those"""
    torch, t2ref, t2, ctx = env
    from j2kgfx import _lib
    from j2kgfx.codec import FramePlan
    pix = _rgba(_frame(W, H, 9, noise=12))
    plan = FramePlan(W, H, 3, precision=8, lossless=False, quality=75, num_resolutions=nres, cb=(cb, cb), tile=tile, coder=_lib.CODER_MQ, ctx=ctx, closed_loop=True)
    d_pix = torch.from_numpy(pix).to(plan.device)
    cs, toffs = plan.encode_frame_pixels(_lib.PIX_RGBA8, d_pix, sop=True, eph=True)
    back = torch.zeros_like(d_pix)
    plan.decode_frame_pixels(cs, int(cs.numel()), back, tile_offs=toffs, sop=True, eph=True)
    plan.frame_status()
    coeff = plan.forward_pixels(_lib.PIX_RGBA8, d_pix)
    want = plan.inverse_pixels(coeff, torch.zeros_like(d_pix))
    # stage by stage as well: parsed blocks -> decoded -> placed == the forward transform's coefficients
    total = int(toffs[-1].item())
    o2, l2, n2 = plan.decode_tile_parts(cs, total, tile_offs=None, sop=True, eph=True)
    placed = plan.place_blocks(plan.decode_blocks(cs, o2, l2, n2))
    plan.frame_status()
    assert torch.equal(placed[:int(plan.info.coeff_elems)], coeff[:int(plan.info.coeff_elems)])
    assert torch.equal(back, want)
    err = (back.cpu().numpy().astype(np.int32).reshape(H, W, 4)[..., :3] - pix.astype(np.int32).reshape(H, W, 4)[..., :3])
    assert np.abs(err).max() <= 128            # (the reference's decode path never dequantises, SURVEY: this is not a reconstruction of the source; it is the same pixels either way)
    plan.close()


@pytest.mark.parametrize("coder", [0, 1])
def test_closed_loop_corrupted_tile_parts_are_reported_not_followed(env, coder):
    """foreign input: bytes of a valid frame overwritten at random (SOT fields, packet headers, bodies), truncations, garbage
    tile-part positions -- the decode either succeeds (the damage hit a body) or reports J2K_ERR_INVALID_ARG; nothing is read
    outside the buffer (the block tables are bounds-checked before the block decoder sees them) and nothing hangs"""
    torch, t2ref, t2, ctx = env
    from j2kgfx import J2KError, _lib
    from j2kgfx.codec import FramePlan
    W, H = 200, 150
    rng = np.random.default_rng(31 + coder)
    frame = _frame(W, H, 4, noise=25)
    plan = FramePlan(W, H, 3, precision=8, lossless=True, num_resolutions=4, cb=(16, 16), tile=(64, 64), coder=coder, ctx=ctx, closed_loop=True)
    coeff = plan.forward(torch.from_numpy(frame.astype(np.int32)).to(plan.device))
    stream, offs, lens, numbps = plan.encode_stream(coeff)
    cs, toffs = plan.encode_tile_parts(stream, offs, lens, numbps, sop=True, eph=True)
    plan.frame_status()
    total = int(toffs[-1].item())
    good = cs[:total].cpu().numpy().copy()
    h_toffs = toffs.cpu().numpy().copy()
    guard = 4096
    outcomes = {"ok": 0, "invalid": 0}
    for it in range(120):
        bad = good.copy()
        kind = it % 6
        n = total
        t_offs = h_toffs.astype(np.uint64).copy()
        if kind == 0:                                            # a few random bytes anywhere
            for _ in range(int(rng.integers(1, 6))):
                bad[int(rng.integers(0, total))] = int(rng.integers(0, 256))
        elif kind == 1:                                          # the first 64 bytes of a tile-part: SOT, SOD, the first packet headers
            t = int(rng.integers(0, len(h_toffs) - 1))
            for _ in range(int(rng.integers(1, 8))):
                bad[int(h_toffs[t]) + int(rng.integers(0, 64))] = int(rng.integers(0, 256))
        elif kind == 2:                                          # truncation
            n = int(rng.integers(0, total))
        elif kind == 3:                                          # runs of zeros / 0xFF over headers (long unary values, stuffing everywhere)
            a = int(rng.integers(0, total - 300))
            bad[a:a + int(rng.integers(8, 300))] = 0 if it % 2 else 0xFF
        elif kind == 4:                                          # tile-part positions from a hostile caller
            t_offs[int(rng.integers(0, len(t_offs)))] = np.uint64([0, total, total + 12345, 2 ** 62, 2 ** 64 - 1 - int(rng.integers(0, 16))][int(rng.integers(0, 5))])
        else:                                                    # everything random
            bad = rng.integers(0, 256, total).astype(np.uint8)
        buf = torch.zeros(total + 2 * guard, dtype=torch.uint8, device=plan.device)
        buf[guard:guard + total] = torch.from_numpy(bad).to(plan.device)
        d_t = torch.from_numpy(t_offs.view(np.int64).copy()).to(plan.device)
        o2, l2, n2 = plan.decode_tile_parts(buf[guard:], n, tile_offs=d_t if (kind == 4 or it % 2) else None, sop=True, eph=True)
        placed = plan.place_blocks(plan.decode_blocks(buf[guard:], o2, l2, n2))
        try:
            plan.frame_status()
            outcomes["ok"] += 1
            failed = False
        except J2KError as e:
            assert e.status == _lib.ERR_INVALID_ARG
            outcomes["invalid"] += 1
            failed = True
        # the packets of a tile side by side (the default for SOP + EPH streams) and one after the other: the same answer on damaged input too
        ctx.set_option("t2_parallel", 0)
        o3, l3, n3 = plan.decode_tile_parts(buf[guard:], n, tile_offs=d_t if (kind == 4 or it % 2) else None, sop=True, eph=True)
        ctx.set_option("t2_parallel", 1)
        try:
            plan.frame_status()
            assert not failed
        except J2KError:
            assert failed
        if not failed:
            nb = int(plan.info.blocks)
            assert torch.equal(o2[:nb], o3[:nb]) and torch.equal(l2[:nb], l3[:nb]) and torch.equal(n2[:nb], n3[:nb]), (it, kind)
        # whatever came out points inside the buffer the caller gave
        o, l = o2.cpu().numpy()[:int(plan.info.blocks)].astype(np.uint64), l2.cpu().numpy()[:int(plan.info.blocks)].astype(np.uint64)
        assert ((o + l) <= n).all()
        assert int(n2.cpu().numpy()[:int(plan.info.blocks)].max()) <= 31
    assert outcomes["invalid"] > 20 and outcomes["ok"] > 0, outcomes
    # and the intact stream still decodes
    o2, l2, n2 = plan.decode_tile_parts(cs, total, sop=True, eph=True)
    back = plan.inverse(plan.place_blocks(plan.decode_blocks(cs, o2, l2, n2)))
    plan.frame_status()
    if coder == 0:
        assert np.array_equal(back.cpu().numpy(), frame.astype(np.int32))
    plan.close()


@pytest.mark.parametrize("coder", [0, 1])
def test_closed_loop_packets_of_a_tile_side_by_side_equal_the_tile_chain(env, coder):
    """SOP + EPH streams: packet starts guessed from the markers, each packet decoded on its own, kept only when every packet ends where
    (and how) the next was started -- the tile chain's result by induction.  Checked here against the tile chain itself (option
    t2_parallel = 0): intact streams (every tile of an MQ frame goes the parallel way), a marker pair planted in a body (that tile falls
    back; same bytes out), headers ending in 0xFF (the carried flag), streams without EPH (nothing to guess from)"""
    torch, t2ref, t2, ctx = env
    from j2kgfx.codec import FramePlan
    rng = np.random.default_rng(77 + coder)
    for (W, H, tile, cb, nres, noise) in [(640, 384, (128, 128), 32, 4, 30), (300, 200, (0, 0), 16, 5, 60), (1024, 512, (512, 512), 64, 6, 8), (96, 64, (32, 32), 8, 3, 90)]:
        frame = _frame(W, H, 5, noise=noise).astype(np.int32)
        plan = FramePlan(W, H, 3, precision=8, lossless=True, num_resolutions=nres, cb=(cb, cb), tile=tile, coder=coder, ctx=ctx, closed_loop=True)
        tiles = int(plan.info.tiles)
        coeff = plan.forward(torch.from_numpy(frame).to(plan.device))
        stream, offs, lens, numbps = plan.encode_stream(coeff)

        def both(cs, total, sop, eph, toffs):
            plan.frame_parallel_tiles()
            nb = int(plan.info.blocks)
            a = plan.decode_tile_parts(cs, total, tile_offs=toffs, sop=sop, eph=eph)
            par = plan.frame_parallel_tiles()
            ctx.set_option("t2_parallel", 0)
            b = plan.decode_tile_parts(cs, total, tile_offs=toffs, sop=sop, eph=eph)
            ctx.set_option("t2_parallel", 1)
            assert plan.frame_parallel_tiles() == 0
            plan.frame_status()
            a = [x[:nb] for x in a]
            for x, y in zip(a, b):
                assert torch.equal(x, y[:nb])
            return a, par

        cs, toffs = plan.encode_tile_parts(stream, offs, lens, numbps, sop=True, eph=True)
        plan.frame_status()
        total = int(toffs[-1].item())
        (o2, l2, n2), par = both(cs, total, True, True, toffs)
        if coder == 0:
            assert par == tiles                                   # an MQ body holds no FF91 / FF92: every tile goes the parallel way
        else:
            assert par >= tiles - 2
        back = plan.inverse(plan.place_blocks(plan.decode_blocks(cs, o2, l2, n2)))
        plan.frame_status()
        if coder == 0:
            assert np.array_equal(back.cpu().numpy(), frame)
        # a marker pair inside the largest body: the list of that tile no longer reads SOP EPH SOP EPH ... and the tile chain takes over
        h_o, h_l = o2.cpu().numpy(), l2.cpu().numpy()
        j = int(np.argmax(h_l))
        if h_l[j] >= 8:
            bad = cs.clone()
            at = int(h_o[j]) + int(rng.integers(1, int(h_l[j]) - 4))
            bad[at] = 0xFF
            bad[at + 1] = 0x91 if rng.integers(0, 2) else 0x92
            (o3, l3, n3), par3 = both(bad, total, True, True, None)
            assert par3 == par - 1 or (coder == 1 and par3 <= par)
            assert torch.equal(o3, o2) and torch.equal(l3, l2) and torch.equal(n3, n2)          # (a body's bytes do not move the packets)
        # ... two of them, in order (a false packet in the middle of a body): the count is off
        if h_l[j] >= 16:
            bad = cs.clone()
            at = int(h_o[j]) + 2
            bad[at:at + 2] = torch.tensor([0xFF, 0x91], dtype=torch.uint8, device=bad.device)
            bad[at + 9:at + 11] = torch.tensor([0xFF, 0x92], dtype=torch.uint8, device=bad.device)
            (o3, l3, n3), par3 = both(bad, total, True, True, toffs)
            assert par3 < tiles and torch.equal(o3, o2) and torch.equal(l3, l2)
        # without EPH (or SOP) there is nothing to take the carried flag from: the tile chain as before
        for sop, eph in [(True, False), (False, True), (False, False)]:
            cs2, toffs2 = plan.encode_tile_parts(stream, offs, lens, numbps, sop=sop, eph=eph)
            plan.frame_status()
            (o4, l4, n4), par4 = both(cs2, int(toffs2[-1].item()), sop, eph, toffs2)
            assert par4 == 0 and torch.equal(l4, l2) and torch.equal(n4, n2)
        plan.close()


def test_closed_loop_side_by_side_packets_with_headers_ending_in_0xff(env):
    """the one thing a packet inherits from the one before it: whether that header's last byte was 0xFF (its own first byte then holds
    7 bits).  Rare in real frames (the header has to end on a byte boundary in eight 1 bits), so block lengths and bit-plane counts
    are drawn at random -- the packet parser reads headers only, a body's bytes are never looked at -- until headers end that way"""
    torch, t2ref, t2, ctx = env
    from j2kgfx.codec import FramePlan
    W, H = 256, 192
    frame = _frame(W, H, 100, noise=60).astype(np.int32)
    plan = FramePlan(W, H, 3, precision=8, lossless=True, num_resolutions=3, cb=(16, 16), tile=(64, 64), coder=0, ctx=ctx, closed_loop=True)
    nb, tiles = int(plan.info.blocks), int(plan.info.tiles)
    coeff = plan.forward(torch.from_numpy(frame).to(plan.device))
    stream, offs, lens, numbps = plan.encode_stream(coeff)
    plan.frame_status()
    h_lens = lens.cpu().numpy()[:nb].astype(np.int64)
    rng = np.random.default_rng(5)
    seen = 0
    for trial in range(400):
        rl = (h_lens * rng.random(nb)).astype(np.int64)             # 0 ... the real length: always inside the stream
        ones = (1 << np.maximum(np.floor(np.log2(np.maximum(h_lens, 1) + 1)).astype(np.int64) - rng.integers(0, 2, nb), 0)) - 1    # 2^k - 1 <= the real length: a length field of 1 bits
        l2 = torch.from_numpy(np.where(rng.random(nb) < 0.5, ones, rl).astype(np.int32)).to(plan.device)
        n2 = torch.from_numpy(rng.integers(1, 17, nb).astype(np.uint8)).to(plan.device)
        lens_t, nbp_t = lens.clone(), numbps.clone()
        lens_t[:nb] = l2
        nbp_t[:nb] = n2
        cs, toffs = plan.encode_tile_parts(stream, offs, lens_t, nbp_t, sop=True, eph=True)
        plan.frame_status()
        total = int(toffs[-1].item())
        h = cs[:total].cpu().numpy()
        eph_at = np.nonzero((h[:-1] == 0xFF) & (h[1:] == 0x92))[0]
        hits = int((h[eph_at - 1] == 0xFF).sum())
        if not hits:
            continue
        seen += hits
        plan.frame_parallel_tiles()
        a = plan.decode_tile_parts(cs, total, tile_offs=toffs, sop=True, eph=True)
        assert plan.frame_parallel_tiles() == tiles
        ctx.set_option("t2_parallel", 0)
        b = plan.decode_tile_parts(cs, total, tile_offs=toffs, sop=True, eph=True)
        ctx.set_option("t2_parallel", 1)
        plan.frame_status()
        for x, y in zip(a, b):
            assert torch.equal(x[:nb], y[:nb])
        assert torch.equal(a[1][:nb], l2)                           # and the lengths that went in
        if seen >= 4:
            break
    assert seen >= 1
    plan.close()


@pytest.mark.parametrize("coder", [0, 1])
@pytest.mark.parametrize("W,H,tile,cb,nres", [(301, 211, (128, 96), 32, 4), (640, 360, (0, 0), 64, 6), (97, 130, (32, 64), 8, 3), (1024, 768, (512, 512), 64, 6)])
def test_closed_loop_frame_encoder_gathers_from_the_slots_what_the_stage_calls_copy(env, coder, W, H, tile, cb, nres):
    """j2k_plan_encode_frame_pixels writes every block's bytes once -- from its coding slot (HT: MagSgn | the MEL zero run made on the way |
    VLC) straight to its place in its packet in its tile-part; the stage calls go through the dense block stream and the same packet coder.
    Same bytes, same tile-part positions, with and without the markers"""
    torch, t2ref, t2, ctx = env
    from j2kgfx import _lib
    from j2kgfx.codec import FramePlan
    frame = _frame(W, H, 21 + coder, noise=25)
    plan = FramePlan(W, H, 3, precision=8, lossless=True, num_resolutions=nres, cb=(cb, cb), tile=tile, coder=coder, ctx=ctx, closed_loop=True)
    d_pix = torch.from_numpy(_rgba(frame)).to(plan.device)
    for sop, eph in [(True, True), (False, False), (True, False)]:
        coeff = plan.forward_pixels(_lib.PIX_RGBA8, d_pix)
        stream, offs, lens, numbps = plan.encode_stream(coeff)
        cs1, t1 = plan.encode_tile_parts(stream, offs, lens, numbps, sop=sop, eph=eph)
        plan.frame_status()
        cs2, t2_ = plan.encode_frame_pixels(_lib.PIX_RGBA8, d_pix, sop=sop, eph=eph)
        plan.frame_status()
        total = int(t1[-1].item())
        assert torch.equal(t1, t2_)
        assert torch.equal(cs1[:total], cs2[:total])
        # and it decodes (MQ: to the source)
        back = torch.zeros_like(d_pix)
        plan.decode_frame_pixels(cs2, total, back, tile_offs=None, sop=sop, eph=eph)
        plan.frame_status()
        if coder == 0:
            assert torch.equal(back, d_pix)
    plan.close()


@pytest.mark.parametrize("W,H,tile,cb,nres", [(640, 360, (256, 256), 64, 5), (301, 211, (128, 96), 32, 4), (130, 70, (0, 0), 16, 3)])
def test_closed_loop_ht_frame_decoder_touches_the_coded_rows_only_and_stays_equal_to_the_stage_calls(env, W, H, tile, cb, nres):
    """j2k_plan_decode_frame_pixels on an HT plan writes only the rows the reference's HT decoder writes (y % 4 == 0), straight into the blocks'
    windows of coefficient planes it zeroed once (no dense blocks, no placement).  Frame after frame on ONE plan -- busy, flat (empty blocks where there were bytes a frame ago), busy again --
    the pixels equal those of the stage calls, which zero, decode and copy every row every time"""
    torch, t2ref, t2, ctx = env
    from j2kgfx import _lib
    from j2kgfx.codec import FramePlan
    plan = FramePlan(W, H, 3, precision=8, lossless=True, num_resolutions=nres, cb=(cb, cb), tile=tile, coder=_lib.CODER_HT, ctx=ctx, closed_loop=True)
    frames = [_frame(W, H, 40, noise=40), np.full((3, H, W), 128, np.uint8), _frame(W, H, 41, noise=3), _frame(W, H, 42, noise=90)]
    frames[1][:, H // 2, W // 3] = 255
    for k, frame in enumerate(frames):
        d_pix = torch.from_numpy(_rgba(frame)).to(plan.device)
        cs, toffs = plan.encode_frame_pixels(_lib.PIX_RGBA8, d_pix, sop=True, eph=True)
        plan.frame_status()
        total = int(toffs[-1].item())
        got = torch.zeros_like(d_pix)
        plan.decode_frame_pixels(cs, total, got, tile_offs=toffs, sop=True, eph=True)
        plan.frame_status()
        o2, l2, n2 = plan.decode_tile_parts(cs, total, tile_offs=toffs, sop=True, eph=True)
        want = plan.inverse_pixels(plan.place_blocks(plan.decode_blocks(cs, o2, l2, n2)), torch.zeros_like(d_pix))
        plan.frame_status()
        assert torch.equal(got, want), k
    plan.close()


@pytest.mark.parametrize("coder", [0, 1])
def test_closed_loop_frame_calls_replay_from_a_hip_graph(env, coder):
    """the frame calls are asynchronous launches on the context's stream with workspaces made at their first call: after one warm-up they can be
    captured (j2k_ctx_capture_begin / _end) and a whole frame -- pixels -> tile-parts -> pixels, some thirty kernels -- replayed as ONE graph launch,
    on new pixel contents in the same buffers"""
    torch, t2ref, t2, ctx = env
    from j2kgfx import _lib
    from j2kgfx.codec import FramePlan
    W, H = 400, 300
    plan = FramePlan(W, H, 3, precision=8, lossless=True, num_resolutions=4, cb=(32, 32), tile=(128, 128), coder=coder, ctx=ctx, closed_loop=True)
    d_pix = torch.from_numpy(_rgba(_frame(W, H, 60, noise=20))).to(plan.device)
    back = torch.zeros_like(d_pix)
    cs = plan.empty(plan.frame_bound(), torch.uint8)
    toffs = plan.empty(int(plan.info.tiles) + 1, torch.int64)[:int(plan.info.tiles) + 1]

    def code():
        plan.encode_frame_pixels(_lib.PIX_RGBA8, d_pix, True, True, cs, toffs)
        plan.decode_frame_pixels(cs, cs.numel(), back, toffs, True, True)
    code()
    plan.frame_status()
    with ctx.capture() as g:
        code()
    for seed in (61, 62):
        d_pix.copy_(torch.from_numpy(_rgba(_frame(W, H, seed, noise=5 * (seed - 58)))).to(plan.device))
        torch.cuda.synchronize()
        back.zero_()
        g.launch()
        plan.frame_status()
        got, cs_g = back.clone(), cs.clone()
        code()
        plan.frame_status()
        assert torch.equal(back, got) and torch.equal(cs, cs_g)
        if coder == 0:
            assert torch.equal(got, d_pix)
    plan.close()


def test_closed_loop_calls_refuse_a_reference_mode_plan(env):
    torch, t2ref, t2, ctx = env
    from j2kgfx import J2KError, _lib
    from j2kgfx.codec import FramePlan
    plan = FramePlan(64, 64, 1, num_resolutions=3, cb=(32, 32), ctx=ctx)
    assert plan.frame_bound() == 0
    with pytest.raises(J2KError) as e:
        plan.place_blocks(plan.empty(plan.info.decoded_elems, torch.int32))
    assert e.value.status == _lib.ERR_UNSUPPORTED
    plan.close()


@pytest.mark.parametrize("closed", [False, True])
def test_host_pixels_one_call_forms(env, closed):
    """j2k_encode_pixels_host / j2k_decode_pixels_host: image.RGBA.Pix in host memory -> tile-parts in host memory (and back, for a
    closed-loop plan) == what the device-buffer calls produce from the same pixels; unaligned stride; capacity reported"""
    torch, t2ref, t2, ctx = env
    from j2kgfx import J2KError, _lib, codestream
    from j2kgfx.codec import FramePlan
    W, H = 600, 300
    stride = W * 4 + 12                                          # a Go sub-image: rows not 16-byte multiples apart
    frame = _frame(W, H, 13, noise=20)
    pix = np.zeros((H, stride), np.uint8)
    pix[:, :W * 4] = _rgba(frame)
    plan = FramePlan(W, H, 3, precision=8, lossless=True, num_resolutions=5, cb=(32, 32), tile=(256, 128), coder=_lib.CODER_MQ, ctx=ctx, closed_loop=closed)
    got = plan.encode_pixels_host(_lib.PIX_RGBA8, pix, sop=True, eph=False)
    d_pix = torch.from_numpy(pix).to(plan.device)
    coeff = plan.forward_pixels(_lib.PIX_RGBA8, d_pix)
    stream, offs, lens, numbps = plan.encode_stream(coeff)
    n = int(plan.info.blocks)
    if closed:
        cs, toffs = plan.encode_tile_parts(stream, offs, lens, numbps, sop=True, eph=False)
        plan.frame_status()
        want = cs[:int(toffs[-1].item())].cpu().numpy()
        assert np.array_equal(got["tile_offs"].astype(np.int64), toffs.cpu().numpy())
    else:
        out, out_len = plan.assemble_tiles(stream, offs)
        ctx.sync()
        want = out[:int(out_len[0].item())].cpu().numpy()
        parts = codestream.parse_tile_parts(got["bytes"].tobytes())
        assert [p.TileIndex for p, _ in parts] == list(range(int(plan.info.tiles)))
        # tile_offs[t] = where tile-part t starts
        starts = [0]
        for _, d in parts:
            starts.append(starts[-1] + 14 + len(d))
        assert [int(v) for v in got["tile_offs"]] == starts
    ctx.sync()
    assert np.array_equal(got["bytes"], want)
    assert np.array_equal(got["lens"], lens.cpu().numpy()[:n].astype(np.uint32)) and np.array_equal(got["numbps"], numbps.cpu().numpy()[:n])
    with pytest.raises(J2KError) as e:
        plan.encode_pixels_host(_lib.PIX_RGBA8, pix, sop=True, eph=False, cap=got["bytes"].size - 1)
    assert e.value.status == _lib.ERR_CAPACITY and plan.encoded_len == got["bytes"].size
    if closed:
        back = plan.decode_pixels_host(got["bytes"], (H, stride), sop=True, eph=False)
        assert np.array_equal(back[:, :W * 4].reshape(H, W, 4)[..., :3], pix[:, :W * 4].reshape(H, W, 4)[..., :3])
        with pytest.raises(J2KError):
            plan.decode_pixels_host(got["bytes"][:-50], (H, stride), sop=True, eph=False)
    else:
        with pytest.raises(J2KError) as e:
            plan.decode_pixels_host(got["bytes"], (H, stride))
        assert e.value.status == _lib.ERR_UNSUPPORTED
    plan.close()
