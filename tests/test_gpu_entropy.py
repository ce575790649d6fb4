"""GPU parity: T1 (MQ) and HT block coders through the C ABI vs the C oracle: encoded BYTES are
compared (the reference only asserts round trips: internal/entropy/t1_test.go:7-96,
coverage_test.go:72-98,414-446,464-535,814-842,883-902; ht_test.go:7-77,126-147)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ent():
    from j2kgfx import entropy
    return entropy


def ref_test_block(w, h):
    """input generator of the reference's own tests: (i*17)%512 with every 7th negated (t1_test.go)"""
    i = np.arange(w * h, dtype=np.int64)
    v = (i * 17) % 512
    v[i % 7 == 0] *= -1
    return v.astype(np.int32).reshape(h, w)


SHAPES = [(4, 4), (8, 8), (16, 16), (32, 32), (1, 1), (8, 1), (1, 8), (8, 5), (64, 64), (13, 9), (5, 3), (64, 7), (33, 64)]


@pytest.mark.parametrize("w,h", SHAPES)
@pytest.mark.parametrize("band", [0, 1, 2, 3])
def test_t1_encode_bytes_and_roundtrip(ent, oracle, w, h, band):
    rng = np.random.default_rng(w * 131 + h * 7 + band)
    for x in (ref_test_block(w, h), rng.integers(-300, 301, (h, w)).astype(np.int32),
              -np.abs(rng.integers(1, 40000, (h, w))).astype(np.int32), np.zeros((h, w), np.int32)):
        t1 = ent.NewT1(w, h)
        t1.SetData(x)
        got = t1.Encode(band)
        want, nb = oracle.t1_encode(x, w, h, band)
        if want.size == 0:
            assert got is None and t1.numBPS == 0
            continue
        assert got == bytes(want) and t1.numBPS == nb
        back = ent.NewT1(w, h).Decode(got, nb, band)
        assert np.array_equal(back.reshape(h, w), x)                     # Decode(Encode(x)) == x


def test_t1_decode_arbitrary_bytes(ent, oracle):
    """decoder parity on streams that are not encoder output (FuzzT1Decode's domain)"""
    rng = np.random.default_rng(3)
    for (w, h) in [(8, 8), (16, 5), (64, 64)]:
        for nb in (1, 5, 12):
            g = rng.integers(0, 256, rng.integers(0, 200)).astype(np.uint8)
            got = ent.NewT1(w, h).Decode(bytes(g), nb, 2)
            assert np.array_equal(got.reshape(h, w), oracle.t1_decode(g, nb, 2, w, h))


def test_t1_large_block_global_workspace(ent, oracle):
    """256x256 = the reference's default real block size (1 << (6+2), encoder.go:606-607)"""
    rng = np.random.default_rng(11)
    x = rng.integers(-2000, 2001, (256, 256)).astype(np.int32)
    t1 = ent.NewT1(256, 256); t1.SetData(x)
    got = t1.Encode(3)
    want, nb = oracle.t1_encode(x, 256, 256, 3)
    assert got == bytes(want) and t1.numBPS == nb
    assert np.array_equal(ent.NewT1(256, 256).Decode(got, nb, 3).reshape(256, 256), x)


@pytest.mark.parametrize("env", [{}, {"J2K_T1_SPLIT": "0"}, {"J2K_T1_LANES": "64"}, {"J2K_T1_LANES": "3"}, {"J2K_T1_SYM_MB": "2"},
                                 {"J2K_T1_SYM_MB": "0"}])
def test_t1_batch_all_encoder_paths(ent, oracle, env, monkeypatch):
    """One batch of blocks of mixed size, band and bit depth through every arrangement of the T1 encoder: contexts and MQ
    coder as two kernels with K blocks per wavefront in lock step (default; K forced to 3 and 64), as one kernel
    (J2K_T1_SPLIT=0), and with a symbol workspace too small for the deep blocks (they fall back to the one-kernel path on
    the device; 0 MiB: all of them).  Every block's bytes and numBPS must be the oracle's."""
    from j2kgfx import Context
    for k, v in env.items():
        monkeypatch.setenv(k, v)                                   # read when a context is created
    ctx = Context(0)
    rng = np.random.default_rng(77)
    PW, PH = 520, 330
    plane = np.zeros((PH, PW), np.int32)
    dims = [(64, 64), (33, 64), (64, 7), (5, 3), (1, 1), (64, 64), (16, 16), (64, 64), (8, 64), (64, 33)]
    blocks = []
    for gy in range(5):
        for gx in range(8):
            w, h = dims[(gy * 8 + gx) % len(dims)]
            x0, y0 = gx * 65, gy * 66
            kind = (gy * 8 + gx) % 7
            if kind == 0:
                v = np.zeros((h, w), np.int64)                                     # all-zero block: nil stream
            elif kind == 1:
                v = rng.integers(-3, 4, (h, w))                                    # 2 bit planes
            elif kind == 2:
                v = rng.integers(-100, 101, (h, w))                                # 25 bit planes, few deep samples (the
                v[rng.integers(0, h), rng.integers(0, w)] = (1 << 24) + 12345      # slot bound is 2 bytes per sample)
            elif kind == 3:
                v = np.where(rng.random((h, w)) < 0.02, rng.integers(-5000, 5000, (h, w)), 0)   # sparse: run-length mode
            else:
                v = rng.integers(-900, 901, (h, w))
            plane[y0:y0 + h, x0:x0 + w] = v
            blocks.append((0, (gy + gx) % 4, x0, y0, w, h))
    blk = np.array(blocks, dtype=ent.BLOCK_DTYPE)
    stream, offs, lens, nb = ent.encode_blocks(0, [plane], blk, ctx=ctx)
    for j, (_, band, x0, y0, w, h) in enumerate(blocks):
        want, wnb = oracle.t1_encode(plane[y0:y0 + h, x0:x0 + w], w, h, band)
        got = stream[int(offs[j]):int(offs[j]) + int(lens[j])]
        assert int(lens[j]) == want.size and int(nb[j]) == wnb, (j, env)
        assert np.array_equal(got, want), (j, env)


@pytest.mark.parametrize("general", ["0", "1"])
def test_t1_batch_both_decoder_kernels(ent, oracle, general, monkeypatch):
    """Blocks of mixed size (some wider than 64) through T1.Decode on the <= 64x64 kernel + the general kernel (default)
    and on the general kernel alone (J2K_T1_DEC_GENERAL=1): encoder output and arbitrary bytes, against the oracle."""
    from j2kgfx import Context
    monkeypatch.setenv("J2K_T1_DEC_GENERAL", general)              # read when a context is created
    ctx = Context(0)
    rng = np.random.default_rng(5)
    dims = [(64, 64), (33, 64), (64, 7), (5, 3), (1, 1), (9, 64), (128, 32), (16, 16), (64, 64), (70, 9), (8, 8), (17, 5)]
    blocks, streams, nbs, wants = [], [], [], []
    for j, (w, h) in enumerate(dims * 2):
        band = j % 4
        if j % 3 == 2:                                             # not encoder output
            g = rng.integers(0, 256, int(rng.integers(0, 300))).astype(np.uint8)
            nb = int(rng.integers(1, 14))
        else:
            x = rng.integers(-700, 701, (h, w)).astype(np.int32)
            g, nb = oracle.t1_encode(x, w, h, band)
        blocks.append((0, band, 0, 0, w, h)); streams.append(np.asarray(g, np.uint8)); nbs.append(nb)
        wants.append(oracle.t1_decode(streams[-1], nb, band, w, h).reshape(h, w))
    blk = np.array(blocks, dtype=ent.BLOCK_DTYPE)
    lens = np.array([g.size for g in streams], np.uint32)
    offs = np.concatenate(([0], np.cumsum(lens)[:-1])).astype(np.uint64)
    got = ent.decode_blocks(0, np.concatenate(streams), offs, lens, np.array(nbs, np.uint8), blk, ctx=ctx)
    for j in range(len(blocks)):
        assert np.array_equal(got[j], wants[j]), (j, blocks[j], general)


@pytest.mark.parametrize("lanes", [2, 1, 0])
def test_t1_batch_plane_stepped_decoder(ent, oracle, monkeypatch, lanes):
    """The plane-stepped T1.Decode (J2K_T1_DEC_SPLIT=1; lanes = 2: the whole decode of 64 blocks per wavefront in one launch, one block
    per lane on row masks, t1_lanes.inc; lanes = 1: the same passes as launches per plane; lanes = 0: round 2's step kernels; MagRef
    chains in lock step in all) on a batch of mixed block
    sizes against the oracle: encoder output, arbitrary bytes (streams that end early, 0xFF runs), zero-length streams,
    bit-plane counts from 0 to 40 (above 31 the one-launch kernel takes the block: bit 0 for p >= 32, t1.go:1291) and
    blocks wider than 64 (general kernel) in the same call."""
    from j2kgfx import Context
    monkeypatch.setenv("J2K_T1_DEC_SPLIT", "1")                    # read when a context is created
    monkeypatch.setenv("J2K_T1_DEC_LANES", str(lanes))
    ctx = Context(0)
    rng = np.random.default_rng(77)
    dims = [(64, 64), (33, 64), (64, 7), (5, 3), (1, 1), (9, 64), (128, 32), (16, 16), (64, 64), (70, 9), (8, 8), (17, 5), (64, 64), (40, 40)]
    blocks, streams, nbs, wants = [], [], [], []
    for j, (w, h) in enumerate(dims * 10):                          # 140 blocks: three groups of lanes
        band = j % 4
        kind = j % 5
        if kind == 2:                                              # not encoder output
            g = rng.integers(0, 256, int(rng.integers(0, 400))).astype(np.uint8)
            nb = int(rng.integers(0, 41))
        elif kind == 3:                                            # 0xFF-rich bytes, deep
            g = np.where(rng.random(int(rng.integers(1, 200))) < 0.5, 0xFF, rng.integers(0, 256)).astype(np.uint8)
            nb = int(rng.integers(25, 33))
        elif kind == 4 and j % 2:                                  # nothing at all
            g = np.zeros(0, np.uint8); nb = int(rng.integers(0, 6))
        else:
            x = rng.integers(-(1 << int(rng.integers(1, 20))), 1 << int(rng.integers(1, 20)), (h, w)).astype(np.int32)
            g, nb = oracle.t1_encode(x, w, h, band)
        blocks.append((0, band, 0, 0, w, h)); streams.append(np.asarray(g, np.uint8)); nbs.append(nb)
        wants.append(oracle.t1_decode(streams[-1], nb, band, w, h).reshape(h, w))
    blk = np.array(blocks, dtype=ent.BLOCK_DTYPE)
    lens = np.array([g.size for g in streams], np.uint32)
    offs = np.concatenate(([0], np.cumsum(lens)[:-1])).astype(np.uint64)
    got = ent.decode_blocks(0, np.concatenate(streams), offs, lens, np.array(nbs, np.uint8), blk, ctx=ctx)
    for j in range(len(blocks)):
        assert np.array_equal(got[j], wants[j]), (j, blocks[j], nbs[j])


HT_SHAPES = [(4, 4), (8, 8), (16, 16), (64, 64), (32, 32), (8, 5), (13, 9), (7, 4), (5, 8), (12, 16), (64, 7), (3, 16), (128, 128),
             (128, 32), (256, 16), (1024, 4), (512, 8), (128, 31), (252, 16), (1020, 3)]


@pytest.mark.parametrize("w,h", HT_SHAPES)
def test_ht_encode_decode_bytes(ent, oracle, w, h):
    rng = np.random.default_rng(w * 17 + h)
    for amp in (1, 3, 300, 40000):
        x = rng.integers(-amp, amp + 1, (h, w)).astype(np.int32)
        if amp == 3:
            x[rng.random((h, w)) < 0.7] = 0
        try:
            want = oracle.ht_encode(x, w, h)
        except ValueError:                       # the Go code would panic on this input
            from j2kgfx import J2KError
            enc = ent.NewHTEncoder(w, h); enc.SetData(x)
            with pytest.raises(J2KError):
                enc.Encode(0)
            continue
        enc = ent.NewHTEncoder(w, h); enc.SetData(x)
        got = enc.Encode(0)
        if want.size == 0:
            assert got is None
            continue
        assert got == bytes(want)
        dec = ent.NewHTDecoder(w, h).Decode(got, 0, 0)
        assert np.array_equal(dec.reshape(h, w), oracle.ht_decode(want, w, h))


@pytest.mark.parametrize("w,h", [(64, 64), (64, 61), (60, 64), (32, 32), (16, 9), (8, 8), (8, 1), (4, 4), (3, 5), (1, 1), (12, 3), (24, 24), (40, 17)])
def test_ht_encode_magsgn_stuffing(ent, oracle, w, h):
    """MagSgn segments full of 0xFF bytes (negative values of the form -(2^n - 1) are all-one fields): the encoder places every
    byte in parallel from the list of 0xFF positions -- runs of ones of every length and alignment, 0xFF as the last byte, as
    the byte before the last, segments shorter than a byte, and ordinary data in between (ht.go:1303-1341)"""
    rng = np.random.default_rng(w * 131 + h)
    cases = [np.full((h, w), -1), np.full((h, w), -3), np.full((h, w), -127), np.full((h, w), -(2 ** 20 - 1)), np.full((h, w), 1)]
    for p_neg in (0.5, 0.9, 0.99):
        for vals in ((-1, 1), (-1, -3, 2), (-1, -7, -15, 5, 0), (-255, -1, 0)):
            x = rng.choice(vals, size=(h, w))
            x[rng.random((h, w)) > p_neg] = rng.integers(-9, 10)
            cases.append(x)
    for run in (7, 8, 9, 15, 16, 17, 23, 30, 31, 64):                  # one run of `run` ones at every offset mod 16, zeros elsewhere is not
        for off in range(0, 16, 3):                                    # expressible (a field starts with its sign) -- use -1 / +1 fields
            x = np.ones((h, w), np.int64)
            flat = x[::4].reshape(-1)                                  # the coded rows, in coding order only for w % 8 == 0; good enough
            flat[off:off + run] = -1
            x[::4] = flat.reshape(x[::4].shape)
            cases.append(x)
    for x in cases:
        x = np.ascontiguousarray(x, dtype=np.int32)
        want = oracle.ht_encode(x, w, h)
        enc = ent.NewHTEncoder(w, h); enc.SetData(x)
        got = enc.Encode(0)
        assert (got is None and want.size == 0) or got == bytes(want)


def test_ht_decoder_edge_inputs(ent, oracle):
    """ht_test.go:126-147: nil, 1-byte and 2-byte inputs; plus bad SCUP and random payloads"""
    rng = np.random.default_rng(4)
    w = h = 16
    for data in (b"", b"\x00", b"\x00\x00", b"\xff\xff", bytes(rng.integers(0, 256, 64).astype(np.uint8))):
        got = ent.NewHTDecoder(w, h).Decode(data, 8, 0)
        assert np.array_equal(got.reshape(h, w), oracle.ht_decode(np.frombuffer(data, np.uint8), w, h))
    x = rng.integers(-50, 51, (h, w)).astype(np.int32)
    good = oracle.ht_encode(x, w, h)
    for _ in range(20):
        g = rng.integers(0, 256, good.size).astype(np.uint8)
        g[-2:] = good[-2:]
        got = ent.NewHTDecoder(w, h).Decode(bytes(g), 8, 0)
        assert np.array_equal(got.reshape(h, w), oracle.ht_decode(g, w, h))


@pytest.mark.parametrize("coder", [0, 1])
@pytest.mark.parametrize("W,H,Cn,nres,cb", [(64, 64, 3, 3, 16), (128, 96, 1, 4, 32), (100, 75, 3, 6, 64), (512, 512, 3, 3, 256)])
def test_encode_frame_matches_encoder(oracle, coder, W, H, Cn, nres, cb):
    """encoder.preprocess + encodeTile (sequential path) on a single tile: coefficients in place and
    the concatenated block bytes, lens and numbps in job order."""
    from j2kgfx.codec import FramePlan
    rng = np.random.default_rng(W + H + coder)
    yy, xx = np.mgrid[0:H, 0:W]
    base = [xx * 255 // W, yy * 255 // H, (xx + yy) * 127 // max(W, H)]          # jpeg2000_test.go:340-352
    planes = [np.clip(base[c % 3] + rng.integers(-16, 17, (H, W)), 0, 255).astype(np.int32) for c in range(Cn)]
    want_coeff = oracle.preprocess(planes, W, H, 8, True, nres)
    want_bytes, want_lens, want_nb = oracle.encode_tile_blocks(want_coeff, W, H, nres, cb, cb, coder)
    plan = FramePlan(W, H, Cn, precision=8, lossless=True, num_resolutions=nres, cb=(cb, cb), coder=coder)
    host = [p.copy().reshape(-1) for p in planes]
    res = plan.encode_frame(host)
    for c in range(Cn):
        assert np.array_equal(host[c].reshape(H, W), want_coeff[c])               # like e.componentData
    assert np.array_equal(res["lens"], want_lens)
    assert np.array_equal(res["numbps"], want_nb)
    assert bytes(res["bytes"]) == bytes(want_bytes)


@pytest.mark.parametrize("coder", [0, 1])
def test_plan_decode_blocks_matches_decoders(oracle, coder):
    """tcd.TileDecoder.DecodeCodeBlock for every job of a 512x512 RGB tile (bench-like content): the decoded
    coefficients equal NewT1(...).Decode / a fresh NewHTDecoder(...).Decode on the same bytes.  For HT this
    content contains blocks whose decoded u exceeds 32 (the reference's bit counter wraps, ht.go:515-519)."""
    import torch
    from j2kgfx.codec import FramePlan
    W = H = 512 if coder == 1 else 128
    rng = np.random.default_rng(0x4A324B30)
    yy, xx = np.mgrid[0:H, 0:W]
    base = np.stack([xx * 255 // W, yy * 255 // H, (xx + yy) * 127 // W])
    frame = np.clip(base + rng.integers(-16, 17, size=(3, H, W)), 0, 255).astype(np.int32)
    plan = FramePlan(W, H, 3, precision=8, lossless=True, num_resolutions=6, cb=(64, 64), coder=coder)
    d = torch.from_numpy(frame).to(plan.device)
    torch.cuda.synchronize()
    coeff = plan.forward(d)
    slots, lens, nb = plan.encode_blocks(coeff)
    offs, stream = plan.compact(slots, lens)
    dec = plan.decode_blocks(stream, offs, lens, nb)
    plan.ctx.sync()
    n = int(plan.info.blocks)
    ho, hl, hn = offs.cpu().numpy(), lens.cpu().numpy(), nb.cpu().numpy()
    hs, hd = stream.cpu().numpy(), dec.cpu().numpy()
    blocks, doffs = plan.blocks(), plan.decoded_offsets()
    big = 0
    for j in range(n):
        w, h, band = int(blocks[j]["w"]), int(blocks[j]["h"]), int(blocks[j]["band"])
        chunk = hs[int(ho[j]):int(ho[j]) + int(hl[j])]
        want = oracle.ht_decode(chunk, w, h) if coder == 1 else oracle.t1_decode(chunk, int(hn[j]), band, w, h)
        got = hd[int(doffs[j]):int(doffs[j]) + w * h].reshape(h, w)
        assert np.array_equal(got, want), j
        big += int(np.abs(want).max() >= (1 << 31) - 1) if want.size else 0


@pytest.mark.parametrize("W,H,cb,corrupt", [(512, 512, 64, False), (200, 150, 32, False), (264, 75, 16, False), (512, 512, 64, True)])
def test_plan_decode_coded_rows_only(oracle, W, H, cb, corrupt):
    """j2k_plan_set_decode_coded_rows_only: the pooled HTDecoder's behaviour (ht.go:1393-1429 -- its data slice is not cleared
    between blocks; the decoder writes row y of every 4-row stripe only, ht.go:589-711).  Into a buffer poisoned with a
    sentinel: the coded rows equal the fresh decoder's (orc_ht_decode) on every block, every other row keeps the sentinel;
    into a zeroed buffer the result IS the fresh decoder's.  `corrupt`: streams with broken SCUP bytes / truncated blocks --
    the invalid-stream and serial-decoder paths must clear the coded rows themselves."""
    import torch
    from j2kgfx.codec import FramePlan
    rng = np.random.default_rng(W + H + cb)
    yy, xx = np.mgrid[0:H, 0:W]
    base = np.stack([xx * 255 // W, yy * 255 // H, (xx + yy) * 127 // max(W, H)])
    frame = np.clip(base + rng.integers(-16, 17, size=(3, H, W)), 0, 255).astype(np.int32)
    plan = FramePlan(W, H, 3, precision=8, lossless=True, num_resolutions=5, cb=(cb, cb), coder=1)
    coeff = plan.forward(torch.from_numpy(frame).to(plan.device))
    stream, offs, lens, nb = plan.encode_stream(coeff)
    plan.ctx.sync()
    n = int(plan.info.blocks)
    ho, hl = offs.cpu().numpy(), lens.cpu().numpy()
    hs = stream.cpu().numpy().copy()
    if corrupt:
        for j in range(0, n, 3):                              # every third block: SCUP out of range, zero, or bytes flipped
            a, l = int(ho[j]), int(hl[j])
            if l < 4:
                continue
            k = (j // 3) % 3
            if k == 0: hs[a + l - 1] = 0xFF; hs[a + l - 2] |= 0x0F
            elif k == 1: hs[a + l - 1] = 0; hs[a + l - 2] &= 0xF0
            else: hs[a + l // 2: a + l // 2 + 8] ^= 0x5A
        stream = torch.from_numpy(hs).to(plan.device)
    fresh = plan.decode_blocks(stream, offs, lens, nb)
    plan.ctx.sync()
    plan.set_decode_coded_rows_only(True)
    SENT = -123456789
    poisoned = torch.full((int(plan.info.decoded_elems),), SENT, dtype=torch.int32, device=plan.device)
    zeroed = torch.zeros_like(poisoned)
    plan.decode_blocks(stream, offs, lens, nb, decoded=poisoned)
    plan.decode_blocks(stream, offs, lens, nb, decoded=zeroed)
    plan.ctx.sync()
    plan.set_decode_coded_rows_only(False)
    again = plan.decode_blocks(stream, offs, lens, nb)
    plan.ctx.sync()
    hf, hp, hz, ha = fresh.cpu().numpy(), poisoned.cpu().numpy(), zeroed.cpu().numpy(), again.cpu().numpy()
    blocks, doffs = plan.blocks(), plan.decoded_offsets()
    for j in range(n):
        w, h = int(blocks[j]["w"]), int(blocks[j]["h"])
        o = int(doffs[j])
        want = oracle.ht_decode(hs[int(ho[j]):int(ho[j]) + int(hl[j])], w, h)
        assert np.array_equal(hf[o:o + w * h].reshape(h, w), want), ("fresh", j)
        assert np.array_equal(ha[o:o + w * h].reshape(h, w), want), ("switched off again", j)
        got = hp[o:o + w * h].reshape(h, w)
        assert np.array_equal(got[0::4], want[0::4]), ("coded rows", j)
        for r in (1, 2, 3):
            assert (got[r::4] == SENT).all(), ("untouched rows", j, r)
        assert np.array_equal(hz[o:o + w * h].reshape(h, w), want), ("zeroed buffer == fresh decoder", j)


def test_standalone_mq_and_raw_coders(ent, oracle):
    """MQEncoder / MQDecoder / RawEncoder / RawDecoder as batch calls, byte-exact against the oracle."""
    rng = np.random.default_rng(14)
    for n, skew in [(0, 0.5), (1, 0.5), (50, 0.5), (5000, 0.5), (20000, 0.9), (20000, 0.02)]:
        ctxs = rng.integers(0, 19, n).astype(np.uint8)
        decs = (rng.random(n) < skew).astype(np.uint8)
        want = oracle.mq_encode(ctxs, decs)
        got = ent.mq_encode(ctxs, decs)
        assert got == want.tobytes()
        assert np.array_equal(ent.mq_decode(got, ctxs), oracle.mq_decode(want, ctxs))
        junk = rng.integers(0, 256, max(n // 4, 3)).astype(np.uint8)                # arbitrary bytes decode the same way
        assert np.array_equal(ent.mq_decode(junk, ctxs), oracle.mq_decode(junk, ctxs))
        bits = (rng.random(n) < skew).astype(np.uint8)
        wr = oracle.raw_encode(bits)
        assert ent.raw_encode(bits) == wr.tobytes()
        assert np.array_equal(ent.raw_decode(wr, n), bits)
        assert np.array_equal(ent.raw_decode(junk, n + 5), oracle.raw_decode(junk, n + 5))
    e = ent.NewMQEncoder()
    for c, d in [(18, 1), (17, 0), (0, 1), (9, 1)]:
        e.Encode(c, d)
    assert e.Flush() == oracle.mq_encode([18, 17, 0, 9], [1, 0, 1, 1]).tobytes()
    from j2kgfx import J2KError
    with pytest.raises(J2KError):
        ent.mq_encode([19], [0])                                                    # contexts [19]: Go index panic


@pytest.mark.parametrize("reversible", [True, False])
@pytest.mark.parametrize("htj2k", [False, True])
def test_tcd_tile_encoder_decoder_mirror(oracle, reversible, htj2k):
    """internal/tcd's hot-path methods through the Python mirror: forward DWT, one code-block out and back, inverse DWT."""
    from j2kgfx import tcd
    rng = np.random.default_rng(5)
    w, h, levels = 96, 80, 3
    src = rng.integers(-300, 300, w * h).astype(np.int32)
    tc = tcd.TileComponent(0, 0, w, h, src.copy())
    enc = tcd.TileEncoder(levels, reversible, htj2k)
    enc.ApplyForwardDWT(tc)
    assert np.array_equal(tc.Data, oracle.tcd_forward_dwt(src.copy(), w, h, levels, reversible).reshape(-1))
    block = tc.Data.reshape(h, w)[:32, :32].copy()
    cb = tcd.CodeBlock(0, 0, 32, 32)
    enc.EncodeCodeBlock(cb, block, 1)
    want = oracle.ht_encode(block, 32, 32) if htj2k else oracle.t1_encode(block, 32, 32, 1)[0]
    assert (cb.Data or b"") == bytes(want)
    dec = tcd.TileDecoder(levels, reversible, htj2k)
    dec.DecodeCodeBlock(cb, 1)
    wantd = oracle.ht_decode(want, 32, 32) if htj2k else oracle.t1_decode(want, cb.TotalBitPlanes, 1, 32, 32)
    assert np.array_equal(cb.Coefficients.reshape(32, 32), wantd)
    back = tcd.TileComponent(0, 0, w, h, tc.Data.copy())
    dec.ApplyInverseDWT(back)
    assert np.array_equal(back.Data, oracle.tcd_inverse_dwt(tc.Data.copy(), w, h, levels, reversible).reshape(-1))


def test_plan_encode_fault_is_reported_at_sync():
    """The asynchronous plan calls fail loudly: a coefficient on which the Go HT encoder never returns (MinInt32,
    ht.go:1159) is reported by the next synchronisation, once, and the context keeps working afterwards."""
    import torch
    from j2kgfx import J2KError
    from j2kgfx.codec import FramePlan
    plan = FramePlan(64, 64, 1, precision=8, lossless=True, num_resolutions=2, cb=(64, 64), coder=1)
    coeff = plan.alloc_coeff()
    coeff.zero_()
    coeff[5] = -2147483648                                            # alone it would be an all-zero block (its int32 magnitude is negative)
    coeff[6] = 3
    plan.encode_blocks(coeff)
    with pytest.raises(J2KError):
        plan.ctx.sync()
    coeff[5] = 7
    slots, lens, nb = plan.encode_blocks(coeff)
    plan.ctx.sync()                                                   # the fault word was cleared when it was reported
    assert int(lens[0].item()) > 0


@pytest.mark.parametrize("W,H,tile,cb,coder", [(3840, 2160, 512, 64, 1), (200, 96, 0, 32, 1), (100, 75, 64, 16, 1), (512, 512, 0, 256, 1),
                                               (256, 128, 128, 64, 0)])
@pytest.mark.parametrize("fuse", ["0", "1"])
def test_encode_stream_equals_encode_blocks_plus_compact(W, H, tile, cb, coder, fuse, monkeypatch):
    """j2k_plan_encode_stream (one kernel with a look-back for HT blocks up to 64x64, the two-step path otherwise) must
    produce the stream, offsets, lengths and bit-plane counts of encode_blocks + compact -- also when launched again
    and again on the same plan (the look-back words are epoch-tagged, never cleared)."""
    import torch
    from j2kgfx import Context
    from j2kgfx.codec import FramePlan
    monkeypatch.setenv("J2K_FUSE_COMPACT", fuse)                   # read when a context is created
    rng = np.random.default_rng(W + cb)
    frame = torch.from_numpy(rng.integers(0, 256, size=(3, H, W)).astype(np.int32))
    plan = FramePlan(W, H, 3, precision=8, lossless=True, num_resolutions=6, cb=(cb, cb), tile=(tile, tile), coder=coder,
                     ctx=Context(0))
    d = frame.to(plan.device)
    coeff = plan.forward(d)
    slots, lens, nb = plan.encode_blocks(coeff)
    offs, stream = plan.compact(slots, lens)
    plan.ctx.sync()
    n = int(plan.info.blocks)
    tot = int(offs[n].item())
    for rep in range(3):
        s2, o2, l2, n2 = plan.encode_stream(coeff)
        plan.ctx.sync()
        assert int(o2[n].item()) == tot
        assert torch.equal(o2[:n + 1], offs[:n + 1]) and torch.equal(l2[:n], lens[:n]) and torch.equal(n2[:n], nb[:n])
        assert torch.equal(s2[:tot], stream[:tot]), rep
        coeff = plan.forward(d)                                    # same values; keeps the launches back to back
    plan.ctx.sync()                                                # nothing in flight when the tensors go out of scope


@pytest.mark.parametrize("W,H,tile,cb,coder", [(3840, 2160, 512, 64, 1), (200, 96, 0, 32, 1), (100, 75, 64, 16, 1), (512, 512, 0, 256, 1),
                                               (256, 128, 128, 64, 0), (64, 64, 0, 64, 1)])
def test_pack_unpack_stream_round_trip(W, H, tile, cb, coder):
    """Transport form of a stream for the multi-GPU gather: unpack(pack(x)) must give back the dense stream, the offsets,
    the lengths and the bit-plane counts byte for byte (on ANOTHER plan of the same geometry, as the root has), the pack
    must be smaller than the stream by the MEL zero runs (HT), and a pack of another geometry must be refused."""
    import torch
    from j2kgfx import Context, J2KError
    from j2kgfx.codec import FramePlan
    rng = np.random.default_rng(W * 3 + cb + coder)
    frame = torch.from_numpy(rng.integers(0, 256, size=(3, H, W)).astype(np.int32))
    kw = dict(precision=8, lossless=True, num_resolutions=6, cb=(cb, cb), tile=(tile, tile), coder=coder)
    plan = FramePlan(W, H, 3, ctx=Context(0), **kw)
    root = FramePlan(W, H, 3, ctx=Context(0), **kw)
    d = frame.to(plan.device)
    coeff = plan.forward(d)
    stream, offs, lens, nb = plan.encode_stream(coeff)
    pack = plan.pack_stream(stream, offs, lens, nb)
    plan.ctx.sync()
    torch.cuda.synchronize()
    n = int(plan.info.blocks)
    tot = int(offs[n].item())
    pbytes = int(pack[:8].view(torch.int64)[0].item())
    assert 0 < pbytes <= plan.pack_bound()
    sent = pack[:pbytes].clone()                                   # what would travel (a torch op on torch's stream: the
    s2, o2, l2, n2 = root.unpack_stream(sent)                      #  wrapper orders the library stream behind it)
    root.ctx.sync()
    assert torch.equal(o2[:n + 1], offs[:n + 1]) and torch.equal(l2[:n], lens[:n]) and torch.equal(n2[:n], nb[:n])
    assert torch.equal(s2[:tot], stream[:tot])
    if coder == 1:
        blocks = plan.blocks()
        mel = sum(max(64, 2 * int(b["w"]) * int(b["h"])) // 4 for b, ln in zip(blocks, lens.cpu().numpy()[:n]) if ln)
        payload = int(pack[8:16].view(torch.int64)[0].item())
        assert payload == tot - mel
    # several packs in one launch (what the root of an N-GPU gather does per frame slot)
    outs = [(root.empty(root.info.bytes_cap, torch.uint8), root.empty(n + 1, torch.int64), root.empty(n, torch.int32),
             root.empty(n, torch.uint8)) for _ in range(3)]
    sent2 = sent.clone()
    root.unpack_streams([sent, sent2, sent], outs)
    root.ctx.sync()
    for s3, o3, l3, n3 in outs:
        assert torch.equal(o3[:n + 1], offs[:n + 1]) and torch.equal(l3[:n], lens[:n]) and torch.equal(n3[:n], nb[:n])
        assert torch.equal(s3[:tot], stream[:tot])
    other = FramePlan(W + 64, H, 3, ctx=Context(0), **kw)
    other.unpack_stream(sent)                                      # temporaries: record_stream keeps them alive until the kernel ran
    with pytest.raises(J2KError):
        other.ctx.sync()
    # ---- a pack is foreign input: truncated packs and wrapping offsets are refused, nothing is written ----
    with pytest.raises(J2KError):                                  # shorter than its own header: refused at the call
        root.unpack_stream(sent[:64].clone())
    cut = sent[:pbytes - 1024].clone() if pbytes > 4096 else None  # header intact, payload cut: the claimed size does not fit
    if cut is not None:
        root.unpack_stream(cut)
        with pytest.raises(J2KError):
            root.ctx.sync()
    # per-block offset of the first non-empty block set to 2^64 - 8: `off + len > end` would wrap and pass
    lens_h = lens.cpu().numpy()[:n]
    j = int(np.flatnonzero(lens_h)[0])
    # section layout (compact.hip pack_layout): 64-byte header | lens u32[n] | maglens u32[n] | numbps u8[n] | offs u64[n+1] |
    # toffs u64[n+1] | payload, every section 16-byte aligned
    a16 = lambda x: (x + 15) & ~15
    offs_o = a16(a16(a16(64 + 4 * n) + 4 * n) + n)
    toffs_o = a16(offs_o + 8 * (n + 1))
    assert a16(toffs_o + 8 * (n + 1)) + int(pack[8:16].view(torch.int64)[0].item()) <= pbytes
    evil = sent.clone()
    offs_sec = evil[offs_o:offs_o + 8 * (n + 1)].view(torch.int64)
    offs_sec[j] = -8
    guard = torch.full((root.info.bytes_cap + 4096,), 0x5A, dtype=torch.uint8, device=root.device)
    out_stream = guard[2048:2048 + root.info.bytes_cap]            # poisoned on both sides: a write before / after the stream shows
    o4, l4, n4 = root.empty(n + 1, torch.int64), root.empty(n, torch.int32), root.empty(n, torch.uint8)
    root.unpack_stream(evil, out_stream, o4, l4, n4)
    with pytest.raises(J2KError):
        root.ctx.sync()
    assert bool((guard[:2048] == 0x5A).all()) and bool((guard[2048 + root.info.bytes_cap:] == 0x5A).all())
    evil = sent.clone()
    toffs_sec = evil[toffs_o:toffs_o + 8 * (n + 1)].view(torch.int64)
    toffs_sec[j] = -8                                              # transport offset wraps: would read before the payload
    root.unpack_stream(evil, out_stream, o4, l4, n4)
    with pytest.raises(J2KError):
        root.ctx.sync()
    # ---- pack_stream only packs the outputs of the LAST encode_stream on the plan ----
    if coder == 1:
        stream_b, offs_b, lens_b, nb_b = plan.encode_stream(coeff)
        with pytest.raises(J2KError):
            plan.pack_stream(stream, offs, lens, nb)               # the older buffers: refused instead of mixing two frames' arrays
        plan.pack_stream(stream_b, offs_b, lens_b, nb_b)
        plan.ctx.sync()


def test_stage_calls_with_temporaries():
    """every stage runs on the library's own stream; the wrappers record that stream on the tensors they are
    given and order it behind torch's, so chained calls on temporaries need no synchronisation in between."""
    import torch
    from j2kgfx import Context
    from j2kgfx.codec import FramePlan
    rng = np.random.default_rng(5)
    W, H = 1024, 512
    plan = FramePlan(W, H, 3, precision=8, lossless=True, num_resolutions=6, cb=(64, 64), tile=(512, 512), coder=1, ctx=Context(0))
    frame = torch.from_numpy(rng.integers(0, 256, size=(3, H, W)).astype(np.int32)).to(plan.device)
    coeff = plan.forward(frame)
    stream, offs, lens, nb = plan.encode_stream(coeff)
    want = plan.decode_blocks(stream, offs, lens, nb)
    plan.ctx.sync()
    want = want.clone()
    for _ in range(8):
        # producers on torch's stream (frame * 1), every intermediate a temporary, allocations in between that would reuse
        # a freed block at once if the library stream were not recorded on it
        got = plan.decode_blocks(*plan.encode_stream(plan.forward(frame * 1)))
        junk = [torch.full((plan.info.coeff_elems,), 7, dtype=torch.int32, device=plan.device) for _ in range(3)]
        plan.ctx.sync()
        assert torch.equal(got, want)
        del junk


def _frame_vs_oracle(oracle, frame, nres, cb, coder, prec):
    import torch
    from j2kgfx.codec import FramePlan
    Cn, H, W = frame.shape
    want = oracle.preprocess([frame[c] for c in range(Cn)], W, H, prec, True, nres)
    wb, wl, wn = oracle.encode_tile_blocks(want, W, H, nres, cb, cb, coder)
    plan = FramePlan(W, H, Cn, precision=prec, lossless=True, num_resolutions=nres, cb=(cb, cb), coder=coder)
    coeff = plan.forward(torch.from_numpy(frame).to(plan.device))
    stream, offs, lens, nb = plan.encode_stream(coeff)
    decoded = plan.decode_blocks(stream, offs, lens, nb)
    plan.ctx.sync()
    n = int(plan.info.blocks)
    assert np.array_equal(lens.cpu().numpy()[:n].astype(np.uint32), wl)
    assert bytes(stream.cpu().numpy()[:int(offs[n].item())]) == bytes(wb)
    blocks = plan.blocks(); doffs = plan.decoded_offsets(); dh = decoded.cpu().numpy()
    pos = 0
    for j in range(n):
        w_, h_, band = int(blocks[j]["w"]), int(blocks[j]["h"]), int(blocks[j]["band"])
        chunk = wb[pos:pos + int(wl[j])]; pos += int(wl[j])
        ref = oracle.ht_decode(chunk, w_, h_) if coder == 1 else oracle.t1_decode(chunk, int(wn[j]), band, w_, h_)
        assert np.array_equal(dh[int(doffs[j]):int(doffs[j]) + w_ * h_].reshape(h_, w_), ref), j


def test_fuzz_regression_walk_with_mixed_block_sizes(oracle):
    """tools/fuzz_gpu.py, seed 1: a 40 x 33 frame with 16 x 16 code-blocks has blocks of many sizes in one walking wavefront,
    some of which leave early (all-zero blocks); the wavefront's trip count was reduced over a wave with holes and cut the
    walk of a 16 x 16 block short (its last two coded rows came out zero)"""
    import os
    frame = np.load(os.path.join(os.path.dirname(__file__), "golden", "fuzz_walk_mixed_blocks.npy"))
    _frame_vs_oracle(oracle, frame, 6, 16, 1, 8)


def test_fuzz_regression_deep_mq_blocks_fit_the_references_buffer(oracle):
    """tools/fuzz_gpu.py, seed 1: 16-bit noise after five lifting levels makes 64 x 64 MQ blocks of ~9.3 KB -- more than
    2wh + 1024 = 9216 but inside the 16384-byte floor of the reference's mqBuf (t1_fast5.go:47-56): they must be coded"""
    rng = np.random.default_rng(5)
    frame = rng.integers(0, 65536, (3, 128, 264)).astype(np.int32)
    _frame_vs_oracle(oracle, frame, 1, 64, 0, 16)
    from j2kgfx import entropy
    assert entropy.block_bound(0, 64, 64) == 16384 and entropy.block_bound(0, 128, 128) == 2 * 128 * 128 + 1024
