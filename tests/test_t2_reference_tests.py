"""CPU (host-only C ABI, no device): the reference's Tier-2 / tile-geometry layer.

Every expectation the reference's own tests state numerically -- internal/tcd/t2_test.go and tcd_test.go, cited per test -- is
checked against BOTH the Python oracle (oracle/t2ref.py: pins the oracle) and the product (csrc/t2.cpp through j2kgfx.t2).
Then the two are compared on randomised inputs, byte for byte and field for field."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "go-jpeg2000_amd"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

import t2ref                                   # noqa: E402
from j2kgfx import J2KError, t2                # noqa: E402
from j2kgfx import _lib                        # noqa: E402


def create_test_precincts(nc, nr, npc):        # t2_test.go:11-21
    return [[[npc] for _ in range(nr)] for _ in range(nc)]


# ---- the two implementations behind one face --------------------------------------------------------------------------
class Prod:
    name = "product"

    @staticmethod
    def seq(nc, nr, nl, prec, order):
        it = t2.PacketIterator(nc, nr, nl, prec, order)
        out = []
        while True:
            p, ok = it.Next()
            if not ok:
                return out
            out.append(p)

    @staticmethod
    def encode(packets):
        """packets: list of (bands, layer, sop, eph, incl_w, imsb_w), bands = [[(data, incl, zbp, passes)]] -> bytes of one encoder"""
        enc = t2.PacketEncoder()
        for bands, layer, sop, eph, iw, mw in packets:
            pr = t2.Precinct([[t2.CodeBlock(*cb) for cb in b] for b in bands], t2.TagTree(iw, 1), t2.TagTree(mw, 1))
            enc.EncodePacket(pr, layer, sop, eph)
        return bytes(enc.buf)

    @staticmethod
    def decode(data, packets):
        """packets: list of (bands, layer, sop, eph): -> [(position, [[(data, incl, zbp, passes)]])] per packet; error -> 'err'"""
        dec = t2.PacketDecoder(data)
        out = []
        for bands, layer, sop, eph in packets:
            pr = t2.Precinct([[t2.CodeBlock(*cb) for cb in b] for b in bands])
            try:
                dec.DecodePacket(pr, layer, sop, eph)
            except J2KError:
                out.append("err")
                break
            out.append((dec.Position(), [[(cb.Data or b"", cb.IncludedInLayers, cb.ZeroBitPlanes, cb.Passes) for cb in b] for b in pr.CodeBlocks]))
        return out

    @staticmethod
    def tagtree(w, h):
        tr = t2.TagTree(w, h)
        return tr.levels, tr.level_sizes

    init_tile = staticmethod(t2.init_tile)


class Orc:
    name = "oracle"

    @staticmethod
    def seq(nc, nr, nl, prec, order):
        return t2ref.PacketIterator(nc, nr, nl, prec, order).all()

    @staticmethod
    def encode(packets):
        enc = t2ref.PacketEncoder()
        for bands, layer, sop, eph, iw, mw in packets:
            pr = t2ref.Precinct([[t2ref.CodeBlock(*cb) for cb in b] for b in bands], iw, mw)
            enc.encode_packet(pr, layer, sop, eph)
        return bytes(enc.out)

    @staticmethod
    def decode(data, packets):
        dec = t2ref.PacketDecoder(data)
        out = []
        for bands, layer, sop, eph in packets:
            pr = t2ref.Precinct([[t2ref.CodeBlock(*cb) for cb in b] for b in bands])
            try:
                dec.decode_packet(pr, layer, sop, eph)
            except t2ref.EOF:
                out.append("err")
                break
            out.append((dec.pos, [[(bytes(cb.data or b""), cb.included_in_layers, cb.zero_bit_planes, cb.num_passes) for cb in b] for b in pr.code_blocks]))
        return out

    @staticmethod
    def tagtree(w, h):
        tr = t2ref.TagTree(w, h)
        return tr.levels, tr.level_sizes

    init_tile = staticmethod(t2ref.init_tile)


IMPLS = [Orc, Prod]
ids = [i.name for i in IMPLS]


# ---- PacketIterator (t2_test.go:64-279, 784-1040, 1205-1257, 1511-1615) --------------------------------------------------
EXPECTED_2x2x2 = {   # (layer, resolution, component, precinct), t2_test.go:71-80, 107-116, 136-145, 164-173, 192-201
    t2ref.LRCP: [(0, 0, 0, 0), (0, 0, 1, 0), (0, 1, 0, 0), (0, 1, 1, 0), (1, 0, 0, 0), (1, 0, 1, 0), (1, 1, 0, 0), (1, 1, 1, 0)],
    t2ref.RLCP: [(0, 0, 0, 0), (0, 0, 1, 0), (1, 0, 0, 0), (1, 0, 1, 0), (0, 1, 0, 0), (0, 1, 1, 0), (1, 1, 0, 0), (1, 1, 1, 0)],
    t2ref.RPCL: [(0, 0, 0, 0), (1, 0, 0, 0), (0, 0, 1, 0), (1, 0, 1, 0), (0, 1, 0, 0), (1, 1, 0, 0), (0, 1, 1, 0), (1, 1, 1, 0)],
    t2ref.PCRL: [(0, 0, 0, 0), (1, 0, 0, 0), (0, 1, 0, 0), (1, 1, 0, 0), (0, 0, 1, 0), (1, 0, 1, 0), (0, 1, 1, 0), (1, 1, 1, 0)],
    t2ref.CPRL: [(0, 0, 0, 0), (1, 0, 0, 0), (0, 1, 0, 0), (1, 1, 0, 0), (0, 0, 1, 0), (1, 0, 1, 0), (0, 1, 1, 0), (1, 1, 1, 0)],
}


@pytest.mark.parametrize("impl", IMPLS, ids=ids)
@pytest.mark.parametrize("order", sorted(EXPECTED_2x2x2))
def test_packet_iterator_orders(impl, order):
    got = impl.seq(2, 2, 2, create_test_precincts(2, 2, 1), order)
    assert got == EXPECTED_2x2x2[order]            # and "no more packets after iteration complete" (t2_test.go:93-97)


@pytest.mark.parametrize("impl", IMPLS, ids=ids)
def test_packet_iterator_precincts_and_counts(impl):
    assert [p[3] for p in impl.seq(1, 1, 1, create_test_precincts(1, 1, 2), t2ref.LRCP)] == [0, 1]     # t2_test.go:241-262
    assert impl.seq(2, 2, 1, [], t2ref.LRCP)[0] == (0, 0, 0, 0)                                         # empty table: one precinct each (:784-794)
    assert impl.seq(1, 1, 1, create_test_precincts(1, 1, 1), t2ref.LRCP) == [(0, 0, 0, 0)]              # :796-814
    assert impl.seq(2, 2, 2, create_test_precincts(2, 2, 1), 99) == []                                  # unknown order (:1040-1051)
    for order in range(5):                                                                              # :888-1038: every order visits L*R*C*P packets
        for (nc, nr, nl, npc) in [(1, 1, 1, 1), (2, 3, 2, 1), (3, 2, 1, 2), (2, 2, 3, 3)]:
            assert len(impl.seq(nc, nr, nl, create_test_precincts(nc, nr, npc), order)) == nc * nr * nl * npc, (order, nc, nr, nl, npc)
    # maxPrecincts = 4 for {{2},{3}},{{1},{4}} (:264-279): PCRL then walks precincts 0..3 for EVERY (component, resolution)
    seq = impl.seq(2, 2, 1, [[[2], [3]], [[1], [4]]], t2ref.PCRL)
    assert len(seq) == 4 * 2 * 2 and max(p[3] for p in seq) == 3


@pytest.mark.parametrize("impl", IMPLS, ids=ids)
def test_packet_iterator_reset(impl):
    if impl is Prod:                                                                                     # t2_test.go:214-239
        it = t2.PacketIterator(2, 2, 2, create_test_precincts(2, 2, 2), t2.LRCP)
        for _ in range(4):
            assert it.Next()[1]
        it.Reset()
        assert it.Next() == ((0, 0, 0, 0), True)
    else:
        it = t2ref.PacketIterator(2, 2, 2, create_test_precincts(2, 2, 2), t2ref.LRCP)
        for _ in range(4):
            assert it.next() is not None
        it.reset()
        assert it.next() == (0, 0, 0, 0)


# ---- PacketEncoder / PacketDecoder (t2_test.go:401-560, 589-782, 816-886, 1053-1178, 1259-1727) --------------------------
def one(data=None, incl=0, zbp=0, passes=0):
    return [[(data, incl, zbp, passes)]]


@pytest.mark.parametrize("impl", IMPLS, ids=ids)
def test_encode_packet_markers_and_presence(impl):
    assert impl.encode([(one(None, 10), 0, False, False, 1, 1)]) == b"\x00"                 # empty packet: presence bit 0, padded (:401-420)
    out = impl.encode([([[]], 5, True, False, 1, 1)])
    assert out[:6] == bytes([0xFF, 0x91, 0x00, 0x04, 0x00, 0x05])                           # SOP, Lsop = 4, Nsop = layer (:449-479)
    assert b"\xff\x92" in impl.encode([([[]], 0, False, True, 1, 1)])                       # EPH (:481-507)
    assert len(impl.encode([([[]], 0, True, True, 1, 1)])) >= 8                             # both (:509-528)
    out = impl.encode([(one(b"\xaa\xbb\xcc", 0, 2, 1), 0, False, False, 1, 1)])             # with data (:422-447)
    # 1 presence | 1 inclusion (value 0) | 001 zero bit-planes = 2 | 0 one pass | 010 length bits = 2 | 11 length = 3 -> 1100 1001 | 011x xxxx
    assert out == bytes([0b11001001, 0b01100000]) + b"\xaa\xbb\xcc"


@pytest.mark.parametrize("impl", IMPLS, ids=ids)
def test_pass_count_length_and_tag_value_round_trips(impl):
    """encodeNumPasses / decodeNumPasses {1..6, 10, 36, 37, 50} (:530-560, 589-630, 1180-1203), encodeLength / decodeLength
    {0, 1, 10, 63, 100, 127} (:562-587, 632-667, 1531-1546), tag-tree values {0, 1, 5, 10} (:710-746, 1463-1474) -- through the
    packets that carry them, the only way in from outside the Go package"""
    for n in (1, 2, 3, 4, 5, 6, 10, 36, 37, 50):
        for length in (1, 10, 63, 100, 127):
            for zbp in (0, 1, 5, 10):
                data = bytes(range(length))
                out = impl.encode([(one(data, 0, zbp, n), 0, False, False, 2, 2)])
                hdr = out[:len(out) - length]
                assert out[len(hdr):] == data
                # the reference's decoder reads bodies from Position(), which a header does not move: decode the header alone,
                # then the body is what stands at position 0 -- the header bytes themselves (t2.go:463-503)
                res = impl.decode(out, [(one(), 0, False, False)])
                pos, cbs = res[0]
                assert cbs[0][0][1:] == (0, zbp, n) and len(cbs[0][0][0]) == length and pos == length
                assert cbs[0][0][0] == out[:length]
    # a length of 0 is "not included" for the encoder (t2.go:326: len(cb.Data) > 0), so length 0 only exists on the decode side
    assert impl.decode(bytes([0b11100000]), [(one(), 0, False, False)])[0][1][0][0] == (b"", 0, 0, 1)      # 1 | 1 | 1 | 0 | 000


@pytest.mark.parametrize("impl", IMPLS, ids=ids)
def test_decode_packet_markers(impl):
    data = bytes([0xFF, 0x91, 0x00, 0x04, 0x00, 0x05, 0x00])                                # :669-690
    res = impl.decode(data, [([[]], 5, True, False)])
    assert res[0][0] >= 6
    assert impl.decode(bytes([0x00, 0xFF, 0x92]), [([[]], 0, False, True)])[0][0] == 0       # :692-708: EPH is looked for at Position() = 0
    assert impl.decode(b"", [(one(), 0, False, False)]) == ["err"]                          # no presence bit: the reader's EOF


@pytest.mark.parametrize("impl", IMPLS, ids=ids)
def test_later_layers_use_single_inclusion_bits(impl):
    # layer 1, block first included in layer 1 (:1130-1178, 1354-1419, 1616-1669): 1 presence | 1 included | 1 zbp = 0 | 0 one pass | 001 1
    out = impl.encode([(one(b"\x7e", 1, 0, 1), 1, False, False, 1, 1)])
    assert out == bytes([0b11100011]) + b"\x7e"
    # not yet included at layer 0 (:1259-1290): IncludedInLayers = 2 -> empty packet
    assert impl.encode([(one(b"\x01\x02", 2, 0, 1), 0, False, False, 1, 1)]) == b"\x00"
    res = impl.decode(bytes([0b11100011, 0x7e]), [(one(None, 5, 9, 0), 1, False, False)])
    assert res[0][1][0][0][1:] == (1, 0, 1)


@pytest.mark.parametrize("impl", IMPLS, ids=ids)
def test_tag_tree_shape_and_divisor(impl):
    for w, h, lv in [(1, 1, 1), (2, 2, 2), (4, 4, 3), (8, 8, 4), (3, 3, 3), (5, 7, 4), (16, 16, 5)]:    # tcd_test.go:81-115
        assert impl.tagtree(w, h)[0] == lv
    assert impl.tagtree(2, 4) == (3, [8, 2, 1])                                                         # :720-745 (node (0,3) = index 6 of 8)
    with pytest.raises((J2KError, t2ref.GoPanic)):                                                       # cbIdx % InclusionTree.width with width 0
        impl.encode([(one(b"\x01", 0, 0, 1), 0, False, False, 0, 1)])


# ---- InitTile (tcd_test.go:171-360, 584-678) -----------------------------------------------------------------------------
def test_header(**kw):
    h = dict(ImageWidth=64, ImageHeight=64, ImageXOffset=0, ImageYOffset=0, TileWidth=64, TileHeight=64, TileXOffset=0, TileYOffset=0,
             NumTilesX=1, NumDecompositions=2, CodeBlockWidthExp=2, CodeBlockHeightExp=2, Subsampling=[(1, 1)])
    h.update(kw)
    return h


test_header.__test__ = False


@pytest.mark.parametrize("impl", IMPLS, ids=ids)
def test_init_tile_reference_expectations(impl):
    tile, comps = impl.init_tile(test_header(), 0)                                           # tcd_test.go:217-285
    assert tile == (0, 0, 64, 64) and len(comps) == 1 and comps[0][0] == (0, 0, 64, 64)
    ress = comps[0][1]
    assert [r[0] for r in ress] == [0, 1, 2] and [len(r[2]) for r in ress] == [1, 3, 3]
    assert [b[0] for b in ress[0][2]] == [0] and [b[0] for b in ress[1][2]] == [1, 2, 3]      # LL; HL, LH, HH (:584-616)
    for r in ress:                                                                           # :618-657
        for (_, b, nx, ny, cbs) in r[2]:
            assert len(cbs) == nx * ny
            if cbs and b[2] > b[0] and b[3] > b[1]:
                assert cbs[0][:2] == b[:2]
    _, comps = impl.init_tile(test_header(Subsampling=[(1, 1), (2, 2), (2, 2)]), 0)            # :287-318
    assert [(c[0][2] - c[0][0], c[0][3] - c[0][1]) for c in comps] == [(64, 64), (32, 32), (32, 32)]
    h = test_header(ImageWidth=128, ImageHeight=128, NumTilesX=2)                             # :320-359
    assert [impl.init_tile(h, i)[0] for i in range(4)] == [(0, 0, 64, 64), (64, 0, 128, 64), (0, 64, 64, 128), (64, 64, 128, 128)]
    assert impl.init_tile(test_header(ImageXOffset=10, ImageYOffset=20), 0)[0][:2] == (10, 20)   # :659-678
    # the band rectangles as initBand writes them (tcd.go:343-361): halves and a quadrant of the RESOLUTION's rectangle
    r1 = impl.init_tile(test_header(), 0)[1][0][1][1]
    assert r1[1] == (0, 0, 32, 32) and [b[1] for b in r1[2]] == [(0, 0, 32, 16), (0, 0, 16, 32), (16, 16, 32, 32)]


# ---- product against oracle on randomised inputs --------------------------------------------------------------------------
def test_packet_sequences_differential():
    rng = np.random.default_rng(7)
    for _ in range(400):
        nc, nr, nl = (int(rng.integers(0, 4)) for _ in range(3))
        prec = [[[int(rng.integers(0, 4))] for _ in range(int(rng.integers(0, nr + 2)))] for _ in range(int(rng.integers(0, nc + 2)))]
        order = int(rng.integers(0, 6))
        assert Prod.seq(nc, nr, nl, prec, order) == Orc.seq(nc, nr, nl, prec, order), (nc, nr, nl, prec, order)


def _rand_bands(rng, ff_heavy):
    bands = []
    for _ in range(int(rng.integers(0, 4))):
        b = []
        for _ in range(int(rng.integers(0, 5))):
            n = int(rng.choice([0, 0, 1, 3, 7, 100, 127, 128, 255, 300]))
            data = None if n == 0 and rng.random() < 0.5 else bytes(rng.integers(0, 256, n).astype(np.uint8))
            passes = int(rng.choice([-1, 0, 1, 2, 3, 5, 6, 36, 37, 164, 165, 200]))
            zbp = int(rng.choice([0, 1, 2, 6, 7, 8, 13, 40] if ff_heavy else [-2, 0, 1, 3, 9]))
            b.append((data, int(rng.integers(-1, 4)), zbp, passes))
        bands.append(b)
    return bands


def test_encode_packets_differential():
    """byte-identical streams from multi-packet encoders: the writer's 'last byte was 0xFF' flag crosses packets, unary runs make
    0xFF header bytes (7-bit bytes behind them), lengths of 128 and more wrap the 3-bit length-of-length"""
    rng = np.random.default_rng(11)
    for it in range(600):
        ff_heavy = it % 2 == 0
        packets = []
        for _ in range(int(rng.integers(1, 5))):
            bands = _rand_bands(rng, ff_heavy)
            if ff_heavy and bands and bands[0]:
                # leading zero-valued inclusion + long unary zero bit-plane runs are rare; all-ones bytes come from pass codes 0x1FF
                bands[0][0] = (b"\x01", 0, 0, 200)
            packets.append((bands, int(rng.integers(0, 3)), bool(rng.integers(0, 2)), bool(rng.integers(0, 2)), 1, 1))
        assert Prod.encode(packets) == Orc.encode(packets), packets


def test_decode_packets_differential():
    rng = np.random.default_rng(13)
    for it in range(600):
        if it % 3 == 0:      # streams the encoder made
            packets = [(_rand_bands(rng, False), int(rng.integers(0, 3)), bool(rng.integers(0, 2)), bool(rng.integers(0, 2)), 1, 1)
                       for _ in range(int(rng.integers(1, 4)))]
            data = Orc.encode(packets)
            shapes = [([[(None, 0, 0, 0) for _ in b] for b in bands], layer, sop, eph) for bands, layer, sop, eph, _, _ in packets]
        else:                # arbitrary bytes, many 0xFF
            data = bytes(rng.choice([0xFF, 0x00, 0x80, 0x7F, int(rng.integers(0, 256))], size=int(rng.integers(0, 40))).astype(np.uint8))
            shapes = [([[(bytes(rng.integers(0, 256, int(rng.integers(0, 4))).astype(np.uint8)) or None, int(rng.integers(0, 3)), 0, 0)
                         for _ in range(int(rng.integers(0, 4)))] for _ in range(int(rng.integers(0, 3)))],
                       int(rng.integers(0, 3)), bool(rng.integers(0, 2)), bool(rng.integers(0, 2))) for _ in range(int(rng.integers(1, 4)))]
        assert Prod.decode(data, shapes) == Orc.decode(data, shapes), (data, shapes)


def test_init_tile_differential():
    rng = np.random.default_rng(17)
    for _ in range(300):
        tw, th = int(rng.integers(1, 300)), int(rng.integers(1, 300))
        ntx = int(rng.integers(1, 5))
        h = dict(ImageWidth=int(rng.integers(1, tw * ntx + 1)), ImageHeight=int(rng.integers(1, th * 3 + 1)), ImageXOffset=int(rng.integers(0, 20)),
                 ImageYOffset=int(rng.integers(0, 20)), TileWidth=tw, TileHeight=th, TileXOffset=int(rng.integers(0, 10)),
                 TileYOffset=int(rng.integers(0, 10)), NumTilesX=ntx, NumDecompositions=int(rng.integers(0, 7)),
                 CodeBlockWidthExp=int(rng.integers(0, 5)), CodeBlockHeightExp=int(rng.integers(0, 5)),
                 Subsampling=[(int(rng.integers(1, 4)), int(rng.integers(1, 4))) for _ in range(int(rng.integers(1, 4)))])
        idx = int(rng.integers(0, ntx * 3))
        try:
            want = t2ref.init_tile(h, idx)
        except t2ref.GoPanic:
            with pytest.raises(J2KError):
                t2.init_tile(h, idx)
            continue
        assert t2.init_tile(h, idx) == want, (h, idx)


def test_c_abi_argument_errors():
    import ctypes as C
    L = _lib.lib()
    n = C.c_size_t(0)
    assert L.j2k_t2_packet_sequence(1, 1, 1, None, None, 0, 0, None, C.c_size_t(0), C.byref(n)) == _lib.ERR_CAPACITY and n.value == 1
    assert L.j2k_t2_packet_sequence(1, 1, 1, None, None, 0, 0, None, C.c_size_t(0), None) == _lib.ERR_INVALID_ARG
    with pytest.raises(J2KError) as e:
        t2.init_tile(test_header(NumTilesX=0), 0)
    assert e.value.status == _lib.ERR_GO_PANIC
    with pytest.raises(J2KError) as e:
        t2.init_tile(test_header(Subsampling=[(0, 1)]), 0)
    assert e.value.status == _lib.ERR_GO_PANIC
