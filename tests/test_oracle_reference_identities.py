"""CPU: pin the C oracle with everything the reference's OWN tests assert for the hot path
 -- round-trip identities, spot values, table properties -- plus the
hand-derived values of SURVEY 8c.  The reference holds no byte- or coefficient-level vectors."""
import numpy as np
import pytest


# ---- internal/dwt/dwt_test.go ---------------------------------------------------
@pytest.mark.parametrize("vec", [[1, 2, 3, 4], [1, 2, 3, 4, 5, 6, 7, 8], [10, 20], [1, 2, 3, 4, 5, 6, 7], [5],
                                 [100, -50, 25, -12, 6, -3], list(range(16)), [0, 0, 0, 0, 0]])
def test_dwt53_1d_roundtrip(oracle, vec):                       # dwt_test.go:8-46
    x = np.array(vec, dtype=np.int32)
    assert np.array_equal(oracle.inv53_1d(oracle.fwd53_1d(x)), x)


def test_hand_derived(oracle):
    assert oracle.fwd53_1d([1, 2, 3, 4]).tolist() == [1, 3, 0, 1]
    assert oracle.fwd53_1d([10, 20]).tolist() == [15, 10]
    assert oracle.fwd53_1d([1, 2, 3, 4, 5, 6, 7]).tolist() == [1, 3, 5, 7, 0, 0, 0]
    y, u, v = oracle.rct_fwd([100], [110], [120])
    assert (y[0], u[0], v[0]) == (110, 10, -10)
    assert oracle.rct_fwd([-100], [50], [-50])[0][0] == -13       # (-50) >> 2 floors


@pytest.mark.parametrize("w,h", [(4, 4), (8, 8), (16, 16), (8, 4), (4, 8), (7, 5)])
def test_dwt53_2d_roundtrip(oracle, w, h):                       # dwt_test.go:81-116
    x = ((np.arange(w * h) * 7) % 256).astype(np.int32)
    assert np.array_equal(oracle.inv53_2d(oracle.fwd53_2d(x, w, h), w, h).reshape(-1), x)


@pytest.mark.parametrize("size,levels", [(8, 1), (16, 2), (32, 3), (64, 4), (64, 5)])
def test_dwt53_multilevel_roundtrip(oracle, size, levels):       # dwt_test.go:152-187
    x = (np.arange(size * size) % 256).astype(np.int32)
    y = oracle.decompose53(x, size, size, levels)
    assert np.array_equal(oracle.reconstruct53(y, size, size, levels).reshape(-1), x)


def test_dwt97_roundtrips(oracle):                               # dwt_test.go:48-79, 118-150, 275-309
    x = np.array([1, 2, 3, 4, 5, 6, 7, 8], dtype=np.float64)
    assert np.max(np.abs(oracle.inv97_1d(oracle.fwd97_1d(x)) - x)) < 1e-10
    x = (np.arange(64, dtype=np.float64) * 3.5) % 100
    assert np.max(np.abs(oracle.inv97_2d(oracle.fwd97_2d(x, 8, 8), 8, 8).reshape(-1) - x)) < 1e-10
    x = (np.arange(32 * 32) % 256).astype(np.float64)
    assert np.max(np.abs(oracle.reconstruct97(oracle.decompose97(x, 32, 32, 3), 32, 32, 3).reshape(-1) - x)) < 1e-9


def test_pooled_buffer_growth_size(oracle):                      # dwt_test.go:455-495 (n = 8192 > pool default 4096)
    x = (np.arange(8192) % 1000).astype(np.int32)
    assert np.array_equal(oracle.inv53_1d(oracle.fwd53_1d(x)), x)


# ---- internal/mct/mct_test.go ----------------------------------------------------
def test_rct_roundtrip_incl_negatives(oracle):                   # mct_test.go:8-39, 533-598
    rng = np.random.default_rng(0)
    r, g, b = (rng.integers(-255, 256, 1000).astype(np.int32) for _ in range(3))
    y, u, v = oracle.rct_fwd(r, g, b)
    r2, g2, b2 = oracle.rct_inv(y, u, v)
    assert np.array_equal(r2, r) and np.array_equal(g2, g) and np.array_equal(b2, b)


def test_ict_roundtrip_tolerance(oracle):                        # mct_test.go:41-69, 600-679 (1e-2)
    rng = np.random.default_rng(1)
    r, g, b = (rng.uniform(0, 255, 1000) for _ in range(3))
    r2, g2, b2 = oracle.ict_inv(*oracle.ict_fwd(r, g, b))
    assert max(np.max(np.abs(r2 - r)), np.max(np.abs(g2 - g)), np.max(np.abs(b2 - b))) < 1e-2


@pytest.mark.parametrize("p", [1, 4, 8, 10, 12, 16])
def test_dc_shift(oracle, p):                                    # mct_test.go:681-717
    x = np.array([0, 1, (1 << p) - 1], dtype=np.int32)
    y = oracle.dc_shift_fwd(x, p)
    assert y.tolist() == [v - (1 << (p - 1)) for v in x.tolist()]
    assert np.array_equal(oracle.dc_shift_inv(y, p), x)


# ---- internal/entropy ---------------------------------------------------------------
def test_mq_table_spot_values(oracle):                           # coverage_test.go:1417-1446
    qe, nm, nl = oracle.mq_table()
    assert qe[0] == 0x5601 and qe[92] == 0x5601
    assert nm.max() < 94 and nl.max() < 94
    assert (nm[92], nl[92], nm[93], nl[93]) == (92, 92, 93, 93)   # uniform context never adapts
    assert (nm[0], nl[0], nm[1], nl[1]) == (2, 3, 3, 2)           # mqc.go:22-23
    assert (nm[10], nl[10]) == (76, 66) and (nm[26], nl[26]) == (58, 42)   # mqc.go:32, 48


def test_mq_symbol_identity(oracle):                             # mqc_test.go:7-66
    seqs = [[0], [1], [0, 0, 0, 0], [1, 1, 1, 1], [0, 1, 0, 1, 0, 1], [1, 0, 0, 1, 1, 0, 1, 0], [0] * 20 + [1] * 20]
    for s in seqs:
        ctx = np.zeros(len(s), dtype=np.uint8)
        assert oracle.mq_decode(oracle.mq_encode(ctx, s), ctx).tolist() == s
    i = np.arange(1000)
    s = ((i * 7 + 3) % 11 < 4).astype(np.uint8)
    ctx = (i % 19).astype(np.uint8)
    assert np.array_equal(oracle.mq_decode(oracle.mq_encode(ctx, s), ctx), s)


def test_lut_spot_values(oracle):                                # coverage_test.go:764-812
    zc, sc, sp = oracle.t1_luts()
    assert zc[0 * 256 + 0] == 0                                   # lutZCCtx[LL, none] == 0
    assert zc[0 * 256 + 0x03] == 8                                # lutZCCtx[LL, W|E] == 8
    assert zc[1 * 256 + 0x0C] == 8                                # HL swaps h and v: N|S -> 8
    assert sc[0] == 0 and sp[0] == 0


def ref_block(w, h):
    i = np.arange(w * h, dtype=np.int64)
    v = (i * 17) % 512
    v[i % 7 == 0] *= -1
    return v.astype(np.int32)


@pytest.mark.parametrize("w,h,band", [(4, 4, 0), (8, 8, 0), (16, 16, 3), (32, 32, 0), (32, 32, 1), (32, 32, 2), (32, 32, 3),
                                      (1, 1, 0), (8, 1, 0), (1, 8, 0), (8, 5, 0), (64, 64, 3)])
def test_t1_decode_of_encode(oracle, w, h, band):                # t1_test.go:7-96, coverage_test.go:72-98,414-446,464-535,814-842
    x = ref_block(w, h)
    b, nb = oracle.t1_encode(x, w, h, band)
    assert nb == int(np.max(np.abs(x))).bit_length()              # t1_test.go:66-80 recomputes numBPS this way
    assert np.array_equal(oracle.t1_decode(b, nb, band, w, h).reshape(-1), x)


def test_t1_all_negative_and_all_zero(oracle):                   # coverage_test.go:883-902
    x = -np.arange(1, 257, dtype=np.int32)
    b, nb = oracle.t1_encode(x, 16, 16, 0)
    assert np.array_equal(oracle.t1_decode(b, nb, 0, 16, 16).reshape(-1), x)
    b, nb = oracle.t1_encode(np.zeros(64, np.int32), 8, 8, 0)
    assert b.size == 0 and nb == 0                                # Encode returns nil


def test_ht_basic_properties(oracle):                            # ht_test.go:7-77, 126-147 (no numeric pin exists)
    x = ((np.arange(64 * 64) % 64) - 32).astype(np.int32)
    b = oracle.ht_encode(x, 64, 64)
    assert b.size > 0
    assert oracle.ht_decode(b, 64, 64).shape == (64, 64)          # "length equality"
    scup = int(b[-1]) + ((int(b[-2]) & 0x0F) << 8)
    assert 2048 + 2 <= scup < 4096                                # 64x64 keeps SCUP parseable
    for data in (np.zeros(0, np.uint8), np.zeros(1, np.uint8), np.zeros(2, np.uint8)):
        assert not oracle.ht_decode(data, 8, 8).any()             # nil / 1-byte / 2-byte inputs -> zeros
    assert oracle.ht_encode(np.zeros(16, np.int32), 4, 4).size == 0


def test_colorspace_reference_test_expectations():
    """The assertions of the reference's own colorspace_test.go (:48-300, 391-500) hold for the oracle's restatement of
    colorspace.go: neutral inputs stay neutral (+-2), CMY / CMYK white and black are exact, Lab mid-grey is grey within
    20, results stay in range, too few components leave the data untouched."""
    import oracle as orc

    def one(cs, vals, prec=8):
        return [int(p[0]) for p in orc.convert_colorspace([np.array([v], np.int32) for v in vals], cs, prec)]

    for cs in (3, 7, 8, 16, 17, 4):                                  # TestConvertSYCC / YCbCr601 / YPbPr709 / EYCC
        r, g, b = one(cs, [128, 128, 128])
        assert abs(r - 128) <= 2 and abs(g - 128) <= 2 and abs(b - 128) <= 2
    assert one(10, [0, 0, 0]) == [255, 255, 255] and one(10, [255, 255, 255]) == [0, 0, 0]           # TestConvertCMYToRGB
    assert one(5, [0, 0, 0, 0])[:3] == [255, 255, 255] and one(5, [0, 0, 0, 255])[0] == 0              # TestConvertCMYKToRGB
    r, g, b = one(12, [128, 128, 128])                                                                 # TestConvertCIELabToRGB
    assert abs(r - g) <= 20 and abs(g - b) <= 20
    for cs in (14, 15, 13, 9):                                                                         # e-sRGB, ROMM, Jab, PhotoYCC
        assert all(0 <= v <= 255 for v in one(cs, [128, 128, 128]))
    assert all(0 <= v <= 65535 for v in one(3, [32768, 32768, 32768], 16))                             # 16bit_precision
    two = [np.array([128], np.int32), np.array([128], np.int32)]
    assert [int(p[0]) for p in orc.convert_colorspace(two, 3, 8)] == [128, 128]                        # insufficient_components
    assert all(0 <= v <= 255 for v in one(11, [128, 156, 156, 0])[:3])                                 # TestConvertYCCKToRGB range
