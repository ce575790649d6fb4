"""CPU: the C oracle against the committed golden vectors (tests/golden/golden_v1.npz), which were
produced by the independent Python transliteration oracle/pyref.py (tests/golden/make_golden.py)."""
import os

import numpy as np
import pytest

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "golden_v1.npz"))


def test_dwt53_and_97(oracle):
    for i, (w, h, L) in enumerate(G["dwt_cases"]):
        assert np.array_equal(oracle.decompose53(G["dwt53_in_%d" % i], w, h, L).reshape(-1), G["dwt53_out_%d" % i])
        assert np.array_equal(oracle.reconstruct53(G["dwt53_out_%d" % i], w, h, L).reshape(-1), G["dwt53_in_%d" % i])
        assert np.array_equal(oracle.decompose97(G["dwt97_in_%d" % i], w, h, L).reshape(-1), G["dwt97_out_%d" % i])
        assert np.array_equal(oracle.reconstruct97(G["dwt97_out_%d" % i], w, h, L).reshape(-1), G["dwt97_rec_%d" % i])


def test_preprocess(oracle):
    for i, (w, h, C, prec, nres, q) in enumerate(G["pre_cases"]):
        planes = G["pre_in_%d" % i]
        for name, lossless in (("ll", 1), ("ly", 0)):
            got = oracle.preprocess([planes[c] for c in range(C)], w, h, prec, lossless, nres, q)
            assert np.array_equal(np.stack(got), G["pre_%s_%d" % (name, i)])


def test_mq(oracle):
    for i in range(4):
        b = oracle.mq_encode(G["mq_ctx_%d" % i], G["mq_dec_%d" % i])
        assert np.array_equal(b, G["mq_bytes_%d" % i])
        assert np.array_equal(oracle.mq_decode(b, G["mq_ctx_%d" % i]), G["mq_dec_%d" % i])


def test_t1(oracle):
    for i, (w, h, band) in enumerate(G["t1_cases"]):
        b, nb = oracle.t1_encode(G["t1_in_%d" % i], w, h, band)
        assert np.array_equal(b, G["t1_bytes_%d" % i]) and nb == int(G["t1_nbps_%d" % i][0])
        assert np.array_equal(oracle.t1_decode(b, nb, band, w, h).reshape(-1), G["t1_in_%d" % i])
        g = G["t1_garbage_%d" % i]
        assert np.array_equal(oracle.t1_decode(g, max(nb, 1), band, w, h).reshape(-1), G["t1_garbage_dec_%d" % i])


def test_ht(oracle):
    for i, (w, h) in enumerate(G["ht_cases"]):
        b = oracle.ht_encode(G["ht_in_%d" % i], w, h)
        assert np.array_equal(b, G["ht_bytes_%d" % i])
        assert np.array_equal(oracle.ht_decode(b, w, h).reshape(-1), G["ht_dec_%d" % i])
        assert np.array_equal(oracle.ht_decode(G["ht_garbage_%d" % i], w, h).reshape(-1), G["ht_garbage_dec_%d" % i])


def test_enumerate_blocks(oracle):
    for i, a in enumerate(G["enum_cases"]):
        got = oracle.enumerate_blocks(*[int(v) for v in a])
        want = G["enum_%d" % i]
        assert len(got) == len(want)
        for j in range(len(got)):
            assert tuple(int(v) for v in got[j]) == tuple(int(v) for v in want[j])
    # block counts stated in SURVEY.md 8a (a10)
    assert len(oracle.enumerate_blocks(3, 512, 512, 3, 256, 256)) == 21
    assert len(oracle.enumerate_blocks(3, 512, 512, 6, 64, 64)) == 210


def test_raw_coders_oracle_vs_python_restatement():
    """RawEncoder / RawDecoder (mqc.go:516-600): the C oracle against a literal Python restatement, and the property the
    bit stuffing exists for (after an 0xFF byte the next byte carries 7 bits, so decode(encode(bits)) == bits)."""
    import oracle as orc

    def py_encode(bits):
        buf, c, ct = [], 0, 8
        for b in bits:
            ct -= 1
            c += (int(b) & 1) << ct
            if ct == 0:
                buf.append(c & 0xFF)
                ct = 7 if (c & 0xFF) == 0xFF else 8
                c = 0
        if ct < 8:
            buf.append(c & 0xFF)
        return bytes(buf)

    def py_decode(data, n):
        out, pos, c, ct = [], 0, 0, 0
        for _ in range(n):
            if ct == 0:
                if c == 0xFF:
                    if pos < len(data) and data[pos] > 0x8F:
                        c, ct = 0xFF, 8
                    elif pos < len(data):
                        c, ct = data[pos], 7
                        pos += 1
                    else:
                        c, ct = 0xFF, 8
                elif pos < len(data):
                    c, ct = data[pos], 8
                    pos += 1
                else:
                    c, ct = 0xFF, 8
            ct -= 1
            out.append((c >> ct) & 1)
        return out

    rng = np.random.default_rng(7)
    for n, p1 in [(0, 0.5), (1, 0.5), (7, 0.5), (8, 1.0), (64, 1.0), (1000, 0.5), (1000, 0.97), (4099, 0.9)]:
        bits = (rng.random(n) < p1).astype(np.uint8)
        enc = orc.raw_encode(bits)
        assert enc.tobytes() == py_encode(bits)
        dec = orc.raw_decode(enc, n)
        assert dec.tolist() == py_decode(enc.tobytes(), n)
        assert dec.tolist() == bits.tolist()
    assert orc.raw_encode(np.ones(8, np.uint8)).tolist() == [0xFF, 0x00]            # ct = 7 < 8 after an 0xFF: Flush appends c
    assert orc.raw_encode(np.ones(15, np.uint8)).tolist() == [0xFF, 0x7F]          # second byte holds 7 bits
    assert orc.raw_decode(np.array([], np.uint8), 9).tolist() == [1] * 9            # exhausted input feeds 0xFF
