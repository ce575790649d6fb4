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
