"""GPU: the HIP kernels (through the C ABI) directly against the committed golden vectors
(tests/golden/golden_v1.npz, produced by the independent Python transliteration)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "golden_v1.npz"))


def test_dwt_golden():
    from j2kgfx import dwt
    for i, (w, h, L) in enumerate(G["dwt_cases"]):
        w, h, L = int(w), int(h), int(L)
        y = G["dwt53_in_%d" % i].copy()
        dwt.DecomposeMultiLevel53(y, w, h, L)
        assert np.array_equal(y, G["dwt53_out_%d" % i])
        dwt.ReconstructMultiLevel53(y, w, h, L)
        assert np.array_equal(y, G["dwt53_in_%d" % i])
        f = G["dwt97_in_%d" % i].copy()
        dwt.DecomposeMultiLevel97(f, w, h, L)
        assert np.array_equal(f, G["dwt97_out_%d" % i])          # bit-exact f64
        dwt.ReconstructMultiLevel97(f, w, h, L)
        assert np.array_equal(f, G["dwt97_rec_%d" % i])


def test_preprocess_golden():
    import torch
    from j2kgfx.codec import FramePlan
    for i, (w, h, C, prec, nres, q) in enumerate(G["pre_cases"]):
        w, h, C, prec, nres, q = (int(v) for v in (w, h, C, prec, nres, q))
        planes = G["pre_in_%d" % i]
        for name, lossless in (("ll", True), ("ly", False)):
            plan = FramePlan(w, h, C, precision=prec, lossless=lossless, quality=q, num_resolutions=nres)
            d = torch.from_numpy(planes.copy()).to(plan.device)
            torch.cuda.synchronize()
            coeff = plan.forward(d)
            plan.ctx.sync()
            hc = coeff.cpu().numpy()
            want = G["pre_%s_%d" % (name, i)]
            for row in plan.planes():
                c, off = int(row[1]), int(row[6])
                assert np.array_equal(hc[off:off + w * h].reshape(h, w), want[c]), (i, name, c)


def test_t1_golden():
    from j2kgfx import entropy
    for i, (w, h, band) in enumerate(G["t1_cases"]):
        w, h, band = int(w), int(h), int(band)
        t1 = entropy.NewT1(w, h)
        t1.SetData(G["t1_in_%d" % i])
        got = t1.Encode(band)
        want = bytes(G["t1_bytes_%d" % i])
        assert (got or b"") == want and t1.numBPS == int(G["t1_nbps_%d" % i][0])
        nb = max(t1.numBPS, 1)
        dec = entropy.NewT1(w, h).Decode(bytes(G["t1_garbage_%d" % i]), nb, band)
        assert np.array_equal(dec, G["t1_garbage_dec_%d" % i])


def test_ht_golden():
    from j2kgfx import entropy
    for i, (w, h) in enumerate(G["ht_cases"]):
        w, h = int(w), int(h)
        enc = entropy.NewHTEncoder(w, h)
        enc.SetData(G["ht_in_%d" % i])
        got = enc.Encode(0) or b""
        assert got == bytes(G["ht_bytes_%d" % i])
        assert np.array_equal(entropy.NewHTDecoder(w, h).Decode(got, 0, 0), G["ht_dec_%d" % i])
        assert np.array_equal(entropy.NewHTDecoder(w, h).Decode(bytes(G["ht_garbage_%d" % i]), 0, 0), G["ht_garbage_dec_%d" % i])


def test_job_enumeration_golden():
    from j2kgfx.codec import FramePlan
    for i, a in enumerate(G["enum_cases"]):
        C, w, h, nres, cbw, cbh = (int(v) for v in a)
        plan = FramePlan(w, h, C, precision=8, lossless=True, num_resolutions=nres, cb=(cbw, cbh))
        got = plan.blocks()
        want = G["enum_%d" % i]      # comp, res, band, x0, y0, w, h
        assert len(got) == len(want)
        for j in range(len(got)):
            assert (int(got[j]["plane"]), int(got[j]["band"]), int(got[j]["x0"]), int(got[j]["y0"]), int(got[j]["w"]), int(got[j]["h"])) == \
                   (int(want[j][0]), int(want[j][2]), int(want[j][3]), int(want[j][4]), int(want[j][5]), int(want[j][6]))
