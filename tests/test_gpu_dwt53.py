"""GPU parity: 5-3 DWT / RCT / DC shift kernels (through the C ABI) vs the C oracle.
Mirrors the reference's own tests: internal/dwt/dwt_test.go:8-46,81-116,152-187,
internal/mct/mct_test.go:8-39,533-598,681-717 -- plus forward-coefficient equality,
which the reference never asserts."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SIZES_1D = [0, 1, 2, 3, 4, 5, 7, 8, 16, 33, 64, 127, 128, 129, 255, 256, 500, 512, 1000, 4096, 8192]
SIZES_2D = [(1, 1), (2, 2), (1, 5), (5, 1), (2, 7), (3, 3), (4, 4), (8, 8), (16, 16), (13, 7), (33, 20),
            (64, 64), (128, 128), (100, 37), (256, 256), (256, 112), (512, 112), (512, 512), (130, 258),
            (520, 36), (1000, 8), (8, 1000), (1026, 34)]


@pytest.fixture(scope="module")
def mods():
    import j2kgfx
    from j2kgfx import dwt, mct
    return j2kgfx, dwt, mct


def rnd(rng, shape, amp=1 << 12):
    return rng.integers(-amp, amp, size=shape).astype(np.int32)


@pytest.mark.parametrize("n", SIZES_1D)
def test_forward53_inverse53_1d(mods, oracle, n):
    _, dwt, _ = mods
    rng = np.random.default_rng(n)
    x = rnd(rng, n)
    y = x.copy()
    dwt.Forward53(y, n)
    assert np.array_equal(y, oracle.fwd53_1d(x)) if n else True
    dwt.Inverse53(y, n)
    assert np.array_equal(y, x)


def test_hand_derived_values(mods):
    _, dwt, mct = mods
    for src, want in (([1, 2, 3, 4], [1, 3, 0, 1]), ([10, 20], [15, 10]), ([1, 2, 3, 4, 5, 6, 7], [1, 3, 5, 7, 0, 0, 0])):
        a = np.array(src, dtype=np.int32)
        dwt.Forward53(a, a.size)
        assert a.tolist() == want
    r, g, b = (np.array([v], dtype=np.int32) for v in (100, 110, 120))
    mct.ForwardRCT(r, g, b)
    assert (r[0], g[0], b[0]) == (110, 10, -10)
    r, g, b = (np.array([v], dtype=np.int32) for v in (-100, 50, -50))
    mct.ForwardRCT(r, g, b)
    assert r[0] == -13


@pytest.mark.parametrize("w,h", SIZES_2D)
def test_forward2d53(mods, oracle, w, h):
    _, dwt, _ = mods
    rng = np.random.default_rng(w * 10007 + h)
    x = rnd(rng, w * h)
    y = x.copy()
    dwt.Forward2D53(y, w, h)
    assert np.array_equal(y.reshape(h, w), oracle.fwd53_2d(x, w, h))
    dwt.Inverse2D53(y, w, h)
    assert np.array_equal(y, x)


@pytest.mark.parametrize("w,h", SIZES_2D)
@pytest.mark.parametrize("levels", [1, 2, 5])
def test_multilevel53(mods, oracle, w, h, levels):
    _, dwt, _ = mods
    rng = np.random.default_rng(w * 31 + h * 7 + levels)
    x = rnd(rng, w * h)
    y = x.copy()
    dwt.DecomposeMultiLevel53(y, w, h, levels)
    assert np.array_equal(y.reshape(h, w), oracle.decompose53(x, w, h, levels))
    # inverse of arbitrary coefficients equals the oracle's inverse, not only the round trip
    z = rnd(rng, w * h)
    z2 = z.copy()
    dwt.ReconstructMultiLevel53(z2, w, h, levels)
    assert np.array_equal(z2.reshape(h, w), oracle.reconstruct53(z, w, h, levels))
    dwt.ReconstructMultiLevel53(y, w, h, levels)
    assert np.array_equal(y, x)


def test_wraparound_extremes(mods, oracle):
    """Go int32 arithmetic wraps; full-range inputs must still match bit for bit."""
    _, dwt, mct = mods
    rng = np.random.default_rng(5)
    w, h = 67, 41
    x = rng.integers(-2**31, 2**31, size=w * h, dtype=np.int64).astype(np.int32)
    y = x.copy()
    dwt.DecomposeMultiLevel53(y, w, h, 3)
    assert np.array_equal(y.reshape(h, w), oracle.decompose53(x, w, h, 3))
    r, g, b = (rng.integers(-2**31, 2**31, size=1000, dtype=np.int64).astype(np.int32) for _ in range(3))
    er, eg, eb = oracle.rct_fwd(r, g, b)
    mct.ForwardRCT(r, g, b)
    assert np.array_equal(r, er) and np.array_equal(g, eg) and np.array_equal(b, eb)
    ir, ig, ib = oracle.rct_inv(r, g, b)
    mct.InverseRCT(r, g, b)
    assert np.array_equal(r, ir) and np.array_equal(g, ig) and np.array_equal(b, ib)


@pytest.mark.parametrize("p", [1, 4, 8, 10, 12, 16])
def test_dc_level_shift(mods, oracle, p):
    _, _, mct = mods
    x = np.arange(-50, 1000, dtype=np.int32)
    y = x.copy()
    mct.DCLevelShiftForward(y, p)
    assert np.array_equal(y, oracle.dc_shift_fwd(x, p))
    mct.DCLevelShiftInverse(y, p)
    assert np.array_equal(y, x)


def test_ict(mods, oracle):
    _, _, mct = mods
    rng = np.random.default_rng(9)
    r, g, b = (rng.uniform(-200, 300, 5000) for _ in range(3))
    er, eg, eb = oracle.ict_fwd(r, g, b)
    mct.ForwardICT(r, g, b)
    assert np.array_equal(r, er) and np.array_equal(g, eg) and np.array_equal(b, eb)   # bit-exact f64
    ir, ig, ib = oracle.ict_inv(r, g, b)
    mct.InverseICT(r, g, b)
    assert np.array_equal(r, ir) and np.array_equal(g, ig) and np.array_equal(b, ib)


FRAMES = [  # (W, H, C, tile, num_resolutions, precision)
    (64, 64, 3, (0, 0), 3, 8), (512, 512, 3, (0, 0), 3, 8), (100, 75, 3, (0, 0), 6, 8), (96, 80, 1, (32, 32), 4, 8),
    (640, 368, 3, (512, 512), 6, 8), (1280, 624, 3, (512, 512), 6, 8), (333, 217, 4, (128, 64), 5, 10),
    (256, 256, 2, (0, 0), 0, 12), (1034, 40, 3, (0, 0), 2, 8)]


@pytest.mark.parametrize("W,H,Cn,tile,nres,prec", FRAMES)
def test_plan_forward_inverse_lossless(mods, oracle, W, H, Cn, tile, nres, prec):
    """encoder.preprocess per tile-component (tile == reference pipeline on the cropped sub-image),
    then the inverse path back to the pixels: bit-exact lossless round trip."""
    import torch
    from j2kgfx.codec import FramePlan
    rng = np.random.default_rng(W + H)
    frame = rng.integers(0, 1 << prec, size=(Cn, H, W)).astype(np.int32)
    plan = FramePlan(W, H, Cn, precision=prec, lossless=True, num_resolutions=nres, tile=tile)
    d_frame = torch.from_numpy(frame).to(plan.device)
    torch.cuda.synchronize()
    coeff = plan.forward(d_frame)
    plan.ctx.sync()
    hc = coeff.cpu().numpy()
    planes = plan.planes()
    for t in np.unique(planes[:, 0]):
        rows = planes[planes[:, 0] == t]
        x0, y0, w, h = (int(v) for v in rows[0, 2:6])
        crop = [frame[c, y0:y0 + h, x0:x0 + w].copy() for c in range(Cn)]
        want = oracle.preprocess(crop, w, h, prec, True, nres)
        for row in rows:
            c, off = int(row[1]), int(row[6])
            got = hc[off:off + w * h].reshape(h, w)
            assert np.array_equal(got, want[c]), (t, c)
    back = plan.inverse(coeff)
    plan.ctx.sync()
    assert np.array_equal(back.cpu().numpy(), frame)
