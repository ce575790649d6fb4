"""GPU parity: 9-7 DWT / ICT / quantisation kernels vs the C oracle.

Tolerance (north_star: "within a stated PSNR tolerance for 9-7 lossy"): the kernels are built
with -ffp-contract=off and follow the reference's operation order, so the bar here is the
strongest one available: float64 results BIT-IDENTICAL to the oracle (np.array_equal), i.e.
PSNR = inf between GPU and CPU outputs.  The reference's own tests use 1e-10/1e-9 round-trip
tolerances (internal/dwt/dwt_test.go:48-79,118-150,275-309) -- also asserted."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SIZES_1D = [1, 2, 3, 4, 5, 7, 8, 16, 33, 64, 127, 128, 255, 256, 300, 512, 1000, 4096]
SIZES_2D = [(1, 1), (2, 2), (1, 5), (5, 1), (2, 7), (3, 3), (4, 4), (8, 8), (16, 16), (13, 7), (33, 20), (64, 64),
            (100, 37), (256, 112), (512, 64), (130, 258), (520, 36), (300, 9), (9, 300)]


@pytest.fixture(scope="module")
def dwt():
    from j2kgfx import dwt
    return dwt


@pytest.mark.parametrize("n", SIZES_1D)
def test_forward97_inverse97_1d(dwt, oracle, n):
    rng = np.random.default_rng(n)
    x = rng.uniform(-500, 500, n)
    y = x.copy()
    dwt.Forward97(y, n)
    assert np.array_equal(y, oracle.fwd97_1d(x))
    z = y.copy()
    dwt.Inverse97(z, n)
    assert np.array_equal(z, oracle.inv97_1d(y))
    assert np.max(np.abs(z - x)) < 1e-10 if n else True


@pytest.mark.parametrize("w,h", SIZES_2D)
def test_forward2d97(dwt, oracle, w, h):
    rng = np.random.default_rng(w * 1009 + h)
    x = rng.uniform(-500, 500, w * h)
    y = x.copy()
    dwt.Forward2D97(y, w, h)
    assert np.array_equal(y.reshape(h, w), oracle.fwd97_2d(x, w, h))
    z = y.copy()
    dwt.Inverse2D97(z, w, h)
    assert np.array_equal(z.reshape(h, w), oracle.inv97_2d(y, w, h))
    assert np.max(np.abs(z - x)) < 1e-9


@pytest.mark.parametrize("w,h", SIZES_2D)
@pytest.mark.parametrize("levels", [1, 3, 5])
def test_multilevel97(dwt, oracle, w, h, levels):
    rng = np.random.default_rng(w + 3 * h + levels)
    x = rng.uniform(-2048, 2048, w * h)
    y = x.copy()
    dwt.DecomposeMultiLevel97(y, w, h, levels)
    assert np.array_equal(y.reshape(h, w), oracle.decompose97(x, w, h, levels))
    z = y.copy()
    dwt.ReconstructMultiLevel97(z, w, h, levels)
    assert np.array_equal(z.reshape(h, w), oracle.reconstruct97(y, w, h, levels))
    assert np.max(np.abs(z - x)) < 1e-8


@pytest.mark.parametrize("w,h,levels", [(64, 64, 5), (100, 37, 3), (256, 112, 5), (33, 33, 2), (512, 40, 4)])
def test_tcd_apply_dwt_irreversible(oracle, w, h, levels):
    """tcd.TileEncoder.ApplyForwardDWT / TileDecoder.ApplyInverseDWT, 9-7 branch (tcd.go:520-532, 428-435)"""
    import ctypes as C
    from j2kgfx import default_context
    ctx = default_context()
    rng = np.random.default_rng(w * h)
    x = rng.integers(-2048, 2048, w * h).astype(np.int32)
    y = x.copy()
    ctx.check(ctx.L.j2k_tcd_apply_forward_dwt(ctx.h, y.ctypes.data_as(C.c_void_p), w, h, levels, 0))
    assert np.array_equal(y.reshape(h, w), oracle.tcd_forward_dwt(x, w, h, levels, 0))
    z = y.copy()
    ctx.check(ctx.L.j2k_tcd_apply_inverse_dwt(ctx.h, z.ctypes.data_as(C.c_void_p), w, h, levels, 0))
    assert np.array_equal(z.reshape(h, w), oracle.tcd_inverse_dwt(y, w, h, levels, 0))
    # reversible branch through the same entry points
    y = x.copy()
    ctx.check(ctx.L.j2k_tcd_apply_forward_dwt(ctx.h, y.ctypes.data_as(C.c_void_p), w, h, levels, 1))
    assert np.array_equal(y.reshape(h, w), oracle.tcd_forward_dwt(x, w, h, levels, 1))


FRAMES = [(64, 64, 3, (0, 0), 3, 8, 75), (100, 75, 3, (0, 0), 6, 12, 75), (96, 80, 1, (32, 32), 4, 8, 0),
          (640, 368, 3, (512, 512), 6, 12, 75), (333, 217, 4, (128, 64), 5, 10, 30), (512, 512, 3, (0, 0), 6, 12, 100),
          # single components whose level 0 takes the workgroup kernels (16 <= w <= 512, w % 8 == 0; dwt97_l0wg.inc SRC = 2): a gray
          # frame in one tile, gray 512-tiles with a ragged last column of 256, four components (no colour transform), odd heights
          (512, 256, 1, (0, 0), 4, 12, 50), (1280, 131, 1, (512, 512), 6, 8, 75), (256, 67, 4, (128, 128), 3, 16, 8191), (64, 2, 1, (0, 0), 2, 8, 1)]


@pytest.mark.parametrize("W,H,Cn,tile,nres,prec,quality", FRAMES)
def test_plan_forward_lossy(oracle, W, H, Cn, tile, nres, prec, quality):
    """encoder.preprocess, lossy branch (DC shift, ICT + round half away, 9-7, v/(1/Quality) +- 0.5):
    quantised int32 coefficients identical to the oracle's for every tile-component; then the
    decode side (tcd ApplyInverseDWT rounding, InverseICT rounding, DC shift) on those coefficients."""
    import torch
    from j2kgfx.codec import FramePlan
    rng = np.random.default_rng(W * H + prec)
    frame = rng.integers(0, 1 << prec, size=(Cn, H, W)).astype(np.int32)
    plan = FramePlan(W, H, Cn, precision=prec, lossless=False, quality=quality, num_resolutions=nres, tile=tile)
    d_frame = torch.from_numpy(frame).to(plan.device)
    torch.cuda.synchronize()
    coeff = plan.forward(d_frame)
    back = plan.inverse(coeff)
    plan.ctx.sync()
    hc = coeff.cpu().numpy(); hb = back.cpu().numpy()
    planes = plan.planes()
    levels = nres - 1 if nres - 1 > 0 else 5
    for t in np.unique(planes[:, 0]):
        rows = planes[planes[:, 0] == t]
        x0, y0, w, h = (int(v) for v in rows[0, 2:6])
        crop = [frame[c, y0:y0 + h, x0:x0 + w].copy() for c in range(Cn)]
        want = oracle.preprocess(crop, w, h, prec, False, nres, quality)
        for row in rows:
            c, off = int(row[1]), int(row[6])
            assert np.array_equal(hc[off:off + w * h].reshape(h, w), want[c]), (t, c)
        inv = [oracle.tcd_inverse_dwt(want[c], w, h, levels, 0) for c in range(Cn)]
        inv = oracle.postprocess(inv, prec, False)
        for c in range(Cn):
            assert np.array_equal(hb[c, y0:y0 + h, x0:x0 + w], inv[c]), ("inverse", t, c)
