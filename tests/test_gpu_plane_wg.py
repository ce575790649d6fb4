"""GPU: the single-component workgroup kernels (dwt53_plane_wg.inc, J2K_PLANE_WG) against the general kernels and the oracle.

The general (marching) kernels are checked against the oracle in test_gpu_dwt53.py; here every shape that exercises a
boundary of the workgroup form -- strips (widths above 512, a last strip narrower than 512, exactly 512), odd and tiny heights,
bands that end at / one row before the plane's end, every level count, the prefix / final split of every level, full-range
int32 input (wraparound), packed Gray16 in and out -- is run with the knob at 0, 4 and 8 and must give identical
coefficients and identical reconstructions; the 0 setting is also compared with the oracle on the smaller shapes."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _ctx(**env):
    from j2kgfx import Context
    old = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        return Context(0)                       # knobs are read when a context is created
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


SHAPES = [(2048, 64, 0), (1024, 130, 0), (776, 37, 0), (520, 18, 0), (512, 512, 0), (512, 7, 0), (256, 256, 0), (64, 10, 0), (16, 2, 0),
          (24, 5, 0), (1536, 48, 512), (1288, 100, 0), (4096, 16, 0), (2048, 2048, 0)]


@pytest.mark.parametrize("nres", [2, 4, 6])
@pytest.mark.parametrize("W,H,tile", SHAPES)
def test_single_component_planes(oracle, W, H, tile, nres):
    import torch
    from j2kgfx.codec import FramePlan
    if (W, H) == (2048, 2048) and nres != 6:
        pytest.skip("the full C5 frame once")
    rng = np.random.default_rng(W * 3 + H + nres)
    full = rng.integers(-2 ** 31, 2 ** 31, (1, H, W), dtype=np.int64).astype(np.int32)       # wraparound everywhere
    small = rng.integers(0, 65536, (1, H, W)).astype(np.int32)
    outs = {}
    for wg in (0, 4, 8):
        plan = FramePlan(W, H, 1, precision=16, lossless=True, num_resolutions=nres, cb=(64, 64), tile=(tile, tile), coder=1,
                         ctx=_ctx(J2K_PLANE_WG=wg))
        res = []
        for frame_h in (full, small):
            frame = torch.from_numpy(frame_h).to(plan.device)
            coeff = plan.forward(frame)
            back = plan.inverse(coeff)
            plan.ctx.sync()
            res += [coeff.cpu().numpy(), back.cpu().numpy()]
            assert np.array_equal(res[-1].reshape(1, H, W), frame_h)                          # lossless, also under wraparound
        # packed Gray16 in and out (C5's boundary format)
        pix = rng.integers(0, 256, (H, W * 2)).astype(np.uint8) if wg == 0 else outs["pix"]
        outs.setdefault("pix", pix)
        dpix = torch.from_numpy(pix).to(plan.device)
        c16 = plan.forward_pixels(1, dpix)
        out = torch.zeros((H, W * 2), dtype=torch.uint8, device=plan.device)
        plan.inverse_pixels(c16, out)
        plan.ctx.sync()
        res += [c16.cpu().numpy(), out.cpu().numpy()]
        outs[wg] = res
    for wg in (4, 8):
        for a, b in zip(outs[0], outs[wg]):
            assert np.array_equal(a, b), wg
    if W * H <= 1 << 18 and tile == 0:                                                         # the oracle on the smaller ones
        want = oracle.preprocess([small[0]], W, H, 16, True, nres)
        assert np.array_equal(outs[4][2].reshape(H, W), want[0])


@pytest.mark.parametrize("W,H,tile", [(1280, 624, 512), (768, 300, 256), (1040, 64, 0)])
def test_levels_above_zero_of_rgb_frames(W, H, tile):
    """three-component frames: level 0 is the RGB-triple kernel, every deeper level runs per component on the new kernels"""
    import torch
    from j2kgfx.codec import FramePlan
    rng = np.random.default_rng(W + H)
    frame_h = rng.integers(0, 256, (3, H, W)).astype(np.int32)
    got = []
    for wg in (0, 4, 8, 12):                   # 12: 4 waves + the triple variant of level 0 (J2K_PLANE_WG3)
        plan = FramePlan(W, H, 3, precision=8, lossless=True, num_resolutions=6, cb=(64, 64), tile=(tile, tile), coder=1,
                         ctx=_ctx(J2K_PLANE_WG=wg & 7 if wg == 12 else wg, J2K_PLANE_WG3=int(wg == 12)))
        frame = torch.from_numpy(frame_h).to(plan.device)
        coeff = plan.forward(frame)
        back = plan.inverse(coeff)
        plan.ctx.sync()
        got.append(coeff.cpu().numpy())
        assert np.array_equal(back.cpu().numpy().reshape(3, H, W), frame_h)
    assert all(np.array_equal(got[0], g) for g in got[1:])


@pytest.mark.parametrize("W,H,tile,prec", [(1280, 624, 512, 8), (3840, 64, 0, 12), (520, 33, 0, 31), (16, 2, 0, 16)])
def test_rgb_triple_level0_variant(oracle, W, H, tile, prec):
    """J2K_PLANE_WG3: level 0 of an RGB triple of int32 planes (DC shift + RCT + lifting, any precision: wraparound) in
    workgroup form, strips included -- against the general kernel and the oracle"""
    import torch
    from j2kgfx.codec import FramePlan
    rng = np.random.default_rng(W + H + prec)
    frame_h = rng.integers(0, 2 ** prec, (3, H, W), dtype=np.int64).astype(np.int32)
    got = []
    for wg3 in (0, 1):
        plan = FramePlan(W, H, 3, precision=prec, lossless=True, num_resolutions=4, cb=(64, 64), tile=(tile, tile), coder=1,
                         ctx=_ctx(J2K_PLANE_WG=4, J2K_PLANE_WG3=wg3))
        frame = torch.from_numpy(frame_h).to(plan.device)
        coeff = plan.forward(frame)
        back = plan.inverse(coeff)
        plan.ctx.sync()
        got.append((coeff.cpu().numpy(), back.cpu().numpy()))
        if prec <= 16:             # (at 31 bits the RCT's (R + 2G + B) >> 2 wraps: the reference does not round-trip either)
            assert np.array_equal(got[-1][1].reshape(3, H, W), frame_h)
    assert np.array_equal(got[0][0], got[1][0]) and np.array_equal(got[0][1], got[1][1])
    got = [g[0] for g in got]
    if tile == 0 and W * H <= 1 << 18:
        want = oracle.preprocess([frame_h[c] for c in range(3)], W, H, prec, True, 4)
        assert np.array_equal(got[1].reshape(3, H, W), np.stack(want))
