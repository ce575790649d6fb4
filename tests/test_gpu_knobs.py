"""GPU: the non-default settings of the round-2 A/B knobs produce the same bits as the default ones ("none of them
changes results").  The knobs are read when a context is created, so each case builds its own Context under a patched
environment and compares with the default context on the same frame."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _ctx(env):
    from j2kgfx import Context
    old = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        return Context(0)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("env", [{"J2K_L0_FUSE": 8}, {"J2K_L0_FUSE": 16}, {"J2K_L0_WG": 0}, {"J2K_L0_WG": 8}, {"J2K_L0_WG": 4, "J2K_L0_XCD": 0}, {"J2K_L0_STORE": 0}, {"J2K_L0_STORE": 4},
                                 {"J2K_L0_WG_INV": 0}, {"J2K_L0_INV_WPE": 6}, {"J2K_HT_ALIAS": 0}, {"J2K_L0_DEAL": 0}])
def test_rgba8_pipeline_knobs(env):
    """packed RGBA8 frame (tiles of 512, 256-wide and short edge tiles): coefficients, HT stream + arrays and reconstructed
    pixels are identical whatever level-0 kernel / store flavour / job order / alias setting produced them"""
    import torch
    from j2kgfx import Context
    from j2kgfx.codec import FramePlan
    W, H = 1280, 624
    rng = np.random.default_rng(11)
    pix_h = rng.integers(0, 256, size=(H, W * 4), dtype=np.uint8)
    pix_h.reshape(H, W, 4)[:, :, 3] = 255
    kw = dict(precision=8, lossless=True, num_resolutions=6, cb=(64, 64), tile=(512, 512), coder=1)
    out = []
    for ctx in (Context(0), _ctx(env)):
        plan = FramePlan(W, H, 3, ctx=ctx, **kw)
        pix = torch.from_numpy(pix_h).to(plan.device)
        coeff = plan.forward_rgba8(pix)
        stream, offs, lens, nb = plan.encode_stream(coeff)
        back = plan.inverse_rgba8(coeff)
        ctx.sync()
        n = int(plan.info.blocks)
        tot = int(offs[n].item())
        out.append((coeff.cpu(), stream[:tot].cpu(), offs[:n + 1].cpu(), lens[:n].cpu(), nb[:n].cpu(), back.cpu()))
        assert torch.equal(out[-1][5], torch.from_numpy(pix_h))
    for a, b in zip(*out):
        assert torch.equal(a, b)


@pytest.mark.parametrize("wg", [0, 6, 12, 16])
def test_lossy_level0_knobs(wg):
    """lossy RGB frame: the quantised coefficients of the general 9-7 kernel and of the workgroup form (any waves per
    workgroup) are identical -- including the Markstein division against the IEEE division of the general kernel"""
    import torch
    from j2kgfx import Context
    from j2kgfx.codec import FramePlan
    W, H = 1024, 600
    rng = np.random.default_rng(12)
    frame_h = rng.integers(0, 4096, size=(3, H, W)).astype(np.int32)
    kw = dict(precision=12, lossless=False, quality=75, num_resolutions=6, cb=(64, 64), tile=(512, 512), coder=0)
    res = []
    for ctx in (Context(0), _ctx({"J2K_L0_WG97": wg})):
        plan = FramePlan(W, H, 3, ctx=ctx, **kw)
        coeff = plan.forward(torch.from_numpy(frame_h).to(plan.device))
        ctx.sync()
        res.append(coeff.cpu())
    assert torch.equal(res[0], res[1])


@pytest.mark.parametrize("quality", [75, 8000])
@pytest.mark.parametrize("wg", [8, 12])
def test_lossy_level0_samples_outside_the_precision(oracle, wg, quality):
    """j2k_plan_forward takes arbitrary int32 planes, and the 9-7 workgroup kernel's fast conversions (v_cvt_i32_f64
    saturates, Go's int32(float64) does not) are only proved for samples inside the declared precision.  Rows with samples
    outside -- up to the full int32 range, where the ICT rounding and the quantiser leave int32 -- must come out as the
    general kernel (go_int32) and the oracle give them; rows inside keep the fast path in the same frame."""
    import torch
    from j2kgfx.codec import FramePlan
    W, H = 512, 96
    rng = np.random.default_rng(quality + wg)
    frame_h = rng.integers(0, 4096, size=(3, H, W)).astype(np.int32)
    frame_h[:, 10:13, :] = rng.integers(-2 ** 31, 2 ** 31, size=(3, 3, W), dtype=np.int64).astype(np.int32)   # rows far outside 12 bits
    frame_h[1, 40, 100:108] = [2 ** 31 - 1, -2 ** 31, 2 ** 31 - 1, 4096, -1, 2 ** 30, -2 ** 30, 65536]
    frame_h[:, 70, :] = 2 ** 31 - 1                                                                       # y + 0.5 lands on 2^31 exactly
    kw = dict(precision=12, lossless=False, quality=quality, num_resolutions=3, cb=(64, 64), tile=(0, 0), coder=0)
    res = []
    for ctx in (_ctx({"J2K_L0_WG97": 0}), _ctx({"J2K_L0_WG97": wg})):
        plan = FramePlan(W, H, 3, ctx=ctx, **kw)
        coeff = plan.forward(torch.from_numpy(frame_h).to(plan.device))
        ctx.sync()
        res.append(coeff.cpu())
    assert torch.equal(res[0], res[1])
    want = oracle.preprocess([frame_h[c] for c in range(3)], W, H, 12, False, 3, quality)
    assert np.array_equal(res[1].numpy().reshape(3, H, W), np.stack(want))


@pytest.mark.parametrize("W,H,tile,nres", [(1024, 600, (512, 512), 6), (512, 77, (0, 0), 3), (64, 3, (0, 0), 2), (16, 2, (0, 0), 2),
                                           (520, 131, (256, 128), 4), (128, 64, (0, 0), 1), (256, 2048, (0, 0), 6)])
@pytest.mark.parametrize("wg", [6, 8, 12])
@pytest.mark.parametrize("span", [12, 31])
def test_lossy_inverse_level0_knobs(W, H, tile, nres, wg, span):
    """lossy RGB inverse on ARBITRARY int32 coefficients: the general 9-7 kernel (J2K_L0_WG97_INV=0) and the workgroup form
    (any waves per workgroup) give the same frame -- odd heights, bands that end inside the halo, one level (no float64
    prefix) and, with span 31, values whose int32(v + 0.5) takes Go's out-of-range conversion (the wave-level redo)"""
    import torch
    from j2kgfx.codec import FramePlan
    rng = np.random.default_rng(W * H + wg + span)
    kw = dict(precision=12, lossless=False, quality=75, num_resolutions=nres, cb=(64, 64), tile=tile, coder=0)
    res = []
    for ctx in (_ctx({"J2K_L0_WG97_INV": 0}), _ctx({"J2K_L0_WG97_INV": wg})):
        plan = FramePlan(W, H, 3, ctx=ctx, **kw)
        if not res:
            n = plan.alloc_coeff().numel()
            lo, hi = -(1 << span), (1 << span)
            coef_h = rng.integers(lo, hi, size=n).astype(np.int32)
            if span == 31: coef_h[rng.random(n) < 0.7] >>= 20          # most windows in range, some not
        back = plan.inverse(torch.from_numpy(coef_h).to(plan.device))
        ctx.sync()
        res.append(back.cpu())
    assert torch.equal(res[0], res[1])


@pytest.mark.parametrize("W,H,tile,cb,prec,lossless", [(512, 384, (256, 256), (64, 64), 12, False), (200, 150, (0, 0), (32, 32), 8, True),
                                                       (256, 256, (0, 0), (128, 128), 12, False), (96, 80, (0, 0), (64, 16), 16, True)])
def test_mq_decode_split_knob(W, H, tile, cb, prec, lossless):
    """MQ block decoder: the plane-stepped paths (J2K_T1_DEC_SPLIT=1: one launch per frame with a block per lane, the same passes as
    launches per plane with J2K_T1_DEC_LANES=1, or round 2's step kernels with J2K_T1_DEC_LANES=0) and the one-launch kernels give the
    same decoded blocks -- mixed block sizes, blocks above 64x64 (which keep the general kernel in all), blocks of very
    different bit-plane counts, all-zero blocks"""
    import torch
    from j2kgfx.codec import FramePlan
    rng = np.random.default_rng(W + H + prec)
    frame_h = rng.integers(0, 1 << prec, size=(3, H, W)).astype(np.int32)
    frame_h[:, :, : (W * 5) // 8] = 1 << (prec - 1)                       # a flat area: blocks with no bit planes at all
    kw = dict(precision=prec, lossless=lossless, quality=75, num_resolutions=4, cb=cb, tile=tile, coder=0)
    res = []
    for ctx in (_ctx({"J2K_T1_DEC_SPLIT": 0}), _ctx({"J2K_T1_DEC_SPLIT": 1}), _ctx({"J2K_T1_DEC_SPLIT": 1, "J2K_T1_DEC_LANES": 0}),
                _ctx({"J2K_T1_DEC_SPLIT": 1, "J2K_T1_DEC_LANES": 1})):
        plan = FramePlan(W, H, 3, ctx=ctx, **kw)
        coeff = plan.forward(torch.from_numpy(frame_h).to(plan.device))
        stream, offs, lens, nb = plan.encode_stream(coeff)
        dec = torch.zeros(int(plan.info.decoded_elems), dtype=torch.int32, device=plan.device)   # (block starts are 4-aligned: the padding is never written)
        plan.decode_blocks(stream, offs, lens, nb, decoded=dec)
        ctx.sync()
        res.append((dec.cpu(), nb.cpu()))
    assert int(res[0][1].max()) > 8
    assert torch.equal(res[0][0], res[1][0])
    assert torch.equal(res[0][0], res[2][0])
    assert torch.equal(res[0][0], res[3][0])


def test_mq_decode_lanes_full_frame():
    """A C3-sized frame (3840 x 2160, 12 bit, 512^2 tiles, 9-7 + Quality 75: 7005 code-blocks, 110 wavefronts of 64 lanes, every lane
    order and plane count the real thing) through the one-launch decoder and through the one-launch-per-frame lanes decoder: the same
    decoded blocks.  (bench.py --config c3 checks the same on its frame 0 after every run.)"""
    import torch
    from j2kgfx.codec import FramePlan
    W, H = 3840, 2160
    rng = np.random.default_rng(33)
    yy, xx = np.mgrid[0:H, 0:W]
    frame_h = np.stack([np.clip(xx * 4095 // W + c * 500 + rng.integers(-300, 301, (H, W)), 0, 4095) for c in range(3)]).astype(np.int32)
    frame_h[:, :200, :] = 2048                                             # a flat band: blocks without bit planes
    res = []
    for ctx in (_ctx({"J2K_T1_DEC_SPLIT": 0}), _ctx({"J2K_T1_DEC_SPLIT": 1})):
        plan = FramePlan(W, H, 3, ctx=ctx, precision=12, lossless=False, quality=75, num_resolutions=6, cb=(64, 64), tile=(512, 512), coder=0)
        coeff = plan.forward(torch.from_numpy(frame_h).to(plan.device))
        stream, offs, lens, nb = plan.encode_stream(coeff)
        dec = torch.zeros(int(plan.info.decoded_elems), dtype=torch.int32, device=plan.device)
        plan.decode_blocks(stream, offs, lens, nb, decoded=dec)
        ctx.sync()
        res.append(dec.cpu())
        assert int(plan.info.blocks) == 7005
        plan.close()
    assert torch.equal(res[0], res[1])


@pytest.mark.parametrize("W,H,tile,nres,quality", [(1024, 600, (512, 512), 6, 75), (512, 77, (0, 0), 4, 30), (256, 2048, (0, 0), 6, 8000),
                                                   (520, 131, (256, 128), 4, 75), (64, 36, (0, 0), 3, 1)])
def test_lossy_deeper_levels_knob(W, H, tile, nres, quality):
    """lossy RGB frame: levels >= 1 of the 9-7 transform on the general marching kernels (J2K_PLANE_WG97=0) and in workgroup
    form give the same quantised coefficients, and the same frame back from ARBITRARY int32 coefficients (odd heights, planes
    down to 16 columns where lanes idle, planes too narrow for the workgroup form next to ones that are not)"""
    import torch
    from j2kgfx.codec import FramePlan
    rng = np.random.default_rng(W * 3 + H + quality)
    frame_h = rng.integers(0, 4096, size=(3, H, W)).astype(np.int32)
    kw = dict(precision=12, lossless=False, quality=quality, num_resolutions=nres, cb=(64, 64), tile=tile, coder=0)
    res = []
    coef_h = None
    for ctx in (_ctx({"J2K_PLANE_WG97": 0}), _ctx({"J2K_PLANE_WG97": 8})):
        plan = FramePlan(W, H, 3, ctx=ctx, **kw)
        coeff = plan.forward(torch.from_numpy(frame_h).to(plan.device))
        if coef_h is None:
            coef_h = rng.integers(-(1 << 20), 1 << 20, size=coeff.numel()).astype(np.int32)
        back = plan.inverse(torch.from_numpy(coef_h).to(plan.device))
        ctx.sync()
        res.append((coeff.cpu(), back.cpu()))
    assert torch.equal(res[0][0], res[1][0])
    assert torch.equal(res[0][1], res[1][1])
