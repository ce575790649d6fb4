"""GPU: every 5-3 level below level 0 in one launch per direction (dwt53_deep.inc, J2K_DEEP) against the per-level launches
and the oracle (dwt.go:524-548).

Shapes chosen for the boundaries of that kernel: the level it streams from memory at 256 / 252 / 128 columns, heights that
leave the deep workgroup 1, 2 or 64 pair-rows and the flat ones none, one or a partial band, odd heights at every level
(missing odd row, mirrored d), a next-level matrix that ends in the middle of a row (odd ceil(h/2)), two to five levels inside
the launch, RGB tiles incl. ragged edge tiles, a frame that does not qualify (falls back), full-range int32 input
(wraparound in both directions)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _ctx(**env):
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


SHAPES = [(512, 512, 0, 6), (512, 200, 0, 5), (1024, 64, 0, 6), (512, 130, 0, 4), (256, 512, 0, 6), (520, 37, 0, 5), (504, 10, 0, 4),
          (512, 5, 0, 4), (512, 3, 0, 4), (512, 510, 0, 6), (512, 259, 0, 6), (2048, 2048, 0, 6), (1024, 1024, 0, 5), (776, 300, 256, 5),
          (16, 16, 0, 3)]


@pytest.mark.parametrize("W,H,tile,nres", SHAPES)
def test_deep_levels_single_component(oracle, W, H, tile, nres):
    import torch
    from j2kgfx.codec import FramePlan
    rng = np.random.default_rng(W * 7 + H + nres)
    full = rng.integers(-2 ** 31, 2 ** 31, (1, H, W), dtype=np.int64).astype(np.int32)
    small = rng.integers(0, 65536, (1, H, W)).astype(np.int32)
    outs = {}
    for deep in (0, 1, 2, 3, 4):                    # 0: per-level launches; 1: default; 2-4: the mid-job variants of both directions
        env = {0: dict(J2K_DEEP=0), 1: dict(J2K_DEEP=1), 2: dict(J2K_DEEP_MID=0), 3: dict(J2K_DEEP_MID_INV=0), 4: dict(J2K_DEEP_MID_INV=2)}[deep]
        env["J2K_DEEP_MIN_PLANES"] = 1              # (by default a frame of fewer than 12 tile-components keeps the per-level launches)
        plan = FramePlan(W, H, 1, precision=16, lossless=True, num_resolutions=nres, cb=(64, 64), tile=(tile, tile), coder=1,
                         ctx=_ctx(**env))
        res = []
        for frame_h in (full, small):
            for rep in range(2):                      # the second pass starts on an idle device (scheduling-dependent bugs)
                frame = torch.from_numpy(frame_h).to(plan.device)
                coeff = plan.forward(frame)
                back = plan.inverse(coeff)
                plan.ctx.sync()
                c, b = coeff.cpu().numpy(), back.cpu().numpy()
                assert np.array_equal(b.reshape(1, H, W), frame_h)
            res += [c, b]
        outs[deep] = res
    for k in (1, 2, 3, 4):
        for a, b in zip(outs[0], outs[k]):
            assert np.array_equal(a, b), k
    if W * H <= 1 << 18 and tile == 0:
        want = oracle.preprocess([small[0]], W, H, 16, True, nres)
        assert np.array_equal(outs[1][2].reshape(H, W), want[0])


@pytest.mark.parametrize("W,H,tile", [(1280, 624, 512), (1536, 1024, 512), (768, 300, 256)])
def test_deep_levels_rgb_tiles(oracle, W, H, tile):
    import torch
    from j2kgfx.codec import FramePlan
    rng = np.random.default_rng(W + H)
    frame_h = rng.integers(0, 256, (3, H, W)).astype(np.int32)
    got = []
    for deep in (0, 1):
        plan = FramePlan(W, H, 3, precision=8, lossless=True, num_resolutions=6, cb=(64, 64), tile=(tile, tile), coder=1,
                         ctx=_ctx(J2K_DEEP=deep, J2K_DEEP_MIN_PLANES=1))
        frame = torch.from_numpy(frame_h).to(plan.device)
        coeff = plan.forward(frame)
        back = plan.inverse(coeff)
        # a decoder's input is arbitrary: the inverse alone on full-range coefficients must agree too
        junk = torch.from_numpy(rng.integers(-2 ** 31, 2 ** 31, coeff.numel(), dtype=np.int64).astype(np.int32)).to(plan.device)
        back2 = plan.inverse(junk) if deep == 0 else plan.inverse(got[0][2].to(plan.device))
        plan.ctx.sync()
        got.append((coeff.cpu().numpy(), back2.cpu().numpy(), junk.cpu()))
        assert np.array_equal(back.cpu().numpy().reshape(3, H, W), frame_h)
    assert np.array_equal(got[0][0], got[1][0]) and np.array_equal(got[0][1], got[1][1])
    # one tile against the oracle's pipeline on the cropped image (top-left tile)
    tw, th = min(tile, W), min(tile, H)
    want = oracle.preprocess([np.ascontiguousarray(frame_h[c, :th, :tw]) for c in range(3)], tw, th, 8, True, 6)
    for c in range(3):
        assert np.array_equal(got[1][0][c * tw * th:(c + 1) * tw * th].reshape(th, tw), want[c])


@pytest.mark.parametrize("W,H,tile", [(1280, 624, 512), (3840, 2160, 512), (512, 512, 0), (768, 300, 256), (1536, 130, 512), (512, 36, 0)])
def test_merged_launches_rgba8(oracle, W, H, tile):
    """J2K_MEGA: the deep levels and the level-0 bands independent of them in one launch per direction (dwt53_mega_*_kernel)
    against level 0 + deep as separate launches (J2K_MEGA=0): same coefficients from packed RGBA8 pixels, same pixels back
    from arbitrary coefficients, both job orders; one tile against the oracle.  Heights that leave no bottom band (36), a single
    short one (130), ragged edge tiles (624 = 512 + 112)."""
    import torch
    from j2kgfx.codec import FramePlan
    rng = np.random.default_rng(W + H + tile)
    pix = rng.integers(0, 256, (H, W * 4)).astype(np.uint8)
    got = []
    junk_h = None
    for mega in (0, 1, 2):
        plan = FramePlan(W, H, 3, precision=8, lossless=True, num_resolutions=6, cb=(64, 64), tile=(tile, tile), coder=1,
                         ctx=_ctx(J2K_MEGA=mega, J2K_DEEP_MIN_PLANES=1))
        dpix = torch.from_numpy(pix).to(plan.device)
        for rep in range(2):
            coeff = plan.forward_rgba8(dpix)
            back = plan.inverse_rgba8(coeff)
            plan.ctx.sync()
            assert np.array_equal(back.cpu().numpy().reshape(H, W, 4)[..., :3], pix.reshape(H, W, 4)[..., :3])
        if junk_h is None:
            junk_h = rng.integers(-2 ** 31, 2 ** 31, coeff.numel(), dtype=np.int64).astype(np.int32)
        back2 = plan.inverse_rgba8(torch.from_numpy(junk_h).to(plan.device))
        plan.ctx.sync()
        got.append((coeff.cpu().numpy(), back2.cpu().numpy()))
    for g in got[1:]:
        assert np.array_equal(got[0][0], g[0]) and np.array_equal(got[0][1], g[1])
    tw, th = min(tile or W, W), min(tile or H, H)
    crop = np.ascontiguousarray(pix.reshape(H, W, 4)[:th, :tw].reshape(th, tw * 4))
    want = oracle.preprocess(oracle.extract_image_data(crop, 2, tw, th), tw, th, 8, True, 6)
    for c in range(3):
        assert np.array_equal(got[1][0][c * tw * th:(c + 1) * tw * th].reshape(th, tw), want[c])
