"""GPU (one device): tile sharding is exact -- the concatenation of every rank's shard output equals
the unsharded frame's output, for any world size; and full-size properties at BASELINE sizes."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def run(plan, frame):
    import torch
    d = torch.from_numpy(frame).to(plan.device)
    torch.cuda.synchronize()
    coeff = plan.forward(d)
    slots, lens, nb = plan.encode_blocks(coeff)
    offs, stream = plan.compact(slots, lens)
    plan.ctx.sync()
    n = int(plan.info.blocks)
    total = int(offs.cpu().numpy()[n])
    return stream.cpu().numpy()[:total].copy(), lens.cpu().numpy()[:n].copy(), nb.cpu().numpy()[:n].copy()


@pytest.mark.parametrize("coder", [0, 1])
@pytest.mark.parametrize("world", [2, 3, 8])
def test_shards_concatenate_to_full(coder, world):
    from j2kgfx import dist as jd
    from j2kgfx.codec import FramePlan
    W, H, Cn, tile = 640, 368, 3, 128
    rng = np.random.default_rng(world)
    frame = rng.integers(0, 256, size=(Cn, H, W)).astype(np.int32)
    kw = dict(precision=8, lossless=True, num_resolutions=4, cb=(32, 32), tile=(tile, tile), coder=coder)
    full = run(FramePlan(W, H, Cn, **kw), frame)
    ntiles = jd.num_tiles(W, H, tile, tile)
    parts = []
    for r in range(world):
        first, count = jd.shard_range(ntiles, r, world)
        if count == 0:
            continue
        parts.append(run(FramePlan(W, H, Cn, tile_first=first, tile_count=count, **kw), frame))
    assert np.array_equal(np.concatenate([p[0] for p in parts]), full[0])
    assert np.array_equal(np.concatenate([p[1] for p in parts]), full[1])
    assert np.array_equal(np.concatenate([p[2] for p in parts]), full[2])


@pytest.mark.parametrize("W,H,Cn,tile,prec", [(3840, 2160, 3, 512, 8), (7680, 4320, 3, 512, 10), (2048, 2048, 1, 0, 16)])
def test_full_size_lossless_roundtrip(W, H, Cn, tile, prec):
    """BASELINE configs C2 / C4 / C5 geometry: size-independent property -- forward then inverse
    returns every pixel (5-3 + RCT are exactly invertible), and HT coding is deterministic."""
    import torch
    from j2kgfx.codec import FramePlan
    rng = np.random.default_rng(W)
    frame = torch.from_numpy(rng.integers(0, 1 << prec, size=(Cn, H, W)).astype(np.int32))
    plan = FramePlan(W, H, Cn, precision=prec, lossless=True, num_resolutions=6, cb=(64, 64), tile=(tile, tile), coder=1)
    d = frame.to(plan.device)
    torch.cuda.synchronize()
    coeff = plan.forward(d)
    back = plan.inverse(coeff)
    s1 = plan.encode_blocks(coeff)
    o1, st1 = plan.compact(s1[0], s1[1])
    plan.ctx.sync()
    assert torch.equal(back.cpu(), frame)
    n = int(plan.info.blocks)
    tot = int(o1[n].item())
    a = st1[:tot].clone()
    s2 = plan.encode_blocks(coeff)
    o2, st2 = plan.compact(s2[0], s2[1])
    plan.ctx.sync()
    assert int(o2[n].item()) == tot and torch.equal(st2[:tot], a)
