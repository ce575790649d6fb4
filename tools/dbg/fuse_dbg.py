import sys, numpy as np, torch
sys.path.insert(0, "go-jpeg2000_amd")
from j2kgfx.codec import FramePlan
for (W, H, tile) in [(1024, 1024, 512), (2048, 2048, 512), (4096, 512, 512), (512, 4096, 512), (3840, 2160, 512), (3840, 2160, 512)]:
    rng = np.random.default_rng(1)
    pix = rng.integers(0, 256, (H, W * 4)).astype(np.uint8)
    plan = FramePlan(W, H, 3, precision=8, lossless=True, num_resolutions=6, cb=(64, 64), tile=(tile, tile), coder=1)
    dpix = torch.from_numpy(pix).to(plan.device)
    planes = np.stack([pix.reshape(H, W, 4)[..., c].astype(np.int32) for c in range(3)])
    frame = torch.from_numpy(planes).to(plan.device)
    want = plan.forward(frame); plan.ctx.sync()
    got = plan.forward_rgba8(dpix); plan.ctx.sync()
    got2 = plan.forward_rgba8(dpix); plan.ctx.sync()
    bad = (got != want).nonzero().flatten().cpu().numpy()
    bad2 = (got2 != want).nonzero().flatten().cpu().numpy()
    print(W, H, tile, "mismatches", bad.size, bad2.size, "same set", np.array_equal(bad, bad2))
    if bad.size:
        # coefficient layout: component-major frame, tile-local dense
        comp = bad // (W * H); rem = bad % (W * H)
        print(" comps", np.unique(comp), "first", bad[:10], "last", bad[-5:])
        print(" rem//tilearea", np.unique(rem // (tile * tile))[:50])
        t0 = rem[(rem // (tile*tile)) == (rem[0] // (tile*tile))] % (tile*tile)
        print(" within first bad tile: rows(512-wide)", np.unique(t0 // 512)[:40], "n", t0.size)
    del plan
