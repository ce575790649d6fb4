import sys, numpy as np, torch
sys.path.insert(0, "go-jpeg2000_amd")
from j2kgfx.codec import FramePlan
W = H = 512; tile = 0
rng = np.random.default_rng(1)
pix = rng.integers(0, 256, (H, W * 4)).astype(np.uint8)
plan = FramePlan(W, H, 3, precision=8, lossless=True, num_resolutions=6, cb=(64, 64), tile=(tile, tile), coder=1)
dpix = torch.from_numpy(pix).to(plan.device)
planes = np.stack([pix.reshape(H, W, 4)[..., c].astype(np.int32) for c in range(3)])
frame = torch.from_numpy(planes).to(plan.device)
want = plan.forward(frame); plan.ctx.sync()
want = want.cpu().numpy().reshape(3, H, W)
for it in range(4):
    got = plan.forward_rgba8(dpix); plan.ctx.sync()
    got = got.cpu().numpy().reshape(3, H, W)
    k, r, c = np.nonzero(got != want)
    print("iter", it, "n", k.size)
    if not k.size: continue
    print(" comps", np.bincount(k, minlength=3), "rows", np.unique(r)[:60], "...", np.unique(r)[-5:])
    print(" cols%64", np.unique(c % 64), "col range", c.min(), c.max())
    t = r - 256
    print(" t%5 hist", np.bincount(t % 5, minlength=5))
    for i in range(0, min(k.size, 12)):
        print("  ", k[i], r[i], c[i], "got", got[k[i], r[i], c[i]], "want", want[k[i], r[i], c[i]], "want row-1/+1", want[k[i], r[i]-1, c[i]], want[k[i], min(r[i]+1,H-1), c[i]])
