import os, sys
import numpy as np
sys.path[:0] = ["go-jpeg2000_amd", "oracle", "."]
import torch, oracle
from j2kgfx.codec import FramePlan
W, H = 512, 96
quality, wg = 75, 8
rng = np.random.default_rng(quality + wg)
frame_h = rng.integers(0, 4096, size=(3, H, W)).astype(np.int32)
frame_h[:, 10:13, :] = rng.integers(-2 ** 31, 2 ** 31, size=(3, 3, W), dtype=np.int64).astype(np.int32)
frame_h[1, 40, 100:108] = [2 ** 31 - 1, -2 ** 31, 2 ** 31 - 1, 4096, -1, 2 ** 30, -2 ** 30, 65536]
frame_h[:, 70, :] = 2 ** 31 - 1
plan = FramePlan(W, H, 3, precision=12, lossless=False, quality=quality, num_resolutions=3, cb=(64, 64), tile=(0, 0), coder=0)
coeff = plan.forward(torch.from_numpy(frame_h).to(plan.device)); plan.ctx.sync()
got = coeff.cpu().numpy().reshape(3, H, W)
want = np.stack(oracle.preprocess([frame_h[c] for c in range(3)], W, H, 12, False, 3, quality))
bad = np.argwhere(got != want)
print(len(bad), "mismatches")
for b in bad[:20]:
    print(tuple(b), got[tuple(b)], want[tuple(b)])
