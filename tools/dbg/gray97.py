import sys
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "go-jpeg2000_amd"))
import numpy as np, torch
from j2kgfx.codec import FramePlan
from j2kgfx import dwt
rng = np.random.default_rng(1)
frame = rng.integers(0, 4096, size=(1, 256, 512)).astype(np.int32)
plan = FramePlan(512, 256, 1, precision=12, lossless=False, quality=50, num_resolutions=4, tile=(0, 0))
d = torch.from_numpy(frame).to(plan.device)
c = plan.forward(d); b = plan.inverse(c); plan.ctx.sync()
x = rng.standard_normal(256 * 112)
y = x.copy(); dwt.DecomposeMultiLevel97(y, 256, 112, 3); dwt.ReconstructMultiLevel97(y, 256, 112, 3)
print("ok")
