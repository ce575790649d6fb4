import sys, numpy as np, torch
sys.path.insert(0, "go-jpeg2000_amd"); sys.path.insert(0, ".")
from j2kgfx.codec import FramePlan
import bench_extra
W, H = 3840, 2160
plan = FramePlan(W, H, 3, precision=8, lossless=True, num_resolutions=6, cb=(64, 64), tile=(512, 512), coder=1)
fr = bench_extra.synth_rgb(np, W, H, 0)
frame = torch.from_numpy(fr).to(plan.device)
coeff = plan.forward(frame)
stream, offs, lens, nb = plan.encode_stream(coeff)
plan.ctx.sync()
for i in range(3):
    dec = plan.decode_blocks(stream, offs, lens, nb)
    plan.ctx.sync()
    print("--- run", i)
