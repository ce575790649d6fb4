"""Dev tool: time the HT encode / decode kernels alone on the C2 frame."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "go-jpeg2000_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import bench
from j2kgfx.codec import FramePlan
plan = FramePlan(bench.W, bench.H, bench.C, precision=8, lossless=True, num_resolutions=6, cb=(64, 64), tile=(512, 512), coder=1)
frame = torch.from_numpy(bench.synth_frame(np, 0)).to(plan.device)
torch.cuda.synchronize()
coeff = plan.forward(frame)
slots, lens, nb = plan.encode_blocks(coeff)
offs, stream = plan.compact(slots, lens)
dec = plan.decode_blocks(stream, offs, lens, nb)
plan.ctx.sync()
s = torch.cuda.ExternalStream(plan.ctx.stream)
def timeit(fn, n=30):
    for _ in range(3): fn()
    plan.ctx.sync()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(n): fn()
    e1.record(s); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print("phases=%s enc %.1f us  dec %.1f us" % (os.environ.get("J2K_HT_DEC_PHASES", "all"),
      timeit(lambda: plan.encode_blocks(coeff, slots, lens, nb)), timeit(lambda: plan.decode_blocks(stream, offs, lens, nb, dec))))
