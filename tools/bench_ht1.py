"""Dev tool: latency of HT decode for a handful of 64x64 blocks (single-block latency)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "go-jpeg2000_amd"))
import numpy as np, torch
from j2kgfx.codec import FramePlan
W = int(sys.argv[1]) if len(sys.argv) > 1 else 64
plan = FramePlan(W, 64, 1, precision=8, lossless=True, num_resolutions=1, cb=(64, 64), coder=1)
rng = np.random.default_rng(0)
frame = torch.from_numpy(rng.integers(0, 256, size=(1, 64, W)).astype(np.int32)).to(plan.device)
torch.cuda.synchronize()
coeff = plan.forward(frame)
slots, lens, nb = plan.encode_blocks(coeff)
offs, stream = plan.compact(slots, lens)
dec = plan.decode_blocks(stream, offs, lens, nb)
plan.ctx.sync()
s = torch.cuda.ExternalStream(plan.ctx.stream)
def timeit(fn, n=50):
    for _ in range(3): fn()
    plan.ctx.sync()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(n): fn()
    e1.record(s); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print("blocks=%d phases=%s dec %.1f us enc %.1f us" % (plan.info.blocks, os.environ.get("J2K_HT_DEC_PHASES", "all"),
      timeit(lambda: plan.decode_blocks(stream, offs, lens, nb, dec)), timeit(lambda: plan.encode_blocks(coeff, slots, lens, nb))))
