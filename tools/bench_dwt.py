"""Quick DWT-only timing (dev tool): python tools/bench_dwt.py [W H C tile nres]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "go-jpeg2000_amd"))
import numpy as np, torch
from j2kgfx.codec import FramePlan
W, H, Cn, tile, nres = (int(v) for v in (sys.argv[1:6] + ["3840", "2160", "3", "512", "6"][len(sys.argv) - 1:]))
plan = FramePlan(W, H, Cn, precision=8, lossless=True, num_resolutions=nres, tile=(tile, tile))
rng = np.random.default_rng(0)
frame = torch.from_numpy(rng.integers(0, 256, size=(Cn, H, W)).astype(np.int32)).to(plan.device)
coeff = plan.alloc_coeff(); back = plan.alloc_frame()
s = torch.cuda.ExternalStream(plan.ctx.stream)
torch.cuda.synchronize()
def timeit(fn, n=50):
    for _ in range(5): fn()
    plan.ctx.sync()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(n): fn()
    e1.record(s); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
tf = timeit(lambda: plan.forward(frame, coeff))
ti = timeit(lambda: plan.inverse(coeff, back))
b = plan.info.dwt_bytes
print("band=%s W=%d H=%d C=%d tile=%d nres=%d: fwd %.1f us %.0f GB/s | inv %.1f us %.0f GB/s | alg bytes %.1f MB (level0 %.1f MB) %.0f Mpx/s fwd"
      % (os.environ.get("J2K_BAND_PROWS", "16"), W, H, Cn, tile, nres, tf * 1e6, b / tf / 1e9, ti * 1e6, b / ti / 1e9, b / 1e6, plan.info.dwt_level0_bytes / 1e6, W * H / tf / 1e6))
assert torch.equal(back, frame)
