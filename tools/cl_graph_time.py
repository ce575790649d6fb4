"""The closed-loop frame calls (HT coder, C2 frame) replayed from ONE HIP graph per frame against launched one by one: per-frame time alone and with
N contexts in flight.   python tools/cl_graph_time.py [contexts]"""
import os
import sys
import time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "go-jpeg2000_amd"))
from j2kgfx import _lib                    # noqa: E402
from j2kgfx.codec import FramePlan         # noqa: E402
from j2kgfx.context import Context         # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2
W, H = 3840, 2160
rng = np.random.default_rng(1)
yy, xx = np.mgrid[0:H, 0:W]
frame = np.clip(np.stack([xx * 255 // W, yy * 255 // H, (xx + yy) * 127 // W]) + rng.integers(-16, 17, (3, H, W)), 0, 255).astype(np.uint8)
pix = np.full((H, W, 4), 255, np.uint8); pix[..., :3] = frame.transpose(1, 2, 0)
lanes = []
for k in range(N):
    ctx = Context(0)
    plan = FramePlan(W, H, 3, precision=8, lossless=True, num_resolutions=6, cb=(64, 64), tile=(512, 512), coder=1, ctx=ctx, closed_loop=True, track_streams=False)
    d_pix = torch.from_numpy(pix.reshape(H, W * 4)).to(plan.device)
    back = torch.zeros_like(d_pix)
    cs = plan.empty(plan.frame_bound(), torch.uint8); toffs = plan.empty(int(plan.info.tiles) + 1, torch.int64)[:int(plan.info.tiles) + 1]
    def code(plan=plan, d_pix=d_pix, back=back, cs=cs, toffs=toffs):
        plan.encode_frame_pixels(_lib.PIX_RGBA8, d_pix, True, True, cs, toffs)
        plan.decode_frame_pixels(cs, cs.numel(), back, toffs, True, True)
    code(); code(); ctx.sync(); plan.frame_status()
    ref = back.clone()
    with ctx.capture() as g:
        code()
    back.zero_()
    g.launch(); ctx.sync(); plan.frame_status()
    assert torch.equal(back, ref), "graph replay differs"
    lanes.append((ctx, plan, code, g))

def run(fn_of_lane, reps):
    for ln in lanes: fn_of_lane(ln)
    for ln in lanes: ln[0].sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        for ln in lanes: fn_of_lane(ln)
    for ln in lanes: ln[0].sync()
    return (time.perf_counter() - t0) / reps / len(lanes)

t_direct = run(lambda ln: ln[2](), 300)
t_graph = run(lambda ln: ln[3].launch(), 300)
print("closed-loop HT, C2 frame, %d context(s) in flight: launched one by one %.1f us per frame (%.1f Gpixel/s), one graph per frame %.1f us (%.1f Gpixel/s)"
      % (N, t_direct * 1e6, W * H / t_direct / 1e9, t_graph * 1e6, W * H / t_graph / 1e9))
for ln in lanes: ln[1].frame_status()
