"""Time of the device packet coder on a C2 frame (3840x2160 RGB8, 512x512 tiles, HT, 64x64 blocks: 8 430 code-blocks, 720 packets):
j2k_plan_t2_fill_cbs + j2k_t2_encode_packets_device per call (the call synchronises: wall time per call over 30 calls)."""
import os
import sys
import time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "go-jpeg2000_amd"))
from j2kgfx import t2                      # noqa: E402
from j2kgfx.codec import FramePlan         # noqa: E402
from j2kgfx.context import Context         # noqa: E402

ctx = Context(0)
W, H = 3840, 2160
rng = np.random.default_rng(1)
yy, xx = np.mgrid[0:H, 0:W]
frame = np.clip(np.stack([xx * 255 // W, yy * 255 // H, (xx + yy) * 127 // W]) + rng.integers(-16, 17, (3, H, W)), 0, 255).astype(np.int32)
for coder, name in ((1, "HT"), (0, "MQ")):
    plan = FramePlan(W, H, 3, precision=8, lossless=True, num_resolutions=6, cb=(64, 64), tile=(512, 512), coder=coder, ctx=ctx)
    coeff = plan.forward(torch.from_numpy(frame).to(plan.device))
    stream, offs, lens, numbps = plan.encode_stream(coeff)
    ctx.sync()
    packets = plan.t2_packets(0)
    d_packets = torch.from_numpy(packets.view(np.uint8).copy()).to(plan.device)
    total_stream = int(offs[-1].item())
    out = torch.zeros(total_stream + 64 * len(packets) + 8 * int(plan.info.blocks), dtype=torch.uint8, device=plan.device)
    poffs = torch.zeros(len(packets) + 1, dtype=torch.int64, device=plan.device)
    cbs = plan.t2_fill_cbs(12, offs, lens, numbps)
    enc = t2.DevicePacketEncoder(ctx)
    for _ in range(3):
        enc.encode(d_packets, len(packets), cbs, stream, True, True, out, poffs)
    t0 = time.perf_counter()
    for _ in range(30):
        plan.t2_fill_cbs(12, offs, lens, numbps, cbs)
        total = enc.encode(d_packets, len(packets), cbs, stream, True, True, out, poffs)
    dt = (time.perf_counter() - t0) / 30
    print("%s: %d blocks, %d packets, %d stream bytes -> %d packet bytes: %.1f us per call (%.2f GB/s of bodies)" %
          (name, int(plan.info.blocks), len(packets), total_stream, total, dt * 1e6, total_stream / dt / 1e9), flush=True)
    plan.close()
