"""A/B of the fused pixel formats (J2K_PIX_FUSE=1 against =2: int32 staging frame for the formats round 4 fused, the lossy RGBA path among them): forward_pixels +
inverse_pixels of a 3840x2160 frame, 512x512 tiles, per format; HIP-event time per call over 50 calls."""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "go-jpeg2000_amd"))
from j2kgfx import pixels                    # noqa: E402
from j2kgfx.codec import FramePlan           # noqa: E402
from j2kgfx.context import Context           # noqa: E402

W, H = 3840, 2160
NAMES = ["Gray", "Gray16", "RGBA", "RGBA64", "NRGBA", "NRGBA64"]
BPP = [1, 2, 4, 8, 4, 8]
for fmt, lossless in [(f, True) for f in range(6)] + [(2, False)]:       # the last: image.RGBA through the lossy path (the reference's default options)
    row = []
    for knob in ("1", "2"):
        os.environ["J2K_PIX_FUSE"] = knob
        ctx = Context(0)
        nc, prec = pixels.components(fmt), (16 if fmt in (1, 3, 5) else 8)
        plan = FramePlan(W, H, nc, precision=prec, lossless=lossless, quality=75, num_resolutions=6, cb=(64, 64), tile=(512, 512), coder=1, ctx=ctx)
        pix = torch.randint(0, 256, (H, W * BPP[fmt]), dtype=torch.uint8, device=plan.device)
        obpp = (1 if nc == 1 else 4) * (prec // 8)
        out = torch.zeros((H, W * obpp), dtype=torch.uint8, device=plan.device)
        coeff = plan.alloc_coeff()
        torch.cuda.synchronize()
        s = torch.cuda.ExternalStream(ctx.stream if not callable(ctx.stream) else ctx.stream())
        res = []
        for fn in (lambda: plan.forward_pixels(fmt, pix, coeff), lambda: plan.inverse_pixels(coeff, out)):
            for _ in range(5):
                fn()
            ctx.sync()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            for _ in range(50):
                fn()
            e1.record(s)
            ctx.sync()
            res.append(e0.elapsed_time(e1) / 50 * 1e3)
        row.append((plan.pixels_fused(fmt, pix), res))
        plan.close(); ctx.close()
    (f1, a), (f2, b) = row
    print("%-14s fused=%d/%d  forward %6.1f us (staged %6.1f)   inverse %6.1f us (staged %6.1f)" % (NAMES[fmt] + ("" if lossless else " lossy"), f1, f2, a[0], b[0], a[1], b[1]), flush=True)
