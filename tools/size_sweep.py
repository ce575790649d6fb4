"""Does the C2 pipeline's rate depend on the working set (infinity cache)?  The C2 step (packed RGBA8 in and out, 5-3 + HT, 512x512
tiles) on frames of several sizes, F frames in flight: Gpixel/s."""
import os
import sys
import time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "go-jpeg2000_amd"))
from j2kgfx.codec import FramePlan         # noqa: E402
from j2kgfx.context import Context         # noqa: E402

for (W, H) in ((3840, 2160), (2560, 1536), (2048, 1024), (1024, 1024)):
    for F in (2, 3, 4, 6, 8):
        lanes = []
        for f in range(F):
            ctx = Context(0)
            p = FramePlan(W, H, 3, precision=8, lossless=True, num_resolutions=6, cb=(64, 64), tile=(512, 512), coder=1, ctx=ctx, track_streams=False)
            rng = np.random.default_rng(f)
            yy, xx = np.mgrid[0:H, 0:W]
            fr = np.clip(np.stack([xx * 255 // W, yy * 255 // H, (xx + yy) * 127 // max(W, H)]) + rng.integers(-16, 17, (3, H, W)), 0, 255)
            rgba = np.concatenate([fr.transpose(1, 2, 0), np.full((H, W, 1), 255)], axis=2).astype(np.uint8).reshape(H, W * 4)
            i = p.info; n = int(i.blocks)
            ln = dict(ctx=ctx, p=p, pix=torch.from_numpy(rgba).to(p.device), coeff=p.alloc_coeff(), stream=p.empty(i.bytes_cap, torch.uint8),
                      lens=p.empty(n, torch.int32), nb=p.empty(n, torch.uint8), offs=p.empty(n + 1, torch.int64),
                      decoded=torch.zeros(max(int(i.decoded_elems), 4), dtype=torch.int32, device=p.device))
            ln["back"] = torch.empty_like(ln["pix"])
            p.set_decode_coded_rows_only(True)
            lanes.append(ln)

        def code(ln):
            p = ln["p"]
            p.forward_rgba8(ln["pix"], ln["coeff"])
            p.encode_stream(ln["coeff"], ln["stream"], ln["offs"], ln["lens"], ln["nb"])
            p.decode_blocks(ln["stream"], ln["offs"], ln["lens"], ln["nb"], ln["decoded"])
            p.inverse_rgba8(ln["coeff"], ln["back"])

        def barrier():
            for ln in lanes:
                ln["ctx"].sync()
            torch.cuda.synchronize()
        for _ in range(10):
            for ln in lanes:
                code(ln)
        barrier()
        steps = max(20, int(200 * (3840 * 2160) / (W * H) / F))
        t0 = time.perf_counter()
        for _ in range(steps):
            for ln in lanes:
                code(ln)
        barrier()
        dt = time.perf_counter() - t0
        print("%4d x %4d, %d in flight: %6.1f Gpixel/s  (%.1f us per frame)" % (W, H, F, F * W * H * steps / dt / 1e9, dt / steps / F * 1e6), flush=True)
        for ln in lanes:
            assert torch.equal(ln["back"], ln["pix"])
            ln["p"].close(); ln["ctx"].close()
