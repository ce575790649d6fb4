"""dev: time the T1 encoder on 256 x 256 blocks (the reference's default code-block size).  python tools/bench_big.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "go-jpeg2000_amd"))
import numpy as np, torch
from j2kgfx.codec import FramePlan
for name, mk in (("gradient", lambda r, W, H: np.stack([np.mgrid[0:H, 0:W][1] * 255 // W, np.mgrid[0:H, 0:W][0] * 255 // H, (np.mgrid[0:H, 0:W][1] + np.mgrid[0:H, 0:W][0]) * 127 // W])),
                 ("noise", lambda r, W, H: r.integers(0, 256, (3, H, W)))):
    W = H = 512
    fr = mk(np.random.default_rng(1), W, H).astype(np.int32)
    p = FramePlan(W, H, 3, precision=8, lossless=True, num_resolutions=3, cb=(256, 256), coder=0)
    d = torch.from_numpy(fr).to(p.device)
    co = p.forward(d)
    st = p.encode_stream(co); p.ctx.sync()
    t0 = time.perf_counter(); st = p.encode_stream(co); p.ctx.sync(); te = time.perf_counter() - t0
    t0 = time.perf_counter(); de = p.decode_blocks(*st); p.ctx.sync(); td = time.perf_counter() - t0
    n = int(p.info.blocks)
    print("%-9s encode %8.2f ms  decode %8.2f ms  bytes %d  blocks %d  lens max %d" % (name, te * 1e3, td * 1e3, int(st[1][n].item()), n, int(st[2][:n].max().item())))
