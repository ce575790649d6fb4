"""dev tool: an 8K 12-bit MQ frame (27 540 code-blocks) through the one-launch block decoder and through the plane-stepped lanes\ndecoder (J2K_T1_DEC_SPLIT=1): same decoded blocks, and what each costs alone.   python tools/check_big_mq.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "go-jpeg2000_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from j2kgfx import Context
from j2kgfx.codec import FramePlan
W, H = 7680, 4320
rng = np.random.default_rng(5)
yy, xx = np.mgrid[0:H, 0:W]
fr = np.stack([np.clip((xx * 4095 // W + c * 300 + rng.integers(-200, 201, (H, W))), 0, 4095) for c in range(3)]).astype(np.int32)
res = []
for env in ({"J2K_T1_DEC_SPLIT": "0"}, {"J2K_T1_DEC_SPLIT": "1"}):
    os.environ.update(env)
    ctx = Context(0)
    p = FramePlan(W, H, 3, precision=12, lossless=False, quality=75, num_resolutions=6, cb=(64, 64), tile=(512, 512), coder=0, ctx=ctx)
    d = torch.from_numpy(fr).to(p.device)
    co = p.forward(d)
    stream, offs, lens, nb = p.encode_stream(co)
    dec = torch.zeros(int(p.info.decoded_elems), dtype=torch.int32, device=p.device)
    import time
    ctx.sync(); t0 = time.perf_counter()
    p.decode_blocks(stream, offs, lens, nb, decoded=dec)
    ctx.sync(); dt = time.perf_counter() - t0
    print(env, "blocks", int(p.info.blocks), "decode %.1f ms" % (dt * 1e3), flush=True)
    res.append(dec.cpu())
    p.close()
print("equal:", bool(torch.equal(res[0], res[1])))
# the encoder's MQ lanes kernel at this size: chains per wavefront (J2K_T1_LANES; 0 = the library's choice for a lone context)
import time
for K in ("0", "16", "32", "64"):
    os.environ["J2K_T1_LANES"] = K
    ctx = Context(0)
    p = FramePlan(W, H, 3, precision=12, lossless=False, quality=75, num_resolutions=6, cb=(64, 64), tile=(512, 512), coder=0, ctx=ctx)
    d = torch.from_numpy(fr).to(p.device)
    co = p.forward(d)
    p.encode_stream(co); ctx.sync()
    t0 = time.perf_counter()
    for _ in range(3): p.encode_stream(co)
    ctx.sync()
    print("J2K_T1_LANES=%s encode_stream %.1f ms" % (K, (time.perf_counter() - t0) / 3 * 1e3), flush=True)
    p.close()
