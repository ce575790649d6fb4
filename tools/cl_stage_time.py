"""Per-stage device time of the closed-loop frame codec on the C2 geometry (3840x2160 RGB8, 512x512 tiles, 64x64 blocks), one frame alone:
HIP events around each stage call, averaged over 30 frames.   python tools/cl_stage_time.py [ht|mq]"""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "go-jpeg2000_amd"))
from j2kgfx import _lib                    # noqa: E402
from j2kgfx.codec import FramePlan         # noqa: E402
from j2kgfx.context import Context         # noqa: E402

coder = 0 if (len(sys.argv) > 1 and sys.argv[1] == "mq") else 1
ctx = Context(0)
W, H = 3840, 2160
rng = np.random.default_rng(1)
yy, xx = np.mgrid[0:H, 0:W]
frame = np.clip(np.stack([xx * 255 // W, yy * 255 // H, (xx + yy) * 127 // W]) + rng.integers(-16, 17, (3, H, W)), 0, 255).astype(np.uint8)
pix = np.full((H, W, 4), 255, np.uint8); pix[..., :3] = frame.transpose(1, 2, 0)
plan = FramePlan(W, H, 3, precision=8, lossless=True, num_resolutions=6, cb=(64, 64), tile=(512, 512), coder=coder, ctx=ctx, closed_loop=True, track_streams=False)
d_pix = torch.from_numpy(pix.reshape(H, W * 4)).to(plan.device)
back = torch.zeros_like(d_pix)
ext = torch.cuda.ExternalStream(ctx.stream)
n = int(plan.info.blocks)
coeff = plan.alloc_coeff(); coeff2 = plan.alloc_coeff()
stream = plan.empty(plan.info.bytes_cap, torch.uint8); offs = plan.empty(n + 1, torch.int64); lens = plan.empty(n, torch.int32); nb = plan.empty(n, torch.uint8)
cs = plan.empty(plan.frame_bound(), torch.uint8); toffs = plan.empty(int(plan.info.tiles) + 1, torch.int64)[:int(plan.info.tiles) + 1]
o2 = plan.empty(n + 1, torch.int64); l2 = plan.empty(n, torch.int32); n2 = plan.empty(n, torch.uint8)
decoded = plan.empty(plan.info.decoded_elems, torch.int32)
cs2 = plan.empty(plan.frame_bound(), torch.uint8); toffs2 = plan.empty(int(plan.info.tiles) + 1, torch.int64)[:int(plan.info.tiles) + 1]
plan.encode_tile_parts(plan.encode_stream(plan.forward_pixels(_lib.PIX_RGBA8, d_pix, coeff), stream, offs, lens, nb)[0], offs, lens, nb, True, True, cs2, toffs2)
cs3 = plan.empty(plan.frame_bound(), torch.uint8); toffs3 = plan.empty(int(plan.info.tiles) + 1, torch.int64)[:int(plan.info.tiles) + 1]
stages = [
    ("encode_frame_pixels (whole: slots -> tile-parts direct)", lambda: plan.encode_frame_pixels(_lib.PIX_RGBA8, d_pix, True, True, cs3, toffs3)),
    ("decode_frame_pixels (whole)", lambda: plan.decode_frame_pixels(cs3, cs3.numel(), back, toffs3, True, True)),
    ("forward_pixels", lambda: plan.forward_pixels(_lib.PIX_RGBA8, d_pix, coeff)),
    ("encode_stream", lambda: plan.encode_stream(coeff, stream, offs, lens, nb)),
    ("encode_tile_parts", lambda: plan.encode_tile_parts(stream, offs, lens, nb, False, False, cs, toffs)),
    ("decode_tile_parts (SOP + EPH: packets side by side)", lambda: plan.decode_tile_parts(cs2, cs2.numel(), toffs2, True, True, o2, l2, n2)),
    ("decode_tile_parts (SOP + EPH, t2_parallel = 0)", lambda: (ctx.set_option("t2_parallel", 0), plan.decode_tile_parts(cs2, cs2.numel(), toffs2, True, True, o2, l2, n2), ctx.set_option("t2_parallel", 1))),
    ("decode_tile_parts", lambda: plan.decode_tile_parts(cs, cs.numel(), toffs, False, False, o2, l2, n2)),
    ("decode_tile_parts (SOT walk)", lambda: plan.decode_tile_parts(cs, cs.numel(), None, False, False, o2, l2, n2)),
    ("decode_blocks", lambda: plan.decode_blocks(cs, o2, l2, n2, decoded)),
    ("place_blocks", lambda: plan.place_blocks(decoded, coeff2)),
    ("inverse_pixels", lambda: plan.inverse_pixels(coeff2, back)),
]
for _, f in stages:
    f()
ctx.sync()
tot = {k: 0.0 for k, _ in stages}
REP = 30
for _ in range(REP):
    for k, f in stages:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(ext); f(); e1.record(ext)
        ctx.sync()
        tot[k] += e0.elapsed_time(e1) * 1e3
plan.frame_status()
if coder == 0:
    assert torch.equal(back, d_pix)
print("closed-loop C2 frame, %s coder, %d blocks, %d bytes of tile-parts; us per stage (one frame alone, events around the call):" % ("MQ" if coder == 0 else "HT", n, int(toffs[-1].item())))
for k, _ in stages:
    print("  %-52s %9.1f" % (k, tot[k] / REP))
plan.frame_parallel_tiles()
plan.decode_tile_parts(cs2, cs2.numel(), toffs2, True, True, o2, l2, n2)
print("tiles parsed packet-parallel: %d of %d" % (plan.frame_parallel_tiles(), int(plan.info.tiles)))
