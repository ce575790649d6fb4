"""dev: C5-like frame (2048x2048 gray 16-bit, 5-3 lossless, untiled, HT) per-stage timing + frames-in-flight throughput.
python tools/bench_c5.py [frames in flight] [io: gray16 | planes]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "go-jpeg2000_amd"))
import numpy as np, torch
from j2kgfx import Context
from j2kgfx.codec import FramePlan
W = H = 2048
F = int(sys.argv[1]) if len(sys.argv) > 1 else 4
IO = sys.argv[2] if len(sys.argv) > 2 else "gray16"
rng = np.random.default_rng(5)
yy, xx = np.mgrid[0:H, 0:W]
fr = np.clip((xx * 65535 // W + yy * 65535 // H) // 2 + rng.integers(-2000, 2001, (H, W)), 0, 65535).astype(np.int32)[None]
lanes = []
for f in range(F):
    ctx = Context(0)
    p = FramePlan(W, H, 1, precision=16, lossless=True, num_resolutions=6, cb=(64, 64), tile=(0, 0), coder=1, ctx=ctx)
    i = p.info; n = int(i.blocks)
    lanes.append(dict(ctx=ctx, p=p, d=torch.from_numpy(fr).to(p.device), co=p.alloc_coeff(), st=p.empty(i.bytes_cap, torch.uint8),
                      pix=torch.from_numpy(np.ascontiguousarray(fr[0].astype(">u2")).view(np.uint8).reshape(H, W * 2)).to(p.device),
                      bpix=torch.zeros((H, W * 2), dtype=torch.uint8, device=p.device),
                      le=p.empty(n, torch.int32), nb=p.empty(n, torch.uint8), of=p.empty(n + 1, torch.int64), de=p.empty(i.decoded_elems, torch.int32), ba=p.alloc_frame()))
b = lanes[0]; p = b["p"]
fwd = (lambda q, b: q.forward_pixels(1, b["pix"], b["co"])) if IO == "gray16" else (lambda q, b: q.forward(b["d"], b["co"]))
inv = (lambda q, b: q.inverse_pixels(b["co"], b["bpix"])) if IO == "gray16" else (lambda q, b: q.inverse(b["co"], b["ba"]))
stages = [("forward", lambda: fwd(p, b)), ("encode_stream", lambda: p.encode_stream(b["co"], b["st"], b["of"], b["le"], b["nb"])),
          ("decode_blocks", lambda: p.decode_blocks(b["st"], b["of"], b["le"], b["nb"], b["de"])), ("inverse", lambda: inv(p, b))]
for _, f in stages: f()
p.ctx.sync()
for name, f in stages:
    K = 20
    t0 = time.perf_counter()
    for _ in range(K): f()
    p.ctx.sync()
    print("%-14s %8.1f us" % (name, (time.perf_counter() - t0) / K * 1e6))
if IO == "planes": assert torch.equal(b["ba"], b["d"])    # (Gray16 pixels >= 32769 do not survive the reference's createImage: v * 65535 / 65535 wraps in int32)
def code(b):
    q = b["p"]
    fwd(q, b); q.encode_stream(b["co"], b["st"], b["of"], b["le"], b["nb"]); q.decode_blocks(b["st"], b["of"], b["le"], b["nb"], b["de"]); inv(q, b)
for _ in range(3):
    for b in lanes: code(b)
for b in lanes: b["ctx"].sync()
K = 30
t0 = time.perf_counter()
for _ in range(K):
    for b in lanes: code(b)
for b in lanes: b["ctx"].sync()
dt = (time.perf_counter() - t0) / K / F
print("F=%d in flight: %.1f us/frame, %.1f Gpx/s, %d blocks, %d compressed bytes" % (F, dt * 1e6, W * H / dt / 1e9, int(lanes[0]["p"].info.blocks), int(lanes[0]["of"][int(lanes[0]["p"].info.blocks)].item())))
