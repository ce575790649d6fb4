"""dev: host enqueue time per frame vs GPU time (is the bench host-bound?)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "go-jpeg2000_amd"))
import numpy as np, torch
from j2kgfx import CODER_HT, Context
from j2kgfx.codec import FramePlan
W, H, C = 3840, 2160, 3
F = int(sys.argv[1]) if len(sys.argv) > 1 else 3
lanes = []
for f in range(F):
    ctx = Context(0)
    p = FramePlan(W, H, C, precision=8, lossless=True, num_resolutions=6, cb=(64, 64), tile=(512, 512), coder=CODER_HT, ctx=ctx)
    i = p.info; n = int(i.blocks)
    fr = torch.randint(0, 256, (C, H, W), dtype=torch.int32, device=p.device)
    b = dict(ctx=ctx, p=p, fr=fr, co=p.alloc_coeff(), sl=p.empty(i.bytes_cap, torch.uint8), st=p.empty(i.bytes_cap, torch.uint8),
             le=p.empty(n, torch.int32), nb=p.empty(n, torch.uint8), of=p.empty(n + 1, torch.int64), de=p.empty(i.decoded_elems, torch.int32), ba=p.alloc_frame())
    lanes.append(b)
def code(b):
    p = b["p"]
    p.forward(b["fr"], b["co"]); p.encode_blocks(b["co"], b["sl"], b["le"], b["nb"]); p.compact(b["sl"], b["le"], b["of"], b["st"])
    p.decode_blocks(b["st"], b["of"], b["le"], b["nb"], b["de"]); p.inverse(b["co"], b["ba"])
for _ in range(5):
    for b in lanes: code(b)
for b in lanes: b["ctx"].sync()
K = 50
t0 = time.perf_counter()
for _ in range(K):
    for b in lanes: code(b)
t1 = time.perf_counter()
for b in lanes: b["ctx"].sync()
t2 = time.perf_counter()
print("F=%d: host enqueue %.1f us/frame, total %.1f us/frame" % (F, (t1 - t0) / K / F * 1e6, (t2 - t0) / K / F * 1e6))
