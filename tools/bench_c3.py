"""dev: C3-like pipeline timing (9-7 lossy 12-bit, MQ coder) per stage.   python tools/bench_c3.py [coder] [lossless]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "go-jpeg2000_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch, bench
from j2kgfx.codec import FramePlan
coder = int(sys.argv[1]) if len(sys.argv) > 1 else 0
lossless = int(sys.argv[2]) if len(sys.argv) > 2 else 0
prec = 8 if lossless else 12
fr = bench.synth_frame(np, 1)
if not lossless: fr = (fr.astype(np.int64) * 4095 // 255).astype(np.int32)
p = FramePlan(3840, 2160, 3, precision=prec, lossless=bool(lossless), quality=75, num_resolutions=6, cb=(64, 64), tile=(512, 512), coder=coder)
d = torch.from_numpy(fr).to(p.device)
i = p.info; n = int(i.blocks)
co = p.alloc_coeff(); sl = p.empty(i.bytes_cap, torch.uint8); st = p.empty(i.bytes_cap, torch.uint8)
le = p.empty(n, torch.int32); nb = p.empty(n, torch.uint8); of = p.empty(n + 1, torch.int64); de = p.empty(i.decoded_elems, torch.int32); ba = p.alloc_frame()
stages = [("forward", lambda: p.forward(d, co)), ("encode_blocks", lambda: p.encode_blocks(co, sl, le, nb)), ("compact", lambda: p.compact(sl, le, of, st)),
          ("decode_blocks", lambda: p.decode_blocks(st, of, le, nb, de)), ("inverse", lambda: p.inverse(co, ba))]
for _, f in stages: f()
p.ctx.sync()
tot = 0
for name, f in stages:
    K = 5
    t0 = time.perf_counter()
    for _ in range(K): f()
    p.ctx.sync()
    dt = (time.perf_counter() - t0) / K
    tot += dt
    print("%-14s %9.1f us" % (name, dt * 1e6))
print("total %.1f us -> %.1f Mpx/s; compressed %d bytes" % (tot * 1e6, 3840 * 2160 / tot / 1e6, int(of[n].item())))
