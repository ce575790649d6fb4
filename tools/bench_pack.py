"""dev: time of the transport pack / unpack of one C2 frame's stream (what a peer and the root add per frame at N > 1)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "go-jpeg2000_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch, bench
from j2kgfx.codec import FramePlan
p = FramePlan(3840, 2160, 3, precision=8, lossless=True, num_resolutions=6, cb=(64, 64), tile=(512, 512), coder=1)
d = torch.from_numpy(bench.synth_frame(np, 1)).to(p.device)
co = p.forward(d)
s, o, l, nb = p.encode_stream(co)
pk = p.pack_stream(s, o, l, nb)
out = p.unpack_stream(pk)
p.ctx.sync()
n = int(p.info.blocks)
print("stream %d bytes, pack %d bytes" % (int(o[n].item()), int(pk[:8].view(torch.int64)[0].item())))
outs7 = [tuple(t.clone() for t in out) for _ in range(7)]
pk7 = [pk.clone() for _ in range(7)]
def seven_launches():
    for a, b in zip(pk7, outs7): p.unpack_stream(a, *b)
for name, f in (("pack", lambda: p.pack_stream(s, o, l, nb, pk)), ("unpack", lambda: p.unpack_stream(pk, *out)),
                ("7 unpacks, 7 launches", seven_launches), ("7 unpacks, 1 launch", lambda: p.unpack_streams(pk7, outs7))):
    f(); p.ctx.sync()
    t0 = time.perf_counter()
    for _ in range(50): f()
    p.ctx.sync()
    print("%-22s %7.1f us" % (name, (time.perf_counter() - t0) / 50 * 1e6))
