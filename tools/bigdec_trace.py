"""dev tool (GPU box): the (context, decision) trace of t1_decode_big_kernel against the Python restatement's, first divergence with the
neighbourhood the reference sees.  bash tools/variant.sh trace t1.hip -DJ2K_TBD_TRACE ; J2K_LIB=.../libj2kgfx_trace.so python tools/bigdec_trace.py W H numBPS band seed"""
import sys, os
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [os.path.join(R, "go-jpeg2000_amd"), os.path.join(R, "oracle")]
import numpy as np
import pyref
from j2kgfx import entropy as ent
w, h, nb, band = (int(v) for v in sys.argv[1:5])
seed = int(sys.argv[5])
rng = np.random.default_rng(seed)
g = rng.integers(0, 256, 4000).astype(np.uint8)
g[g == 0xFF] = 0x7F
got = ent.NewT1(w, h).Decode(bytes(g), nb, band).view(np.uint16)
tr = []
last = {}
orig = pyref.MQDecoder.decode
class Stop(Exception): pass
def rec(self, ctx):
    d = orig(self, ctx)
    i = len(tr)
    tr.append((ctx, d))
    if i < got.size:
        c, dd = int(got[i]) & 0xFF, int(got[i]) >> 8
        if (c, dd) != (ctx, d):
            raise Stop()
    return d
pyref.MQDecoder.decode = rec
for name in ("zc", "sc", "mr"):
    def mk(name):
        o = getattr(pyref.T1, name)
        def f(self, x, y):
            last["at"] = (name, x, y); last["t"] = self
            return o(self, x, y)
        return f
    setattr(pyref.T1, name, mk(name))
try:
    pyref.t1_decode(bytes(g), nb, band, w, h)
    print("traces equal over", min(len(tr), got.size), "decisions of", len(tr))
except Stop:
    i = len(tr) - 1
    print("first divergence at decision", i, "gpu", (int(got[i]) & 0xFF, int(got[i]) >> 8), "ref", tr[i], "at", last["at"])
    t = last["t"]; _, x, y = last["at"]
    for yy in range(y - 1, y + 2):
        print("   ", [("." if not (0 <= xx < w and 0 <= yy < h) else ("-" if t.neg(xx, yy) else "+") if t.sig(xx, yy) else "0") for xx in range(x - 2, x + 3)])
    print("ref  ", tr[max(0, i - 12):i + 1])
    print("gpu  ", [(int(v) & 0xFF, int(v) >> 8) for v in got[max(0, i - 12):i + 1]])
