"""dev tool (GPU box): phase spans of the deep 5-3 kernels from a -DJ2K_DEEP_STAMP build.
bash tools/variant.sh stamp dwt53.hip -DJ2K_DEEP_STAMP ; J2K_LIB=go-jpeg2000_amd/build/libj2kgfx_stamp.so python tools/deep_stamps.py"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "go-jpeg2000_amd"), ROOT]
import torch
from j2kgfx import _lib
from j2kgfx.codec import FramePlan
L = _lib.lib()
W, H = 3840, 2160
plan = FramePlan(W, H, 3, precision=8, lossless=True, num_resolutions=6, cb=(64, 64), tile=(512, 512), coder=1)
rng = np.random.default_rng(1)
pix = torch.from_numpy(rng.integers(0, 256, (H, W * 4)).astype(np.uint8)).to(plan.device)
out = torch.zeros_like(pix)
for it in range(3):
    coeff = plan.forward_rgba8(pix)
    plan.inverse_rgba8(coeff, out)
    plan.ctx.sync()
buf = np.zeros((2, 1024, 8), dtype=np.uint64)
rc = L.j2k_dev_deep_stamps(buf.ctypes.data_as(C.c_void_p))
assert rc == 0, rc
for d, name in ((0, "fwd"), (1, "inv")):
    s = buf[d].astype(np.int64)
    live = s[:, 0] > 0
    t0 = s[live, 0].min()
    print(name, "workgroups", int(live.sum()), " (10 ns ticks, relative to the first workgroup's start)")
    isdeep = (s[:, 3] > 0) if d == 0 else (s[:, 1] > 0)
    for sel, label in ((isdeep, "deep"), (~isdeep, "flat")):
        sub = s[sel & live]
        if not len(sub): continue
        rel = sub - t0
        cols = []
        for k in range(8):
            v = rel[:, k][sub[:, k] > 0]
            cols.append("k%d: %s" % (k, "-" if not len(v) else "%d/%d/%d" % (v.min(), int(np.median(v)), v.max())))
        print("  %s n=%d  min/med/max  %s" % (label, len(sub), "  ".join(cols)))
        order = [k for k in ((0, 1, 2, 3, 4, 5) if d == 0 else (0, 3, 2, 1, 6, 7)) if (sub[:, k] > 0).all()]
        dl = []
        for a, b in zip(order, order[1:]):
            v = sub[:, b] - sub[:, a]
            dl.append("k%d->k%d: %d/%d/%d" % (a, b, v.min(), int(np.median(v)), v.max()))
        print("     per-workgroup spans  " + "  ".join(dl))

wb = np.zeros((256, 16, 4), dtype=np.uint64)
if hasattr(L, "j2k_dev_deep_wstamps") or True:
    try:
        rc = L.j2k_dev_deep_wstamps(wb.ctypes.data_as(C.c_void_p))
        wb = wb.astype(np.int64)
        for wg in (0, 1, 40, 100):
            t0 = wb[wg, :, 0].min()
            print("wg %d first LDS level, per wave begin/end/after-barrier (10 ns, rel.):" % wg, " ".join("%d/%d/%d" % tuple(wb[wg, k, :3] - t0) for k in range(16)))
    except AttributeError:
        pass
