"""Wall time of the one-call host forms on the C2 frame (3840x2160 RGBA8, 512x512 tiles, HT): j2k_encode_pixels_host from pageable
and from pinned host memory; closed-loop MQ encode + decode.   python tools/host_call_time.py"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "go-jpeg2000_amd"))
from j2kgfx import _lib                    # noqa: E402
from j2kgfx.codec import FramePlan         # noqa: E402
from j2kgfx.context import Context         # noqa: E402
ctx = Context(0)
W, H = 3840, 2160
rng = np.random.default_rng(1)
yy, xx = np.mgrid[0:H, 0:W]
frame = np.clip(np.stack([xx * 255 // W, yy * 255 // H, (xx + yy) * 127 // W]) + rng.integers(-16, 17, (3, H, W)), 0, 255).astype(np.uint8)
pix = np.full((H, W, 4), 255, np.uint8); pix[..., :3] = frame.transpose(1, 2, 0); pix = pix.reshape(H, W * 4)
pinned = torch.from_numpy(pix).pin_memory().numpy()
for name, coder, closed in (("HT, reference mode", 1, False), ("MQ, closed loop", 0, True)):
    plan = FramePlan(W, H, 3, precision=8, lossless=True, num_resolutions=6, cb=(64, 64), tile=(512, 512), coder=coder, ctx=ctx, closed_loop=closed)
    for label, src in (("pageable", pix), ("pinned", pinned)):
        got = plan.encode_pixels_host(_lib.PIX_RGBA8, src)
        t0 = time.perf_counter()
        for _ in range(5):
            got = plan.encode_pixels_host(_lib.PIX_RGBA8, src)
        dt = (time.perf_counter() - t0) / 5
        print("%s: j2k_encode_pixels_host from %s memory: %.2f ms per 4K frame = %.2f Gpixel/s (%d bytes out)" % (name, label, dt * 1e3, W * H / dt / 1e9, got["bytes"].size), flush=True)
    if closed:
        back = plan.decode_pixels_host(got["bytes"], (H, W * 4))
        t0 = time.perf_counter()
        for _ in range(3):
            back = plan.decode_pixels_host(got["bytes"], (H, W * 4))
        dt = (time.perf_counter() - t0) / 3
        assert np.array_equal(back.reshape(H, W, 4)[..., :3], pix.reshape(H, W, 4)[..., :3])
        print("%s: j2k_decode_pixels_host: %.2f ms per 4K frame = %.2f Gpixel/s, pixels equal" % (name, dt * 1e3, W * H / dt / 1e9), flush=True)
    plan.close()
