"""GPU fuzz (dev tool, run on the GPU box): random frame geometries and contents through the plan API, every stage against the
oracle -- coefficients (encoder.preprocess), block bytes / lengths / bit-plane counts (encodeTile job order), decoded blocks
(HTDecoder.Decode / T1.Decode on the oracle's bytes) and the lossless reconstruction.  Untiled frames (the oracle's
preprocess works on whole components); tiles are covered by the sharded tests.
    python tools/fuzz_gpu.py [seconds] [seed]        (J2K_FUZZ_HT_SHARE=0: MQ coder only; with J2K_T1_DEC_SPLIT=1 the lanes decoder)"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "go-jpeg2000_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as orc                                     # noqa: E402
from j2kgfx import CODER_HT, CODER_MQ                    # noqa: E402
from j2kgfx.codec import FramePlan                       # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
t0 = time.time()
n = npanic = 0
t_say = t0
while time.time() - t0 < budget:
    if time.time() - t_say > 60:                          # (a run that prints nothing for minutes is taken to be hung)
        t_say = time.time(); print("... %d frames, %.0f s" % (n, t_say - t0), flush=True)
    Cn = int(rng.choice([1, 3]))
    W = int(rng.choice([16, 24, 40, 64, 100, 128, 200, 256, 264, 512, 520, 776, 1024, 1032]))
    H = int(rng.choice([2, 3, 5, 16, 33, 64, 75, 128, 200]))
    if W * H * Cn > 600000:
        continue
    nres = int(rng.integers(1, 7))
    cb = int(rng.choice([16, 32, 64]))
    coder = CODER_HT if rng.random() < float(os.environ.get("J2K_FUZZ_HT_SHARE", "0.75")) else CODER_MQ
    if coder == CODER_MQ and W * H * Cn > 120000:
        coder = CODER_HT
    prec = int(rng.choice([8, 10, 12, 16]))
    kind = int(rng.integers(0, 5))
    top = (1 << prec) - 1
    if kind == 0:
        frame = rng.integers(0, top + 1, (Cn, H, W))
    elif kind == 1:                                      # smooth + small noise: long zero / one runs in the block coders
        yy, xx = np.mgrid[0:H, 0:W]
        frame = np.clip(np.stack([(xx * top // max(W, 1) + c * 3) for c in range(Cn)]) + rng.integers(-2, 3, (Cn, H, W)), 0, top)
    elif kind == 2:
        frame = np.full((Cn, H, W), int(rng.integers(0, top + 1)))
    elif kind == 3:                                      # sparse spikes
        frame = np.zeros((Cn, H, W), np.int64)
        m = rng.random((Cn, H, W)) < 0.02
        frame[m] = rng.integers(0, top + 1, int(m.sum()))
    else:                                                # two-level texture: many -1 / +1 coefficients (0xFF-rich MagSgn)
        frame = (top // 2) + rng.integers(-1, 2, (Cn, H, W))
    frame = frame.astype(np.int32)
    desc = (Cn, W, H, nres, cb, coder, prec, kind)
    want = orc.preprocess([frame[c] for c in range(Cn)], W, H, prec, True, nres)
    try:
        wb, wl, wn = orc.encode_tile_blocks(want, W, H, nres, cb, cb, coder)
        panics = False
    except ValueError:                                   # the Go code panics on this input (a block overruns its own buffer)
        panics = True
    plan = FramePlan(W, H, Cn, precision=prec, lossless=True, num_resolutions=nres, cb=(cb, cb), coder=coder)
    d = torch.from_numpy(frame).to(plan.device)
    coeff = plan.forward(d)
    back = plan.inverse(coeff)
    plan.ctx.sync()
    if panics:
        from j2kgfx import J2KError
        try:
            plan.encode_stream(coeff)
            plan.ctx.sync()
            raise AssertionError(("no Go-panic status", desc))
        except J2KError:
            npanic += 1
        plan.close()
        n += 1
        continue
    try:
        stream, offs, lens, nb = plan.encode_stream(coeff)
        decoded = plan.decode_blocks(stream, offs, lens, nb)
        plan.ctx.sync()
    except Exception as exc:
        raise AssertionError(("encode / decode raised", desc, str(exc)))
    hc = coeff.cpu().numpy()
    for c in range(Cn):
        assert np.array_equal(hc[c * W * H:(c + 1) * W * H].reshape(H, W), want[c]), ("coefficients", desc)
    nblk = int(plan.info.blocks)
    tot = int(offs[nblk].item())
    assert np.array_equal(lens.cpu().numpy()[:nblk].astype(np.uint32), wl), ("lens", desc)
    assert np.array_equal(nb.cpu().numpy()[:nblk], wn), ("numbps", desc)
    assert bytes(stream.cpu().numpy()[:tot]) == bytes(wb), ("bytes", desc)
    blocks = plan.blocks(); doffs = plan.decoded_offsets(); dh = decoded.cpu().numpy()
    pos = 0
    for j in range(nblk):
        w_, h_, band = int(blocks[j]["w"]), int(blocks[j]["h"]), int(blocks[j]["band"])
        chunk = wb[pos:pos + int(wl[j])]; pos += int(wl[j])
        ref = orc.ht_decode(chunk, w_, h_) if coder == CODER_HT else orc.t1_decode(chunk, int(wn[j]), band, w_, h_)
        assert np.array_equal(dh[int(doffs[j]):int(doffs[j]) + w_ * h_].reshape(h_, w_), ref), ("decoded", desc, j)
    assert np.array_equal(back.cpu().numpy().reshape(Cn, H, W), frame), ("round trip", desc)
    plan.close()
    n += 1
print("fuzz ok: %d frames (%d in the Go-panic domain) in %.0f s (seed %d)" % (n, npanic, time.time() - t0, seed))
