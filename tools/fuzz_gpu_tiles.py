"""GPU fuzz, second part (dev tool, run on the GPU box): TILED frames, lossless and lossy (9-7 + quantisation), both block coders,
and the packed-pixel entry points -- every tile against the oracle's per-tile pipeline (tile t == the reference pipeline on the
cropped planes): coefficients, block bytes / lengths / bit-plane counts, decoded blocks, reconstructions.
    python tools/fuzz_gpu_tiles.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "go-jpeg2000_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as orc                                     # noqa: E402
from j2kgfx import J2KError                              # noqa: E402
from j2kgfx.codec import FramePlan                       # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
t0 = time.time()
n = npanic = npix = nfused = 0
t_say = t0
while time.time() - t0 < budget:
    if time.time() - t_say > 60:                          # (a run that prints nothing for minutes is taken to be hung)
        t_say = time.time(); print("... %d frames, %.0f s" % (n, t_say - t0), flush=True)
    Cn = int(rng.choice([1, 3, 3, 4]))
    tile = int(rng.choice([32, 64, 128, 256, 512]))
    W = int(rng.choice([tile, tile + 8, 2 * tile, 2 * tile + 24, 3 * tile - 16, 100, 264, 520, 1032]))
    H = int(rng.choice([tile, tile + 3, 2 * tile - 5, 33, 64, 130]))
    if W * H * Cn > 500000 or W < 8 or H < 2:
        continue
    nres = int(rng.integers(2, 7))
    cb = int(rng.choice([16, 32, 64]))
    coder = int(rng.integers(0, 2))
    lossless = bool(rng.random() < 0.6)
    quality = 0 if lossless else int(rng.choice([1, 10, 75, 100, 400]))
    prec = int(rng.choice([8, 12, 16]))
    if coder == 0 and W * H * Cn > 100000:
        coder = 1
    top = (1 << prec) - 1
    kind = int(rng.integers(0, 3))
    if kind == 0:
        frame = rng.integers(0, top + 1, (Cn, H, W))
    elif kind == 1:
        yy, xx = np.mgrid[0:H, 0:W]
        frame = np.clip(np.stack([(xx * top // W + yy + c * 5) for c in range(Cn)]) + rng.integers(-3, 4, (Cn, H, W)), 0, top)
    else:
        frame = (top // 3) + rng.integers(-1, 2, (Cn, H, W))
    frame = frame.astype(np.int32)
    desc = (Cn, W, H, tile, nres, cb, coder, lossless, quality, prec, kind)
    plan = FramePlan(W, H, Cn, precision=prec, lossless=lossless, quality=quality, num_resolutions=nres, cb=(cb, cb), tile=(tile, tile), coder=coder)
    d = torch.from_numpy(frame).to(plan.device)
    coeff = plan.forward(d)
    back = plan.inverse(coeff)
    plan.ctx.sync()
    hco = coeff.cpu().numpy()
    planes, blocks, doffs = plan.planes(), plan.blocks(), plan.decoded_offsets()
    nblk = int(plan.info.blocks)
    tiles_x = (W + tile - 1) // tile
    ntiles = tiles_x * ((H + tile - 1) // tile)
    wants = []
    panics = False
    for tl in range(ntiles):
        x0, y0 = (tl % tiles_x) * tile, (tl // tiles_x) * tile
        w, h = min(tile, W - x0), min(tile, H - y0)
        crop = [np.ascontiguousarray(frame[c, y0:y0 + h, x0:x0 + w]) for c in range(Cn)]
        want_c = orc.preprocess(crop, w, h, prec, lossless, nres, quality)
        for c in range(Cn):
            row = planes[tl * Cn + c]
            got = hco[int(row[6]):int(row[6]) + w * h].reshape(h, w)
            assert np.array_equal(got, want_c[c]), ("coefficients", desc, tl, c)
        if not lossless:            # decode side of the lossy path: tcd ApplyInverseDWT rounding, InverseICT rounding, DC shift
            levels = nres - 1 if nres - 1 > 0 else 5
            inv = orc.postprocess([orc.tcd_inverse_dwt(want_c[c], w, h, levels, 0) for c in range(Cn)], prec, False)
            hb = back.cpu().numpy().reshape(Cn, H, W)
            for c in range(Cn):
                assert np.array_equal(hb[c, y0:y0 + h, x0:x0 + w], inv[c]), ("lossy inverse", desc, tl, c)
        try:
            wants.append(orc.encode_tile_blocks(want_c, w, h, nres, cb, cb, coder))
        except ValueError:
            panics = True
    if lossless:
        assert np.array_equal(back.cpu().numpy().reshape(Cn, H, W), frame), ("round trip", desc)
    if panics:
        try:
            plan.encode_stream(coeff); plan.ctx.sync()
            raise AssertionError(("no Go-panic status", desc))
        except J2KError:
            npanic += 1
    else:
        try:
            stream, offs, lens, nb = plan.encode_stream(coeff)
            dec = plan.decode_blocks(stream, offs, lens, nb)
            plan.ctx.sync()
        except Exception as exc:
            raise AssertionError(("encode / decode raised", desc, str(exc)))
        hl, hn, ho, hs, hd = lens.cpu().numpy()[:nblk], nb.cpu().numpy()[:nblk], offs.cpu().numpy(), stream.cpu().numpy(), dec.cpu().numpy()
        j0 = 0
        for tl, (data, wl, wn) in enumerate(wants):
            nj = len(wl)
            assert np.array_equal(hl[j0:j0 + nj], wl.astype(hl.dtype)), ("lens", desc, tl)
            assert np.array_equal(hn[j0:j0 + nj], wn), ("numbps", desc, tl)
            a, b = int(ho[j0]), int(ho[j0 + nj])
            assert np.array_equal(hs[a:b], data), ("bytes", desc, tl)
            for k in range(0, nj, 3):
                j = j0 + k
                bw, bh, band = int(blocks[j]["w"]), int(blocks[j]["h"]), int(blocks[j]["band"])
                p0 = int(ho[j]) - a
                chunk = data[p0:p0 + int(wl[k])]
                want_d = orc.ht_decode(chunk, bw, bh) if coder == 1 else orc.t1_decode(chunk, int(wn[k]), band, bw, bh)
                assert np.array_equal(hd[int(doffs[j]):int(doffs[j]) + bw * bh].reshape(bh, bw), want_d), ("decoded", desc, tl, k)
            j0 += nj
    # packed pixels in and out -- image.Gray / Gray16 / RGBA / RGBA64 / NRGBA / NRGBA64 by component count and precision -- against the
    # planar entry points and decoder.createImage (row strides with and without the 16-byte alignment the fused kernels need)
    if prec in (8, 16) and (lossless or (Cn in (1, 3) and prec == 8)):    # (lossy: image.RGBA / image.Gray at 8 bit, the reference's default path)
        fmt = {1: 0, 3: 2, 4: 4}[Cn] + (1 if prec == 16 else 0)
        ch, sb = (1 if Cn == 1 else 4), prec // 8
        samp = np.zeros((H, W, ch), np.int64)
        samp[..., :Cn] = frame.transpose(1, 2, 0)
        if Cn == 3:
            samp[..., 3] = rng.integers(0, top + 1, (H, W))                 # alpha of an RGBA source: ignored (encoder.go:108-138)
        row = (samp.astype(">u2").view(np.uint8) if sb == 2 else samp.astype(np.uint8)).reshape(H, W * ch * sb)
        stride = (row.shape[1] + 15) // 16 * 16 + int(rng.choice([0, 0, 16, 8, 3 if ch * sb == 1 else 24])) if rng.random() < 0.8 else row.shape[1]
        pix = rng.integers(0, 256, (H, stride)).astype(np.uint8)
        pix[:, :row.shape[1]] = row
        dp = torch.from_numpy(pix).to(plan.device)
        nfused += int(plan.pixels_fused(fmt, dp))
        c2 = plan.forward_pixels(fmt, dp)
        out = torch.full((H, stride), 0x5A, dtype=torch.uint8, device=plan.device)
        plan.inverse_pixels(c2, out)
        if Cn == 3 and prec == 8 and stride % 4 == 0:                        # the RGBA8 entry points of round 2
            c3 = plan.forward_rgba8(dp)
            plan.ctx.sync()
            assert torch.equal(c3, coeff), ("rgba8 forward", desc)
        plan.ctx.sync()
        assert torch.equal(c2, coeff), ("pixel forward", desc, fmt, stride)
        # decoder.createImage: alpha 255 / 65535 for three components, component 3 for four; 16 bit goes through the reference's
        # wrapping v * 65535 / 65535
        want_pix = orc.create_image([frame[c] for c in range(Cn)], prec) if lossless else orc.create_image([p_ for p_ in back.cpu().numpy().reshape(Cn, H, W)], prec)
        o = out.cpu().numpy()
        assert np.array_equal(o[:, :row.shape[1]], want_pix), ("pixel inverse", desc, fmt, stride)
        assert (o[:, row.shape[1]:] == 0x5A).all(), ("pixel inverse: row padding written", desc, fmt, stride)
        npix += 1
    plan.close()
    n += 1
print("fuzz (tiles) ok: %d frames (%d in the Go-panic domain, %d through the pixel entry points, %d of those read by the level-0 kernels themselves) in %.0f s (seed %d)" % (n, npanic, npix, nfused, time.time() - t0, seed))
