"""GPU fuzz, third part (dev tool, run on the GPU box): the CLOSED-LOOP mode (j2k_params.closed_loop; this library's, not the
reference's) -- random geometries (tiles down to one sample wide, odd sizes, 1..6 resolutions, 4..64 code-blocks), contents and
precisions, both coders, SOP / EPH on and off.  Every frame: tile-parts == the oracle's composition (preprocess, the job list with
partitioning windows, the block coder, t2ref.PacketEncoder(len_bits=5) per tile, createTileHeader); parse -> block decode ->
placement == orc.decode_tile_blocks; MQ frames come back bit-exact; tile-part positions given and found by walking the SOTs.
    python tools/fuzz_gpu_closed_loop.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "go-jpeg2000_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as orc                                     # noqa: E402
import t2ref                                             # noqa: E402
from j2kgfx import J2KError, _lib                        # noqa: E402
from j2kgfx.codec import FramePlan                       # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
t0 = time.time()
n = npanic = npar_tiles = nser_tiles = nframe_enc = 0
t_say = t0
while time.time() - t0 < budget:
    if time.time() - t_say > 60:
        t_say = time.time(); print("... %d frames, %.0f s" % (n, t_say - t0), flush=True)
    Cn = int(rng.choice([1, 3, 3, 4]))
    tw = int(rng.choice([1, 2, 3, 5, 16, 32, 64, 100, 128, 256]))
    th = int(rng.choice([1, 2, 7, 16, 33, 64, 128]))
    W = int(rng.choice([tw, tw + 1, 2 * tw, 2 * tw + 3, 3 * tw - 1, 37, 100, 264]))
    H = int(rng.choice([th, th + 3, 2 * th - 1, 5, 33, 64, 130]))
    if W < 1 or H < 1 or W * H * Cn > 120000 or ((W + tw - 1) // tw) * ((H + th - 1) // th) > 400:
        continue
    nres = int(rng.integers(1, 7))
    cb = int(rng.choice([4, 8, 16, 32, 64]))
    coder = int(rng.integers(0, 2))
    prec = int(rng.choice([8, 12, 16]))
    sop, eph = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    if rng.integers(0, 3) == 0:
        sop = eph = True                                        # (about half the frames: SOP + EPH, the packets of a tile parsed side by side)
    top = (1 << prec) - 1
    kind = int(rng.integers(0, 4))
    if kind == 0:
        frame = rng.integers(0, top + 1, (Cn, H, W))
    elif kind == 1:
        yy, xx = np.mgrid[0:H, 0:W]
        frame = np.clip(np.stack([(xx * top // max(W, 1) + yy + c * 5) for c in range(Cn)]) + rng.integers(-3, 4, (Cn, H, W)), 0, top)
    elif kind == 2:
        frame = (top // 3) + rng.integers(-1, 2, (Cn, H, W))
    else:
        frame = np.full((Cn, H, W), top // 2)                    # flat: all-zero bands, empty packets
        frame[:, rng.integers(0, H), rng.integers(0, W)] = top
    frame = frame.astype(np.int32)
    desc = (Cn, W, H, tw, th, nres, cb, coder, prec, kind, sop, eph)
    plan = FramePlan(W, H, Cn, precision=prec, lossless=True, num_resolutions=nres, cb=(cb, cb), tile=(tw, th), coder=coder, closed_loop=True)
    d = torch.from_numpy(frame).to(plan.device)
    coeff = plan.forward(d)
    try:
        stream, offs, lens, numbps = plan.encode_stream(coeff)
        plan.ctx.sync()
    except J2KError as e:
        assert e.status == -5 and coder == 1, desc               # the reference's HT encoder panics on this input
        npanic += 1
        plan.close()
        continue
    cs, toffs = plan.encode_tile_parts(stream, offs, lens, numbps, sop=sop, eph=eph)
    plan.frame_status()
    h_cs, h_t = cs.cpu().numpy(), toffs.cpu().numpy()
    total = int(h_t[-1])
    planes = plan.planes()
    tiles_x = (W + tw - 1) // tw
    ntiles = tiles_x * ((H + th - 1) // th)
    levels = nres - 1 if nres - 1 > 0 else 5
    nres_jobs = nres if nres > 0 else 6
    panics = False
    ref = []
    for tl in range(ntiles):
        x0, y0 = (tl % tiles_x) * tw, (tl // tiles_x) * th
        w, h = min(tw, W - x0), min(th, H - y0)
        crop = [np.ascontiguousarray(frame[c, y0:y0 + h, x0:x0 + w]) for c in range(Cn)]
        want_c = orc.preprocess(crop, w, h, prec, True, nres)
        try:
            by, ln, nb = orc.encode_tile_blocks(want_c, w, h, nres_jobs, cb, cb, coder, windows=1)
        except ValueError:
            panics = True
            break
        jobs = orc.enumerate_blocks(Cn, w, h, nres_jobs, cb, cb, 1)
        enc = t2ref.PacketEncoder(len_bits=5)
        pos, j = 0, 0
        while j < len(jobs):
            k, blocks = j, []
            while k < len(jobs) and jobs[k]["comp"] == jobs[j]["comp"] and jobs[k]["res"] == jobs[j]["res"]:
                l_, n_ = int(ln[k]), int(nb[k])
                blocks.append(t2ref.CodeBlock(bytes(by[pos:pos + l_]), 1 if l_ == 0 else 0, max(31 - n_, 0), 0 if n_ == 0 else (1 if coder == 1 else 3 * n_ - 2)))
                pos += l_
                k += 1
            enc.encode_packet(t2ref.Precinct([blocks]), 0, sop, eph)
            j = k
        part = orc.create_tile_header(tl, bytes(enc.out))
        assert bytes(h_cs[int(h_t[tl]):int(h_t[tl + 1])]) == part, ("tile-part", desc, tl)
        ref.append((x0, y0, w, h, by, ln, nb))
    assert not panics, ("oracle panics where the product did not", desc)
    if Cn == 3 and prec == 8:
        # the one-call frame encoder (blocks gathered from their coding slots straight into the tile-parts) == the stage calls
        pix = np.full((H, W, 4), 255, np.uint8)
        pix[..., :3] = frame.transpose(1, 2, 0)
        cs_f, toffs_f = plan.encode_frame_pixels(_lib.PIX_RGBA8, torch.from_numpy(pix.reshape(H, W * 4)).to(plan.device), sop=sop, eph=eph)
        plan.frame_status()
        assert np.array_equal(toffs_f.cpu().numpy(), h_t) and np.array_equal(cs_f.cpu().numpy()[:total], h_cs[:total]), ("frame encoder", desc)
        nframe_enc += 1
    for given in (True, False):
        plan.frame_parallel_tiles()
        offs2, lens2, nb2 = plan.decode_tile_parts(cs, total, tile_offs=toffs if given else None, sop=sop, eph=eph)
        decoded = plan.decode_blocks(cs, offs2, lens2, nb2)
        placed = plan.place_blocks(decoded)
        back = plan.inverse(placed)
        try:
            plan.frame_status()
        except J2KError as e:
            raise AssertionError(("decode status", e.status, desc, given, total, [int(x) for x in h_t[:6]]))
        par = plan.frame_parallel_tiles()
        npar_tiles += par
        nser_tiles += ntiles - par if (sop and eph) else 0
        assert par == 0 or (sop and eph), desc
        hp, hb = placed.cpu().numpy(), back.cpu().numpy().reshape(Cn, H, W)
        for tl, (x0, y0, w, h, by, ln, nb) in enumerate(ref):
            want_p = orc.decode_tile_blocks(by, ln, nb, Cn, w, h, nres_jobs, cb, cb, coder, 1)
            for c in range(Cn):
                row = planes[tl * Cn + c]
                got = hp[int(row[6]):int(row[6]) + w * h].reshape(h, w)
                assert np.array_equal(got, want_p[c]), ("placed", desc, tl, c, given)
        if coder == 0:
            assert np.array_equal(hb, frame), ("round trip", desc, given)
        if Cn == 3 and prec == 8 and given:
            # the one-call frame decoder (HT: coded rows straight into the planes' windows, planes zeroed once) == the stage calls, frame after frame
            want_px = plan.inverse_pixels(placed, torch.zeros(H, W * 4, dtype=torch.uint8, device=plan.device))
            got_px = torch.zeros_like(want_px)
            for _ in range(2):
                plan.decode_frame_pixels(cs, total, got_px, tile_offs=toffs, sop=sop, eph=eph)
                plan.frame_status()
                assert torch.equal(got_px, want_px), ("frame decoder", desc)
    plan.close()
    n += 1
print("closed-loop fuzz: %d frames clean (%d in the reference's HT panic domain; SOP + EPH frames: %d tiles parsed packet-parallel, %d fell back to the tile chain; %d frames also through the one-call frame encoder) in %.0f s, seed %d"
      % (n, npanic, npar_tiles, nser_tiles, nframe_enc, time.time() - t0, seed))
