"""The closed-loop frame by the ORACLE (test infrastructure): the reference's own functions composed per tile -- preprocess (DC shift, RCT, 5-3),
the job list with partitioning windows, the block coder, one packet per (component, resolution) through the restated PacketEncoder with the
closed-loop flags, createTileHeader.  Used by tests/test_gpu_decode_body.py, tests/test_closed_loop_golden.py and tests/golden/make_closed_loop_golden.py."""
import numpy as np


def frame(W, H, seed, noise=16):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:H, 0:W]
    f = np.stack([xx * 255 // W, yy * 255 // H, (xx + yy) * 127 // max(W, H)]) + rng.integers(-noise, noise + 1, (3, H, W))
    return np.clip(f, 0, 255).astype(np.uint8)


def oracle_frame(frm, W, H, tw, th, nres, cb, coder, sop, eph, orc, t2ref, tiles=None):
    """Returns per tile: dict(coeff, bytes, lens, numbps, part, w, h, x0, y0)."""
    out = {}
    tx_n, ty_n = (W + tw - 1) // tw, (H + th - 1) // th
    for t in range(tx_n * ty_n):
        if tiles is not None and t not in tiles:
            continue
        tx, ty = t % tx_n, t // tx_n
        x0, y0 = tx * tw, ty * th
        w, h = min(tw, W - x0), min(th, H - y0)
        sub = [np.ascontiguousarray(frm[c, y0:y0 + h, x0:x0 + w]).astype(np.int32) for c in range(frm.shape[0])]
        coeff = orc.preprocess(sub, w, h, 8, True, nres)
        by, lens, nb = orc.encode_tile_blocks(coeff, w, h, nres, cb, cb, coder, windows=1)
        jobs = orc.enumerate_blocks(len(sub), w, h, nres, cb, cb, 1)
        enc = t2ref.PacketEncoder(len_bits=5)
        pos, j = 0, 0
        while j < len(jobs):
            k = j
            blocks = []
            while k < len(jobs) and jobs[k]["comp"] == jobs[j]["comp"] and jobs[k]["res"] == jobs[j]["res"]:
                ln, n_b = int(lens[k]), int(nb[k])
                blocks.append(t2ref.CodeBlock(bytes(by[pos:pos + ln]), 1 if ln == 0 else 0, max(31 - n_b, 0), 0 if n_b == 0 else (1 if coder == 1 else 3 * n_b - 2)))
                pos += ln
                k += 1
            enc.encode_packet(t2ref.Precinct([blocks]), 0, sop, eph)
            j = k
        out[t] = dict(coeff=coeff, bytes=by, lens=lens, numbps=nb, part=orc.create_tile_header(t, bytes(enc.out)), w=w, h=h, x0=x0, y0=y0)
    return out


# the frames whose closed-loop tile-parts are pinned by digest (tests/golden/closed_loop_v1.json)
GOLDEN_CASES = [
    dict(name="mq_ragged_sop_eph", W=97, H=70, tile=(32, 48), cb=8, nres=3, coder=0, sop=True, eph=True, seed=201, noise=20),
    dict(name="ht_ragged_sop_eph", W=97, H=70, tile=(32, 48), cb=8, nres=3, coder=1, sop=True, eph=True, seed=202, noise=6),
    dict(name="mq_one_tile_bare", W=64, H=64, tile=(64, 64), cb=16, nres=4, coder=0, sop=False, eph=False, seed=203, noise=40),
    dict(name="ht_flat_empty_packets", W=80, H=48, tile=(40, 48), cb=16, nres=3, coder=1, sop=True, eph=False, seed=204, noise=0),
]


def golden_stream(case, orc, t2ref):
    """the frame of a golden case and its tile-parts end to end, by the oracle"""
    frm = frame(case["W"], case["H"], case["seed"], noise=case["noise"])
    if case["noise"] == 0:
        frm = np.full_like(frm, 128)
        frm[:, case["H"] // 2, case["W"] // 3] = 255
    want = oracle_frame(frm, case["W"], case["H"], case["tile"][0], case["tile"][1], case["nres"], case["cb"], case["coder"], case["sop"], case["eph"], orc, t2ref)
    return frm, b"".join(want[t]["part"] for t in sorted(want))
