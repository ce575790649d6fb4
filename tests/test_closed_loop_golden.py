"""The closed-loop mode is this library's own stream format (no reference behaviour to be equal to): its tile-parts for
four small frames are pinned by digest in tests/golden/closed_loop_v1.json (written by tests/golden/make_closed_loop_golden.py from the
oracle's composition of the reference's functions).  CPU: the oracle still composes exactly those bytes, and the restated packet decoder reads
them back.  GPU: the product writes exactly those bytes -- stage calls and the one-call frame encoder -- and decodes them."""
import hashlib
import json
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (os.path.join(ROOT, "oracle"), os.path.join(ROOT, "go-jpeg2000_amd"), HERE):
    if p not in sys.path:
        sys.path.insert(0, p)
import closed_loop_ref as ref  # noqa: E402

GOLDEN = json.load(open(os.path.join(HERE, "golden", "closed_loop_v1.json")))


def test_golden_file_covers_the_cases():
    assert sorted(GOLDEN) == sorted(c["name"] for c in ref.GOLDEN_CASES)


@pytest.mark.parametrize("case", ref.GOLDEN_CASES, ids=[c["name"] for c in ref.GOLDEN_CASES])
def test_oracle_composes_the_pinned_tile_parts_and_reads_them_back(case):
    import oracle as orc
    import t2ref
    frm, stream = ref.golden_stream(case, orc, t2ref)
    g = GOLDEN[case["name"]]
    assert len(stream) == g["bytes"] and stream[:24].hex() == g["head"]
    assert hashlib.sha256(stream).hexdigest() == g["sha256"]
    # read back with the restated PacketDecoder (closed-loop flags): every tile-part's packets give the block lengths that went in
    tw, th = case["tile"]
    want = ref.oracle_frame(frm, case["W"], case["H"], tw, th, case["nres"], case["cb"], case["coder"], case["sop"], case["eph"], orc, t2ref)
    at = 0
    for t in sorted(want):
        part = want[t]["part"]
        assert stream[at:at + len(part)] == part
        assert part[:2] == b"\xff\x90" and part[12:14] == b"\xff\x93" and int.from_bytes(part[6:10], "big") == len(part)
        jobs = orc.enumerate_blocks(3, want[t]["w"], want[t]["h"], case["nres"], case["cb"], case["cb"], 1)
        dec = t2ref.PacketDecoder(part[14:], len_bits=5, seated=True)
        j, got = 0, []
        while j < len(jobs):
            k = j
            while k < len(jobs) and jobs[k]["comp"] == jobs[j]["comp"] and jobs[k]["res"] == jobs[j]["res"]:
                k += 1
            blocks = [t2ref.CodeBlock(None, 0, 0, 0) for _ in range(k - j)]
            dec.decode_packet(t2ref.Precinct([blocks]), 0, case["sop"], case["eph"])
            got += [b.dlen() for b in blocks]
            j = k
        assert got == [int(x) for x in want[t]["lens"]], t
        at += len(part)
    assert at == len(stream)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ref.GOLDEN_CASES, ids=[c["name"] for c in ref.GOLDEN_CASES])
def test_product_writes_the_pinned_tile_parts(case):
    import torch
    from j2kgfx import _lib
    from j2kgfx.codec import FramePlan
    from j2kgfx.context import Context
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    ctx = Context(0)
    frm = ref.frame(case["W"], case["H"], case["seed"], noise=case["noise"])
    if case["noise"] == 0:
        frm = np.full_like(frm, 128)
        frm[:, case["H"] // 2, case["W"] // 3] = 255
    W, H = case["W"], case["H"]
    plan = FramePlan(W, H, 3, precision=8, lossless=True, num_resolutions=case["nres"], cb=(case["cb"], case["cb"]), tile=case["tile"], coder=case["coder"],
                     ctx=ctx, closed_loop=True)
    g = GOLDEN[case["name"]]
    pix = np.full((H, W, 4), 255, np.uint8)
    pix[..., :3] = frm.transpose(1, 2, 0)
    d_pix = torch.from_numpy(pix.reshape(H, W * 4)).to(plan.device)
    # stage calls
    coeff = plan.forward(torch.from_numpy(frm.astype(np.int32)).to(plan.device))
    stream, offs, lens, numbps = plan.encode_stream(coeff)
    cs, toffs = plan.encode_tile_parts(stream, offs, lens, numbps, sop=case["sop"], eph=case["eph"])
    plan.frame_status()
    total = int(toffs[-1].item())
    assert total == g["bytes"]
    assert hashlib.sha256(cs[:total].cpu().numpy().tobytes()).hexdigest() == g["sha256"]
    # the one-call frame encoder
    cs2, toffs2 = plan.encode_frame_pixels(_lib.PIX_RGBA8, d_pix, sop=case["sop"], eph=case["eph"])
    plan.frame_status()
    assert int(toffs2[-1].item()) == total and hashlib.sha256(cs2[:total].cpu().numpy().tobytes()).hexdigest() == g["sha256"]
    # and back (MQ: to the source)
    back = torch.zeros_like(d_pix)
    plan.decode_frame_pixels(cs2, total, back, tile_offs=None, sop=case["sop"], eph=case["eph"])
    plan.frame_status()
    if case["coder"] == 0:
        assert torch.equal(back, d_pix)
    plan.close()
    ctx.close()
