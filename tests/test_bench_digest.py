"""CPU: bench.py's committed DECODED_SHA256 (what its HT block decoder must produce for frame 0 of the C2 workload) is what
the ORACLE produces: encoder.preprocess + encodeTile (HT) + HTDecoder.Decode of every block, job order, every tile."""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_bench_decoded_digest_matches_oracle(oracle):
    import bench
    import bench_extra
    cfg = bench_extra.CONFIGS["c2"]
    frame = bench_extra.synth_frame(np, cfg, 0)
    assert np.array_equal(frame, bench.synth_frame(np, 0))
    W, H, C, T = cfg["W"], cfg["H"], cfg["C"], cfg["tile"]
    h = hashlib.sha256()
    for y0 in range(0, H, T):
        for x0 in range(0, W, T):
            w, hh = min(T, W - x0), min(T, H - y0)
            crop = [np.ascontiguousarray(frame[c, y0:y0 + hh, x0:x0 + w]) for c in range(C)]
            coeff = oracle.preprocess(crop, w, hh, cfg["prec"], True, cfg["nres"])
            data, lens, _ = oracle.encode_tile_blocks(coeff, w, hh, cfg["nres"], cfg["cb"], cfg["cb"], 1)
            pos = 0
            for b, ln in zip(oracle.enumerate_blocks(C, w, hh, cfg["nres"], cfg["cb"], cfg["cb"]), lens):
                bw, bh = int(b["w"]), int(b["h"])
                dec = oracle.ht_decode(data[pos:pos + int(ln)], bw, bh).astype(np.int32).reshape(-1)
                pos += int(ln)
                h.update(dec.tobytes())
                pad = (-dec.size) % 4                                    # decoded blocks start at multiples of 4 elements
                if pad:
                    h.update(np.zeros(pad, np.int32).tobytes())
    assert h.hexdigest() == bench.DECODED_SHA256
