#!/usr/bin/env python3
"""Generate tests/golden/golden_v1.npz with the literal Python transliteration oracle/pyref.py
(the restatement that is independent of the C oracle).  Inputs are seeded; outputs are what the
reference's algorithms produce according to pyref.  Run in the build container:

    python tests/golden/make_golden.py

The fixtures pin (a) the C oracle on CPU (tests/test_oracle_golden.py) and (b) the HIP kernels on
the GPU box (tests/test_gpu_golden.py), where neither pyref's slowness nor /root/reference is
acceptable.  Input generators follow the reference's own tests where they exist
(i%256 ramps, (i*17)%512 with every 7th negated: internal/entropy/t1_test.go, coverage_test.go)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import pyref as p  # noqa: E402

tabs = p.load_ht_tables(os.path.join(HERE, "..", "..", "oracle", "ht_tables.h"))
rng = np.random.default_rng(20261003)
out = {}


def ref_pattern(n):
    i = np.arange(n, dtype=np.int64)
    v = (i * 17) % 512
    v[i % 7 == 0] *= -1
    return v.astype(np.int32)


# ---- DWT 5-3 / 9-7 multi-level (prefix layout) ----
dwt_cases = [(8, 8, 1), (16, 16, 3), (13, 7, 3), (1, 5, 2), (5, 1, 2), (2, 2, 1), (33, 20, 4), (64, 64, 5), (40, 24, 2)]
out["dwt_cases"] = np.array(dwt_cases, dtype=np.int32)
for i, (w, h, L) in enumerate(dwt_cases):
    x = rng.integers(-2000, 2000, w * h).astype(np.int32)
    d = [int(v) for v in x]
    p.decompose53(d, w, h, L)
    out["dwt53_in_%d" % i] = x
    out["dwt53_out_%d" % i] = np.array(d, dtype=np.int32)
    xf = rng.uniform(-500, 500, w * h)
    df = [float(v) for v in xf]
    p.decompose97(df, w, h, L)
    out["dwt97_in_%d" % i] = xf
    out["dwt97_out_%d" % i] = np.array(df, dtype=np.float64)
    rf = list(df)
    p.reconstruct97(rf, w, h, L)
    out["dwt97_rec_%d" % i] = np.array(rf, dtype=np.float64)

# ---- encoder.preprocess, lossless and lossy ----
pre_cases = [(16, 16, 3, 8, 3, 0), (20, 12, 3, 12, 4, 75), (8, 8, 1, 8, 0, 30), (24, 10, 4, 10, 3, 100)]
out["pre_cases"] = np.array(pre_cases, dtype=np.int32)
for i, (w, h, C, prec, nres, q) in enumerate(pre_cases):
    planes = rng.integers(0, 1 << prec, (C, h, w)).astype(np.int32)
    out["pre_in_%d" % i] = planes
    for name, lossless in (("ll", 1), ("ly", 0)):
        res = p.preprocess([[int(v) for v in planes[c].reshape(-1)] for c in range(C)], w, h, prec, lossless, nres, q)
        out["pre_%s_%d" % (name, i)] = np.array(res, dtype=np.int32).reshape(C, h, w)

# ---- MQ coder: raw (ctx, decision) sequences ----
for i, n in enumerate((1, 7, 100, 1000)):
    ctx = rng.integers(0, 19, n).astype(np.uint8)
    dec = (rng.random(n) < 0.3).astype(np.uint8)
    e = p.MQEncoder()
    for c, d in zip(ctx, dec):
        e.encode(int(c), int(d))
    out["mq_ctx_%d" % i] = ctx
    out["mq_dec_%d" % i] = dec
    out["mq_bytes_%d" % i] = np.frombuffer(e.flush(), dtype=np.uint8).copy()

# ---- T1: encoded bytes + numBPS, all four bands, edge shapes ----
t1_cases = []
for (w, h) in [(4, 4), (8, 8), (16, 16), (1, 1), (8, 1), (1, 8), (8, 5), (13, 9), (32, 32), (64, 64), (5, 3)]:
    for band in range(4):
        if (w, h) in ((32, 32), (64, 64)) and band not in (0, 3):
            continue
        t1_cases.append((w, h, band))
out["t1_cases"] = np.array(t1_cases, dtype=np.int32)
for i, (w, h, band) in enumerate(t1_cases):
    kind = i % 3
    if kind == 0:
        x = ref_pattern(w * h)
    elif kind == 1:
        x = rng.integers(-300, 301, w * h).astype(np.int32)
        x[rng.random(w * h) < 0.5] = 0
    else:
        x = -np.abs(rng.integers(1, 30000, w * h)).astype(np.int32)
    b, nb = p.t1_encode([int(v) for v in x], w, h, band)
    assert p.t1_decode(b, nb, band, w, h) == [int(v) for v in x]
    out["t1_in_%d" % i] = x
    out["t1_bytes_%d" % i] = np.frombuffer(b, dtype=np.uint8).copy()
    out["t1_nbps_%d" % i] = np.array([nb], dtype=np.int32)
    g = rng.integers(0, 256, max(3, len(b) // 2)).astype(np.uint8)       # decoder on non-encoder bytes
    out["t1_garbage_%d" % i] = g
    out["t1_garbage_dec_%d" % i] = np.array(p.t1_decode(bytes(g), max(nb, 1), band, w, h), dtype=np.int32)

# ---- HT: encoded bytes and what the reference decoder makes of them ----
ht_cases = [(4, 4), (8, 8), (16, 16), (64, 64), (32, 32), (8, 5), (13, 9), (7, 4), (5, 8), (12, 16), (64, 7), (3, 16)]
out["ht_cases"] = np.array(ht_cases, dtype=np.int32)
for i, (w, h) in enumerate(ht_cases):
    amp = (1, 3, 300, 4000)[i % 4]
    x = rng.integers(-amp, amp + 1, w * h).astype(np.int32)
    if i % 3 == 0:
        x[rng.random(w * h) < 0.6] = 0
    b = p.HTEncoder(w, h, tabs).encode([int(v) for v in x])
    out["ht_in_%d" % i] = x
    out["ht_bytes_%d" % i] = np.frombuffer(b, dtype=np.uint8).copy()
    out["ht_dec_%d" % i] = np.array(p.HTDecoder(w, h, tabs).decode(b), dtype=np.int32)
    g = rng.integers(0, 256, max(len(b), 8)).astype(np.uint8)
    if len(b) > 2:
        g[-2:] = np.frombuffer(b[-2:], dtype=np.uint8)
    out["ht_garbage_%d" % i] = g
    out["ht_garbage_dec_%d" % i] = np.array(p.HTDecoder(w, h, tabs).decode(bytes(g)), dtype=np.int32)

# ---- encodeTile job enumeration ----
enum_cases = [(3, 512, 512, 3, 256, 256), (3, 512, 512, 6, 64, 64), (1, 256, 112, 6, 64, 64), (3, 100, 37, 4, 32, 16), (2, 512, 112, 0, 64, 64)]
out["enum_cases"] = np.array(enum_cases, dtype=np.int32)
for i, a in enumerate(enum_cases):
    out["enum_%d" % i] = np.array(p.enumerate_blocks(*a), dtype=np.int32)

np.savez_compressed(os.path.join(HERE, "golden_v1.npz"), **out)
print("wrote golden_v1.npz with %d arrays" % len(out))
