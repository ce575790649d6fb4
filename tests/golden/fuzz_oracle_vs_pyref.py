#!/usr/bin/env python3
"""Differential fuzz: the C oracle (oracle/j2k_oracle.c, follows EncodeFast5 + LUT contexts) against the independent
literal Python restatement (oracle/pyref.py, follows EncodeSafe + rule contexts) on random blocks and planes.

    python tests/golden/fuzz_oracle_vs_pyref.py --n 12000 --procs 7 --seed 20261004 --out tests/golden/fuzz_r02_summary.json

Families (case i -> family i % 6, own seed = seed * 1000003 + i, so any case can be replayed with --replay i):
  t1      w, h in 1..64 (half uniform, half log-uniform), band 0..3, magnitudes up to 31 bits, several sparsity / sign mixes: encoded bytes,
          numBPS, both decoders on the encoder's bytes (== the input) and on arbitrary bytes
  ht      same block shapes, up to 31 bits: Go-panic domain (C status -2 <=> pyref GoPanic), bytes, both decoders on the
          encoder's bytes and on arbitrary bytes with a valid-looking SCUP
  dwt53   w, h in 1..64, 1..5 levels, full int32 range (Go wraparound): forward prefix layout + inverse
  dwt97   same shapes, f64 bit-identical forward and inverse
  pre     encoder.preprocess, lossless and lossy, 1..4 components, precision 1..16
  mq      raw (context, decision) sequences through MQEncoder / MQDecoder
Exit status 1 and the failing case indices on any mismatch.  The summary JSON (counts, digest of every compared output) is
committed; tests/test_oracle_fuzz.py replays a slice of it in the CPU suite.
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import oracle as orc  # noqa: E402
import pyref as p  # noqa: E402

TABS = None
FAMILIES = ("t1", "ht", "dwt53", "dwt97", "pre", "mq")


def tabs():
    global TABS
    if TABS is None:
        TABS = p.load_ht_tables(os.path.join(HERE, "..", "..", "oracle", "ht_tables.h"))
    return TABS


def dim(rng):
    if rng.random() < 0.5:
        return int(rng.integers(1, 65))                               # uniform 1..64
    return int(min(64, max(1, round(2.0 ** rng.uniform(0, 6.02)))))    # log-uniform: small and thin blocks


def block(rng, w, h):
    bits = int(rng.integers(1, 32))
    hi = (1 << bits) - 1
    x = rng.integers(-hi, hi + 1, w * h, dtype=np.int64)
    mode = int(rng.integers(0, 5))
    if mode == 0:
        x[rng.random(w * h) < rng.uniform(0.3, 0.98)] = 0            # sparse
    elif mode == 1:
        x = -np.abs(x)                                                # all negative
    elif mode == 2:
        x = (x >> int(rng.integers(0, bits))).astype(np.int64)        # mixed depths
    elif mode == 3:
        x[:] = 0
        k = int(rng.integers(0, 4))
        if k:
            x[rng.integers(0, w * h, k)] = rng.integers(-hi, hi + 1, k)   # a few isolated samples (run-length paths)
    return x.astype(np.int32)


def case(seed, i):
    """Runs case i; returns (family, digest bytes, error string or None, tag)."""
    tag = None
    rng = np.random.default_rng(seed * 1000003 + i)
    fam = FAMILIES[i % len(FAMILIES)]
    hsh = hashlib.sha256()
    err = None

    def same(name, a, b):
        nonlocal err
        a = np.asarray(a); b = np.asarray(b)
        hsh.update(np.ascontiguousarray(a).tobytes())
        if err is None and not (a.shape == b.shape and a.dtype == b.dtype and a.tobytes() == b.tobytes()):
            err = "%s differs" % name

    if fam == "t1":
        w, h, band = dim(rng), dim(rng), int(rng.integers(0, 4))
        x = block(rng, w, h)
        cb, cnb = orc.t1_encode(x, w, h, band)
        pb, pnb = p.t1_encode([int(v) for v in x], w, h, band)
        same("t1 bytes", cb, np.frombuffer(pb, np.uint8))
        same("t1 numBPS", np.int32(cnb), np.int32(pnb))
        cd = orc.t1_decode(cb, cnb, band, w, h).reshape(-1)
        same("t1 decode(encode) == input", cd, x)
        same("t1 decode C vs pyref", cd, np.array(p.t1_decode(pb, pnb, band, w, h), np.int32))
        g = rng.integers(0, 256, int(rng.integers(0, max(4, len(pb)))) , dtype=np.uint8)
        nb = int(rng.integers(0, 33))
        same("t1 decode of arbitrary bytes", orc.t1_decode(g, nb, band, w, h).reshape(-1),
             np.array(p.t1_decode(bytes(g), nb, band, w, h), np.int32))
    elif fam == "ht":
        w, h = dim(rng), dim(rng)
        x = block(rng, w, h)
        try:
            pb = p.HTEncoder(w, h, tabs()).encode([int(v) for v in x])
        except p.GoPanic:
            pb = None
        try:
            cb = orc.ht_encode(x, w, h)
        except ValueError:
            cb = None
        hsh.update(b"panic" if pb is None else b"ok")
        tag = "go_panic" if pb is None else None
        if (pb is None) != (cb is None):
            err = "HT Go-panic domain differs (pyref %s, C %s)" % (pb is None, cb is None)
        elif pb is not None:
            same("ht bytes", cb, np.frombuffer(pb, np.uint8))
            same("ht decode", orc.ht_decode(cb, w, h).reshape(-1), np.array(p.HTDecoder(w, h, tabs()).decode(pb), np.int32))
        n = int(rng.integers(2, 3 * max(w * h, 8)))
        g = rng.integers(0, 256, n, dtype=np.uint8)
        if rng.random() < 0.8:                                        # mostly with a SCUP the decoder accepts
            scup = int(rng.integers(2, min(n, 4095) + 1))
            g[-1] = scup & 0xFF; g[-2] = (g[-2] & 0xF0) | (scup >> 8)
        same("ht decode of arbitrary bytes", orc.ht_decode(g, w, h).reshape(-1), np.array(p.HTDecoder(w, h, tabs()).decode(bytes(g)), np.int32))
    elif fam == "dwt53":
        w, h, L = dim(rng), dim(rng), int(rng.integers(1, 6))
        bits = int(rng.integers(2, 33))
        x = rng.integers(-(1 << (bits - 1)), 1 << (bits - 1), w * h, dtype=np.int64).astype(np.int32)
        d = [int(v) for v in x]
        p.decompose53(d, w, h, L)
        c = orc.decompose53(x, w, h, L)
        same("decompose53", c.reshape(-1), np.array(d, np.int32))
        p.reconstruct53(d, w, h, L)
        same("reconstruct53", orc.reconstruct53(c, w, h, L).reshape(-1), np.array(d, np.int32))
    elif fam == "dwt97":
        w, h, L = dim(rng), dim(rng), int(rng.integers(1, 6))
        x = rng.uniform(-1, 1, w * h) * 2.0 ** rng.uniform(0, 24)
        d = [float(v) for v in x]
        p.decompose97(d, w, h, L)
        c = orc.decompose97(x, w, h, L)
        same("decompose97", c.reshape(-1), np.array(d, np.float64))
        p.reconstruct97(d, w, h, L)
        same("reconstruct97", orc.reconstruct97(c, w, h, L).reshape(-1), np.array(d, np.float64))
    elif fam == "pre":
        w, h, C = dim(rng), dim(rng), int(rng.integers(1, 5))
        prec, nres, q = int(rng.integers(1, 17)), int(rng.integers(0, 7)), int(rng.integers(0, 101))
        planes = rng.integers(0, 1 << prec, (C, h, w)).astype(np.int32)
        for lossless in (1, 0):
            want = p.preprocess([[int(v) for v in planes[c].reshape(-1)] for c in range(C)], w, h, prec, lossless, nres, q)
            got = orc.preprocess([planes[c] for c in range(C)], w, h, prec, bool(lossless), nres, q)
            same("preprocess lossless=%d" % lossless, np.stack([np.asarray(g).reshape(-1) for g in got]), np.array(want, np.int32))
    else:
        n = int(rng.integers(0, 3000))
        ctx = rng.integers(0, 19, n).astype(np.uint8)
        dec = (rng.random(n) < rng.uniform(0.02, 0.98)).astype(np.uint8)
        e = p.MQEncoder()
        for c_, d_ in zip(ctx, dec):
            e.encode(int(c_), int(d_))
        pb = e.flush()
        cb = orc.mq_encode(ctx, dec)
        same("mq bytes", cb, np.frombuffer(pb, np.uint8))
        same("mq decode", orc.mq_decode(cb, ctx), dec)
        g = rng.integers(0, 256, int(rng.integers(0, 64)), dtype=np.uint8)
        d2 = p.MQDecoder(bytes(g))
        same("mq decode of arbitrary bytes", orc.mq_decode(g, ctx), np.array([d2.decode(int(c_)) for c_ in ctx], np.uint8))
    return fam, hsh.digest(), err, tag


def run_range(args):
    seed, lo, hi = args
    orc.lib()
    res = []
    for i in range(lo, hi):
        try:
            fam, dg, err, tag = case(seed, i)
        except Exception as e:  # noqa: BLE001 -- a crash in either restatement is a finding too
            fam, dg, err, tag = FAMILIES[i % len(FAMILIES)], b"", "exception %r" % (e,), None
        res.append((i, fam, dg, err, tag))
    return res


def fuzz(n, seed, procs=1, first=0):
    chunks = [(seed, first + a, first + min(a + 50, n)) for a in range(0, n, 50)]
    if procs > 1:
        import multiprocessing as mp
        with mp.get_context("fork").Pool(procs) as pool:
            parts = pool.map(run_range, chunks, chunksize=1)
    else:
        parts = [run_range(c) for c in chunks]
    rows = sorted(r for part in parts for r in part)
    total = hashlib.sha256()
    counts = {f: 0 for f in FAMILIES}
    bad = []
    panics = 0
    head = None
    for n_done, (i, fam, dg, err, tag) in enumerate(rows):
        if n_done == 300:
            head = total.hexdigest()
        total.update(dg)
        counts[fam] += 1
        panics += tag == "go_panic"
        if err:
            bad.append((i, fam, err))
    return {"cases": len(rows), "first": first, "seed": seed, "per_family": counts, "ht_cases_in_go_panic_domain": panics,
            "mismatches": len(bad), "failed": bad[:50], "digest": total.hexdigest(), "digest_first_300": head or total.hexdigest()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=12000)
    ap.add_argument("--seed", type=int, default=20261004)
    ap.add_argument("--procs", type=int, default=max(1, (os.cpu_count() or 2) - 1))
    ap.add_argument("--out", default=None)
    ap.add_argument("--replay", type=int, default=None, help="run one case index and print its verdict")
    a = ap.parse_args()
    if a.replay is not None:
        print(case(a.seed, a.replay)[::2])
        return 0
    t0 = time.time()
    s = fuzz(a.n, a.seed, a.procs)
    s["seconds"] = round(time.time() - t0, 1)
    s["what"] = "C oracle (oracle/j2k_oracle.c) vs pyref (oracle/pyref.py), tests/golden/fuzz_oracle_vs_pyref.py"
    print(json.dumps(s, indent=1))
    if a.out:
        with open(a.out, "w") as f:
            json.dump(s, f, indent=1)
            f.write("\n")
    return 1 if s["mismatches"] else 0


if __name__ == "__main__":
    sys.exit(main())
