"""Writes tests/golden/closed_loop_v1.json: SHA-256 / length / first bytes of the closed-loop tile-parts of the frames in
tests/closed_loop_ref.GOLDEN_CASES as the ORACLE composes them (oracle/j2k_oracle.c + oracle/t2ref.py).  The closed-loop mode is this library's own
stream format: the digests pin it across rounds -- a change of the format has to change this file on purpose.
    python tests/golden/make_closed_loop_golden.py"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
import oracle as orc          # noqa: E402
import t2ref                  # noqa: E402
import closed_loop_ref as ref  # noqa: E402

out = {}
for case in ref.GOLDEN_CASES:
    _, stream = ref.golden_stream(case, orc, t2ref)
    out[case["name"]] = {"bytes": len(stream), "sha256": hashlib.sha256(stream).hexdigest(), "head": stream[:24].hex()}
json.dump(out, open(os.path.join(HERE, "closed_loop_v1.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
