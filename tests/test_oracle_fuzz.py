"""CPU: the differential fuzz C oracle vs pyref (tests/golden/fuzz_oracle_vs_pyref.py).  The full run (24 000 cases, all six
families, 0 mismatches) is committed as tests/golden/fuzz_r02_summary.json; here a slice of it is replayed (same seed: the
digest of the first 300 cases must match the committed one, i.e. both restatements still produce what they produced then)
and a fresh seed is tried."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))


def test_committed_fuzz_summary():
    s = json.load(open(os.path.join(HERE, "golden", "fuzz_r02_summary.json")))
    assert s["cases"] >= 10000 and s["mismatches"] == 0 and not s["failed"]
    assert all(v >= 1500 for v in s["per_family"].values())
    assert s["ht_cases_in_go_panic_domain"] > 0          # the HT panic domain was actually exercised


def test_fuzz_replay_and_fresh_seed(oracle):
    import fuzz_oracle_vs_pyref as fz
    s = json.load(open(os.path.join(HERE, "golden", "fuzz_r02_summary.json")))
    r = fz.fuzz(300, s["seed"])
    assert r["mismatches"] == 0, r["failed"]
    assert r["digest"] == s["digest_first_300"]
    r = fz.fuzz(300, 77)
    assert r["mismatches"] == 0, r["failed"]
