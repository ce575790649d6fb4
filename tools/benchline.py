"""One-line digest of a bench.py JSON line on stdin (dev tool): python bench.py ... | python tools/benchline.py [label]"""
import json
import sys
d = json.loads(sys.stdin.readline())
r = d.get("roofline", {})
F = d.get("config", {}).get("frames_in_flight", 1) or 1
print("%s value %.0f %s  ms/step %.4f (host issue %.4f)  per frame %.1f us  kernel %.2f us frac %.3f" % (
    " ".join(sys.argv[1:]), d["value"], d["unit"], d["ms_per_step"], d.get("host_issue_ms_per_step", 0), d["ms_per_step"] * 1000 / F,
    r.get("avg_launch_us", 0), r.get("frac", 0)))
