import sys,json
for ln in sys.stdin:
    if ln.startswith("{"):
        d=json.loads(ln); print(d["value"], d["ms_per_step"])
