"""dev tool: per-launch durations of the plane-stepped decoder's kernels from a rocprofv3 kernel trace (last frame).
python tools/t1_trace.py <dir>"""
import csv, glob, os, sys
fs = sorted(glob.glob(os.path.join(sys.argv[1], "*", "*kernel_trace.csv")), key=os.path.getmtime)
rows = list(csv.DictReader(open(fs[-1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = ("t1_dec_sig_lanes", "t1_dec_plane", "t1_dec_magref_lanes", "t1_dec_step")
for nm in names:
    ds = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if nm in r["Kernel_Name"]]
    if not ds: continue
    per = 32 if nm != "t1_dec_magref_lanes" else 31
    last = ds[-per:]
    print(nm, "launches", len(ds), "last frame sum %.1f us:" % sum(last), " ".join("%.0f" % d for d in last))
