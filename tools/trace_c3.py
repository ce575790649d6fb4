"""dev tool: overlap of the MQ-coder kernels of several frames in flight.  python tools/trace_c3.py <rocprofv3 out dir> [frames]
Reads *_kernel_trace.csv (steady-state part): time with n kernels active, per kernel its average duration and its summed
duration per frame."""
import csv, glob, os, sys, collections
d = sys.argv[1]
f = sorted(glob.glob(os.path.join(d, "*", "*kernel_trace.csv")), key=os.path.getmtime)[-1]
rows = []
for r in csv.DictReader(open(f)):
    nm = r["Kernel_Name"].replace("void j2k::", "").replace("j2k::", "").split("(")[0].split("<")[0]
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), nm, r.get("Queue_Id", "")))
rows.sort()
j2k = [x for x in rows if x[2].startswith(("dwt", "t1_", "gather", "scan", "compact"))]
t_lo = j2k[int(len(j2k) * 0.4)][0]
j2k = [x for x in j2k if x[0] >= t_lo]
ev = []
for i, x in enumerate(j2k):
    ev.append((x[0], 1, i)); ev.append((x[1], -1, i))
ev.sort()
active = set(); last = None; conc = collections.Counter()
for t, kind, i in ev:
    if last is not None and active: conc[len(active)] += t - last
    if kind == 1: active.add(i)
    else: active.discard(i)
    last = t
cnt = collections.Counter(x[2] for x in j2k); dur = collections.Counter()
for x in j2k: dur[x[2]] += x[1] - x[0]
nfr = cnt["t1_mq_lanes_kernel"] or 1
span = j2k[-1][1] - j2k[0][0]
print("frames %d  span %.2f ms/frame  queues %s" % (nfr, span / nfr / 1e6, sorted(set(x[3] for x in j2k))))
print("time with n kernels active (ms/frame):", {k: round(v / nfr / 1e6, 2) for k, v in sorted(conc.items())})
print("%-34s %8s %10s %12s" % ("kernel", "calls/fr", "avg us", "sum ms/frame"))
for nm in sorted(dur, key=lambda n: -dur[n]):
    print("%-34s %8.1f %10.1f %12.2f" % (nm, cnt[nm] / nfr, dur[nm] / cnt[nm] / 1e3, dur[nm] / nfr / 1e6))
