"""dev tool: how the kernels of several frames in flight overlap.  python tools/trace_overlap.py <rocprofv3 out dir>
Reads *_kernel_trace.csv: per kernel its average duration in the trace, the time it ran ALONE on the device and the time it
shared with others; plus the union busy time per frame."""
import csv, glob, os, sys, collections
d = sys.argv[1]
f = sorted(glob.glob(os.path.join(d, "*", "*kernel_trace.csv")), key=os.path.getmtime)[-1]
rows = []
for r in csv.DictReader(open(f)):
    nm = r["Kernel_Name"].replace("void j2k::", "").replace("j2k::", "").split("(")[0].split("<")[0]
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), nm))
rows.sort()
# keep the steady-state part: the last 60 % of the j2k kernels
j2k = [x for x in rows if x[2].startswith(("dwt", "ht_", "gather", "scan"))]
t_lo = j2k[int(len(j2k) * 0.4)][0]
j2k = [x for x in j2k if x[0] >= t_lo]
ev = []
for i, (s, e, nm) in enumerate(j2k):
    ev.append((s, 1, i)); ev.append((e, -1, i))
ev.sort()
alone = collections.Counter(); shared = collections.Counter(); active = set(); last = None; busy = 0; conc_time = collections.Counter()
for t, kind, i in ev:
    if last is not None and active:
        dt = t - last
        busy += dt
        conc_time[len(active)] += dt
        for a in active:
            (alone if len(active) == 1 else shared)[j2k[a][2]] += dt
    if kind == 1: active.add(i)
    else: active.discard(i)
    last = t
cnt = collections.Counter(x[2] for x in j2k); dur = collections.Counter()
for s, e, nm in j2k: dur[nm] += e - s
nfr = cnt["ht_walk_kernel"]
span = j2k[-1][1] - j2k[0][0]
print("frames %d  span %.1f us/frame  device busy %.1f us/frame" % (nfr, span / nfr / 1e3, busy / nfr / 1e3))
print("concurrency (us/frame):", {k: round(v / nfr / 1e3, 1) for k, v in sorted(conc_time.items())})
print("%-34s %8s %8s %8s" % ("kernel", "avg us", "alone", "shared"))
for nm in sorted(dur, key=lambda n: -dur[n]):
    print("%-34s %8.1f %8.1f %8.1f" % (nm, dur[nm] / cnt[nm] / 1e3, alone[nm] / nfr / 1e3, shared[nm] / nfr / 1e3))
