"""dev tool: per-kernel average of one rocprofv3 --pmc counter.  python tools/pmc_sum.py dir [substr]"""
import csv, glob, os, sys, collections
d = sys.argv[1]; sub = sys.argv[2] if len(sys.argv) > 2 else ""
f = sorted(glob.glob(os.path.join(d, "*", "*counter_collection.csv")), key=os.path.getmtime)[-1]
acc = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    nm = r["Kernel_Name"].replace("void j2k::", "").split("(")[0]
    if sub not in nm: continue
    k = (nm, r["Counter_Name"])
    a = acc.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += float(r["Counter_Value"])
for (nm, c), (n, v) in acc.items():
    print("%-60s %s n=%d avg=%.1f" % (nm[:60], c, n, v / n))
