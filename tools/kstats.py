"""dev tool: print per-kernel rows of rocprofv3 kernel_stats.csv files.  python tools/kstats.py <substr> dir..."""
import csv, glob, os, sys
sub = sys.argv[1]
for d in sys.argv[2:]:
    fs = sorted(glob.glob(os.path.join(d, "*", "*kernel_stats.csv")), key=os.path.getmtime)
    if not fs: print(d, "no stats"); continue
    out = []
    for r in csv.DictReader(open(fs[-1])):
        if sub in r["Name"]:
            nm = r["Name"].replace("void j2k::", "").split("(")[0]
            out.append("%s n=%s avg=%.1fus" % (nm, r["Calls"], float(r["AverageNs"]) / 1e3))
    print(os.path.basename(d), " | ".join(out))
