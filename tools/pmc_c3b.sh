# SQ instruction counters of tools/bench_c3.py kernels, totals per kernel name (run on the GPU box).  tools/pmc_c3b.sh <tag> <substr>
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $R/gpurun_out/${1}_pmc -- python $R/tools/bench_c3.py 0 0 > /dev/null 2>&1
python3 - $R/gpurun_out/${1}_pmc "$2" <<'PY'
import csv, glob, os, sys, collections
d, sub = sys.argv[1], sys.argv[2]
f = sorted(glob.glob(os.path.join(d, "*", "*counter_collection.csv")), key=os.path.getmtime)[-1]
acc = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    nm = r["Kernel_Name"].replace("void j2k::", "").split("(")[0]
    if sub not in nm: continue
    a = acc.setdefault((nm, r["Counter_Name"]), [0, 0.0]); a[0] += 1; a[1] += float(r["Counter_Value"])
for (nm, c), (n, v) in acc.items(): print("%-40s %-14s launches=%d total=%.3e per-launch=%.3e" % (nm[:40], c, n, v, v / n))
PY
