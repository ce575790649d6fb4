"""After `gpurun -- bash tools/round_profiles.sh`: copy what is judged from gpurun_out/round/ into profiles/ (tracked).
python tools/collect_profiles.py <tag>   e.g. r02b -> profiles/<tag>_bench_default_3inflight_kernel_stats.csv, ..._inflight1_...,
..._inflight1_pmc_summary.txt (+ the per-dispatch rows of the transform kernels), <tag>_c3_kernel_stats.csv"""
import csv, glob, os, shutil, subprocess, sys
tag = sys.argv[1]
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = os.path.join(R, "gpurun_out", "round"); P = os.path.join(R, "profiles")
def newest(pat):
    return sorted(glob.glob(os.path.join(O, pat)), key=os.path.getmtime)[-1]
shutil.copy(newest("stats_default/*/*kernel_stats.csv"), os.path.join(P, tag + "_bench_default_3inflight_kernel_stats.csv"))
shutil.copy(newest("stats_inflight1/*/*kernel_stats.csv"), os.path.join(P, tag + "_bench_inflight1_kernel_stats.csv"))
shutil.copy(newest("stats_c3/*/*kernel_stats.csv"), os.path.join(P, tag + "_c3_kernel_stats.csv"))
lines = ["# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), python bench.py --steps 10 --warmup 3 --no-cpu-baseline --inflight 1; per-kernel average, KiB",
         "# (gfx950: FETCH_SIZE counts 64 B per 128-B request of a wide coalesced read -> double it; WRITE_SIZE is exact)"]
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    out = subprocess.run([sys.executable, os.path.join(R, "tools", "pmc_sum.py"), os.path.join(O, "pmc_" + c)], capture_output=True, text=True).stdout
    lines += [l for l in out.splitlines() if l.strip()]
    src = newest("pmc_%s/*/*counter_collection.csv" % c)
    rows = list(csv.reader(open(src)))
    keep = [rows[0]] + [r for r in rows[1:] if "dwt" in ",".join(r)]
    with open(os.path.join(P, "%s_bench_inflight1_pmc_%s_dwt.csv" % (tag, c)), "w", newline="") as f:
        csv.writer(f).writerows(keep)
open(os.path.join(P, tag + "_bench_inflight1_pmc_summary.txt"), "w").write("\n".join(lines) + "\n")
print(open(os.path.join(P, tag + "_bench_inflight1_pmc_summary.txt")).read())
if glob.glob(os.path.join(O, "stats_c3bench/*/*kernel_stats.csv")):
    shutil.copy(newest("stats_c3bench/*/*kernel_stats.csv"), os.path.join(P, tag + "_c3_bench_kernel_stats.csv"))
    lines = ["# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), python bench.py --config c3 --steps 2 --warmup 1 --no-cpu-baseline --inflight 1;",
             "# per-kernel average, KiB, transform kernels only (gfx950: double FETCH_SIZE for wide coalesced reads; WRITE_SIZE is exact)"]
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        out = subprocess.run([sys.executable, os.path.join(R, "tools", "pmc_sum.py"), os.path.join(O, "pmc_c3_" + c), "dwt97"], capture_output=True, text=True).stdout
        lines += [l for l in out.splitlines() if l.strip()]
    open(os.path.join(P, tag + "_c3_pmc_summary.txt"), "w").write("\n".join(lines) + "\n")
    print(open(os.path.join(P, tag + "_c3_pmc_summary.txt")).read())

if glob.glob(os.path.join(O, "stats_c3lanes/*/*kernel_stats.csv")):
    shutil.copy(newest("stats_c3lanes/*/*kernel_stats.csv"), os.path.join(P, tag + "_c3_lanes_kernel_stats.csv"))
    if os.path.exists(os.path.join(O, "c3lanes_per_launch.txt")):
        shutil.copy(os.path.join(O, "c3lanes_per_launch.txt"), os.path.join(P, tag + "_c3_lanes_per_launch.txt"))
for name in ("c1gpu", "c5"):
    if glob.glob(os.path.join(O, "stats_%s/*/*kernel_stats.csv" % name)):
        shutil.copy(newest("stats_%s/*/*kernel_stats.csv" % name), os.path.join(P, "%s_%s_bench_kernel_stats.csv" % (tag, name)))
if glob.glob(os.path.join(O, "stats_cl/*/*kernel_stats.csv")):
    shutil.copy(newest("stats_cl/*/*kernel_stats.csv"), os.path.join(P, tag + "_closed_loop_bench_kernel_stats.csv"))
    if glob.glob(os.path.join(O, "stats_clht/*/*kernel_stats.csv")):
        shutil.copy(newest("stats_clht/*/*kernel_stats.csv"), os.path.join(P, tag + "_closed_loop_ht_inflight1_kernel_stats.csv"))
if glob.glob(os.path.join(O, "pmc_invg16_FETCH/*/*counter_collection.csv")):
    lines = ["# inverse level 0 (dwt53_inv_rgba8_wg_kernel<4, 5, false>), one frame in flight: FETCH_SIZE (rocprofv3 --pmc, KiB per launch; x2 on gfx950) and",
             "# the kernel's average time (rocprofv3 --kernel-trace --stats) with 8 (default) and 16 consecutive bands per XCD (J2K_L0_XCD_GROUP):",
             "# algorithmic read = 3840 x 2160 x 12 B = 97200 KiB"]
    for g in (8, 16):
        out = subprocess.run([sys.executable, os.path.join(R, "tools", "pmc_sum.py"), os.path.join(O, "pmc_invg%d_FETCH" % g), "dwt53_inv_rgba8"], capture_output=True, text=True).stdout
        us = subprocess.run([sys.executable, os.path.join(R, "tools", "kstats.py"), "dwt53_inv_rgba8", os.path.join(O, "stats_invg%d" % g)], capture_output=True, text=True).stdout
        lines += ["group %2d: %s  |  %s" % (g, out.strip().replace("\n", " ; "), us.strip())]
    open(os.path.join(P, tag + "_inv_level0_xcd_group_ab.txt"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))
