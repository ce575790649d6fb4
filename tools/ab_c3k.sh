# per-kernel averages of the 9-7 kernels in tools/bench_c3.py under rocprofv3 (run on the GPU box).  tools/ab_c3k.sh <tag> [substr]
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/c3k_$1 -- python $R/tools/bench_c3.py 0 0 > $R/gpurun_out/c3k_$1.log 2>&1
python $R/tools/kstats.py ${2:-dwt97} $R/gpurun_out/c3k_$1
