cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${1:-c3prof} -- python $R/tools/bench_c3.py 0 0 > $R/gpurun_out/${1:-c3prof}.log 2>&1
