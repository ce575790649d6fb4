# HBM-traffic pass (run on the GPU box): FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 passes, kernel-trace only.
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-pmc}
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_$c -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $R/gpurun_out/${TAG}_$c.log 2>&1
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_cal_$c -- $R/tools/probe/copy_bw > /dev/null 2>&1
done
ls $R/gpurun_out/${TAG}_FETCH_SIZE/*/ | head
