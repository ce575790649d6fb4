# HBM traffic per kernel (run on the GPU box): bash tools/pmc_traffic.sh <tag> [bench args]; FETCH_SIZE / WRITE_SIZE in separate passes
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-pmc}; shift
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_$c -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --inflight 1 "$@" > $R/gpurun_out/${TAG}_$c.log 2>&1
  python $R/tools/pmc_sum.py $R/gpurun_out/${TAG}_$c
done
