# rocprofv3 kernel stats of the default bench (run on the GPU box): bash tools/prof_bench.sh <tag>
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-prof}
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline > $R/gpurun_out/$TAG.log 2>&1
tail -1 $R/gpurun_out/$TAG.log
