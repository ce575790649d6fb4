# rocprofv3 kernel stats of the bench (run on the GPU box): bash tools/prof_bench.sh <tag> [bench args...]
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-prof}; shift
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline "$@" > $R/gpurun_out/$TAG.log 2>&1
tail -1 $R/gpurun_out/$TAG.log | cut -c1-400
python $R/tools/kstats.py "" $R/gpurun_out/$TAG | tr '|' '\n'
