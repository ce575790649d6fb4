# rocprofv3 kernel stats of another bench configuration (run on the GPU box): bash tools/prof_cfg.sh <tag> <bench args...>
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-prof}; shift
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG -- python $R/bench.py --no-cpu-baseline "$@" > $R/gpurun_out/$TAG.log 2>&1
tail -1 $R/gpurun_out/$TAG.log | cut -c1-300
python $R/tools/kstats.py "" $R/gpurun_out/$TAG | tr '|' '\n'
