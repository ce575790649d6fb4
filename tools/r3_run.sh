# usage: bash tools/r3_run.sh <tag> "<pytest args or ->" [bench args...]   (GPU box): tests, then bench kernel stats at one frame in flight
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; shift
PT=$1; shift
O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
if [ "$PT" != "-" ]; then
  timeout -k 10 900 python -m pytest $PT -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -15 $O/pytest.log
  if [ $rc -ne 0 ]; then exit $rc; fi
fi
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_inflight1 -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --inflight 1 "$@" > $O/stats_inflight1.log 2>&1
tail -1 $O/stats_inflight1.log | cut -c1-300
python $R/tools/kstats.py "" $O/stats_inflight1 | tr '|' '\n' | grep -v "at::native\|rocclr"
