# round-3 baseline: GPU suite, bench default, kernel stats at one frame in flight
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3base; mkdir -p $O
cd $R
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" ; tail -3 $O/pytest.log
python bench.py > $O/bench.log 2>&1 ; tail -1 $O/bench.log | cut -c1-600
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_inflight1 -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --inflight 1 > $O/stats_inflight1.log 2>&1
python $R/tools/kstats.py "" $O/stats_inflight1 | tr '|' '\n'
