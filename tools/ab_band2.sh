cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for b in 3 4 5 6 7 9; do
  J2K_BENCH_INFLIGHT=1 J2K_BAND_PROWS=$b rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pb_$b -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
done
