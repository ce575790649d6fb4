cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for rep in 1 2; do for b in 2 3 4 5; do
  J2K_BENCH_INFLIGHT=1 J2K_BAND_PROWS=$b rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pb3_${rep}_$b -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
done; done
