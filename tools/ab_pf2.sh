cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for pf in 0 1; do for b in 3 5; do
  J2K_BENCH_INFLIGHT=1 J2K_FWD_PF=$pf J2K_BAND_PROWS=$b rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ppf_${pf}_$b -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
done; done
