cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for b in 3 5 7 9 11 13; do
  J2K_BAND_PROWS=$b rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/band_$b -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
done
