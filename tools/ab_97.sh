cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for b in 8 9 11 13 17; do
  J2K_BAND_PROWS_97=$b rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/b97_$b -- python $R/tools/bench_c3.py 0 0 > /dev/null 2>&1
done
