cd /tmp; export TMPDIR=/tmp
for x in 0 1; do for b in 4 8; do J2K_XCD_MAP=$x J2K_BAND_PROWS=$b rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/xcd_${x}_$b -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1; done; done
