cd /tmp; export TMPDIR=/tmp
for c in 8 4; do for b in 2 4 8; do J2K_CPL0=$c J2K_BAND_PROWS=$b rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/cpl_${c}_$b -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1; done; done
