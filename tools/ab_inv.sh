# dev A/B: inverse 5-3 linked bands inside the default bench (run on the GPU box)
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for link in 0 1; do for b in 3 4 5 6 8; do
  J2K_INV_LINK=$link J2K_BAND_PROWS_INV=$b rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/inv_${link}_$b -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
done; done
