# dev A/B: forward 5-3 linked bands inside the default bench (run on the GPU box)
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for link in 0 1; do for pf in 0 1; do for b in 4 5 6 8 12; do
  J2K_FWD_LINK=$link J2K_FWD_PF=$pf J2K_BAND_PROWS=$b rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/link_${link}_${pf}_$b -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
done; done; done
