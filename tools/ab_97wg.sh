# A/B of the lossy level-0 forward workgroup kernel's waves per workgroup (J2K_L0_WG97) through rocprofv3 kernel stats of tools/bench_c3.py (run on the GPU box)
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for wg in ${@:-8 10 12 14 16}; do
  J2K_L0_WG97=$wg rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/fwd97_$wg -- python $R/tools/bench_c3.py 0 0 > $R/gpurun_out/fwd97_$wg.log 2>&1
  python $R/tools/kstats.py dwt97_fwd_rgb $R/gpurun_out/fwd97_$wg
done
