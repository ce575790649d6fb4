# A/B: blocks per wavefront in the lane-parallel MQ kernel (C3 workload)
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for K in ${KS:-2 4 7 8 14 16 32 64}; do
  echo "K=$K $(J2K_T1_LANES=$K python $R/tools/bench_c3.py 0 0 2>&1 | grep encode_blocks)"
done
J2K_T1_LANES=7 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/t1lanes -- python $R/tools/bench_c3.py 0 0 > $R/gpurun_out/t1lanes.log 2>&1
python $R/tools/kstats.py $(ls -t $R/gpurun_out/t1lanes/*/*_kernel_stats.csv | head -1) | head -8
