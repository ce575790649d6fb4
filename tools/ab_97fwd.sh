# A/B of library variants (tools/variant.sh) on the 9-7 kernels of tools/bench_c3.py under rocprofv3 (run on the GPU box).  tools/ab_97fwd.sh <tags...>  ("-" = the shipped library)
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for lib in "$@"; do
  if [ "$lib" != "-" ]; then export J2K_LIB=$R/go-jpeg2000_amd/build/libj2kgfx_$lib.so; else unset J2K_LIB; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/v97_$lib -- python $R/tools/bench_c3.py 0 0 > $R/gpurun_out/v97_$lib.log 2>&1
  python $R/tools/kstats.py rgb_wg_kernel $R/gpurun_out/v97_$lib
done
