# A/B of the lossy inverse level-0 workgroup kernel (J2K_L0_WG97_INV; optional J2K_LIB variants) through rocprofv3 kernel stats
# of tools/bench_c3.py (run on the GPU box).   tools/ab_97inv.sh "<libtags or ->" <wg...>
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
libs=${1:--}; shift
for lib in $libs; do
for wg in ${@:-8 6 10 12 0}; do
  if [ "$lib" != "-" ]; then export J2K_LIB=$R/go-jpeg2000_amd/build/libj2kgfx_$lib.so; else unset J2K_LIB; fi
  J2K_L0_WG97_INV=$wg rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/inv97_${lib}_$wg -- python $R/tools/bench_c3.py 0 0 > $R/gpurun_out/inv97_${lib}_$wg.log 2>&1
  python $R/tools/kstats.py dwt97_inv_rgb $R/gpurun_out/inv97_${lib}_$wg
done
done
