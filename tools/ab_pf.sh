# dev A/B: forward 5-3 level-0 variants inside the default bench (run on the GPU box)
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for pf in 0 1; do for c in 8 4; do for b in 2 4 8; do
  J2K_FWD_PF=$pf J2K_CPL0=$c J2K_BAND_PROWS=$b rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pf_${pf}_${c}_$b -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
done; done; done
for b in 2 4 8; do
  J2K_LIB=$R/go-jpeg2000_amd/build/libj2kgfx_w3.so J2K_FWD_PF=1 J2K_CPL0=8 J2K_BAND_PROWS=$b rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pf_w3_8_$b -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
done
cd $R
for d in gpurun_out/pf_*; do f=$(ls -t $d/*/*kernel_stats.csv | head -1); echo "$d $(grep dwt53_fwd_kernel $f | awk -F, '{printf "%s calls=%s avg=%s | ", substr($1,1,40), $2, $4}')"; done
