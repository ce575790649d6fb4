# A/B: variants of the lane-parallel MQ kernel built as alternative libraries (go-jpeg2000_amd/build/alt)
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for V in ${VS:-0 1 2}; do
  for K in ${KS:-4 7}; do
    echo "V=$V K=$K $(J2K_LIB=$R/go-jpeg2000_amd/build/alt/libj2kgfx_v$V.so J2K_T1_LANES=$K python $R/tools/bench_c3.py 0 0 2>&1 | grep encode_blocks)"
  done
done
