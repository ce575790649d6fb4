# A/B: general T1 decode kernel (1) against the <= 64x64 kernel (0), C3 workload
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for K in ${KS:-1 0}; do
  echo "K=$K $(J2K_T1_DEC_GENERAL=$K timeout -k 10 120 python $R/tools/bench_c3.py 0 0 2>&1 | grep decode_blocks)"
done
