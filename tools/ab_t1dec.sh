# A/B: blocks per wavefront in the T1 decoder (C3 workload); -1 = one block per wavefront
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for K in ${KS:--1 1 2 4 7 8 12}; do
  echo "K=$K $(J2K_T1_DEC_LANES=$K timeout -k 10 120 python $R/tools/bench_c3.py 0 0 2>&1 | grep decode_blocks)"
done
