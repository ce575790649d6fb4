# DEV (needs a library built with -DJ2K_DEV): marginal cost of each kernel with three frames in flight (J2K_DEV_SKIP leaves launches out; results are invalid by design)
cd $GRAFT_REPO_ROOT
for m in 0 1 2 4 8 16 32 64 128 0x100 0x200 0x400 0; do
  J2K_DEV_SKIP=$m python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python tools/benchline.py skip $m
done
