# do the coefficient planes (100 MB per frame slot) stay in the 256 MB Infinity Cache when the level-0 stores are plain?  (J2K_L0_STORE: 0 plain, 1 nt)
cd $GRAFT_REPO_ROOT
for st in 1 0; do for inf in 1 2 3; do
  J2K_L0_STORE=$st python bench.py --steps 100 --warmup 10 --no-cpu-baseline --inflight $inf 2>/dev/null | python tools/benchline.py store $st inflight $inf
done; done
