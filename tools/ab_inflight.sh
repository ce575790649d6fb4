# headline bench at 1..6 frames in flight (how far overlap of the latency-bound HT kernels carries)
cd $GRAFT_REPO_ROOT
for inf in 1 2 3 4 5 6; do
  python bench.py --steps 60 --warmup 10 --no-cpu-baseline --inflight $inf 2>/dev/null | python tools/benchline.py inflight $inf
done
