# frames in flight x GPU_MAX_HW_QUEUES (ROCm maps HIP streams onto that many hardware queues; default 4)
cd $GRAFT_REPO_ROOT
for q in 4 8; do for inf in 3 4 6; do
  GPU_MAX_HW_QUEUES=$q python bench.py --steps 100 --warmup 10 --no-cpu-baseline --inflight $inf 2>/dev/null | python tools/benchline.py hwq $q inflight $inf
done; done
