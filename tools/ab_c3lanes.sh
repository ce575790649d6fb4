# C3 throughput against MQ-kernel blocks per wavefront (J2K_T1_LANES) and frames in flight (run on the GPU box)
cd $GRAFT_REPO_ROOT
for lanes in ${LANES:-0 16 64}; do for f in ${FL:-2 4 6}; do
  J2K_T1_LANES=$lanes python bench.py --config c3 --steps 3 --warmup 1 --no-cpu-baseline --inflight $f 2>/dev/null | python tools/benchline.py lanes=$lanes inflight=$f
done; done
