# A/B of the fused level-0+1 forward launch at the default three frames in flight (J2K_L0_FUSE: 0 = separate level-1 launch, 8 / 16 = fused)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for f in 0 8 10; do
    J2K_L0_FUSE=$f python bench.py --steps 150 --warmup 10 --no-cpu-baseline 2>/dev/null | python tools/benchline.py fuse $f
done
done
