# A/B of the lossy level-0 workgroup kernel's waves per workgroup (J2K_L0_WG97) through bench.py --config c3 (run on the GPU box)
cd $GRAFT_REPO_ROOT
for wg in 8 10 12 14 16; do
  J2K_L0_WG97=$wg python bench.py --config c3 --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | python tools/benchline.py wg97 $wg
done
