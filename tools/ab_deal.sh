# A/B of J2K_L0_DEAL (short bands last per XCD in the RGBA8 level-0 job table) through bench.py (run on the GPU box)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for d in 0 1; do for f in 1 3; do
  J2K_L0_DEAL=$d python bench.py --steps 40 --warmup 5 --no-cpu-baseline --inflight $f 2>/dev/null | python tools/benchline.py deal=$d inflight=$f
done; done; done
