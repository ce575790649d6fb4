# A/B of the fused level-0+1 forward launch (J2K_L0_FUSE: 0 = separate level-1 launch, 8 / 16 = fused workgroups of that many waves)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for f in 0 8 16; do
  for inf in 1 3; do
    echo "== J2K_L0_FUSE=$f inflight=$inf"
    J2K_L0_FUSE=$f python bench.py --steps 100 --warmup 10 --no-cpu-baseline --inflight $inf 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('value %.0f Mpx/s  ms/step %.4f  level0 %.2f us (in timed region %.2f us) frac %.3f' % (d['value'], d['ms_per_step'], r['avg_launch_us'], r['avg_launch_us_in_timed_region'], r['frac']))"
  done
done
done
