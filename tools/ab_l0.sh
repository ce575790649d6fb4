# A/B of the packed-RGBA8 level-0 forward kernel (J2K_L0_WG: 0 = round-1 kernel, 4 / 8 = workgroup form; J2K_L0_STORE) through bench.py
cd $GRAFT_REPO_ROOT
for cfg in "0 1" "8 0" "8 1" "4 1"; do
  set -- $cfg
  for inf in 1 3; do
    echo "== J2K_L0_WG=$1 J2K_L0_STORE=$2 inflight=$inf"
    J2K_L0_WG=$1 J2K_L0_STORE=$2 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --inflight $inf 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('value %.0f Mpx/s  ms/step %.4f  level0 %.2f us (in timed region %.2f us) frac %.3f' % (d['value'], d['ms_per_step'], r['avg_launch_us'], r['avg_launch_us_in_timed_region'], r['frac']))"
  done
done
