# A/B of J2K_NT_MODE builds (libj2kgfx_nt{1,2,3}.so beside the default) through bench.py at 1 and 3 frames in flight
cd $GRAFT_REPO_ROOT
for lib in libj2kgfx.so libj2kgfx_nt1.so libj2kgfx_nt2.so libj2kgfx_nt3.so; do
  for inf in 1 3; do
    echo "== $lib inflight=$inf"
    J2K_LIB=$GRAFT_REPO_ROOT/go-jpeg2000_amd/$lib python bench.py --steps 100 --warmup 10 --no-cpu-baseline --inflight $inf | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('value %.0f Mpx/s  ms/step %.4f  level0 %.2f us (in timed region %.2f us) frac %.3f' % (d['value'], d['ms_per_step'], r['avg_launch_us'], r['avg_launch_us_in_timed_region'], r['frac']))"
  done
done
