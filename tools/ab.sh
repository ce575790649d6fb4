# A/B on the GPU box: kernel stats of the default bench at one frame in flight under each setting of an environment knob.
# usage: bash tools/ab.sh <kernel-substr> VAR=val1,val2,... [VAR2=...]   (cartesian product; "-" = unset)
cd /tmp; export TMPDIR=/tmp; export J2K_TUNING=1   # (the library reads its J2K_* switches only with this set)
R=$GRAFT_REPO_ROOT
SUB=$1; shift
combos=("")
for spec in "$@"; do
  var=${spec%%=*}; vals=${spec#*=}
  new=()
  for c in "${combos[@]}"; do for v in ${vals//,/ }; do new+=("$c $var=$v"); done; done
  combos=("${new[@]}")
done
i=0
for c in "${combos[@]}"; do
  i=$((i+1)); O=$R/gpurun_out/ab/$i; rm -rf $O; mkdir -p $O
  envs=""; for kv in $c; do if [ "${kv#*=}" != "-" ]; then export $kv; envs="$envs $kv"; else unset ${kv%%=*}; fi; done
  rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-other-configs --inflight 1 > $O.log 2>&1
  echo "$c :: $(python $R/tools/kstats.py "$SUB" $O | cut -d' ' -f2- )"
  for kv in $c; do unset ${kv%%=*}; done
done
