cd $GRAFT_REPO_ROOT
for m in 0 1 2 3; do cp go-jpeg2000_amd/libj2kgfx_nt$m.so go-jpeg2000_amd/libj2kgfx.so; cd /tmp; export TMPDIR=/tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/nt_$m -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1; cd $GRAFT_REPO_ROOT; done
