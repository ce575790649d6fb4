# DEV (library built with -DJ2K_DEV): what one MORE launch of each kernel costs a frame at three frames in flight.
# J2K_DEV_DUP issues the selected launches twice (every launch is idempotent, so results stay valid and bench.py's checks run).
cd $GRAFT_REPO_ROOT
export J2K_LIB=$PWD/go-jpeg2000_amd/build/libj2kgfx_dev.so
for m in 0 1 2 4 8 16 32 64 128 0x100 0x200 0x400 0; do
  J2K_DEV_DUP=$m python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python tools/benchline.py dup $m
done
