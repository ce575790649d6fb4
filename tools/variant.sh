# dev: build go-jpeg2000_amd/build/libj2kgfx_<tag>.so with one source recompiled under extra flags.   tools/variant.sh <tag> <src.hip> <flags...>
# (use with J2K_LIB=go-jpeg2000_amd/build/libj2kgfx_<tag>.so; build/ is git-ignored, the .so travels to the GPU box)
set -e
cd "$(dirname "$0")/../go-jpeg2000_amd"
tag=$1; src=$2; shift 2
make -s
/opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wno-unused-function "$@" -c csrc/$src -o build/var_$tag.o 2>/dev/null
objs=$(ls build/*.o | grep -v "build/var_" | grep -v "build/$src.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o build/libj2kgfx_$tag.so $objs build/var_$tag.o
echo built build/libj2kgfx_$tag.so
