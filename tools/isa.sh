# dev: device ISA of one source -> /tmp/<name>.s and per-kernel instruction counts.   bash tools/isa.sh t1.hip [kernel-substr]
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -Igo-jpeg2000_amd/csrc -Iinclude -S --cuda-device-only go-jpeg2000_amd/csrc/$1 -o /tmp/${1%.*}.s 2>&1 | grep -i " error" -A3
python tools/isa_count.py /tmp/${1%.*}.s $2
