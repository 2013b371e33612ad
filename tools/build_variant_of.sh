#!/bin/bash
# tuning build of ONE kernel source: tools/build_variant_of.sh ugs_align NAME -mllvm -amdgpu-sched-strategy=iterative-maxocc ...
# -> usearch12_amd/variants/libugs_NAME.so (the other objects are the in-tree ones; use with UGS_LIB=...; git-ignored)
set -e
cd "$(dirname "$0")/../usearch12_amd"
src=$1; name=$2; shift; shift
mkdir -p variants
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -x hip "$@" -c csrc/$src.hip -o variants/${src}_$name.o
objs=$(ls csrc/*.o | grep -v "$src.o\|ugs_gather.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libugs_$name.so $objs variants/${src}_$name.o
echo built variants/libugs_$name.so
