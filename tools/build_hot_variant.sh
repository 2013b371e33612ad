#!/bin/bash
# tuning build of k_rank's HOT translation unit alone (compiles in under a minute): tools/build_hot_variant.sh NAME [-D.. | -mllvm ..] ...
# -> usearch12_amd/variants/libugs_NAME.so (the other objects are the in-tree ones; use with UGS_LIB=...; git-ignored)
# NOSCHED=1 drops the build's scheduler option for this unit.
set -e
cd "$(dirname "$0")/../usearch12_amd"
name=$1; shift
mkdir -p variants
sched="-mllvm -amdgpu-sched-strategy=iterative-maxocc"
[ -n "$NOSCHED" ] && sched=""
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -x hip -DUGS_RANK_TU=1 $sched "$@" -c csrc/ugs_rank.hip -o variants/hot_$name.o
objs=$(ls csrc/*.o | grep -v "ugs_rank_hot.o\|ugs_gather.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libugs_$name.so $objs variants/hot_$name.o
echo built variants/libugs_$name.so
