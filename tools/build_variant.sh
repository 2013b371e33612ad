#!/bin/bash
# tuning builds of the ranking kernel: tools/build_variant.sh NAME -DUGS_SCAN_V=1 -DUGS_RANK_WGS=4 ...
# -> usearch12_amd/variants/libugs_NAME.so (use with UGS_LIB=...; git-ignored, travels with gpurun)
set -e
cd "$(dirname "$0")/../usearch12_amd"
name=$1; shift
mkdir -p variants
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -x hip "$@" -c csrc/ugs_rank.hip -o variants/rank_$name.o
objs=$(ls csrc/*.o | grep -v "ugs_rank.o\|ugs_rank_hot.o\|ugs_gather.o")   # (one unit here: UGS_RANK_TU undefined)
hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libugs_$name.so $objs variants/rank_$name.o
echo built variants/libugs_$name.so
