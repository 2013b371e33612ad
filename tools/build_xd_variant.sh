#!/bin/bash
# A/B builds of the gapped x-drop kernels: tools/build_xd_variant.sh NAME [SRCDIR] -DUGS_XD_SYNC=0 ...
# -> usearch12_amd/variants/libugs_NAME.so (use with UGS_LIB=...; git-ignored, travels with gpurun).
# SRCDIR (optional, a directory holding other versions of ugs_xdrop.hip / ugs_local.hip / ugs_xdrop_dev.h) lets an
# earlier revision be measured beside the current one: git show REV:path > SRCDIR/...
set -e
cd "$(dirname "$0")/../usearch12_amd"
name=$1; shift
src=csrc
if [ -d "$1" ]; then src=$1; shift; fi
mkdir -p variants
for f in ugs_xdrop ugs_local; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -x hip -I csrc "$@" -c $src/$f.hip -o variants/${f}_$name.o
done
objs=$(ls csrc/*.o | grep -v "ugs_xdrop.o\|ugs_local.o\|ugs_gather.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libugs_$name.so $objs variants/ugs_xdrop_$name.o variants/ugs_local_$name.o
echo built variants/libugs_$name.so
