#!/bin/bash
# timing builds of the bitmap ranking kernel: tools/r2_probe.sh NAME [-D...]  -> usearch12_amd/variants/libugs_NAME.so (UGS_LIB=...)
set -e
cd "$(dirname "$0")/.."
tools/build_variant_of.sh ugs_rank2 "$@"
