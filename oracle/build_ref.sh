#!/bin/bash
# TEST INFRASTRUCTURE ONLY.
# Compiles the *unmodified* reference sources where they lie (/root/reference/src)
# into oracle/_ref/usearch12 with plain gcc/g++ (the reference's own Makefile is NOT
# run: it hard-codes ccache and -march=native).  Nothing is copied into the repo;
# only object files + the binary land under oracle/_ref/ (git-ignored, but they
# travel to the GPU box with the gpurun snapshot, where the binary serves as the
# "reference" CPU baseline and as a parity cross-check).
# Flags follow /root/reference/src/Makefile:10-19 except -march=native is replaced
# by -march=x86-64-v2 so the static binary runs on whatever host the GPU box has.
set -e
REF=${REF:-/root/reference/src}
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/_ref"
if [ ! -d "$REF" ]; then
  echo "build_ref: $REF not present (GPU box?) - keeping prebuilt $OUT/usearch12 if any"
  exit 0
fi
mkdir -p "$OUT/o"
CXXFLAGS="-DNDEBUG -pthread -O3 -ffast-math -march=x86-64-v2 --std=c++11 -w"
CFLAGS="-O3 -ffast-math -march=x86-64-v2 -w"
JOBS=${JOBS:-$(nproc)}
build_one() {
  src="$1"; base="$(basename "$src")"; obj="$OUT/o/${base%.*}.o"
  if [ "$obj" -nt "$src" ]; then return 0; fi
  case "$src" in
    *.cpp) g++ $CXXFLAGS -I"$REF" -c "$src" -o "$obj" ;;
    *.c)   gcc $CFLAGS   -I"$REF" -c "$src" -o "$obj" ;;
  esac
}
export -f build_one; export OUT REF CXXFLAGS CFLAGS
ls "$REF"/*.cpp "$REF"/*.c | xargs -P "$JOBS" -I{} bash -c 'build_one {}'
g++ -O3 -pthread -static -o "$OUT/usearch12" "$OUT"/o/*.o -lpthread
echo "build_ref: built $OUT/usearch12"
# x-drop known-answer driver: our own main() (oracle/ref_xdrop_main.cpp) over the reference's objects
g++ $CXXFLAGS -I"$REF" -c "$HERE/ref_xdrop_main.cpp" -o "$OUT/ref_xdrop_main.o"
g++ -O3 -pthread -static -o "$OUT/ref_xdrop" "$OUT/ref_xdrop_main.o" $(ls "$OUT"/o/*.o | grep -v '/usearch_main\.o$') -lpthread
echo "build_ref: built $OUT/ref_xdrop"
