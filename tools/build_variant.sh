#!/bin/bash
# A variant of libskm.so for timing studies: skm_gmm.hip recompiled with extra -D flags, the other objects reused.
#   tools/build_variant.sh <name> -DSKM_ABL_NOLSE ...   ->  tools/ab/libskm_<name>.so   (select with SKM_LIB=...)
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
CS="$ROOT/speaker-recognition_amd/csrc"
name="$1"; shift
mkdir -p "$ROOT/tools/ab"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I"$ROOT/include" "$@" -c "$CS/skm_gmm.hip" -o "$ROOT/tools/ab/skm_gmm_$name.o"
# every other object of the in-tree build is reused (the team / segment kernels live in their own translation units:
# a flag that changes THEM needs those recompiled too)
others=$(ls "$CS"/obj/*.o | grep -v "/skm_gmm.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC "$ROOT/tools/ab/skm_gmm_$name.o" $others -o "$ROOT/tools/ab/libskm_$name.so"
rm -f "$ROOT/tools/ab/skm_gmm_$name.o"
echo "built tools/ab/libskm_$name.so"
