#!/bin/bash
# Build a variant of libistvt_hip.so for a same-box A/B: one source recompiled with extra flags, the other objects
# reused from the normal build.   tools/build_variant.sh <out.so> <source.hip> <flags...>
# (use with ISTVT_LIB=<out.so> python bench.py, or GB_LIB=<out.so> python tools/gemm_bench.py)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CSRC=$ROOT/2023-tifs-istvt_amd/csrc
out=$1; src=$2; shift 2
mkdir -p "$(dirname "$out")"
tmp=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result -Wno-inline-asm -ffp-contract=fast "$@" -c "$CSRC/$src" -o "$tmp/v.o"
objs=""
for o in "$CSRC"/build/*.o; do
    [ "$(basename "$o")" = "${src%.hip}.o" ] || objs="$objs $o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$out" $objs "$tmp/v.o"
rm -rf "$tmp"
echo "$out"
