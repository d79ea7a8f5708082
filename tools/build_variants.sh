#!/bin/bash
# Developer tool: build experimental variants of the kernel library (tools/exp/lib_<name>.so)
# usage: tools/build_variants.sh name1:"-DFLAG1 -DFLAG2" name2:"..."
# Only matcher.hip is recompiled with the flags (every VC_EXP_* / VC2_* / VC_WAVES switch lives
# there); the other objects are the ones `make -C vit_colmap_amd/csrc` left behind.  FILE=gemm.hip (or
# attention.hip ...) selects another translation unit for the flags.
set -e
cd "$(dirname "$0")/.."
make -s -j8 -C vit_colmap_amd/csrc
mkdir -p tools/exp
FILE=${FILE:-matcher.hip}
OTHERS=$(ls vit_colmap_amd/csrc/*.o | grep -v "/${FILE%.hip}.o")
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  ( /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -ffp-contract=off $flags \
       -c vit_colmap_amd/csrc/$FILE -o tools/exp/${name}.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/exp/lib_${name}.so tools/exp/${name}.o $OTHERS ) &
done
wait
ls -la tools/exp/
