#!/bin/bash
# Developer tool: build experimental variants of the kernel library (tools/exp/lib_<name>.so)
# usage: tools/build_variants.sh name1:"-DFLAG1 -DFLAG2" name2:"..."
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/exp
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -ffp-contract=off -shared $flags \
     vit_colmap_amd/csrc/capi.hip vit_colmap_amd/csrc/matcher.hip $(ls vit_colmap_amd/csrc/select.hip vit_colmap_amd/csrc/heatmap.hip vit_colmap_amd/csrc/preprocess.hip vit_colmap_amd/csrc/vit_ops.hip vit_colmap_amd/csrc/attention.hip vit_colmap_amd/csrc/gemm.hip 2>/dev/null) \
     -o tools/exp/lib_${name}.so &
done
wait
ls -la tools/exp/
