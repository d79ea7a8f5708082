#!/bin/bash
# Developer tool: run tools/bench_matcher.py over every tools/exp/lib_*.so variant (output: gpurun_out/ablate.txt)
cd "$(dirname "$0")/.."
out=gpurun_out/ablate.txt; : > $out
for lib in tools/exp/lib_*.so; do
  for cfg in "--images 50 --kind vit" "--images 200 --kind vit" "--images 200 --kind scene"; do
    echo "== $lib $cfg" >> $out
    VITCOLMAP_HIP_LIB=$PWD/$lib timeout -k 10 120 python tools/bench_matcher.py $cfg --iters 20 2>&1 | grep -v prepare >> $out || exit 1
  done
done
./tools/exp/mfma_peak >> $out 2>&1
