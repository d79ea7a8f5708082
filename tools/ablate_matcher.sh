#!/bin/bash
# Developer tool: run tools/bench_matcher.py over tools/exp/lib_<name>.so variants (output: gpurun_out/ablate.txt)
# usage: tools/ablate_matcher.sh name1 name2 ...   (default: every variant present)
cd "$(dirname "$0")/.."
out=gpurun_out/ablate.txt; mkdir -p gpurun_out; : > $out
names="$@"; [ -z "$names" ] && names=$(ls tools/exp/lib_*.so | sed 's/.*lib_//; s/\.so//')
for n in $names; do
  for cfg in "--images 50 --kind vit" "--images 200 --kind vit" "--images 200 --kind scene"; do
    echo -n "$n | $cfg | " >> $out
    VITCOLMAP_HIP_LIB=$PWD/tools/exp/lib_$n.so timeout -k 10 120 python tools/bench_matcher.py $cfg --iters 20 > gpurun_out/bm.log 2>&1 || { cat gpurun_out/bm.log; exit 1; }
    if grep -q "Memory access fault" gpurun_out/bm.log; then echo "GPU FAULT ($n)"; exit 3; fi
    grep "pairs/s" gpurun_out/bm.log | sed 's/.*launch, //; s/GB\/s algorithmic //; s/ Tops int8//' >> $out
  done
done
cat $out
