#!/bin/bash
# Developer tool (GPU box): matcher parity tests, then the matcher micro-benchmarks; stops at the first failure or GPU fault.
set -eo pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests/test_matcher_gpu.py tests/test_bindings.py -x -q -m gpu 2>&1 | tee gpurun_out/matcher_tests.log | tail -15
if grep -q "Memory access fault" gpurun_out/matcher_tests.log; then echo "GPU FAULT in tests"; exit 3; fi
for cfg in "--images 50 --kind vit" "--images 200 --kind vit" "--images 50 --kind scene" "--images 200 --kind scene" "--images 50 --n 2048 --d 256 --kind vit --iters 5"; do
  timeout -k 10 120 python tools/bench_matcher.py $cfg > gpurun_out/bm.log 2>&1 || { cat gpurun_out/bm.log; exit 4; }
  if grep -q "Memory access fault" gpurun_out/bm.log; then cat gpurun_out/bm.log; echo "GPU FAULT in bench"; exit 3; fi
  grep -v "prepare\|amdgpu.ids" gpurun_out/bm.log
done
