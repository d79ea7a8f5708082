#!/bin/bash
# Developer tool: time the x-stationary GEMM experiment builds (tools/exp/lib_*.so) on the GPU box
for v in "$@"; do
  echo "== $v"
  VITCOLMAP_HIP_LIB=tools/exp/lib_$v.so XSONLY=1 python tools/bench_gemm.py 2>&1 | grep " xs "
done
