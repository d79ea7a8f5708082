#!/bin/bash
# Developer tool (GPU box): the round's profiles — kernel-trace stats of bench.py, then PMC passes on the matcher micro-bench
# (each counter group in its own pass; never combined with other trace domains).  Raw output under gpurun_out/prof_r02/.
set -eo pipefail
cd "$(dirname "$0")/.."
tools/prof_stats.sh prof_r02/stats -- python3 bench.py --steps 10 --warmup 3 --match-launches 13 --no-cpu-baseline | tee gpurun_out/prof_r02_stats.txt
cp "$(find gpurun_out/prof_r02/stats -name '*kernel_stats.csv' | head -1)" gpurun_out/prof_r02_kernel_stats.csv
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"; do
  name=$(echo $grp | cut -d' ' -f1)
  echo "=== PMC $grp"
  tools/prof_pmc.sh prof_r02/pmc_$name "$grp" -- python3 tools/bench_matcher.py --images 50 --kind vit --iters 10 | grep -A12 "pair2_kernel" | tee -a gpurun_out/prof_r02_pmc.txt
done
