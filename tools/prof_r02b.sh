#!/bin/bash
# Developer tool (GPU box): the round-2 final profiles — kernel-trace stats of the default bench command, the per-context
# split of the pair kernel, then PMC passes on the matcher micro-bench (each counter group in its own pass; never combined
# with other trace domains).  Raw output under gpurun_out/prof_r02b/.
set -eo pipefail
cd "$(dirname "$0")/.."
bash tools/prof_stats.sh prof_r02b/stats -- python3 bench.py --no-cpu-baseline | tee gpurun_out/prof_r02b_stats.txt
cp "$(find gpurun_out/prof_r02b/stats -name '*kernel_stats.csv' | head -1)" gpurun_out/prof_r02b_kernel_stats.csv
grep '"metric"' gpurun_out/prof_r02b/stats/stdout.log | tail -1 > gpurun_out/prof_r02b_bench_line.json
python3 tools/prof_bench_contexts.py "$(find gpurun_out/prof_r02b/stats -name '*kernel_trace.csv' | head -1)" 155 | tee gpurun_out/prof_r02b_contexts.txt
rm -f $(find gpurun_out/prof_r02b/stats -name '*kernel_trace.csv')   # (tens of MB: not merged back)
: > gpurun_out/prof_r02b_pmc.txt
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"; do
  name=$(echo $grp | cut -d' ' -f1)
  echo "=== PMC $grp" | tee -a gpurun_out/prof_r02b_pmc.txt
  bash tools/prof_pmc.sh prof_r02b/pmc_$name "$grp" -- python3 tools/bench_matcher.py --images 50 --kind vit --iters 10 | grep -A12 "pair2_kernel" | tee -a gpurun_out/prof_r02b_pmc.txt
done
