#!/bin/bash
# Developer tool (GPU box): the round-3 profiles.
#   1. rocprofv3 --kernel-trace --stats of the driver's default bench command (kernel table, per-context split of the pair kernel)
#   2. PMC passes on the matcher micro-bench, sparse (configs[2] input) and dense, each counter group in its own pass
#   3. one SQ pass over the ViT-S forward (single stream, so that per-dispatch counters are per kernel)
# PMC is never combined with another trace domain.  Raw output under gpurun_out/prof_r03/.
set -eo pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/prof_r03
bash tools/prof_stats.sh prof_r03/stats -- python3 bench.py --no-cpu-baseline --no-strong-anchor | tee gpurun_out/prof_r03_stats.txt
cp "$(find gpurun_out/prof_r03/stats -name '*kernel_stats.csv' | head -1)" gpurun_out/prof_r03_kernel_stats.csv
grep '"metric"' gpurun_out/prof_r03/stats/stdout.log | tail -1 > gpurun_out/prof_r03_bench_line.json
python3 tools/prof_bench_contexts.py "$(find gpurun_out/prof_r03/stats -name '*kernel_trace.csv' | head -1)" 155 "pair2_kernel<12>" | tee gpurun_out/prof_r03_contexts.txt
rm -f $(find gpurun_out/prof_r03/stats -name '*kernel_trace.csv')   # (tens of MB: not merged back)
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"
: > gpurun_out/prof_r03_pmc.txt
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" "$SQ"; do
  name=$(echo $grp | cut -d' ' -f1)
  echo "=== PMC sparse $grp" | tee -a gpurun_out/prof_r03_pmc.txt
  bash tools/prof_pmc.sh prof_r03/pmc_$name "$grp" -- python3 tools/bench_matcher.py --images 50 --kind vit --iters 10 | grep -A12 "pair2_kernel" | tee -a gpurun_out/prof_r03_pmc.txt
done
echo "=== PMC dense $SQ" | tee -a gpurun_out/prof_r03_pmc.txt
bash tools/prof_pmc.sh prof_r03/pmc_dense "$SQ" -- python3 tools/bench_matcher.py --images 50 --kind scene --iters 10 | grep -A12 "pair2_kernel" | tee -a gpurun_out/prof_r03_pmc.txt
echo "=== PMC dense FETCH_SIZE" | tee -a gpurun_out/prof_r03_pmc.txt
bash tools/prof_pmc.sh prof_r03/pmc_dense_fetch "FETCH_SIZE" -- python3 tools/bench_matcher.py --images 50 --kind scene --iters 10 | grep -A4 "pair2_kernel" | tee -a gpurun_out/prof_r03_pmc.txt
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
mkdir -p gpurun_out/prof_r03/vit
VITCOLMAP_VIT_SHARDS=1 ITERS=3 rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY \
  --output-format csv -d gpurun_out/prof_r03/vit -- python3 tools/vit_layer_pmc.py > gpurun_out/prof_r03/vit/stdout.log 2>&1 || { tail -20 gpurun_out/prof_r03/vit/stdout.log; exit 1; }
python3 tools/pmc_summary.py gpurun_out/prof_r03/vit gpurun_out/prof_r03_vit_mfma_utilisation.json gpurun_out/prof_r03_vit_pmc_sq.csv
rm -rf gpurun_out/prof_r03/vit gpurun_out/prof_r03/pmc_* gpurun_out/prof_r03/stats   # raw traces stay on the box
echo "prof_r03 done"
