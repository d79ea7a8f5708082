#!/usr/bin/env python3
"""Developer tool: turn the raw output of tools/prof_r03.sh (gpurun_out/prof_r03_*) into the committed summaries
profiles/r03_bench_kernel_stats.{csv,md}, r03_bench_line_profiled.json, r03_matcher_pmc.txt, r03_matcher_traffic.json,
r03_vit_mfma_utilisation.json, r03_vit_pmc_sq.csv."""
import csv
import json
import re
import shutil

shutil.copy("gpurun_out/prof_r03_kernel_stats.csv", "profiles/r03_bench_kernel_stats.csv")
shutil.copy("gpurun_out/prof_r03_bench_line.json", "profiles/r03_bench_line_profiled.json")
shutil.copy("gpurun_out/prof_r03_pmc.txt", "profiles/r03_matcher_pmc.txt")
shutil.copy("gpurun_out/prof_r03_vit_mfma_utilisation.json", "profiles/r03_vit_mfma_utilisation.json")
shutil.copy("gpurun_out/prof_r03_vit_pmc_sq.csv", "profiles/r03_vit_pmc_sq.csv")
d = json.load(open("profiles/r03_bench_line_profiled.json"))
rows = list(csv.DictReader(open("profiles/r03_bench_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
ctx = open("gpurun_out/prof_r03_contexts.txt").read().strip().splitlines()
sparse_us = float(ctx[1].split("|")[3])
dense_us = float(ctx[2].split("|")[3])
gbs = 1225 * 405504 / sparse_us / 1e3
per_layer = {}
for r in rows:
    for key, lab in (("mlp2_kernel", "fused MLP"), ("attention_kernel<2, true", "attention"), ("xs_kernel<0, true", "qkv"), ("xs_kernel<2, false", "proj")):
        if key in r["Name"]:
            per_layer[lab] = float(r["AverageNs"]) / 1e3
md = f"""# rocprofv3 --kernel-trace --stats — `python bench.py --no-cpu-baseline --no-strong-anchor` (round 3, 1x MI355X)

155 end-to-end steps (5 warm-up + 150 timed, 50 images each; consecutive steps are pipelined, whole batches alternating between two HIP streams), then the
matcher loops of the same process on 512 x 384 blocks (3 warm-up + 1 probe + {d['roofline']['launches_timed']} timed launches on the configs[2] input —
`roofline` — and the same on the dense input — `roofline_dense`) and the two short 2048 x 256 legs (`matcher_c5_shape`,
`pair2_kernel<8>`).  Raw per-kernel table: `r03_bench_kernel_stats.csv`; the JSON line the profiled run printed:
`r03_bench_line_profiled.json` (tracing slows the 150 steps by a few percent: {d['ms_per_step']} ms per step here; with kernels of two
batches overlapping, a kernel's traced duration includes the time it shares the chip with the other batch's kernels, so the per-layer
averages below are longer than the single-stream figures of round 2 although the step is shorter).  Collected by `tools/prof_r03.sh`,
summarised by `tools/write_profiles_r03.py`.

`pair2_kernel<12>` is launched in three contexts; per context, from the kernel trace of the same run (`tools/prof_bench_contexts.py`):

| context | dispatches | average us | min | max |
|---|---:|---:|---:|---:|
{chr(10).join(ctx)}

The bench line of this run reports `roofline.launch_ms` = {d['roofline']['launch_ms']} and `roofline_dense.launch_ms` = {d['roofline_dense']['launch_ms']} (HIP events around
the timed loops): the two clocks agree.  1225 pairs x 405 504 B / {sparse_us:.2f} us = {gbs:.0f} GB/s = **{gbs/80:.1f} % of the 8 TB/s contract**
({d['pair_matches_per_s']/1e6:.2f} M pairs/s; nominal {d['roofline']['int8_tops_nominal']/1e3:.2f} Pop/s int8 of which the kernel EXECUTES one third on this input, see
`r03_matcher_traffic.json`); dense {d['roofline_dense']['achieved']:.0f} GB/s = {d['roofline_dense']['frac']*100:.1f} % ({d['roofline_dense']['int8_tops']/1e3:.2f} Pop/s int8, every MAC executed
= {d['roofline_dense']['int8_mfma_frac']*100:.1f} % of the 5 Pop/s dense int8 peak).

| kernel | calls | total ms | avg us | % of GPU time |
|---|---:|---:|---:|---:|
"""
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:18]:
    md += f"| `{r['Name'][:110]}` | {int(r['Calls'])} | {float(r['TotalDurationNs'])/1e6:.3f} | {float(r['AverageNs'])/1e3:.2f} | {float(r['TotalDurationNs'])/tot*100:.2f} |\n"
md += "\nPer launch (two launches per layer and step, one per batch shard of 25 images): " + ", ".join(f"{k} {v:.1f} us" for k, v in per_layer.items()) + ".\n"
open("profiles/r03_bench_kernel_stats.md", "w").write(md)

pm = open("profiles/r03_matcher_pmc.txt").read()
sparse_txt, dense_txt = pm.split("=== PMC dense", 1)


def g(txt, name):
    m = re.search(r"pair2_kernel[^\n]*\n(?:[^\n]*\n)*?\s+" + name + r"\s+total\s+\d+\s+per-dispatch\s+([\d.]+)", txt)
    return float(m.group(1))


fetch, wr, rd, hit, miss = (g(sparse_txt, n) for n in ("FETCH_SIZE", "WRITE_SIZE", "TCC_EA0_RDREQ_sum", "TCC_HIT_sum", "TCC_MISS_sum"))
gui, nm, nv, busy, wany, wcyc = (g(sparse_txt, n) for n in ("GRBM_GUI_ACTIVE", "SQ_INSTS_MFMA", "SQ_INSTS_VALU", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES"))
dgui, dnm, dnv, dbusy, dwany, dwcyc = (g("=== PMC dense" + dense_txt, n) for n in ("GRBM_GUI_ACTIVE", "SQ_INSTS_MFMA", "SQ_INSTS_VALU", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES"))
dfetch = g("=== PMC dense" + dense_txt.split("=== PMC dense FETCH_SIZE", 1)[1], "FETCH_SIZE")
NOMINAL = 1225 * 8 * 16 * 24
t = {
    "kernel": "pair2_kernel<12>",
    "workload": "configs[2]: 50 x 512 x 384 uint8 blocks, 1225 pairs, one launch (tools/bench_matcher.py --images 50 --kind vit); dense: --kind scene",
    "collected": "tools/prof_r03.sh: rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum / --pmc SQ_* GRBM_GUI_ACTIVE (separate passes, 13 dispatches each; raw sums in r03_matcher_pmc.txt)",
    "FETCH_SIZE_KiB_per_launch": fetch, "WRITE_SIZE_KiB_per_launch": wr, "TCC_EA0_RDREQ_per_launch": rd, "TCC_HIT_per_launch": hit, "TCC_MISS_per_launch": miss,
    "correction": f"gfx950: FETCH_SIZE counts 128-B requests at 64 B for wide (16 B/lane) coalesced reads -> x2 (MI355X_MICROARCH.md, HBM section): {fetch} KiB x 2 = {fetch*2*1024/1e6:.1f} MB; cross-check TCC_EA0_RDREQ x 128 B = {rd*128/1e6:.1f} MB.  WRITE_SIZE is exact for these stores.",
    "hbm_bytes_per_launch": int(round(fetch * 2 * 1024 + wr * 1024, -5)),
    "note": f"memory-side (fabric) bytes per launch = {fetch*2*1024/1e6:.1f} MB read + {wr*1024/1e6:.2f} MB written = {fetch*2*1024/496.7e6:.2f}x the 496.7 MB algorithmic bytes.  The prepared set is 10 MB and lives in the Infinity Cache: true HBM traffic is a fraction of this.",
    "mfma": {"SQ_INSTS_MFMA_per_launch": nm, "nominal_mfma_per_launch": NOMINAL, "executed_fraction_of_nominal": round(nm / NOMINAL, 4),
             "SQ_VALU_MFMA_BUSY_CYCLES_per_launch": busy, "GRBM_GUI_ACTIVE_per_launch_sum_over_8_XCDs": gui, "kernel_cycles": int(gui / 8),
             "mfma_pipe_busy": round(busy / (1024 * gui / 8), 3), "SQ_INSTS_VALU_per_launch": nv, "valu_per_mfma": round(nv / nm, 2),
             "wave_parked_frac": round(wany / wcyc, 3),
             "note": "nominal = 1225 pairs x 8 waves x 16 column tiles x 24 MFMAs (the full product).  On this input every tile is cut after the 4 head k-steps of 12, so a third of the nominal MACs execute: the contract roofline counts the work of the full product, the matrix pipe does a third of it."},
    "dense": {"SQ_INSTS_MFMA_per_launch": dnm, "executed_fraction_of_nominal": round(dnm / NOMINAL, 4), "kernel_cycles": int(dgui / 8),
              "mfma_pipe_busy": round(dbusy / (1024 * dgui / 8), 3), "valu_per_mfma": round(dnv / dnm, 2), "wave_parked_frac": round(dwany / dwcyc, 3),
              "FETCH_SIZE_KiB_per_launch": dfetch, "fabric_read_MB_per_launch": round(dfetch * 2 * 1024 / 1e6, 1)},
}
json.dump(t, open("profiles/r03_matcher_traffic.json", "w"), indent=1)
print(md.split("| kernel |")[0][-700:])
print(json.dumps({k: t[k] for k in ("mfma", "dense")}, indent=1)[:900])
v = json.load(open("profiles/r03_vit_mfma_utilisation.json"))
print("ViT block MFMA pipe busy:", v["transformer_block_mfma_pipe_busy"], {k: x["mfma_pipe_busy"] for k, x in v["kernels"].items()})
