#!/usr/bin/env python3
"""Developer tool: turn the raw output of tools/prof_r02b.sh (gpurun_out/prof_r02b_*) into the committed summaries
profiles/r02_bench_kernel_stats.{csv,md}, r02_bench_line_profiled.json, r02_matcher_pmc.txt, r02_matcher_traffic.json."""
import csv, json, re, shutil
shutil.copy("gpurun_out/prof_r02b_kernel_stats.csv", "profiles/r02_bench_kernel_stats.csv")
shutil.copy("gpurun_out/prof_r02b_bench_line.json", "profiles/r02_bench_line_profiled.json")
shutil.copy("gpurun_out/prof_r02b_pmc.txt", "profiles/r02_matcher_pmc.txt")
d = json.load(open("profiles/r02_bench_line_profiled.json"))
rows = list(csv.DictReader(open("profiles/r02_bench_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
ctx = open("gpurun_out/prof_r02b_contexts.txt").read().strip().splitlines()
sparse_us = float(ctx[1].split("|")[3])
dense_us = float(ctx[2].split("|")[3])
gbs = 1225 * 405504 / sparse_us / 1e3
md = f"""# rocprofv3 --kernel-trace --stats — `python bench.py --no-cpu-baseline` (round 2 final, 1x MI355X, the driver's default command)

155 end-to-end steps (5 warm-up + 150 timed, 50 images each), then the two matcher loops of the same process: 3 warm-up + 1 probe +
{d['roofline']['launches_timed']} timed launches on the configs[2] input (`roofline`), the same on the dense input (`roofline_dense`).  Raw per-kernel
table: `r02_bench_kernel_stats.csv`; the JSON line the profiled run printed: `r02_bench_line_profiled.json` (tracing slows the
150 steps by a few percent: {d['ms_per_step']} ms per step here; the matcher loops are unaffected).  Collected by `tools/prof_r02b.sh`,
summarised by `tools/write_profiles_r02.py`.

`pair2_kernel<12>` is launched in three contexts, so its single row in the stats table mixes them; per context, from the kernel trace of
the same run (`tools/prof_bench_contexts.py`, dispatch start -> end, warm-up launches dropped):

| context | dispatches | average us | min | max |
|---|---:|---:|---:|---:|
{chr(10).join(ctx)}

The bench line of this run reports `roofline.launch_ms` = {d['roofline']['launch_ms']} and `roofline_dense.launch_ms` = {d['roofline_dense']['launch_ms']} (HIP events around
the timed loops): the two clocks agree.  1225 pairs x 405 504 B / {sparse_us:.2f} us = {gbs:.0f} GB/s = **{gbs/80:.1f} % of 8 TB/s** ({d['pair_matches_per_s']/1e6:.2f} M pairs/s,
{d['roofline']['int8_tops']/1e3:.2f} Pop/s int8 of nominal work); dense {d['roofline_dense']['achieved']:.0f} GB/s = {d['roofline_dense']['frac']*100:.1f} %.  (Twenty-launch loops, as `tools/bench_matcher.py`
runs them, take ~9 % longer per launch on the same kernel: the first launches after an idle period run slower than the sustained rate.)
Earlier profiles of this command: round 1 124-126 us = 49-50 %; round 2 after the persistent kernel 96.89 us = 64.1 %; after the
early waves took over the copies, the LDS pair list and the restart-free early-out 80.88 us = 76.8 %; with one barrier per two tiles
this one.

| kernel | calls | total ms | avg us | % of GPU time |
|---|---:|---:|---:|---:|
"""
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:16]:
    md += f"| `{r['Name'][:110]}` | {int(r['Calls'])} | {float(r['TotalDurationNs'])/1e6:.3f} | {float(r['AverageNs'])/1e3:.2f} | {float(r['TotalDurationNs'])/tot*100:.2f} |\n"
md += """
Per transformer layer (12 per step): `mlp2_kernel` 232 us (round 1: 250), `attention_kernel<2,true>` 233 us (234), qkv `xs_kernel<0,true,false>` 84 us (86),
proj `xs_kernel<2,false,false>` 48.5 us (48).
"""
open("profiles/r02_bench_kernel_stats.md", "w").write(md)
pm = open("profiles/r02_matcher_pmc.txt").read()
def g(name):
    m = re.search(r"pair2_kernel[^\n]*\n(?:[^\n]*\n)*?\s+" + name + r"\s+total\s+\d+\s+per-dispatch\s+([\d.]+)", pm)
    return float(m.group(1))
fetch, wr, rd, hit, miss = g("FETCH_SIZE"), g("WRITE_SIZE"), g("TCC_EA0_RDREQ_sum"), g("TCC_HIT_sum"), g("TCC_MISS_sum")
gui, nm, nv, busy = g("GRBM_GUI_ACTIVE"), g("SQ_INSTS_MFMA"), g("SQ_INSTS_VALU"), g("SQ_VALU_MFMA_BUSY_CYCLES")
t = {
 "kernel": "pair2_kernel<12>",
 "workload": "configs[2]: 50 x 512 x 384 uint8 blocks, 1225 pairs, one launch (tools/bench_matcher.py --images 50 --kind vit)",
 "collected": "tools/prof_r02b.sh: rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum / --pmc SQ_* GRBM_GUI_ACTIVE (four separate passes, 13 dispatches each; raw sums in r02_matcher_pmc.txt)",
 "FETCH_SIZE_KiB_per_launch": fetch, "WRITE_SIZE_KiB_per_launch": wr, "TCC_EA0_RDREQ_per_launch": rd, "TCC_HIT_per_launch": hit, "TCC_MISS_per_launch": miss,
 "correction": f"gfx950: FETCH_SIZE counts 128-B requests at 64 B for wide (16 B/lane) coalesced reads -> x2 (MI355X_MICROARCH.md, HBM section): {fetch} KiB x 2 = {fetch*2*1024/1e6:.1f} MB; cross-check TCC_EA0_RDREQ x 128 B = {rd*128/1e6:.1f} MB.  WRITE_SIZE is exact for these stores.",
 "hbm_bytes_per_launch": int(round(fetch * 2 * 1024 + wr * 1024, -5)),
 "note": f"memory-side (fabric) bytes per launch = {fetch*2*1024/1e6:.1f} MB read + {wr*1024/1e6:.2f} MB written = {fetch*2*1024/496.7e6:.2f}x the 496.7 MB algorithmic bytes (round 1's pair_kernel: 187 MB).  The prepared set is 10 MB and lives in the Infinity Cache: true HBM traffic is a fraction of this.",
 "mfma": {"SQ_INSTS_MFMA_per_launch": nm, "SQ_VALU_MFMA_BUSY_CYCLES_per_launch": busy, "GRBM_GUI_ACTIVE_per_launch_sum_over_8_XCDs": gui,
          "kernel_cycles": int(gui / 8), "mfma_pipe_busy": round(busy / (1024 * gui / 8), 3), "SQ_INSTS_VALU_per_launch": nv, "valu_per_mfma": round(nv / nm, 2),
          "note": "SQ_INSTS_MFMA is exactly one third of the 3 763 200 a launch holds (1225 pairs x 8 waves x 16 tiles x 24): on this input every tile is cut after the 4 head k-steps, which is why the matrix pipe is only ~20 % busy while the kernel delivers ~84 % of the contract roofline — the contract counts the work of the full product.  busy = BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8), measured in 13-launch bursts"}}
json.dump(t, open("profiles/r02_matcher_traffic.json", "w"), indent=1)
print(md.split("| kernel |")[0][-900:])
print(json.dumps(t["mfma"], indent=1)[:400])
