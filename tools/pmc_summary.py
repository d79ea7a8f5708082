#!/usr/bin/env python3
"""Developer tool: rocprofv3 counter_collection.csv of tools/vit_layer_pmc.py -> per-kernel MFMA pipe utilisation JSON
(profiles/r01_vit_mfma_utilisation.json) and a compact per-kernel CSV (profiles/r01_vit_pmc_sq.csv).

usage: tools/pmc_summary.py <dir with *counter_collection.csv> <out.json> <out.csv>"""
import collections, csv, glob, json, sys

NAMES = [  # substring of the kernel name -> label, in forward order
    ("gemm_kernel<3>", "patch+pos embedding"),
    ("xs_kernel<0, true", "LN1+qkv"),
    ("attention_kernel", "attention (q pre-scaled, deferred max)"),
    ("xs_kernel<2, false", "proj+residual"),
    ("mlp2_kernel", "fused MLP (LN2+fc1+GELU+fc2+residual)"),
    ("xs_kernel<1, true", "LN2+fc1+GELU"),
    ("gemm_kernel<2>", "fc2+residual"),
    ("add_layernorm_kernel", "final LayerNorm"),
]
PER_BLOCK = {"LN1+qkv", "attention (q pre-scaled, deferred max)", "proj+residual", "fused MLP (LN2+fc1+GELU+fc2+residual)",
             "LN2+fc1+GELU", "fc2+residual"}
SIMDS = 1024

def main(src, out_json, out_csv):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(collections.Counter)
    for f in glob.glob(src + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            label = next((lab for sub, lab in NAMES if sub in row["Kernel_Name"]), None)
            if label is None:
                continue
            agg[label][row["Counter_Name"]] += float(row["Counter_Value"])
            cnt[label][row["Counter_Name"]] += 1
    kernels, tot_busy, tot_cycles = {}, 0.0, 0.0
    for _, label in NAMES:
        if label not in agg:
            continue
        pd = {c: round(v / cnt[label][c], 1) for c, v in sorted(agg[label].items())}
        cyc = pd["GRBM_GUI_ACTIVE"] / 8.0
        k = {"dispatches": cnt[label]["GRBM_GUI_ACTIVE"], "per_dispatch": pd, "kernel_cycles": round(cyc),
             "mfma_pipe_busy": round(pd["SQ_VALU_MFMA_BUSY_CYCLES"] / (SIMDS * cyc), 4),
             "valu_issue_lower_bound": round(4 * pd["SQ_INSTS_VALU"] / (SIMDS * cyc), 4)}
        if "SQ_WAVE_CYCLES" in pd and pd["SQ_WAVE_CYCLES"] > 0:
            k["wave_parked_frac"] = round(pd.get("SQ_WAIT_ANY", 0.0) / pd["SQ_WAVE_CYCLES"], 3)
            k["issue_stall_frac"] = round(pd.get("SQ_WAIT_INST_ANY", 0.0) / pd["SQ_WAVE_CYCLES"], 3)
        kernels[label] = k
        if label in PER_BLOCK:
            tot_busy += pd["SQ_VALU_MFMA_BUSY_CYCLES"]
            tot_cycles += cyc
    doc = {"workload": "tools/vit_layer_pmc.py: ViT-S/14 bf16 forward, 50 images x 1530 patches (+cls), random weights",
           "collected": "rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE "
                        "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY (one pass, no other trace domain)",
           "definitions": "kernel cycles = GRBM_GUI_ACTIVE / 8 (summed over 8 XCDs); mfma_pipe_busy = SQ_VALU_MFMA_BUSY_CYCLES / "
                          "(1024 SIMDs x kernel cycles); valu_issue = 4 cycles x SQ_INSTS_VALU / (1024 x kernel cycles) "
                          "(transcendentals cost 8, so a lower bound)",
           "kernels": kernels,
           "transformer_block_mfma_pipe_busy": round(tot_busy / (SIMDS * tot_cycles), 4) if tot_cycles else None}
    json.dump(doc, open(out_json, "w"), indent=1)
    with open(out_csv, "w", newline="") as fh:
        w = csv.writer(fh)
        ctrs = sorted({c for k in kernels.values() for c in k["per_dispatch"]})
        w.writerow(["kernel", "dispatches"] + ctrs + ["kernel_cycles", "mfma_pipe_busy"])
        for lab, k in kernels.items():
            w.writerow([lab, k["dispatches"]] + [k["per_dispatch"].get(c, "") for c in ctrs] + [k["kernel_cycles"], k["mfma_pipe_busy"]])
    for lab, k in kernels.items():
        print(f"{lab:45s} cycles {k['kernel_cycles']:8d}  mfma busy {k['mfma_pipe_busy']:.3f}  valu>= {k['valu_issue_lower_bound']:.3f}")
    print("per transformer block:", doc["transformer_block_mfma_pipe_busy"])

if __name__ == "__main__":
    main(*sys.argv[1:4])
