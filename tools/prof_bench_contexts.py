#!/usr/bin/env python3
"""Developer tool: split the pair-kernel dispatches of a `rocprofv3 --kernel-trace` run of bench.py into the three
contexts bench.py launches it in (inside the timed steps / the configs[2] loop / the dense loop) and print the per-context
average duration — the configs[2] loop is the launch `roofline.launch_ms` times with HIP events.
usage: tools/prof_bench_contexts.py <kernel_trace.csv> <steps+warmup> [kernel name substring] """
import csv, sys
sub = sys.argv[3] if len(sys.argv) > 3 else None   # e.g. "pair2_kernel<12>": leaves the 2048 x 256 legs (pair2_kernel<8>) out
rows = [r for r in csv.DictReader(open(sys.argv[1]))
        if (sub in r["Kernel_Name"] if sub else ("pair2_kernel" in r["Kernel_Name"] or "pair_kernel" in r["Kernel_Name"]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
n_step = int(sys.argv[2])
rest = d[n_step:]
half = len(rest) // 2          # the two loops run the same number of launches only approximately: split at the jump
jump = max(range(8, len(rest) - 8), key=lambda i: sum(rest[i:i + 8]) / 8 - sum(rest[i - 8:i]) / 8)
ctx = {"in the timed steps (ragged blocks, ~115 keypoints)": d[:n_step], "configs[2] loop (sparse)": rest[:jump], "dense loop": rest[jump:]}
for k, v in ctx.items():
    tail = v[4:] if len(v) > 8 else v     # drop the warm-up launches and the one-launch timing probe
    print(f"| {k} | {len(v)} | {sum(tail)/len(tail):.2f} | {min(tail):.2f} | {max(tail):.2f} |")
