#!/bin/bash
# usage: tools/prof_pmc.sh <outdir-under-gpurun_out> "<counters>" -- <program args...>
# Runs rocprofv3 with PMC counters only (kernel-trace, no other trace domains) and prints per-kernel sums.
set -e
out="gpurun_out/$1"; shift
ctrs="$1"; shift; shift
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
mkdir -p "$out"
rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d "$out" -- "$@" > "$out/stdout.log" 2>&1 || { tail -20 "$out/stdout.log"; exit 1; }
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
files = glob.glob(out + "/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in files:
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"][:60]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
        n[(k, row["Counter_Name"])] += 1
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:32s} total {v:16.0f}   per-dispatch {v / n[(k, c)]:14.1f}  (n={n[(k, c)]})")
PY
