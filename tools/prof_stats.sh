#!/bin/bash
# usage: tools/prof_stats.sh <name> -- <program args...>   (kernel trace + stats only; no PMC)
set -e
name="$1"; shift; shift
out="gpurun_out/$name"
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -- "$@" > "$out/stdout.log" 2>&1 || { tail -30 "$out/stdout.log"; exit 1; }
f=$(find "$out" -name "*kernel_stats.csv" | head -1)
echo "== $f"
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"{'kernel':90s} {'calls':>7s} {'total ms':>10s} {'avg us':>10s} {'%':>6s}")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:28]:
    print(f"{r['Name'][:90]:90s} {int(r['Calls']):7d} {float(r['TotalDurationNs'])/1e6:10.3f} {float(r['AverageNs'])/1e3:10.2f} {float(r['TotalDurationNs'])/tot*100:6.2f}")
PY
