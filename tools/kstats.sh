#!/usr/bin/env bash
# Runs ON the GPU box: rocprofv3 kernel stats of the default bench command -> gpurun_out/<tag>/kernel_stats.csv (+ a per-step summary)
# usage: gpurun -- 'bash tools/kstats.sh <tag> [steps]'
set -u
tag=${1:-ks}; steps=${2:-20}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$tag
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -- python3 "$R/bench.py" --steps $steps --warmup 3 --no-cpu-baseline --no-roofline --no-extras > "$O/stats.out" 2>&1
f=$(find "$O/stats" -name "*kernel_stats.csv" | head -1)
cp "$f" "$O/kernel_stats.csv"
find "$O/stats" -name "*kernel_trace.csv" -delete; find "$O/stats" -name "*agent_info.csv" -delete
python3 - "$O/kernel_stats.csv" $((steps + 3)) <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2])
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total {tot / n / 1e6:.3f} ms per step over {n} steps (incl. set-up kernels)")
other = 0.0
for r in rows:
    name = r["Name"]
    ms = float(r["TotalDurationNs"]) / n / 1e6
    calls = int(r["Calls"]) / n
    lpi = "anonymous namespace" in name or name.startswith("_ZN12_GLOBAL") or "lpi" in name
    if not lpi:
        other += ms
    if ms > 0.02 or not lpi:
        print(f"{ms:8.3f} ms  {calls:7.2f}/step  {float(r['AverageNs']) / 1e3:8.1f} us  {'   ' if lpi else 'EXT'} {name[:110]}")
print(f"non-LPI kernels: {other:.3f} ms per step")
PY
