#!/usr/bin/env bash
# Runs ON the GPU box: rocprofv3 kernel stats of the ViT-L/14 workload (BASELINE configs[4] on one GPU) -> gpurun_out/<tag>/
set -u
tag=${1:-ksl}; steps=${2:-5}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$tag
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -- python3 "$R/bench.py" --model ViT-L/14 --batch 512 --depth 12 --rank 8 --prompt-layers 12 --steps $steps --warmup 2 --no-cpu-baseline --no-roofline --no-extras > "$O/stats.out" 2>&1
f=$(find "$O/stats" -name "*kernel_stats.csv" | head -1)
cp "$f" "$O/kernel_stats.csv"
find "$O/stats" -name "*kernel_trace.csv" -delete; find "$O/stats" -name "*agent_info.csv" -delete
python3 - "$O/kernel_stats.csv" $((steps + 2)) <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2])
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total {tot / n / 1e6:.3f} ms per step over {n} steps (incl. set-up kernels)")
for r in rows[:16]:
    m = re.search(r'(\w+_kernel)(<[^>]*>)?', r["Name"]); s = (m.group(0) if m else r["Name"])[:70]
    print(f"{float(r['TotalDurationNs']) / n / 1e6:8.3f} ms  {int(r['Calls']) / n:7.2f}/step  {float(r['AverageNs']) / 1e3:8.1f} us  {s}")
PY
