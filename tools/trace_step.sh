#!/usr/bin/env bash
# Runs ON the GPU box: steady-state kernel statistics + the launch sequence of one step (tools/steady_stats.py over a --kernel-trace run of the default workload)
# usage: gpurun -- 'bash tools/trace_step.sh <tag> [bench.py workload arguments]'   -> gpurun_out/<tag>/
set -u
tag=${1:-trace}; shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$tag
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --output-format csv -d "$O/trace" -- python3 "$R/bench.py" "$@" --steps 12 --warmup 3 --no-cpu-baseline --no-roofline --no-extras > "$O/bench.out" 2>&1
python3 "$R/tools/steady_stats.py" "$O/trace" "$O/ss" 5 | tee "$O/summary.txt"
rm -rf "$O/trace"
