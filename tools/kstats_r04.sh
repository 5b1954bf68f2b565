#!/usr/bin/env bash
# Runs ON the GPU box: rocprofv3 kernel trace of the default bench command, cut to a STEADY-STATE window of whole steps by tools/steady_stats.py
# -> gpurun_out/<tag>/{ss_kernel_stats.csv, ss_step_sequence.txt, summary.txt}.   usage: gpurun -- 'bash tools/kstats_r04.sh <tag> [steps] [extra bench args]'
set -u
tag=${1:-ks4}; steps=${2:-12}; shift 2 || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$tag
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --output-format csv -d "$O/trace" -- python3 "$R/bench.py" --steps $steps --warmup 3 --no-cpu-baseline --no-roofline --no-extras "$@" > "$O/bench.out" 2>&1
python3 "$R/tools/steady_stats.py" "$O/trace" "$O/ss" 5 | tee "$O/summary.txt"
rm -rf "$O/trace"
