#!/usr/bin/env bash
# Runs ON the GPU box: rocprofv3 kernel + memory-copy trace of the plugin's hot loop -> gpurun_out/ptrace/summary_{f32,u8}.txt
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/ptrace; mkdir -p "$O"; cd /tmp; export TMPDIR=/tmp
for pf in f32 u8; do
  timeout 400 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$O/t_$pf" -- python3 "$R/tools/plugin_loop.py" 40 $pf > "$O/run_$pf.out" 2>&1
  echo "== pixel_format $pf" | tee "$O/summary_$pf.txt"
  nb=154140672; [ $pf = u8 ] && nb=38535168
  python3 "$R/tools/plugin_trace_summary.py" "$O/t_$pf" 8 $nb | tee -a "$O/summary_$pf.txt"
  rm -rf "$O/t_$pf"
done
