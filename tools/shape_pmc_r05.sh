#!/usr/bin/env bash
# Runs ON the GPU box: per-shape FETCH_SIZE / WRITE_SIZE of the persistent GEMM with the weight slices on (default) and off (LPI_TUNING=15=-1)
# -> gpurun_out/shapes/{slices,noslices}.json      usage: gpurun -- 'bash tools/shape_pmc_r05.sh'
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/shapes; mkdir -p "$O"; cd /tmp; export TMPDIR=/tmp
for arm in slices noslices; do
  T=""; [ $arm = noslices ] && T="15=-1"
  LPI_TUNING=$T timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$O/f_$arm" -- python3 "$R/tools/gemm_shape_pmc.py" > "$O/table_$arm.txt" 2>/dev/null
  LPI_TUNING=$T timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$O/w_$arm" -- python3 "$R/tools/gemm_shape_pmc.py" > /dev/null 2>&1
  echo "== $arm"; python3 "$R/tools/gemm_shape_pmc_summary.py" "$O/table_$arm.txt" "$O/f_$arm" "$O/w_$arm" "$O/$arm.json"
  rm -rf "$O/f_$arm" "$O/w_$arm"
done
