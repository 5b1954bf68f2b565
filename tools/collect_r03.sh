#!/usr/bin/env bash
# Runs ON the GPU box (via gpurun, from the repo root): kernel-stats trace and the PMC passes of the default bench command.
# usage: gpurun --timeout 1500 -- 'bash tools/collect_r03.sh <tag>'      -> gpurun_out/<tag>/...
# Every rocprofv3 line has the program itself after `--` (python3 bench.py: no wrapper, no exec hop) and --pmc is never combined with
# anything but --kernel-trace.
set -u
tag=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$tag
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
A="--steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-extras"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -- python3 "$R/bench.py" --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-extras > "$O/stats.out" 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$O/pmc_fetch" -- python3 "$R/bench.py" $A > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$O/pmc_write" -- python3 "$R/bench.py" $A > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$O/pmc_mfma" -- python3 "$R/bench.py" $A > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_VALU --kernel-trace --output-format csv -d "$O/pmc_wait" -- python3 "$R/bench.py" $A > /dev/null 2>&1
# keep only what the summaries need (the merged gpurun_out is capped at 64 MiB)
find "$O" -name "*agent_info.csv" -delete
for d in pmc_fetch pmc_write pmc_mfma pmc_wait; do
  n=$(find "$O/$d" -name "*counter_collection.csv" | head -1); [ -n "$n" ] && echo "$d: $(wc -l < "$n") rows"
done
find "$O/stats" -name "*kernel_trace.csv" -delete
ls -la "$O" "$O/stats"/* | head -30
du -sh "$O"
