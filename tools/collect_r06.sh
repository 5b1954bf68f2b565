#!/usr/bin/env bash
# Runs ON the GPU box (via gpurun, from the repo root): steady-state kernel statistics and the PMC passes of one bench workload, summarised
# on the box so that only the small files travel back.
# usage: gpurun --timeout 1200 -- 'bash tools/collect_r06.sh <tag> [bench.py workload arguments]'   -> gpurun_out/<tag>/
#   kernel statistics: tools/steady_stats.py over a --kernel-trace run, cut to whole steps after the fifth optimiser launch
#   pmc.json         : tools/pmc_summary.py over four separate --pmc passes (FETCH_SIZE | WRITE_SIZE | MFMA + clock | waits + LDS), stamped
#                      with the workload / dtype / tuning / library version that bench.py matches before it attaches the numbers
# Every rocprofv3 line has the program itself after `--` (python3 bench.py: no wrapper, no exec hop); --pmc goes with --kernel-trace only.
set -u
tag=${1:-r06}; shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$tag
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
W="$*"
A="$W --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-extras"
timeout 400 rocprofv3 --kernel-trace --output-format csv -d "$O/trace" -- python3 "$R/bench.py" $W --steps 12 --warmup 3 --no-cpu-baseline --no-roofline --no-extras > "$O/bench.out" 2>&1
python3 "$R/tools/steady_stats.py" "$O/trace" "$O/ss" 5 | tee "$O/summary.txt"
rm -rf "$O/trace"
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$O/pmc_fetch" -- python3 "$R/bench.py" $A > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$O/pmc_write" -- python3 "$R/bench.py" $A > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$O/pmc_mfma" -- python3 "$R/bench.py" $A > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_VALU --kernel-trace --output-format csv -d "$O/pmc_wait" -- python3 "$R/bench.py" $A > /dev/null 2>&1
for d in pmc_fetch pmc_write pmc_mfma pmc_wait; do
  n=$(find "$O/$d" -name "*counter_collection.csv" | head -1); [ -n "$n" ] && echo "$d: $(wc -l < "$n") rows"
done
# the library build and the tuning vector bench.py will compare against (the same LPI_TUNING environment as the passes above)
read -r libv tun < <(python3 -c "import sys; sys.path.insert(0, '$R'); from lpi_amd import _lib; L = _lib.load(); print(L.lpi_version(), ','.join(str(int(L.lpi_get_tuning(k))) for k in range(8)))")
python3 "$R/tools/pmc_summary.py" "$O/pmc.json" bf16 "$tun" "$libv" "python3 bench.py $W (+ --steps 3 --warmup 1 in each counter pass)" \
  "$O/pmc_fetch" "$O/pmc_write" "$O/pmc_mfma" "$O/pmc_wait" | tee -a "$O/summary.txt"
rm -rf "$O/pmc_fetch" "$O/pmc_write" "$O/pmc_mfma" "$O/pmc_wait"
ls -la "$O"
