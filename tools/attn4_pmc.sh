#!/usr/bin/env bash
# Runs ON the GPU box: PMC passes of the attention-backward micro-benchmark (tools/attn4_time.py) -> gpurun_out/<tag>/pmc.json
# usage: gpurun -- 'bash tools/attn4_pmc.sh <tag> <keys...>'
set -u
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$tag
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$O/pmc_fetch" -- python3 "$R/tools/attn4_time.py" "$@" > /dev/null 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$O/pmc_write" -- python3 "$R/tools/attn4_time.py" "$@" > /dev/null 2>&1
timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$O/pmc_mfma" -- python3 "$R/tools/attn4_time.py" "$@" > /dev/null 2>&1
timeout 200 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_VALU --kernel-trace --output-format csv -d "$O/pmc_wait" -- python3 "$R/tools/attn4_time.py" "$@" > /dev/null 2>&1
timeout 200 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES --kernel-trace --output-format csv -d "$O/pmc_inst" -- python3 "$R/tools/attn4_time.py" "$@" > /dev/null 2>&1
find "$O" -name "*agent_info.csv" -delete
python3 "$R/tools/pmc_summary.py" "$O/pmc.json" bf16 "0" 0 "tools/attn4_time.py $*" "$O/pmc_fetch" "$O/pmc_write" "$O/pmc_mfma" "$O/pmc_wait" "$O/pmc_inst" > /dev/null 2>&1
find "$O" -name "*counter_collection.csv" -size +2M -delete
python3 - <<PY
import json
d=json.load(open("$O/pmc.json"))
for k,v in d.get("kernels",d).items():
    if "attn" in k: print(k, json.dumps(v))
PY
