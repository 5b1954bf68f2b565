#!/usr/bin/env bash
# Runs ON the GPU box (via gpurun, from the repo root): the four bench lines, the two PMC passes and the two kernel-stat summaries
# that profiles/ is refreshed from (tools/pmc_summary.py and a copy step run afterwards in the build container).
# usage: gpurun --timeout 1500 -- 'bash tools/collect_artifacts.sh <tag>'      -> gpurun_out/<tag>_*
set -u
tag=${1:-final}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p "$O"
python bench.py > "$O/${tag}_bf16.json" 2>/dev/null
python bench.py --dtype f32 --no-cpu-baseline > "$O/${tag}_f32.json" 2>/dev/null
python bench.py --fwd-only --no-cpu-baseline > "$O/${tag}_fwd.json" 2>/dev/null
python bench.py --model ViT-L/14 --depth 12 --rank 8 --prompt-layers 12 --no-cpu-baseline --steps 10 --warmup 3 > "$O/${tag}_vitl14.json" 2>/dev/null
cd /tmp && export TMPDIR=/tmp
A="--steps 2 --warmup 1 --no-cpu-baseline --no-roofline"
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$O/${tag}_pmc_fetch" -- python3 "$R/bench.py" $A > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$O/${tag}_pmc_write" -- python3 "$R/bench.py" $A > /dev/null 2>&1
timeout 250 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/${tag}_stats_default" -- python3 "$R/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > /dev/null 2>&1
timeout 250 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/${tag}_stats_overlap" -- python3 "$R/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --overlap > /dev/null 2>&1
echo "collected $tag"
