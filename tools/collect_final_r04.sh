#!/usr/bin/env bash
# Runs ON the GPU box: the round's bench lines and diagnostics of the final build -> gpurun_out/fin/ (copied into profiles/r04_* afterwards).
#   bench.json          python bench.py (the driver's command: main record + parity_mode / fwd_only / f16_mode / vit_l14 / eval_path + cpu_baseline)
#   dp4.json, dp4ll.json  bench.py --gpus 4 --share-gpu --batch 64, default loss mode and --local-loss --gather-with-grad
#   aten_trace.txt      tools/aten_trace.py: ATen ops / foreign kernels inside one steady-state step
#   gemm_stamps.json    tools/gemm_stamps.py on the diagnostic build, gemm_product.json: the same shapes timed on the product library
# usage: gpurun --timeout 1200 -- 'bash tools/collect_final_r04.sh'
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/fin
mkdir -p "$O"
cd "$R"
python3 bench.py > "$O/bench.json" 2> "$O/bench.err"; echo "bench rc=$?"
python3 bench.py --gpus 4 --share-gpu --batch 64 > "$O/dp4.json" 2> "$O/dp4.err"; echo "dp4 rc=$?"
python3 bench.py --gpus 4 --share-gpu --batch 64 --local-loss --gather-with-grad > "$O/dp4ll.json" 2> "$O/dp4ll.err"; echo "dp4ll rc=$?"
python3 tools/aten_trace.py > "$O/aten_trace.txt" 2>&1; echo "aten rc=$?"
LPI_STAMP_PRODUCT=1 python3 tools/gemm_stamps.py "$O/gemm_product.json" > "$O/gemm_product.txt" 2>&1; echo "product rc=$?"
python3 tools/gemm_stamps.py "$O/gemm_stamps.json" > "$O/gemm_stamps.txt" 2>&1; echo "stamps rc=$?"
python3 tools/show_bench.py "$O/bench.json" 2>/dev/null | head -12
cat "$O/gemm_product.txt" "$O/gemm_stamps.txt" | grep -v amdgpu.ids
