mkdir -p gpurun_out/r2f
timeout 1800 python -m pytest tests -x -q -m gpu -p no:cacheprovider 2>&1 | tail -5 > gpurun_out/r2f/pytest.txt; cat gpurun_out/r2f/pytest.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke
timeout 600 python bench.py > gpurun_out/r2f/bench.json 2> gpurun_out/r2f/bench.err; tail -c 200 gpurun_out/r2f/bench.json; echo
timeout 300 python bench.py --model ViT-L/14 --depth 12 --rank 8 --prompt-layers 12 --no-cpu-baseline --no-extras --steps 10 --warmup 3 > gpurun_out/r2f/bench_vitl14.json 2>/dev/null; head -c 300 gpurun_out/r2f/bench_vitl14.json; echo
timeout 300 python bench.py --gpus 2 --share-gpu --steps 5 --warmup 2 --batch 64 --no-roofline > gpurun_out/r2f/bench_dp2_shared.json 2>/dev/null; head -c 200 gpurun_out/r2f/bench_dp2_shared.json; echo
bash tools/collect_r02.sh r2f_prof > gpurun_out/r2f/collect.txt 2>&1; tail -2 gpurun_out/r2f/collect.txt
