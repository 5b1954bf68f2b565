mkdir -p gpurun_out/r2b
timeout 1500 python -m pytest tests -x -q -m gpu -p no:cacheprovider 2>&1 | tail -25 > gpurun_out/r2b/pytest.txt
cat gpurun_out/r2b/pytest.txt
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r2b/bench.json 2> gpurun_out/r2b/bench.err
tail -3 gpurun_out/r2b/bench.err; cat gpurun_out/r2b/bench.json
timeout 300 python bench.py --gpus 2 --share-gpu --steps 5 --warmup 2 --batch 64 --no-roofline > gpurun_out/r2b/bench_dp2_shared.json 2> gpurun_out/r2b/bench_dp2.err
tail -5 gpurun_out/r2b/bench_dp2.err; cat gpurun_out/r2b/bench_dp2_shared.json
