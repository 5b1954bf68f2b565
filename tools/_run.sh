# full GPU suite + default bench + profiles for the docs
mkdir -p gpurun_out/r2e
timeout 1500 python -m pytest tests -x -q -m gpu -p no:cacheprovider 2>&1 | tail -5 > gpurun_out/r2e/pytest.txt; cat gpurun_out/r2e/pytest.txt
timeout 600 python bench.py > gpurun_out/r2e/bench.json 2> gpurun_out/r2e/bench.err; tail -c 300 gpurun_out/r2e/bench.json
bash tools/collect_r02.sh r2e_prof > gpurun_out/r2e/collect.txt 2>&1; tail -3 gpurun_out/r2e/collect.txt
