mkdir -p gpurun_out/r2c
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_plugin_gpu.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -15 > gpurun_out/r2c/pytest.txt
cat gpurun_out/r2c/pytest.txt
bash tools/collect_r02.sh r2c_prof
