timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "single_pass or attention" -p no:cacheprovider 2>&1 | tail -4
