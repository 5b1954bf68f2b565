timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm" -p no:cacheprovider 2>&1 | tail -3
timeout 300 python tools/gemm_ab.py 2 2 0 2>&1 | tail -14
