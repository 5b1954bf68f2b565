timeout 900 python -m pytest tests/test_model_gpu.py tests/test_kernels_gpu.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -6
for v in 0 1 0 1; do
LPI_L0_PROMPT_ROWS=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-roofline 2>/dev/null | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' | tr '\n' ' '; echo " <= L0_PROMPT_ROWS=$v"
done
