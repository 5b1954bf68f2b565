timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "attention" -p no:cacheprovider 2>&1 | tail -4
timeout 300 python tools/attn_bench.py 2>&1 | tail -6
for i in 1 2; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-roofline 2>&1 | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' | tr '\n' ' '; echo
done
