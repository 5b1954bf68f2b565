timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "f16" -p no:cacheprovider 2>&1 | tail -15
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "f16" -p no:cacheprovider -s 2>&1 | tail -15
for dt in bf16 f16 bf16 f16; do
python bench.py --dtype $dt --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-roofline 2>/dev/null | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' | tr '\n' ' '; echo " <= $dt"
done
