for v in 2 0 2 0; do
LPI_TUNING="2=$v" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-roofline 2>/dev/null | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' | tr '\n' ' '; echo " <= key2=$v"
done
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -3
