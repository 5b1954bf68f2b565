for cfg in "" "--overlap" "--overlap --text-cus 32" "--overlap --text-cus 40" "--overlap --text-cus 48" ""; do
python bench.py $cfg --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-roofline 2>&1 | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|Error.*' | head -3 | tr '\n' ' '; echo " <= $cfg"
done
