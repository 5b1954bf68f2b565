mkdir -p gpurun_out/r2a
for cfg in "--cu-lanes 1 --cu-mode half" "--cu-lanes 2 --cu-mode half" "SIDE --cu-lanes 2 --cu-mode half" "SIDE --cu-lanes 1 --cu-mode half" "SIDE"; do
  if [[ "$cfg" == SIDE* ]]; then export LPI_MAIN_STREAM=side; c="${cfg#SIDE}"; else unset LPI_MAIN_STREAM; c="$cfg"; fi
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline $c 2>&1 | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|Error.*\|error.*' | tr '\n' ' '
  echo " <= $cfg"
done
