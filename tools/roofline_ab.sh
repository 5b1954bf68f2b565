#!/usr/bin/env bash
# Runs ON the GPU box: step time AND the GEMM roofline block of bench.py for environment settings, interleaved: bash tools/roofline_ab.sh <reps> "VAR=a" "-" ...
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
reps=$1; shift
cfgs=("$@")
for rep in $(seq 1 $reps); do
  for cfg in "${cfgs[@]}"; do
    e=""; [ "$cfg" != "-" ] && e="$cfg"
    env $e timeout -k 10 300 python3 "$R/bench.py" --steps 30 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=r['roofline']
print('%30s: %.3f ms  %.0f pairs/s   GEMM %.1f TF frac %.4f  %.3f ms/step over %d launches' % ('$cfg', r['ms_per_step'], r['value'], f['achieved'], f['frac'], f['gemm_ms_per_step'], f['launches_per_step']), flush=True)"
  done
done
