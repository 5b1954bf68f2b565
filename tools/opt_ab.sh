#!/usr/bin/env bash
# Runs ON the GPU box: whole-step A/B of bench.py argument sets, interleaved `reps` times.
# usage: bash tools/opt_ab.sh <reps> "--engine-opt stream_pool=0" "" ...      ("" = the defaults)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/optab
mkdir -p "$O"
reps=$1; shift
A="--steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-extras"
for rep in $(seq 1 $reps); do
  i=0
  for cfg in "$@"; do
    i=$((i + 1))
    timeout -k 10 200 python3 "$R/bench.py" $A $cfg > "$O/b_${i}_$rep.json" 2> "$O/b_${i}_$rep.err" || { echo "bench failed for $cfg"; tail -5 "$O/b_${i}_$rep.err"; exit 1; }
    python3 - "$O/b_${i}_$rep.json" "$cfg" <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{sys.argv[2] or '(defaults)':>40}: {r['ms_per_step']:.3f} ms  {r['value']:.0f} pairs/s  {r['launches_per_step']} launches", flush=True)
PY
  done
done
