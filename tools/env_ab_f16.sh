#!/usr/bin/env bash
# as tools/env_ab.sh, in the f16 operand mode
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
reps=$1; shift
A="--dtype f16 --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-extras"
for rep in $(seq 1 $reps); do
  for cfg in "$@"; do
    e=""; [ "$cfg" != "-" ] && e="$cfg"
    env $e timeout -k 10 200 python3 "$R/bench.py" $A 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%28s: %.3f ms  %.0f pairs/s' % ('$cfg', r['ms_per_step'], r['value']), flush=True)"
  done
done
