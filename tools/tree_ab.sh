#!/usr/bin/env bash
# Runs ON the GPU box: whole-step A/B of two source TREES (each with its own built liblpi_hip.so), interleaved `reps` times on one box.
# usage: bash tools/tree_ab.sh <reps> <tree A> <tree B> [extra bench args]     e.g. tools/tree_ab.sh 3 build_var/base .
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
reps=$1; ta=$2; tb=$3; shift 3
A="--steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-extras $*"
for rep in $(seq 1 $reps); do
  for t in "$ta" "$tb"; do
    (cd "$R/$t" && timeout -k 10 300 python3 bench.py $A 2>/dev/null | python3 -c "
import json, sys
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(f'{sys.argv[1]:>20}: {r[\"ms_per_step\"]:.3f} ms (median {r[\"median_ms_per_step\"]:.3f})  {r[\"value\"]:.0f} pairs/s', flush=True)" "$t") || { echo "bench failed in $t"; exit 1; }
  done
done
