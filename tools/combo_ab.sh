#!/usr/bin/env bash
# Runs ON the GPU box: interleaved whole-step A/B of (library build, environment) pairs: bash tools/combo_ab.sh <reps> "suffix|base VAR=x ..." ...
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
reps=$1; shift
cfgs=("$@")
A="--steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-extras"
for rep in $(seq 1 $reps); do
  for cfg in "${cfgs[@]}"; do
    set -- $cfg; v=$1; shift; e="$*"
    lib=$R/lpi_amd/csrc/liblpi_hip.so; [ "$v" != base ] && lib=$R/lpi_amd/csrc/variants/liblpi_hip_$v.so
    env LPI_LIB=$lib $e timeout -k 10 200 python3 "$R/bench.py" $A 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%40s: %.3f ms  %.0f pairs/s' % ('$cfg', r['ms_per_step'], r['value']), flush=True)"
  done
done
