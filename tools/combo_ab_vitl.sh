#!/usr/bin/env bash
# as tools/combo_ab.sh, on the ViT-L/14 workload of BASELINE configs[4] (512 pairs, depth 12, r 8)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
reps=$1; shift
cfgs=("$@")
A="--model ViT-L/14 --batch 512 --depth 12 --rank 8 --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-extras"
for rep in $(seq 1 $reps); do
  for cfg in "${cfgs[@]}"; do
    set -- $cfg; v=$1; shift; e="$*"
    lib=$R/lpi_amd/csrc/liblpi_hip.so; [ "$v" != base ] && lib=$R/lpi_amd/csrc/variants/liblpi_hip_$v.so
    env LPI_LIB=$lib $e timeout -k 10 300 python3 "$R/bench.py" $A 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%40s: %.3f ms  %.0f pairs/s' % ('$cfg', r['ms_per_step'], r['value']), flush=True)"
  done
done
