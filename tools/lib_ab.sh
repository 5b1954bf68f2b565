#!/usr/bin/env bash
# Runs ON the GPU box: whole-step A/B of library builds (bench.py with LPI_LIB), interleaved `reps` times.
# usage: bash tools/lib_ab.sh <reps> <suffix|base> [<suffix|base> ...]      suffix -> lpi_amd/csrc/variants/liblpi_hip_<suffix>.so
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/libab
mkdir -p "$O"
reps=$1; shift
A="--steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-extras"
for rep in $(seq 1 $reps); do
  for v in "$@"; do
    lib=$R/lpi_amd/csrc/liblpi_hip.so; [ "$v" != base ] && lib=$R/lpi_amd/csrc/variants/liblpi_hip_$v.so
    LPI_LIB=$lib timeout -k 10 200 python3 "$R/bench.py" $A > "$O/b_${v}_$rep.json" 2> "$O/b_${v}_$rep.err" || { echo "bench failed for $v"; tail -5 "$O/b_${v}_$rep.err"; exit 1; }
    python3 - "$O/b_${v}_$rep.json" "$v" <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{sys.argv[2]:>10}: {r['ms_per_step']:.3f} ms  {r['value']:.0f} pairs/s", flush=True)
PY
  done
done
