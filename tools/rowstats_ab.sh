#!/usr/bin/env bash
# Runs ON the GPU box: whole-step A/B of LPI_ROWSTATS x LPI_LN_FOLD (bench.py, interleaved twice)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/rsab
mkdir -p "$O"
A="--steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-extras"
for rep in 1 2; do
  for cfg in "0 1" "0 2" "1 2" "2 2"; do
    set -- $cfg
    LPI_ROWSTATS=$1 LPI_LN_FOLD=$2 timeout -k 10 200 python3 "$R/bench.py" $A > "$O/b_$1_$2_$rep.json" 2> "$O/b_$1_$2_$rep.err" || { echo "bench failed rowstats=$1 fold=$2"; tail -5 "$O/b_$1_$2_$rep.err"; exit 1; }
    python3 - "$O/b_$1_$2_$rep.json" "$1" "$2" <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"rowstats={sys.argv[2]} fold={sys.argv[3]}: {r['ms_per_step']:.3f} ms  {r['value']:.0f} pairs/s", flush=True)
PY
  done
done
