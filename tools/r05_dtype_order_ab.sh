cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5e; O=gpurun_out/r5e
python bench.py --no-cpu-baseline --no-roofline --no-extras > $O/bf16_a.json 2>/dev/null
python bench.py --dtype f16 --no-cpu-baseline --no-roofline --no-extras > $O/f16_a.json 2>/dev/null
python bench.py --no-cpu-baseline --no-roofline --no-extras > $O/bf16_b.json 2>/dev/null
python bench.py --dtype f16 --no-cpu-baseline --no-roofline --no-extras > $O/f16_b.json 2>/dev/null
python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-roofline --no-extras > $O/bf16_long.json 2>/dev/null
(rocm-smi --showpower --showclocks 2>&1 | head -30) > $O/smi.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5e/*.json')):
    j=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], j['value'], j['ms_per_step'], j['median_ms_per_step'])
PY
head -30 $O/smi.txt
