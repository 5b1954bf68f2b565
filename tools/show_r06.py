#!/usr/bin/env python3
"""Summarise the round's full bench lines (gpurun_out/r06_v3_*.json, one box each: the FINAL tree; --prefix r06_v2_ = library 604 before the last two kernel changes, --prefix r06_final_ =
library 603): every line's records, the MEDIAN run by `value` -> profiles/r06_bench_bf16.json,
all lines -> profiles/r06_bench_bf16_boxes.txt.   python tools/show_r06.py [--write]"""
import glob
import json
import os
import shutil
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
runs = []
PREFIX = sys.argv[sys.argv.index("--prefix") + 1] if "--prefix" in sys.argv else "r06_v3_"
for p in sorted(glob.glob(os.path.join(REPO, "gpurun_out", PREFIX + "*.json"))):
    try:
        d = json.loads(open(p).read().strip().splitlines()[-1])
    except Exception:      # noqa: BLE001
        continue
    runs.append((p, d))
runs.sort(key=lambda r: r[1]["value"])
g = lambda d, *ks: (lambda v: v)(__import__("functools").reduce(lambda a, k: (a or {}).get(k) if isinstance(a, dict) else None, ks, d))  # noqa: E731
rows = []
for p, d in runs:
    rows.append(dict(file=os.path.basename(p), value=d["value"], ms=d["ms_per_step"], mfu=d.get("step_mfma_frac"), hfu=d.get("hw_flop_frac"),
                     gemm_tf=g(d, "roofline", "achieved"), frac=g(d, "roofline", "frac"), us=g(d, "roofline", "avg_launch_us"), traffic=g(d, "roofline", "traffic"),
                     f16=g(d, "f16_mode", "value"), f32=g(d, "parity_mode", "value"), fwd=g(d, "fwd_only", "value"), vitl=g(d, "vit_l14", "value"),
                     noshared=g(d, "without_shared_text_prefix", "value"), plugin=g(d, "plugin_step", "value"), plugin_ratio=g(d, "plugin_step", "vs_bare_step"),
                     plugin_u8=g(d, "plugin_step_u8", "vs_bare_step"), ref_order=g(d, "plugin_step_reference_order", "value"), cpu=g(d, "cpu_baseline", "value"),
                     launches=d.get("launches_per_step")))
for r in rows:
    print(" ".join(f"{k}={v}" for k, v in r.items()))
if rows:
    med = runs[(len(runs) - 1) // 2]
    print("median run:", os.path.basename(med[0]), med[1]["value"])
    par = med[1].get("parity", {})
    for k, v in par.items():
        print("  parity", k, {a: (round(b, 8) if isinstance(b, float) else b) for a, b in v.items() if a not in ("fixture", "bar", "text_layout")})
    if "--write" in sys.argv:
        shutil.copy(med[0], os.path.join(REPO, "profiles", "r06_bench_bf16.json"))
        with open(os.path.join(REPO, "profiles", "r06_bench_bf16_boxes.txt"), "w") as f:
            f.write("# python bench.py, one gpurun call = one box each, the FINAL tree of round 6 (library 604), sorted by `value`; the committed line\n"
                    "# profiles/r06_bench_bf16.json is the MEDIAN run (" + os.path.basename(med[0]) + ")\n")
            for p, d in runs:
                f.write(json.dumps(d) + "\n")
