#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes into profiles/<tag>_pmc_traffic.json.

HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE/WRITE_SIZE are in KiB and, on gfx950, FETCH_SIZE
reports exactly half of the bytes of a wide coalesced streaming read (MI355X_MICROARCH.md, HBM section); the two counters
are collected in separate passes (they do not fit one pass).

usage: tools/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> [command string]
"""
import collections
import csv
import json
import re
import sys


def load(path):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        n = r["Kernel_Name"]
        m = re.match(r"_ZN\d+_GLOBAL__N_1(\d+)", n)      # rocprofv3 leaves some instantiations (fp16 template arguments) mangled
        if m:
            k = int(m.group(1))
            n = n[m.end():m.end() + k]
        n = re.sub(r"\(anonymous namespace\)::", "", n)
        n = re.sub(r"<.*", "", n).replace("void ", "")
        a = agg[n]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return agg


f, w = load(sys.argv[1]), load(sys.argv[2])
out = {"command": sys.argv[4] if len(sys.argv) > 4 else None,
       "formula": "(2*FETCH_SIZE + WRITE_SIZE) * 1024 bytes per launch; separate --pmc passes; gfx950 FETCH_SIZE half-count corrected",
       "kernels": {}}
for k in sorted(f, key=lambda k: -f[k][1]):
    if not k.startswith(("gemm", "attn", "ln_", "vis_", "txt_", "pool", "rows_sum", "patchify")):
        continue
    nf, vf = f[k]
    nw, vw = w.get(k, (1, 0.0))
    out["kernels"][k] = {"launches": nf, "fetch_kib_per_launch": round(vf / nf, 1), "write_kib_per_launch": round(vw / max(nw, 1), 1),
                         "hbm_mb_per_launch": round((2 * vf / nf + vw / max(nw, 1)) * 1024 / 1e6, 2)}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out["kernels"].get("gemm256_kernel")))
