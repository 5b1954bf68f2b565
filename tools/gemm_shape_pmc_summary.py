#!/usr/bin/env python3
"""Per-shape HBM traffic of the persistent GEMM from the rocprofv3 --pmc passes of tools/gemm_shape_pmc.py.
usage: gemm_shape_pmc_summary.py <table.json> <fetch pass dir> <write pass dir> <out.json>"""
import csv
import glob
import json
import os
import sys

table = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])


def rows(d, counter):
    out = []
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            if "gemm256" in r["Kernel_Name"] and r["Counter_Name"] == counter:
                out.append((int(r["Dispatch_Id"]), float(r["Counter_Value"]), float(r.get("End_Timestamp", 0) or 0) - float(r.get("Start_Timestamp", 0) or 0)))
    return sorted(out)


f, w = rows(sys.argv[2], "FETCH_SIZE"), rows(sys.argv[3], "WRITE_SIZE")
res, i = [], 0
for t in table:
    n = t["reps"]
    fs, ws = f[i:i + n], w[i:i + n]
    i += n
    fetch = 2 * 1024 * sum(x[1] for x in fs[1:]) / max(len(fs) - 1, 1)          # gfx950: FETCH_SIZE counts half of a wide streaming read (KiB)
    write = 1024 * sum(x[1] for x in ws[1:]) / max(len(ws) - 1, 1)
    us = sum(x[2] for x in fs[1:]) / max(len(fs) - 1, 1) / 1e3
    res.append({**t, "fetch_mb": round(fetch / 1e6, 1), "write_mb": round(write / 1e6, 1), "alg_read_mb": round(t["reads"] / 1e6, 1),
                "alg_write_mb": round(t["writes"] / 1e6, 1), "fetch_over_alg": round(fetch / t["reads"], 3), "write_over_alg": round(write / t["writes"], 3),
                "us_in_counter_pass": round(us, 1)})
json.dump(res, open(sys.argv[4], "w"), indent=1)
print(f"{'shape':16s} {'M':>6s} {'N':>5s} {'K':>5s} {'fetch MB':>9s} {'alg':>7s} {'ratio':>6s} {'write MB':>9s} {'alg':>7s} {'ratio':>6s} {'us':>7s}")
for r in res:
    print(f"{r['name']:16s} {r['M']:6d} {r['N']:5d} {r['K']:5d} {r['fetch_mb']:9.1f} {r['alg_read_mb']:7.1f} {r['fetch_over_alg']:6.2f} {r['write_mb']:9.1f} {r['alg_write_mb']:7.1f} {r['write_over_alg']:6.2f} {r['us_in_counter_pass']:7.1f}")
