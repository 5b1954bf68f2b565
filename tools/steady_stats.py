#!/usr/bin/env python3
"""Steady-state kernel statistics of a rocprofv3 --kernel-trace run of bench.py: only the kernels of WHOLE training steps.

A step of the bench workload ends with exactly one `sgd_step_kernel` launch, so the trace is cut at those launches: everything up to the
`skip`-th one (engine construction, weight conversion / folding, workspace zero fills, warm-up) is dropped and what remains is N whole steps.

usage: python3 tools/steady_stats.py <dir with *kernel_trace.csv> <out prefix> [steps to skip, default 3]
writes <out prefix>_kernel_stats.csv  (Name, Calls, CallsPerStep, TotalDurationNs, AverageNs, MsPerStep, Percentage, MinNs, MaxNs)
       <out prefix>_step_sequence.txt (the launches of the last step in order: start offset, duration, gap to the previous kernel, name)
and prints a summary (launches per step, busy / wall per step, non-LPI kernels).
"""
import collections
import csv
import glob
import re
import sys


def short(n):
    n = re.sub(r"^void ", "", n)
    n = n.replace("(anonymous namespace)::", "")
    m = re.match(r"_ZN12_GLOBAL__N_1\d+([A-Za-z0-9_]+?_kernel)I(.*?)E+v", n)
    if m:
        return f"{m.group(1)}<{m.group(2)[:40]}>"
    n = re.sub(r"at::native::", "", n)
    n = re.sub(r"\((?:[^()]|\([^()]*\))*\)$", "", n)
    return n[:100]


def is_lpi(n):
    return "anonymous namespace" in n or n.startswith("_ZN12_GLOBAL") or "lpi" in n or n.startswith("ln_stats_finalize") or "StatFinP" in n


def main():
    d, out = sys.argv[1], sys.argv[2]
    skip = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
    rows = list(csv.DictReader(open(f[0])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    ends = [i for i, r in enumerate(rows) if "sgd_step_kernel" in r["Kernel_Name"]]
    if len(ends) <= skip + 1:
        raise SystemExit(f"only {len(ends)} steps in the trace")
    seg = rows[ends[skip - 1] + 1: ends[-1] + 1]
    nsteps = len(ends) - skip
    agg = collections.OrderedDict()
    for r in seg:
        n = r["Kernel_Name"]
        t = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        a = agg.setdefault(n, [0, 0, 1 << 62, 0])
        a[0] += 1
        a[1] += t
        a[2] = min(a[2], t)
        a[3] = max(a[3], t)
    tot = sum(a[1] for a in agg.values())
    with open(out + "_kernel_stats.csv", "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["Name", "Calls", "CallsPerStep", "TotalDurationNs", "AverageNs", "MsPerStep", "Percentage", "MinNs", "MaxNs"])
        for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            w.writerow([n, a[0], round(a[0] / nsteps, 3), a[1], round(a[1] / a[0], 1), round(a[1] / nsteps / 1e6, 4), round(100.0 * a[1] / tot, 3), a[2], a[3]])
    wall = int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])
    ext = {n: a for n, a in agg.items() if not is_lpi(n)}
    print(f"steady-state window: {nsteps} steps, {len(seg) / nsteps:.1f} launches per step, busy {tot / nsteps / 1e6:.3f} ms per step, "
          f"wall {wall / nsteps / 1e6:.3f} ms per step, {len(agg)} distinct kernels")
    print(f"non-LPI kernels in the window: {sum(a[0] for a in ext.values()) / nsteps:.2f} launches per step, "
          f"{sum(a[1] for a in ext.values()) / nsteps / 1e3:.1f} us per step")
    for n, a in ext.items():
        print(f"   EXT x{a[0] / nsteps:.2f}/step {a[1] / a[0] / 1e3:.1f} us  {short(n)}")
    last = rows[ends[-2] + 1: ends[-1] + 1]
    t0 = int(last[0]["Start_Timestamp"])
    with open(out + "_step_sequence.txt", "w") as fh:
        fh.write(f"# the {len(last)} launches of one steady-state step: start offset us | duration us | gap to previous us | kernel\n")
        prev = None
        for r in last:
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            gap = 0.0 if prev is None else (s - prev) / 1e3
            fh.write(f"{(s - t0) / 1e3:10.1f} {(e - s) / 1e3:9.2f} {gap:7.2f}  {short(r['Kernel_Name'])}\n")
            prev = e
    small = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in last]
    print(f"last step: {len(last)} launches; {sum(1 for t in small if t < 15000)} of them under 15 us = {sum(t for t in small if t < 15000) / 1e3:.1f} us; "
          f"under 50 us: {sum(1 for t in small if t < 50000)} = {sum(t for t in small if t < 50000) / 1e3:.1f} us")


if __name__ == "__main__":
    main()
