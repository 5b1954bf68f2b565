#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace --memory-copy-trace run of tools/plugin_loop.py: per steady-state iteration the kernel busy time, the host-to-device copy
time and bytes, and how much of the copy time lies UNDER kernels (overlapped).   python tools/plugin_trace_summary.py <dir> [skip iterations]"""
import csv
import glob
import os
import sys

d = sys.argv[1]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ker, cp = [], []
for p in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(p)):
        ker.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
for p in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(p)):
        cp.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Direction", r.get("Name", "")), int(float(r.get("Bytes", r.get("Size", 0)) or 0))))
ker.sort()
cp.sort()
sgd = [k for k in ker if "sgd_step" in k[2]]
if len(sgd) < skip + 4:
    raise SystemExit(f"only {len(sgd)} optimiser launches in the trace")
t0, t1 = sgd[skip][1], sgd[-1][1]
iters = len(sgd) - 1 - skip
kw = [k for k in ker if t0 <= k[0] < t1]
busy = sum(e - s for s, e, _ in kw)
# this rocprofv3 version's memory-copy rows carry no byte count: the batch's image copy is told from the index arrays' by its duration (> 0.2 ms)
nominal = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0          # bytes of one image batch
big = [(c[0], c[1], c[2], nominal) for c in cp if t0 <= c[0] < t1 and c[1] - c[0] > 200_000]
small = [c for c in cp if t0 <= c[0] < t1 and c[1] - c[0] <= 200_000]


def overlap(c):
    s, e = c[0], c[1]
    tot = 0
    for ks, ke, _ in kw:
        if ke <= s:
            continue
        if ks >= e:
            break
        tot += min(e, ke) - max(s, ks)
    return tot


cp_time = sum(c[1] - c[0] for c in big)
ov = sum(overlap(c) for c in big)
print(f"steady window: {iters} iterations, wall {(t1 - t0) / iters / 1e6:.3f} ms per iteration, kernels busy {busy / iters / 1e6:.3f} ms per iteration, "
      f"{len(kw) / iters:.1f} kernel launches per iteration")
print(f"image-batch copies (host to device): {len(big) / iters:.2f} per iteration, {sum(c[3] for c in big) / iters / 1e6:.1f} MB per iteration, "
      f"{cp_time / iters / 1e6:.3f} ms per iteration ({sum(c[3] for c in big) / max(cp_time, 1):.1f} GB/s), {100.0 * ov / max(cp_time, 1):.1f} % of that time under kernels")
print(f"small copies (token ids, row starts, the loss line): {len(small) / iters:.2f} per iteration, {sum(c[1] - c[0] for c in small) / iters / 1e3:.1f} us per iteration")
names = sorted({k[2] for k in kw if "anonymous namespace" not in k[2] and "_GLOBAL__N_" not in k[2] and "lpi" not in k[2].lower()})
print("kernels that are not the library's:", names or "none")
