#!/usr/bin/env python3
"""Idle time between consecutive kernels of a steady-state window of a rocprofv3 --kernel-trace run.

usage: python3 tools/kgaps.py <dir with *kernel_trace.csv> [fraction of the trace to skip, default 0.5]

Prints the busy / wall / gap totals of the window, a histogram of the gaps, and the gaps summed by the pair
(kernel before, kernel after) so that the launches that leave the card idle can be named.
"""
import collections
import csv
import glob
import re
import statistics
import sys


def short(n):
    n = re.sub(r"\(.*", "", n)
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"at::native::", "", n)
    return n[:60]


def main():
    d = sys.argv[1]
    skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
    f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
    rows = list(csv.DictReader(open(f[0])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    seg = rows[int(len(rows) * skip):]
    s = [int(r["Start_Timestamp"]) for r in seg]
    e = [int(r["End_Timestamp"]) for r in seg]
    busy = sum(b - a for a, b in zip(s, e))
    wall = e[-1] - s[0]
    gaps = [s[i + 1] - e[i] for i in range(len(seg) - 1)]
    print(f"launches {len(seg)}  busy {busy / 1e6:.3f} ms  wall {wall / 1e6:.3f} ms  gaps {sum(gaps) / 1e6:.3f} ms "
          f"({100.0 * sum(gaps) / wall:.2f} %)  median gap {statistics.median(gaps) / 1e3:.2f} us  "
          f"mean {statistics.mean(gaps) / 1e3:.2f} us")
    h = collections.Counter(min(int(g / 1000), 20) if g >= 0 else -1 for g in gaps)
    print("gap histogram (us bucket: count):", sorted(h.items()))
    by = collections.defaultdict(lambda: [0, 0])
    for i, g in enumerate(gaps):
        k = (short(seg[i]["Kernel_Name"]), short(seg[i + 1]["Kernel_Name"]))
        by[k][0] += g
        by[k][1] += 1
    for k, (t, n) in sorted(by.items(), key=lambda kv: -kv[1][0])[:25]:
        print(f"{t / 1e6:8.3f} ms  n {n:5d}  mean {t / n / 1e3:7.2f} us   {k[0]}  ->  {k[1]}")


if __name__ == "__main__":
    main()
