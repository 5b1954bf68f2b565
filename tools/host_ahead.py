#!/usr/bin/env python3
"""Does the host keep ahead of the GPU?  Times the ENQUEUE of twelve training steps of the bench workload (no synchronisation inside) and the wall clock
including the final synchronisation.  Round 4, one MI355X box: 3.9-6.4 ms of host time per step against 22.2 ms of GPU time — the host runs many steps
ahead, so the 46 + 25 us gaps at the head of a step in a rocprofv3 trace (profiles/r04_bench_bf16_step_sequence.txt) are the profiler's, and the wall
clock of a step is the sum of its kernels."""
import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
sys.argv = sys.argv[:1]
a = bench.parse_args()
w = bench.Workload(a, "cuda:0", 0, "bf16", False, None)
for _ in range(5): w.step()
torch.cuda.synchronize()
ts = []
t0 = time.perf_counter()
for i in range(12):
    s = time.perf_counter(); w.step(); ts.append((time.perf_counter() - s) * 1e3)
host_total = (time.perf_counter() - t0) * 1e3
torch.cuda.synchronize()
total = (time.perf_counter() - t0) * 1e3
print("host enqueue ms per step:", [round(t, 2) for t in ts])
print(f"host loop total {host_total:.1f} ms, with final sync {total:.1f} ms for 12 steps ({total/12:.2f} per step)")
