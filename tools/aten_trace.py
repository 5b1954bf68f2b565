#!/usr/bin/env python3
"""Which ATen ops (and from which source line) launch GPU kernels inside ONE steady-state training step of the bench workload."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

sys.argv = sys.argv[:1]
a = bench.parse_args()
w = bench.Workload(a, "cuda:0", 0, "bf16", False, None)
for _ in range(4):
    w.step()
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile  # noqa: E402

with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    w.step()
    torch.cuda.synchronize()
ka = prof.key_averages(group_by_stack_n=6)
for e in sorted(ka, key=lambda e: -getattr(e, "device_time_total", getattr(e, "cuda_time_total", 0))):
    dt = getattr(e, "self_device_time_total", getattr(e, "self_cuda_time_total", 0))
    if dt <= 0 or not e.key.startswith("aten::"):
        continue
    stack = [s for s in e.stack if "/lpi_amd/" in s or "bench.py" in s or "/torch/optim" in s][:3]
    print(f"{e.key:34s} x{e.count:3d} self device {dt:8.1f} us  shapes {str(e.input_shapes)[:60]:60s} {' <- '.join(s.split('/')[-1] for s in stack)}")

print("---- device activities that are not LPI kernels")
import collections
acc = collections.Counter(); tim = collections.Counter()
for e in prof.events():
    if e.device_type.name != "CPU" and "anonymous namespace" not in e.name and "_GLOBAL__N_" not in e.name:
        acc[e.name[:90]] += 1; tim[e.name[:90]] += e.device_time if hasattr(e, "device_time") else e.cuda_time
for k, n in acc.most_common():
    print(f"x{n:3d} {tim[k]:8.1f} us  {k}")
print("---- aten::copy_ / aten::to call sites")
for e in ka:
    if e.key in ("aten::copy_", "aten::to", "aten::_to_copy", "aten::contiguous", "aten::clone") :
        stack = [s for s in e.stack if "/lpi_amd/" in s or "bench.py" in s][:2]
        print(f"{e.key:20s} x{e.count:3d} {str(e.input_shapes)[:50]:50s} {' <- '.join(s.split('/')[-1] for s in stack)}")
