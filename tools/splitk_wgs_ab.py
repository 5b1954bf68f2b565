#!/usr/bin/env python3
"""Whole-step sweep of engine.SPLITK_WGS (the workgroups a few-row split-K launch aims at) in ONE process, interleaved: python tools/splitk_wgs_ab.py [values...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from lpi_amd import engine as E  # noqa: E402

vals = [int(v) for v in sys.argv[1:]] or [128, 192, 256, 384, 512, 768]
sys.argv = sys.argv[:1]
a = bench.parse_args()
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
wl = bench.Workload(a, dev, 0, "bf16", False, None)
for _ in range(5):
    wl.step()
for rep in range(3):
    for v in vals:
        E.SPLITK_WGS = v
        for _ in range(3):
            wl.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(30):
            wl.step()
        torch.cuda.synchronize()
        print(f"SPLITK_WGS = {v:4d}: {(time.perf_counter() - t0) / 30 * 1e3:.3f} ms per step", flush=True)
