#!/usr/bin/env python3
"""ViT-L/14, 512 pairs: the pooled-row GEMMs have M = 512.  Whole-step A/B in one process, interleaved: engine._few_rows as shipped (M <= 256: these go to the
256 x 256 kernels, 8-32 tiles) against M <= 512 (the one-launch 32 x 32 kernel, 512-2048 workgroups).  python tools/probe/few_rows_m512.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import bench  # noqa: E402
from lpi_amd import _lib, engine as E  # noqa: E402
from lpi_amd.engine import F32  # noqa: E402

sys.argv = sys.argv[:1] + ["--model", "ViT-L/14", "--batch", "512", "--depth", "12", "--rank", "8", "--prompt-layers", "12"]
a = bench.parse_args()
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
wl = bench.Workload(a, dev, 0, "bf16", False, None)
for _ in range(3):
    wl.step()
shipped = E._few_rows


def upto512(dt, M, N, K):
    return M <= 512 and M % 128 == 0 and N % 128 == 0 and K % (32 if dt == F32 else 64) == 0


for rep in range(3):
    for tag, fn in (("M <= 256 (shipped)", shipped), ("M <= 512", upto512)):
        E._few_rows = fn
        for _ in range(2):
            wl.step()
        torch.cuda.synchronize()
        l0 = _lib.launch_count()
        t0 = time.perf_counter()
        for _ in range(8):
            wl.step()
        torch.cuda.synchronize()
        print(f"ViT-L/14 512 pairs, {tag:20s}: {(time.perf_counter() - t0) / 8 * 1e3:.3f} ms per step, {(_lib.launch_count() - l0) // 8} launches", flush=True)
