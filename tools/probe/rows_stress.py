"""Chunked timing of back-to-back launches: does a kernel show rare long stalls?  100 chunks of 200 launches each, per kernel; prints median / max chunk.
The rows binding builds a ctypes descriptor array per call: the series runs with the Python garbage collector on, then off."""
import gc
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lpi_amd import _lib, engine as E  # noqa: E402

dev = torch.device("cuda:0")
s = torch.cuda.current_stream().cuda_stream
B, td, dt = 256, torch.bfloat16, E.BF16
probs = [dict(M=B, N=n, K=k, a=torch.randn(B, k, device=dev).to(td), b=(0.05 * torch.randn(n, k, device=dev)).to(td), c=torch.zeros(B, n, device=dev, dtype=td),
              bias=torch.randn(n, device=dev), aux=None) for n, k in ((768, 3072), (512, 2048))]
x = torch.zeros(4096, device=dev)


def rows():
    _lib.gemm_rows(dt, dt, E.EPI_NONE, 1.0, probs, s)


def k128():
    p = probs[1]
    _lib.call("lpi_gemm_nt", dt, dt, p["M"], p["N"], p["K"], p["a"], p["K"], p["b"], p["K"], p["c"], p["N"], p["bias"], None, 0, 0, None, 0, 1.0, s)


def tiny():
    x.add_(1.0)


for name, fn in (("rows pair", rows), ("128x128 one problem", k128), ("torch add_ 4096", tiny), ("rows pair", rows), ("rows pair, gc off", rows), ("rows pair, gc off", rows)):
    (gc.disable if "gc off" in name else gc.enable)()
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    ts, wall = [], []
    for c in range(100):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(200):
            fn()
        e1.record()
        torch.cuda.synchronize()
        wall.append((time.perf_counter() - t0) * 1e6 / 200)
        ts.append(e0.elapsed_time(e1) * 1e3 / 200)
    st = sorted(ts)
    print(f"{name:22s}: per launch median {st[50]:7.2f} us, min {st[0]:7.2f}, max {st[-1]:7.2f}, chunks above 2x median: {sum(t > 2 * st[50] for t in ts)}; "
          f"host wall per launch median {sorted(wall)[50]:6.2f} max {max(wall):7.2f}", flush=True)
