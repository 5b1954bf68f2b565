#!/usr/bin/env python3
"""Reference point only (never on the product path): liblpi_hip's GEMM vs the vendor library (torch.matmul -> hipBLASLt) on the
bench's bf16 shapes, plain C = A.B^T (+bias for ours), bf16 out.  Usage: python tools/gemm_vs_vendor.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lpi_amd import engine as E  # noqa: E402
from lpi_amd._lib import BF16  # noqa: E402

dev = "cuda:0"
Mv, Mt = 54528, 19712
shapes = [("v.qkv", Mv, 2304, 768), ("v.out", Mv, 768, 768), ("v.fc", Mv, 3072, 768), ("v.proj", Mv, 768, 3072),
          ("v.dqkv", Mv, 768, 2304), ("t.qkv", Mt, 1536, 512), ("t.fc", Mt, 2048, 512), ("t.proj", Mt, 512, 2048)]


def timeit(fn, n=10, reps=3):
    best = 1e9
    for _ in range(reps):
        for _ in range(2):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best


print(f"{'shape':8s} {'M':>6s} {'N':>5s} {'K':>5s} | {'lpi us':>8s} {'TF':>7s} | {'vendor us':>9s} {'TF':>7s}")
for name, M, N, K in shapes:
    a = torch.randn(M, K, device=dev).bfloat16()
    b = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    c = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
    bt = b.t()
    t1 = timeit(lambda: E.gemm(BF16, a, b, c, M, N, K))
    t2 = timeit(lambda: torch.matmul(a, bt, out=c))
    fl = 2.0 * M * N * K
    print(f"{name:8s} {M:6d} {N:5d} {K:5d} | {t1:8.1f} {fl / t1 / 1e6:7.1f} | {t2:9.1f} {fl / t2 / 1e6:7.1f}")
