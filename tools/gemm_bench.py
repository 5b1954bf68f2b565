#!/usr/bin/env python3
"""GEMM micro-benchmark on the bench's shapes: 128x128 kernel vs 256x256 8-phase kernel, interleaved in one process,
random data (cdna guide rules 24/25).  Usage: python tools/gemm_bench.py [bf16|f32]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lpi_amd import engine as E  # noqa: E402
from lpi_amd._lib import BF16, F32, call  # noqa: E402

dt = F32 if (len(sys.argv) > 1 and sys.argv[1] == "f32") else BF16
TD = torch.float32 if dt == F32 else torch.bfloat16
dev = "cuda:0"
Mv, Mt = 54528, 19712
shapes = [  # (name, M, N, K, out f32?, epi, residual)
    ("v.qkv", Mv, 2304, 768, False, 0, False), ("v.out+res", Mv, 768, 768, True, 0, True), ("v.fc+gelu", Mv, 3072, 768, False, 1, False),
    ("v.proj+res", Mv, 768, 3072, True, 0, True), ("v.dproj*dgelu", Mv, 3072, 768, False, 2, False), ("v.dfc", Mv, 768, 3072, True, 0, False),
    ("v.dout", Mv, 768, 768, False, 0, False), ("v.dqkv", Mv, 768, 2304, True, 0, False),
    ("t.qkv", Mt, 1536, 512, False, 0, False), ("t.out+res", Mt, 512, 512, True, 0, True), ("t.fc+gelu", Mt, 2048, 512, False, 1, False),
    ("t.proj+res", Mt, 512, 2048, True, 0, True), ("t.dfc", Mt, 512, 2048, True, 0, False),
]
torch.manual_seed(0)
print(f"{'shape':16s} {'M':>6s} {'N':>5s} {'K':>5s} | {'128 us':>8s} {'TF':>7s} | {'256 us':>8s} {'TF':>7s}")
for name, M, N, K, f32out, epi, res in shapes:
    a = torch.randn(M, K, device=dev).to(TD)
    b = (torch.randn(N, K, device=dev) * 0.05).to(TD)
    c = torch.zeros(M, N, device=dev, dtype=torch.float32 if (f32out or dt == F32) else TD)
    bias = torch.randn(N, device=dev)
    r = torch.randn(M, N, device=dev) if res else None
    aux = torch.randn(M, N, device=dev).to(TD) if epi else None
    t = {}
    for rnd in range(3):
        for kern, key in (("128", 1 << 30), ("256", 1)):
            call("lpi_set_tuning", 0, key)
            for _ in range(2):
                E.gemm(dt, a, b, c, M, N, K, bias=bias, residual=r, epi=epi, aux=aux)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                E.gemm(dt, a, b, c, M, N, K, bias=bias, residual=r, epi=epi, aux=aux)
            e1.record()
            torch.cuda.synchronize()
            t.setdefault(kern, []).append(e0.elapsed_time(e1) / 10 * 1e3)
    fl = 2.0 * M * N * K
    u1, u2 = min(t["128"]), min(t["256"])
    print(f"{name:16s} {M:6d} {N:5d} {K:5d} | {u1:8.1f} {fl / u1 / 1e6:7.1f} | {u2:8.1f} {fl / u2 / 1e6:7.1f}")
call("lpi_set_tuning", 0, 1)
