#!/usr/bin/env python3
"""A/B a tuning knob of the GEMM on the bench's shapes, interleaved in one process on random data.
usage: python tools/gemm_ab.py KEY VALUE_A VALUE_B [bf16|f32]      (e.g. 2 -1 0: L2 warm-up of the next tile off / on)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lpi_amd import engine as E  # noqa: E402
from lpi_amd._lib import BF16, F32, call  # noqa: E402

key, va, vb = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dt = F32 if (len(sys.argv) > 4 and sys.argv[4] == "f32") else BF16
TD = torch.float32 if dt == F32 else torch.bfloat16
dev = "cuda:0"
Mv, Mt = 54528, 15104
shapes = [  # (name, M, N, K, c dtype, epi, residual)
    ("v.qkv", Mv, 2304, 768, TD, 0, False), ("v.out+res", Mv, 768, 768, torch.float16, 0, True), ("v.fc+gelu", Mv, 3072, 768, TD, 1, False),
    ("v.proj+res", Mv, 768, 3072, torch.float16, 0, True), ("v.dproj*dgelu", Mv, 3072, 768, TD, 2, False), ("v.dfc", Mv, 768, 3072, TD, 0, False),
    ("v.dout", Mv, 768, 768, TD, 0, False), ("v.dqkv", Mv, 768, 2304, TD, 0, False),
    ("t.qkv", Mt, 1536, 512, TD, 0, False), ("t.fc+gelu", Mt, 2048, 512, TD, 1, False), ("t.dproj", Mt, 2048, 512, TD, 2, False),
]
torch.manual_seed(0)
tot = {va: 0.0, vb: 0.0}
print(f"{'shape':16s} {'M':>6s} {'N':>5s} {'K':>5s} | key {key}={va:>3d} us {'TF':>7s} | key {key}={vb:>3d} us {'TF':>7s} | ratio")
for name, M, N, K, cdt, epi, res in shapes:
    if dt == F32:
        cdt = torch.float32
    a = torch.randn(M, K, device=dev).to(TD)
    b = (torch.randn(N, K, device=dev) * 0.05).to(TD)
    c = torch.zeros(M, N, device=dev, dtype=cdt)
    bias = torch.randn(N, device=dev)
    r = (torch.randn(M, N, device=dev).to(cdt) if cdt == torch.float16 else torch.randn(M, N, device=dev)) if res else None
    aux = torch.randn(M, N, device=dev).to(TD) if epi else None
    t = {va: [], vb: []}
    for rnd in range(4):
        for v in (va, vb):
            call("lpi_set_tuning", key, v)
            for _ in range(2):
                E.gemm(dt, a, b, c, M, N, K, bias=bias, residual=r, epi=epi, aux=aux)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                E.gemm(dt, a, b, c, M, N, K, bias=bias, residual=r, epi=epi, aux=aux)
            e1.record()
            torch.cuda.synchronize()
            t[v].append(e0.elapsed_time(e1) / 10 * 1e3)
    fl = 2.0 * M * N * K
    ua, ub = min(t[va]), min(t[vb])
    tot[va] += ua
    tot[vb] += ub
    print(f"{name:16s} {M:6d} {N:5d} {K:5d} | {ua:12.1f} {fl / ua / 1e6:7.1f} | {ub:12.1f} {fl / ub / 1e6:7.1f} | {ub / ua:5.3f}")
print(f"sum: {tot[va]:.1f} us vs {tot[vb]:.1f} us  ({tot[vb] / tot[va]:.4f})")
call("lpi_set_tuning", key, 0)
