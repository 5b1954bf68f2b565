#!/usr/bin/env python3
"""Sweep a tuning key of the 256x256 GEMM over the bench's vision shapes: python tools/gemm_sweep.py <key> v1 v2 ..."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lpi_amd import engine as E
from lpi_amd._lib import BF16, call
key = int(sys.argv[1]); vals = [int(v) for v in sys.argv[2:]]
dev = "cuda:0"; Mv = 54528
shapes = [("v.qkv", Mv, 2304, 768, False, 0, False), ("v.out+res", Mv, 768, 768, True, 0, True), ("v.fc+gelu", Mv, 3072, 768, False, 1, False),
          ("v.proj+res", Mv, 768, 3072, True, 0, True), ("v.dproj", Mv, 3072, 768, False, 2, False), ("v.dfc", Mv, 768, 3072, False, 0, False),
          ("v.dout", Mv, 768, 768, False, 0, False), ("v.dqkv", Mv, 768, 2304, False, 0, False)]
tot = {v: 0.0 for v in vals}
for name, M, N, K, f32o, epi, res in shapes:
    a = torch.randn(M, K, device=dev).bfloat16(); b = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    c = torch.zeros(M, N, device=dev, dtype=torch.float32 if f32o else torch.bfloat16); bias = torch.randn(N, device=dev)
    r = torch.randn(M, N, device=dev) if res else None; aux = torch.randn(M, N, device=dev).bfloat16() if epi else None
    out = []
    for v in vals:
        call("lpi_set_tuning", key, v)
        best = 1e9
        for rep in range(3):
            for _ in range(2): E.gemm(BF16, a, b, c, M, N, K, bias=bias, residual=r, epi=epi, aux=aux)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
            for _ in range(10): E.gemm(BF16, a, b, c, M, N, K, bias=bias, residual=r, epi=epi, aux=aux)
            e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1) * 100)
        out.append(f"{v}:{best:.0f}"); tot[v] += best
    print(f"{name:12s}", " ".join(out))
print("sum         ", " ".join(f"{v}:{t:.0f}" for v, t in tot.items()))
