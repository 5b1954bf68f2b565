#!/usr/bin/env python3
"""bf16 vs f16 operands on the same GEMM / attention shapes, interleaved in one process (random data): is the fp16 MFMA as fast?"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lpi_amd import engine as E  # noqa: E402
from lpi_amd._lib import BF16, F16, call  # noqa: E402

dev = "cuda:0"
TD = {BF16: torch.bfloat16, F16: torch.float16}
s = lambda: torch.cuda.current_stream().cuda_stream  # noqa: E731


def timeit(fn, n=10):
    fn(); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


torch.manual_seed(0)
Mv = 54528
for name, M, N, K, epi in (("v.qkv", Mv, 2304, 768, 0), ("v.dfc-like", Mv, 768, 3072, 0), ("v.fc+gelu", Mv, 3072, 768, 1)):
    res = {}
    ops = {}
    for dt in (BF16, F16):
        a = torch.randn(M, K, device=dev).to(TD[dt])
        b = (torch.randn(N, K, device=dev) * 0.05).to(TD[dt])
        c = torch.zeros(M, N, device=dev, dtype=TD[dt])
        bias = torch.randn(N, device=dev)
        aux = torch.zeros(M, N, device=dev, dtype=torch.bfloat16) if epi else None
        ops[dt] = (a, b, c, bias, aux)
    for rnd in range(3):
        for dt in (BF16, F16):
            a, b, c, bias, aux = ops[dt]
            t = timeit(lambda: E.gemm(dt, a, b, c, M, N, K, bias=bias, epi=epi, aux=aux))
            res.setdefault(dt, []).append(t)
    print(f"{name:12s} bf16 {min(res[BF16]):7.1f} us   f16 {min(res[F16]):7.1f} us   ratio {min(res[F16]) / min(res[BF16]):.3f}")
B, L, H = 256, 213, 12
d = H * 64
res = {}
ops = {}
for dt in (BF16, F16):
    qkv = torch.randn(B * L, 3 * d, device=dev).to(TD[dt])
    ctx = torch.zeros(B * L, d, device=dev, dtype=TD[dt])
    lse = torch.zeros(B, H, L, device=dev)
    dctx = torch.randn(B * L, d, device=dev).to(torch.bfloat16)
    dqkv = torch.zeros(B * L, 3 * d, device=dev, dtype=torch.bfloat16)
    delta = torch.zeros(B, H, L, device=dev)
    ops[dt] = (qkv, ctx, lse, dctx, dqkv, delta)
for rnd in range(3):
    for dt in (BF16, F16):
        qkv, ctx, lse, dctx, dqkv, delta = ops[dt]
        tf = timeit(lambda: call("lpi_attn_fwd", dt, B, L, H, qkv, 3 * d, ctx, d, lse, 0, s()))
        tb = timeit(lambda: call("lpi_attn_bwd", dt, B, L, H, qkv, 3 * d, ctx, d, dctx, d, lse, delta, dqkv, 3 * d, 0, s()))
        res.setdefault(dt, []).append((tf, tb))
for i, nm in ((0, "attn fwd"), (1, "attn bwd")):
    b_, f_ = min(r[i] for r in res[BF16]), min(r[i] for r in res[F16])
    print(f"{nm:12s} bf16 {b_:7.1f} us   f16 {f_:7.1f} us   ratio {f_ / b_:.3f}")
