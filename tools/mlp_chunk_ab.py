#!/usr/bin/env python3
"""Does the MLP's hidden activation survive in the 256 MB Infinity Cache between its producer and its consumer when the batch is processed
in chunks?  fc+QuickGELU (saves u, writes g) followed by c_proj (+ fp16 residual) on M = 54 528 rows at once vs in 2 / 4 row chunks; the same
for the backward pair d c_proj * gelu'(u) -> d c_fc.   usage: python tools/mlp_chunk_ab.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lpi_amd import engine as E  # noqa: E402
from lpi_amd._lib import BF16  # noqa: E402

dev = "cuda:0"
M, d = 54528, 768
T = torch.bfloat16
torch.manual_seed(0)
h = torch.randn(M, d, device=dev).to(T)
wfc = (torch.randn(4 * d, d, device=dev) * 0.05).to(T)
wpr = (torch.randn(d, 4 * d, device=dev) * 0.05).to(T)
bfc, bpr = torch.randn(4 * d, device=dev), torch.randn(d, device=dev)
u = torch.zeros(M, 4 * d, device=dev, dtype=T)
g = torch.zeros(M, 4 * d, device=dev, dtype=T)
xmid = torch.randn(M, d, device=dev).half()
xout = torch.zeros(M, d, device=dev, dtype=torch.float16)
dx = torch.randn(M, d, device=dev).to(T)
du = torch.zeros(M, 4 * d, device=dev, dtype=T)
dh = torch.zeros(M, d, device=dev, dtype=T)
wprt, wfct = wpr.t().contiguous(), wfc.t().contiguous()
big = torch.zeros(600 << 20, device=dev, dtype=torch.uint8)      # cache flusher


def fwd(chunks):
    step = M // chunks // 256 * 256
    r = 0
    while r < M:
        n = min(step, M - r) if r + 2 * step <= M else M - r
        E.gemm(BF16, h[r:r + n], wfc, g[r:r + n], n, 4 * d, d, bias=bfc, epi=E.EPI_QUICKGELU, aux=u[r:r + n])
        E.gemm(BF16, g[r:r + n], wpr, xout[r:r + n], n, d, 4 * d, bias=bpr, residual=xmid[r:r + n])
        r += n


def bwd(chunks):
    step = M // chunks // 256 * 256
    r = 0
    while r < M:
        n = min(step, M - r) if r + 2 * step <= M else M - r
        E.gemm(BF16, dx[r:r + n], wprt, du[r:r + n], n, 4 * d, d, epi=E.EPI_DQUICKGELU, aux=u[r:r + n])
        E.gemm(BF16, du[r:r + n], wfct, dh[r:r + n], n, d, 4 * d)
        r += n


for name, fn in (("fc+gelu -> proj+res", fwd), ("dproj*gelu' -> dfc", bwd)):
    out = []
    for chunks in (1, 2, 3, 4, 6):
        best = 1e9
        for rep in range(4):
            fn(chunks)
            big.fill_(1)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                fn(chunks)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 5 * 1e3)
        out.append(f"{chunks} chunk(s): {best:6.1f} us")
    print(f"{name:22s} " + " | ".join(out))
