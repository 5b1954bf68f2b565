#!/usr/bin/env python3
"""MLP in row chunks, second look (round 4): does the hidden activation survive in the 256 MB Infinity Cache between fc + QuickGELU and c_proj when the
rows are processed in chunks whose TILE COUNTS are kept friendly (round 2 measured equal chunks only: 3 x 71 row panels gained 6 % although the c_proj
chunks ran 213 tiles = 0.83 of a round each)?  Splits are given in row panels of 256 rows (213 panels = the vision tower at 256 pairs).
usage: python tools/mlp_chunk_ab2.py"""
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


def rows(split):
    r = 0
    for p in split:
        yield r, p * 256
        r += p * 256
    assert r == M, (r, M)


def fwd(split):
    for r, n in rows(split):
        E.gemm(BF16, h[r:r + n], wfc, g[r:r + n], n, 4 * d, d, bias=bfc, epi=E.EPI_QUICKGELU, aux=u[r:r + n])
        E.gemm(BF16, g[r:r + n], wpr, xout[r:r + n], n, d, 4 * d, bias=bpr, residual=xmid[r:r + n])


def bwd(split):
    for r, n in rows(split):
        E.gemm(BF16, dx[r:r + n], wprt, du[r:r + n], n, 4 * d, d, epi=E.EPI_DQUICKGELU, aux=u[r:r + n])
        E.gemm(BF16, du[r:r + n], wfct, dh[r:r + n], n, d, 4 * d)


SPLITS = [(213,), (107, 106), (71, 71, 71), (85, 86, 42), (85, 85, 43), (42, 86, 85), (64, 64, 85), (128, 85), (85, 128), (43, 42, 43, 42, 43), (53, 53, 53, 54)]
for name, fn in (("fc+gelu -> proj+res", fwd), ("dproj*gelu' -> dfc", bwd)):
    for split in SPLITS:
        ts = []
        for rep in range(4):
            fn(split)
            big.fill_(1)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                fn(split)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 5 * 1e3)
        ts.sort()
        print(f"{name:22s} {str(split):28s} best {ts[0]:7.1f} us  median {ts[len(ts) // 2]:7.1f} us", flush=True)
