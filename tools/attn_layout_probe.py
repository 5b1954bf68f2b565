#!/usr/bin/env python3
"""Does the attention of the vision tower run faster on a HEAD-CONTIGUOUS q/k/v layout?  The kernels read a (sample, head) slice as 213 pieces of 128 bytes at a
stride of 4 608 bytes (qkv is [M, 3 d], heads interleaved in the columns).  Proxy without touching the kernels: the same arithmetic as B' = B H samples of ONE head
(qkv' [B H L, 192]: a slice is 213 rows of 384 contiguous bytes).  python tools/attn_layout_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lpi_amd._lib import BF16, call  # noqa: E402

dev = "cuda:0"
s = lambda: torch.cuda.current_stream().cuda_stream  # noqa: E731


def timed(fn, n=10):
    best = 1e9
    for _ in range(3):
        fn(); fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / n)
    return best


for name, B, L, H, causal in (("vision: heads interleaved", 256, 213, 12, 0), ("vision: head-contiguous proxy", 256 * 12, 213, 1, 0),
                              ("text: heads interleaved", 256, 59, 8, 1), ("text: head-contiguous proxy", 256 * 8, 59, 1, 1)):
    d = H * 64
    M = B * L
    Mp = (M + 255) // 256 * 256
    qkv = torch.randn(Mp, 3 * d, device=dev).bfloat16()
    dctx = torch.randn(Mp, d, device=dev).bfloat16()
    ctx = torch.zeros(Mp, d, device=dev, dtype=torch.bfloat16)
    dqkv = torch.zeros(Mp, 3 * d, device=dev, dtype=torch.bfloat16)
    lse, delta = torch.zeros(B, H, L, device=dev), torch.zeros(B, H, L, device=dev)
    fwd = lambda: call("lpi_attn_fwd", BF16, B, L, H, qkv, 3 * d, ctx, d, lse, causal, s())  # noqa: E731
    bwd = lambda: call("lpi_attn_bwd", BF16, B, L, H, qkv, 3 * d, ctx, d, dctx, d, lse, delta, dqkv, 3 * d, causal, s())  # noqa: E731
    fwd()
    byt = M * d * 2 * (4 + 0)      # fwd: q, k, v in, ctx out
    tf, tb = timed(fwd), timed(bwd)
    print(f"{name:>32}: fwd {tf:6.1f} us ({M * d * 2 * 4 / tf / 1e6:5.2f} TB/s)   bwd {tb:6.1f} us ({M * d * 2 * 8 / tb / 1e6:5.2f} TB/s)", flush=True)
