#!/usr/bin/env python3
"""The text tower's attention backward (fused one-head kernel, H = 8) on the bench's caption lengths: plain packed layout and shared-prefix layout, by
tuning key 9 (minimum waves per workgroup on ragged batches).   python tools/text_attn_bench.py [key9 values ...]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lpi_amd import synth  # noqa: E402
from lpi_amd._lib import BF16, call  # noqa: E402
from lpi_amd.engine import PackedIds  # noqa: E402

dev = "cuda:0"
B, H, d = 256, 8, 512
s = lambda: torch.cuda.current_stream().cuda_stream  # noqa: E731
ids = synth.token_ids(B)
keys = [int(x) for x in sys.argv[1:]] or [0, 3, 4, 6, 8]


def timed(fn, n=20):
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


for name, pre in (("plain", 0), ("shared", 17)):
    pk = PackedIds(ids, pre).to(dev)
    M, L = pk.rows, pk.shape[1]
    Mp = (M + 255) // 256 * 256
    qkv = torch.randn(Mp, 3 * d, device=dev).bfloat16()
    dctx = torch.randn(Mp, d, device=dev).bfloat16()
    ctx = torch.zeros(Mp, d, device=dev, dtype=torch.bfloat16)
    dqkv = torch.zeros(Mp, 3 * d, device=dev, dtype=torch.bfloat16)
    lse, delta = torch.zeros(B + 1, H, L, device=dev), torch.zeros(B + 1, H, L, device=dev)
    scratch = torch.zeros(B * 17 * 2 * d, device=dev)
    rs = pk.row_start_dev
    if pre:
        fwd = lambda: call("lpi_attn_fwd_shared", BF16, B, L, rs, pre, H, qkv, 3 * d, ctx, d, lse, s())  # noqa: E731
        bwd = lambda: call("lpi_attn_bwd_shared", BF16, B, L, rs, pre, L, H, qkv, 3 * d, ctx, d, dctx, d, lse, delta, dqkv, 3 * d, scratch, s())  # noqa: E731
        bw0 = lambda: call("lpi_attn_bwd_shared", BF16, B, L, rs, pre, pre, H, qkv, 3 * d, ctx, d, dctx, d, lse, delta, dqkv, 3 * d, scratch, s())  # noqa: E731
    else:
        fwd = lambda: call("lpi_attn_fwd_varlen", BF16, B, L, rs, H, qkv, 3 * d, ctx, d, lse, 1, s())  # noqa: E731
        bwd = lambda: call("lpi_attn_bwd_varlen", BF16, B, L, rs, H, qkv, 3 * d, ctx, d, dctx, d, lse, delta, dqkv, 3 * d, 1, s())  # noqa: E731
        bw0 = lambda: call("lpi_attn_bwd_prefix", BF16, B, L, rs, 17, H, qkv, 3 * d, ctx, d, dctx, d, lse, delta, dqkv, 3 * d, 1, s())  # noqa: E731
    fwd()
    for k9 in keys:
        call("lpi_set_tuning", 9, k9)
        print(f"{name:>6} rows {M:6d} L {L}  key9 {k9}: fwd {timed(fwd):6.1f} us   bwd {timed(bwd):6.1f} us   first-block bwd {timed(bw0):6.1f} us", flush=True)
call("lpi_set_tuning", 9, 0)
