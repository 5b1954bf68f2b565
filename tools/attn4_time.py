#!/usr/bin/env python3
"""Time the vision attention backward (B 256, L 213, H 12, bf16) of the library named by LPI_LIB: tuning key 7 values from argv (default 1 5)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lpi_amd._lib import BF16, call  # noqa: E402

dev = "cuda:0"
s = lambda: torch.cuda.current_stream().cuda_stream  # noqa: E731
B, L, H = 256, 213, 12
d = H * 64
qkv = torch.randn(B * L, 3 * d, device=dev).to(torch.bfloat16)
dctx = torch.randn(B * L, d, device=dev).to(torch.bfloat16)
ctx = torch.zeros(B * L, d, device=dev, dtype=torch.bfloat16)
dqkv = torch.zeros(B * L, 3 * d, device=dev, dtype=torch.bfloat16)
lse = torch.zeros(B, H, L, device=dev)
delta = torch.zeros(B, H, L, device=dev)
call("lpi_attn_fwd", BF16, B, L, H, qkv, 3 * d, ctx, d, lse, 0, s())
keys = [a for a in sys.argv[1:]] or ["1", "5"]
fn = lambda: call("lpi_attn_bwd", BF16, B, L, H, qkv, 3 * d, ctx, d, dctx, d, lse, delta, dqkv, 3 * d, 0, s())  # noqa: E731
times = {k: [] for k in keys}
for rnd in range(9):          # configurations interleaved, so that clock / box drift hits them alike
    for key in keys:
        k7, _, k12 = key.partition(":")       # "5:1" = generation 5 with A/B flags (tuning key 12) = 1
        call("lpi_set_tuning", 7, int(k7))
        call("lpi_set_tuning", 12, int(k12 or 0))
        fn(); fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            fn()
        e1.record()
        torch.cuda.synchronize()
        if rnd:
            times[key].append(e0.elapsed_time(e1) * 1000 / 30)
for key in keys:
    t = sorted(times[key])
    print(f"{os.environ.get('LPI_LIB', 'base'):40s} key7={key}: bwd median {t[len(t) // 2]:7.1f} us  min {t[0]:7.1f}  max {t[-1]:7.1f}", flush=True)
call("lpi_set_tuning", 7, 0)
call("lpi_set_tuning", 12, 0)
