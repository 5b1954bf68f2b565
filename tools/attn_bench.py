#!/usr/bin/env python3
"""Attention micro-benchmark on the bench's shapes (random data): forward, backward (dq + dkv passes)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lpi_amd._lib import BF16, F32, call  # noqa: E402

dt = F32 if (len(sys.argv) > 1 and sys.argv[1] == "f32") else BF16
TD = torch.float32 if dt == F32 else torch.bfloat16
dev = "cuda:0"
s = lambda: torch.cuda.current_stream().cuda_stream  # noqa: E731
for name, B, L, H, causal in (("vision", 256, 213, 12, 0), ("text", 256, 77, 8, 1), ("text59", 256, 59, 8, 1)):
    d = H * 64
    qkv = torch.randn(B * L, 3 * d, device=dev).to(TD)
    dctx = torch.randn(B * L, d, device=dev).to(TD)
    ctx = torch.zeros(B * L, d, device=dev, dtype=TD)
    dqkv = torch.zeros(B * L, 3 * d, device=dev, dtype=TD)
    lse = torch.zeros(B, H, L, device=dev)
    delta = torch.zeros(B, H, L, device=dev)
    fwd = lambda: call("lpi_attn_fwd", dt, B, L, H, qkv, 3 * d, ctx, d, lse, causal, s())  # noqa: E731
    bwd = lambda: call("lpi_attn_bwd", dt, B, L, H, qkv, 3 * d, ctx, d, dctx, d, lse, delta, dqkv, 3 * d, causal, s())  # noqa: E731
    out = []
    for fn, fl, key3, key7 in ((fwd, 4.0 * L * L * 64 * H * B, 0, 1), (bwd, 8.0 * L * L * 64 * H * B, 0, 1), (bwd, 8.0 * L * L * 64 * H * B, 1, 1),
                               (bwd, 8.0 * L * L * 64 * H * B, 0, 0)):
        call("lpi_set_tuning", 3, key3)
        call("lpi_set_tuning", 7, key7)
        best = 1e9
        for _ in range(3):
            fn(); fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fn()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 100)
        out.append(f"{best:7.1f} us {fl / best / 1e6:6.1f} TF")
    call("lpi_set_tuning", 3, 0)
    call("lpi_set_tuning", 7, 0)
    print(f"{name:7s} L={L:3d} attention.hip: fwd {out[0]} | bwd fused {out[1]} | bwd two-pass {out[2]}  (dense algorithmic FLOPs)")
    print(f"{name:7s} L={L:3d} default dispatch (the streamed single-pass backward of attention4.hip where it applies): bwd {out[3]}")
