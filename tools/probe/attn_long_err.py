"""bf16 / f16 attention error against f64 autograd: the one-workgroup-per-head kernels (L = 273) beside the long-sequence kernels (L = 289 .. 586)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lpi_amd import _lib  # noqa: E402
from lpi_amd._lib import BF16, F16, F32, call  # noqa: E402

DEV = "cuda:0"
s = torch.cuda.current_stream().cuda_stream


def relerr(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / b.abs().max())


def rms(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt())


for dt, td, tg in ((BF16, torch.bfloat16, torch.bfloat16), (F16, torch.float16, torch.bfloat16), (F32, torch.float32, torch.float32)):
    for L in (273, 289, 586):
        B, H = 2, 4
        d = H * 64
        g = torch.Generator().manual_seed(1)
        qkv = torch.randn(B * L, 3 * d, generator=g).to(td)
        dctx = torch.randn(B * L, d, generator=g).to(tg)
        qd = qkv.to(DEV)
        ctx = torch.zeros(B * L, d, device=DEV, dtype=td)
        lse = torch.zeros(B, H, L, device=DEV)
        call("lpi_attn_fwd", dt, B, L, H, qd, 3 * d, ctx, d, lse, 0, s)
        qr = qkv.double().requires_grad_(True)
        q, k, v = qr.reshape(B, L, 3, H, 64).permute(2, 0, 3, 1, 4)
        p = torch.softmax((q * 0.125) @ k.transpose(-1, -2), -1)
        o = (p @ v).transpose(1, 2).reshape(B * L, d)
        o.backward(dctx.double())
        dqkv = torch.zeros(B * L, 3 * d, device=DEV, dtype=tg)
        delta = torch.zeros(B, H, L, device=DEV)
        call("lpi_attn_bwd", dt, B, L, H, qd, 3 * d, ctx, d, dctx.to(DEV), d, lse, delta, dqkv, 3 * d, 0, s)
        print(f"dtype {dt} L {L}: ctx max {relerr(ctx, o.detach()):.2e} rms {rms(ctx, o.detach()):.2e} | "
              + " | ".join(f"{n} max {relerr(dqkv[:, sl], qr.grad[:, sl]):.2e} rms {rms(dqkv[:, sl], qr.grad[:, sl]):.2e}"
                           for n, sl in (("dq", slice(0, d)), ("dk", slice(d, 2 * d)), ("dv", slice(2 * d, 3 * d)))))
