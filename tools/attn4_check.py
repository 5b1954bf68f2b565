#!/usr/bin/env python3
"""Fourth-generation attention backward (attention4.hip) against an f64 reference, and timed against the second generation."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lpi_amd._lib import BF16, F16, call  # noqa: E402

dev = "cuda:0"
s = lambda: torch.cuda.current_stream().cuda_stream  # noqa: E731


def ref(qkv, dctx, B, L, H):
    d = H * 64
    qr = qkv.double().requires_grad_(True)
    q, k, v = qr.reshape(B, L, 3, H, 64).permute(2, 0, 3, 1, 4)
    sc = (q * 0.125) @ k.transpose(-1, -2)
    p = torch.softmax(sc, -1)
    o = (p @ v).transpose(1, 2).reshape(B * L, d)
    o.backward(dctx.double())
    return o.detach(), torch.logsumexp(sc, -1).detach(), qr.grad


def relerr(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / b.abs().max())


ok = True
for dt, TD in ((BF16, torch.bfloat16), (F16, torch.float16)):
    for B, L, H, cap in ((2, 213, 3, 0), (2, 213, 3, 1), (3, 213, 2, 2), (1, 197, 1, 0), (2, 21, 2, 1), (2, 32, 1, 0), (2, 100, 2, 1), (1, 224, 2, 0), (2, 161, 1, 1)):
        d = H * 64
        g = torch.Generator().manual_seed(5)
        qkv = torch.randn(B * L, 3 * d, generator=g).to(TD)
        dctx = torch.randn(B * L, d, generator=g).to(torch.bfloat16)
        o, lse_ref, gref = ref(qkv.float(), dctx.float(), B, L, H)
        qd, dd = qkv.to(dev), dctx.to(dev)
        ctx = torch.zeros(B * L, d, device=dev, dtype=TD)
        lse = torch.zeros(B, H, L, device=dev)
        call("lpi_attn_fwd", dt, B, L, H, qd, 3 * d, ctx, d, lse, 0, s())
        out = {}
        for key in (1, 5):
            call("lpi_set_tuning", 7, key)
            call("lpi_set_tuning", 11, cap)
            dqkv = torch.full((B * L, 3 * d), float("nan"), device=dev, dtype=torch.bfloat16)
            delta = torch.zeros(B, H, L, device=dev)
            call("lpi_attn_bwd", dt, B, L, H, qd, 3 * d, ctx, d, dd, d, lse, delta, dqkv, 3 * d, 0, s())
            torch.cuda.synchronize()
            out[key] = (dqkv.clone(), delta.clone())
        call("lpi_set_tuning", 7, 0)
        call("lpi_set_tuning", 11, 0)
        e = {k: [relerr(v[0][:, i * d:(i + 1) * d], gref[:, i * d:(i + 1) * d]) for i in range(3)] for k, v in out.items()}
        dref = (dctx.double() * o).reshape(B, L, H, 64).sum(-1).permute(0, 2, 1)
        ed = relerr(out[1][1], dref)          # the two-pass kernels leave delta in the scratch; the streamed one keeps it in LDS
        good = max(e[5]) < 4e-2 and ed < 2e-2 and bool(torch.isfinite(out[5][0].float()).all())
        ok &= good
        print(f"dt={dt} B={B} L={L} H={H} cap={cap}: gen4 dq/dk/dv err {e[5][0]:.2e} {e[5][1]:.2e} {e[5][2]:.2e} delta {ed:.2e} | gen1 {e[1][0]:.2e} {e[1][1]:.2e} {e[1][2]:.2e}  {'ok' if good else 'FAIL'}", flush=True)
print("ALL OK" if ok else "FAILED", flush=True)

# timing at the benchmarked shape
B, L, H = 256, 213, 12
d = H * 64
qkv = torch.randn(B * L, 3 * d, device=dev).to(torch.bfloat16)
dctx = torch.randn(B * L, d, device=dev).to(torch.bfloat16)
ctx = torch.zeros(B * L, d, device=dev, dtype=torch.bfloat16)
dqkv = torch.zeros(B * L, 3 * d, device=dev, dtype=torch.bfloat16)
lse = torch.zeros(B, H, L, device=dev)
delta = torch.zeros(B, H, L, device=dev)
call("lpi_attn_fwd", BF16, B, L, H, qkv, 3 * d, ctx, d, lse, 0, s())
res = {}
for key in (1, 5, 1, 5):
    call("lpi_set_tuning", 7, key)
    fn = lambda: call("lpi_attn_bwd", BF16, B, L, H, qkv, 3 * d, ctx, d, dctx, d, lse, delta, dqkv, 3 * d, 0, s())  # noqa: E731
    best = 1e9
    for _ in range(3):
        fn(); fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 50)
    print(f"key7={key}: bwd {best:7.1f} us", flush=True)
    if key == 5:
        r1 = dqkv.clone()
        fn(); torch.cuda.synchronize()
        print("   bitwise reproducible:", bool(torch.equal(r1, dqkv)), flush=True)
call("lpi_set_tuning", 7, 0)
