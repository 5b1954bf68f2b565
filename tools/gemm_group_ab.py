#!/usr/bin/env python3
"""Grouped (vision + text problem of the same layer op in ONE persistent launch, lpi_gemm_nt_grouped) vs two separate launches, on
the bench's shapes, interleaved in one process on random data.   usage: python tools/gemm_group_ab.py [text_rows_per_sample]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lpi_amd import _lib  # noqa: E402
from lpi_amd._lib import BF16, F16, call  # noqa: E402

dev = "cuda:0"
Lt = int(sys.argv[1]) if len(sys.argv) > 1 else 59
Mv, Mt = 54528, (256 * Lt + 255) // 256 * 256
TD = torch.bfloat16
ops = [  # (name, Nv, Kv, Nt, Kt, c dtype, epi, residual)
    ("qkv", 2304, 768, 1536, 512, TD, 0, False), ("out+res", 768, 768, 512, 512, torch.float16, 0, True),
    ("fc+gelu", 3072, 768, 2048, 512, TD, 1, False), ("proj+res", 768, 3072, 512, 2048, torch.float16, 0, True),
    ("dproj*dgelu", 3072, 768, 2048, 512, TD, 2, False), ("dfc", 768, 3072, 512, 2048, TD, 0, False),
    ("dout", 768, 768, 512, 512, TD, 0, False), ("dqkv", 768, 2304, 512, 1536, TD, 0, False),
]
torch.manual_seed(0)
s = torch.cuda.current_stream().cuda_stream
tot = [0.0, 0.0]
print(f"{'op':12s} | separate us (TF) | grouped us (TF) | ratio")
for name, Nv, Kv, Nt, Kt, cdt, epi, res in ops:
    probs = []
    for M, N, K in ((Mv, Nv, Kv), (Mt, Nt, Kt)):
        p = dict(M=M, N=N, K=K, a=torch.randn(M, K, device=dev).to(TD), b=(torch.randn(N, K, device=dev) * 0.05).to(TD),
                 c=torch.zeros(M, N, device=dev, dtype=cdt), bias=None if epi == 2 else torch.randn(N, device=dev))
        if res:
            p["residual"] = torch.randn(M, N, device=dev).to(cdt)
        if epi:
            p["aux"] = torch.randn(M, N, device=dev).to(TD)
        probs.append(p)
    c_dt = F16 if cdt == torch.float16 else BF16
    t = [[], []]
    for rnd in range(4):
        for mode in (0, 1):
            call("lpi_set_tuning", 8, 1 - mode)
            for _ in range(2):
                _lib.gemm_grouped(BF16, c_dt, epi, 1.0, probs, s)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                g = _lib.gemm_grouped(BF16, c_dt, epi, 1.0, probs, s)
            e1.record()
            torch.cuda.synchronize()
            assert g == bool(mode)
            t[mode].append(e0.elapsed_time(e1) / 10 * 1e3)
    fl = sum(2.0 * p["M"] * p["N"] * p["K"] for p in probs)
    a, b = min(t[0]), min(t[1])
    tot[0] += a
    tot[1] += b
    print(f"{name:12s} | {a:8.1f} ({fl / a / 1e6:6.1f}) | {b:8.1f} ({fl / b / 1e6:6.1f}) | {b / a:5.3f}")
print(f"sum per layer: separate {tot[0]:.1f} us, grouped {tot[1]:.1f} us ({tot[1] / tot[0]:.4f})")
call("lpi_set_tuning", 8, 0)
