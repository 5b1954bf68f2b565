#!/usr/bin/env python3
"""Phase timeline of workgroup 0 of the fourth-generation attention backward (diagnostic build -DLPI_ABL4_STAMPS, loaded through LPI_LIB):
per head 2 stamps (head start, after the prologue) and per iteration 8 (start, DMA issued, delta pass, first half, second half, dQ, vmcnt
wait, barrier).  Prints the mean cycles of every phase per wave."""
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
call("lpi_set_tuning", 7, 5)
for _ in range(5):
    call("lpi_attn_bwd", BF16, B, L, H, qkv, 3 * d, ctx, d, dctx, d, lse, delta, dqkv, 3 * d, 0, s())
torch.cuda.synchronize()
st = delta.reshape(-1)[:2 * 8 * 2048].view(torch.int64).reshape(8, 2048).cpu()
NSL, NH, PER_IT, PER_HEAD = 7, 12, 8, 2
names = ["issue", "delta", "half1", "half2", "dq", "vmwait", "barrier", "->next"]
for w in range(8):
    t = st[w]
    rows = []
    idx = 0
    head_pro, head_tot = [], []
    for h in range(NH):
        h0, h1 = int(t[idx]), int(t[idx + 1])
        idx += PER_HEAD
        head_pro.append(h1 - h0)
        for it in range(NSL):
            v = [int(x) for x in t[idx:idx + PER_IT]]
            idx += PER_IT
            nxt = int(t[idx]) if idx < 2048 and int(t[idx]) else v[-1]
            rows.append([v[i + 1] - v[i] for i in range(PER_IT - 1)] + [nxt - v[-1]])
    rows = torch.tensor(rows[NSL:-NSL], dtype=torch.float64)       # skip the first and the last head
    m = rows.mean(0)
    print(f"wave {w}: prologue {sum(head_pro[1:]) / (NH - 1):7.0f} | " + " ".join(f"{n} {x:6.0f}" for n, x in zip(names, m.tolist())) + f" | iteration {m.sum():7.0f} cycles (100 MHz ticks? see s_memtime)")
call("lpi_set_tuning", 7, 0)
