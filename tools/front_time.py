#!/usr/bin/env python3
"""The vision front end (lpi_vis_assemble_fwd) with and without the output-row statistics, and the statistics pass they replace; medians, us."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lpi_amd._lib import BF16, F16, call  # noqa: E402

dev = "cuda:0"
B, G2, P, d = 256, 196, 16, 768
L = 1 + P + G2
g = torch.Generator().manual_seed(0)
pe = torch.randn(B * G2, d, generator=g).to(dev)
cls, pos, pr0 = torch.randn(d, generator=g).to(dev), torch.randn(1 + G2, d, generator=g).to(dev), torch.randn(P, d, generator=g).to(dev)
gam, bet = torch.ones(d, device=dev), torch.zeros(d, device=dev)
x0 = torch.zeros(B * L, d, dtype=torch.float16, device=dev)
st, so = torch.zeros(2, B * L, device=dev), torch.zeros(2, B * L, device=dev)
s = torch.cuda.current_stream().cuda_stream
fns = {
    "assemble": lambda: call("lpi_vis_assemble_fwd", F16, B, G2, P, d, pe, d, cls, pos, pr0, 0, gam, bet, x0, st[0], st[1], None, None, s),
    "assemble + output statistics": lambda: call("lpi_vis_assemble_fwd", F16, B, G2, P, d, pe, d, cls, pos, pr0, 0, gam, bet, x0, st[0], st[1], so[0], so[1], s),
    "statistics pass": lambda: call("lpi_layernorm_fwd", BF16, F16, B * L, d, x0, d, None, None, None, 0, so[0], so[1], s),
}
acc = {k: [] for k in fns}
for rep in range(6):
    for k, f in fns.items():
        f()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(8)]
        for a, b in ev:
            a.record(); f(); b.record()
        torch.cuda.synchronize()
        if rep:
            acc[k] += [a.elapsed_time(b) * 1e3 for a, b in ev]
print("  ".join(f"{k}: {statistics.median(v):.1f} us" for k, v in acc.items()))
