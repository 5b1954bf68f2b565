"""The long-sequence attention kernels alone at the ViT-L/14@336px shape (L = 593, H = 16), bf16: TFLOP/s of forward and backward.  LPI_LIB selects a build."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lpi_amd import _lib  # noqa: E402
from lpi_amd._lib import BF16, call  # noqa: E402

DEV = "cuda:0"
s = torch.cuda.current_stream().cuda_stream
B, L, H = 64, 593, 16
d = H * 64
qkv = torch.randn(B * L, 3 * d, device=DEV).bfloat16()
dctx = torch.randn(B * L, d, device=DEV).bfloat16()
ctx = torch.zeros(B * L, d, device=DEV, dtype=torch.bfloat16)
lse = torch.zeros(B, H, L, device=DEV)
delta = torch.zeros(B, H, L, device=DEV)
dqkv = torch.zeros(B * L, 3 * d, device=DEV, dtype=torch.bfloat16)


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


tf = timed(lambda: call("lpi_attn_fwd", BF16, B, L, H, qkv, 3 * d, ctx, d, lse, 0, s))
tb = timed(lambda: call("lpi_attn_bwd", BF16, B, L, H, qkv, 3 * d, ctx, d, dctx, d, lse, delta, dqkv, 3 * d, 0, s))
fl = 4.0 * L * L * 64 * H * B
print(f"{os.environ.get('LPI_LIB', 'default build')}: forward {tf:.0f} us = {fl / tf / 1e6:.0f} TFLOP/s, backward (dQ + dK/dV launches) {tb:.0f} us = {2.5 * fl / tb / 1e6:.0f} TFLOP/s")
