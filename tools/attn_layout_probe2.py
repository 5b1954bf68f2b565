#!/usr/bin/env python3
"""The vision tower's attention kernels THEMSELVES on three q / k / v layouts (round 6) — not a proxy: lpi_attn_fwd_pair with layout strides
and lpi_attn_bwd_layout (the streamed single-pass backward):
  interleaved  [M, 3 H 64]            the GEMMs' natural output (q | k | v, heads contiguous by 64): a (sample, head) slice = L pieces of 128 B at a stride of 6 H 64 B
  head-grouped [M, H, (q|k|v) 64]     the same matrix with in_proj's weight rows permuted (free: the weights are frozen): L pieces of 384 B
  blocked      [3 H][Mp][64] planes   every (sample, head) slice of q, k, v, ctx, dctx, dq, dk, dv one contiguous run of L x 128 B
Checks that the three give the same bits, then times forward and backward.  python tools/attn_layout_probe2.py [B]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lpi_amd import _lib  # noqa: E402
from lpi_amd._lib import BF16, call  # noqa: E402

dev = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
L, H = 213, 12
d = H * 64
M = B * L
Mp = (M + 255) // 256 * 256
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


g = torch.Generator(device=dev).manual_seed(1)
qkv_i = torch.randn(Mp, 3, H, 64, device=dev, generator=g).bfloat16()          # [row][which][head][c] = interleaved
dctx_i = torch.randn(Mp, H, 64, device=dev, generator=g).bfloat16()
# a small text problem beside it, as in the step (the pair launch)
Bt, Lt, Ht = B, 59, 8
qkv_t = torch.randn(Bt * Lt, 3 * Ht * 64, device=dev, generator=g).bfloat16()
ctx_t = torch.zeros(Bt * Lt, Ht * 64, device=dev, dtype=torch.bfloat16)
lse_t = torch.zeros(Bt, Ht, Lt, device=dev)

layouts = {
    "interleaved": dict(qkv=qkv_i.reshape(Mp, 3 * d).contiguous(), ld=3 * d, hs=64, vs=d, dctx=dctx_i.reshape(Mp, d).contiguous(), cld=d, chs=64),
    "head-grouped": dict(qkv=qkv_i.permute(0, 2, 1, 3).contiguous().reshape(Mp, 3 * d), ld=3 * d, hs=192, vs=64, dctx=dctx_i.reshape(Mp, d).contiguous(), cld=d, chs=64),
    "blocked": dict(qkv=qkv_i.permute(1, 2, 0, 3).contiguous().reshape(3 * H * Mp, 64), ld=64, hs=Mp * 64, vs=H * Mp * 64,
                    dctx=dctx_i.permute(1, 0, 2).contiguous().reshape(H * Mp, 64), cld=64, chs=Mp * 64),
}
res = {}
for name, lay in layouts.items():
    ctx = torch.zeros_like(lay["dctx"])
    dqkv = torch.zeros_like(lay["qkv"])
    lse, delta = torch.zeros(B, H, L, device=dev), torch.zeros(B, H, L, device=dev)
    va = (B, L, None, H, lay["qkv"], lay["ld"], ctx, lay["cld"], lse, 0, 0, (lay["hs"], lay["vs"], lay["chs"]))
    ta = (Bt, Lt, None, Ht, qkv_t, 3 * Ht * 64, ctx_t, Ht * 64, lse_t, 1, 0)
    fwd = lambda: _lib.attn_fwd_pair(BF16, va, ta, s())  # noqa: E731
    larr = (ctypes.c_int32 * 6)(lay["hs"], lay["vs"], lay["hs"], lay["vs"], lay["chs"], lay["chs"])
    bwd = lambda: call("lpi_attn_bwd_layout", BF16, B, L, L, H, lay["qkv"], lay["ld"], ctx, lay["cld"], lay["dctx"], lay["cld"], lse, delta, dqkv, lay["ld"],  # noqa: E731
                       ctypes.cast(larr, ctypes.c_void_p), s())
    fwd(); bwd()
    torch.cuda.synchronize()
    # back to [row][which][head][c] / [row][head][c] for the comparison
    if name == "interleaved":
        c_n, dq_n = ctx.reshape(Mp, H, 64), dqkv.reshape(Mp, 3, H, 64)
    elif name == "head-grouped":
        c_n, dq_n = ctx.reshape(Mp, H, 64), dqkv.reshape(Mp, H, 3, 64).permute(0, 2, 1, 3)
    else:
        c_n, dq_n = ctx.reshape(H, Mp, 64).permute(1, 0, 2), dqkv.reshape(3, H, Mp, 64).permute(2, 0, 1, 3)
    res[name] = (c_n[:M].clone(), dq_n[:M].clone(), lse.clone())
    tf, tb = timed(fwd), timed(bwd)
    # ... and AS THE STEP RUNS THEM: the forward right behind the GEMM that wrote qkv, the backward right behind the GEMM that wrote dctx (a device copy of
    # the same bytes stands in for the GEMM's stores: what matters is that the operand was just written, i.e. sits in the Infinity Cache), a 335 MB
    # stream of other traffic in between as the MLP GEMMs of a layer would leave; only the attention kernel is between the events
    src_q, src_d = lay["qkv"].clone(), lay["dctx"].clone()
    other = torch.empty(168 * 1024 * 1024, device=dev, dtype=torch.bfloat16)

    def step_like(kernel, refresh):
        ts = []
        for _ in range(12):
            other.add_(1)
            refresh()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); kernel(); e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        return sorted(ts[2:])[len(ts[2:]) // 2]
    sf = step_like(fwd, lambda: lay["qkv"].copy_(src_q))
    sb = step_like(bwd, lambda: lay["dctx"].copy_(src_d))
    print(f"{name:>13}: forward pair {tf:6.1f} us   streamed backward {tb:6.1f} us ({M * d * 2 * 8 / tb / 1e6:4.2f} TB/s)   |  behind the operand's producer: "
          f"forward {sf:6.1f} us   backward {sb:6.1f} us", flush=True)
    del src_q, src_d, other
ref = res["interleaved"]
for name, (c_n, dq_n, lse) in res.items():
    same = torch.equal(c_n, ref[0]) and torch.equal(dq_n, ref[1]) and torch.equal(lse, ref[2])
    print(f"{name:>13}: ctx / dqkv / lse bit-identical to the interleaved layout: {same}")
    assert same, name
