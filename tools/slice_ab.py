#!/usr/bin/env python3
"""Weight slices of the persistent GEMM (tuning key 15) on the three wide-N shapes of the ViT-B/16 step, A/B interleaved: in_proj with the LayerNorm fold
(N = 2304: 3 slices), c_fc with LayerNorm fold + QuickGELU + saved derivative (N = 3072: 2 slices), d c_proj x gelu' (N = 3072: 2 slices); each alone and
grouped with the text tower's problem of the same op, as the step launches them.  Median of REPS launches per arm and round, microseconds."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lpi_amd import _lib, engine as E  # noqa: E402
from lpi_amd._lib import BF16, F16, call  # noqa: E402

dev = "cuda:0"
REPS, ROUNDS = 20, 3
Mv, Mt = 54528, 10240
torch.manual_seed(0)


def operands(M, N, K, kind):
    if kind in ("qkv", "fc"):        # LayerNorm-fold GEMMs: fp16 stream x fp16 weights, LN operand block
        a = torch.randn(M, K, device=dev).half()
        b = (torch.randn(N, K, device=dev) * 0.05).bfloat16().half()
        blk = torch.zeros(3 * M + N, device=dev)
        blk[:M] = a.float().mean(1)
        blk[M:2 * M] = 1.0 / (a.float().var(1, unbiased=False) + 1e-5).sqrt()
        blk[2 * M:2 * M + N] = b.float().sum(1)
        c = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
        aux = torch.zeros(M, N, device=dev, dtype=torch.bfloat16) if kind == "fc" else None
        return dict(M=M, N=N, K=K, a=a, b=b, c=c, bias=torch.randn(N, device=dev), residual=blk, ldr=M, aux=aux), F16, (E.EPI_LN_QUICKGELU if kind == "fc" else E.EPI_LN)
    a = torch.randn(M, K, device=dev).bfloat16()
    b = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    c = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
    aux = torch.randn(M, N, device=dev).bfloat16()
    return dict(M=M, N=N, K=K, a=a, b=b, c=c, aux=aux), BF16, E.EPI_DQUICKGELU


def timed(fn):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(REPS)]
    for e0, e1 in ev:
        e0.record()
        fn()
        e1.record()
    torch.cuda.synchronize()
    return float(np.median([e0.elapsed_time(e1) for e0, e1 in ev])) * 1e3


s = torch.cuda.current_stream().cuda_stream
for kind, (Nv, Kv), (Nt, Kt) in (("qkv", (2304, 768), (1536, 512)), ("fc", (3072, 768), (2048, 512)), ("dproj", (3072, 768), (2048, 512))):
    pv, dt, epi = operands(Mv, Nv, Kv, kind)
    pt, _, _ = operands(Mt, Nt, Kt, kind)
    cdt = BF16
    runs = {"alone": lambda: _lib.gemm_grouped(dt, cdt, epi, 1.0, [pv], s), "grouped with text": lambda: _lib.gemm_grouped(dt, cdt, epi, 1.0, [pv, pt], s)}
    for name, fn in runs.items():
        res = {-1: [], 0: []}
        ref = None
        for _ in range(ROUNDS):
            for key in (-1, 0):
                call("lpi_set_tuning", 15, key)
                fn()
                torch.cuda.synchronize()
                if ref is None:
                    ref = pv["c"].clone()
                else:
                    assert torch.equal(ref.view(torch.int16), pv["c"].view(torch.int16)), "slices changed bits"
                res[key].append(timed(fn))
        call("lpi_set_tuning", 15, 0)
        off, on = np.median(res[-1]), np.median(res[0])
        print(f"{kind:6s} {name:18s} off {off:7.1f} us  slices {on:7.1f} us  ({100 * (on / off - 1):+.1f} %)   rounds off {[round(x, 1) for x in res[-1]]} on {[round(x, 1) for x in res[0]]}", flush=True)
