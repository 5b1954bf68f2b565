#!/usr/bin/env python3
"""The persistent 256x256 GEMM, one bench shape after the other (REPS launches each, fixed order), for rocprofv3 --pmc passes: the counter rows of
the gemm256p_kernel dispatches are attributed to shapes by dispatch order (tools/gemm_shape_pmc_summary.py).  Prints the shape table (name, M, N, K,
algorithmic read / write bytes) as JSON on the last line.   usage: python3 tools/gemm_shape_pmc.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lpi_amd import engine as E  # noqa: E402
from lpi_amd._lib import BF16, call  # noqa: E402

REPS = 8
TD = torch.bfloat16
dev = "cuda:0"
Mv, Mt = 54528, 10752          # vision rows (256 x 213) and packed text rows (256 x 42 -> whole 256-row tiles: the LN-fold epilogues live in the 256x256 kernel)
shapes = [  # (name, M, N, K, c dtype, epi, residual)
    ("v.qkv", Mv, 2304, 768, TD, 0, False), ("v.out+res", Mv, 768, 768, torch.float16, 0, True), ("v.fc+gelu", Mv, 3072, 768, TD, 1, False),
    ("v.proj+res", Mv, 768, 3072, torch.float16, 0, True), ("v.dproj*dgelu", Mv, 3072, 768, TD, 2, False), ("v.dfc", Mv, 768, 3072, TD, 0, False),
    ("v.dout", Mv, 768, 768, TD, 0, False), ("v.dqkv", Mv, 768, 2304, TD, 0, False),
    ("t.qkv", Mt, 1536, 512, TD, 0, False), ("t.fc+gelu", Mt, 2048, 512, TD, 1, False), ("t.dproj*dgelu", Mt, 2048, 512, TD, 2, False),
    ("t.dfc", Mt, 512, 2048, TD, 0, False),
]
torch.manual_seed(0)
table = []
for name, M, N, K, cdt, epi, res in shapes:
    fold = name.endswith(".qkv") or name.endswith(".fc+gelu")      # round 5: as the step launches them — LayerNorm folded in (fp16 stream x fp16 weights)
    OT = torch.float16 if fold else TD
    a = torch.randn(M, K, device=dev).to(OT)
    b = (torch.randn(N, K, device=dev) * 0.05).to(TD).to(OT)
    c = torch.zeros(M, N, device=dev, dtype=cdt)
    bias = torch.randn(N, device=dev)
    r = torch.randn(M, N, device=dev).to(cdt) if res else None
    aux = torch.randn(M, N, device=dev).to(TD) if epi else None
    if fold:
        blk = torch.zeros(3 * M + N, device=dev)
        blk[M:2 * M] = 1.0
        for _ in range(REPS):
            try:
                E.gemm(E.F16, a, b, c, M, N, K, bias=bias, residual=blk, ldr=M, epi=(E.EPI_LN_QUICKGELU if epi else E.EPI_LN), aux=aux)
            except Exception as e:
                raise RuntimeError(f"{name}: M {M} N {N} K {K} c {c.dtype} aux {None if aux is None else aux.dtype}: {e}")
    for _ in range(0 if fold else REPS):
        E.gemm(BF16, a, b, c, M, N, K, bias=bias, residual=r, epi=epi, aux=aux)
    torch.cuda.synchronize()
    rd = 2 * (M * K + N * K) + (2 * M * N if res else 0) + (2 * M * N if epi == 2 else 0)
    wr = 2 * M * N + (2 * M * N if epi == 1 else 0)
    table.append({"name": name, "M": M, "N": N, "K": K, "reads": rd, "writes": wr, "reps": REPS, "gflop": 2.0 * M * N * K / 1e9})
    del a, b, c, r, aux
print(json.dumps(table))
