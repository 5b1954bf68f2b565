#!/usr/bin/env python3
"""Residual-epilogue GEMM with and without the row-statistics epilogue (LPI_EPI_RES_ROWSTATS), the finalize kernel and the statistics pass it
replaces: interleaved medians, microseconds.  usage: python3 tools/rowstats_gemm_time.py [M]"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lpi_amd import engine as E  # noqa: E402
from lpi_amd._lib import BF16, F16, call  # noqa: E402

DEV = "cuda:0"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 54528


def timed(fn, n=20):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    return [a.elapsed_time(b) * 1e3 for a, b in ev]


def main():
    s = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device="cpu").manual_seed(0)
    shapes = ((768, 768), (768, 3072))
    if len(sys.argv) > 2:
        shapes = tuple((768, int(k)) for k in sys.argv[2].split(","))
    for N, K in shapes:
        a = torch.randn(M, K, generator=g).bfloat16().to(DEV)
        b = (torch.randn(N, K, generator=g) * 0.05).bfloat16().to(DEV)
        bias = torch.randn(N, generator=g).to(DEV)
        res = torch.randn(M, N, generator=g).half().to(DEV)
        c = torch.zeros(M, N, dtype=torch.float16, device=DEV)
        part = torch.zeros(2 * (N // 128), M, device=DEV)
        mean, rstd = torch.zeros(M, device=DEV), torch.zeros(M, device=DEV)
        fns = {
            "res": lambda: call("lpi_gemm_nt", BF16, F16, M, N, K, a, K, b, K, c, N, bias, res, N, E.EPI_NONE, None, 0, 1.0, s),
            "res+stats": lambda: call("lpi_gemm_nt", BF16, F16, M, N, K, a, K, b, K, c, N, bias, res, N, E.EPI_RES_ROWSTATS, part, M, 1.0, s),
            "finalize": lambda: call("lpi_ln_stats_finalize", M, N, part, M, 1e-5, mean, rstd, s),
            "stats pass": lambda: call("lpi_layernorm_fwd", BF16, F16, M, N, c, N, None, None, None, 0, mean, rstd, s),
        }
        for f in fns.values():
            f()
        acc = {k: [] for k in fns}
        for _ in range(5):
            for k, f in fns.items():
                acc[k] += timed(f, 10)
        print(f"M {M} N {N} K {K}: " + "  ".join(f"{k} {statistics.median(v):.1f}" for k, v in acc.items()), flush=True)


if __name__ == "__main__":
    main()
