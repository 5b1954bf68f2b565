#!/usr/bin/env python3
"""Where a tile's time goes, per instantiation of the persistent 256x256 GEMM (diagnostic build -DLPI_GEMM_STAMPS: s_memtime stamps of wave 0 around the K
loops, the epilogues and the hand-over to the next tile; tools/build_variant.sh stamps gemm256p -DLPI_GEMM_STAMPS).  For every GEMM shape of the bench:
launch time, TFLOP/s, and the median workgroup's share of K-loop / epilogue / hand-over cycles, cycles per tile, and the in-kernel clock estimate
(cycles of the median workgroup / launch time).   usage: python3 tools/gemm_stamps.py [out.json]"""
import ctypes
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lpi_amd._lib as L  # noqa: E402
PRODUCT = os.environ.get("LPI_STAMP_PRODUCT") == "1"      # 1: the PRODUCT library, launch times only (no stamp executes there): the table's `us_product` column
if not PRODUCT:
    L.LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lpi_amd/csrc/variants/liblpi_hip_stamps.so")
from lpi_amd import engine as E  # noqa: E402
from lpi_amd._lib import BF16, F16, EPI_LN, EPI_LN_QUICKGELU, EPI_RES_ROWSTATS  # noqa: E402

dev = "cuda:0"
Mv, Mt = 54528, 11008
lib = L.load()
if not PRODUCT:
    rd = lib.lpi_gemm_stamps_read
    rd.argtypes = [ctypes.c_void_p]
buf = np.zeros((1024, 8), dtype=np.uint64)
TD = torch.bfloat16
torch.manual_seed(0)


def mk(M, N, K, kind):
    """-> a closure issuing the GEMM of the step: kind = qkv (LN fold) | fc (LN fold + QuickGELU, saves gelu') | res (fp16 residual + row statistics) |
    dgelu (x gelu') | plain"""
    bias = torch.randn(N, device=dev)
    b = (torch.randn(N, K, device=dev) * 0.05)
    if kind in ("qkv", "fc"):
        a = torch.randn(M, K, device=dev).half()
        bw = b.to(torch.bfloat16).half()      # 8 significant bits in an fp16 container, as LnLinear in bf16 mode
        lnb = torch.zeros(2 * M + N + 64, device=dev)
        lnb[M:2 * M] = 1.0
        c = torch.zeros(M, N, device=dev, dtype=TD)
        aux = torch.zeros(M, N, device=dev, dtype=TD) if kind == "fc" else None
        return lambda: E.gemm(F16, a, bw, c, M, N, K, bias=bias, residual=lnb, ldr=M, epi=EPI_LN_QUICKGELU if kind == "fc" else EPI_LN, aux=aux)
    a = torch.randn(M, K, device=dev).to(TD)
    bw = b.to(TD)
    if kind == "res":
        c = torch.zeros(M, N, device=dev, dtype=torch.float16)
        r = torch.randn(M, N, device=dev).half()
        part = torch.zeros(2 * (N // 128), M, device=dev)
        return lambda: E.gemm(BF16, a, bw, c, M, N, K, bias=bias, residual=r, epi=EPI_RES_ROWSTATS, aux=part)
    c = torch.zeros(M, N, device=dev, dtype=TD)
    if kind == "dgelu":
        aux = torch.randn(M, N, device=dev).to(TD)
        return lambda: E.gemm(BF16, a, bw, c, M, N, K, epi=E.EPI_DQUICKGELU, aux=aux)
    return lambda: E.gemm(BF16, a, bw, c, M, N, K)


shapes = [("in_proj (LN fold)", Mv, 2304, 768, "qkv"), ("out_proj + res + rowstats", Mv, 768, 768, "res"), ("c_fc (LN fold) + QuickGELU", Mv, 3072, 768, "fc"),
          ("c_proj + res + rowstats", Mv, 768, 3072, "res"), ("d c_proj x gelu'", Mv, 3072, 768, "dgelu"), ("d c_fc", Mv, 768, 3072, "plain"),
          ("d out_proj", Mv, 768, 768, "plain"), ("d in_proj", Mv, 768, 2304, "plain")]
only = os.environ.get("LPI_STAMP_ONLY")      # e.g. "dgelu": just the shapes of that kind (quick A/B runs with LPI_TUNING)
if only:
    shapes = [sh for sh in shapes if sh[4] in only.split(",")]
rows = []
for name, M, N, K, kind in shapes:
    fn = mk(M, N, K, kind)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    us = 1e9
    for _ in range(3):          # the fastest of three rounds of ten launches (a round that meets a clock ramp or another process's tail reads high)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = min(us, e0.elapsed_time(e1) * 100)
    if PRODUCT:
        rows.append({"gemm": name, "M": M, "N": N, "K": K, "us_product": round(us, 1), "tflops_product": round(2.0 * M * N * K / us / 1e6, 1)})
        print(f"{name:28s} {us:7.1f} us {rows[-1]['tflops_product']:7.1f} TF (product library)", flush=True)
        continue
    assert rd(buf.ctypes.data) == 0
    st = buf[:256].astype(np.float64)
    tot = st[:, 0] + st[:, 1] + st[:, 2]
    med = int(np.argsort(tot)[128])
    m, e, n, t = st[med][:4]
    ep = st[med][4:8] / max(t, 1)
    rows.append({"gemm": name, "M": M, "N": N, "K": K, "us": round(us, 1), "tflops": round(2.0 * M * N * K / us / 1e6, 1),
                 "kloop_frac": round(m / tot[med], 4), "epilogue_frac": round(e / tot[med], 4), "handover_frac": round(n / tot[med], 4),
                 "tiles_of_median_workgroup": int(t), "cycles_per_tile": round(tot[med] / max(t, 1)), "epilogue_cycles_per_tile": round(e / max(t, 1)),
                 "handover_cycles_per_tile": round(n / max(t - 1, 1)), "clock_ghz_estimate": round(tot[med] / us / 1e3, 2),
                 "epilogue_split_cycles_per_tile": {"barrier before the staging writes": round(ep[0]), "staging writes (+ side-tile wait)": round(ep[1]),
                                                    "barrier behind them": round(ep[2]), "staging reads, epilogue arithmetic, stores": round(ep[3])}})
    print(f"{name:28s} {us:7.1f} us {rows[-1]['tflops']:7.1f} TF  K loop {m / tot[med]:.3f}  epilogue {e / tot[med]:.3f}  hand-over {n / tot[med]:.3f}  "
          f"[epi split {ep[0]:.0f} / {ep[1]:.0f} / {ep[2]:.0f} / {ep[3]:.0f}]  tiles {int(t)}  cycles/tile {tot[med] / max(t, 1):.0f} (epilogue {e / max(t, 1):.0f}, hand-over {n / max(t - 1, 1):.0f})  ~{tot[med] / us / 1e3:.2f} GHz", flush=True)
if len(sys.argv) > 1:
    json.dump({"note": "diagnostic build (-DLPI_GEMM_STAMPS): wave 0's s_memtime stamps; the stamps themselves cost a few percent; full tiles of the persistent loop only "
                       "(the hybrid half-tile round is outside the stamps)", "shapes": rows}, open(sys.argv[1], "w"), indent=1)
