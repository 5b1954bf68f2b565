#!/usr/bin/env python3
"""Time (and fingerprint) the eight vision-layer GEMMs of the bench with an alternate build of the library:
   python tools/gemm_variant.py base|<suffix>      (suffix -> lpi_amd/csrc/variants/liblpi_hip_<suffix>.so)
Used for A/B-ing kernel variants compiled with -D switches; each variant runs in its own process."""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.getcwd())
import lpi_amd._lib as L  # noqa: E402
if sys.argv[1] != "base":
    L.LIB_PATH = os.path.join(os.getcwd(), "lpi_amd/csrc/variants/liblpi_hip_%s.so" % sys.argv[1])
from lpi_amd import engine as E  # noqa: E402
from lpi_amd._lib import BF16, F32, call  # noqa: E402

dev = "cuda:0"
Mv = 54528
torch.manual_seed(0)
fp = []
for dt, TD in ((BF16, torch.bfloat16), (F32, torch.float32)):
    for (M, N, K) in ((256, 256, 128 if dt == BF16 else 64), (512, 768, 768), (768, 512, 3072)):
        a = torch.randn(M, K, device=dev).to(TD)
        b = (torch.randn(N, K, device=dev) * 0.05).to(TD)
        c = torch.zeros(M, N, device=dev, dtype=TD)
        call("lpi_set_tuning", 1, 1)
        E.gemm(dt, a, b, c, M, N, K, bias=torch.ones(N, device=dev))
        torch.cuda.synchronize()
        fp.append(hashlib.md5(c.cpu().view(torch.uint8).numpy().tobytes()).hexdigest()[:8])
print(sys.argv[1], "fingerprints", " ".join(fp))
shapes = [("qkv", Mv, 2304, 768, False, 0), ("out+res", Mv, 768, 768, True, 0), ("fc+gelu", Mv, 3072, 768, False, 1),
          ("proj+res", Mv, 768, 3072, True, 0), ("dproj", Mv, 3072, 768, False, 2), ("dfc", Mv, 768, 3072, False, 0),
          ("dout", Mv, 768, 768, False, 0), ("dqkv", Mv, 768, 2304, False, 0)]
out, tot = [], 0.0
for name, M, N, K, res, epi in shapes:
    a = torch.randn(M, K, device=dev).bfloat16()
    b = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    c = torch.zeros(M, N, device=dev, dtype=torch.float16 if res else torch.bfloat16)
    bias = torch.randn(N, device=dev)
    r = torch.randn(M, N, device=dev).half() if res else None
    aux = torch.randn(M, N, device=dev).bfloat16() if epi else None
    best = 1e9
    for rep in range(3):
        for _ in range(2):
            E.gemm(BF16, a, b, c, M, N, K, bias=bias, residual=r, epi=epi, aux=aux)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            E.gemm(BF16, a, b, c, M, N, K, bias=bias, residual=r, epi=epi, aux=aux)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 100)
    out.append(f"{best:6.0f}")
    tot += best
print(f"{sys.argv[1]:8s}", " ".join(out), f" sum {tot:.0f}")
