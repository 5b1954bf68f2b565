"""The few-row GEMMs of a step (pooled rows of the last block, heads), two towers per launch (gemm_rows.hip: 32 x 32 tiles over the whole K range) against the
128 x 128 kernel on the same operands (two launches of a handful of tiles).  Round 6 measured it against the split-K launch pairs it replaced:
profiles/r06_rows_ab.txt.  usage: python tools/gemm_rows_bench.py [bf16|f16] [B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lpi_amd import _lib, engine as E  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "bf16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dt = {"bf16": E.BF16, "f16": E.F16}[mode]
td = {"bf16": torch.bfloat16, "f16": torch.float16}[mode]
dev = torch.device("cuda:0")
s = torch.cuda.current_stream().cuda_stream
OPS = [("q / out_proj", (768, 768), (512, 512), E.EPI_NONE, False), ("c_fc + QuickGELU", (3072, 768), (2048, 512), E.EPI_QUICKGELU, True),
       ("c_proj", (768, 3072), (512, 2048), E.EPI_NONE, False), ("d c_proj x gelu'", (3072, 768), (2048, 512), E.EPI_DQUICKGELU, True),
       ("d c_fc", (768, 3072), (512, 2048), E.EPI_NONE, False), ("heads", (512, 768), (512, 512), E.EPI_NONE, False)]


def timed(fn, n=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


tot = [0.0, 0.0]
for name, (n0, k0), (n1, k1), epi, aux in OPS + OPS[:1]:      # the first op again at the end: the first measurement of a process has been seen 30x off
    probs = []
    for (n, k) in ((n0, k0), (n1, k1)):
        auxt = torch.bfloat16
        probs.append(dict(M=B, N=n, K=k, a=torch.randn(B, k, device=dev).to(td), b=(0.05 * torch.randn(n, k, device=dev)).to(td),
                          c=torch.zeros(B, n, device=dev, dtype=td), bias=None if epi == E.EPI_DQUICKGELU else torch.randn(n, device=dev),
                          aux=torch.randn(B, n, device=dev).to(auxt) if aux else None))
    def two_launches():
        for p in probs:
            _lib.call("lpi_gemm_nt", dt, dt, p["M"], p["N"], p["K"], p["a"], p["K"], p["b"], p["K"], p["c"], p["N"], p["bias"], None, 0, epi, p["aux"],
                      p["N"] if p["aux"] is not None else 0, 1.0, s)
    t_sk = timed(two_launches)
    c_sk = [p["c"].clone() for p in probs]
    t_rw = timed(lambda: _lib.gemm_rows(dt, dt, epi, 1.0, probs, s))
    err = max(float((p["c"].float() - c.float()).abs().max() / (c.float().abs().max() + 1e-30)) for p, c in zip(probs, c_sk))
    tot[0] += t_sk
    tot[1] += t_rw
    print(f"{name:20s} B {B}  N x K {n0} x {k0} | {n1} x {k1}   128 x 128 tiles, two launches {t_sk:6.1f} us   one launch {t_rw:6.1f} us   max rel diff {err:.1e}")
print(f"sum over the six ops: {tot[0]:.1f} -> {tot[1]:.1f} us")
