"""Host-side engine of the MI355X LPI hot path: owns the frozen CLIP weights (pre-converted and pre-transposed
for the matrix cores), a workspace arena sized for 288 GB of HBM, and sequences the HIP kernels of
``liblpi_hip.so`` for the prompted dual-encoder forward and its dgrad-only backward.

What it stands in for in the reference (paths relative to /root/reference/retrieval/):
  * ``VisionTransformer.forward``                    models/clip/model.py:227-259
  * ``TextEncoder.forward`` + ``PromptLearner.forward``  models/clip/prompt_learner.py:52-63, 128-163
  * ``Transformer`` / ``ResidualAttentionBlock``     models/clip/model.py:168-207
  * their autograd backward w.r.t. the prompt tensors (the backbone is frozen: sprompt.py:230-237)

PyTorch is used for device memory, streams and (in ``dp.py``) torch.distributed only; all arithmetic on the
path runs in the HIP kernels.  There is no CPU / eager fallback: without the library or a GPU this raises.

Layout: token rows are batch-major (row = b*L + l), padded to a multiple of 128 rows for the 128x128 GEMM
tiles; the residual stream, LN statistics and every reduction are f32; ``dtype`` ('f32' | 'bf16') selects the
matrix-core operand type (see include/lpi_hip.h).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch

from . import _lib
from ._lib import BF16, EPI_DQUICKGELU, EPI_LN, EPI_LN_QUICKGELU, EPI_NONE, EPI_QUICKGELU, EPI_RES_ROWSTATS, F16, F32, call
from .synth import ClipConfig

import ctypes
import os as _os

_DT = {"f32": F32, "fp32": F32, "float32": F32, "bf16": BF16, "bfloat16": BF16, "f16": F16, "fp16": F16, "float16": F16, "half": F16}


def _grad_dtype(dt: int) -> int:
    """Operand / storage type of the BACKWARD for a forward operand type: f32 stays f32; bf16 and f16 forwards both back-propagate in
    bf16 (gradients do not fit fp16's range without loss scaling; the reference, which runs fp16 end to end, simply lives with that)."""
    return F32 if dt == F32 else BF16
_TORCH_DT = {F32: torch.float32, BF16: torch.bfloat16, F16: torch.float16}


@dataclass(frozen=True)
class EngineOptions:
    """Everything that selects a code path of the engine, as ONE typed per-engine object (round 6; it replaced sixteen module globals read from the
    environment at import).  Three fields can be set from the environment (from_env(): documented fall-backs an operator may want without touching
    code); the rest are constructor arguments that the exactness tests use.  Every settled A/B arm of rounds 1-5 is gone from the code (row-chunked MLP,
    un-grouped launches, the pooled-MLP-only last block, no split-K, full-mantissa LayerNorm-fold weights): their measurements are in profiles/.

    residual_f16 (LPI_RESIDUAL=f32 -> False): the 2-byte modes store the forward residual stream in fp16 — the reference's own activation type (it runs fp16
        end to end, model.py:371-392) — which halves the bytes of the HBM-bound LayerNorms and residual epilogues; statistics, accumulation and the
        pooled rows stay f32.  False keeps an f32 stream (-5.6 % throughput) and with it LayerNorm as its own kernel.  f32 mode always uses f32.
    ln_fold (LPI_LN_FOLD): 0 = LayerNorm as its own kernel in front of in_proj / c_fc (f32 mode always); 1 = ln_1 folded into in_proj; 2 = ln_2 folded
        into c_fc as well (default: 22.82-22.91 ms per step against 22.97-23.01 with 1, and the more accurate one: LN(x) is never rounded to bf16).
    rowstats (LPI_ROWSTATS): 2 = the statistics the folded LayerNorms need come out of the GEMM that WRITES the stream (out_proj / c_proj + residual,
        LPI_EPI_RES_ROWSTATS + a finalize launch) instead of a pass over it; 1 = ln_2's only; 0 = the two-sweep statistics pass everywhere (what the
        guard below falls back to).  Same values up to the order of an f32 sum.
    rowstat_guard: the one-sweep statistics (E[x^2] - mean^2 in f32) lose digits as (mean / std)^2 * 1e-7; the kernels count the rows with
        mean^2 > 64 var (lpi_rowstat_guard), the engine reads the count without synchronising and drops BOTH towers to rowstats = 0 from the next forward
        on, with a warning (tests/test_round6_gpu.py: rows 12 sigma off zero).
    pooled_last: exact dead-row elimination in the last block (only the pooled token's query / softmax row / out_proj / MLP are evaluated, K and V for
        every token).  False evaluates the whole block: the reference's literal order, kept for the test that holds the two equal.
    stream_pool (round 6; needs pooled_last): the last block of a tower with uniform sequences (the vision tower) does not compute K and V at all — with one
        live query per (sample, head), q.k_l = LN1(x_l).(W_k^T q) + const and sum_l p_l v_l = W_v (sum_l p_l LN1(x_l)) + b_v, so two passes over the sample's
        rows of the residual stream replace the [B L, 2 d] projection, its dgrad and the single-query attention over them (lpi_spool_attn_fwd / _bwd,
        csrc/attn_stream.hip: exact algebra; 2-byte modes with the fp16 stream).  False keeps the K / V path (the exactness test's other arm).
    l0_prompt_rows: the first block's backward reads dL/dx_0 at the prompt rows only (nothing upstream of the prompt slots is trainable), so its in_proj
        dgrad and LN1 backward run on B*P rows.  False computes every row (the exactness test's other arm).
    qkv_grouped: in_proj's output features of the non-causal tower re-ordered to [head][q|k|v][64] (Tower.__init__).  Same numbers; the attention
        kernels' (sample, head) slices become 384-byte pieces.  OFF: measured in the step it buys nothing (attn_bwd4 170.9 us either way, forward pair
        98.4 -> 96.5 us, step 22.27 ms both: profiles/r06_experiments.md) although the kernels alone gain 7 % — kept as the tested proof of that."""
    residual_f16: bool = True
    ln_fold: int = 2
    rowstats: int = 2
    rowstat_guard: bool = True
    pooled_last: bool = True
    stream_pool: bool = True
    l0_prompt_rows: bool = True
    qkv_grouped: bool = False

    def __post_init__(self):
        if self.ln_fold not in (0, 1, 2) or self.rowstats not in (0, 1, 2):
            raise ValueError("ln_fold and rowstats are 0, 1 or 2")

    @staticmethod
    def from_env(**over):
        """The defaults, with LPI_RESIDUAL (f16 | f32), LPI_LN_FOLD (0 | 1 | 2) and LPI_ROWSTATS (0 | 1 | 2) applied — read HERE, when an engine is built,
        not at import — and then the keyword overrides."""
        env = {}
        if "LPI_RESIDUAL" in _os.environ:
            v = _os.environ["LPI_RESIDUAL"]
            if v not in ("f16", "f32"):
                raise ValueError(f"LPI_RESIDUAL={v!r}: f16 or f32")
            env["residual_f16"] = v == "f16"
        for name, key in (("LPI_LN_FOLD", "ln_fold"), ("LPI_ROWSTATS", "rowstats")):
            if name in _os.environ:
                env[key] = int(_os.environ[name])
        env.update(over)
        return EngineOptions(**env)


def _pad(n: int, m: int = 128) -> int:
    return (n + m - 1) // m * m


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _require_gpu(device):
    if device.type != "cuda" or not torch.cuda.is_available():
        raise _lib.LpiError("lpi_amd runs on an MI355X only (device must be cuda:N); there is no CPU fallback")


class Linear:
    """A frozen nn.Linear prepared for the NT GEMM: W [out,in] and W^T [in,out] in the operand dtype, f32 bias."""

    def __init__(self, w: torch.Tensor, b: Optional[torch.Tensor], dt: int, device, k_pad: Optional[int] = None):
        w = w.to(device=device, dtype=torch.float32)
        if k_pad is not None and k_pad != w.shape[1]:
            w = torch.nn.functional.pad(w, (0, k_pad - w.shape[1]))
        self.out_features, self.in_features = w.shape
        self.w = w.to(_TORCH_DT[dt]).contiguous()                                         # forward operand
        self.wt = w.t().contiguous().to(_TORCH_DT[_grad_dtype(dt)]).contiguous()          # dgrad operand (W^T, K-contiguous)
        self.b = None if b is None else b.to(device=device, dtype=torch.float32).contiguous()


class LnLinear:
    """LayerNorm FOLDED into the frozen Linear behind it (LPI_EPI_LN, include/lpi_hip.h): LN(x) W^T + b = rstd (x (gamma o W)^T - mean c1) + c2 with
    c1[n] = sum_k (gamma o W)[n,k] (of the ROUNDED fp16 operand, so that the mean term cancels exactly) and c2 = W beta + b.  The GEMM then reads
    the fp16 residual stream itself; LN(x) is never written (model.py:172-177: ln_1 -> in_proj, ln_2 -> c_fc; weights are frozen, so nothing
    downstream needs LN(x) either).  Operands are fp16 in bf16 mode too: x is the fp16 stream, and an fp16 gamma o W keeps three more bits."""

    def __init__(self, w: torch.Tensor, b: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, device, bf16_weights: bool = False):
        w = w.to(device=device, dtype=torch.float64)
        g, be = gamma.to(device=device, dtype=torch.float64), beta.to(device=device, dtype=torch.float64)
        wl = w * g[None, :]
        if bf16_weights:
            # bf16 mode: gamma o W keeps bf16's 8 significant bits (the precision of every other weight of the mode) inside the fp16 container.
            # Not for accuracy — for POWER: the chip is power-limited under the GEMMs and the matrix pipe's energy follows the operands'
            # mantissa activity; with the three low mantissa bits of B zero the fp16-operand in_proj GEMM loses most of its 4 % penalty
            # against the bf16 one (24.21 -> 24.08 ms per step, profiles/r02_gemm_experiments.md).  (full fp16 mantissas: 24.21 ms.)
            wl = wl.to(torch.bfloat16)
        self.w = wl.to(torch.float16).contiguous()
        self.c1 = self.w.double().sum(dim=1).float().contiguous()
        self.c2 = (w @ be + b.to(device=device, dtype=torch.float64)).float().contiguous()


# When set to a list, every gemm() launch is bracketed by HIP events on the launch stream and
# (start, end, algorithmic_flops) is appended: bench.py's live roofline measurement of the dominant kernel.
GEMM_PROFILE = None


def _few_rows(dt, M, N, K):
    """True for a GEMM with a few rows (pooled rows of the last block, heads, the loss's gradient GEMMs): it runs as ONE launch of 32 x 32 tiles over the whole K
    range (lpi_gemm_nt_rows) instead of a handful of 128 x 128 or 256 x 256 tiles.  Rounds 2-5 ran these as split-K partial + reduction (two launches, f32
    partials through HBM): the one-launch kernel halves their time, -0.15 ms per step at ViT-B/16 with 256 pairs (profiles/r06_experiments.md section 10);
    512 rows (ViT-L/14 with 512 pairs: 8-32 tiles of the 256 x 256 kernels before) -0.6 ms of 179 (tools/probe/few_rows_m512.py)."""
    return M <= 512 and M % 128 == 0 and N % 128 == 0 and K % (32 if dt == F32 else 64) == 0


def gemm(dt, a, b, c, M, N, K, bias=None, residual=None, epi=EPI_NONE, aux=None, alpha=1.0, m_real=None, ldr=None):
    """c[M,N] = epi(alpha * a[M,K] @ b[N,K]^T + bias) + residual   (all row-major, contiguous rows).
    m_real: un-padded row count, used only for algorithmic-FLOP accounting."""
    cdt = F32 if c.dtype == torch.float32 else (F16 if c.dtype == torch.float16 else BF16)
    prof = GEMM_PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    if ldr is None:
        ldr = residual.stride(0) if residual is not None else 0
    ln = epi in (EPI_LN, EPI_LN_QUICKGELU)      # `residual` is the LN operand block then (include/lpi_hip.h), ldr its vector stride
    if (cdt != F16 or dt == F16) and not ln and epi != EPI_RES_ROWSTATS and _few_rows(dt, M, N, K):
        _lib.gemm_rows(dt, cdt, epi, alpha, [dict(M=M, N=N, K=K, a=a, b=b, c=c, bias=bias, residual=residual, aux=aux, ldr=ldr)], _stream())
    else:
        call("lpi_gemm_nt", dt, cdt, M, N, K, a, a.stride(0), b, b.stride(0), c, c.stride(0), bias, residual,
             ldr, epi, aux, aux.stride(0) if aux is not None else 0,
             float(alpha), _stream())
    if prof is not None:
        e1.record()
        mr = m_real or M
        nbytes = (mr * K + N * K) * a.element_size() + mr * N * (c.element_size() + (residual.element_size() if residual is not None and not ln else 0)
                                                                 + (aux.element_size() if aux is not None and epi != EPI_RES_ROWSTATS else 0))
        kind = int(_lib.load().lpi_gemm_last_kernel())      # LPI_GEMM_K_*: which kernel the dispatcher launched (same host thread)
        prof.append((e0, e1, 2.0 * mr * N * K, nbytes, kind))


_LN_FOLD_OK = {}


def _ln_fold_ok(Mp, d):
    """True if the persistent 256x256 kernel (the one with the LN-fold epilogues) takes the in_proj / c_fc GEMMs of a [Mp, d] stream."""
    key = (Mp, d)
    ok = _LN_FOLD_OK.get(key)
    if ok is None:
        lib = _lib.load()
        ok = _LN_FOLD_OK[key] = bool(lib.lpi_gemm_ln_supported(F16, Mp, 2 * d, d) and lib.lpi_gemm_ln_supported(F16, Mp, 3 * d, d)
                                     and lib.lpi_gemm_ln_supported(F16, Mp, 4 * d, d))
    return ok


class Req:
    """Base of everything a tower generator yields to its driver (run_alone / run_lockstep).  `tag` names the op — requests with EQUAL tags of the two
    towers are issued as one pair; `optional` says that a tower emits this request CONDITIONALLY (its partner may have nothing at that point): run_lockstep
    then issues it alone without advancing the partner, so the towers re-align instead of running shifted — and ungrouped — for the rest of the pass.  Every
    request class states it (a class default or a per-instance value); LOCKSTEP_STATS['shifted'] counts the issues that found no partner although neither side
    was optional: expected where the towers differ in depth (ViT-L/14: 24 vision blocks against 12 text blocks), 0 for towers of equal depth — asserted for
    the benchmarked configuration (tests/test_round5_gpu.py), where a request class that forgot the flag would show up."""
    __slots__ = ()
    optional = False


LOCKSTEP_STATS = {"paired": 0, "alone_optional": 0, "shifted": 0}


class GemmReq(Req):
    """A GEMM a tower wants issued: the towers' forward / backward are generators that YIELD their GEMMs instead of launching them, so
    that a driver (run_lockstep) can launch the vision and the text tower's GEMM of the same layer op as ONE grouped launch
    (lpi_gemm_nt_grouped).  `tag` names the op ("3.qkv", "last.kv", ...): only requests with equal tags are paired; None = never."""
    __slots__ = ("tag", "dt", "a", "b", "c", "M", "N", "K", "kw", "optional")

    def __init__(self, tag, dt, a, b, c, M, N, K, optional=False, **kw):
        # optional: a request only THIS tower emits (e.g. a statistics pass the other tower does not take): run_lockstep issues it alone
        self.tag, self.dt, self.a, self.b, self.c, self.M, self.N, self.K, self.kw, self.optional = tag, dt, a, b, c, M, N, K, kw, optional

    def issue(self):
        gemm(self.dt, self.a, self.b, self.c, self.M, self.N, self.K, **self.kw)


class AttnFwdReq(Req):
    """An attention forward a tower wants issued (args of lpi_attn_fwd_varlen after the dtype): the two towers' go out as one launch."""
    __slots__ = ("tag", "dt", "args")

    def __init__(self, tag, dt, *args):
        self.tag, self.dt, self.args = tag, dt, args

    def issue(self):
        a = self.args
        if len(a) > 11 and a[11] is not None:      # layout strides (Tower.qkv_lay): the descriptor form is the one that takes them
            _lib.attn_fwd_one(self.dt, a, _stream())
        elif len(a) > 10 and a[10]:      # shared prefix (PackedIds(shared=...)): B, L, row_start, H, qkv, ldqkv, ctx, ldctx, lse, causal, shared rows
            call("lpi_attn_fwd_shared", self.dt, a[0], a[1], a[2], a[10], *a[3:9], _stream())
        else:
            call("lpi_attn_fwd_varlen", self.dt, *a[:10], _stream())


class SoloReq(Req):
    """A launch only THIS tower makes at a step position both towers tag alike (fn = None: nothing at all — the tower has no such op, but says so, so that the
    towers' request streams stay aligned): the partner's request of the same tag is issued alone."""
    __slots__ = ("tag", "fn")

    def __init__(self, tag, fn=None):
        self.tag, self.fn = tag, fn

    def issue(self):
        if self.fn is not None:
            self.fn()


class LnReq(Req):
    """A LayerNorm a tower wants issued (forward: kind "fwd", args of lpi_layernorm_fwd after the two dtypes; backward: "bwd", args of
    lpi_layernorm_bwd after the three dtypes).  Yielded like a GemmReq so that the two towers' LayerNorms of the same layer go out as
    ONE launch (lpi_layernorm_fwd_pair / _bwd_pair): the text tower's alone is a few-microsecond kernel that is mostly launch ramp."""
    __slots__ = ("tag", "kind", "dts", "args", "optional")

    def __init__(self, tag, kind, dts, *args, optional=False):
        # optional: a request this tower emits CONDITIONALLY (the forward LayerNorms / statistics passes: a tower whose statistics come out of a
        # GEMM epilogue emits a StatFinReq or nothing at that point) — run_lockstep may issue it alone without advancing the partner
        self.tag, self.kind, self.dts, self.args, self.optional = tag, kind, dts, args, optional

    def issue(self):
        call("lpi_layernorm_fwd" if self.kind == "fwd" else "lpi_layernorm_bwd", *self.dts, *self.args, _stream())


class StatFinReq(Req):
    """The row statistics of a residual-stream GEMM output from the slot sums its epilogue left (LPI_EPI_RES_ROWSTATS -> lpi_ln_stats_finalize):
    args = (rows, d, part, ld, mean, rstd).  The two towers' go out as one launch."""
    __slots__ = ("tag", "args")
    optional = True      # emitted only by a tower that takes its statistics from the producing GEMM's epilogue (see LnReq.optional)

    def __init__(self, tag, *args):
        self.tag, self.args = tag, args

    def issue(self):
        rows, d, part, ld, mean, rstd = self.args
        call("lpi_ln_stats_finalize", rows, d, part, ld, 1e-5, mean, rstd, _stream())


class RowReq(Req):
    """Small row kernels a tower wants issued (a list of _lib.RowJob of ONE dependency level: lpi_row_jobs).  Yielded like a GemmReq so that the two
    towers' jobs of the same op go out as ONE launch — each alone is a ~5 us launch on B or B*P rows, and the tail of a step is a chain of them."""
    __slots__ = ("tag", "jobs", "optional")

    def __init__(self, tag, jobs, optional=False):
        self.tag, self.jobs, self.optional = tag, list(jobs), optional      # optional: see LnReq

    def issue(self):
        _lib.row_jobs(self.jobs, _stream())


class RowsSumReq(Req):
    """A batch sum of prompt rows (args of lpi_rows_sum_over_batch_varlen after the dtype, before the stream); the towers' pair = one launch."""
    __slots__ = ("tag", "dt", "args")
    optional = True      # emitted by the layers below the prompt depth only: a tower with fewer such layers must not be shifted (run_lockstep)

    def __init__(self, tag, dt, *args):
        self.tag, self.dt, self.args = tag, dt, args

    def issue(self):
        call("lpi_rows_sum_over_batch_varlen", self.dt, *self.args, _stream())


class PoolAttnReq(Req):
    """The pooled-row attention of the last block (forward or backward): `desc` = the lpi_attn_pooled_desc fields of this tower."""
    __slots__ = ("tag", "dt", "bwd", "desc")

    def __init__(self, tag, dt, bwd, **desc):
        self.tag, self.dt, self.bwd, self.desc = tag, dt, bwd, desc

    def issue(self):
        q = self.desc
        if q.get("shared_rows"):
            _lib.attn_pooled_one(self.dt, q, _stream(), backward=self.bwd)
        elif self.bwd:
            call("lpi_attn_pooled_bwd_varlen", self.dt, q["B"], q["L"], q["row_start"], q["H"], q["q"], q["ldq"], q["qkv"], q["ldqkv"], q["idx"], q["dctx"],
                 q["lddctx"], q["lse"], q["dq"], q["lddq"], q["dqkv"], q["lddqkv"], q["causal"], _stream())
        else:
            call("lpi_attn_pooled_fwd_varlen", self.dt, q["B"], q["L"], q["row_start"], q["H"], q["q"], q["ldq"], q["qkv"], q["ldqkv"], q["idx"], q["ctx"],
                 q["ldctx"], q["lse"], q["causal"], _stream())


def _cdt(c):
    return F32 if c.dtype == torch.float32 else (F16 if c.dtype == torch.float16 else BF16)


def _issue_pair(r0: GemmReq, r1: GemmReq):
    """The two towers' GEMM of the same op: one grouped launch where the library can (two large bf16 / f16 problems of the same epilogue
    kind), else two launches — the same bits either way."""
    if isinstance(r0, SoloReq) or isinstance(r1, SoloReq):
        r0.issue()
        r1.issue()
        return
    if isinstance(r0, RowReq) or isinstance(r1, RowReq):
        if isinstance(r0, RowReq) and isinstance(r1, RowReq):
            _lib.row_jobs(r0.jobs + r1.jobs, _stream())
        else:
            r0.issue()
            r1.issue()
        return
    if isinstance(r0, RowsSumReq) or isinstance(r1, RowsSumReq):
        if isinstance(r0, RowsSumReq) and isinstance(r1, RowsSumReq) and r0.dt == r1.dt:
            _lib.rows_sum_pair(r0.dt, r0.args, r1.args, _stream())
        else:
            r0.issue()
            r1.issue()
        return
    if isinstance(r0, PoolAttnReq) or isinstance(r1, PoolAttnReq):
        if isinstance(r0, PoolAttnReq) and isinstance(r1, PoolAttnReq) and r0.dt == r1.dt and r0.bwd == r1.bwd:
            _lib.attn_pooled_pair(r0.dt, r0.desc, r1.desc, _stream(), backward=r0.bwd)
        else:
            r0.issue()
            r1.issue()
        return
    if isinstance(r0, AttnFwdReq) or isinstance(r1, AttnFwdReq):
        if isinstance(r0, AttnFwdReq) and isinstance(r1, AttnFwdReq) and r0.dt == r1.dt and r0.dt != F32:
            _lib.attn_fwd_pair(r0.dt, r0.args, r1.args, _stream())
        else:
            r0.issue()
            r1.issue()
        return
    if isinstance(r0, StatFinReq) or isinstance(r1, StatFinReq):
        if isinstance(r0, StatFinReq) and isinstance(r1, StatFinReq):
            a, b = r0.args, r1.args
            call("lpi_ln_stats_finalize_pair", a[0], a[1], a[2], a[3], a[4], a[5], b[0], b[1], b[2], b[3], b[4], b[5], 1e-5, _stream())
        else:
            r0.issue()
            r1.issue()
        return
    if isinstance(r0, LnReq) or isinstance(r1, LnReq):
        stats_only = [r.kind == "fwd" and r.args[6] is None for r in (r0, r1) if isinstance(r, LnReq)]      # one tower folds its LayerNorms, the other not
        if (isinstance(r0, LnReq) and isinstance(r1, LnReq) and r0.kind == r1.kind and r0.dts == r1.dts
                and stats_only[0] == stats_only[1]):
            (_lib.layernorm_fwd_pair if r0.kind == "fwd" else _lib.layernorm_bwd_pair)(*r0.dts, r0.args, r1.args, _stream())
        else:
            r0.issue()
            r1.issue()
        return
    k0, k1 = r0.kw, r1.kw
    same = (r0.dt == r1.dt and r0.dt != F32 and r0.c.dtype == r1.c.dtype and k0.get("epi", EPI_NONE) == k1.get("epi", EPI_NONE)
            and (k0.get("residual") is None) == (k1.get("residual") is None) and (k0.get("aux") is None) == (k1.get("aux") is None)
            and (k0.get("bias") is None) == (k1.get("bias") is None) and k0.get("alpha", 1.0) == k1.get("alpha", 1.0)
            )
    cdt = _cdt(r0.c)
    # few-row GEMMs (pooled rows of the last block, heads): the two towers' in one launch (lpi_gemm_nt_rows)
    few = bool(same and (cdt != F16 or r0.dt == F16) and k0.get("epi", EPI_NONE) not in (EPI_LN, EPI_LN_QUICKGELU, EPI_RES_ROWSTATS)
               and _few_rows(r0.dt, r0.M, r0.N, r0.K) and _few_rows(r1.dt, r1.M, r1.N, r1.K))
    if not same or not (few or min(r0.M, r1.M) > 256):
        r0.issue()
        r1.issue()
        return
    prof = GEMM_PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    probs = [dict(M=r.M, N=r.N, K=r.K, a=r.a, b=r.b, c=r.c, bias=r.kw.get("bias"), residual=r.kw.get("residual"), aux=r.kw.get("aux"), ldr=r.kw.get("ldr"))
             for r in (r0, r1)]
    ln = k0.get("epi", EPI_NONE) in (EPI_LN, EPI_LN_QUICKGELU)
    if few:
        _lib.gemm_rows(r0.dt, cdt, k0.get("epi", EPI_NONE), k0.get("alpha", 1.0), probs, _stream())
    else:
        _lib.gemm_grouped(r0.dt, cdt, k0.get("epi", EPI_NONE), k0.get("alpha", 1.0), probs, _stream())
    if prof is not None:
        e1.record()
        fl = nb = 0.0
        for r in (r0, r1):
            mr = r.kw.get("m_real") or r.M
            res, aux = (None if ln else r.kw.get("residual")), r.kw.get("aux")
            fl += 2.0 * mr * r.N * r.K
            nb += (mr * r.K + r.N * r.K) * r.a.element_size() + mr * r.N * (r.c.element_size() + (res.element_size() if res is not None else 0)
                                                                         + (aux.element_size() if aux is not None and k0.get("epi") != EPI_RES_ROWSTATS else 0))
        prof.append((e0, e1, fl, nb, int(_lib.load().lpi_gemm_last_kernel())))


def run_alone(gen):
    """Drive one tower generator: every GEMM it yields is launched at once.  Returns the generator's return value."""
    try:
        while True:
            next(gen).issue()
    except StopIteration as e:
        return e.value


def run_lockstep(g0, g1):
    """Drive two tower generators in lock step: requests with EQUAL tags are issued as a pair (_issue_pair); an untagged request is
    issued alone and only its tower advances.  Two different tags: if exactly one of them is an OPTIONAL request (one a tower emits
    conditionally: the statistics passes / finalizes, which depend on the tower's fold level, width and LPI_ROWSTATS), it is issued alone and only ITS
    tower advances — the towers re-align at the next common request instead of running shifted by one (and ungrouped) for the rest of the pass;
    otherwise both are issued one after the other and both advance.  Returns both return values."""
    gens, cur, done, ret = [g0, g1], [None, None], [False, False], [None, None]

    def adv(i):
        try:
            cur[i] = next(gens[i])
        except StopIteration as e:
            cur[i], done[i], ret[i] = None, True, e.value

    adv(0)
    adv(1)
    while not (done[0] and done[1]):
        for i in (0, 1):                 # untagged requests (and a tower whose partner has finished) run alone
            while not done[i] and (cur[i].tag is None or done[1 - i]):
                cur[i].issue()
                adv(i)
        if done[0] or done[1]:
            continue
        if cur[0].tag == cur[1].tag:
            _issue_pair(cur[0], cur[1])
            LOCKSTEP_STATS["paired"] += 1
        else:
            o0, o1 = bool(cur[0].optional), bool(cur[1].optional)
            if o0 != o1:
                i = 0 if o0 else 1
                cur[i].issue()
                LOCKSTEP_STATS["alone_optional"] += 1
                adv(i)
                continue
            cur[0].issue()
            cur[1].issue()
            LOCKSTEP_STATS["shifted"] += 1
        adv(0)
        adv(1)
    return ret[0], ret[1]


@dataclass
class TowerSpec:
    width: int
    heads: int
    layers: int
    causal: bool


class Tower:
    """One transformer tower (vision or text): weights + forward/backward over a [B*L, d] residual stream."""

    def __init__(self, sd: dict, prefix: str, spec: TowerSpec, dt: int, device, opt: Optional[EngineOptions] = None):
        self.spec, self.dt, self.device = spec, dt, device
        self.opt = opt = opt if opt is not None else EngineOptions.from_env()
        self.gdt = _grad_dtype(dt)                                      # operand / storage type of the backward
        self.xdt = F16 if (dt != F32 and opt.residual_f16) else F32   # storage type of the forward residual stream
        self.blocks = []
        f = lambda k: torch.as_tensor(np.asarray(sd[k])) if not torch.is_tensor(sd[k]) else sd[k]  # noqa: E731
        # HEAD-GROUPED q / k / v (round 6): in_proj's output features re-ordered from [q | k | v][head][64] to [head][q | k | v][64] — a permutation of the
        # frozen weight's rows (and of its bias, of W^T's columns and of the LayerNorm fold's c1 / c2), prepared once, so the GEMMs run unchanged and produce
        # the same numbers in another column order.  A (sample, head) slice of qkv / dqkv is then L pieces of 384 contiguous bytes instead of 3 L pieces of
        # 128: the streamed attention backward, which is bound by its memory path, 198.6 -> 183.8 us on the vision shape (tools/attn_layout_probe2.py:
        # the kernels take the layout as strides and give the same bits).  The non-causal tower in the 2-byte modes; its last block keeps the
        # interleaved order (the pooled-row kernels read K and V as the rows d .. 3 d of the weight).  EngineOptions.qkv_grouped: OFF by default (in the step
        # it measured nothing: see there).
        self.qkv_grouped = bool(opt.qkv_grouped and dt != F32 and not spec.causal and spec.layers > 1)
        d_, H_ = spec.width, spec.heads
        self.qkv_lay = (3 * 64, 64, 64)                                 # (head stride, q -> k -> v stride, ctx head stride) of a grouped block
        perm = torch.arange(3 * d_).view(3, H_, 64).permute(1, 0, 2).reshape(-1)      # new feature h*192 + w*64 + c  <-  old w*d + h*64 + c
        for i in range(spec.layers):
            p = f"{prefix}resblocks.{i}."
            grouped = self.qkv_grouped and i < spec.layers - 1
            wq, bq = f(p + "attn.in_proj_weight"), f(p + "attn.in_proj_bias")
            if grouped:
                wq, bq = wq[perm.to(wq.device)], bq[perm.to(bq.device)]
            blk = {
                "qkv": Linear(wq, bq, dt, device),
                "grouped": grouped,
                "out": Linear(f(p + "attn.out_proj.weight"), f(p + "attn.out_proj.bias"), dt, device),
                "fc": Linear(f(p + "mlp.c_fc.weight"), f(p + "mlp.c_fc.bias"), dt, device),
                "proj": Linear(f(p + "mlp.c_proj.weight"), f(p + "mlp.c_proj.bias"), dt, device),
            }
            for nm in ("ln_1", "ln_2"):
                blk[nm + ".w"] = f(p + nm + ".weight").to(device=device, dtype=torch.float32).contiguous()
                blk[nm + ".b"] = f(p + nm + ".bias").to(device=device, dtype=torch.float32).contiguous()
            if dt != F32 and self.xdt == F16 and opt.ln_fold:
                w8 = dt == BF16      # bf16 mode: gamma o W keeps bf16's 8 significant bits inside the fp16 container (LnLinear: power, not accuracy)
                blk["qkv_ln"] = LnLinear(wq, bq, blk["ln_1.w"], blk["ln_1.b"], device, w8)
                blk["fc_ln"] = LnLinear(f(p + "mlp.c_fc.weight"), f(p + "mlp.c_fc.bias"), blk["ln_2.w"], blk["ln_2.b"], device, w8)
            if i == spec.layers - 1 and dt != F32:
                # W^T of in_proj in the FORWARD's operand type, for the last block without K and V (stream_pool: csrc/attn_stream.hip reads a head's 64 output
                # features as one contiguous run): bf16 mode has it already (the dgrad operand), f16 mode gets an fp16 copy (3.5 MB at ViT-B/16)
                blk["qkv_wT"] = blk["qkv"].wt if dt == self.gdt else blk["qkv"].w.t().contiguous()
                # ... and the row-major weight in the BACKWARD's operand type (bf16) beside its transpose (= the dgrad operand wt)
                blk["qkv_wb"] = blk["qkv"].w if dt == self.gdt else blk["qkv"].wt.t().contiguous()
            self.blocks.append(blk)
        self._ws = {}
        self.rowstats = opt.rowstats      # per tower, mutable (the guard sets it to 0; a test sets one tower to 0: the lock-stepped towers must re-align)
        self.serial = 0      # bumped by every forward: a backward checks that its forward was the tower's LAST one (see DualEncoder._ctx)

    def _stream_pool_shape(self, L):
        """True if the last block of this tower may run without K and V for sequences of L tokens (EngineOptions.stream_pool): a non-causal tower (uniform
        sequences, the pooled token attends to every row) in a 2-byte mode with the fp16 stream, and a shape the kernels take."""
        sp = self.spec
        return bool(self.opt.stream_pool and self.opt.pooled_last and self.dt != F32 and self.xdt == F16 and not sp.causal and len(self.blocks) > 1
                    and _lib.load().lpi_spool_attn_supported(int(L), sp.heads, sp.width) == 1)

    def _check_depth(self, prompts, depth):
        """model.py:191 indexes prompts[:, layer_id]: the reference raises IndexError past the stack; so do we (before any kernel)."""
        if prompts is not None and not (1 <= depth <= prompts.shape[-3]):
            raise ValueError(f"prompt depth {depth} outside 1..{prompts.shape[-3]} (layers of the prompt stack)")
        if depth > len(self.blocks):
            raise ValueError(f"prompt depth {depth} exceeds the tower's {len(self.blocks)} blocks")

    # ------------------------------------------------------------------ workspace
    def workspace(self, B: int, L: int, train: bool, cap: Optional[int] = None, packed=None):
        """Arena for B samples of L tokens.  It is allocated for `cap` (>= L) tokens per sample and reused for any shorter L (the
        text tower's L varies with the longest caption of the batch, see trim_token_ids): a call only re-binds L and the padded
        row count; the [M, *] buffers are used by their first M rows, the [B, H, L] ones as flat storage.
        packed: a PackedIds (ragged batch: sample b owns rows row_start[b] .. row_start[b+1]-1, M = sum of the lengths, L = the longest)."""
        def bind(ws, L):
            ws["B"], ws["Bp"] = B, _pad(B)
            ws["L"] = L
            ws["M"] = B * L if packed is None else packed.rows
            ws["Mp"] = _pad(ws["M"], 256)
            ws["rs"] = None if packed is None else packed.row_start_dev
            ws["pool_abs"] = None if packed is None else packed.pool_rows_dev
            # shared prefix (PackedIds(shared=n), include/lpi_hip.h): rows [0, n) are the positions every sample has in common; the attention backward
            # needs an f32 scratch for the samples' partial dK / dV of those keys
            ws["pre"] = pre = 0 if packed is None else int(getattr(packed, "shared", 0))
            if pre and train:
                need = ws["B"] * pre * 2 * self.spec.width
                if ws.get("shared_dkv") is None or ws["shared_dkv"].numel() < need:
                    ws["shared_dkv"] = torch.empty(need, dtype=torch.float32, device=self.device)
            return ws
        # an arena serves any batch of its mode that FITS it (B <= its batch capacity, L <= its token capacity): the odd last batch of an epoch
        # (DataLoader drop_last=False) and a shorter longest caption only re-bind row counts — every [M, *] buffer is used by its first rows,
        # the per-sample ones as flat storage; rows past the bound batch hold stale but finite data that no kernel reads into a live row.
        for (bcap, tr), ws in self._ws.items():
            if tr == train and bcap >= B and ws["Lcap"] >= L:
                return bind(ws, L)
        # one live arena per tower and MODE: the training arena and the evaluation one (a sixth of its size) stay side by side, so that the
        # train / eval alternation of an epoch loop does not free and re-allocate gigabytes; a LARGER batch of the same mode replaces its arena
        key = (B, train)
        for k in [k for k in self._ws if k[1] == train]:
            del self._ws[k]
        d, H, nl = self.spec.width, self.spec.heads, self.spec.layers
        POOLED_LAST = self.opt.pooled_last
        Lreal, L = L, max(L, cap or L)
        Mp = _pad(B * L, 256)      # whole 256x256 GEMM tiles (the 128x128 kernel takes any multiple of 128)
        Bp = _pad(B)
        T, TX, TG = _TORCH_DT[self.dt], _TORCH_DT[self.xdt], _TORCH_DT[self.gdt]
        TU = TG if self.dt == F16 else T      # the saved QuickGELU DERIVATIVE gelu'(u) ("u" buffers): read by the backward only (gemm_epilogue.h, AuxT)
        dev = self.device
        z = lambda *s, dtype=torch.float32: torch.zeros(*s, dtype=dtype, device=dev)  # noqa: E731
        keep = nl if train else 1
        ws = {
            "B": B, "L": L, "Mp": Mp, "Lcap": L,
            "x": [z(Mp, d, dtype=TX) for _ in range(nl + 1 if train else 2)],
            "xmid": [z(Mp, d, dtype=TX) for _ in range(keep)],
            "qkv": [z(Mp, 3 * d, dtype=T) for _ in range(keep)],
            "ctx": [z(Mp, d, dtype=T) for _ in range(keep)],
            "lse": [z(B + 1, H, L) for _ in range(keep)],      # + 1: the shared sequence of a shared-prefix batch (sample index B)
            "u": [z(Mp, 4 * d, dtype=TU) for _ in range(keep)] if train else [None],
            # per layer, per LayerNorm: mean[Mp] | rstd[Mp] | c1[<= 4d] (the LN operand block of the LN-fold GEMM epilogues, include/lpi_hip.h)
            # (+ Mp: a row CHUNK of the MLP passes the block shifted by its first row, and reads c1 shifted by as much: a copy of c1 sits there)
            "lnblk": [z(2, 3 * Mp + 4 * d) for _ in range(nl)],
            "h": z(Mp, d, dtype=T),
            "g": z(Mp, 4 * d, dtype=T),
            # slot sums of the row statistics a residual GEMM's epilogue leaves (LPI_EPI_RES_ROWSTATS): d/128 slots x (sum, sum of squares) x Mp
            "rstat": z(2 * max(d // 128, 1), Mp),
            # the LAST block's MLP runs on the B pooled rows only (exact: the heads read nothing else of its output)
            "Bp": Bp, "c_xmid": z(Bp, d), "c_h": z(Bp, d, dtype=T), "c_g": z(Bp, 4 * d, dtype=T),
            "c_u": z(Bp, 4 * d, dtype=TU) if train else None, "c_xout": z(Bp, d), "c_stat": z(2, Bp),
            # ... and so do its query, attention row and out_proj (K and V still cover every token)
            "c_q": z(Bp, d, dtype=T), "c_ctx": z(Bp, d, dtype=T), "c_lse": z(B * H), "c_xin": z(Bp, d), "c_stat1": z(2, Bp),
            # the last block without K and V (EngineOptions.stream_pool): qt | hbar | dhbar | dqt, f32 [B, H, d] each, and the scores' log-sum-exp
            "sp_scratch": z(4 * B * H * d) if self._stream_pool_shape(L) else None, "sp_lse": z(B * H),
        }
        if train:
            ws.update({
                "dx": z(Mp, d) if self.dt == F32 else None,      # bf16 mode: the bf16 stream dxT is the only gradient stream
                "dh": z(Mp, d, dtype=TG), "dctx": z(Mp, d, dtype=TG), "dqkv": z(Mp, 3 * d, dtype=TG),
                "delta": z(B + 1, H, L),
                "dxT": z(Mp, d, dtype=TG) if self.dt != F32 else None,
                "c_dx": z(Bp, d), "c_dxT": z(Bp, d, dtype=TG) if self.dt != F32 else None, "c_dh": z(Bp, d, dtype=TG),
                "c_dctx": z(Bp, d, dtype=TG), "c_dq": z(Bp, d, dtype=TG),
                # the first block's prompt-row backward: up to 32 prompt rows per sample, packed
                "p_dqkv": z(_pad(B * 32, 256), 3 * d, dtype=TG), "p_dh": z(_pad(B * 32, 256), d, dtype=TG),
                # the [*, 4d] MLP buffers double as dL/dg in the backward: a view in the gradient type (same element size)
                "du": None, "c_du": None,
            })
        if train:
            ws["du"], ws["c_du"] = ws["g"].view(TG), ws["c_g"].view(TG)
        # "stat"[i] = (ln1 mean, ln1 rstd, ln2 mean, ln2 rstd) of layer i: views into the LN blocks
        ws["stat"] = [(b[0, :Mp], b[0, Mp:2 * Mp], b[1, :Mp], b[1, Mp:2 * Mp]) for b in ws["lnblk"]]
        ws["ln_ld"] = Mp
        for i, blk in enumerate(self.blocks):
            if "qkv_ln" in blk:
                c1 = blk["qkv_ln"].c1
                if i == nl - 1 and POOLED_LAST:      # the last block's in_proj GEMM covers K and V only (rows d.. of the weight)
                    c1 = c1[d:]
                ws["lnblk"][i][0, 2 * Mp:2 * Mp + c1.numel()].copy_(c1)
                ws["lnblk"][i][1, 2 * Mp:2 * Mp + 4 * d].copy_(blk["fc_ln"].c1)
        self._ws[key] = ws
        return bind(ws, Lreal)

    # ------------------------------------------------------------------ forward
    def ln1_stats_out(self, ws):
        """(mean, rstd) of the first block's ln_1 if the front end should leave them there (its rows pass through a wave: lpi_vis_assemble_fwd /
        lpi_txt_embed_fwd take them as out_mean / out_rstd), else (None, None): no statistics pass over x_0 then.  Pass ln1_ready=True to forward_gen."""
        if self.rowstats >= 2 and self.blocks and "qkv_ln" in self.blocks[0] and _ln_fold_ok(ws["Mp"], self.spec.width):
            st = ws["stat"][0]
            return st[0], st[1]
        return None, None

    def forward_gen(self, ws, prompts=None, prompt_bstride=0, depth=1, train=True, pool_idx=None, ln1_ready=False):
        """GENERATOR (see GemmReq): runs the blocks over ws['x'][0], yielding its GEMMs; returns the POOLED rows of the output residual stream, [Bp, d] f32: row b is token
        pool_idx[b] of sample b (None: token 0 = CLS).  Only those rows are ever read by the heads (model.py:255,
        prompt_learner.py:61), so the last block's MLP is evaluated on them alone.

        prompts: f32 tensor whose element (b, layer, p, :) sits at  b*prompt_bstride + (layer*P + p)*d."""
        sp, dt, xdt, s = self.spec, self.dt, self.xdt, _stream()
        POOLED_LAST, LN_FOLD = self.opt.pooled_last, self.opt.ln_fold
        d, H = sp.width, sp.heads
        B, L, Mp, M = ws["B"], ws["L"], ws["Mp"], ws["M"]
        rs = ws["rs"]                                  # ragged batch: row starts (device int32 [B+1]); None = B x L rows
        # kernels that address ONE token per sample: (L, token index) or, ragged, (0, absolute row) — see include/lpi_hip.h
        Lx, pidx = (L, pool_idx) if rs is None else (0, ws["pool_abs"])
        P = prompts.shape[-2] if prompts is not None else 0
        pre = ws.get("pre", 0)                         # shared prefix: the prompt rows exist ONCE, as rows 1 .. P of "a batch of one sample of `pre` rows"
        Bq, Lq, rsq = (1, pre, None) if pre else (B, L, rs)
        self._check_depth(prompts, depth)
        self.serial += 1
        have_ln1 = bool(ln1_ready)      # ln_1's statistics of the coming block already written (by the front end / the previous block's c_proj epilogue)
        for i, blk in enumerate(self.blocks):
            lt = "last" if i == len(self.blocks) - 1 else i      # GEMM tag: towers of different depth pair layer i with layer i, last with last
            k = i if train else 0
            x_in = ws["x"][i if train else i % 2]
            x_out = ws["x"][i + 1 if train else (i + 1) % 2]
            xmid, qkv, ctx, lse, u, st = ws["xmid"][k], ws["qkv"][k], ws["ctx"][k], ws["lse"][k], ws["u"][k], ws["stat"][i]
            if prompts is not None and 0 < i < depth:      # model.py:189-193 with the intended guard
                # the rows it rewrites get their ln_1 statistics from the same kernel (the epilogue's are of their old contents)
                so = (st[0], st[1]) if have_ln1 else (None, None)
                yield RowReq(f"{lt}.padd", [_lib.row_job(_lib.ROWOP_PROMPT_ADD, B=Bq, L=Lq, row_start=rsq, P=P, d=d, dt_a=xdt, out=x_in,
                                                         a=prompts.view(-1)[i * P * d:], bstride=prompt_bstride, mean=so[0], rstd=so[1])], optional=True)
            # LayerNorm folded into the GEMM behind it (LnLinear): a statistics pass over the stream, then the GEMM reads the stream itself
            fold = "qkv_ln" in blk and _ln_fold_ok(Mp, d)
            lnb, ln_ld = ws["lnblk"][i], ws["ln_ld"]
            rowstats = self.rowstats if (fold and d % 128 == 0) else 0
            if fold and have_ln1:
                pass      # the previous block's c_proj left this LayerNorm's statistics (finalised behind it)
            elif fold:
                yield LnReq(f"{lt}.ln1", "fwd", (dt, xdt), M, d, x_in, d, None, None, None, 0, st[0], st[1], optional=True)
            else:
                yield LnReq(f"{lt}.ln1", "fwd", (dt, xdt), M, d, x_in, d, blk["ln_1.w"], blk["ln_1.b"], ws["h"], d, st[0], st[1], optional=True)
            if i == len(self.blocks) - 1 and POOLED_LAST:
                # last block: Q / softmax row / out_proj / MLP for the pooled token only; K and V for every token — or, with stream_pool, not at all
                Bp, cst, cst1 = ws["Bp"], ws["c_stat"], ws["c_stat1"]
                wq, bq = blk["qkv"].w, blk["qkv"].b
                spool = rs is None and not pre and ws.get("sp_scratch") is not None and self._stream_pool_shape(L)
                if spool:
                    yield SoloReq(f"{lt}.kv")      # no K / V projection: the text tower's (same tag) goes out alone
                elif fold:
                    ql = blk["qkv_ln"]
                    yield GemmReq(f"{lt}.kv", F16, x_in, ql.w[d:], qkv[:, d:], Mp, 2 * d, d, bias=ql.c2[d:], residual=lnb[0], ldr=ln_ld, epi=EPI_LN, m_real=M)
                else:
                    yield GemmReq(f"{lt}.kv", dt, ws["h"], wq[d:], qkv[:, d:], Mp, 2 * d, d, bias=bq[d:], m_real=M)
                # ln_1 of the pooled rows and the rows themselves (f32: the residual operand of the pooled out_proj) in one job
                yield RowReq(f"{lt}.pln1", [_lib.row_job(_lib.ROWOP_POOL_LN_FWD, B=B, L=Lx, d=d, dt_a=xdt, dt_b=dt, a=x_in, idx=pidx, gamma=blk["ln_1.w"],
                                                         beta=blk["ln_1.b"], out=ws["c_h"], ld_c=d, mean=cst1[0], rstd=cst1[1], out2=ws["c_xin"])])
                yield GemmReq(f"{lt}.cq", dt, ws["c_h"], wq[:d], ws["c_q"], Bp, d, d, bias=bq[:d], m_real=B)
                if spool:
                    x_sp, st_sp = x_in, st      # (bound now: the generator's loop variables move on)
                    yield SoloReq(f"{lt}.pattn", lambda: call("lpi_spool_attn_fwd", dt, B, L, H, ws["c_q"], d, wq, d, blk["qkv_wT"], 3 * d, bq, x_sp, d, st_sp[0], st_sp[1], blk["ln_1.w"],
                                                              blk["ln_1.b"], ws["sp_scratch"], ws["sp_lse"], ws["c_ctx"], d, _stream()))
                else:
                    yield PoolAttnReq(f"{lt}.pattn", dt, False, B=B, L=L, row_start=rs, H=H, q=ws["c_q"], ldq=d, qkv=qkv, ldqkv=3 * d, idx=pool_idx, ctx=ws["c_ctx"],
                                      ldctx=d, lse=ws["c_lse"], causal=int(sp.causal), shared_rows=pre)
                yield GemmReq(f"{lt}.cout", dt, ws["c_ctx"], blk["out"].w, ws["c_xmid"], Bp, d, d, bias=blk["out"].b, residual=ws["c_xin"], m_real=B)
                yield RowReq(f"{lt}.pln2", [_lib.row_job(_lib.ROWOP_POOL_LN_FWD, B=B, L=1, d=d, dt_a=F32, dt_b=dt, a=ws["c_xmid"], gamma=blk["ln_2.w"],
                                                         beta=blk["ln_2.b"], out=ws["c_h"], ld_c=d, mean=cst[0], rstd=cst[1])])
                yield GemmReq(f"{lt}.cfc", dt, ws["c_h"], blk["fc"].w, ws["c_g"], Bp, 4 * d, d, bias=blk["fc"].b, epi=EPI_QUICKGELU, aux=ws["c_u"], m_real=B)
                yield GemmReq(f"{lt}.cproj", dt, ws["c_g"], blk["proj"].w, ws["c_xout"], Bp, d, 4 * d, bias=blk["proj"].b, residual=ws["c_xmid"], m_real=B)
                return ws["c_xout"]
            if fold:
                ql = blk["qkv_ln"]
                yield GemmReq(f"{lt}.qkv", F16, x_in, ql.w, qkv, Mp, 3 * d, d, bias=ql.c2, residual=lnb[0], ldr=ln_ld, epi=EPI_LN, m_real=M)
            else:
                yield GemmReq(f"{lt}.qkv", dt, ws["h"], blk["qkv"].w, qkv, Mp, 3 * d, d, bias=blk["qkv"].b, m_real=M)
            yield AttnFwdReq(f"{lt}.attn", dt, B, L, rs, H, qkv, 3 * d, ctx, d, lse, int(sp.causal), pre, self.qkv_lay if blk["grouped"] else None)
            ln2_stats = rowstats >= 1 and LN_FOLD >= 2 and not (i == len(self.blocks) - 1 and POOLED_LAST)
            if ln2_stats:      # x + attn(..) and the slot sums of its rows in one epilogue; ln_2's mean / rstd from them
                yield GemmReq(f"{lt}.out", dt, ctx, blk["out"].w, xmid, Mp, d, d, bias=blk["out"].b, residual=x_in, m_real=M, epi=EPI_RES_ROWSTATS,
                              aux=ws["rstat"])
                yield StatFinReq(f"{lt}.fin2", M, d, ws["rstat"], ws["rstat"].stride(0), st[2], st[3])
            else:
                yield GemmReq(f"{lt}.out", dt, ctx, blk["out"].w, xmid, Mp, d, d, bias=blk["out"].b, residual=x_in, m_real=M)
            # the next block's ln_1 statistics come out of c_proj's epilogue; rows that deep prompts rewrite first get theirs from prompt_add (above)
            nxt = i + 1
            have_ln1 = rowstats >= 2 and nxt < len(self.blocks) and "qkv_ln" in self.blocks[nxt]
            proj_kw = dict(bias=blk["proj"].b, residual=xmid, m_real=M)
            if have_ln1:
                proj_kw.update(epi=EPI_RES_ROWSTATS, aux=ws["rstat"])
            if fold and LN_FOLD >= 2:
                fl = blk["fc_ln"]
                if not ln2_stats:
                    yield LnReq(f"{lt}.ln2", "fwd", (dt, xdt), M, d, xmid, d, None, None, None, 0, st[2], st[3], optional=True)
                yield GemmReq(f"{lt}.fc", F16, xmid, fl.w, ws["g"], Mp, 4 * d, d, bias=fl.c2, residual=lnb[1], ldr=ln_ld, epi=EPI_LN_QUICKGELU, aux=u, m_real=M)
            else:
                yield LnReq(f"{lt}.ln2", "fwd", (dt, xdt), M, d, xmid, d, blk["ln_2.w"], blk["ln_2.b"], ws["h"], d, st[2], st[3], optional=True)
                yield GemmReq(f"{lt}.fc", dt, ws["h"], blk["fc"].w, ws["g"], Mp, 4 * d, d, bias=blk["fc"].b, epi=EPI_QUICKGELU, aux=u, m_real=M)
            yield GemmReq(f"{lt}.proj", dt, ws["g"], blk["proj"].w, x_out, Mp, d, 4 * d, **proj_kw)
            if have_ln1:
                nst = ws["stat"][nxt]
                yield StatFinReq(f"{lt}.fin1", M, d, ws["rstat"], ws["rstat"].stride(0), nst[0], nst[1])
        call("lpi_gather_rows", xdt, B, Lx, d, x_out, pidx, ws["c_xout"], s)      # pooled_last = False: full last block, then pool
        return ws["c_xout"]

    def forward(self, ws, prompts=None, prompt_bstride=0, depth=1, train=True, pool_idx=None):
        """forward_gen driven alone: every GEMM launched as it is yielded."""
        return run_alone(self.forward_gen(ws, prompts, prompt_bstride, depth, train, pool_idx))

    # ------------------------------------------------------------------ backward (dgrad only)
    def backward_gen(self, ws, prompts=None, depth=1, dprompts=None, pool_idx=None, acc=0):
        """GENERATOR (see GemmReq).  ws['c_dx'] (f32 [Bp, d]) [and ws['c_dxT']] hold dL/d(pooled output rows) on entry; ws['dx'] holds dL/dx_0 on exit.
        dprompts: f32 [Lyr, P, d]; rows of layers 1..depth-1 receive the batch-summed deep-prompt gradients (acc = 1: ADDED to what the rows hold — the
        alignment-loss gradient the step seeded them with, DualEncoder.seed_prompt_grads)."""
        sp, dt, xdt, s = self.spec, self.gdt, self.xdt, _stream()        # dt: the BACKWARD's operand / storage type from here on
        adt = F16 if self.dt == F16 else dt       # attention backward: F16 = "saved q, k, v, ctx are fp16; gradients and operands bf16"
        POOLED_LAST, L0_PROMPT_ROWS = self.opt.pooled_last, self.opt.l0_prompt_rows
        d, H = sp.width, sp.heads
        B, L, Mp, M = ws["B"], ws["L"], ws["Mp"], ws["M"]
        rs = ws["rs"]                                  # ragged batch: row starts (device int32 [B+1]); None = B x L rows
        # kernels that address ONE token per sample: (L, token index) or, ragged, (0, absolute row) — see include/lpi_hip.h
        Lx, pidx = (L, pool_idx) if rs is None else (0, ws["pool_abs"])
        dx, dh, dctx, dqkv = ws["dx"], ws["dh"], ws["dctx"], ws["dqkv"]
        dxT = ws["dxT"] if dt != F32 else dx          # the stream the dgrad GEMMs read; in bf16 mode also the accumulator
        P = prompts.shape[-2] if prompts is not None else 0
        pre = ws.get("pre", 0)                         # shared prefix (forward_gen): the gradient stream's rows 1 .. P ARE the batch sums
        Bq, Lq, rsq = (1, pre, None) if pre else (B, L, rs)
        self._check_depth(prompts, depth)
        for i in reversed(range(len(self.blocks))):
            blk = self.blocks[i]
            # GEMM tag: the backward pairs the towers' layers by their distance from the END (both start at their last block)
            lt = "last" if i == len(self.blocks) - 1 else f"r{len(self.blocks) - 1 - i}"
            x_in, xmid, qkv, ctx, lse, u, st = ws["x"][i], ws["xmid"][i], ws["qkv"][i], ws["ctx"][i], ws["lse"][i], ws["u"][i], ws["stat"][i]
            if i == len(self.blocks) - 1 and not POOLED_LAST:
                call("lpi_zero", dxT, dxT.numel() * dxT.element_size(), s)
                call("lpi_scatter_rows", dt, B, Lx, d, ws["c_dx"], pidx, dx, None if dt == F32 else dxT, s)
            if i == len(self.blocks) - 1 and POOLED_LAST:
                # last block: MLP backward on the pooled rows, then scatter into the (zeroed) full-size gradient stream
                Bp, cst = ws["Bp"], ws["c_stat"]
                c_dx = ws["c_dx"]
                c_dxT = ws["c_dxT"] if dt != F32 else c_dx
                yield GemmReq(f"{lt}.cdproj", dt, c_dxT, blk["proj"].wt, ws["c_du"], Bp, 4 * d, d, epi=EPI_DQUICKGELU, aux=ws["c_u"], m_real=B)
                yield GemmReq(f"{lt}.cdfc", dt, ws["c_du"], blk["fc"].wt, ws["c_dh"], Bp, d, 4 * d, m_real=B)
                yield RowReq(f"{lt}.cdln2", [_lib.row_job(_lib.ROWOP_LN_BWD, B=B, d=d, dt_a=dt, dt_b=dt, a=ws["c_dh"], ld_a=d, b=ws["c_xmid"], ld_b=d,
                                                          gamma=blk["ln_2.w"], mean_in=cst[0], rstd_in=cst[1], out=c_dx, out2=None if dt == F32 else c_dxT,
                                                          ld_c=d, flag=1)])
                # ... and, on the same pooled rows, the attention branch:
                # attention branch of the pooled rows: dctx, dQ on B rows; dK, dV on every row; d(LN1 out) = dKV.Wkv (+ dQ.Wq at the pooled rows)
                wqt = blk["qkv"].wt
                yield GemmReq(f"{lt}.cdout", dt, c_dxT, blk["out"].wt, ws["c_dctx"], Bp, d, d, m_real=B)
                if rs is None and not pre and ws.get("sp_scratch") is not None and self._stream_pool_shape(L):
                    # the forward ran without K and V (forward_gen): d LN1(x_l) of every row and the pooled queries' gradient from two passes over the stream
                    yield SoloReq(f"{lt}.pdattn", lambda: call("lpi_spool_attn_bwd", B, L, H, blk["qkv_wb"], d, blk["qkv"].wt, 3 * d, x_in, d, st[0], st[1], blk["ln_1.w"],
                                                               ws["sp_scratch"], ws["sp_lse"], ws["c_dctx"], d, ws["c_dq"], d, dh, d, _stream()))
                    yield SoloReq(f"{lt}.dkv")
                else:
                    yield PoolAttnReq(f"{lt}.pdattn", adt, True, B=B, L=L, row_start=rs, H=H, q=ws["c_q"], ldq=d, qkv=qkv, ldqkv=3 * d, idx=pool_idx,
                                      dctx=ws["c_dctx"], lddctx=d, lse=ws["c_lse"], dq=ws["c_dq"], lddq=d, dqkv=dqkv, lddqkv=3 * d, causal=int(sp.causal),
                                      shared_rows=pre, shared_dkv=ws.get("shared_dkv") if pre else None)
                    yield GemmReq(f"{lt}.dkv", dt, dqkv[:, d:], wqt[:, d:], dh, Mp, d, 2 * d, m_real=M)
                yield GemmReq(f"{lt}.cdq", dt, ws["c_dq"], wqt[:, :d], ws["c_dh"], Bp, d, d, m_real=B)
                yield RowReq(f"{lt}.sadd1", [_lib.row_job(_lib.ROWOP_SCATTER_ADD, B=B, L=Lx, d=d, dt_a=dt, a=ws["c_dh"], ld_a=d, idx=pidx, out=dh, ld_c=d)])
                # the gradient stream starts here: LN1's backward WRITES it (no zero-fill of the [M, d] stream), then the residual
                # path of the pooled rows is added
                yield LnReq(f"{lt}.dln1", "bwd", (dt, dt, xdt), M, d, dh, d, x_in, d, blk["ln_1.w"], st[0], st[1], dx, d,
                            None if dt == F32 else dxT, d, 0)
                yield RowReq(f"{lt}.sadd2", [_lib.row_job(_lib.ROWOP_SCATTER_ADD, B=B, L=Lx, d=d, dt_a=dt, a=c_dxT, ld_a=d, idx=pidx, out=dxT, ld_c=d)])
                if prompts is not None and dprompts is not None and 0 < i < depth:
                    yield RowsSumReq(f"{lt}.psum", dt, Bq, Lq, rsq, 1, P, d, dxT, dprompts[i], acc)
                continue
            if not (i == len(self.blocks) - 1 and POOLED_LAST):
                du = ws["du"]
                yield GemmReq(f"{lt}.dproj", dt, dxT, blk["proj"].wt, du, Mp, 4 * d, d, epi=EPI_DQUICKGELU, aux=u, m_real=M)          # d c_proj, * gelu'
                yield GemmReq(f"{lt}.dfc", dt, du, blk["fc"].wt, dh, Mp, d, 4 * d, m_real=M)                                         # d c_fc
                yield LnReq(f"{lt}.dln2", "bwd", (dt, dt, xdt), M, d, dh, d, xmid, d, blk["ln_2.w"], st[2], st[3], dx, d,
                            None if dt == F32 else dxT, d, 1)      # dx is None in bf16 mode: dxT accumulates in place
            yield GemmReq(f"{lt}.dout", dt, dxT, blk["out"].wt, dctx, Mp, d, d, m_real=M)                                         # d out_proj
            l0_rows = i == 0 and L0_PROMPT_ROWS and prompts is not None and 0 < P <= 32 and len(self.blocks) > 1
            # first block: only dQ / dK / dV of the prompt rows 1 .. P are read below -> the attention backward skips the row blocks behind them
            if pre:
                call("lpi_attn_bwd_shared", adt, B, L, rs, pre, (1 + P) if l0_rows else L, H, qkv, 3 * d, ctx, d, dctx, d, lse, ws["delta"], dqkv, 3 * d,
                     ws["shared_dkv"], s)
            elif blk["grouped"]:
                hs_, vs_, chs_ = self.qkv_lay
                lay = (ctypes.c_int32 * 6)(hs_, vs_, hs_, vs_, chs_, chs_)
                call("lpi_attn_bwd_layout", adt, B, L, (1 + P) if l0_rows else L, H, qkv, 3 * d, ctx, d, dctx, d, lse, ws["delta"], dqkv, 3 * d,
                     ctypes.cast(lay, ctypes.c_void_p), s)
            else:
                call("lpi_attn_bwd_prefix", adt, B, L, rs, (1 + P) if l0_rows else L, H, qkv, 3 * d, ctx, d, dctx, d, lse, ws["delta"], dqkv, 3 * d,
                     int(sp.causal), s)
            if l0_rows:
                # first block: dL/dx_0 is read at the prompt rows 1..P only (vis_assemble_bwd / rows_sum_over_batch below; the patch,
                # CLS and token embeddings are frozen) -> in_proj dgrad and LN1 backward on the packed B*P prompt rows.  The residual
                # path of those rows is already in the stream; every other row of it is left without this block's attention term.
                pq, ph = ws["p_dqkv"], ws["p_dh"]
                esz = 4 if dt == F32 else 2
                yield RowReq(f"{lt}.gath", [_lib.row_job(_lib.ROWOP_GATHER_BATCH_ROWS, B=Bq, L=Lq, row_start=rsq, row0=1, P=P, d=3 * d * esz // 16, a=dqkv,
                                                         ld_a=3 * d * esz // 16, out=pq, ld_c=3 * d * esz // 16)])
                yield GemmReq(f"{lt}.dqkv_p", dt, pq, blk["qkv"].wt, ph, _pad(Bq * P, 256), d, 3 * d, m_real=Bq * P)
                if dt == BF16 and xdt == F16 and d % 8 == 0:      # the 16-byte half-wave kernel, with the towers' launches paired
                    yield RowReq(f"{lt}.dln1p", [_lib.row_job(_lib.ROWOP_LN_BWD_ROWS_H16, B=Bq, L=Lq, row_start=rsq, row0=1, P=P, d=d, a=ph, ld_a=d, b=x_in, ld_b=d,
                                                              gamma=blk["ln_1.w"], mean_in=st[0], rstd_in=st[1], out2=dxT, ld_c=d, flag=1)])
                else:
                    call("lpi_layernorm_bwd_rows_varlen", dt, dt, xdt, Bq, Lq, rsq, 1, P, d, ph, d, x_in, d, blk["ln_1.w"], st[0], st[1], dx, d,
                         None if dt == F32 else dxT, d, 1, s)
                continue
            yield GemmReq(f"{lt}.dqkv", dt, dqkv, blk["qkv"].wt, dh, Mp, d, 3 * d, m_real=M)                                      # d in_proj
            yield LnReq(f"{lt}.dln1", "bwd", (dt, dt, xdt), M, d, dh, d, x_in, d, blk["ln_1.w"], st[0], st[1], dx, d,
                        None if dt == F32 else dxT, d, 1)
            if prompts is not None and dprompts is not None and 0 < i < depth:
                yield RowsSumReq(f"{lt}.psum", dt, Bq, Lq, rsq, 1, P, d, dxT, dprompts[i], acc)
        return dxT

    def backward(self, ws, prompts=None, depth=1, dprompts=None, pool_idx=None, acc=0):
        """backward_gen driven alone."""
        return run_alone(self.backward_gen(ws, prompts, depth, dprompts, pool_idx, acc))


class DualEncoder:
    """CLIP ViT + text transformer with prompt side inputs, forward and dgrad backward, on one MI355X."""

    def __init__(self, cfg: ClipConfig, state_dict: dict, dtype: str = "f32", device="cuda:0", n_ctx: int = 16, options: Optional[EngineOptions] = None):
        """options: an EngineOptions (None = EngineOptions.from_env(): the defaults with the three documented environment fall-backs applied)."""
        self.cfg = cfg
        self.opt = options if options is not None else EngineOptions.from_env()
        self.device = torch.device(device)
        _require_gpu(self.device)
        _lib.load()
        self.dt = _DT[dtype]
        self.gdt = _grad_dtype(self.dt)
        self.dtype_name = {F32: "f32", BF16: "bf16", F16: "f16"}[self.dt]
        self.n_ctx = n_ctx
        dev, dt = self.device, self.dt
        t = lambda k: (state_dict[k] if torch.is_tensor(state_dict[k]) else torch.as_tensor(np.asarray(state_dict[k]))).to(  # noqa: E731
            device=dev, dtype=torch.float32).contiguous()
        self.vis = Tower(state_dict, "visual.transformer.", TowerSpec(cfg.vision_width, cfg.vision_heads, cfg.vision_layers, False), dt, dev, self.opt)
        self.txt = Tower(state_dict, "transformer.", TowerSpec(cfg.transformer_width, cfg.transformer_heads, cfg.transformer_layers, True), dt, dev, self.opt)
        ps = cfg.vision_patch_size
        self.kp = _pad(3 * ps * ps, 32 if dt == F32 else 64)
        self.conv = Linear(t("visual.conv1.weight").reshape(cfg.vision_width, -1), None, dt, dev, k_pad=self.kp)
        self.cls = t("visual.class_embedding")
        self.vpos = t("visual.positional_embedding")
        self.ln_pre = (t("visual.ln_pre.weight"), t("visual.ln_pre.bias"))
        self.ln_post = (t("visual.ln_post.weight"), t("visual.ln_post.bias"))
        self.vproj = Linear(t("visual.proj").t().contiguous(), None, dt, dev)        # Linear weight [E, d]
        self.tok = t("token_embedding.weight")
        self.tpos = t("positional_embedding")
        self.ln_final = (t("ln_final.weight"), t("ln_final.bias"))
        self.tproj = Linear(t("text_projection").t().contiguous(), None, dt, dev)
        self.logit_scale = t("logit_scale")
        self.logit_scale_exp = float(math.exp(float(np.asarray(state_dict["logit_scale"] if not torch.is_tensor(state_dict["logit_scale"]) else state_dict["logit_scale"].cpu()))))
        self._head_ws = {}
        # guard of the one-sweep LayerNorm statistics (EngineOptions.rowstat_guard): a device counter, and a word of PINNED host memory the kernels flag
        self._guard = torch.zeros(1, dtype=torch.int32, device=dev) if self.opt.rowstat_guard else None
        self._guard_flag = torch.zeros(1, dtype=torch.int32).pin_memory() if self.opt.rowstat_guard else None
        self.rowstat_guard_tripped = 0          # rows counted when the guard switched the towers to the statistics pass (0 = never)

    # ------------------------------------------------------------------ uint8 pixels
    def pixel_lut(self, mean=None, std=None):
        """f32 [3, 256] on the device: lut[c][v] = ((v / 255) - mean[c]) / std[c], evaluated with the torch operations of the loader's ToTensor +
        Normalize (utils/data.py:201-204 -> lpi_amd.retrieval.utils.data._to_normalised_tensor: .float().div_(255.0), then (a - mean) / std in f32), so
        that lpi_patchify_u8 reproduces the f32 pipeline bit for bit.  Default statistics: ImageNet's, as the reference's transforms use."""
        key = (tuple(mean) if mean is not None else None, tuple(std) if std is not None else None)
        lut = self.__dict__.setdefault("_pixel_luts", {}).get(key)
        if lut is None:
            lut = self._pixel_luts[key] = make_pixel_lut(mean, std).to(self.device)
        return lut

    # ------------------------------------------------------------------ one-sweep statistics guard
    def _guard_begin(self):
        """Start of a forward: if an earlier forward's kernels have flagged the host word (a plain read of this process's own memory: no copy, no event,
        no wait), drop both towers to the two-sweep statistics pass; then register counter and flag for the kernels this thread is about to launch."""
        lib = _lib.load()
        if self._guard is None:
            return
        if int(self._guard_flag[0]) != 0 and (self.vis.rowstats or self.txt.rowstats):
            n = int(self._guard.item())          # the one synchronisation, once, on the way out of the fast path
            import warnings
            warnings.warn(f"lpi_amd: {n} residual-stream rows with |mean| > 8 std were seen by the one-sweep LayerNorm statistics (E[x^2] - mean^2 loses "
                          "digits there); both towers use the two-sweep statistics pass from now on (rowstats = 0 behaviour: exact, ~0.6 % slower)",
                          RuntimeWarning, stacklevel=3)
            self.vis.rowstats = self.txt.rowstats = 0
            self.rowstat_guard_tripped = max(n, 1)
        active = bool(self.vis.rowstats or self.txt.rowstats)
        rc = lib.lpi_rowstat_guard(self._guard.data_ptr() if active else None, self._guard_flag.data_ptr() if active else None)
        if rc != 0:
            raise _lib.LpiError(f"lpi_rowstat_guard failed with code {rc} (the pinned flag word is not device-accessible)")
        self._guard_depth = self.__dict__.get("_guard_depth", 0) + 1

    def _guard_end(self):
        """End of a forward (the second one's, when the towers run in lock step): take the registration back, so that no later launch of this thread —
        another engine's, a test's direct call — writes to a counter whose engine may be gone."""
        if self._guard is None:
            return
        self._guard_depth = max(0, self.__dict__.get("_guard_depth", 1) - 1)
        if self._guard_depth == 0:
            _lib.load().lpi_rowstat_guard(None, None)

    # ------------------------------------------------------------------ lanes
    def lane(self, i: int, stream=None):
        """Lane i > 0: an engine that SHARES the frozen weights but owns its workspace arena and HIP stream, so that
        micro-batches can be in flight concurrently (step.py); lane 0 is this engine on the caller's stream.
        stream: use this torch stream instead of creating a plain one."""
        if i == 0:
            return self
        lanes = self.__dict__.setdefault("_lanes", {})
        if i in lanes and stream is not None and lanes[i].stream is not stream:
            lanes[i].stream = stream
        if i not in lanes:
            import copy
            c = copy.copy(self)
            c.vis, c.txt = copy.copy(self.vis), copy.copy(self.txt)
            c.vis._ws, c.txt._ws, c._head_ws = {}, {}, {}
            c.__dict__.pop("_lanes", None)
            c.__dict__.pop("_side_stream", None)
            c.stream = stream if stream is not None else torch.cuda.Stream(device=self.device)
            lanes[i] = c
        return lanes[i]

    # ------------------------------------------------------------------ helpers
    def _head(self, tag, B, d):
        key = (tag, B)
        hw = self._head_ws.get(key)
        if hw is None:
            Bp, E, dev, T, TG = _pad(B), self.cfg.embed_dim, self.device, _TORCH_DT[self.dt], _TORCH_DT[self.gdt]
            z = lambda *s, dtype=torch.float32: torch.zeros(*s, dtype=dtype, device=dev)  # noqa: E731
            hw = {"pooled": z(Bp, d, dtype=T), "stat": z(2, Bp), "feat": z(Bp, E), "inv": z(Bp),
                  "dfeat": z(Bp, E), "dfeatT": z(Bp, E, dtype=TG) if self.dt != F32 else None, "dpooled": z(Bp, d),
                  "idx": torch.zeros(Bp, dtype=torch.int32, device=dev)}
            self._head_ws[key] = hw
        return hw

    @staticmethod
    def _prompt_args(prompts, B):
        """prompts: None | [Lyr,P,d] (broadcast over the batch, slinet.py:119) | [B,Lyr,P,d] (per sample, slinet.py:215)."""
        if prompts is None:
            return None, 0, 0
        if prompts.dim() == 4 and prompts.stride(0) == 0:
            prompts = prompts[0]
        if prompts.dim() == 3:
            return prompts.contiguous().float(), 0, prompts.shape[-2]
        if prompts.shape[0] != B:
            raise ValueError("per-sample prompts must have batch dimension B")
        p = prompts.contiguous().float()
        return p, p.stride(0), p.shape[-2]

    # ------------------------------------------------------------------ vision
    def _stale(self, tower, serial, what):
        if serial != tower.serial:
            raise _lib.LpiError(
                f"{what}: the {'vision' if tower is self.vis else 'text'} tower ran another forward since the forward this backward belongs to; "
                "its workspace arena (saved activations) has been overwritten.  Run forward -> backward 1:1 per engine (use "
                "DualEncoder.lane(i) for micro-batches that must be in flight together).")

    def encode_image(self, image, prompts=None, depth=1, train=False, normalise=True, return_ctx=False):
        """image [B,3,R,R] f32 (device) -> features [B,E] f32 (L2-normalised like slinet.py:122 unless normalise=False).
        return_ctx: also return the backward context (held by the autograd node, functional.EncodeImageFn)."""
        out, ctx = run_alone(self.encode_image_gen(image, prompts, depth, train, normalise))
        return (out, ctx) if return_ctx else out

    def encode_image_gen(self, image, prompts=None, depth=1, train=False, normalise=True):
        """GENERATOR form of encode_image (yields its GEMMs, see GemmReq); returns (features, backward context)."""
        cfg, dt, s = self.cfg, self.dt, _stream()
        self._guard_begin()
        B = image.shape[0]
        u8 = image.dtype == torch.uint8      # decoded pixels: ToTensor + Normalize happen inside the im2col kernel (pixel_lut)
        image = image.to(device=self.device).contiguous() if u8 else image.to(device=self.device, dtype=torch.float32).contiguous()
        pr, pbs, P = self._prompt_args(prompts, B)
        G2, d = cfg.n_patches, cfg.vision_width
        L = 1 + P + G2
        ws = self.vis.workspace(B, L, train)
        fe = ws.get("front")
        if fe is None:
            rows = _pad(B * G2, 256)
            fe = {"cols": torch.zeros(rows, self.kp, dtype=_TORCH_DT[dt], device=self.device),
                  "pe": torch.zeros(rows, d, device=self.device), "stat": torch.zeros(2, ws["Mp"], device=self.device)}
            ws["front"] = fe
        if u8:
            call("lpi_patchify_u8", dt, B, cfg.image_resolution, cfg.vision_patch_size, image, self.pixel_lut(), fe["cols"], self.kp, s)
        else:
            call("lpi_patchify", dt, B, cfg.image_resolution, cfg.vision_patch_size, image, fe["cols"], self.kp, s)
        yield GemmReq(None, dt, fe["cols"], self.conv.w, fe["pe"], fe["cols"].shape[0], d, self.kp, m_real=B * G2)
        call("lpi_vis_assemble_fwd", self.vis.xdt, B, G2, P, d, fe["pe"], d, self.cls, self.vpos, pr, pbs, self.ln_pre[0], self.ln_pre[1],
             ws["x"][0], fe["stat"][0], fe["stat"][1], *self.vis.ln1_stats_out(ws), s)
        xo = yield from self.vis.forward_gen(ws, pr, pbs, depth, train, None, ln1_ready=self.vis.ln1_stats_out(ws)[0] is not None)      # pooled (CLS) rows [Bp, d]
        hw = self._head("v", B, d)
        out = yield from self._head_fwd_gen(hw, xo, self.ln_post, self.vproj, B, d, normalise)
        self._guard_end()
        ctx = (ws, pr, pbs, P, depth, B, L, out, self.vis.serial)
        self._vis_ctx = ctx
        return out, ctx

    def _head_fwd_gen(self, hw, xo, ln, proj, B, d, normalise):
        """ln_post / ln_final on the pooled rows, the projection, the L2 normalisation (model.py:255-257, prompt_learner.py:57-63, slinet.py:122,133)."""
        cfg, dt = self.cfg, self.dt
        E_ = cfg.embed_dim
        yield RowReq("head.ln", [_lib.row_job(_lib.ROWOP_POOL_LN_FWD, B=B, L=1, d=d, dt_a=F32, dt_b=dt, a=xo, gamma=ln[0], beta=ln[1], out=hw["pooled"], ld_c=d,
                                              mean=hw["stat"][0], rstd=hw["stat"][1])])
        yield GemmReq("head", dt, hw["pooled"], proj.w, hw["feat"], hw["pooled"].shape[0], E_, d)
        out = torch.empty(B, E_, device=self.device)
        if normalise:
            yield RowReq("head.l2", [_lib.row_job(_lib.ROWOP_L2NORM_FWD, B=B, d=E_, a=hw["feat"], ld_a=E_, out=out, ld_c=E_, mean=hw["inv"])])
        else:
            out.copy_(hw["feat"][:B])
        return out

    def _head_bwd_gen(self, hw, ws, out, dout, ln, proj, B, d):
        """The backward of _head_fwd_gen: leaves dL/d(pooled output rows) in ws['c_dx'] (and its bf16 copy in ws['c_dxT'])."""
        dt, E_ = self.gdt, self.cfg.embed_dim
        dout = dout.contiguous().float()
        # L2-norm backward; in the 2-byte modes the same job leaves the bf16 operand of the dgrad GEMM (no separate cast launch)
        yield RowReq("dhead.l2", [_lib.row_job(_lib.ROWOP_L2NORM_BWD, B=B, d=E_, a=out, ld_a=E_, b=dout, ld_b=E_, mean_in=hw["inv"], out=hw["dfeat"], ld_c=E_,
                                               out2=None if dt == F32 else hw["dfeatT"], dt_b=BF16)])
        dfe = hw["dfeat"] if dt == F32 else hw["dfeatT"]
        yield GemmReq("dhead", dt, dfe, proj.wt, hw["dpooled"], dfe.shape[0], d, E_)
        yield RowReq("dhead.ln", [_lib.row_job(_lib.ROWOP_POOL_LN_BWD, B=B, L=1, d=d, dt_b=dt, a=hw["dpooled"], ld_a=d, b=ws["c_xout"], gamma=ln[0],
                                               mean_in=hw["stat"][0], rstd_in=hw["stat"][1], out=ws["c_dx"], out2=None if dt == F32 else ws["c_dxT"])])

    def _dprompts(self, ws, Lyr, P, d, depth, seeded=False):
        """The tower's prompt-gradient buffer [Lyr, P, d] f32, kept in the workspace.  Plain mode: rows of layers < depth are OVERWRITTEN by every
        backward (the prompt-row sums write, they do not accumulate), the rows behind them are zero from the allocation on — no fill kernel in the
        step; re-zeroed if the depth shrinks or after a seeded use.  Seeded mode (seed_prompt_grads): the caller has written a gradient of ALL rows
        into it (the alignment loss's) and the backward ADDS the towers' rows to it.
        The buffer is PERSISTENT: the tensor encode_*_backward returns is this buffer and is overwritten by the engine's next backward — consume it (or
        clone it) before the next step.  The autograd Functions (functional.py) hand autograd a clone unless the step seeded the buffers."""
        key = (Lyr, P, d)
        ent = ws.get("dprompts")
        if ent is None or ent[0] != key:
            ent = ws["dprompts"] = [key, torch.zeros(Lyr, P, d, device=self.device), depth, False]
        elif not seeded and (depth < ent[2] or ent[3]):
            ent[1].zero_()
        ent[2], ent[3] = depth, seeded
        return ent[1]

    def seed_prompt_grads(self, vis_ctx, txt_ctx):
        """-> (dvis, dtxt): the two towers' prompt-gradient buffers [Lyr, P, d] for the NEXT backward of these contexts, to be filled by the caller with a
        gradient the towers' own should be ADDED to (step.train_step: the alignment loss writes its gradient there, so that no separate sum of the
        two gradients is needed).  One-shot: the next encode_*_backward of each tower accumulates onto the buffer and clears the request."""
        bufs = []
        for ctx, d in ((vis_ctx, self.cfg.vision_width), (txt_ctx, self.cfg.transformer_width)):
            ws, pr, P, depth = ctx[0], ctx[1], ctx[3], ctx[4]
            if pr is None:
                raise ValueError("seed_prompt_grads needs a forward with prompts")
            buf = self._dprompts(ws, pr.shape[-3], P, d, depth, seeded=True)
            ws["dprompts_seeded"] = True
            bufs.append(buf)
        return bufs[0], bufs[1]

    def encode_image_backward(self, dout, ctx=None):
        """dL/d(normalised features) [B,E] -> dL/d(prompts) [Lyr,P,d] summed over the batch.
        ctx: the context encode_image(return_ctx=True) returned (default: the engine's last forward)."""
        return run_alone(self.encode_image_backward_gen(dout, ctx))

    def encode_image_backward_gen(self, dout, ctx=None):
        cfg, dt, s = self.cfg, self.gdt, _stream()         # the backward's operand / storage type
        ws, pr, pbs, P, depth, B, L, out, serial = ctx if ctx is not None else self._vis_ctx
        self._stale(self.vis, serial, "encode_image_backward")
        d, E = cfg.vision_width, cfg.embed_dim
        hw = self._head("v", B, d)
        yield from self._head_bwd_gen(hw, ws, out, dout, self.ln_post, self.vproj, B, d)
        if pr is None:
            return None
        Lyr = pr.shape[-3]
        acc = 1 if ws.pop("dprompts_seeded", False) else 0
        dpr = self._dprompts(ws, Lyr, P, d, depth, seeded=bool(acc))
        yield from self.vis.backward_gen(ws, pr, depth, dpr, None, acc)
        # lpi_vis_assemble_bwd as its two kernels: ln_pre's backward on the prompt rows (this tower only), then the batch sum (paired with the text tower's)
        dx0 = ws["dx"] if dt == F32 else ws["dxT"]
        yield RowReq(None, [_lib.row_job(_lib.ROWOP_VIS_PROMPT_ROWS_BWD, B=B, L=L, P=P, d=d, dt_a=dt, out=dx0, a=pr, bstride=pbs, gamma=self.ln_pre[0],
                                         mean_in=ws["front"]["stat"][0], rstd_in=ws["front"]["stat"][1])])
        yield RowsSumReq("front.psum", dt, B, L, None, 1, P, d, dx0, dpr[0], acc)
        return dpr

    # ------------------------------------------------------------------ text
    def encode_text(self, ids, prompts=None, depth=1, train=False, use_ctx=True, normalise=True, return_ctx=False):
        """ids [B,77] int64 (device).  prompts as in encode_image; row 0 of the prompt stack is the ctx spliced over
        positions 1..n_ctx (slinet.py:130, prompt_learner.py:155-163); use_ctx=False = extract_vector (:118-126)."""
        out, ctx = run_alone(self.encode_text_gen(ids, prompts, depth, train, use_ctx, normalise))
        return (out, ctx) if return_ctx else out

    def encode_text_gen(self, ids, prompts=None, depth=1, train=False, use_ctx=True, normalise=True):
        """GENERATOR form of encode_text; returns (features, backward context)."""
        cfg, dt, s = self.cfg, self.dt, _stream()
        self._guard_begin()
        packed = ids if isinstance(ids, PackedIds) else None
        if packed is not None:
            ids = packed.on(self.device)
        B, L = ids.shape
        ids = ids.to(device=self.device, dtype=torch.int64).contiguous()
        pr, pbs, P = self._prompt_args(prompts, B)
        if pr is not None and P != self.n_ctx:
            raise ValueError("prompt length must equal n_ctx")
        d = cfg.transformer_width
        if L > cfg.context_length:
            raise ValueError("more tokens than the context length")
        if packed is not None and int(packed.lengths.min()) < self.n_ctx + 2:
            raise ValueError("a packed caption must hold SOT, the n_ctx context slots and EOT")
        pre = packed.shared if packed is not None else 0
        if pre:
            # the first 1 + n_ctx positions are the same rows for every sample only if the context is spliced in and BROADCAST (slinet.py:119-130: training);
            # per-sample prompts (slinet.py:215: inference) and extract_vector (use_ctx=False) need the plain packed layout
            if pr is None or pbs != 0 or not use_ctx or pre != 1 + self.n_ctx:
                raise ValueError("PackedIds(shared=1 + n_ctx) needs broadcast prompts spliced into the caption (the training forward)")
        ws = self.txt.workspace(B, L, train, cap=cfg.context_length, packed=packed)
        hw = self._head("t", B, d)
        if packed is not None:       # the EOT positions came with the packed layout (host side): no argmax kernel
            eot_idx = packed.eot_dev
        else:
            eot_idx = hw["idx"]
            call("lpi_eot_index", B, L, ids, eot_idx, s)
        ctx = pr if (pr is not None and use_ctx) else None
        if pre:
            call("lpi_txt_embed_fwd_shared", self.txt.xdt, B, L, ws["rs"], pre, self.n_ctx, d, ids, self.tok, self.tpos, ctx, ws["x"][0],
                 *self.txt.ln1_stats_out(ws), s)
        else:
            call("lpi_txt_embed_fwd_varlen", self.txt.xdt, B, L, ws["rs"], self.n_ctx, d, ids, self.tok, self.tpos, ctx, pbs, ws["x"][0],
                 *self.txt.ln1_stats_out(ws), s)
        xo = yield from self.txt.forward_gen(ws, pr, pbs, depth, train, eot_idx, ln1_ready=self.txt.ln1_stats_out(ws)[0] is not None)      # pooled (EOT) rows
        out = yield from self._head_fwd_gen(hw, xo, self.ln_final, self.tproj, B, d, normalise)
        self._guard_end()
        ctx = (ws, pr, pbs, P, depth, B, L, out, self.txt.serial, eot_idx)
        self._txt_ctx = ctx
        return out, ctx

    def encode_text_backward(self, dout, ctx=None):
        return run_alone(self.encode_text_backward_gen(dout, ctx))

    def encode_text_backward_gen(self, dout, ctx=None):
        cfg, dt, s = self.cfg, self.gdt, _stream()         # the backward's operand / storage type
        ws, pr, pbs, P, depth, B, L, out, serial, eot_idx = ctx if ctx is not None else self._txt_ctx
        self._stale(self.txt, serial, "encode_text_backward")
        d, E = cfg.transformer_width, cfg.embed_dim
        hw = self._head("t", B, d)
        yield from self._head_bwd_gen(hw, ws, out, dout, self.ln_final, self.tproj, B, d)
        if pr is None:
            return None
        Lyr = pr.shape[-3]
        acc = 1 if ws.pop("dprompts_seeded", False) else 0
        dpr = self._dprompts(ws, Lyr, P, d, depth, seeded=bool(acc))
        yield from self.txt.backward_gen(ws, pr, depth, dpr, eot_idx, acc)
        if ws.get("pre", 0):      # shared prefix: rows 1 .. P of the stream hold the batch sum already
            yield RowsSumReq("front.psum", dt, 1, ws["pre"], None, 1, P, d, ws["dx"] if dt == F32 else ws["dxT"], dpr[0], acc)
        else:
            yield RowsSumReq("front.psum", dt, B, L, ws["rs"], 1, P, d, ws["dx"] if dt == F32 else ws["dxT"], dpr[0], acc)
        return dpr


    # ------------------------------------------------------------------ both towers in lock step
    def encode_both(self, image, ids, vis_prompts=None, txt_prompts=None, depth=1, train=False):
        """encode_image and encode_text with the two towers' GEMMs of the same layer op issued as ONE grouped launch
        (lpi_gemm_nt_grouped: the text tower's tiles fill the vision tower's partial last round of CUs).  The towers are independent
        (slinet.py:121-133 runs them one after the other), every kernel is the one encode_image / encode_text launch, so the results are
        the same bits.  Returns ((img_f, vis_ctx), (txt_f, txt_ctx))."""
        return run_lockstep(self.encode_image_gen(image, vis_prompts, depth, train), self.encode_text_gen(ids, txt_prompts, depth, train))

    def encode_both_backward(self, dimg, dtxt, vis_ctx, txt_ctx):
        """The two backward passes in lock step -> (dL/d vis prompts, dL/d txt prompts)."""
        return run_lockstep(self.encode_image_backward_gen(dimg, vis_ctx), self.encode_text_backward_gen(dtxt, txt_ctx))


def make_pixel_lut(mean=None, std=None) -> torch.Tensor:
    """[3, 256] f32 on the host: lut[c][v] = ToTensor + Normalize of byte v in channel c, in the loader's own torch operations (see DualEncoder.pixel_lut)."""
    m = torch.tensor(mean if mean is not None else (0.485, 0.456, 0.406)).view(3, 1)
    sd = torch.tensor(std if std is not None else (0.229, 0.224, 0.225)).view(3, 1)
    a = torch.arange(256, dtype=torch.uint8).view(1, 256).expand(3, 256).float().div_(255.0)
    return ((a - m) / sd).contiguous()


class PackedIds:
    """Token ids of a text batch together with the PACKED row layout of the text tower: sample b owns rows row_start[b] ..
    row_start[b+1]-1, its tokens 0 .. eot_b — the rows behind a sample's own EOT are not computed at all.

    Exact dead-row elimination, per sample (trim_token_ids cuts the whole batch at the LONGEST caption; this cuts every caption at
    its own end): under the causal mask (model.py:347-353) a position attends only to earlier ones and the tower's output is read at
    the EOT position alone (prompt_learner.py:61), so those rows can reach neither a feature nor a gradient.  Built on the HOST,
    where the tokenizer's output lives (prompt_learner.py:128-133): the row count and the longest caption are known without a
    device synchronisation.  Accepted by DualEncoder.encode_text / encode_both and the autograd Functions in place of the id tensor."""

    def __init__(self, ids, shared=0):
        """shared = n > 0: the SHARED-PREFIX layout of the training forward (include/lpi_hip.h, lpi_attn_fwd_shared): positions 0 .. n-1 (SOT and the
        n_ctx context slots, which the broadcast prompts overwrite — slinet.py:119-130) are stored once, as rows [0, n); sample b owns only the rows of
        its positions n, n + 1, ... .  Exact under the causal mask; the text tower then computes 40 % fewer rows on COCO-length captions."""
        a = np.ascontiguousarray(ids.numpy() if torch.is_tensor(ids) else np.asarray(ids))
        if torch.is_tensor(ids) and ids.is_cuda:
            raise ValueError("PackedIds is built from host token ids (before the upload)")
        self.lengths = a.argmax(-1).astype(np.int64) + 1            # eot_b + 1 (ids.argmax(-1) is the EOT position, prompt_learner.py:61)
        self.shape = (a.shape[0], int(self.lengths.max()))
        self.ids = torch.from_numpy(np.ascontiguousarray(a[:, :self.shape[1]]).astype(np.int64))
        self.shared = int(shared)
        rs = np.zeros(a.shape[0] + 1, dtype=np.int64)
        if self.shared:
            if int(self.lengths.min()) <= self.shared:
                raise ValueError("every caption must continue behind the shared positions (at least its EOT)")
            if not (a[:, 0] == a[0, 0]).all():
                raise ValueError("the captions do not start with the same token (SOT)")
            rs[0] = self.shared
            np.cumsum(self.lengths - self.shared, out=rs[1:])
            rs[1:] += self.shared
        else:
            np.cumsum(self.lengths, out=rs[1:])
        self.rows = int(rs[-1])
        self.row_start = torch.from_numpy(rs.astype(np.int32))
        self.pool_rows = torch.from_numpy((rs[1:] - 1).astype(np.int32))       # absolute row of every sample's EOT token
        self.eot = torch.from_numpy((self.lengths - 1).astype(np.int32))       # ... and its index within the sample (= ids.argmax(-1))
        self._dev = None
        self.row_start_dev = self.pool_rows_dev = self.eot_dev = None

    def on(self, device, non_blocking=False):
        """Upload once (ids, row starts, pooled rows); returns the device id matrix [B, Lmax].  non_blocking: the four arrays go through PINNED copies
        (kept on this object until it dies) and the calls only enqueue on the current stream — the input pipeline's producer thread must never wait for
        the stream it feeds (lpi_amd/pipeline.py)."""
        device = torch.device(device)
        if self._dev is None or self._dev.device != device:
            if non_blocking:
                # four small pinned copies (each is a blit kernel of a few microseconds on the copy's stream).  Packing the four arrays into ONE pinned buffer
                # and one copy was tried in round 5 and made the training loop 1.7x SLOWER (38.5 against 22.1 ms per iteration, tools/plugin_loop.py): the
                # producer then spent 17 ms in the upload and 18 ms in the image gather — the byte-view upload did not stay asynchronous
                self._pinned = [t.pin_memory() for t in (self.ids, self.row_start, self.pool_rows, self.eot)]
                self._dev, self.row_start_dev, self.pool_rows_dev, self.eot_dev = (t.to(device, non_blocking=True) for t in self._pinned)
            else:
                self._dev = self.ids.to(device)
                self.row_start_dev = self.row_start.to(device)
                self.pool_rows_dev = self.pool_rows.to(device)
                self.eot_dev = self.eot.to(device)
        return self._dev

    def to(self, device, non_blocking=False):
        self.on(device, non_blocking)
        return self

    def record_stream(self, stream):
        """The device arrays were allocated on the stream that uploaded them (lpi_amd.pipeline: a side stream); tell the allocator who else reads them."""
        for t in (self._dev, self.row_start_dev, self.pool_rows_dev, self.eot_dev):
            if t is not None:
                t.record_stream(stream)

    def slice(self, lo, hi):
        """The sub-batch of samples lo .. hi-1 (data-parallel shards, micro-batches)."""
        return PackedIds(self.ids[lo:hi], self.shared)

    def __getitem__(self, sl):
        if not isinstance(sl, slice) or sl.step not in (None, 1):
            raise TypeError("PackedIds takes contiguous sample slices only")
        lo, hi, _ = sl.indices(self.shape[0])
        return self.slice(lo, hi)


def trim_token_ids(ids):
    """Drop the token columns after the longest caption's EOT: ids [B, n] (HOST tensor or numpy array) -> ids[:, :max_b(eot_b) + 1].

    Exact dead-row elimination for the text tower: under the causal mask (model.py:347-353) a position attends only to earlier ones,
    and the tower's output is read at the EOT position alone (prompt_learner.py:61), so rows behind every sample's EOT can reach
    neither a feature nor a gradient; the engine takes any L <= context_length.  Done on the host, where the tokenizer's output
    lives (prompt_learner.py:128-133), so it costs no device synchronisation."""
    if torch.is_tensor(ids):
        if ids.is_cuda:
            raise ValueError("trim_token_ids works on host token ids (before the upload)")
        n = int(ids.argmax(-1).max()) + 1
    else:
        n = int(np.asarray(ids).argmax(-1).max()) + 1
    return ids[:, :n]


# ---------------------------------------------------------------------------------------------- loss / prompt ops
_LOSS_WS = {}


def _gemm_ready(t, n):
    """True if the f32 matrix t [n, E] can be a GEMM operand as it is (whole 128-row tiles, unit inner stride, 16-byte aligned rows)."""
    return (t.dtype == torch.float32 and n % 128 == 0 and t.stride(1) == 1 and (t.stride(0) * 4) % 16 == 0 and t.data_ptr() % 16 == 0)


def clip_loss_fwd_bwd(img_all, txt_all, scale: float, need_grad=True, r0: int = 0, nloc: Optional[int] = None):
    """Symmetric CE over the (global) n x n logits (loss/loss.py:75-87, slinet.py:139-141) and its gradient w.r.t. the LOCAL rows
    r0 .. r0+nloc of both feature matrices (data parallel: `local_loss=False` semantics of sprompt.py:75-80 — every rank evaluates
    the full loss but back-propagates only through its own rows; one rank: r0 = 0, nloc = n).
    img_all/txt_all: f32 [n, E] on device (row-strided views are fine).  Returns (loss[1], logits[n,n] view, dimg[nloc,E], dtxt[nloc,E]).
    Temporaries live in a per-shape workspace; the returned gradients are fresh tensors (they are saved on the autograd node)."""
    n, E = img_all.shape
    nloc = n if nloc is None else nloc
    if not (0 <= r0 and r0 + nloc <= n):
        raise ValueError("local rows outside the global batch")
    dev = img_all.device
    npad, lpad = _pad(n), _pad(nloc)
    s = _stream()
    key = (n, nloc, E, dev, torch.cuda.current_stream().cuda_stream)
    ws = _LOSS_WS.get(key)
    if ws is None:
        z = lambda *sh: torch.zeros(*sh, device=dev)  # noqa: E731
        ws = _LOSS_WS[key] = {"A": z(npad, E), "B": z(npad, E), "logits": z(npad, npad), "lse": z(2, npad),
                              "At": z(E, npad), "Bt": z(E, npad), "g": z(lpad, npad), "gt": z(lpad, npad)}
    logits, lse = ws["logits"], ws["lse"]
    ops = []
    for src, name in ((img_all, "A"), (txt_all, "B")):
        if _gemm_ready(src, n):
            ops.append(src)
        else:       # small / odd shapes: pack into the zero-padded operand buffer (rows >= n stay zero)
            srcf = src if (src.dtype == torch.float32 and src.stride(1) == 1) else src.float().contiguous()
            call("lpi_copy_rows", n, E, srcf, srcf.stride(0), ws[name], E, s)
            ops.append(ws[name])
    A, Bm = ops
    gemm(F32, A, Bm, logits, npad, npad, E, alpha=scale)
    loss = torch.empty(1, device=dev)
    if not need_grad:
        call("lpi_clip_loss_local", n, logits, npad, 1.0, 0, 0, loss, lse[0], lse[1], None, None, 0, s)
        return loss, logits[:n, :n], None, None
    # dI_loc = scale * g . T,  dT_loc = scale * gt . I  with g / gt the local rows of dlogits / dlogits^T (NT form: B operand = T^T / I^T).
    # Two launches for the two log-sum-exp vectors, the loss value and the local rows of both gradients (lpi_clip_loss_local), one for both transposes,
    # and the two small gradient GEMMs as one few-row launch (round 4: nine launches became five; round 6: four)
    At, Bt, g, gt = ws["At"], ws["Bt"], ws["g"], ws["gt"]
    call("lpi_clip_loss_local", n, logits, npad, 1.0, r0, nloc, loss, lse[0], lse[1], g, gt, npad, s)
    call("lpi_transpose2", F32, npad if A is ws["A"] else n, E, A, A.stride(0), At, npad, npad if Bm is ws["B"] else n, E, Bm, Bm.stride(0), Bt, npad, s)
    dI, dT = torch.empty(lpad, E, device=dev), torch.empty(lpad, E, device=dev)
    if GEMM_PROFILE is None and _few_rows(F32, lpad, E, npad):
        _lib.gemm_rows(F32, F32, EPI_NONE, scale, [dict(M=lpad, N=E, K=npad, a=g, b=Bt, c=dI), dict(M=lpad, N=E, K=npad, a=gt, b=At, c=dT)], s)
    else:
        gemm(F32, g, Bt, dI, lpad, E, npad, alpha=scale)
        gemm(F32, gt, At, dT, lpad, E, npad, alpha=scale)
    return loss, logits[:n, :n], dI[:nloc], dT[:nloc]


def _padded_rows(t, n, npad, name, ws, s):
    """t [n, E] f32 (any row stride) as a GEMM operand with npad rows: itself if it already is one, else packed into the zero-padded ws[name]."""
    if npad == n and _gemm_ready(t, n):
        return t
    E = t.shape[1]
    buf = ws.get(name)
    if buf is None or buf.shape != (npad, E):
        buf = ws[name] = torch.zeros(npad, E, device=t.device)
    tf = t if (t.dtype == torch.float32 and t.stride(1) == 1) else t.float().contiguous()
    call("lpi_copy_rows", n, E, tf, tf.stride(0), buf, E, s)
    return buf


def clip_loss_local_fwd_bwd(img_all, txt_all, scale: float, r0: int, nloc: int, need_grad=True, key_grads=False):
    """`local_loss=True` form of the contrastive loss (sprompt.py:278-283 + loss/loss.py:62-87): this rank's nloc images against ALL n
    texts and its nloc texts against all n images, labels r0 + i; loss = (CE(logits_per_image) + CE(logits_per_text)) / 2 over the
    nloc local rows.  Returns (loss[1], dI_q [nloc,E], dT_q [nloc,E], dI_k, dT_k): the gradients w.r.t. the LOCAL features as queries
    (rows of the two logit blocks) and, with key_grads (`gather_with_grad=True`: the gathered features carry gradient,
    sprompt.py:67-69), w.r.t. ALL n features as keys ([n,E] each; the caller reduce-scatters them to their owners) — else None."""
    n, E = img_all.shape
    if not (0 <= r0 and r0 + nloc <= n):
        raise ValueError("local rows outside the global batch")
    dev = img_all.device
    s = _stream()
    npad, lpad = _pad(n), _pad(nloc)
    key = ("local", n, nloc, E, dev, torch.cuda.current_stream().cuda_stream)
    ws = _LOSS_WS.get(key)
    if ws is None:
        z = lambda *sh: torch.zeros(*sh, device=dev)  # noqa: E731
        ws = _LOSS_WS[key] = {"li": z(lpad, npad), "lt": z(lpad, npad), "rows": z(2, lpad), "At": z(E, npad), "Bt": z(E, npad),
                              "git": z(npad, lpad), "gtt": z(npad, lpad), "ilt": z(E, lpad), "tlt": z(E, lpad)}
    Ia = _padded_rows(img_all, n, npad, "Ia", ws, s)
    Ta = _padded_rows(txt_all, n, npad, "Ta", ws, s)
    Il = _padded_rows(img_all[r0:r0 + nloc], nloc, lpad, "Il", ws, s)
    Tl = _padded_rows(txt_all[r0:r0 + nloc], nloc, lpad, "Tl", ws, s)
    li, lt, rows = ws["li"], ws["lt"], ws["rows"]
    gemm(F32, Il, Ta, li, lpad, npad, E, alpha=scale)            # logits_per_image = scale * I_loc . T_all^T
    gemm(F32, Tl, Ia, lt, lpad, npad, E, alpha=scale)            # logits_per_text  = scale * T_loc . I_all^T
    up = 0.5 / nloc
    g = li if need_grad else None                                # the gradients overwrite the logit blocks
    gt = lt if need_grad else None
    call("lpi_ce_rows_fwd_bwd", nloc, n, li, npad, r0, up, rows[0], g, npad, s)
    call("lpi_ce_rows_fwd_bwd", nloc, n, lt, npad, r0, up, rows[1], gt, npad, s)
    loss = torch.empty(1, device=dev)
    call("lpi_sum_scaled", nloc, rows[0], rows[1], up, loss, s)
    if not need_grad:
        return loss, None, None, None, None
    # rows >= nloc and columns >= n of the two blocks are zero (products with the zero padding rows of the operands) and the CE kernel
    # writes rows < nloc, columns < n only: the padded GEMMs below see zeros there
    At, Bt = ws["At"], ws["Bt"]
    call("lpi_transpose", F32, npad, E, Ia, Ia.stride(0), At, npad, s)
    call("lpi_transpose", F32, npad, E, Ta, Ta.stride(0), Bt, npad, s)
    dIq, dTq = torch.empty(lpad, E, device=dev), torch.empty(lpad, E, device=dev)
    gemm(F32, li, Bt, dIq, lpad, E, npad, alpha=scale)           # d I_loc (queries) = scale * g  . T_all
    gemm(F32, lt, At, dTq, lpad, E, npad, alpha=scale)           # d T_loc (queries) = scale * gt . I_all
    if not key_grads:
        return loss, dIq[:nloc], dTq[:nloc], None, None
    git, gtt, ilt, tlt = ws["git"], ws["gtt"], ws["ilt"], ws["tlt"]
    call("lpi_transpose", F32, lpad, npad, li, npad, git, lpad, s)
    call("lpi_transpose", F32, lpad, npad, lt, npad, gtt, lpad, s)
    call("lpi_transpose", F32, lpad, E, Il, Il.stride(0), ilt, lpad, s)
    call("lpi_transpose", F32, lpad, E, Tl, Tl.stride(0), tlt, lpad, s)
    dTk, dIk = torch.empty(npad, E, device=dev), torch.empty(npad, E, device=dev)
    gemm(F32, git, ilt, dTk, npad, E, lpad, alpha=scale)         # d T_all (keys of logits_per_image) = scale * g^T  . I_loc
    gemm(F32, gtt, tlt, dIk, npad, E, lpad, alpha=scale)         # d I_all (keys of logits_per_text)  = scale * gt^T . T_loc
    return loss, dIq[:nloc], dTq[:nloc], dIk[:n], dTk[:n]


def clip_loss_full_grad(img_all, txt_all, scale: float):
    """The global loss with gradients w.r.t. ALL n features (`local_loss=False, gather_with_grad=True`, sprompt.py:67-69: every rank
    back-propagates through every gathered feature; the caller reduce-scatters).  Returns (loss[1], dI_all [n,E], dT_all [n,E])."""
    n = img_all.shape[0]
    loss, _, dI, dT = clip_loss_fwd_bwd(img_all, txt_all, scale, True, 0, n)
    return loss, dI, dT


def score_matrix(img_feats, txt_feats):
    """The N_img x N_txt cosine score matrix of the evaluation (sprompt.py:509 `(image_feats @ text_feats.t()).t()`) and its transpose,
    through the f32 NT GEMM (exact f32 products, f32 accumulation) and lpi_transpose: -> (score_i2t [Ni, Nt], score_t2i [Nt, Ni])."""
    ni, E = img_feats.shape
    nt = txt_feats.shape[0]
    dev = img_feats.device
    s = _stream()
    nip, ntp = _pad(ni), _pad(nt)
    ops = []
    for src, n, npad in ((img_feats, ni, nip), (txt_feats, nt, ntp)):
        if _gemm_ready(src, n):
            ops.append(src)
        else:
            buf = torch.zeros(npad, E, device=dev)
            srcf = src if (src.dtype == torch.float32 and src.stride(1) == 1) else src.float().contiguous()
            call("lpi_copy_rows", n, E, srcf, srcf.stride(0), buf, E, s)
            ops.append(buf)
    full = torch.empty(nip, ntp, device=dev)
    gemm(F32, ops[0], ops[1], full, nip, ntp, E)
    i2t = torch.empty(ni, nt, device=dev)
    call("lpi_copy_rows", ni, nt, full, ntp, i2t, nt, s)
    t2i = torch.empty(nt, ni, device=dev)
    call("lpi_transpose", F32, ni, nt, i2t, nt, t2i, ni, s)
    return i2t, t2i


def prompt_cp_fwd2(d1, d2v, d2t, d3v, d3t, scale=1.0):
    """Both prompt stacks of a DecomposedPrompt (prompts.py:38-57) in one launch -> (vis [Lyr,P,Dv], txt [Lyr,P,Dt])."""
    Lyr, r = d1.shape
    P, Dv, Dt = d2v.shape[0], d3v.shape[0], d3t.shape[0]
    outv, outt = torch.empty(Lyr, P, Dv, device=d1.device), torch.empty(Lyr, P, Dt, device=d1.device)
    call("lpi_prompt_cp_fwd2", Lyr, P, Dv, Dt, r, d1, d2v, d2t, d3v, d3t, float(scale), outv, outt, _stream())
    return outv, outt


_CP_SCRATCH = {}


def prompt_cp_bwd2(d1, d2v, d2t, d3v, d3t, doutv, doutt, scale=1.0, out=None):
    """The five factor gradients from the two stacks' gradients in two launches (lpi_prompt_cp_bwd2).  out: (g1, g2v, g2t, g3v, g3t) to write into
    (contiguous f32, e.g. the slices of a flat gradient buffer: optim.flatten); default: new tensors."""
    Lyr, r = d1.shape
    P, Dv, Dt = d2v.shape[0], d3v.shape[0], d3t.shape[0]
    g1, g2v, g2t, g3v, g3t = out if out is not None else (torch.empty_like(d1), torch.empty_like(d2v), torch.empty_like(d2t), torch.empty_like(d3v),
                                                            torch.empty_like(d3t))
    key = (d1.device, torch.cuda.current_stream().cuda_stream, 2 * Lyr * P * r)
    scratch = _CP_SCRATCH.get(key)
    if scratch is None:
        scratch = _CP_SCRATCH[key] = torch.empty(2 * Lyr * P * r, device=d1.device)
    call("lpi_prompt_cp_bwd2", Lyr, P, Dv, Dt, r, d1, d2v, d2t, d3v, d3t, float(scale), doutv.contiguous(), doutt.contiguous(), g1, g2v, g2t, g3v, g3t,
         scratch, _stream())
    return g1, g2v, g2t, g3v, g3t


def prompt_cp_fwd(d1, d2, d3, scale=1.0):
    Lyr, r = d1.shape
    P, D = d2.shape[0], d3.shape[0]
    out = torch.empty(Lyr, P, D, device=d1.device)
    call("lpi_prompt_cp_fwd", Lyr, P, D, r, d1, d2, d3, float(scale), out, _stream())
    return out


def prompt_cp_bwd(d1, d2, d3, dout, g1, accumulate_g1, scale=1.0, out=None):
    """out: (g2, g3) to write into (contiguous f32, e.g. slices of a flat gradient buffer: optim.flatten); default: new tensors."""
    Lyr, r = d1.shape
    P, D = d2.shape[0], d3.shape[0]
    g2, g3 = out if out is not None else (torch.empty_like(d2), torch.empty_like(d3))
    scratch = torch.empty(Lyr * P * r, device=d1.device)
    call("lpi_prompt_cp_bwd", Lyr, P, D, r, d1, d2, d3, float(scale), dout.contiguous(), g1, g2, g3, int(accumulate_g1), scratch, _stream())
    return g2, g3


_ALIGN_SCRATCH = {}


def align_loss_fwd_bwd(vis, txt, temp=0.01, weight=0.1, need_grad=True, out=None):
    """slinet.py:143-158 -> (loss [1], dvis, dtxt).  out: (dvis, dtxt) buffers to write the dense gradients into (contiguous f32 [Lyr, P, D]: the towers'
    prompt-gradient buffers, DualEncoder.seed_prompt_grads); default: new tensors.  Two launches (lpi_align_loss_fwd_bwd2)."""
    Lyr, P, Dv = vis.shape
    Dt = txt.shape[-1]
    loss = torch.empty(1, device=vis.device)          # written, not accumulated
    dv = dtx = None
    if need_grad:
        dv, dtx = out if out is not None else (torch.empty_like(vis), torch.empty_like(txt))
    key = (vis.device, torch.cuda.current_stream().cuda_stream, 2 * Lyr * P)
    scratch = _ALIGN_SCRATCH.get(key)
    if scratch is None:
        scratch = _ALIGN_SCRATCH[key] = torch.empty(2 * Lyr * P, device=vis.device)
    call("lpi_align_loss_fwd_bwd2", Lyr, P, Dv, Dt, vis.contiguous(), txt.contiguous(), float(temp), float(weight), loss, dv, dtx, scratch, _stream())
    return loss, dv, dtx
