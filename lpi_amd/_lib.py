"""ctypes binding of ``liblpi_hip.so`` (C ABI declared in ``include/lpi_hip.h``).

There is NO CPU fallback: if the shared library is missing, or a call returns an error, this raises.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_float, c_int, c_long, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# LPI_LIB: an alternate build of the library (tools/build_variant.sh: A/B and ablation builds for the measurement tools)
LIB_PATH = os.environ.get("LPI_LIB") or os.path.join(_HERE, "csrc", "liblpi_hip.so")

F32, BF16, F16 = 0, 1, 2
EPI_NONE, EPI_QUICKGELU, EPI_DQUICKGELU, EPI_LN, EPI_LN_QUICKGELU, EPI_RES_ROWSTATS = 0, 1, 2, 3, 4, 5


class LpiError(RuntimeError):
    pass


class GemmDesc(ctypes.Structure):
    """``lpi_gemm_desc`` of include/lpi_hip.h: one problem of lpi_gemm_nt_grouped (host struct, device pointers inside)."""
    _fields_ = [("M", c_int), ("N", c_int), ("K", c_int), ("A", c_void_p), ("lda", c_int), ("B", c_void_p), ("ldb", c_int),
                ("C", c_void_p), ("ldc", c_int), ("bias", c_void_p), ("residual", c_void_p), ("ldr", c_int), ("aux", c_void_p),
                ("ldaux", c_int)]


_P, _I, _F, _L = c_void_p, c_int, c_float, c_long

# name -> argtypes (restype is int unless listed in _RESTYPES); must list EVERY symbol of include/lpi_hip.h
SIGNATURES = {
    "lpi_version": [],
    "lpi_launch_count": [],
    "lpi_set_tuning": [_I, _I],
    "lpi_get_tuning": [_I],
    "lpi_gemm_last_kernel": [],
    "lpi_gemm_nt": [_I, _I, _I, _I, _I, _P, _I, _P, _I, _P, _I, _P, _P, _I, _I, _P, _I, _F, _P],
    "lpi_gemm_nt_grouped": [_I, _I, _I, _F, _I, _P, _P],
    "lpi_gemm_last_grouped": [],
    "lpi_gemm_ln_supported": [_I, _I, _I, _I],
    "lpi_ln_stats_finalize": [_I, _I, _P, _I, _F, _P, _P, _P],
    "lpi_rowstat_guard": [_P, _P],
    "lpi_ln_stats_finalize_pair": [_I, _I, _P, _I, _P, _P, _I, _I, _P, _I, _P, _P, _F, _P],
    "lpi_gemm_nt_rows": [_I, _I, _I, _F, _I, _P, _P],
    "lpi_gemm_nt_rows_supported": [_I, _I, _I, _I],
    "lpi_layernorm_fwd": [_I, _I, _I, _I, _P, _I, _P, _P, _P, _I, _P, _P, _P],
    "lpi_layernorm_bwd": [_I, _I, _I, _I, _I, _P, _I, _P, _I, _P, _P, _P, _P, _I, _P, _I, _I, _P],
    "lpi_layernorm_fwd_pair": [_I, _I, _P, _P],
    "lpi_layernorm_bwd_pair": [_I, _I, _I, _P, _P],
    "lpi_layernorm_bwd_rows": [_I, _I, _I, _I, _I, _I, _I, _I, _P, _I, _P, _I, _P, _P, _P, _P, _I, _P, _I, _I, _P],
    "lpi_gather_batch_rows": [_I, _I, _I, _I, _I, _I, _P, _I, _P, _I, _P],
    "lpi_attn_fwd": [_I, _I, _I, _I, _P, _I, _P, _I, _P, _I, _P],
    "lpi_attn_bwd": [_I, _I, _I, _I, _P, _I, _P, _I, _P, _I, _P, _P, _P, _I, _I, _P],
    "lpi_attn_pooled_fwd": [_I, _I, _I, _I, _P, _I, _P, _I, _P, _P, _I, _P, _I, _P],
    "lpi_attn_pooled_bwd": [_I, _I, _I, _I, _P, _I, _P, _I, _P, _P, _I, _P, _P, _I, _P, _I, _I, _P],
    # ragged (packed) batches: the same kernels with a row_start array (NULL = uniform length L)
    "lpi_attn_fwd_varlen": [_I, _I, _I, _P, _I, _P, _I, _P, _I, _P, _I, _P],
    "lpi_attn_bwd_varlen": [_I, _I, _I, _P, _I, _P, _I, _P, _I, _P, _I, _P, _P, _P, _I, _I, _P],
    "lpi_attn_fwd_pair": [_I, _P, _P],
    "lpi_attn_bwd_layout": [_I, _I, _I, _I, _I, _P, _I, _P, _I, _P, _I, _P, _P, _P, _I, _P, _P],
    "lpi_attn_fwd_one": [_I, _P, _P],
    "lpi_spool_attn_supported": [_I, _I, _I],
    "lpi_spool_attn_fwd": [_I, _I, _I, _I, _P, _I, _P, _I, _P, _I, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _I, _P],
    "lpi_spool_attn_bwd": [_I, _I, _I, _P, _I, _P, _I, _P, _I, _P, _P, _P, _P, _P, _P, _I, _P, _I, _P, _I, _P],
    "lpi_attn_bwd_prefix": [_I, _I, _I, _P, _I, _I, _P, _I, _P, _I, _P, _I, _P, _P, _P, _I, _I, _P],
    "lpi_attn_pooled_fwd_varlen": [_I, _I, _I, _P, _I, _P, _I, _P, _I, _P, _P, _I, _P, _I, _P],
    "lpi_attn_pooled_bwd_varlen": [_I, _I, _I, _P, _I, _P, _I, _P, _I, _P, _P, _I, _P, _P, _I, _P, _I, _I, _P],
    "lpi_layernorm_bwd_rows_varlen": [_I, _I, _I, _I, _I, _P, _I, _I, _I, _P, _I, _P, _I, _P, _P, _P, _P, _I, _P, _I, _I, _P],
    "lpi_gather_batch_rows_varlen": [_I, _I, _I, _P, _I, _I, _I, _P, _I, _P, _I, _P],
    "lpi_rows_sum_over_batch_varlen": [_I, _I, _I, _P, _I, _I, _I, _P, _P, _I, _P],
    "lpi_prompt_add_varlen": [_I, _I, _I, _P, _I, _I, _P, _P, _L, _P, _P, _P],
    "lpi_txt_embed_fwd_varlen": [_I, _I, _I, _P, _I, _I, _P, _P, _P, _P, _L, _P, _P, _P, _P],
    "lpi_scatter_add_rows": [_I, _I, _I, _I, _P, _I, _P, _P, _I, _P],
    "lpi_prompt_cp_fwd": [_I, _I, _I, _I, _P, _P, _P, _F, _P, _P],
    "lpi_prompt_cp_bwd": [_I, _I, _I, _I, _P, _P, _P, _F, _P, _P, _P, _P, _I, _P, _P],
    "lpi_align_loss_fwd_bwd": [_I, _I, _I, _I, _P, _P, _F, _F, _P, _P, _P, _P],
    "lpi_nt_bxent_fwd_bwd": [_I, _I, _I, _P, _P, _F, _F, _P, _P, _I, _P, _P],
    "lpi_patchify": [_I, _I, _I, _I, _P, _P, _I, _P],
    "lpi_patchify_u8": [_I, _I, _I, _I, _P, _P, _P, _I, _P],
    "lpi_vis_assemble_fwd": [_I, _I, _I, _I, _I, _P, _I, _P, _P, _P, _L, _P, _P, _P, _P, _P, _P, _P, _P],
    "lpi_vis_assemble_bwd": [_I, _I, _I, _I, _I, _P, _P, _L, _P, _P, _P, _P, _P],
    "lpi_txt_embed_fwd": [_I, _I, _I, _I, _I, _P, _P, _P, _P, _L, _P, _P, _P, _P],
    "lpi_rows_sum_over_batch": [_I, _I, _I, _I, _I, _I, _P, _P, _I, _P],
    "lpi_prompt_add": [_I, _I, _I, _I, _I, _P, _P, _L, _P, _P, _P],
    "lpi_pool_ln_fwd": [_I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _I, _P, _P, _P],
    "lpi_pool_ln_bwd": [_I, _I, _I, _I, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P],
    "lpi_gather_rows": [_I, _I, _I, _I, _P, _P, _P, _P],
    "lpi_scatter_rows": [_I, _I, _I, _I, _P, _P, _P, _P, _P],
    "lpi_l2norm_fwd": [_I, _I, _P, _I, _P, _I, _P, _P],
    "lpi_l2norm_bwd": [_I, _I, _P, _I, _P, _I, _P, _P, _I, _P],
    "lpi_eot_index": [_I, _I, _P, _P, _P],
    "lpi_clip_loss_fwd_bwd": [_I, _P, _I, _F, _P, _P, _I, _P, _P, _P],
    "lpi_clip_loss_local_grad": [_I, _P, _I, _P, _P, _F, _I, _I, _P, _P, _I, _P],
    "lpi_clip_loss_local": [_I, _P, _I, _F, _I, _I, _P, _P, _P, _P, _P, _I, _P],
    "lpi_prompt_cp_fwd2": [_I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _F, _P, _P, _P],
    "lpi_prompt_cp_bwd2": [_I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _F, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "lpi_align_loss_fwd_bwd2": [_I, _I, _I, _I, _P, _P, _F, _F, _P, _P, _P, _P, _P],
    "lpi_transpose2": [_I, _I, _I, _P, _I, _P, _I, _I, _I, _P, _I, _P, _I, _P],
    "lpi_row_jobs": [_I, _P, _P],
    "lpi_rows_sum_over_batch_pair": [_I, _P, _P],
    "lpi_attn_pooled_fwd_pair": [_I, _P, _P],
    "lpi_attn_pooled_bwd_pair": [_I, _P, _P],
    "lpi_attn_pooled_fwd_desc": [_I, _P, _P],
    "lpi_attn_pooled_bwd_desc": [_I, _P, _P],
    "lpi_txt_embed_fwd_shared": [_I, _I, _I, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P],
    "lpi_attn_fwd_shared": [_I, _I, _I, _P, _I, _I, _P, _I, _P, _I, _P, _P],
    "lpi_attn_bwd_shared": [_I, _I, _I, _P, _I, _I, _I, _P, _I, _P, _I, _P, _I, _P, _P, _P, _I, _P, _P],
    "lpi_shared_kv_reduce": [_I, _I, _I, _I, _P, _P, _I, _I, _P],
    "lpi_ce_rows_fwd_bwd": [_I, _I, _P, _I, _I, _F, _P, _P, _I, _P],
    "lpi_sum_scaled": [_I, _P, _P, _F, _P, _P],
    "lpi_zero": [_P, _L, _P],
    "lpi_copy_rows": [_I, _I, _P, _L, _P, _L, _P],
    "lpi_l1_task_id": [_I, _I, _I, _I, _P, _I, _P, _P, _P, _P],
    "lpi_kmeans_sqdist": [_I, _I, _I, _P, _I, _P, _P, _P],
    "lpi_kmeans_assign": [_I, _I, _I, _P, _I, _P, _P, _P, _P, _P],
    "lpi_kmeans_update": [_I, _I, _I, _P, _I, _P, _P, _P, _P],
    "lpi_kmeans_colstats": [_I, _I, _P, _I, _P, _P, _P, _P],
    "lpi_sgd_step": [_L, _P, _P, _P, _F, _F, _F, _I, _P],
    "lpi_cast": [_I, _I, _L, _P, _P, _P],
    "lpi_transpose": [_I, _I, _I, _P, _I, _P, _I, _P],
    "lpi_retrieval_rank": [_I, _I, _P, _I, _P, _I, _P, _P],
    "lpi_topk": [_I, _I, _I, _P, _I, _P, _P, _P],
    # host side: batch assembly of the input pipeline (pipeline.py), BPE tokenizer (a6)
    "lpi_host_gather": [_P, _P, _I, _L, _I],
    "lpi_bpe_create": [_P, _L],
    "lpi_bpe_destroy": [_P],
    "lpi_interact_workspace_floats": [_I, _I, _I, _I],
    "lpi_interact_fwd": [_I, _I, _I, _I, _I, _I, _P, _I, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _F, _F, _P, _I, _P, _I, _P, _P],
    "lpi_interact_bwd": [_I, _I, _I, _I, _I, _I, _P, _I, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _F, _F, _P, _I, _P, _I, _P, _I, _P, _I,
                         _P, _P, _P],
    "lpi_bpe_encode": [_P, _P, _P, _I],
    "lpi_bpe_tokenize": [_P, _P, _I, _I, _I, _P],
}
_RESTYPES = {"lpi_launch_count": c_uint64, "lpi_bpe_create": c_void_p, "lpi_bpe_destroy": None}

# The C ABI this binding was written against (lpi_version()).  Bumped with every change of a signature or of an argument's meaning: a stale
# liblpi_hip.so (or an LPI_LIB variant of another commit) would otherwise take shifted arguments silently.
EXPECTED_ABI = 604
VARIANT_OFFSET = 1000000      # lpi_version() of a tools/build_variant.sh build = EXPECTED_ABI + this

_lib = None


def load() -> ctypes.CDLL:
    """Load the library once; raise (never fall back) if it is absent or lacks a declared symbol."""
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own libamdhip64 (same SONAME as /opt/rocm's): import it first so that this library binds to the
    # HIP runtime instance that owns torch's device context and streams, whichever module the process imported first.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise LpiError(
            f"{LIB_PATH} not found: the MI355X HIP extension is not built. Run "
            f"`python -c 'import __graft_entry__ as g; g.build()'` or lpi_amd/csrc/build.sh. "
            f"There is no CPU fallback for the product path.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise LpiError(f"{LIB_PATH} does not export {name}") from e
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, c_int)
    got = int(lib.lpi_version())
    if got == EXPECTED_ABI + VARIANT_OFFSET and os.path.abspath(LIB_PATH) != os.path.join(_HERE, "csrc", "liblpi_hip.so"):
        # a diagnostic / ablation build (tools/build_variant.sh) of THIS ABI, asked for by name (LPI_LIB / a tool's LIB_PATH): never picked up silently
        import sys
        print(f"lpi_amd: loaded the VARIANT library {LIB_PATH} (diagnostic build; not the product library)", file=sys.stderr)
        got = EXPECTED_ABI
    if got != EXPECTED_ABI:
        raise LpiError(f"{LIB_PATH} has C-ABI version {got}, this binding expects {EXPECTED_ABI}: rebuild it (lpi_amd/csrc/build.sh)")
    _lib = lib
    # LPI_TUNING="key=value,key=value": speed-only knobs of lpi_set_tuning applied at load (A/B runs of bench.py; never results)
    for kv in filter(None, os.environ.get("LPI_TUNING", "").split(",")):
        k, v = kv.split("=")
        if lib.lpi_set_tuning(int(k), int(v)) != 0:
            raise LpiError(f"LPI_TUNING: bad knob {kv!r}")
    return lib


def _ptr(t):
    if t is None:
        return None
    return t.data_ptr() if hasattr(t, "data_ptr") else int(t)


def call(name: str, *args):
    """Call an int-returning entry point with tensors converted to device pointers; raise on error."""
    fn = getattr(load(), name)
    conv = [(_ptr(a) if (a is None or hasattr(a, "data_ptr")) else a) for a in args]
    rc = fn(*conv)
    if rc != 0:
        raise LpiError(f"{name} failed with code {rc}" + (" (invalid argument)" if rc == -22 else ""))


class LnFwdDesc(ctypes.Structure):
    """``lpi_ln_fwd_desc``"""
    _fields_ = [("rows", c_int), ("d", c_int), ("x", c_void_p), ("ldx", c_int), ("gamma", c_void_p), ("beta", c_void_p), ("y", c_void_p),
                ("ldy", c_int), ("mean", c_void_p), ("rstd", c_void_p)]


class LnBwdDesc(ctypes.Structure):
    """``lpi_ln_bwd_desc``"""
    _fields_ = [("rows", c_int), ("d", c_int), ("dy", c_void_p), ("lddy", c_int), ("x", c_void_p), ("ldx", c_int), ("gamma", c_void_p),
                ("mean", c_void_p), ("rstd", c_void_p), ("dx", c_void_p), ("lddx", c_int), ("dx_cast", c_void_p), ("ldcast", c_int),
                ("accumulate", c_int)]


class AttnFwdDesc(ctypes.Structure):
    """``lpi_attn_fwd_desc``"""
    _fields_ = [("B", c_int), ("L", c_int), ("H", c_int), ("row_start", c_void_p), ("qkv", c_void_p), ("ldqkv", c_int), ("ctx", c_void_p),
                ("ldctx", c_int), ("lse", c_void_p), ("causal", c_int), ("shared_rows", c_int), ("qkv_hs", c_int), ("qkv_vs", c_int), ("ctx_hs", c_int)]


def _fill_attn_fwd(q, t):
    q.B, q.L, q.row_start, q.H, q.qkv, q.ldqkv, q.ctx, q.ldctx, q.lse, q.causal = (t[0], t[1], _ptr(t[2]), t[3], _ptr(t[4]), t[5], _ptr(t[6]), t[7], _ptr(t[8]), t[9])
    q.shared_rows = t[10] if len(t) > 10 else 0
    q.qkv_hs, q.qkv_vs, q.ctx_hs = t[11] if len(t) > 11 and t[11] is not None else (0, 0, 0)      # layout strides (include/lpi_hip.h): 0 = interleaved


def attn_fwd_one(dt, a, stream):
    """One argument tuple of attn_fwd_pair through lpi_attn_fwd_one (the single-problem call that takes the layout strides)."""
    q = AttnFwdDesc()
    _fill_attn_fwd(q, a)
    rc = load().lpi_attn_fwd_one(dt, ctypes.cast(ctypes.pointer(q), c_void_p), stream)
    if rc != 0:
        raise LpiError(f"lpi_attn_fwd_one failed with code {rc}")


def attn_fwd_pair(dt, a, b, stream):
    """Two argument tuples (B, L, row_start, H, qkv, ldqkv, ctx, ldctx, lse, causal[, shared_rows]) of lpi_attn_fwd_varlen / _shared in one launch."""
    arr = (AttnFwdDesc * 2)()
    for q, t in zip(arr, (a, b)):
        _fill_attn_fwd(q, t)
    rc = load().lpi_attn_fwd_pair(dt, ctypes.cast(arr, c_void_p), stream)
    if rc != 0:
        raise LpiError(f"lpi_attn_fwd_pair failed with code {rc}")


def layernorm_fwd_pair(dt, xdt, a, b, stream):
    """Two lpi_layernorm_fwd argument tuples (rows, d, x, ldx, gamma, beta, y, ldy, mean, rstd) in one launch."""
    arr = (LnFwdDesc * 2)()
    for q, t in zip(arr, (a, b)):
        q.rows, q.d, q.x, q.ldx, q.gamma, q.beta, q.y, q.ldy, q.mean, q.rstd = (t[0], t[1], _ptr(t[2]), t[3], _ptr(t[4]), _ptr(t[5]), _ptr(t[6]), t[7],
                                                                               _ptr(t[8]), _ptr(t[9]))
    rc = load().lpi_layernorm_fwd_pair(dt, xdt, ctypes.cast(arr, c_void_p), stream)
    if rc != 0:
        raise LpiError(f"lpi_layernorm_fwd_pair failed with code {rc}")


def layernorm_bwd_pair(dydt, cdt, xdt, a, b, stream):
    """Two lpi_layernorm_bwd argument tuples (rows, d, dy, lddy, x, ldx, gamma, mean, rstd, dx, lddx, dx_cast, ldcast, accumulate)."""
    arr = (LnBwdDesc * 2)()
    for q, t in zip(arr, (a, b)):
        (q.rows, q.d, q.dy, q.lddy, q.x, q.ldx, q.gamma, q.mean, q.rstd, q.dx, q.lddx, q.dx_cast, q.ldcast, q.accumulate) = (
            t[0], t[1], _ptr(t[2]), t[3], _ptr(t[4]), t[5], _ptr(t[6]), _ptr(t[7]), _ptr(t[8]), _ptr(t[9]), t[10], _ptr(t[11]), t[12], t[13])
    rc = load().lpi_layernorm_bwd_pair(dydt, cdt, xdt, ctypes.cast(arr, c_void_p), stream)
    if rc != 0:
        raise LpiError(f"lpi_layernorm_bwd_pair failed with code {rc}")


def gemm_grouped(dt: int, cdt: int, epi: int, alpha: float, problems, stream) -> bool:
    """lpi_gemm_nt_grouped over `problems` = dicts with M, N, K, a, b, c and optional bias / residual / aux tensors.
    Returns True if they ran as one grouped launch."""
    lib = load()
    arr = (GemmDesc * len(problems))()
    for d, p in zip(arr, problems):
        d.M, d.N, d.K = p["M"], p["N"], p["K"]
        d.A, d.lda = p["a"].data_ptr(), p["a"].stride(0)
        d.B, d.ldb = p["b"].data_ptr(), p["b"].stride(0)
        d.C, d.ldc = p["c"].data_ptr(), p["c"].stride(0)
        bias, res, aux = p.get("bias"), p.get("residual"), p.get("aux")
        d.bias = None if bias is None else bias.data_ptr()
        d.residual, d.ldr = (None, 0) if res is None else (res.data_ptr(), p.get("ldr") or res.stride(0))
        d.aux, d.ldaux = (None, 0) if aux is None else (aux.data_ptr(), aux.stride(0))
    rc = lib.lpi_gemm_nt_grouped(dt, cdt, epi, float(alpha), len(problems), ctypes.cast(arr, c_void_p), stream)
    if rc != 0:
        raise LpiError(f"lpi_gemm_nt_grouped failed with code {rc}" + (" (invalid argument)" if rc == -22 else ""))
    return bool(lib.lpi_gemm_last_grouped())


def gemm_rows(dt: int, cdt: int, epi: int, alpha: float, problems, stream):
    """lpi_gemm_nt_rows over one or two `problems` (dicts as for gemm_grouped): the few-row GEMM in one launch."""
    arr = (GemmDesc * len(problems))()
    for d, p in zip(arr, problems):
        d.M, d.N, d.K = p["M"], p["N"], p["K"]
        d.A, d.lda = p["a"].data_ptr(), p["a"].stride(0)
        d.B, d.ldb = p["b"].data_ptr(), p["b"].stride(0)
        d.C, d.ldc = p["c"].data_ptr(), p["c"].stride(0)
        bias, res, aux = p.get("bias"), p.get("residual"), p.get("aux")
        d.bias = None if bias is None else bias.data_ptr()
        d.residual, d.ldr = (None, 0) if res is None else (res.data_ptr(), p.get("ldr") or res.stride(0))
        d.aux, d.ldaux = (None, 0) if aux is None else (aux.data_ptr(), aux.stride(0))
    rc = load().lpi_gemm_nt_rows(dt, cdt, epi, float(alpha), len(problems), ctypes.cast(arr, c_void_p), stream)
    if rc != 0:
        raise LpiError(f"lpi_gemm_nt_rows failed with code {rc}" + (" (invalid argument)" if rc == -22 else ""))


ROWOP_POOL_LN_FWD, ROWOP_L2NORM_FWD, ROWOP_L2NORM_BWD, ROWOP_POOL_LN_BWD, ROWOP_LN_BWD = 1, 2, 3, 4, 5
ROWOP_SCATTER_ADD, ROWOP_GATHER_BATCH_ROWS, ROWOP_PROMPT_ADD, ROWOP_LN_BWD_ROWS_H16, ROWOP_VIS_PROMPT_ROWS_BWD = 6, 7, 8, 9, 10
ROW_JOBS_MAX = 4


class RowJob(ctypes.Structure):
    """``lpi_row_job`` (include/lpi_hip.h): one small row kernel of an lpi_row_jobs launch."""
    _fields_ = [("op", c_int), ("B", c_int), ("L", c_int), ("d", c_int), ("P", c_int), ("row0", c_int), ("dt_a", c_int), ("dt_b", c_int),
                ("ld_a", c_int), ("ld_b", c_int), ("ld_c", c_int), ("flag", c_int), ("bstride", c_long),
                ("a", c_void_p), ("b", c_void_p), ("idx", c_void_p), ("row_start", c_void_p),
                ("gamma", c_void_p), ("beta", c_void_p), ("mean_in", c_void_p), ("rstd_in", c_void_p),
                ("out", c_void_p), ("out2", c_void_p), ("mean", c_void_p), ("rstd", c_void_p)]


_ROWJOB_PTRS = {"a", "b", "idx", "row_start", "gamma", "beta", "mean_in", "rstd_in", "out", "out2", "mean", "rstd"}


def row_job(op, **kw):
    """A RowJob from keyword fields; tensors become device pointers (None = NULL).  The job keeps a reference to every tensor it points at (`_keep`), so
    that a tensor the caller drops between building the job and issuing it is not recycled by the allocator under the launch."""
    j = RowJob()
    j.op = op
    j._keep = [v for k, v in kw.items() if k in _ROWJOB_PTRS and v is not None]
    for k, v in kw.items():
        setattr(j, k, (_ptr(v) if k in _ROWJOB_PTRS else v))
    return j


def row_jobs(jobs, stream):
    """lpi_row_jobs: up to ROW_JOBS_MAX independent small row kernels in one launch (more: several launches)."""
    for i in range(0, len(jobs), ROW_JOBS_MAX):
        part = jobs[i:i + ROW_JOBS_MAX]
        arr = (RowJob * len(part))(*part)
        rc = load().lpi_row_jobs(len(part), ctypes.cast(arr, c_void_p), stream)
        if rc != 0:
            raise LpiError(f"lpi_row_jobs failed with code {rc} (ops {[j.op for j in part]})" + (" (invalid argument)" if rc == -22 else ""))


class RowsSumDesc(ctypes.Structure):
    """``lpi_rows_sum_desc``"""
    _fields_ = [("B", c_int), ("L", c_int), ("row0", c_int), ("P", c_int), ("d", c_int), ("accumulate", c_int), ("row_start", c_void_p),
                ("dx", c_void_p), ("out", c_void_p)]


def rows_sum_pair(dt, a, b, stream):
    """Two argument tuples (B, L, row_start, row0, P, d, dx, out, accumulate) of lpi_rows_sum_over_batch_varlen in one launch."""
    arr = (RowsSumDesc * 2)()
    for q, t in zip(arr, (a, b)):
        q.B, q.L, q.row_start, q.row0, q.P, q.d, q.dx, q.out, q.accumulate = t[0], t[1], _ptr(t[2]), t[3], t[4], t[5], _ptr(t[6]), _ptr(t[7]), t[8]
    rc = load().lpi_rows_sum_over_batch_pair(dt, ctypes.cast(arr, c_void_p), stream)
    if rc != 0:
        raise LpiError(f"lpi_rows_sum_over_batch_pair failed with code {rc}")


class AttnPooledDesc(ctypes.Structure):
    """``lpi_attn_pooled_desc``"""
    _fields_ = [("B", c_int), ("L", c_int), ("H", c_int), ("row_start", c_void_p), ("q", c_void_p), ("ldq", c_int), ("qkv", c_void_p), ("ldqkv", c_int),
                ("idx", c_void_p), ("ctx", c_void_p), ("ldctx", c_int), ("lse", c_void_p), ("dctx", c_void_p), ("lddctx", c_int), ("dq", c_void_p),
                ("lddq", c_int), ("dqkv", c_void_p), ("lddqkv", c_int), ("causal", c_int), ("shared_rows", c_int), ("shared_dkv", c_void_p)]


_POOLED_PTRS = {"row_start", "q", "qkv", "idx", "ctx", "lse", "dctx", "dq", "dqkv", "shared_dkv"}


def attn_pooled_one(dt, a, stream, backward=False):
    """One dict of lpi_attn_pooled_desc fields -> lpi_attn_pooled_fwd_desc / _bwd_desc."""
    q = AttnPooledDesc()
    for k, v in a.items():
        setattr(q, k, (_ptr(v) if k in _POOLED_PTRS else v))
    fn = load().lpi_attn_pooled_bwd_desc if backward else load().lpi_attn_pooled_fwd_desc
    rc = fn(dt, ctypes.cast(ctypes.pointer(q), c_void_p), stream)
    if rc != 0:
        raise LpiError(f"lpi_attn_pooled_{'bwd' if backward else 'fwd'}_desc failed with code {rc}")


def attn_pooled_pair(dt, a, b, stream, backward=False):
    """Two dicts of lpi_attn_pooled_desc fields (tensors or ints) -> lpi_attn_pooled_fwd_pair / _bwd_pair."""
    arr = (AttnPooledDesc * 2)()
    ptrs = _POOLED_PTRS
    for q, t in zip(arr, (a, b)):
        for k, v in t.items():
            setattr(q, k, (_ptr(v) if k in ptrs else v))
    fn = load().lpi_attn_pooled_bwd_pair if backward else load().lpi_attn_pooled_fwd_pair
    rc = fn(dt, ctypes.cast(arr, c_void_p), stream)
    if rc != 0:
        raise LpiError(f"lpi_attn_pooled_{'bwd' if backward else 'fwd'}_pair failed with code {rc}")


def launch_count() -> int:
    return int(load().lpi_launch_count())
