// Fused GEMM epilogue shared by the 128x128 and 256x256 kernels: the lane holds 4 consecutive output columns of one row.
#pragma once
#include "common.h"

// Storage type of what the QuickGELU epilogue saves for the backward (`aux` = gelu'(u)): the operand type, except in f16 operand mode, where it is kept in bf16 —
// it is read only by the BACKWARD's gelu'(u) epilogue, whose operands (gradients) are bf16 (they do not fit fp16's range).
template <typename T> struct AuxT { typedef T type; };
template <> struct AuxT<f16_t> { typedef bf16_t type; };

// RES (residual present) and SAVE_U (QuickGELU pre-activation wanted) are COMPILE-TIME: a run-time "pointer or not" test per
// element makes hipcc branch around every load and wait vmcnt(0) each time — 32 serial HBM round trips per lane
// (cdna_hip_programming.md, "Three .s-level traps" (c)).  Without branches the unrolled loads are batched.
// NTC: the output C is stored with a STREAMING (non-temporal) store.  Measured on the whole step: +3.2 % with every GEMM output streamed
// (26.19 -> 25.37 ms: the outputs are 84-335 MB, the next kernel reads them through L2 misses either way, and without write-allocated
// lines the A / B tiles of the running GEMM stay in the XCD L2); the persistent kernel chooses per problem (gemm256p.hip).
#ifndef LPI_NO_NT_C
#define LPI_NTC_DEFAULT true
#else
#define LPI_NTC_DEFAULT false
#endif
template <typename T, typename TC, int EPI, bool RES, bool SAVE_U = true, bool NTC = LPI_NTC_DEFAULT>
__device__ __forceinline__ f32x4 gemm_epilogue_store(f32x4 acc, int row, int col, TC* __restrict__ C, int ldc, f32x4 bv, float alpha,
                                                    const float* __restrict__ residual, int ldr, typename AuxT<T>::type* __restrict__ aux,
                                                    int ldaux, f32x4 c1v = f32x4{0.f, 0.f, 0.f, 0.f}, float ln_mu = 0.f, float ln_rs = 1.f) {
    typedef typename AuxT<T>::type TA;
    f32x4 v;
    if constexpr (EPI == LPI_EPI_LN || EPI == LPI_EPI_LN_QUICKGELU) {
        // LayerNorm folded into the GEMM (include/lpi_hip.h): c1v = this lane's four c1 values, (ln_mu, ln_rs) = the row's mean and 1/std —
        // the caller loads them (`residual` = mean[ldr] | rstd[ldr] | c1[N]): the persistent kernel with scalar loads, eight rows at a time.
        // ((alpha rstd) acc + (bias - (mean rstd) c1) was tried: two packed operations fewer per four elements, three more plain ones for the
        // per-row products, which a lane uses for only four elements - no gain)
        // alpha is 1 for these epilogues (lpi_gemm_nt refuses anything else): no multiply by it — one vector operation per element less in an epilogue
        // that is bound by them.  The two multiply-adds are written out (no contraction choice left to the compiler): every instantiation rounds alike, so
        // the forward that saves gelu' and the one that does not (evaluation) give the same bits.  The QuickGELU instantiation that saves nothing keeps an
        // (exact) multiply by alpha = 1 in front: without it hipcc's allocation of the persistent kernel spills 40 bytes per lane there
        // (tests/test_no_spills.py).  -DLPI_LN_ALPHA_MUL=true: the multiply everywhere (A/B).
#ifndef LPI_LN_ALPHA_MUL
#define LPI_LN_ALPHA_MUL false
#endif
        f32x4 a0 = acc;
        if constexpr (LPI_LN_ALPHA_MUL || (EPI == LPI_EPI_LN_QUICKGELU && !SAVE_U)) a0 = acc * alpha;
        const f32x4 nmu = f32x4{-ln_mu, -ln_mu, -ln_mu, -ln_mu}, rs4 = f32x4{ln_rs, ln_rs, ln_rs, ln_rs};
        v = __builtin_elementwise_fma(__builtin_elementwise_fma(nmu, c1v, a0), rs4, bv);
    } else {
        v = acc * alpha + bv;
    }
    if constexpr (EPI == LPI_EPI_QUICKGELU || EPI == LPI_EPI_LN_QUICKGELU) {
        // aux (if wanted) receives the DERIVATIVE gelu'(u) = s (1 + 1.702 u (1 - s)), not u: the backward's d c_proj epilogue then is a plain
        // multiply.  gelu' shares this epilogue's sigmoid, and this epilogue's vector work hides behind its two output stores, whereas the
        // backward's gelu'(u) evaluation was exposed (5 us of 27 per 256x256 tile of the d c_proj GEMM).
        if constexpr (SAVE_U) {
            f32x4 g, dg;
#if LPI_IEEE_DIV || defined(LPI_SCALAR_GELU)
#pragma unroll
            for (int j = 0; j < 4; ++j) { g[j] = quick_gelu(v[j]); dg[j] = quick_gelu_grad(v[j]); }
#else
            quick_gelu_both_x4(v, g, dg);
#endif
            st4_nt<TA>(aux + (size_t)row * ldaux + col, dg);      // read again only by the backward, a whole forward later: streaming store
            v = g;
        } else {
#if LPI_IEEE_DIV || defined(LPI_SCALAR_GELU)      /* A/B: the one-value-at-a-time form */
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = quick_gelu(v[j]);
#else
            v = quick_gelu_x4(v);
#endif
        }
    } else if constexpr (EPI == LPI_EPI_DQUICKGELU) {
        v *= Elem<TA>::ld4(aux + (size_t)row * ldaux + col);      // aux = gelu'(u), saved by the forward's QuickGELU epilogue
    }
    if constexpr (RES) {
        // the residual stream has C's storage type when C is fp16 (bf16 mode), f32 otherwise; ldr counts elements of that type
        if constexpr (sizeof(TC) == 2 && !__is_same(TC, bf16_t)) v += Elem<TC>::ld4(reinterpret_cast<const TC*>(residual) + (size_t)row * ldr + col);
        else v += *reinterpret_cast<const f32x4*>(residual + (size_t)row * ldr + col);
    }
    // the fp16 residual stream with plain stores (the next GEMM's A operand: gemm256p.hip has the measurement), everything else streams
    if constexpr (NTC && !(RES && __is_same(TC, f16_t))) st4_nt<TC>(C + (size_t)row * ldc + col, v);
    else Elem<TC>::st4(C + (size_t)row * ldc + col, v);
    return v;      // what was stored, before the rounding to TC (the row-statistics epilogue sums the rounded values)
}
