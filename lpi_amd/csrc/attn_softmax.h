// Online-softmax step of one 32-key tile, shared by the attention forwards (attention.hip: one head per workgroup; attention2.hip: persistent).
// Scores are TRANSPOSED tiles (mfma(K, Q)): lane l holds query column (l & 15), keys kb + 4 (l >> 4) + 0..3 (s0) and + 16 (s1), for NB 16-query
// blocks.  replaces: the softmax inside nn.MultiheadAttention (retrieval/models/clip/model.py:183-185).
#pragma once
#include "common.h"

// max of three.  The attention sources are compiled with -fno-honor-nans (build.sh): with NaNs honoured every fmaxf on an MFMA result gets a
// canonicalising v_max x, x in front (11 extra instructions per 16-query block and tile); without, the chain below is v_max3_f32.  (An
// inline-asm v_max3_f32 is NOT an option: the compiler does not place the MFMA -> VALU read wait states for asm operands — stale scores,
// run-dependent bits and overflowing p were the result.)
__device__ __forceinline__ float attn_max3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
// maximum over the 4 lanes sharing (lane & 15): lane ^ 16 by v_permlane16_swap, lane ^ 32 by v_permlane32_swap
__device__ __forceinline__ float attn_group_max(float v) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

#ifndef LPI_ATTN_DEFER_MAX
#define LPI_ATTN_DEFER_MAX 1
#endif
constexpr float ATTN_DEFER_THR = 8.0f;      // log2 domain: p <= 256

// In: raw scores s0 / s1 (f32 accumulators).  Out: p = exp2(c s - m) in their place; m (scaled log2 domain), lsum (this lane's partial row sum) and
// the running output o rescaled.  c = scale * log2(e).
//
// DEFERRED running maximum (LPI_ATTN_DEFER_MAX, default): the reference point m of a row moves only when some score of the tile exceeds it by
// more than ATTN_DEFER_THR, so p <= 2^THR instead of <= 1 — the same softmax (any reference point gives the same quotient, and the bf16 / f32
// roundings are scale-free), but in all tiles but the first few the cross-lane maximum, the rescale factor and its branch per 16-query block
// shrink to ONE lane-local max3 chain and one wave-uniform test: ~114 instead of ~183 instructions per tile (the forward's compute phase is
// vector-issue bound at 3.5 waves per SIMD; measured 96 -> 86 us on the vision shape).  The decision is wave-uniform (a ballot), so a row's
// bits depend on which rows share its wave — the row -> wave map is a function of L alone (wave w owns rows 32 w ..), so results are
// reproducible run to run and equal between the one-problem and the pair launch.  Floating-point contraction is off inside: where the
// compiler fuses a multiply into a later add must not depend on the instantiation.
template <int NB, bool MASKED, bool CAUSAL>
__device__ __forceinline__ void attn_softmax_tile(f32x4 (&s0)[NB], f32x4 (&s1)[NB], float (&m)[NB], float (&lsum)[NB], f32x4 (&o)[NB][4], int kb, int g,
                                                  int L, const int (&qrow)[NB], float c) {
#pragma clang fp contract(off)
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        if constexpr (MASKED) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k0 = kb + 4 * g + r, k1 = k0 + 16;
                if (!(k0 < L && (!CAUSAL || k0 <= qrow[j]))) s0[j][r] = -INFINITY;
                if (!(k1 < L && (!CAUSAL || k1 <= qrow[j]))) s1[j][r] = -INFINITY;
            }
        }
    }
#if LPI_ATTN_DEFER_MAX
    float mt[NB];
    bool grow = false;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        mt[j] = attn_max3(attn_max3(s0[j][0], s0[j][1], s0[j][2]), attn_max3(s0[j][3], s1[j][0], s1[j][1]), attn_max3(s1[j][2], s1[j][3], s1[j][3]));
        grow |= fmaf(mt[j], c, -ATTN_DEFER_THR) > m[j];
    }
    if (__builtin_amdgcn_ballot_w64(grow) != 0) {       // wave-uniform; always at the first tile (m = -inf), rarely later
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const float mn = fmaxf(m[j], attn_group_max(mt[j]) * c);
            const float msafe = (mn == -INFINITY) ? 0.f : mn;      // nothing but masked keys so far
            const float alpha = __builtin_amdgcn_exp2f(m[j] - msafe);           // m = -inf: 0 (lsum and o are 0 then)
            lsum[j] *= alpha;
#pragma unroll
            for (int i = 0; i < 4; ++i) o[j][i] *= alpha;
            m[j] = mn;
        }
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const float msafe = (MASKED && m[j] == -INFINITY) ? 0.f : m[j];
        float ps = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            s0[j][r] = __builtin_amdgcn_exp2f(fmaf(s0[j][r], c, -msafe));
            s1[j][r] = __builtin_amdgcn_exp2f(fmaf(s1[j][r], c, -msafe));
            ps += s0[j][r] + s1[j][r];
        }
        lsum[j] += ps;
    }
#else       /* the classic form: exact running maximum, one rescale factor per tile and block (A/B) */
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        float mt = fmaxf(fmaxf(fmaxf(s0[j][0], s0[j][1]), fmaxf(s0[j][2], s0[j][3])), fmaxf(fmaxf(s1[j][0], s1[j][1]), fmaxf(s1[j][2], s1[j][3])));
        mt = attn_group_max(mt) * c;
        const float mn = fmaxf(m[j], mt);
        const float msafe = (MASKED && mn == -INFINITY) ? 0.f : mn;   // fully masked so far (padded / causal-early rows)
        const float alpha = __builtin_amdgcn_exp2f(m[j] - msafe);
        float ps = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            s0[j][r] = __builtin_amdgcn_exp2f(fmaf(s0[j][r], c, -msafe));
            s1[j][r] = __builtin_amdgcn_exp2f(fmaf(s1[j][r], c, -msafe));
            ps += s0[j][r] + s1[j][r];
        }
        lsum[j] = fmaf(lsum[j], alpha, ps);
        m[j] = mn;
        // rescale the running output only when some lane's maximum moved (wave-uniform test; x * 1.0f is exact)
        if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) o[j][i] *= alpha;
        }
    }
#endif
}
