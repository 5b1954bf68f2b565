// Multi-head attention for LONG sequences (288 < L <= 1024 tokens, non-causal, uniform): forward and backward tiled over the keys with an online softmax.
//
// replaces: nn.MultiheadAttention of the reference's vision tower (retrieval/models/clip/model.py:183-185) at the one CLIP ViT of clip.available_models()
// (clip.py:30-40) whose sequences do not fit the one-workgroup-per-(sample, head) kernels of attention.hip / attention4.hip: ViT-L/14@336px, 577 tokens + prompts.
// Those kernels keep a head's K and V in LDS (L <= 288); here a workgroup owns 128 rows of one (sample, head) — 4 waves of two 16-row tiles — and walks the OTHER index in
// blocks of 32 (2-byte types) / 16 (f32):
//   forward     own = queries, walk keys:    S = Q K^T / 8 (MFMA), running max / sum per query, O += P V (MFMA), lse = max + log sum
//   backward dQ own = queries, walk keys:    P = exp(S - lse), dP = dO V^T, dS = P (dP - delta) / 8, dQ += dS K; also writes delta = rowsum(dO o O)
//   backward dK, dV  own = keys, walk queries: the same products transposed (S^T = K Q^T), dV += P^T dO, dK += dS^T Q
// One layout trick carries all three: with the operands swapped (A = the walked rows, B = the own rows) a lane holds, of its own row (lane & 15), the four
// consecutive walked positions 4 (lane >> 4) + i of each 16-wide score tile — which is exactly one k-group of the NEXT product's B operand (P, dS or their
// transposes), so the probabilities go from accumulators to MFMA operand without leaving the registers, and softmax statistics are per lane (+ two cross-lane
// steps).  The walked block is staged in LDS in the MFMA operand type, row-major — the accumulate products read it through the transposing LDS read (f32: a second,
// transposed image) —, converted on the way
// in: f32 -> f32, bf16 -> bf16, fp16 -> fp16 (forward of the f16 mode), fp16 -> bf16 (its backward: saved q / k / v / ctx fp16, gradients bf16).
// A correctness-first kernel family (the benchmarked configurations never reach it), then tuned where it was cheap (OwnTiles below); deterministic, no atomics.
#include "common.h"
#include "../../include/lpi_hip.h"

namespace {

constexpr int HD = 64;
constexpr int NTHR = 256;
constexpr float SCALE = 0.125f;      // 1 / sqrt(64)
constexpr float C2 = SCALE * 1.44269504088896341f;      // scores in the scaled log2 domain: exp(s / 8 - x) = exp2(C2 s - x log2 e), one multiply and v_exp_f32
constexpr float LOG2E_ = 1.44269504088896341f, LN2_ = 0.69314718055994531f;

template <typename TM>
struct LT {
    static constexpr int EPC = Elem<TM>::EPC;             // elements per 16-byte chunk: 4 / 8
    static constexpr int NB = 4 * EPC;                    // walked rows per block = one MFMA k-step of the accumulate products: 16 / 32
    static constexpr int NTB = NB / 16;                   // 16-wide score tiles per block
    static constexpr int KSD = HD / (4 * EPC);            // k-steps over the head dimension: 4 / 2
    static constexpr int RS = HD * (int)sizeof(TM) + (sizeof(TM) == 2 ? 32 : 16);  // row-major image: bytes per row (160 for the 2-byte types, as attention.hip: the transposing reads)
    static constexpr int RST = NB * (int)sizeof(TM) + 16; // transposed image: bytes per row (one head-dimension index)
    static constexpr int IMG = NB * RS;
    static constexpr int IMGT = HD * RST;
};

// NB rows x 64 columns of a [rows, ld] matrix (this head's columns) -> LDS in the MFMA type TM: row-major image and / or transposed image; rows >= nvalid are
// zero.  In two halves, so that the NEXT block's global loads are in flight while this block is multiplied (a thread owns 8 consecutive elements of one row).
struct BlockRegs { f32x4 lo, hi; };
template <typename TS, typename TM>
__device__ __forceinline__ BlockRegs load_block(const TS* __restrict__ src, int ld, int nvalid) {
    const int i = threadIdx.x, row = i >> 3, grp = i & 7;
    BlockRegs r;
    r.lo = r.hi = f32x4{0.f, 0.f, 0.f, 0.f};
    if (i < LT<TM>::NB * 8 && row < nvalid) {
        r.lo = Elem<TS>::ld4(src + (size_t)row * ld + grp * 8);
        r.hi = Elem<TS>::ld4(src + (size_t)row * ld + grp * 8 + 4);
    }
    return r;
}
template <typename TM, bool ROWM, bool TRANS>
__device__ __forceinline__ void store_block(const BlockRegs& r, char* img, char* imgt) {
    const int i = threadIdx.x, row = i >> 3, grp = i & 7;
    if (i >= LT<TM>::NB * 8) return;
    if constexpr (ROWM) {
        TM* d = reinterpret_cast<TM*>(img + row * LT<TM>::RS) + grp * 8;
        Elem<TM>::st4(d, r.lo);
        Elem<TM>::st4(d + 4, r.hi);
    }
    if constexpr (TRANS) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            Elem<TM>::st(reinterpret_cast<TM*>(imgt + (grp * 8 + e) * LT<TM>::RST) + row, r.lo[e]);
            Elem<TM>::st(reinterpret_cast<TM*>(imgt + (grp * 8 + 4 + e) * LT<TM>::RST) + row, r.hi[e]);
        }
    }
}

// fragment k-step ks of one of the lane's OWN rows, from global memory (converted to TM): elements (g + 4 ks) EPC ... of the row
template <typename TS, typename TM>
__device__ __forceinline__ Chunk own_frag(const TS* __restrict__ row, bool valid, int g, int ks) {
    constexpr int EPC = LT<TM>::EPC;
    Chunk c;
    c.u = make_uint4(0, 0, 0, 0);
    if (!valid) return c;
    const TS* p = row + (g + 4 * ks) * EPC;
    if constexpr (EPC == 4) {
        c.f = Elem<TS>::ld4(p);
    } else {
        const f32x4 a = Elem<TS>::ld4(p), b = Elem<TS>::ld4(p + 4);
        c.u = make_uint4(pack2_t<TM>(a[0], a[1]), pack2_t<TM>(a[2], a[3]), pack2_t<TM>(b[0], b[1]), pack2_t<TM>(b[2], b[3]));
    }
    return c;
}

template <typename TM>
__device__ __forceinline__ Chunk img_frag(const char* img, int row, int g, int ks) {
    Chunk c;
    c.u = *reinterpret_cast<const uint4*>(img + row * LT<TM>::RS + (g + 4 * ks) * 16);
    return c;
}

// transposed image, row = head-dimension index: the k-group g of the block = walked positions 4 g .. 4 g + 3 of every 16-wide tile
template <typename TM>
__device__ __forceinline__ Chunk imgt_frag(const char* imgt, int dim, int g) {
    Chunk c;
    const char* p = imgt + dim * LT<TM>::RST;
    if constexpr (LT<TM>::EPC == 4) {
        c.u = *reinterpret_cast<const uint4*>(p + g * 16);
    } else {
        const uint2 lo = *reinterpret_cast<const uint2*>(p + g * 8), hi = *reinterpret_cast<const uint2*>(p + 32 + g * 8);
        c.u = make_uint4(lo.x, lo.y, hi.x, hi.y);
    }
    return c;
}

// The accumulate products' A operand: rows = head-dimension index 16 dt + (lane & 15), k-group g = walked positions 4 g .. 4 g + 3 of every 16-wide tile.
// 2-byte types read it TRANSPOSED out of the ROW-MAJOR image (ds_read_b64_tr_b16: lane 4 q + p of a 16-lane group addresses row q, columns 4 p .. 4 p + 3 of a
// 4 x 16 block and receives column 4 q + p of its four rows) — no transposed image, no 2-byte scatter stores (the first version staged one: 16 bank-conflicting
// ds_write_b16 per thread and block cost as much as the block's MFMAs); f32 has no transposing read and keeps the transposed image.
typedef __attribute__((ext_vector_type(4))) short short4v_;
template <typename TM>
__device__ __forceinline__ Chunk acc_operand(const char* img, const char* imgt, int dt, int r16, int g) {
    if constexpr (LT<TM>::EPC == 4) {
        return imgt_frag<TM>(imgt, 16 * dt + r16, g);
    } else {
        const char* tp = img + (4 * g + (r16 >> 2)) * LT<TM>::RS + (r16 & 3) * 8 + dt * 32;
        const short4v_ lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v_*)(tp));
        const short4v_ hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v_*)(tp + 16 * LT<TM>::RS));
        const uint2 lo2 = __builtin_bit_cast(uint2, lo), hi2 = __builtin_bit_cast(uint2, hi);
        Chunk c;
        c.u = make_uint4(lo2.x, lo2.y, hi2.x, hi2.y);
        return c;
    }
}

// the score tiles of a block (accumulator layout: walked positions 4 g + i of tile t) as the B-operand k-group g of the accumulate product
template <typename TM>
__device__ __forceinline__ Chunk acc_chunk(const f32x4 (&t)[LT<TM>::NTB]) {
    Chunk c;
    if constexpr (LT<TM>::EPC == 4) {
        c.f = t[0];
    } else {
        c.u = make_uint4(pack2_t<TM>(t[0][0], t[0][1]), pack2_t<TM>(t[0][2], t[0][3]), pack2_t<TM>(t[1][0], t[1][1]), pack2_t<TM>(t[1][2], t[1][3]));
    }
    return c;
}

// reductions over the 4 lanes sharing (lane & 15): lane ^ 16 by v_permlane16_swap, lane ^ 32 by v_permlane32_swap (with vdst = src = v one instruction leaves
// {own, partner} in its two results for every lane) — vector instructions, where __shfl_xor goes through the LDS crossbar
__device__ __forceinline__ float quad_max(float v) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float quad_sum(float v) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

struct LongArgs {
    int B, L, H;
    const void* qkv; int ldqkv;
    const void* ctx; int ldctx;      // forward: output; backward: the saved output
    const void* dctx; int lddctx;
    float* lse; float* delta;        // [B, H, L]
    void* dqkv; int lddqkv;
};

// NOWN = 16-row tiles of the own index per wave: a staged block of the walked index serves 64 NOWN own rows of the workgroup.  Measured at the ViT-L/14@336px
// shape (L = 593, H = 16, 64 samples, bf16; tools/probe/attn_long_bench.py), forward / backward TFLOP/s: first version (a transposed LDS image written with 2-byte
// stores, __shfl_xor reductions, natural-log arithmetic) 1 / 2 / 4 tiles -> 186 / 197 / 141 and 189 / 228 / 167; the next block prefetched into registers: no change;
// exp2-domain scores, the row sum reduced once after the loop, row constants as initial accumulators in the backward: 207 / 236; transposing reads of the
// row-major image instead of the transposed image + permlane reductions: 231 / 268; 160-byte image rows: 233 / 274.  About 0.6 of the tuned short-sequence
// kernels' rate.  Two image sets with ONE barrier per block: 234 / 271 (no gain: not kept) — what remains is the serial S -> softmax -> PV chain of a wave
// at one or two waves per SIMD (172-260 VGPRs).
template <typename TM> struct OwnTiles { static constexpr int N = 2; };

// ---- forward: grid (ceil(L / (64 NOWN)), B H) -----------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(NTHR) void attn_long_fwd_kernel(LongArgs A) {
    typedef LT<T> C;
    constexpr int NO = OwnTiles<T>::N;
    constexpr bool TR = C::EPC != 4;      // 2-byte types: the accumulate products read the row-major images transposed (acc_operand)
    __shared__ __attribute__((aligned(16))) char kimg[C::IMG];
    __shared__ __attribute__((aligned(16))) char vimg[TR ? C::IMG : C::IMGT];      // V: row-major (2-byte) or transposed (f32)
    const int L = A.L, H = A.H, bh = blockIdx.y, b = bh / H, h = bh % H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, r16 = lane & 15;
    const int q0 = blockIdx.x * (64 * NO) + wave * (16 * NO) + r16;      // own row of tile j: q0 + 16 j
    const T* base = reinterpret_cast<const T*>(A.qkv) + (size_t)b * L * A.ldqkv + h * HD;
    Chunk qf[NO][C::KSD];
    float m[NO], l[NO];
    f32x4 o[NO][4];
#pragma unroll
    for (int j = 0; j < NO; ++j) {
        const int q = q0 + 16 * j;
#pragma unroll
        for (int ks = 0; ks < C::KSD; ++ks) qf[j][ks] = own_frag<T, T>(base + (size_t)(q < L ? q : 0) * A.ldqkv, q < L, g, ks);
        m[j] = -INFINITY;
        l[j] = 0.f;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[j][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    BlockRegs kr = load_block<T, T>(base + H * HD, A.ldqkv, L), vr = load_block<T, T>(base + 2 * H * HD, A.ldqkv, L);
    for (int k0 = 0; k0 < L; k0 += C::NB) {
        __syncthreads();
        store_block<T, true, false>(kr, kimg, nullptr);
        store_block<T, TR, !TR>(vr, vimg, vimg);
        __syncthreads();
        if (k0 + C::NB < L) {      // the next block's rows: in flight under this block's products
            kr = load_block<T, T>(base + (size_t)(k0 + C::NB) * A.ldqkv + H * HD, A.ldqkv, L - k0 - C::NB);
            vr = load_block<T, T>(base + (size_t)(k0 + C::NB) * A.ldqkv + 2 * H * HD, A.ldqkv, L - k0 - C::NB);
        }
        Chunk kfr[C::NTB][C::KSD], vfr[4];
#pragma unroll
        for (int t = 0; t < C::NTB; ++t)
#pragma unroll
            for (int ks = 0; ks < C::KSD; ++ks) kfr[t][ks] = img_frag<T>(kimg, 16 * t + r16, g, ks);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) vfr[dt] = acc_operand<T>(vimg, vimg, dt, r16, g);
#pragma unroll
        for (int j = 0; j < NO; ++j) {
            f32x4 s[C::NTB];
            float mx = -INFINITY;
#pragma unroll
            for (int t = 0; t < C::NTB; ++t) {
                s[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < C::KSD; ++ks) mma_chunk<T>(s[t], kfr[t][ks], qf[j][ks]);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    s[t][i] = k0 + 16 * t + 4 * g + i < L ? s[t][i] * C2 : -INFINITY;
                    mx = fmaxf(mx, s[t][i]);
                }
            }
            mx = quad_max(mx);
            const float mn = fmaxf(m[j], mx);      // finite: every block has at least one live key
            float rs = 0.f;
#pragma unroll
            for (int t = 0; t < C::NTB; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    s[t][i] = __builtin_amdgcn_exp2f(s[t][i] - mn);
                    rs += s[t][i];
                }
            const float alpha = __builtin_amdgcn_exp2f(m[j] - mn);
            l[j] = l[j] * alpha + rs;      // this lane's share of the row sum: the row's four lanes rescale alike (one max), they are added up once, after the loop
            m[j] = mn;
            const Chunk pc = acc_chunk<T>(s);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                o[j][dt] *= alpha;
                mma_chunk<T>(o[j][dt], vfr[dt], pc);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < NO; ++j) {
        const int q = q0 + 16 * j;
        if (q < L) {
            const float ltot = quad_sum(l[j]);
            const float inv = 1.f / ltot;
            T* dst = reinterpret_cast<T*>(const_cast<void*>(A.ctx)) + ((size_t)b * L + q) * A.ldctx + h * HD + 4 * g;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) Elem<T>::st4(dst + 16 * dt, o[j][dt] * inv);
            if (g == 0) A.lse[((size_t)b * H + h) * L + q] = (m[j] + __log2f(ltot)) * LN2_;
        }
    }
}

// ---- backward, dQ: grid (ceil(L / (64 NOWN)), B H); TS = type of the saved q / k / v / ctx, TG = type of dctx / dqkv and of the MFMA operands -----------
template <typename TS, typename TG>
__global__ __launch_bounds__(NTHR) void attn_long_bwd_dq_kernel(LongArgs A) {
    typedef LT<TG> C;
    constexpr int NO = OwnTiles<TG>::N;
    constexpr bool TR = C::EPC != 4;
    __shared__ __attribute__((aligned(16))) char kimg[C::IMG];
    __shared__ __attribute__((aligned(16))) char vimg[C::IMG];
    __shared__ __attribute__((aligned(16))) char kimgt[TR ? 16 : C::IMGT];      // f32 only
    const int L = A.L, H = A.H, bh = blockIdx.y, b = bh / H, h = bh % H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, r16 = lane & 15;
    const int q0 = blockIdx.x * (64 * NO) + wave * (16 * NO) + r16;
    const TS* base = reinterpret_cast<const TS*>(A.qkv) + (size_t)b * L * A.ldqkv + h * HD;
    Chunk qf[NO][C::KSD], df[NO][C::KSD];
    float dl[NO], lse[NO];
    f32x4 dq[NO][4];
#pragma unroll
    for (int j = 0; j < NO; ++j) {
        const int q = q0 + 16 * j;
        const bool qv = q < L;
        const size_t qrow = (size_t)b * L + (qv ? q : 0);
        const TS* orow = reinterpret_cast<const TS*>(A.ctx) + qrow * A.ldctx + h * HD;
        const TG* drow = reinterpret_cast<const TG*>(A.dctx) + qrow * A.lddctx + h * HD;
        float d_ = 0.f;
#pragma unroll
        for (int ks = 0; ks < C::KSD; ++ks) {
            qf[j][ks] = own_frag<TS, TG>(base + (size_t)(qv ? q : 0) * A.ldqkv, qv, g, ks);
            df[j][ks] = own_frag<TG, TG>(drow, qv, g, ks);
            if (qv) {      // delta = sum_d dO O over the row: this lane's chunks, then the row's four lanes
#pragma unroll
                for (int e = 0; e < C::EPC; e += 4) {
                    const f32x4 a = Elem<TG>::ld4(drow + (g + 4 * ks) * C::EPC + e), o4 = Elem<TS>::ld4(orow + (g + 4 * ks) * C::EPC + e);
                    d_ += a[0] * o4[0] + a[1] * o4[1] + a[2] * o4[2] + a[3] * o4[3];
                }
            }
        }
        d_ = quad_sum(d_);
        dl[j] = d_;
        const size_t sidx = ((size_t)b * H + h) * L + (qv ? q : 0);
        lse[j] = qv ? A.lse[sidx] : INFINITY;      // exp(s - inf) = 0: nothing from the rows behind L
        if (qv && g == 0) A.delta[sidx] = d_;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dq[j][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    BlockRegs kr = load_block<TS, TG>(base + H * HD, A.ldqkv, L), vr = load_block<TS, TG>(base + 2 * H * HD, A.ldqkv, L);
    for (int k0 = 0; k0 < L; k0 += C::NB) {
        __syncthreads();
        store_block<TG, true, !TR>(kr, kimg, kimgt);
        store_block<TG, true, false>(vr, vimg, nullptr);
        __syncthreads();
        if (k0 + C::NB < L) {
            kr = load_block<TS, TG>(base + (size_t)(k0 + C::NB) * A.ldqkv + H * HD, A.ldqkv, L - k0 - C::NB);
            vr = load_block<TS, TG>(base + (size_t)(k0 + C::NB) * A.ldqkv + 2 * H * HD, A.ldqkv, L - k0 - C::NB);
        }
        Chunk kfr[C::NTB][C::KSD], vfr[C::NTB][C::KSD], ktf[4];
#pragma unroll
        for (int t = 0; t < C::NTB; ++t)
#pragma unroll
            for (int ks = 0; ks < C::KSD; ++ks) {
                kfr[t][ks] = img_frag<TG>(kimg, 16 * t + r16, g, ks);
                vfr[t][ks] = img_frag<TG>(vimg, 16 * t + r16, g, ks);
            }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) ktf[dt] = acc_operand<TG>(kimg, kimgt, dt, r16, g);
#pragma unroll
        for (int j = 0; j < NO; ++j) {
            f32x4 ds[C::NTB];
#pragma unroll
            for (int t = 0; t < C::NTB; ++t) {
                // 2-byte types: the row constants are the INITIAL accumulators — S - 8 lse and dP - delta come out of the MFMA chains, p = exp2(C2 S') needs no
                // subtraction (a row behind L has lse = +inf: p = 0); the factor 1 / 8 of dS waits for the store.  f32 keeps the explicit form (parity mode).
                constexpr bool INIT = C::EPC != 4;
                const float s_in = INIT ? -8.f * lse[j] : 0.f, d_in = INIT ? -dl[j] : 0.f;
                f32x4 s = f32x4{s_in, s_in, s_in, s_in}, dp = f32x4{d_in, d_in, d_in, d_in};
#pragma unroll
                for (int ks = 0; ks < C::KSD; ++ks) {
                    mma_chunk<TG>(s, kfr[t][ks], qf[j][ks]);
                    mma_chunk<TG>(dp, vfr[t][ks], df[j][ks]);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const bool live = k0 + 16 * t + 4 * g + i < L;
                    if constexpr (INIT) {
                        ds[t][i] = (live ? __builtin_amdgcn_exp2f(s[i] * C2) : 0.f) * dp[i];
                    } else {
                        const float p = live ? __expf(s[i] * SCALE - lse[j]) : 0.f;
                        ds[t][i] = p * (dp[i] - dl[j]);
                    }
                }
            }
            const Chunk dc = acc_chunk<TG>(ds);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) mma_chunk<TG>(dq[j][dt], ktf[dt], dc);
        }
    }
#pragma unroll
    for (int j = 0; j < NO; ++j) {
        const int q = q0 + 16 * j;
        if (q < L) {
            TG* dst = reinterpret_cast<TG*>(A.dqkv) + ((size_t)b * L + q) * A.lddqkv + h * HD + 4 * g;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) Elem<TG>::st4(dst + 16 * dt, dq[j][dt] * SCALE);
        }
    }
}

// ---- backward, dK and dV: grid (ceil(L / (64 NOWN)), B H); own rows = keys, the queries are walked (needs delta of the dQ kernel) ------------------------
template <typename TS, typename TG>
__global__ __launch_bounds__(NTHR) void attn_long_bwd_dkv_kernel(LongArgs A) {
    typedef LT<TG> C;
    constexpr int NO = OwnTiles<TG>::N;
    constexpr bool TR = C::EPC != 4;
    __shared__ __attribute__((aligned(16))) char qimg[C::IMG];
    __shared__ __attribute__((aligned(16))) char dimg[C::IMG];
    __shared__ __attribute__((aligned(16))) char qimgt[TR ? 16 : C::IMGT];      // f32 only
    __shared__ __attribute__((aligned(16))) char dimgt[TR ? 16 : C::IMGT];
    __shared__ float lse_l[C::NB], dl_l[C::NB];
    const int L = A.L, H = A.H, bh = blockIdx.y, b = bh / H, h = bh % H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, r16 = lane & 15;
    const int kk0 = blockIdx.x * (64 * NO) + wave * (16 * NO) + r16;
    const TS* base = reinterpret_cast<const TS*>(A.qkv) + (size_t)b * L * A.ldqkv + h * HD;
    const TG* dbase = reinterpret_cast<const TG*>(A.dctx) + (size_t)b * L * A.lddctx + h * HD;
    Chunk kf[NO][C::KSD], vf[NO][C::KSD];
    f32x4 dk[NO][4], dv[NO][4];
#pragma unroll
    for (int j = 0; j < NO; ++j) {
        const int k = kk0 + 16 * j;
        const bool kv = k < L;
#pragma unroll
        for (int ks = 0; ks < C::KSD; ++ks) {
            kf[j][ks] = own_frag<TS, TG>(base + (size_t)(kv ? k : 0) * A.ldqkv + H * HD, kv, g, ks);
            vf[j][ks] = own_frag<TS, TG>(base + (size_t)(kv ? k : 0) * A.ldqkv + 2 * H * HD, kv, g, ks);
        }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dk[j][dt] = dv[j][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const size_t s0 = ((size_t)b * H + h) * L;
    BlockRegs qr = load_block<TS, TG>(base, A.ldqkv, L), dr = load_block<TG, TG>(dbase, A.lddctx, L);
    float lse_r = INFINITY, dl_r = 0.f;      // exp(s - inf) = 0 for the rows behind L
    if ((int)threadIdx.x < C::NB && (int)threadIdx.x < L) { lse_r = A.lse[s0 + threadIdx.x]; dl_r = A.delta[s0 + threadIdx.x]; }
    for (int q0 = 0; q0 < L; q0 += C::NB) {
        __syncthreads();
        store_block<TG, true, !TR>(qr, qimg, qimgt);
        store_block<TG, true, !TR>(dr, dimg, dimgt);
        if ((int)threadIdx.x < C::NB) {
            lse_l[threadIdx.x] = lse_r;
            dl_l[threadIdx.x] = dl_r;
        }
        __syncthreads();
        if (q0 + C::NB < L) {
            const int qn = q0 + C::NB;
            qr = load_block<TS, TG>(base + (size_t)qn * A.ldqkv, A.ldqkv, L - qn);
            dr = load_block<TG, TG>(dbase + (size_t)qn * A.lddctx, A.lddctx, L - qn);
            lse_r = INFINITY;
            dl_r = 0.f;
            if ((int)threadIdx.x < C::NB && qn + (int)threadIdx.x < L) { lse_r = A.lse[s0 + qn + threadIdx.x]; dl_r = A.delta[s0 + qn + threadIdx.x]; }
        }
        Chunk qfr[C::NTB][C::KSD], dfr[C::NTB][C::KSD], qtf[4], dtf[4];
        f32x4 lq[C::NTB], dlq[C::NTB];
#pragma unroll
        for (int t = 0; t < C::NTB; ++t) {
#pragma unroll
            for (int ks = 0; ks < C::KSD; ++ks) {
                qfr[t][ks] = img_frag<TG>(qimg, 16 * t + r16, g, ks);
                dfr[t][ks] = img_frag<TG>(dimg, 16 * t + r16, g, ks);
            }
            lq[t] = *reinterpret_cast<const f32x4*>(lse_l + 16 * t + 4 * g);
            dlq[t] = *reinterpret_cast<const f32x4*>(dl_l + 16 * t + 4 * g);
        }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            qtf[dt] = acc_operand<TG>(qimg, qimgt, dt, r16, g);
            dtf[dt] = acc_operand<TG>(dimg, dimgt, dt, r16, g);
        }
#pragma unroll
        for (int j = 0; j < NO; ++j) {
            f32x4 pt[C::NTB], dst_[C::NTB];
#pragma unroll
            for (int t = 0; t < C::NTB; ++t) {
                constexpr bool INIT = C::EPC != 4;      // see the dQ kernel
                f32x4 s = INIT ? lq[t] * -8.f : f32x4{0.f, 0.f, 0.f, 0.f}, dp = INIT ? -dlq[t] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < C::KSD; ++ks) {
                    mma_chunk<TG>(s, qfr[t][ks], kf[j][ks]);       // s[i] = score(key r16 of tile j, query 16 t + 4 g + i)
                    mma_chunk<TG>(dp, dfr[t][ks], vf[j][ks]);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if constexpr (INIT) {
                        pt[t][i] = __builtin_amdgcn_exp2f(s[i] * C2);
                        dst_[t][i] = pt[t][i] * dp[i];
                    } else {
                        pt[t][i] = __expf(s[i] * SCALE - lq[t][i]);
                        dst_[t][i] = pt[t][i] * (dp[i] - dlq[t][i]);
                    }
                }
            }
            const Chunk pc = acc_chunk<TG>(pt), dc = acc_chunk<TG>(dst_);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                mma_chunk<TG>(dv[j][dt], dtf[dt], pc);
                mma_chunk<TG>(dk[j][dt], qtf[dt], dc);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < NO; ++j) {
        const int k = kk0 + 16 * j;
        if (k < L) {
            TG* dst = reinterpret_cast<TG*>(A.dqkv) + ((size_t)b * L + k) * A.lddqkv + h * HD + 4 * g;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                Elem<TG>::st4(dst + H * HD + 16 * dt, dk[j][dt] * SCALE);
                Elem<TG>::st4(dst + 2 * H * HD + 16 * dt, dv[j][dt]);
            }
        }
    }
}

}  // namespace

bool lpi_attn_long_ok(int L, int causal, const void* row_start) { return !causal && !row_start && L > 288 && L <= 1024; }

int lpi_attn_long_fwd(int dtype, int B, int L, int H, const void* qkv, int ldqkv, void* ctx, int ldctx, float* lse, hipStream_t s) {
    const LongArgs A{B, L, H, qkv, ldqkv, ctx, ldctx, nullptr, 0, lse, nullptr, nullptr, 0};
    const int own = 64 * (dtype == LPI_F32 ? OwnTiles<float>::N : OwnTiles<bf16_t>::N);
    const dim3 grid((L + own - 1) / own, B * H);
    if (dtype == LPI_F32) LPI_LAUNCH((attn_long_fwd_kernel<float>), grid, dim3(NTHR), 0, s, A);
    else if (dtype == LPI_BF16) LPI_LAUNCH((attn_long_fwd_kernel<bf16_t>), grid, dim3(NTHR), 0, s, A);
    else if (dtype == LPI_F16) LPI_LAUNCH((attn_long_fwd_kernel<f16_t>), grid, dim3(NTHR), 0, s, A);
    else return LPI_EINVAL;
    LPI_CHECK_LAST();
    return 0;
}

// dtype: LPI_F32 (everything f32), LPI_BF16 (everything bf16), LPI_F16 (saved qkv / ctx fp16; dctx, dqkv and the MFMA operands bf16)
int lpi_attn_long_bwd(int dtype, int B, int L, int H, const void* qkv, int ldqkv, const void* ctx, int ldctx, const void* dctx, int lddctx, const float* lse,
                      float* delta, void* dqkv, int lddqkv, hipStream_t s) {
    const LongArgs A{B, L, H, qkv, ldqkv, ctx, ldctx, dctx, lddctx, const_cast<float*>(lse), delta, dqkv, lddqkv};
    const int own = 64 * (dtype == LPI_F32 ? OwnTiles<float>::N : OwnTiles<bf16_t>::N);
    const dim3 grid((L + own - 1) / own, B * H);
    if (dtype == LPI_F32) {
        LPI_LAUNCH((attn_long_bwd_dq_kernel<float, float>), grid, dim3(NTHR), 0, s, A);
        LPI_LAUNCH((attn_long_bwd_dkv_kernel<float, float>), grid, dim3(NTHR), 0, s, A);
    } else if (dtype == LPI_BF16) {
        LPI_LAUNCH((attn_long_bwd_dq_kernel<bf16_t, bf16_t>), grid, dim3(NTHR), 0, s, A);
        LPI_LAUNCH((attn_long_bwd_dkv_kernel<bf16_t, bf16_t>), grid, dim3(NTHR), 0, s, A);
    } else if (dtype == LPI_F16) {
        LPI_LAUNCH((attn_long_bwd_dq_kernel<f16_t, bf16_t>), grid, dim3(NTHR), 0, s, A);
        LPI_LAUNCH((attn_long_bwd_dkv_kernel<f16_t, bf16_t>), grid, dim3(NTHR), 0, s, A);
    } else return LPI_EINVAL;
    LPI_CHECK_LAST();
    return 0;
}
