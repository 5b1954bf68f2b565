// Multi-head attention for LONG sequences (288 < L <= 1024 tokens, non-causal, uniform): forward and backward tiled over the keys with an online softmax.
//
// replaces: nn.MultiheadAttention of the reference's vision tower (retrieval/models/clip/model.py:183-185) at the one CLIP ViT of clip.available_models()
// (clip.py:30-40) whose sequences do not fit the one-workgroup-per-(sample, head) kernels of attention.hip / attention4.hip: ViT-L/14@336px, 577 tokens + prompts.
// Those kernels keep a head's K and V in LDS (L <= 288); here a workgroup owns 64 rows of one (sample, head) — 4 waves of 16 — and walks the OTHER index in
// blocks of 32 (2-byte types) / 16 (f32):
//   forward     own = queries, walk keys:    S = Q K^T / 8 (MFMA), running max / sum per query, O += P V (MFMA), lse = max + log sum
//   backward dQ own = queries, walk keys:    P = exp(S - lse), dP = dO V^T, dS = P (dP - delta) / 8, dQ += dS K; also writes delta = rowsum(dO o O)
//   backward dK, dV  own = keys, walk queries: the same products transposed (S^T = K Q^T), dV += P^T dO, dK += dS^T Q
// One layout trick carries all three: with the operands swapped (A = the walked rows, B = the own rows) a lane holds, of its own row (lane & 15), the four
// consecutive walked positions 4 (lane >> 4) + i of each 16-wide score tile — which is exactly one k-group of the NEXT product's B operand (P, dS or their
// transposes), so the probabilities go from accumulators to MFMA operand without leaving the registers, and softmax statistics are per lane (+ two cross-lane
// steps).  The walked block is staged in LDS in the MFMA operand type, row-major (score products) and transposed (accumulate products), converted on the way
// in: f32 -> f32, bf16 -> bf16, fp16 -> fp16 (forward of the f16 mode), fp16 -> bf16 (its backward: saved q / k / v / ctx fp16, gradients bf16).
// A correctness-first kernel family (the benchmarked configurations never reach it); deterministic, no atomics.
#include "common.h"
#include "../../include/lpi_hip.h"

namespace {

constexpr int HD = 64;
constexpr int NTHR = 256;
constexpr float SCALE = 0.125f;      // 1 / sqrt(64)

template <typename TM>
struct LT {
    static constexpr int EPC = Elem<TM>::EPC;             // elements per 16-byte chunk: 4 / 8
    static constexpr int NB = 4 * EPC;                    // walked rows per block = one MFMA k-step of the accumulate products: 16 / 32
    static constexpr int NTB = NB / 16;                   // 16-wide score tiles per block
    static constexpr int KSD = HD / (4 * EPC);            // k-steps over the head dimension: 4 / 2
    static constexpr int RS = HD * (int)sizeof(TM) + 16;  // row-major image: bytes per row
    static constexpr int RST = NB * (int)sizeof(TM) + 16; // transposed image: bytes per row (one head-dimension index)
    static constexpr int IMG = NB * RS;
    static constexpr int IMGT = HD * RST;
};

// NB rows x 64 columns of a [rows, ld] matrix (this head's columns) -> LDS in the MFMA type TM: row-major image and / or transposed image; rows >= nvalid are zero
template <typename TS, typename TM, bool ROWM, bool TRANS>
__device__ __forceinline__ void stage_block(const TS* __restrict__ src, int ld, int nvalid, char* img, char* imgt) {
    constexpr int NB = LT<TM>::NB;
    for (int i = threadIdx.x; i < NB * 8; i += NTHR) {
        const int row = i >> 3, grp = i & 7;
        f32x4 lo = f32x4{0.f, 0.f, 0.f, 0.f}, hi = lo;
        if (row < nvalid) {
            lo = Elem<TS>::ld4(src + (size_t)row * ld + grp * 8);
            hi = Elem<TS>::ld4(src + (size_t)row * ld + grp * 8 + 4);
        }
        if constexpr (ROWM) {
            TM* d = reinterpret_cast<TM*>(img + row * LT<TM>::RS) + grp * 8;
            Elem<TM>::st4(d, lo);
            Elem<TM>::st4(d + 4, hi);
        }
        if constexpr (TRANS) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                Elem<TM>::st(reinterpret_cast<TM*>(imgt + (grp * 8 + e) * LT<TM>::RST) + row, lo[e]);
                Elem<TM>::st(reinterpret_cast<TM*>(imgt + (grp * 8 + 4 + e) * LT<TM>::RST) + row, hi[e]);
            }
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

__device__ __forceinline__ float quad_max(float v) { return fmaxf(fmaxf(v, __shfl_xor(v, 16)), fmaxf(__shfl_xor(v, 32), __shfl_xor(v, 48))); }
__device__ __forceinline__ float quad_sum(float v) {
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}

struct LongArgs {
    int B, L, H;
    const void* qkv; int ldqkv;
    const void* ctx; int ldctx;      // forward: output; backward: the saved output
    const void* dctx; int lddctx;
    float* lse; float* delta;        // [B, H, L]
    void* dqkv; int lddqkv;
};

// ---- forward: grid (ceil(L / 64), B H) ---------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(NTHR) void attn_long_fwd_kernel(LongArgs A) {
    typedef LT<T> C;
    __shared__ __attribute__((aligned(16))) char kimg[C::IMG];
    __shared__ __attribute__((aligned(16))) char vimgt[C::IMGT];
    const int L = A.L, H = A.H, bh = blockIdx.y, b = bh / H, h = bh % H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, r16 = lane & 15;
    const int q = blockIdx.x * 64 + wave * 16 + r16;
    const bool qv = q < L;
    const T* base = reinterpret_cast<const T*>(A.qkv) + (size_t)b * L * A.ldqkv + h * HD;
    Chunk qf[C::KSD];
#pragma unroll
    for (int ks = 0; ks < C::KSD; ++ks) qf[ks] = own_frag<T, T>(base + (size_t)(qv ? q : 0) * A.ldqkv, qv, g, ks);
    float m = -INFINITY, l = 0.f;
    f32x4 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < L; k0 += C::NB) {
        __syncthreads();
        stage_block<T, T, true, false>(base + (size_t)k0 * A.ldqkv + H * HD, A.ldqkv, L - k0, kimg, nullptr);
        stage_block<T, T, false, true>(base + (size_t)k0 * A.ldqkv + 2 * H * HD, A.ldqkv, L - k0, nullptr, vimgt);
        __syncthreads();
        f32x4 s[C::NTB];
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < C::NTB; ++t) {
            s[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < C::KSD; ++ks) mma_chunk<T>(s[t], img_frag<T>(kimg, 16 * t + r16, g, ks), qf[ks]);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                s[t][i] = k0 + 16 * t + 4 * g + i < L ? s[t][i] * SCALE : -INFINITY;
                mx = fmaxf(mx, s[t][i]);
            }
        }
        mx = quad_max(mx);
        const float mn = fmaxf(m, mx);      // finite: every block has at least one live key
        float rs = 0.f;
#pragma unroll
        for (int t = 0; t < C::NTB; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                s[t][i] = __expf(s[t][i] - mn);
                rs += s[t][i];
            }
        rs = quad_sum(rs);
        const float alpha = __expf(m - mn);
        l = l * alpha + rs;
        m = mn;
        const Chunk pc = acc_chunk<T>(s);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            o[dt] *= alpha;
            mma_chunk<T>(o[dt], imgt_frag<T>(vimgt, 16 * dt + r16, g), pc);
        }
    }
    if (qv) {
        const float inv = 1.f / l;
        T* dst = reinterpret_cast<T*>(const_cast<void*>(A.ctx)) + ((size_t)b * L + q) * A.ldctx + h * HD + 4 * g;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) Elem<T>::st4(dst + 16 * dt, o[dt] * inv);
        if (g == 0) A.lse[((size_t)b * H + h) * L + q] = m + __logf(l);
    }
}

// ---- backward, dQ: grid (ceil(L / 64), B H); TS = type of the saved q / k / v / ctx, TG = type of dctx / dqkv and of the MFMA operands -----------------
template <typename TS, typename TG>
__global__ __launch_bounds__(NTHR) void attn_long_bwd_dq_kernel(LongArgs A) {
    typedef LT<TG> C;
    __shared__ __attribute__((aligned(16))) char kimg[C::IMG];
    __shared__ __attribute__((aligned(16))) char vimg[C::IMG];
    __shared__ __attribute__((aligned(16))) char kimgt[C::IMGT];
    const int L = A.L, H = A.H, bh = blockIdx.y, b = bh / H, h = bh % H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, r16 = lane & 15;
    const int q = blockIdx.x * 64 + wave * 16 + r16;
    const bool qv = q < L;
    const size_t qrow = (size_t)b * L + (qv ? q : 0);
    const TS* base = reinterpret_cast<const TS*>(A.qkv) + (size_t)b * L * A.ldqkv + h * HD;
    const TS* orow = reinterpret_cast<const TS*>(A.ctx) + qrow * A.ldctx + h * HD;
    const TG* drow = reinterpret_cast<const TG*>(A.dctx) + qrow * A.lddctx + h * HD;
    Chunk qf[C::KSD], df[C::KSD];
    float dl = 0.f;
#pragma unroll
    for (int ks = 0; ks < C::KSD; ++ks) {
        qf[ks] = own_frag<TS, TG>(base + (size_t)(qv ? q : 0) * A.ldqkv, qv, g, ks);
        df[ks] = own_frag<TG, TG>(drow, qv, g, ks);
        if (qv) {      // delta = sum_d dO O over the row: this lane's chunks, then the row's four lanes
#pragma unroll
            for (int e = 0; e < C::EPC; e += 4) {
                const f32x4 a = Elem<TG>::ld4(drow + (g + 4 * ks) * C::EPC + e), o4 = Elem<TS>::ld4(orow + (g + 4 * ks) * C::EPC + e);
                dl += a[0] * o4[0] + a[1] * o4[1] + a[2] * o4[2] + a[3] * o4[3];
            }
        }
    }
    dl = quad_sum(dl);
    const size_t sidx = ((size_t)b * H + h) * L + (qv ? q : 0);
    const float lse = qv ? A.lse[sidx] : 0.f;
    if (qv && g == 0) A.delta[sidx] = dl;
    f32x4 dq[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < L; k0 += C::NB) {
        __syncthreads();
        stage_block<TS, TG, true, true>(base + (size_t)k0 * A.ldqkv + H * HD, A.ldqkv, L - k0, kimg, kimgt);
        stage_block<TS, TG, true, false>(base + (size_t)k0 * A.ldqkv + 2 * H * HD, A.ldqkv, L - k0, vimg, nullptr);
        __syncthreads();
        f32x4 ds[C::NTB];
#pragma unroll
        for (int t = 0; t < C::NTB; ++t) {
            f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f}, dp = s;
#pragma unroll
            for (int ks = 0; ks < C::KSD; ++ks) {
                mma_chunk<TG>(s, img_frag<TG>(kimg, 16 * t + r16, g, ks), qf[ks]);
                mma_chunk<TG>(dp, img_frag<TG>(vimg, 16 * t + r16, g, ks), df[ks]);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float p = (qv && k0 + 16 * t + 4 * g + i < L) ? __expf(s[i] * SCALE - lse) : 0.f;
                ds[t][i] = p * (dp[i] - dl) * SCALE;
            }
        }
        const Chunk dc = acc_chunk<TG>(ds);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) mma_chunk<TG>(dq[dt], imgt_frag<TG>(kimgt, 16 * dt + r16, g), dc);
    }
    if (qv) {
        TG* dst = reinterpret_cast<TG*>(A.dqkv) + ((size_t)b * L + q) * A.lddqkv + h * HD + 4 * g;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) Elem<TG>::st4(dst + 16 * dt, dq[dt]);
    }
}

// ---- backward, dK and dV: grid (ceil(L / 64), B H); own rows = keys, the queries are walked (needs delta of the dQ kernel) ------------------------------
template <typename TS, typename TG>
__global__ __launch_bounds__(NTHR) void attn_long_bwd_dkv_kernel(LongArgs A) {
    typedef LT<TG> C;
    __shared__ __attribute__((aligned(16))) char qimg[C::IMG];
    __shared__ __attribute__((aligned(16))) char dimg[C::IMG];
    __shared__ __attribute__((aligned(16))) char qimgt[C::IMGT];
    __shared__ __attribute__((aligned(16))) char dimgt[C::IMGT];
    __shared__ float lse_l[C::NB], dl_l[C::NB];
    const int L = A.L, H = A.H, bh = blockIdx.y, b = bh / H, h = bh % H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, r16 = lane & 15;
    const int k = blockIdx.x * 64 + wave * 16 + r16;
    const bool kv = k < L;
    const TS* base = reinterpret_cast<const TS*>(A.qkv) + (size_t)b * L * A.ldqkv + h * HD;
    const TG* dbase = reinterpret_cast<const TG*>(A.dctx) + (size_t)b * L * A.lddctx + h * HD;
    Chunk kf[C::KSD], vf[C::KSD];
#pragma unroll
    for (int ks = 0; ks < C::KSD; ++ks) {
        kf[ks] = own_frag<TS, TG>(base + (size_t)(kv ? k : 0) * A.ldqkv + H * HD, kv, g, ks);
        vf[ks] = own_frag<TS, TG>(base + (size_t)(kv ? k : 0) * A.ldqkv + 2 * H * HD, kv, g, ks);
    }
    f32x4 dk[4], dv[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dk[dt] = dv[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const size_t s0 = ((size_t)b * H + h) * L;
    for (int q0 = 0; q0 < L; q0 += C::NB) {
        __syncthreads();
        stage_block<TS, TG, true, true>(base + (size_t)q0 * A.ldqkv, A.ldqkv, L - q0, qimg, qimgt);
        stage_block<TG, TG, true, true>(dbase + (size_t)q0 * A.lddctx, A.lddctx, L - q0, dimg, dimgt);
        if ((int)threadIdx.x < C::NB) {
            const bool v = q0 + (int)threadIdx.x < L;
            lse_l[threadIdx.x] = v ? A.lse[s0 + q0 + threadIdx.x] : INFINITY;      // exp(s - inf) = 0 for the rows behind L
            dl_l[threadIdx.x] = v ? A.delta[s0 + q0 + threadIdx.x] : 0.f;
        }
        __syncthreads();
        f32x4 pt[C::NTB], dst_[C::NTB];
#pragma unroll
        for (int t = 0; t < C::NTB; ++t) {
            f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f}, dp = s;
#pragma unroll
            for (int ks = 0; ks < C::KSD; ++ks) {
                mma_chunk<TG>(s, img_frag<TG>(qimg, 16 * t + r16, g, ks), kf[ks]);       // s[i] = score(key r16, query 16 t + 4 g + i)
                mma_chunk<TG>(dp, img_frag<TG>(dimg, 16 * t + r16, g, ks), vf[ks]);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int qi = 16 * t + 4 * g + i;
                const float p = __expf(s[i] * SCALE - lse_l[qi]);
                pt[t][i] = p;
                dst_[t][i] = p * (dp[i] - dl_l[qi]) * SCALE;
            }
        }
        const Chunk pc = acc_chunk<TG>(pt), dc = acc_chunk<TG>(dst_);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            mma_chunk<TG>(dv[dt], imgt_frag<TG>(dimgt, 16 * dt + r16, g), pc);
            mma_chunk<TG>(dk[dt], imgt_frag<TG>(qimgt, 16 * dt + r16, g), dc);
        }
    }
    if (kv) {
        TG* dst = reinterpret_cast<TG*>(A.dqkv) + ((size_t)b * L + k) * A.lddqkv + h * HD + 4 * g;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            Elem<TG>::st4(dst + H * HD + 16 * dt, dk[dt]);
            Elem<TG>::st4(dst + 2 * H * HD + 16 * dt, dv[dt]);
        }
    }
}

}  // namespace

bool lpi_attn_long_ok(int L, int causal, const void* row_start) { return !causal && !row_start && L > 288 && L <= 1024; }

int lpi_attn_long_fwd(int dtype, int B, int L, int H, const void* qkv, int ldqkv, void* ctx, int ldctx, float* lse, hipStream_t s) {
    const LongArgs A{B, L, H, qkv, ldqkv, ctx, ldctx, nullptr, 0, lse, nullptr, nullptr, 0};
    const dim3 grid((L + 63) / 64, B * H);
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
    const dim3 grid((L + 63) / 64, B * H);
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
