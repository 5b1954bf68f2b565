// Prompted multi-head attention (head_dim 64) forward + backward on the gfx950 matrix cores.
//
// replaces: nn.MultiheadAttention as called from ResidualAttentionBlock.attention
// (retrieval/models/clip/model.py:183-185) — softmax(q k^T / 8 + mask) v per (sample, head) over the
// [CLS | prompts | patches] (L = 213) or [SOT | ctx | caption] (L = 77, causal mask model.py:347-353)
// sequence — and its autograd backward.
//
// Design (MI355X first).  Sequences are short (L <= 288), so ONE workgroup owns one (sample, head) and keeps the
// whole K and V (forward, dQ pass) or Q and dO (dK/dV pass) of that head in LDS, row-major with a padded row
// stride; nothing N x N ever touches HBM.  Scores are computed TRANSPOSED (mfma(K, Q)) so that a lane holds one
// query column: the online softmax needs only two 16-lane shuffles, and the probability tile sitting in the
// accumulator registers is directly the B operand of the following P.V product (no LDS round trip).  The
// transposed operand of that product (V^T, K^T, dO^T, Q^T) is read straight from the row-major LDS image with
// ds_read_b64_tr_b16 (bf16) or 4-byte strided reads (f32): no transposed copy is ever staged.
// Backward recomputes P from the saved log-sum-exp and runs as two kernels so that every accumulator stays in
// one wave's registers (no atomics, bitwise reproducible):
//   pass A (a wave owns 16 queries): delta = rowsum(dO*O);  dQ = scale * dS K
//   pass B (a wave owns 16 keys)   : dV = P^T dO;           dK = scale * dS^T Q
// Fragment conventions (cdna_hip_programming.md section 3): 16x16 MFMA tiles, lane l supplies row (l & 15) and
// k-group g = l >> 4 of each operand as one 16-byte chunk; it receives column (l & 15), rows 4g..4g+3.
#include "common.h"

namespace {

constexpr int HD = 64;  // head_dim of every CLIP tower (width / 64 heads, model.py:292)
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;
constexpr float SCALE = 0.125f;  // HD ** -0.5

template <typename T> struct AT;
template <> struct AT<float> {
    static constexpr int KS = 4;        // k-steps of 4 chunks per 64-element row
    static constexpr int RS = 272;      // LDS row stride in bytes (64 f32 + 16 B pad)
};
template <> struct AT<bf16_t> {
    static constexpr int KS = 2;
    static constexpr int RS = 144;      // 64 bf16 + 16 B pad
};

// stage rows [0, L) x 64 elements of a head (global row stride ld elements) into LDS, zero rows [L, Lp)
template <typename T>
__device__ __forceinline__ void stage_rows(char* lds, const T* g, int ld, int L, int Lp) {
    constexpr int NCH = HD * (int)sizeof(T) / 16;
    for (int i = threadIdx.x; i < Lp * NCH; i += blockDim.x) {
        const int row = i / NCH, c = i % NCH;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (row < L) v = *reinterpret_cast<const uint4*>(g + (size_t)row * ld + c * Elem<T>::EPC);
        *reinterpret_cast<uint4*>(lds + row * AT<T>::RS + c * 16) = v;
    }
}

// this lane's KS row chunks (chunk g + 4*ks) of row `row` of a global matrix; zeros if !valid
template <typename T>
__device__ __forceinline__ void load_row_chunks(Chunk (&q)[AT<T>::KS], const T* g, size_t row, int ld, int grp, bool valid) {
#pragma unroll
    for (int ks = 0; ks < AT<T>::KS; ++ks) {
        q[ks].u = make_uint4(0, 0, 0, 0);
        if (valid) q[ks].u = *reinterpret_cast<const uint4*>(g + row * ld + (grp + 4 * ks) * Elem<T>::EPC);
    }
}

// acc(16 rows of LDS matrix starting at r0) x (register operand rows)^T : acc[col = reg operand row][rows = lds rows]
template <typename T>
__device__ __forceinline__ f32x4 mma_lds_rows(const char* lds, int r0, int lane, const Chunk (&b)[AT<T>::KS], bool lds_is_A) {
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    const char* p = lds + (r0 + (lane & 15)) * AT<T>::RS + (lane >> 4) * 16;
#pragma unroll
    for (int ks = 0; ks < AT<T>::KS; ++ks) {
        Chunk a;
        a.u = *reinterpret_cast<const uint4*>(p + ks * 64);
        if (lds_is_A) mma_chunk<T>(acc, a, b[ks]);
        else mma_chunk<T>(acc, b[ks], a);
    }
    return acc;
}

// acc[dt] += X^T[d = 16dt + .., k = r0 .. r0+31] . Preg[k][col], k running over 32 consecutive LDS rows of X.
// p0/p1: the two 16-row accumulator tiles (rows 4g..4g+3 of rows r0.. and r0+16..) holding the B operand.
template <typename T>
__device__ __forceinline__ void mma_transposed(f32x4 (&acc)[4], const char* lds, int r0, int lane, f32x4 p0, f32x4 p1) {
    const int g = lane >> 4;
    if constexpr (sizeof(T) == 2) {
        Chunk b;
        b.h[0] = (__bf16)p0[0]; b.h[1] = (__bf16)p0[1]; b.h[2] = (__bf16)p0[2]; b.h[3] = (__bf16)p0[3];
        b.h[4] = (__bf16)p1[0]; b.h[5] = (__bf16)p1[1]; b.h[6] = (__bf16)p1[2]; b.h[7] = (__bf16)p1[3];
        // ds_read_b64_tr_b16: lane i = 4q+p of a 16-lane group addresses row q, columns 4p..4p+3 of a 4 x 16 block
        // and receives column i of its 4 rows.  Block rows = LDS rows r0 + 4g (+16), block columns = d 16dt..16dt+15.
        const int i = lane & 15;
        const char* base = lds + (r0 + 4 * g + (i >> 2)) * AT<T>::RS + (i & 3) * 8;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(base + dt * 32));
            short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(base + 16 * AT<T>::RS + dt * 32));
            Chunk a;
            const uint2 lo2 = __builtin_bit_cast(uint2, lo), hi2 = __builtin_bit_cast(uint2, hi);
            a.u = make_uint4(lo2.x, lo2.y, hi2.x, hi2.y);
            mma_chunk<T>(acc[dt], a, b);
        }
    } else {
        Chunk b0, b1;
        b0.f = p0;
        b1.f = p1;
        const float* base = reinterpret_cast<const float*>(lds + (r0 + 4 * g) * AT<T>::RS) + (lane & 15);
        constexpr int RSF = AT<T>::RS / 4;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            Chunk a0, a1;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                a0.f[s] = base[s * RSF + dt * 16];
                a1.f[s] = base[(16 + s) * RSF + dt * 16];
            }
            mma_chunk<T>(acc[dt], a0, b0);
            mma_chunk<T>(acc[dt], a1, b1);
        }
    }
}

__device__ __forceinline__ float group_max(float v) {  // over the 4 lanes sharing (lane & 15)
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float group_sum(float v) {
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
}

// ------------------------------------------------------------------------------------------------ forward
template <typename T, bool CAUSAL>
__global__ __launch_bounds__(512) void attn_fwd_kernel(int L, int Lp, int H, const T* __restrict__ qkv, int ldqkv,
                                                      T* __restrict__ ctx, int ldctx, float* __restrict__ lse) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int dm = H * HD;
    const T* qg = qkv + (size_t)b * L * ldqkv + h * HD;
    char* k_lds = smem;
    char* v_lds = smem + Lp * AT<T>::RS;
    stage_rows<T>(k_lds, qg + dm, ldqkv, L, Lp);
    stage_rows<T>(v_lds, qg + 2 * dm, ldqkv, L, Lp);
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int g = lane >> 4;
    const float c = SCALE * LOG2E;
    for (int q0 = wave * 16; q0 < L; q0 += nw * 16) {
        const int qrow = q0 + (lane & 15);
        Chunk q[AT<T>::KS];
        load_row_chunks<T>(q, qg, qrow, ldqkv, g, qrow < L);
        float m = -INFINITY, lsum = 0.f;
        f32x4 o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int kend = CAUSAL ? min(Lp, ((q0 + 15) / 32 + 1) * 32) : Lp;
        for (int kb = 0; kb < kend; kb += 32) {
            f32x4 s0 = mma_lds_rows<T>(k_lds, kb, lane, q, true);
            f32x4 s1 = mma_lds_rows<T>(k_lds, kb + 16, lane, q, true);
            float mt = -INFINITY;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k0 = kb + 4 * g + r, k1 = k0 + 16;
                s0[r] = (k0 < L && (!CAUSAL || k0 <= qrow)) ? s0[r] * c : -INFINITY;
                s1[r] = (k1 < L && (!CAUSAL || k1 <= qrow)) ? s1[r] * c : -INFINITY;
                mt = fmaxf(mt, fmaxf(s0[r], s1[r]));
            }
            mt = group_max(mt);
            const float mn = fmaxf(m, mt);
            const float msafe = (mn == -INFINITY) ? 0.f : mn;  // fully masked so far (padded query rows only)
            const float alpha = exp2f(m - msafe);
            float ps = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                s0[r] = exp2f(s0[r] - msafe);
                s1[r] = exp2f(s1[r] - msafe);
                ps += s0[r] + s1[r];
            }
            lsum = lsum * alpha + ps;
            m = mn;
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] *= alpha;
            mma_transposed<T>(o, v_lds, kb, lane, s0, s1);
        }
        const float ltot = group_sum(lsum);
        const float inv = 1.0f / ltot;
        if (qrow < L) {
            T* dst = ctx + ((size_t)b * L + qrow) * ldctx + h * HD + 4 * g;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) Elem<T>::st4(dst + dt * 16, o[dt] * inv);
            if (g == 0) lse[((size_t)b * H + h) * L + qrow] = (m + log2f(ltot)) * LN2;
        }
    }
}

// ------------------------------------------------------------------------------------------------ backward A
template <typename T, bool CAUSAL>
__global__ __launch_bounds__(512) void attn_bwd_dq_kernel(int L, int Lp, int H, const T* __restrict__ qkv, int ldqkv,
                                                         const T* __restrict__ ctx, int ldctx, const T* __restrict__ dctx, int lddctx,
                                                         const float* __restrict__ lse, float* __restrict__ delta,
                                                         T* __restrict__ dqkv, int lddqkv) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int dm = H * HD;
    const T* qg = qkv + (size_t)b * L * ldqkv + h * HD;
    char* k_lds = smem;
    char* v_lds = smem + Lp * AT<T>::RS;
    stage_rows<T>(k_lds, qg + dm, ldqkv, L, Lp);
    stage_rows<T>(v_lds, qg + 2 * dm, ldqkv, L, Lp);
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int g = lane >> 4;
    const float c = SCALE * LOG2E;
    for (int q0 = wave * 16; q0 < L; q0 += nw * 16) {
        const int qrow = q0 + (lane & 15);
        const bool valid = qrow < L;
        const size_t grow = (size_t)b * L + qrow;
        Chunk q[AT<T>::KS], dO[AT<T>::KS];
        load_row_chunks<T>(q, qg, qrow, ldqkv, g, valid);
        load_row_chunks<T>(dO, dctx + h * HD, grow, lddctx, g, valid);
        float dl = 0.f;
        {
            Chunk oc[AT<T>::KS];
            load_row_chunks<T>(oc, ctx + h * HD, grow, ldctx, g, valid);
#pragma unroll
            for (int ks = 0; ks < AT<T>::KS; ++ks) {
                if constexpr (sizeof(T) == 4) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) dl += oc[ks].f[j] * dO[ks].f[j];
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) dl += (float)oc[ks].h[j] * (float)dO[ks].h[j];
                }
            }
        }
        dl = group_sum(dl);
        const float lq = valid ? lse[((size_t)b * H + h) * L + qrow] * LOG2E : INFINITY;
        if (valid && g == 0) delta[((size_t)b * H + h) * L + qrow] = dl;
        f32x4 dq[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) dq[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int kend = CAUSAL ? min(Lp, ((q0 + 15) / 32 + 1) * 32) : Lp;
        for (int kb = 0; kb < kend; kb += 32) {
            f32x4 s0 = mma_lds_rows<T>(k_lds, kb, lane, q, true);
            f32x4 s1 = mma_lds_rows<T>(k_lds, kb + 16, lane, q, true);
            f32x4 p0 = mma_lds_rows<T>(v_lds, kb, lane, dO, true);
            f32x4 p1 = mma_lds_rows<T>(v_lds, kb + 16, lane, dO, true);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k0 = kb + 4 * g + r, k1 = k0 + 16;
                const float e0 = (k0 < L && (!CAUSAL || k0 <= qrow)) ? exp2f(s0[r] * c - lq) : 0.f;
                const float e1 = (k1 < L && (!CAUSAL || k1 <= qrow)) ? exp2f(s1[r] * c - lq) : 0.f;
                s0[r] = e0 * (p0[r] - dl) * SCALE;
                s1[r] = e1 * (p1[r] - dl) * SCALE;
            }
            mma_transposed<T>(dq, k_lds, kb, lane, s0, s1);
        }
        if (valid) {
            T* dst = dqkv + grow * lddqkv + h * HD + 4 * g;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) Elem<T>::st4(dst + dt * 16, dq[dt]);
        }
    }
}

// ------------------------------------------------------------------------------------------------ backward B
template <typename T, bool CAUSAL>
__global__ __launch_bounds__(512) void attn_bwd_dkv_kernel(int L, int Lp, int H, const T* __restrict__ qkv, int ldqkv,
                                                          const T* __restrict__ dctx, int lddctx, const float* __restrict__ lse,
                                                          const float* __restrict__ delta, T* __restrict__ dqkv, int lddqkv) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int dm = H * HD;
    const T* qg = qkv + (size_t)b * L * ldqkv + h * HD;
    char* q_lds = smem;
    char* do_lds = smem + Lp * AT<T>::RS;
    float* lse_lds = reinterpret_cast<float*>(smem + 2 * Lp * AT<T>::RS);
    float* dl_lds = lse_lds + Lp;
    stage_rows<T>(q_lds, qg, ldqkv, L, Lp);
    stage_rows<T>(do_lds, dctx + (size_t)b * L * lddctx + h * HD, lddctx, L, Lp);
    for (int i = threadIdx.x; i < Lp; i += blockDim.x) {
        lse_lds[i] = i < L ? lse[((size_t)b * H + h) * L + i] * LOG2E : INFINITY;  // padded queries -> P = 0
        dl_lds[i] = i < L ? delta[((size_t)b * H + h) * L + i] : 0.f;
    }
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int g = lane >> 4;
    const float c = SCALE * LOG2E;
    for (int k0 = wave * 16; k0 < L; k0 += nw * 16) {
        const int krow = k0 + (lane & 15);
        const bool valid = krow < L;
        Chunk kk[AT<T>::KS], vv[AT<T>::KS];
        load_row_chunks<T>(kk, qg + dm, krow, ldqkv, g, valid);
        load_row_chunks<T>(vv, qg + 2 * dm, krow, ldqkv, g, valid);
        f32x4 dk[4], dv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { dk[i] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        const int qstart = CAUSAL ? (k0 / 32) * 32 : 0;
        for (int qb = qstart; qb < Lp; qb += 32) {
            // S[q][key]: rows q = qb + 16t + 4g + r, column key = krow
            f32x4 s0 = mma_lds_rows<T>(q_lds, qb, lane, kk, true);
            f32x4 s1 = mma_lds_rows<T>(q_lds, qb + 16, lane, kk, true);
            f32x4 p0 = mma_lds_rows<T>(do_lds, qb, lane, vv, true);
            f32x4 p1 = mma_lds_rows<T>(do_lds, qb + 16, lane, vv, true);
            const f32x4 l0 = *reinterpret_cast<const f32x4*>(lse_lds + qb + 4 * g);
            const f32x4 l1 = *reinterpret_cast<const f32x4*>(lse_lds + qb + 16 + 4 * g);
            const f32x4 d0 = *reinterpret_cast<const f32x4*>(dl_lds + qb + 4 * g);
            const f32x4 d1 = *reinterpret_cast<const f32x4*>(dl_lds + qb + 16 + 4 * g);
            f32x4 e0, e1;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int qa = qb + 4 * g + r, qc = qa + 16;
                e0[r] = (valid && (!CAUSAL || krow <= qa)) ? exp2f(s0[r] * c - l0[r]) : 0.f;
                e1[r] = (valid && (!CAUSAL || krow <= qc)) ? exp2f(s1[r] * c - l1[r]) : 0.f;
                s0[r] = e0[r] * (p0[r] - d0[r]) * SCALE;
                s1[r] = e1[r] * (p1[r] - d1[r]) * SCALE;
            }
            mma_transposed<T>(dv, do_lds, qb, lane, e0, e1);
            mma_transposed<T>(dk, q_lds, qb, lane, s0, s1);
        }
        if (valid) {
            T* dst = dqkv + ((size_t)b * L + krow) * lddqkv + h * HD + 4 * g;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                Elem<T>::st4(dst + dm + dt * 16, dk[dt]);
                Elem<T>::st4(dst + 2 * dm + dt * 16, dv[dt]);
            }
        }
    }
}

inline int pick_waves(int L) {
    const int nqb = (L + 15) / 16;
    const int rounds = (nqb + 7) / 8;
    return (nqb + rounds - 1) / rounds;
}

// allow the full 160 KiB of a CU's LDS for this kernel (once per kernel instance)
template <typename K> int set_lds(K kern, size_t bytes) {
    if (bytes > 160 * 1024) return LPI_EINVAL;
    static bool done = false;   // one static per instantiation (K is a distinct function-pointer VALUE, so key on it)
    static const void* last = nullptr;
    if (done && last == (const void*)kern) return 0;
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    done = true;
    last = (const void*)kern;
    return 0;
}

template <typename T>
int fwd_launch(int B, int L, int H, const void* qkv, int ldqkv, void* ctx, int ldctx, float* lse, int causal, hipStream_t s) {
    const int Lp = (L + 31) / 32 * 32;
    const size_t lds = (size_t)2 * Lp * AT<T>::RS;
    const int thr = 64 * pick_waves(L);
    if (causal) {
        int e = set_lds(attn_fwd_kernel<T, true>, lds);
        if (e) return e;
        LPI_LAUNCH((attn_fwd_kernel<T, true>), dim3(B * H), dim3(thr), lds, s, L, Lp, H, (const T*)qkv, ldqkv, (T*)ctx, ldctx, lse);
    } else {
        int e = set_lds(attn_fwd_kernel<T, false>, lds);
        if (e) return e;
        LPI_LAUNCH((attn_fwd_kernel<T, false>), dim3(B * H), dim3(thr), lds, s, L, Lp, H, (const T*)qkv, ldqkv, (T*)ctx, ldctx, lse);
    }
    LPI_CHECK_LAST();
    return 0;
}

template <typename T, bool CAUSAL>
int bwd_launch(int B, int L, int H, const void* qkv, int ldqkv, const void* ctx, int ldctx, const void* dctx, int lddctx,
               const float* lse, float* delta, void* dqkv, int lddqkv, hipStream_t s) {
    const int Lp = (L + 31) / 32 * 32;
    const size_t ldsA = (size_t)2 * Lp * AT<T>::RS;
    const size_t ldsB = ldsA + (size_t)2 * Lp * sizeof(float);
    const int thr = 64 * pick_waves(L);
    int e = set_lds(attn_bwd_dq_kernel<T, CAUSAL>, ldsA);
    if (e) return e;
    e = set_lds(attn_bwd_dkv_kernel<T, CAUSAL>, ldsB);
    if (e) return e;
    LPI_LAUNCH((attn_bwd_dq_kernel<T, CAUSAL>), dim3(B * H), dim3(thr), ldsA, s, L, Lp, H, (const T*)qkv, ldqkv, (const T*)ctx, ldctx,
                       (const T*)dctx, lddctx, lse, delta, (T*)dqkv, lddqkv);
    LPI_CHECK_LAST();
    LPI_LAUNCH((attn_bwd_dkv_kernel<T, CAUSAL>), dim3(B * H), dim3(thr), ldsB, s, L, Lp, H, (const T*)qkv, ldqkv, (const T*)dctx, lddctx,
                       lse, delta, (T*)dqkv, lddqkv);
    LPI_CHECK_LAST();
    return 0;
}

inline bool bad_attn(int dtype, int B, int L, int H, int ld) {
    const int esz = dtype == LPI_F32 ? 4 : 2;
    return B <= 0 || H <= 0 || L <= 0 || L > 288 || ld < 3 * H * HD || (ld * esz) % 16;
}

}  // namespace

extern "C" int lpi_attn_fwd(int dtype, int B, int L, int H, const void* qkv, int ldqkv, void* ctx, int ldctx, float* lse, int causal,
                            void* stream) {
    if (!qkv || !ctx || !lse || bad_attn(dtype, B, L, H, ldqkv) || ldctx < H * HD || (ldctx & 7)) return LPI_EINVAL;
    if (((uintptr_t)qkv | (uintptr_t)ctx) & 15) return LPI_EINVAL;
    if (dtype == LPI_F32) return fwd_launch<float>(B, L, H, qkv, ldqkv, ctx, ldctx, lse, causal, (hipStream_t)stream);
    if (dtype == LPI_BF16) return fwd_launch<bf16_t>(B, L, H, qkv, ldqkv, ctx, ldctx, lse, causal, (hipStream_t)stream);
    return LPI_EINVAL;
}

extern "C" int lpi_attn_bwd(int dtype, int B, int L, int H, const void* qkv, int ldqkv, const void* ctx, int ldctx, const void* dctx,
                            int lddctx, const float* lse, float* delta, void* dqkv, int lddqkv, int causal, void* stream) {
    if (!qkv || !ctx || !dctx || !lse || !delta || !dqkv || bad_attn(dtype, B, L, H, ldqkv) || bad_attn(dtype, B, L, H, lddqkv)) return LPI_EINVAL;
    const int esz = dtype == LPI_F32 ? 4 : 2;
    if (ldctx < H * HD || lddctx < H * HD || (ldctx * esz) % 16 || (lddctx * esz) % 16) return LPI_EINVAL;
    if (((uintptr_t)qkv | (uintptr_t)ctx | (uintptr_t)dctx | (uintptr_t)dqkv) & 15) return LPI_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == LPI_F32)
        return causal ? bwd_launch<float, true>(B, L, H, qkv, ldqkv, ctx, ldctx, dctx, lddctx, lse, delta, dqkv, lddqkv, s)
                      : bwd_launch<float, false>(B, L, H, qkv, ldqkv, ctx, ldctx, dctx, lddctx, lse, delta, dqkv, lddqkv, s);
    if (dtype == LPI_BF16)
        return causal ? bwd_launch<bf16_t, true>(B, L, H, qkv, ldqkv, ctx, ldctx, dctx, lddctx, lse, delta, dqkv, lddqkv, s)
                      : bwd_launch<bf16_t, false>(B, L, H, qkv, ldqkv, ctx, ldctx, dctx, lddctx, lse, delta, dqkv, lddqkv, s);
    return LPI_EINVAL;
}
