// Attention with ONE query row per sample: the last block of a tower, where only the pooled token (CLS / EOT) is read by the
// heads (model.py:255, prompt_learner.py:61), so Q, the softmax row, the context row and their gradients exist for that row alone;
// K and V (and dK, dV) still cover every token of the sample.  Exact dead-row elimination of nn.MultiheadAttention
// (model.py:179-184), not an approximation.  HBM-bound: one pass over the sample's K and V (forward) / K, V, dK, dV (backward).
//
// One 4-wave workgroup per (sample, head), keys interleaved over the waves.  Scores: 16 lanes x 4 dims per key, 4 keys per wave
// instruction (each key row is one contiguous 128 B / 256 B read), 4-step xor reduction.  Softmax row in LDS; every wave reduces
// it for itself.  Context / dQ / dK / dV: lane = head dim, loop over keys, so every row access is one whole head row; the four
// partial context / dQ rows are summed through LDS in a fixed order.  No atomics: bitwise reproducible.
#include "common.h"

namespace {

constexpr int DH = 64;
constexpr int WPB = 4;   // waves per workgroup

__device__ __forceinline__ float reduce16(float v) {
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 8, 64);
    return v;
}

__device__ __forceinline__ float dot4(const f32x4& a, const f32x4& b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3]; }

template <typename T>
__device__ __forceinline__ void attn_pooled_fwd_body(
    int bh, float* sm, int B, int Lmax, const int* __restrict__ rs, int H, int Lp, const T* __restrict__ q, int ldq, const T* __restrict__ qkv, int ld, const int* __restrict__ idx,
    T* __restrict__ ctx, int ldo, float* __restrict__ lse, int causal, int pre = 0)
{
    // pre > 0 (shared prefix, attention.hip: attn_fwd_body): key position j < pre is global row j, position j >= pre the sample's own row j - pre
    float* s = sm;                 // scores
    float* p = sm + Lp;            // exp(score - max)
    float* red = sm + 2 * Lp;      // [WPB][64] partial context rows
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int b = bh / H, h = bh - b * H, d = H * DH;
    const int row = idx ? idx[b] : 0;
    // ragged batch (rs = row starts): sample b owns rows rs[b] .. rs[b+1]-1 of qkv
    const size_t r0 = rs ? (size_t)rs[b] : (size_t)b * Lmax;
    const int L = rs ? rs[b + 1] - rs[b] : Lmax;
    const int nv = causal ? row + 1 : L;           // keys the query row may attend to
    const int kk = lane >> 4, g = lane & 15;
    const f32x4 q4 = Elem<T>::ld4(q + (size_t)b * ldq + h * DH + 4 * g);
    const T* kbase = qkv + d + h * DH;
    const T* vbase = kbase + d;
    const long seg = (long)r0 - pre;
    auto krow = [&](int j) -> size_t { return (size_t)(j < pre ? (long)j : (long)j + seg); };
    for (int j0 = wave * 4; j0 < nv; j0 += 4 * WPB) {
        const int j = j0 + kk;
        float v = 0.f;
        if (j < nv) v = dot4(q4, Elem<T>::ld4(kbase + krow(j) * ld + 4 * g));
        v = reduce16(v);
        if (g == 0 && j < nv) s[j] = v * 0.125f;   // 1/sqrt(64)
    }
    __syncthreads();
    float m = -INFINITY;                           // every wave reduces the whole row itself: no cross-wave exchange
    for (int j = lane; j < nv; j += WAVE) m = fmaxf(m, s[j]);
    m = wave_max(m);
    float sum = 0.f;
    for (int j = lane; j < nv; j += WAVE) sum += __expf(s[j] - m);
    sum = wave_sum(sum);
    for (int j = threadIdx.x; j < nv; j += WAVE * WPB) p[j] = __expf(s[j] - m);
    __syncthreads();
    float acc = 0.f;
#pragma unroll 8
    for (int j = wave; j < nv; j += WPB) acc += p[j] * Elem<T>::ld(vbase + krow(j) * ld + lane);
    red[wave * DH + lane] = acc;
    __syncthreads();
    if (wave == 0) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < WPB; ++w) t += red[w * DH + lane];
        Elem<T>::st(ctx + (size_t)b * ldo + h * DH + lane, t / sum);
        if (lane == 0) lse[bh] = m + __logf(sum);
    }
}
template <typename T>
__global__ __launch_bounds__(WAVE * WPB) void attn_pooled_fwd_kernel(
    int B, int Lmax, const int* __restrict__ rs, int H, int Lp, const T* __restrict__ q, int ldq, const T* __restrict__ qkv, int ld, const int* __restrict__ idx,
    T* __restrict__ ctx, int ldo, float* __restrict__ lse, int causal, int pre)
{
    extern __shared__ float sm[];
    attn_pooled_fwd_body<T>(blockIdx.x, sm, B, Lmax, rs, H, Lp, q, ldq, qkv, ld, idx, ctx, ldo, lse, causal, pre);
}
// the two towers' pooled-row attention of the last block in ONE launch: workgroups [0, nb0) are problem 0's (sample, head) pairs
struct PoolFwdP { int B, Lmax, H, Lp, ldq, ld, ldo, causal; const int* rs; const void* q; const void* qkv; const int* idx; void* ctx; float* lse; int pre; };
template <typename T>
__global__ __launch_bounds__(WAVE * WPB) void attn_pooled_fwd_pair_kernel(PoolFwdP p0, PoolFwdP p1, int nb0)
{
    extern __shared__ float sm[];
    const bool z = (int)blockIdx.x >= nb0;
    const PoolFwdP& p = z ? p1 : p0;
    attn_pooled_fwd_body<T>(z ? blockIdx.x - nb0 : blockIdx.x, sm, p.B, p.Lmax, p.rs, p.H, p.Lp, (const T*)p.q, p.ldq, (const T*)p.qkv, p.ld, p.idx, (T*)p.ctx, p.ldo, p.lse,
                            p.causal, p.pre);
}

// TS: storage type of the SAVED q / qkv (fp16 after an f16-mode forward, else T); dctx, dq, dqkv are T.  All arithmetic is f32.
template <typename T, typename TS = T>
__device__ __forceinline__ void attn_pooled_bwd_body(
    int bh, float* sm, int B, int Lmax, const int* __restrict__ rs, int H, int Lp, const TS* __restrict__ q, int ldq, const TS* __restrict__ qkv, int ld, const int* __restrict__ idx,
    const T* __restrict__ dctx, int ldo, const float* __restrict__ lse, T* __restrict__ dq, int lddq, T* __restrict__ dqkv, int ldg, int causal, int pre = 0,
    float* __restrict__ part = nullptr)
{
    // pre > 0 (shared prefix): dK / dV of the shared keys go to part[b][key][dK (d) | dV (d)] as f32 (summed over the samples by lpi_shared_kv_reduce)
    float* p = sm;                 // softmax row
    float* dp = sm + Lp;           // dctx . V_j
    float* ds = sm + 2 * Lp;       // P_j (dP_j - delta) / 8
    float* red = sm + 3 * Lp;      // [WPB][64] partial dQ rows
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int b = bh / H, h = bh - b * H, d = H * DH;
    const int row = idx ? idx[b] : 0;
    const size_t r0 = rs ? (size_t)rs[b] : (size_t)b * Lmax;      // ragged batch: see the forward
    const int L = rs ? rs[b + 1] - rs[b] : Lmax;
    const int nv = causal ? row + 1 : L;
    const int kk = lane >> 4, g = lane & 15;
    const f32x4 q4 = Elem<TS>::ld4(q + (size_t)b * ldq + h * DH + 4 * g);
    const f32x4 o4 = Elem<T>::ld4(dctx + (size_t)b * ldo + h * DH + 4 * g);
    const TS* kbase = qkv + d + h * DH;
    const TS* vbase = kbase + d;
    const long seg = (long)r0 - pre;
    auto krow = [&](int j) -> size_t { return (size_t)(j < pre ? (long)j : (long)j + seg); };
    const float ls = lse[bh];
    for (int j0 = wave * 4; j0 < nv; j0 += 4 * WPB) {
        const int j = j0 + kk;
        float sc = 0.f, dv = 0.f;
        if (j < nv) {
            sc = dot4(q4, Elem<TS>::ld4(kbase + krow(j) * ld + 4 * g));
            dv = dot4(o4, Elem<TS>::ld4(vbase + krow(j) * ld + 4 * g));
        }
        sc = reduce16(sc);
        dv = reduce16(dv);
        if (g == 0 && j < nv) {
            p[j] = __expf(sc * 0.125f - ls);
            dp[j] = dv;
        }
    }
    __syncthreads();
    float delta = 0.f;
    for (int j = lane; j < nv; j += WAVE) delta += p[j] * dp[j];
    delta = wave_sum(delta);
    for (int j = threadIdx.x; j < nv; j += WAVE * WPB) ds[j] = p[j] * (dp[j] - delta) * 0.125f;
    __syncthreads();
    const float qd = Elem<TS>::ld(q + (size_t)b * ldq + h * DH + lane);
    const float od = Elem<T>::ld(dctx + (size_t)b * ldo + h * DH + lane);
    T* dk = dqkv + d + h * DH + lane;
    T* dv = dk + d;
    float* pk = part + ((size_t)b * pre) * 2 * d + h * DH + lane;      // used for j < pre only
    float acc = 0.f;
#pragma unroll 4
    for (int j = wave; j < nv; j += WPB) {
        const float dsj = ds[j];
        acc += dsj * Elem<TS>::ld(kbase + krow(j) * ld + lane);
        if (j < pre) {
            pk[(size_t)j * 2 * d] = dsj * qd;
            pk[(size_t)j * 2 * d + d] = p[j] * od;
        } else {
            Elem<T>::st(dk + krow(j) * ldg, dsj * qd);
            Elem<T>::st(dv + krow(j) * ldg, p[j] * od);
        }
    }
    for (int j = nv + wave; j < L + pre; j += WPB) {     // keys behind the causal mask get no gradient
        if (j < pre) {
            pk[(size_t)j * 2 * d] = 0.f;
            pk[(size_t)j * 2 * d + d] = 0.f;
        } else {
            Elem<T>::st(dk + krow(j) * ldg, 0.f);
            Elem<T>::st(dv + krow(j) * ldg, 0.f);
        }
    }
    red[wave * DH + lane] = acc;
    __syncthreads();
    if (wave == 0) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < WPB; ++w) t += red[w * DH + lane];
        Elem<T>::st(dq + (size_t)b * lddq + h * DH + lane, t);
    }
}
template <typename T, typename TS = T>
__global__ __launch_bounds__(WAVE * WPB) void attn_pooled_bwd_kernel(
    int B, int Lmax, const int* __restrict__ rs, int H, int Lp, const TS* __restrict__ q, int ldq, const TS* __restrict__ qkv, int ld, const int* __restrict__ idx,
    const T* __restrict__ dctx, int ldo, const float* __restrict__ lse, T* __restrict__ dq, int lddq, T* __restrict__ dqkv, int ldg, int causal, int pre,
    float* __restrict__ part)
{
    extern __shared__ float sm[];
    attn_pooled_bwd_body<T, TS>(blockIdx.x, sm, B, Lmax, rs, H, Lp, q, ldq, qkv, ld, idx, dctx, ldo, lse, dq, lddq, dqkv, ldg, causal, pre, part);
}
struct PoolBwdP { int B, Lmax, H, Lp, ldq, ld, ldo, lddq, ldg, causal; const int* rs; const void* q; const void* qkv; const int* idx; const void* dctx; const float* lse;
                  void* dq; void* dqkv; int pre; float* part; };
template <typename T, typename TS = T>
__global__ __launch_bounds__(WAVE * WPB) void attn_pooled_bwd_pair_kernel(PoolBwdP p0, PoolBwdP p1, int nb0)
{
    extern __shared__ float sm[];
    const bool z = (int)blockIdx.x >= nb0;
    const PoolBwdP& p = z ? p1 : p0;
    attn_pooled_bwd_body<T, TS>(z ? blockIdx.x - nb0 : blockIdx.x, sm, p.B, p.Lmax, p.rs, p.H, p.Lp, (const TS*)p.q, p.ldq, (const TS*)p.qkv, p.ld, p.idx, (const T*)p.dctx,
                                p.ldo, p.lse, (T*)p.dq, p.lddq, (T*)p.dqkv, p.ldg, p.causal, p.pre, p.part);
}

template <typename T>
__global__ __launch_bounds__(256) void scatter_add_rows_kernel(int B, int L, int d, const T* __restrict__ src, int lds_, const int* __restrict__ idx,
                                                                T* __restrict__ dst, int ldd)
{
    const int b = blockIdx.x;
    const int row = idx ? idx[b] : 0;
    T* o = dst + ((size_t)b * L + row) * ldd;
    const T* i = src + (size_t)b * lds_;
    for (int c = threadIdx.x * 4; c < d; c += 256 * 4) {
        f32x4 a = Elem<T>::ld4(o + c), v = Elem<T>::ld4(i + c);
        a[0] += v[0]; a[1] += v[1]; a[2] += v[2]; a[3] += v[3];
        Elem<T>::st4(o + c, a);
    }
}

}  // namespace

extern "C" int lpi_attn_pooled_fwd_varlen(int dtype, int B, int L, const int32_t* row_start, int H, const void* q, int ldq, const void* qkv, int ld,
                                          const int* idx, void* ctx, int ldo, float* lse, int causal, void* stream)
{
    const int* rs = row_start;
    if (!q || !qkv || !ctx || !lse || B <= 0 || L <= 0 || H <= 0) return LPI_EINVAL;
    const int esz = dtype == LPI_F32 ? 4 : 2;
    if (ld < 3 * H * DH || ldq < H * DH || ldo < H * DH || (ld * esz) % 16 || (ldq * esz) % 16) return LPI_EINVAL;
    if (((uintptr_t)q | (uintptr_t)qkv) & 15) return LPI_EINVAL;
    const int Lp = (L + 63) / 64 * 64;
    const size_t lds = (size_t)(2 * Lp + WPB * DH) * sizeof(float);
    if (lds > 64 * 1024) return LPI_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(B * H), block(WAVE * WPB);
    if (dtype == LPI_F32)
        LPI_LAUNCH((attn_pooled_fwd_kernel<float>), grid, block, lds, s, B, L, rs, H, Lp, (const float*)q, ldq, (const float*)qkv, ld, idx,
                   (float*)ctx, ldo, lse, causal, 0);
    else if (dtype == LPI_BF16)
        LPI_LAUNCH((attn_pooled_fwd_kernel<bf16_t>), grid, block, lds, s, B, L, rs, H, Lp, (const bf16_t*)q, ldq, (const bf16_t*)qkv, ld, idx,
                   (bf16_t*)ctx, ldo, lse, causal, 0);
    else if (dtype == LPI_F16)
        LPI_LAUNCH((attn_pooled_fwd_kernel<f16_t>), grid, block, lds, s, B, L, rs, H, Lp, (const f16_t*)q, ldq, (const f16_t*)qkv, ld, idx,
                   (f16_t*)ctx, ldo, lse, causal, 0);
    else
        return LPI_ENOSYS;
    LPI_CHECK_LAST();
    return 0;
}

extern "C" int lpi_attn_pooled_bwd_varlen(int dtype, int B, int L, const int32_t* row_start, int H, const void* q, int ldq, const void* qkv, int ld,
                                          const int* idx, const void* dctx, int ldo, const float* lse, void* dq, int lddq, void* dqkv, int ldg,
                                          int causal, void* stream)
{
    const int* rs = row_start;
    if (!q || !qkv || !dctx || !lse || !dq || !dqkv || B <= 0 || L <= 0 || H <= 0) return LPI_EINVAL;
    const int esz = dtype == LPI_F32 ? 4 : 2;
    if (ld < 3 * H * DH || ldg < 3 * H * DH || ldq < H * DH || ldo < H * DH || lddq < H * DH) return LPI_EINVAL;
    if ((ld * esz) % 16 || (ldq * esz) % 16 || (ldo * esz) % 16) return LPI_EINVAL;
    if (((uintptr_t)q | (uintptr_t)qkv | (uintptr_t)dctx) & 15) return LPI_EINVAL;
    const int Lp = (L + 63) / 64 * 64;
    const size_t lds = (size_t)(3 * Lp + WPB * DH) * sizeof(float);
    if (lds > 64 * 1024) return LPI_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(B * H), block(WAVE * WPB);
    if (dtype == LPI_F32)
        LPI_LAUNCH((attn_pooled_bwd_kernel<float>), grid, block, lds, s, B, L, rs, H, Lp, (const float*)q, ldq, (const float*)qkv, ld, idx,
                   (const float*)dctx, ldo, lse, (float*)dq, lddq, (float*)dqkv, ldg, causal, 0, nullptr);
    else if (dtype == LPI_BF16)
        LPI_LAUNCH((attn_pooled_bwd_kernel<bf16_t>), grid, block, lds, s, B, L, rs, H, Lp, (const bf16_t*)q, ldq, (const bf16_t*)qkv, ld, idx,
                   (const bf16_t*)dctx, ldo, lse, (bf16_t*)dq, lddq, (bf16_t*)dqkv, ldg, causal, 0, nullptr);
    else if (dtype == LPI_F16)      // saved q / qkv are fp16 (f16-mode forward); the gradients in and out are bf16
        LPI_LAUNCH((attn_pooled_bwd_kernel<bf16_t, f16_t>), grid, block, lds, s, B, L, rs, H, Lp, (const f16_t*)q, ldq, (const f16_t*)qkv, ld, idx,
                   (const bf16_t*)dctx, ldo, lse, (bf16_t*)dq, lddq, (bf16_t*)dqkv, ldg, causal, 0, nullptr);
    else
        return LPI_ENOSYS;
    LPI_CHECK_LAST();
    return 0;
}

// ---- the two towers' pooled attention in one launch (same kernels' bodies, bit for bit); desc[i]: the arguments of the _varlen entry points
extern "C" int lpi_attn_pooled_fwd_pair(int dtype, const lpi_attn_pooled_desc* d, void* stream)
{
    if (!d) return LPI_EINVAL;
    const int esz = dtype == LPI_F32 ? 4 : 2;
    PoolFwdP p[2];
    size_t lds = 0;
    for (int i = 0; i < 2; ++i) {
        const lpi_attn_pooled_desc& q = d[i];
        if (!q.q || !q.qkv || !q.ctx || !q.lse || q.B <= 0 || q.L <= 0 || q.H <= 0) return LPI_EINVAL;
        if (q.ldqkv < 3 * q.H * DH || q.ldq < q.H * DH || q.ldctx < q.H * DH || (q.ldqkv * esz) % 16 || (q.ldq * esz) % 16) return LPI_EINVAL;
        if (((uintptr_t)q.q | (uintptr_t)q.qkv) & 15) return LPI_EINVAL;
        const int Lp = (q.L + 63) / 64 * 64;
        lds = std::max(lds, (size_t)(2 * Lp + WPB * DH) * sizeof(float));
        if (q.shared_rows < 0 || (q.shared_rows > 0 && (!q.causal || !q.row_start || !q.idx || q.shared_rows >= q.L))) return LPI_EINVAL;
        p[i] = PoolFwdP{q.B, q.L, q.H, Lp, q.ldq, q.ldqkv, q.ldctx, q.causal, q.row_start, q.q, q.qkv, q.idx, q.ctx, q.lse, q.shared_rows};
    }
    if (lds > 64 * 1024) return LPI_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int nb0 = p[0].B * p[0].H;
    const dim3 grid(nb0 + p[1].B * p[1].H), block(WAVE * WPB);
    if (dtype == LPI_F32) LPI_LAUNCH((attn_pooled_fwd_pair_kernel<float>), grid, block, lds, s, p[0], p[1], nb0);
    else if (dtype == LPI_BF16) LPI_LAUNCH((attn_pooled_fwd_pair_kernel<bf16_t>), grid, block, lds, s, p[0], p[1], nb0);
    else if (dtype == LPI_F16) LPI_LAUNCH((attn_pooled_fwd_pair_kernel<f16_t>), grid, block, lds, s, p[0], p[1], nb0);
    else return LPI_ENOSYS;
    LPI_CHECK_LAST();
    return 0;
}
extern "C" int lpi_attn_pooled_bwd_pair(int dtype, const lpi_attn_pooled_desc* d, void* stream)
{
    if (!d) return LPI_EINVAL;
    const int esz = dtype == LPI_F32 ? 4 : 2;
    PoolBwdP p[2];
    size_t lds = 0;
    for (int i = 0; i < 2; ++i) {
        const lpi_attn_pooled_desc& q = d[i];
        if (!q.q || !q.qkv || !q.dctx || !q.lse || !q.dq || !q.dqkv || q.B <= 0 || q.L <= 0 || q.H <= 0) return LPI_EINVAL;
        if (q.ldqkv < 3 * q.H * DH || q.lddqkv < 3 * q.H * DH || q.ldq < q.H * DH || q.lddctx < q.H * DH || q.lddq < q.H * DH) return LPI_EINVAL;
        if ((q.ldqkv * esz) % 16 || (q.ldq * esz) % 16 || (q.lddctx * esz) % 16) return LPI_EINVAL;
        if (((uintptr_t)q.q | (uintptr_t)q.qkv | (uintptr_t)q.dctx) & 15) return LPI_EINVAL;
        const int Lp = (q.L + 63) / 64 * 64;
        lds = std::max(lds, (size_t)(3 * Lp + WPB * DH) * sizeof(float));
        if (q.shared_rows < 0 || (q.shared_rows > 0 && (!q.causal || !q.row_start || !q.idx || q.shared_rows >= q.L || !q.shared_dkv))) return LPI_EINVAL;
        p[i] = PoolBwdP{q.B, q.L, q.H, Lp, q.ldq, q.ldqkv, q.lddctx, q.lddq, q.lddqkv, q.causal, q.row_start, q.q, q.qkv, q.idx, q.dctx, q.lse, q.dq, q.dqkv, q.shared_rows,
                        q.shared_dkv};
    }
    if (lds > 64 * 1024) return LPI_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int nb0 = p[0].B * p[0].H;
    const dim3 grid(nb0 + p[1].B * p[1].H), block(WAVE * WPB);
    if (dtype == LPI_F32) LPI_LAUNCH((attn_pooled_bwd_pair_kernel<float>), grid, block, lds, s, p[0], p[1], nb0);
    else if (dtype == LPI_BF16) LPI_LAUNCH((attn_pooled_bwd_pair_kernel<bf16_t>), grid, block, lds, s, p[0], p[1], nb0);
    else if (dtype == LPI_F16) LPI_LAUNCH((attn_pooled_bwd_pair_kernel<bf16_t, f16_t>), grid, block, lds, s, p[0], p[1], nb0);
    else return LPI_ENOSYS;
    LPI_CHECK_LAST();
    for (int i = 0; i < 2; ++i)      // shared prefix: no query sits on the shared rows here, so their dK / dV are the samples' sum alone (written, not added)
        if (d[i].shared_rows > 0)
            if (int e = lpi_shared_kv_reduce(dtype, d[i].B, d[i].shared_rows, d[i].H, d[i].shared_dkv, d[i].dqkv, d[i].lddqkv, 0, stream)) return e;
    return 0;
}

// one problem through the descriptor form (the shared-prefix fields exist there only); a second, empty problem rides along
extern "C" int lpi_attn_pooled_fwd_desc(int dtype, const lpi_attn_pooled_desc* d, void* stream)
{
    if (!d) return LPI_EINVAL;
    if (!d->shared_rows)
        return lpi_attn_pooled_fwd_varlen(dtype, d->B, d->L, d->row_start, d->H, d->q, d->ldq, d->qkv, d->ldqkv, d->idx, d->ctx, d->ldctx, d->lse, d->causal, stream);
    if (!d->q || !d->qkv || !d->ctx || !d->lse || !d->row_start || !d->idx || !d->causal || d->B <= 0 || d->L <= 0 || d->H <= 0 || d->shared_rows < 0 || d->shared_rows >= d->L)
        return LPI_EINVAL;
    const int esz = dtype == LPI_F32 ? 4 : 2;
    if (d->ldqkv < 3 * d->H * DH || d->ldq < d->H * DH || d->ldctx < d->H * DH || (d->ldqkv * esz) % 16 || (d->ldq * esz) % 16) return LPI_EINVAL;
    if (((uintptr_t)d->q | (uintptr_t)d->qkv) & 15) return LPI_EINVAL;
    const int Lp = (d->L + 63) / 64 * 64;
    const size_t lds = (size_t)(2 * Lp + WPB * DH) * sizeof(float);
    if (lds > 64 * 1024) return LPI_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(d->B * d->H), block(WAVE * WPB);
    if (dtype == LPI_F32)
        LPI_LAUNCH((attn_pooled_fwd_kernel<float>), grid, block, lds, s, d->B, d->L, d->row_start, d->H, Lp, (const float*)d->q, d->ldq, (const float*)d->qkv, d->ldqkv, d->idx,
                   (float*)d->ctx, d->ldctx, d->lse, 1, d->shared_rows);
    else if (dtype == LPI_BF16)
        LPI_LAUNCH((attn_pooled_fwd_kernel<bf16_t>), grid, block, lds, s, d->B, d->L, d->row_start, d->H, Lp, (const bf16_t*)d->q, d->ldq, (const bf16_t*)d->qkv, d->ldqkv, d->idx,
                   (bf16_t*)d->ctx, d->ldctx, d->lse, 1, d->shared_rows);
    else if (dtype == LPI_F16)
        LPI_LAUNCH((attn_pooled_fwd_kernel<f16_t>), grid, block, lds, s, d->B, d->L, d->row_start, d->H, Lp, (const f16_t*)d->q, d->ldq, (const f16_t*)d->qkv, d->ldqkv, d->idx,
                   (f16_t*)d->ctx, d->ldctx, d->lse, 1, d->shared_rows);
    else
        return LPI_ENOSYS;
    LPI_CHECK_LAST();
    return 0;
}
extern "C" int lpi_attn_pooled_bwd_desc(int dtype, const lpi_attn_pooled_desc* d, void* stream)
{
    if (!d) return LPI_EINVAL;
    if (!d->shared_rows)
        return lpi_attn_pooled_bwd_varlen(dtype, d->B, d->L, d->row_start, d->H, d->q, d->ldq, d->qkv, d->ldqkv, d->idx, d->dctx, d->lddctx, d->lse, d->dq, d->lddq, d->dqkv,
                                          d->lddqkv, d->causal, stream);
    if (!d->q || !d->qkv || !d->dctx || !d->lse || !d->dq || !d->dqkv || !d->shared_dkv || !d->row_start || !d->idx || !d->causal || d->B <= 0 || d->L <= 0 || d->H <= 0 ||
        d->shared_rows < 0 || d->shared_rows >= d->L)
        return LPI_EINVAL;
    if (d->ldqkv < 3 * d->H * DH || d->lddqkv < 3 * d->H * DH || d->ldq < d->H * DH || d->lddctx < d->H * DH || d->lddq < d->H * DH) return LPI_EINVAL;
    const int esz = dtype == LPI_F32 ? 4 : 2;
    if ((d->ldqkv * esz) % 16 || (d->ldq * esz) % 16 || (d->lddctx * esz) % 16) return LPI_EINVAL;
    if (((uintptr_t)d->q | (uintptr_t)d->qkv | (uintptr_t)d->dctx) & 15) return LPI_EINVAL;
    const int Lp = (d->L + 63) / 64 * 64;
    const size_t lds = (size_t)(3 * Lp + WPB * DH) * sizeof(float);
    if (lds > 64 * 1024) return LPI_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(d->B * d->H), block(WAVE * WPB);
    if (dtype == LPI_F32)
        LPI_LAUNCH((attn_pooled_bwd_kernel<float>), grid, block, lds, s, d->B, d->L, d->row_start, d->H, Lp, (const float*)d->q, d->ldq, (const float*)d->qkv, d->ldqkv, d->idx,
                   (const float*)d->dctx, d->lddctx, d->lse, (float*)d->dq, d->lddq, (float*)d->dqkv, d->lddqkv, 1, d->shared_rows, d->shared_dkv);
    else if (dtype == LPI_BF16)
        LPI_LAUNCH((attn_pooled_bwd_kernel<bf16_t>), grid, block, lds, s, d->B, d->L, d->row_start, d->H, Lp, (const bf16_t*)d->q, d->ldq, (const bf16_t*)d->qkv, d->ldqkv, d->idx,
                   (const bf16_t*)d->dctx, d->lddctx, d->lse, (bf16_t*)d->dq, d->lddq, (bf16_t*)d->dqkv, d->lddqkv, 1, d->shared_rows, d->shared_dkv);
    else if (dtype == LPI_F16)
        LPI_LAUNCH((attn_pooled_bwd_kernel<bf16_t, f16_t>), grid, block, lds, s, d->B, d->L, d->row_start, d->H, Lp, (const f16_t*)d->q, d->ldq, (const f16_t*)d->qkv, d->ldqkv, d->idx,
                   (const bf16_t*)d->dctx, d->lddctx, d->lse, (bf16_t*)d->dq, d->lddq, (bf16_t*)d->dqkv, d->lddqkv, 1, d->shared_rows, d->shared_dkv);
    else
        return LPI_ENOSYS;
    LPI_CHECK_LAST();
    return lpi_shared_kv_reduce(dtype, d->B, d->shared_rows, d->H, d->shared_dkv, d->dqkv, d->lddqkv, 0, stream);
}

extern "C" int lpi_attn_pooled_fwd(int dtype, int B, int L, int H, const void* q, int ldq, const void* qkv, int ld, const int* idx,
                                   void* ctx, int ldo, float* lse, int causal, void* stream)
{
    return lpi_attn_pooled_fwd_varlen(dtype, B, L, nullptr, H, q, ldq, qkv, ld, idx, ctx, ldo, lse, causal, stream);
}
extern "C" int lpi_attn_pooled_bwd(int dtype, int B, int L, int H, const void* q, int ldq, const void* qkv, int ld, const int* idx,
                                   const void* dctx, int ldo, const float* lse, void* dq, int lddq, void* dqkv, int ldg, int causal,
                                   void* stream)
{
    return lpi_attn_pooled_bwd_varlen(dtype, B, L, nullptr, H, q, ldq, qkv, ld, idx, dctx, ldo, lse, dq, lddq, dqkv, ldg, causal, stream);
}

extern "C" int lpi_scatter_add_rows(int dtype, int B, int L, int d, const void* src, int ld_src, const int* idx, void* dst, int ld_dst,
                                    void* stream)
{
    if (!src || !dst || B <= 0 || L < 0 || (L == 0 && !idx) || d <= 0 || (d & 3) || ld_src < d || ld_dst < d) return LPI_EINVAL;      // L == 0: idx holds absolute rows
    const int esz = dtype == LPI_F32 ? 4 : 2;
    if ((ld_src * esz) % 8 || (ld_dst * esz) % 8 || (((uintptr_t)src | (uintptr_t)dst) & 7)) return LPI_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == LPI_F32)
        LPI_LAUNCH((scatter_add_rows_kernel<float>), dim3(B), dim3(256), 0, s, B, L, d, (const float*)src, ld_src, idx, (float*)dst, ld_dst);
    else if (dtype == LPI_BF16)
        LPI_LAUNCH((scatter_add_rows_kernel<bf16_t>), dim3(B), dim3(256), 0, s, B, L, d, (const bf16_t*)src, ld_src, idx, (bf16_t*)dst, ld_dst);
    else
        return LPI_ENOSYS;
    LPI_CHECK_LAST();
    return 0;
}
