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
// query column: the online softmax needs only two lane swaps (v_permlane16/32_swap, no LDS round trip), and the
// probability tile sitting in the accumulator registers is directly the B operand of the following P.V product.  The
// transposed operand of that product (V^T, K^T, dO^T, Q^T) is read straight from the row-major LDS image with
// ds_read_b64_tr_b16 (bf16) or 4-byte strided reads (f32): no transposed copy is ever staged.
// A wave works on NB = 2 independent 16-row blocks at once: they share every LDS operand read and give the scheduler
// two independent MFMA -> softmax -> MFMA chains to interleave (one chain alone is latency bound: ~1000 cycles per
// 32-key tile with ~250 cycles of issue).
// Backward recomputes P from the saved log-sum-exp and runs as two kernels so that every accumulator stays in
// one wave's registers (no atomics, bitwise reproducible):
//   pass A (a wave owns 2x16 queries): delta = rowsum(dO*O);  dQ = scale * dS K
//   pass B (a wave owns 2x16 keys)   : dV = P^T dO;           dK = scale * dS^T Q
// Fragment conventions (cdna_hip_programming.md section 3): 16x16 MFMA tiles, lane l supplies row (l & 15) and
// k-group g = l >> 4 of each operand as one 16-byte chunk; it receives column (l & 15), rows 4g..4g+3.
#include <type_traits>
#include "common.h"
#include "attn_softmax.h"

extern int g_lpi_tuning[16];

namespace {

constexpr int HD = 64;  // head_dim of every CLIP tower (width / 64 heads, model.py:292)
constexpr int NB = 2;   // 16-row blocks a wave processes together
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;
constexpr float SCALE = 0.125f;  // HD ** -0.5

template <typename T> struct AT;
template <> struct AT<float> {
    static constexpr int KS = 4;        // k-steps of 4 chunks per 64-element row
    static constexpr int RS = 272;      // LDS row stride in bytes (64 f32 + 16 B pad)
};
template <> struct AT<f16_t> {
    static constexpr int KS = 2;
    static constexpr int RS = 160;      // as bf16
};
template <> struct AT<bf16_t> {
    static constexpr int KS = 2;
    static constexpr int RS = 160;      // 64 bf16 + 32 B pad: conflict-free on the 64-bank LDS for both the ds_read_b128 row reads and the ds_read_b64_tr_b16 reads (144 B was 2-way on both: 43 % of LDS cycles were bank conflicts, profiles/r02_pmc.json)
};

// stage rows [0, L) x 64 elements of two matrices (global row strides ld0/ld1 elements) into two LDS images, zero rows
// [L, Lp).  8 independent 16-byte loads are kept in flight per thread before any LDS write: the loop is latency bound
// (the qkv buffer is far larger than the caches), and a one-load-at-a-time loop costs a full HBM round trip per iteration.
// CV0 / CV1: the matrix is stored as fp16 (saved by an f16-mode forward) and is converted to bf16 on its way into LDS (the backward's
// operands are bf16: gradients do not fit fp16's range)
// SW: the images have UNPADDED 128-byte rows (2-byte types) with the 16-byte chunk index XOR-ed with (row & 6) — conflict-free for the row-fragment and
// the transposing reads alike (tools/lds_swizzle_check.py; the layout of attention4.hip's images) — instead of rows padded to 160 bytes: a head of
// 257 .. 288 tokens (ViT-L/14: 273) then takes 72 KiB instead of 90, so that TWO workgroups fit a CU's 160 KiB like they do at L = 213
// pre / seg (shared-prefix ragged batches, see attn_fwd_body): image rows [0, pre) come from global rows [0, pre) of g0 / g1, image rows >= pre from global
// rows r + seg (g0 / g1 then point at global row 0); pre = 0, seg = 0: image row r = global row r of the sample's own base pointer
template <typename T, bool CV0 = false, bool CV1 = false, bool SW = false>
__device__ __forceinline__ void stage_rows2(char* lds0, const T* g0, char* lds1, const T* g1, int ld0, int ld1, int L, int Lp, int pre = 0, long seg = 0) {
    constexpr int NCH = HD * (int)sizeof(T) / 16;
    const int n = Lp * NCH, nt = blockDim.x;
    for (int base = threadIdx.x; base < n; base += 4 * nt) {
        uint4 v0[4], v1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = base + j * nt, row = i / NCH, c = i % NCH;
            v0[j] = make_uint4(0, 0, 0, 0);
            v1[j] = make_uint4(0, 0, 0, 0);
            if (i < n && row < L) {
                const size_t srow = (size_t)(row < pre ? (long)row : (long)row + seg);
                v0[j] = *reinterpret_cast<const uint4*>(g0 + srow * ld0 + c * Elem<T>::EPC);
                v1[j] = *reinterpret_cast<const uint4*>(g1 + srow * ld1 + c * Elem<T>::EPC);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = base + j * nt, row = i / NCH, c = i % NCH;
            if (i < n) {
                if constexpr (CV0) { Chunk t; t.u = v0[j]; chunk_f16_to_bf16(t); v0[j] = t.u; }
                if constexpr (CV1) { Chunk t; t.u = v1[j]; chunk_f16_to_bf16(t); v1[j] = t.u; }
                const int lo = SW ? row * 128 + ((c ^ (row & 6)) << 4) : row * AT<T>::RS + c * 16;
                *reinterpret_cast<uint4*>(lds0 + lo) = v0[j];
                *reinterpret_cast<uint4*>(lds1 + lo) = v1[j];
            }
        }
    }
}

// FOUR images in one sweep (the fused backward: K, V with La valid rows and the pre / seg row map, Q, dO with Lb valid rows): all of a thread's loads are
// in flight before the first LDS write — one HBM round trip where two stage_rows2 calls in a row take two (the text tower's backward is a chain of such
// round trips: 2 056 workgroups of a few dozen rows).  CVA: K, V, Q are fp16 and become bf16 on the way (f16 mode); dO never is.
template <typename T, bool CVA>
__device__ __forceinline__ void stage_rows4(char* lds_k, const T* gk, char* lds_v, const T* gv, int ldkv, int La, int pre, long seg,
                                            char* lds_q, const T* gq, int ldq, char* lds_do, const T* gdo, int lddo, int Lb, int Lp) {
    constexpr int NCH = HD * (int)sizeof(T) / 16;
    const int n = Lp * NCH, nt = blockDim.x;
    for (int base = threadIdx.x; base < n; base += 4 * nt) {
        uint4 vk[4], vv[4], vq[4], vd[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = base + j * nt, row = i / NCH, c = i % NCH;
            vk[j] = vv[j] = vq[j] = vd[j] = make_uint4(0, 0, 0, 0);
            if (i < n && row < La) {
                const size_t srow = (size_t)(row < pre ? (long)row : (long)row + seg);
                vk[j] = *reinterpret_cast<const uint4*>(gk + srow * ldkv + c * Elem<T>::EPC);
                vv[j] = *reinterpret_cast<const uint4*>(gv + srow * ldkv + c * Elem<T>::EPC);
            }
            if (i < n && row < Lb) {
                vq[j] = *reinterpret_cast<const uint4*>(gq + (size_t)row * ldq + c * Elem<T>::EPC);
                vd[j] = *reinterpret_cast<const uint4*>(gdo + (size_t)row * lddo + c * Elem<T>::EPC);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = base + j * nt, row = i / NCH, c = i % NCH;
            if (i < n) {
                if constexpr (CVA) {
                    Chunk t;
                    t.u = vk[j]; chunk_f16_to_bf16(t); vk[j] = t.u;
                    t.u = vv[j]; chunk_f16_to_bf16(t); vv[j] = t.u;
                    t.u = vq[j]; chunk_f16_to_bf16(t); vq[j] = t.u;
                }
                const int lo = row * AT<T>::RS + c * 16;
                *reinterpret_cast<uint4*>(lds_k + lo) = vk[j];
                *reinterpret_cast<uint4*>(lds_v + lo) = vv[j];
                *reinterpret_cast<uint4*>(lds_q + lo) = vq[j];
                *reinterpret_cast<uint4*>(lds_do + lo) = vd[j];
            }
        }
    }
}

// this lane's KS row chunks of LDS row `row` (rows >= L were zero filled by the staging)
template <typename T>
__device__ __forceinline__ void lds_row_chunks(Chunk (&q)[AT<T>::KS], const char* lds, int row, int grp) {
#pragma unroll
    for (int ks = 0; ks < AT<T>::KS; ++ks) q[ks].u = *reinterpret_cast<const uint4*>(lds + row * AT<T>::RS + (grp + 4 * ks) * 16);
}

// this lane's KS row chunks (chunk g + 4*ks) of row `row` of a global matrix; zeros if !valid
template <typename T, bool CV = false>
__device__ __forceinline__ void load_row_chunks(Chunk (&q)[AT<T>::KS], const T* g, size_t row, int ld, int grp, bool valid) {
#pragma unroll
    for (int ks = 0; ks < AT<T>::KS; ++ks) {
        q[ks].u = make_uint4(0, 0, 0, 0);
        if (valid) q[ks].u = *reinterpret_cast<const uint4*>(g + row * ld + (grp + 4 * ks) * Elem<T>::EPC);
        if constexpr (CV) chunk_f16_to_bf16(q[ks]);
    }
}

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }   // v_exp_f32; exp2(-inf) = 0

typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){a, b}, bf16x2));   // one v_cvt_pk_bf16_f32
}

// 16 LDS rows x (NB register operands)^T, the LDS chunk read once.  p = this lane's row pointer:
// lds + (r0 + (lane & 15)) * RS + (lane >> 4) * 16.  acc[j]: column = operand j's row (lane & 15), rows = LDS rows r0+4g+0..3.
template <typename T>
__device__ __forceinline__ void mma_lds_rows(f32x4 (&acc)[NB], const char* p, const Chunk (&b)[NB][AT<T>::KS]) {
#pragma unroll
    for (int j = 0; j < NB; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < AT<T>::KS; ++ks) {
        Chunk a;
        a.u = *reinterpret_cast<const uint4*>(p + ks * 64);
#pragma unroll
        for (int j = 0; j < NB; ++j) mma_chunk<T>(acc[j], a, b[j][ks]);
    }
}

// acc[j][dt] += X^T[d = 16dt + .., k] . P_j[k][col], k = 32 consecutive LDS rows of X; p0[j]/p1[j] = the two 16-row accumulator
// tiles (this lane: rows 4g..4g+3 of each) holding operand j.  The transposed X fragment is read once for all j.
// tp = this lane's transposed-read pointer:
//   bf16: lds + (r0 + 4g + ((lane & 15) >> 2)) * RS + (lane & 3) * 8    (ds_read_b64_tr_b16: lane i = 4q+p of a 16-lane group
//         addresses row q, columns 4p..4p+3 of a 4 x 16 block and receives column i of its 4 rows)
//   f32 : lds + (r0 + 4g) * RS + (lane & 15) * 4
template <typename T>
__device__ __forceinline__ void mma_transposed(f32x4 (&acc)[NB][4], const char* tp, const f32x4 (&p0)[NB], const f32x4 (&p1)[NB]) {
    if constexpr (sizeof(T) == 2) {
        Chunk b[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j)
            b[j].u = make_uint4(pack2_t<T>(p0[j][0], p0[j][1]), pack2_t<T>(p0[j][2], p0[j][3]), pack2_t<T>(p1[j][0], p1[j][1]), pack2_t<T>(p1[j][2], p1[j][3]));
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(tp + dt * 32));
            short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(tp + 16 * AT<T>::RS + dt * 32));
            Chunk a;
            const uint2 lo2 = __builtin_bit_cast(uint2, lo), hi2 = __builtin_bit_cast(uint2, hi);
            a.u = make_uint4(lo2.x, lo2.y, hi2.x, hi2.y);
#pragma unroll
            for (int j = 0; j < NB; ++j) mma_chunk<T>(acc[j][dt], a, b[j]);
        }
    } else {
        const float* base = reinterpret_cast<const float*>(tp);
        constexpr int RSF = AT<T>::RS / 4;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            Chunk a0, a1;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                a0.f[s] = base[s * RSF + dt * 16];
                a1.f[s] = base[(16 + s) * RSF + dt * 16];
            }
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                Chunk b0, b1;
                b0.f = p0[j];
                b1.f = p1[j];
                mma_chunk<T>(acc[j][dt], a0, b0);
                mma_chunk<T>(acc[j][dt], a1, b1);
            }
        }
    }
}

// the same two products on the swizzled images (SW): `base` = the image row block (rows r0 .. r0 + 15, r0 a multiple of 16), ro[ks] / to[dt] = this lane's
// byte offsets of its row chunk of k-step ks / of its transposing read of head-dim block dt (16 rows further: + 16 * 128, same swizzle)
template <typename T>
__device__ __forceinline__ void mma_lds_rows_sw(f32x4 (&acc)[NB], const char* base, const int (&ro)[AT<T>::KS], const Chunk (&b)[NB][AT<T>::KS]) {
#pragma unroll
    for (int j = 0; j < NB; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < AT<T>::KS; ++ks) {
        Chunk a;
        a.u = *reinterpret_cast<const uint4*>(base + ro[ks]);
#pragma unroll
        for (int j = 0; j < NB; ++j) mma_chunk<T>(acc[j], a, b[j][ks]);
    }
}
template <typename T>
__device__ __forceinline__ void mma_transposed_sw(f32x4 (&acc)[NB][4], const char* base, const int (&to)[4], const f32x4 (&p0)[NB], const f32x4 (&p1)[NB]) {
    Chunk b[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j)
        b[j].u = make_uint4(pack2_t<T>(p0[j][0], p0[j][1]), pack2_t<T>(p0[j][2], p0[j][3]), pack2_t<T>(p1[j][0], p1[j][1]), pack2_t<T>(p1[j][2], p1[j][3]));
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
        short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(base + to[dt]));
        short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(base + 16 * 128 + to[dt]));
        Chunk a;
        const uint2 lo2 = __builtin_bit_cast(uint2, lo), hi2 = __builtin_bit_cast(uint2, hi);
        a.u = make_uint4(lo2.x, lo2.y, hi2.x, hi2.y);
#pragma unroll
        for (int j = 0; j < NB; ++j) mma_chunk<T>(acc[j][dt], a, b[j]);
    }
}

template <typename T> __device__ __forceinline__ int row_ptr_off(int lane) { return (lane & 15) * AT<T>::RS + (lane >> 4) * 16; }
template <typename T> __device__ __forceinline__ int tr_ptr_off(int lane) {
    if constexpr (sizeof(T) == 2) return (4 * (lane >> 4) + ((lane & 15) >> 2)) * AT<T>::RS + (lane & 3) * 8;
    else return 4 * (lane >> 4) * AT<T>::RS + (lane & 15) * 4;
}

// reductions over the 4 lanes sharing (lane & 15): lane ^ 16 by v_permlane16_swap (odd rows of vdst <-> even rows of src),
// lane ^ 32 by v_permlane32_swap (upper half of vdst <-> lower half of src).  With vdst = src = v, one instruction leaves
// {own, partner} in its two results for every lane.
__device__ __forceinline__ float group_max(float v) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float group_sum(float v) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// ------------------------------------------------------------------------------------------------ forward
// SHARED PREFIX (round 5; causal, ragged batches only: pre > 0).  In training the text tower's first `pre` positions of every caption — SOT and the context
// slots — are the same rows for all samples (the prompts are broadcast, slinet.py:119-130) and, under the causal mask (model.py:347-353), stay the same
// through every block.  They are then stored ONCE, as global rows [0, pre); sample b owns rows rs[b] .. rs[b+1]-1 = its positions pre, pre + 1, ...; its keys
// and values are [shared rows | own rows], its queries the own rows (position = pre + own index).  Workgroups of sample index `bshared` (= B, one past the
// tail samples) compute the shared sequence itself: rows [0, pre), plain causal attention among themselves.
template <typename T, bool CAUSAL, bool SW = false>
__device__ __forceinline__ void attn_fwd_body(int bh, char* smem, int Lmax, int Lpmax, const int* __restrict__ rs, int H, const T* __restrict__ qkv, int ldqkv,
                                              T* __restrict__ ctx, int ldctx, float* __restrict__ lse, int pre = 0, int bshared = -1,
                                              int qkv_hs = 0, int qkv_vs = 0, int ctx_hs = 0) {
    // LAYOUT of q / k / v and ctx (round 6; include/lpi_hip.h, lpi_attn_fwd_desc): element (row, head h, which in {q, k, v}, c) of qkv sits at
    // row ldqkv + h hs + which vs + c.  0 = the interleaved default [B L, 3 H 64]: hs = 64, vs = H 64.  Head-BLOCKED planes [3 H][rows][64]: ldqkv = 64,
    // hs = rows 64, vs = H rows 64 — a (sample, head) slice is then one contiguous run of L x 128 bytes instead of L pieces at a stride of 6 H 64 bytes.
    const int hs = qkv_hs ? qkv_hs : HD, chs = ctx_hs ? ctx_hs : HD;
    const int b = bh / H, h = bh % H;
    // ragged batch (rs = row starts, B + 1 ints): sample b owns rows rs[b] .. rs[b+1]-1; else every sample has Lmax rows
    int L = Lmax, Lp = Lpmax;         // L: the sample's OWN rows (its queries); Lk below: its keys
    size_t row0 = (size_t)b * Lmax;
    int pb = 0;                       // shared positions in front of this sample's own rows
    if (CAUSAL && pre > 0 && b == bshared) {
        row0 = 0;
        L = pre;
        Lp = (L + 31) / 32 * 32;
    } else if (rs) {
        const int r = rs[b];
        row0 = (size_t)r;
        L = rs[b + 1] - r;
        if (CAUSAL && pre > 0) pb = pre;
        Lp = (pb + L + 31) / 32 * 32;
    }
    const int Lk = pb + L;
    const size_t lse0 = ((size_t)b * H + h) * Lmax;       // lse / delta stay [B, H, Lmax]
    const int dm = qkv_vs ? qkv_vs : H * HD;
    const T* qg = qkv + row0 * ldqkv + (size_t)h * hs;
    constexpr int RSX = SW ? 128 : AT<T>::RS;      // image row stride (SW: see stage_rows2)
    static_assert(!SW || sizeof(T) == 2, "the swizzled images are for the 2-byte operand types");
    char* k_lds = smem;
    char* v_lds = smem + Lp * RSX;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int g = lane >> 4;
    // the wave's first query blocks are fetched before the staging so that their HBM latency hides under it
    Chunk q[NB][AT<T>::KS];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int qr = (wave * NB + j) * 16 + (lane & 15);
        load_row_chunks<T>(q[j], qg, qr, ldqkv, g, qr < L);
    }
    if (pb) stage_rows2<T, false, false, SW>(k_lds, qkv + (size_t)h * hs + dm, v_lds, qkv + (size_t)h * hs + 2 * (size_t)dm, ldqkv, ldqkv, Lk, Lp, pb, (long)row0 - pb);
    else stage_rows2<T, false, false, SW>(k_lds, qg + dm, v_lds, qg + 2 * (size_t)dm, ldqkv, ldqkv, L, Lp);
    __syncthreads();

    const float c = SCALE * LOG2E;
    const char* const kp0 = k_lds + row_ptr_off<T>(lane);
    const char* const vt0 = v_lds + tr_ptr_off<T>(lane);
    int ro[AT<T>::KS], to[4];      // SW: this lane's swizzled offsets inside a 16-row block of an image
    {
        const int r = lane & 15, rr = 4 * g + (r >> 2), pq = lane & 3;
#pragma unroll
        for (int ks = 0; ks < AT<T>::KS; ++ks) ro[ks] = r * 128 + (((g + 4 * ks) ^ (r & 6)) << 4);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) to[dt] = rr * 128 + (((2 * dt + (pq >> 1)) ^ (rr & 6)) << 4) + (pq & 1) * 8;
    }
    for (int q0 = wave * 16 * NB; q0 < L; q0 += nw * 16 * NB) {
        int qrow[NB], qpos[NB];       // own row index (loads / stores), position in the sequence (masks)
#pragma unroll
        for (int j = 0; j < NB; ++j) { qrow[j] = q0 + 16 * j + (lane & 15); qpos[j] = qrow[j] + pb; }
        if (q0 != wave * 16 * NB) {
#pragma unroll
            for (int j = 0; j < NB; ++j) load_row_chunks<T>(q[j], qg, qrow[j], ldqkv, g, qrow[j] < L);
        }
        float m[NB], lsum[NB];    // m in the scaled log2 domain
        f32x4 o[NB][4];
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            m[j] = -INFINITY;
            lsum[j] = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) o[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        // one 32-key tile; MASKED only for tiles that can contain padded keys (last tile) or the causal diagonal
        auto tile = [&](int kb, auto masked_tag) {
            constexpr bool MASKED = decltype(masked_tag)::value;
            f32x4 s0[NB], s1[NB];
            if constexpr (SW) {
                const char* kb_ = k_lds + kb * 128;
                mma_lds_rows_sw<T>(s0, kb_, ro, q);
                mma_lds_rows_sw<T>(s1, kb_ + 16 * 128, ro, q);
                attn_softmax_tile<NB, MASKED, CAUSAL>(s0, s1, m, lsum, o, kb, g, Lk, qpos, c);
                mma_transposed_sw<T>(o, v_lds + kb * 128, to, s0, s1);
            } else {
                const char* kp = kp0 + kb * AT<T>::RS;
                mma_lds_rows<T>(s0, kp, q);
                mma_lds_rows<T>(s1, kp + 16 * AT<T>::RS, q);
                attn_softmax_tile<NB, MASKED, CAUSAL>(s0, s1, m, lsum, o, kb, g, Lk, qpos, c);
                mma_transposed<T>(o, vt0 + kb * AT<T>::RS, s0, s1);
            }
        };
        const int qlast = q0 + pb + 16 * NB - 1;      // positions
        const int kend = CAUSAL ? min(Lp, (qlast / 32 + 1) * 32) : Lp;
        // unmasked tiles: every key < Lk and (causal) every key <= the smallest query position of the blocks
        const int kfull = CAUSAL ? min((Lk / 32) * 32, ((q0 + pb) / 32) * 32) : (Lk / 32) * 32;
        int kb = 0;
        for (; kb < kfull; kb += 32) tile(kb, std::false_type{});
        for (; kb < kend; kb += 32) tile(kb, std::true_type{});
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const float ltot = group_sum(lsum[j]);
            const float inv = 1.0f / ltot;
            if constexpr (sizeof(T) == 2) {
                f32x4 os[4];
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) os[dt] = o[j][dt] * inv;
                store_row16_t<T>(ctx + (row0 + qrow[j]) * ldctx + (size_t)h * chs, os, g, qrow[j] < L);
                if (qrow[j] < L && g == 0) lse[lse0 + qrow[j]] = (m[j] + log2f(ltot)) * LN2;
            } else if (qrow[j] < L) {
                T* dst = ctx + (row0 + qrow[j]) * ldctx + (size_t)h * chs + 4 * g;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) Elem<T>::st4(dst + dt * 16, o[j][dt] * inv);
                if (g == 0) lse[lse0 + qrow[j]] = (m[j] + log2f(ltot)) * LN2;
            }
        }
    }
}

template <typename T, bool CAUSAL, bool SW = false>
__global__ __launch_bounds__(512) void attn_fwd_kernel(int Lmax, int Lpmax, const int* __restrict__ rs, int H, const T* __restrict__ qkv, int ldqkv,
                                                      T* __restrict__ ctx, int ldctx, float* __restrict__ lse, int pre, int bshared, int hs, int vs, int chs) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    attn_fwd_body<T, CAUSAL, SW>(blockIdx.x, smem, Lmax, Lpmax, rs, H, qkv, ldqkv, ctx, ldctx, lse, pre, bshared, hs, vs, chs);
}
// TWO attention forwards in one launch (the vision tower's and the text tower's of the same layer): workgroups [0, nb0) run problem 0, the
// rest problem 1.  The text tower's forward alone is a 15 us kernel — a chain of dependent HBM round trips with the chip nearly idle; here its
// workgroups fill in behind the vision tower's.  The launch takes the larger thread count and LDS size of the two.
template <typename T>
struct AttnFwdP {
    int L, Lp, H, ldqkv, ldctx;
    const int* rs; const T* qkv; T* ctx; float* lse;
    int pre, bshared;      // shared prefix (attn_fwd_body): 0, -1 = none
    int hs, vs, chs;       // layout strides (attn_fwd_body): 0 = the interleaved default
};
template <typename T, bool C0, bool C1, bool SW0 = false>
__global__ __launch_bounds__(512) void attn_fwd_pair_kernel(AttnFwdP<T> p0, AttnFwdP<T> p1, int nb0) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if ((int)blockIdx.x < nb0) attn_fwd_body<T, C0, SW0>(blockIdx.x, smem, p0.L, p0.Lp, p0.rs, p0.H, p0.qkv, p0.ldqkv, p0.ctx, p0.ldctx, p0.lse, p0.pre, p0.bshared, p0.hs, p0.vs, p0.chs);
    else attn_fwd_body<T, C1>(blockIdx.x - nb0, smem, p1.L, p1.Lp, p1.rs, p1.H, p1.qkv, p1.ldqkv, p1.ctx, p1.ldctx, p1.lse, p1.pre, p1.bshared, p1.hs, p1.vs, p1.chs);
}

// ------------------------------------------------------------------------------------------------ backward A
template <typename T, bool CAUSAL, bool SV16 = false>
__global__ __launch_bounds__(512) void attn_bwd_dq_kernel(int Lmax, int Lpmax, const int* __restrict__ rs, int H, const T* __restrict__ qkv, int ldqkv,
                                                         const T* __restrict__ ctx, int ldctx, const T* __restrict__ dctx, int lddctx,
                                                         const float* __restrict__ lse, float* __restrict__ delta,
                                                         T* __restrict__ dqkv, int lddqkv, int pre, int bshared, int q_hs, int q_vs, int dq_hs, int dq_vs) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    // ragged batch (rs = row starts, B + 1 ints): sample b owns rows rs[b] .. rs[b+1]-1; else every sample has Lmax rows
    // shared prefix (pre > 0): as in attn_bwd_fused_kernel — keys [shared rows | own rows], queries the own rows, masks on positions
    int L = Lmax, Lp = Lpmax;
    size_t row0 = (size_t)b * Lmax;
    int pb = 0;
    if (CAUSAL && pre > 0 && b == bshared) {
        row0 = 0;
        L = pre;
        Lp = (L + 31) / 32 * 32;
    } else if (rs) {
        const int r = rs[b];
        row0 = (size_t)r;
        L = rs[b + 1] - r;
        if (CAUSAL && pre > 0) pb = pre;
        Lp = (pb + L + 31) / 32 * 32;
    }
    const int Lk = pb + L;
    const size_t lse0 = ((size_t)b * H + h) * Lmax;       // lse / delta stay [B (+ 1), H, Lmax], indexed by the own row
    // layout strides of qkv / dqkv (attn_fwd_body has the definition; 0 = the interleaved default).  dm stays H * HD: the shared-prefix partials' own layout
    const int hsq = q_hs ? q_hs : HD, vsq = q_vs ? q_vs : H * HD, hsd = dq_hs ? dq_hs : HD, vsd = dq_vs ? dq_vs : H * HD;
    const T* qg = qkv + row0 * ldqkv + (size_t)h * hsq;
    char* k_lds = smem;
    char* v_lds = smem + Lp * AT<T>::RS;
    if (pb) stage_rows2<T, SV16, SV16>(k_lds, qkv + (size_t)h * hsq + vsq, v_lds, qkv + (size_t)h * hsq + 2 * (size_t)vsq, ldqkv, ldqkv, Lk, Lp, pb, (long)row0 - pb);
    else stage_rows2<T, SV16, SV16>(k_lds, qg + vsq, v_lds, qg + 2 * (size_t)vsq, ldqkv, ldqkv, L, Lp);
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int g = lane >> 4;
    const float c = SCALE * LOG2E;
    const char* const kp0 = k_lds + row_ptr_off<T>(lane);
    const char* const vp0 = v_lds + row_ptr_off<T>(lane);
    const char* const kt0 = k_lds + tr_ptr_off<T>(lane);
    for (int q0 = wave * 16 * NB; q0 < L; q0 += nw * 16 * NB) {
        int qrow[NB];
        Chunk q[NB][AT<T>::KS], dO[NB][AT<T>::KS];
        float dls[NB], lq[NB];
        f32x4 dq[NB][4];
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            qrow[j] = q0 + 16 * j + (lane & 15);
            const bool valid = qrow[j] < L;
            const size_t grow = row0 + qrow[j];
            load_row_chunks<T, SV16>(q[j], qg, qrow[j], ldqkv, g, valid);
            load_row_chunks<T>(dO[j], dctx + h * HD, grow, lddctx, g, valid);
            Chunk oc[AT<T>::KS];
            load_row_chunks<T>(oc, ctx + h * HD, grow, ldctx, g, valid);
            float dl = 0.f;
#pragma unroll
            for (int ks = 0; ks < AT<T>::KS; ++ks) {
                if constexpr (sizeof(T) == 4) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) dl += oc[ks].f[e] * dO[j][ks].f[e];
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) dl += (SV16 ? (float)oc[ks].hh[e] : (float)oc[ks].h[e]) * (float)dO[j][ks].h[e];
                }
            }
            dl = group_sum(dl);
            lq[j] = valid ? lse[lse0 + qrow[j]] * LOG2E : INFINITY;     // padded queries -> P = 0
            if (valid && g == 0) delta[lse0 + qrow[j]] = dl;
            dls[j] = dl * SCALE;
#pragma unroll
            for (int i = 0; i < 4; ++i) dq[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        auto tile = [&](int kb, auto masked_tag) {
            constexpr bool MASKED = decltype(masked_tag)::value;
            const char* kp = kp0 + kb * AT<T>::RS;
            const char* vp = vp0 + kb * AT<T>::RS;
            f32x4 s0[NB], s1[NB], p0[NB], p1[NB];
            mma_lds_rows<T>(s0, kp, q);
            mma_lds_rows<T>(s1, kp + 16 * AT<T>::RS, q);
            mma_lds_rows<T>(p0, vp, dO);
            mma_lds_rows<T>(p1, vp + 16 * AT<T>::RS, dO);
#pragma unroll
            for (int j = 0; j < NB; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float e0 = fast_exp2(fmaf(s0[j][r], c, -lq[j]));
                    float e1 = fast_exp2(fmaf(s1[j][r], c, -lq[j]));
                    if constexpr (MASKED) {
                        const int k0 = kb + 4 * g + r, k1 = k0 + 16;
                        if (!(k0 < Lk && (!CAUSAL || k0 <= qrow[j] + pb))) e0 = 0.f;
                        if (!(k1 < Lk && (!CAUSAL || k1 <= qrow[j] + pb))) e1 = 0.f;
                    }
                    s0[j][r] = e0 * fmaf(p0[j][r], SCALE, -dls[j]);       // P * (dP - delta) * scale
                    s1[j][r] = e1 * fmaf(p1[j][r], SCALE, -dls[j]);
                }
            mma_transposed<T>(dq, kt0 + kb * AT<T>::RS, s0, s1);
        };
        const int qlast = q0 + pb + 16 * NB - 1;      // positions
        const int kend = CAUSAL ? min(Lp, (qlast / 32 + 1) * 32) : Lp;
        const int kfull = CAUSAL ? min((Lk / 32) * 32, ((q0 + pb) / 32) * 32) : (Lk / 32) * 32;
        int kb = 0;
        for (; kb < kfull; kb += 32) tile(kb, std::false_type{});
        for (; kb < kend; kb += 32) tile(kb, std::true_type{});
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            if constexpr (sizeof(T) == 2) {
                store_row_bf16_t(reinterpret_cast<bf16_t*>(dqkv) + (row0 + qrow[j]) * lddqkv + (size_t)h * hsd, dq[j], g, qrow[j] < L);
            } else if (qrow[j] < L) {
                T* dst = dqkv + (row0 + qrow[j]) * lddqkv + (size_t)h * hsd + 4 * g;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) Elem<T>::st4(dst + dt * 16, dq[j][dt]);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ backward B
template <typename T, bool CAUSAL, bool SV16 = false>
__global__ __launch_bounds__(512) void attn_bwd_dkv_kernel(int Lmax, int Lpmax, const int* __restrict__ rs, int H, const T* __restrict__ qkv, int ldqkv,
                                                          const T* __restrict__ dctx, int lddctx, const float* __restrict__ lse,
                                                          const float* __restrict__ delta, T* __restrict__ dqkv, int lddqkv, int pre, int bshared,
                                                          float* __restrict__ shared_dkv, int q_hs, int q_vs, int dq_hs, int dq_vs) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    // ragged batch (rs = row starts, B + 1 ints): sample b owns rows rs[b] .. rs[b+1]-1; else every sample has Lmax rows
    // shared prefix (pre > 0): as in attn_bwd_fused_kernel's phase B — a tail sample's dK / dV of the shared keys leave as f32 partials
    int L = Lmax, Lp = Lpmax;
    size_t row0 = (size_t)b * Lmax;
    int pb = 0;
    if (CAUSAL && pre > 0 && b == bshared) {
        row0 = 0;
        L = pre;
        Lp = (L + 31) / 32 * 32;
    } else if (rs) {
        const int r = rs[b];
        row0 = (size_t)r;
        L = rs[b + 1] - r;
        if (CAUSAL && pre > 0) pb = pre;
        Lp = (pb + L + 31) / 32 * 32;
    }
    const int Lk = pb + L;
    const size_t lse0 = ((size_t)b * H + h) * Lmax;       // lse / delta stay [B (+ 1), H, Lmax], indexed by the own row
    const int dm = H * HD;
    // layout strides of qkv / dqkv (attn_fwd_body has the definition; 0 = the interleaved default).  dm stays H * HD: the shared-prefix partials' own layout
    const int hsq = q_hs ? q_hs : HD, vsq = q_vs ? q_vs : H * HD, hsd = dq_hs ? dq_hs : HD, vsd = dq_vs ? dq_vs : H * HD;
    const T* qg = qkv + row0 * ldqkv + (size_t)h * hsq;
    const T* kg = qkv + (size_t)h * hsq + vsq;                      // keys / values by GLOBAL row (key position p -> row p, or row0 + p - pb behind the shared ones)
    auto krow_global = [&](int p) -> size_t { return (size_t)(p < pb ? (long)p : (long)row0 + p - pb); };
    char* q_lds = smem;
    char* do_lds = smem + Lp * AT<T>::RS;
    float* lse_lds = reinterpret_cast<float*>(smem + 2 * Lp * AT<T>::RS);
    float* dl_lds = lse_lds + Lp;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int g = lane >> 4;
    Chunk kk[NB][AT<T>::KS], vv[NB][AT<T>::KS];
#pragma unroll
    for (int j = 0; j < NB; ++j) {          // first key blocks: fetched ahead of the staging
        const int kr = (wave * NB + j) * 16 + (lane & 15);
        load_row_chunks<T, SV16>(kk[j], kg, krow_global(kr), ldqkv, g, kr < Lk);
        load_row_chunks<T, SV16>(vv[j], kg + vsq, krow_global(kr), ldqkv, g, kr < Lk);
    }
    stage_rows2<T, SV16, false>(q_lds, qg, do_lds, dctx + row0 * lddctx + h * HD, ldqkv, lddctx, L, Lp);
    for (int i = threadIdx.x; i < Lp; i += blockDim.x) {
        lse_lds[i] = i < L ? lse[lse0 + i] * LOG2E : INFINITY;  // padded queries -> P = 0
        dl_lds[i] = i < L ? delta[lse0 + i] * SCALE : 0.f;
    }
    __syncthreads();

    const float c = SCALE * LOG2E;
    const char* const qp0 = q_lds + row_ptr_off<T>(lane);
    const char* const dp0 = do_lds + row_ptr_off<T>(lane);
    const char* const qt0 = q_lds + tr_ptr_off<T>(lane);
    const char* const dt0 = do_lds + tr_ptr_off<T>(lane);
    for (int k0 = wave * 16 * NB; k0 < Lk; k0 += nw * 16 * NB) {
        int krow[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) krow[j] = k0 + 16 * j + (lane & 15);
        if (k0 != wave * 16 * NB) {
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                load_row_chunks<T, SV16>(kk[j], kg, krow_global(krow[j]), ldqkv, g, krow[j] < Lk);
                load_row_chunks<T, SV16>(vv[j], kg + vsq, krow_global(krow[j]), ldqkv, g, krow[j] < Lk);
            }
        }
        f32x4 dk[NB][4], dv[NB][4];
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) { dk[j][i] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[j][i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        // padded queries need no mask: their lse is +inf in LDS, so P = exp2(-inf) = 0; padded KEY rows are never stored.
        auto tile = [&](int qb, auto masked_tag) {
            constexpr bool MASKED = decltype(masked_tag)::value;       // only the causal diagonal tiles
            const char* qp = qp0 + qb * AT<T>::RS;
            const char* dp = dp0 + qb * AT<T>::RS;
            // S[q][key]: rows q = qb + 16t + 4g + r, column key = krow
            f32x4 s0[NB], s1[NB], p0[NB], p1[NB], e0[NB], e1[NB];
            mma_lds_rows<T>(s0, qp, kk);
            mma_lds_rows<T>(s1, qp + 16 * AT<T>::RS, kk);
            mma_lds_rows<T>(p0, dp, vv);
            mma_lds_rows<T>(p1, dp + 16 * AT<T>::RS, vv);
            const f32x4 l0 = *reinterpret_cast<const f32x4*>(lse_lds + qb + 4 * g);
            const f32x4 l1 = *reinterpret_cast<const f32x4*>(lse_lds + qb + 16 + 4 * g);
            const f32x4 d0 = *reinterpret_cast<const f32x4*>(dl_lds + qb + 4 * g);     // delta * scale
            const f32x4 d1 = *reinterpret_cast<const f32x4*>(dl_lds + qb + 16 + 4 * g);
#pragma unroll
            for (int j = 0; j < NB; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    e0[j][r] = fast_exp2(fmaf(s0[j][r], c, -l0[r]));
                    e1[j][r] = fast_exp2(fmaf(s1[j][r], c, -l1[r]));
                    if constexpr (MASKED) {
                        const int qa = qb + pb + 4 * g + r, qc = qa + 16;      // query positions
                        if (krow[j] > qa) e0[j][r] = 0.f;
                        if (krow[j] > qc) e1[j][r] = 0.f;
                    }
                    s0[j][r] = e0[j][r] * fmaf(p0[j][r], SCALE, -d0[r]);
                    s1[j][r] = e1[j][r] * fmaf(p1[j][r], SCALE, -d1[r]);
                }
            mma_transposed<T>(dv, dt0 + qb * AT<T>::RS, e0, e1);
            mma_transposed<T>(dk, qt0 + qb * AT<T>::RS, s0, s1);
        };
        // query tiles by OWN index; a key at position k is seen by the own queries i with i + pb >= k
        const int kq = k0 - pb;
        int qb = CAUSAL ? (kq > 0 ? (kq / 32) * 32 : 0) : 0;
        if constexpr (CAUSAL) {
            const int qdiag = (kq + 16 * NB - 1 >= 0) ? min(Lp, ((kq + 16 * NB - 1) / 32 + 1) * 32) : 0;      // tiles that can hold q < key
            for (; qb < qdiag; qb += 32) tile(qb, std::true_type{});
        }
        for (; qb < Lp; qb += 32) tile(qb, std::false_type{});
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            if (pb && krow[j] < pb) {      // a SHARED key: this sample's f32 partial (summed over the samples by lpi_shared_kv_reduce)
                float* dst = shared_dkv + ((size_t)b * pb + krow[j]) * 2 * dm + h * HD + 4 * g;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    *reinterpret_cast<f32x4*>(dst + dt * 16) = dk[j][dt];
                    *reinterpret_cast<f32x4*>(dst + dm + dt * 16) = dv[j][dt];
                }
                continue;
            }
            const size_t orow = row0 + (krow[j] - pb);      // own key
            if constexpr (sizeof(T) == 2) {
                bf16_t* dst = reinterpret_cast<bf16_t*>(dqkv) + orow * lddqkv + (size_t)h * hsd;
                store_row_bf16_t(dst + vsd, dk[j], g, krow[j] < Lk);
                store_row_bf16_t(dst + 2 * (size_t)vsd, dv[j], g, krow[j] < Lk);
            } else if (krow[j] < Lk) {
                T* dst = dqkv + orow * lddqkv + (size_t)h * hsd + 4 * g;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    Elem<T>::st4(dst + vsd + dt * 16, dk[j][dt]);
                    Elem<T>::st4(dst + 2 * (size_t)vsd + dt * 16, dv[j][dt]);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ backward, fused
// One kernel for both passes when Q, K, V and dO of a head all fit in LDS (bf16: 4 x Lp x 144 B = 129 KB at L = 213): the head's
// qkv and dctx rows are read from HBM once instead of twice — the two-pass backward is HBM bound (about 1 GB per vision layer),
// this one moves a third less.  Phase A (queries) produces delta into LDS and dQ; after one barrier phase B (keys) produces
// dK, dV.  Same tile bodies, same fixed summation order as the two-pass kernels.
template <typename T, bool CAUSAL, bool SV16 = false>
__global__ __launch_bounds__(512) void attn_bwd_fused_kernel(int Lmax, int Lpmax, const int* __restrict__ rs, int H, const T* __restrict__ qkv, int ldqkv,
                                                            const T* __restrict__ ctx, int ldctx, const T* __restrict__ dctx, int lddctx,
                                                            const float* __restrict__ lse, float* __restrict__ delta,
                                                            T* __restrict__ dqkv, int lddqkv, int rows_hi, int pre, int bshared, float* __restrict__ shared_dkv,
                                                            int q_hs, int q_vs, int dq_hs, int dq_vs) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    // ragged batch (rs = row starts, B + 1 ints): sample b owns rows rs[b] .. rs[b+1]-1; else every sample has Lmax rows
    // shared prefix (pre > 0, causal ragged batches; see attn_fwd_body): the sample's keys are [shared rows | own rows], its queries the own rows;
    // dK / dV of the SHARED keys go, as f32 partials, to shared_dkv[b][key][dK (H 64) | dV (H 64)] — lpi_shared_kv_reduce adds them over the samples
    // onto the shared rows of dqkv, which the workgroups of sample `bshared` (the shared sequence itself) write
    int L = Lmax, Lp = Lpmax;         // L: the sample's OWN rows (queries); Lk: its keys
    size_t row0 = (size_t)b * Lmax;
    int pb = 0;
    if (CAUSAL && pre > 0 && b == bshared) {
        row0 = 0;
        L = pre;
        Lp = (L + 31) / 32 * 32;
    } else if (rs) {
        const int r = rs[b];
        row0 = (size_t)r;
        L = rs[b + 1] - r;
        if (CAUSAL && pre > 0) pb = pre;
        Lp = (pb + L + 31) / 32 * 32;
    }
    const int Lk = pb + L;
    const size_t lse0 = ((size_t)b * H + h) * Lmax;       // lse / delta stay [B (+ 1), H, Lmax], indexed by the own row
    const int dm = H * HD;
    // layout strides of qkv / dqkv (attn_fwd_body has the definition; 0 = the interleaved default).  dm stays H * HD: the shared-prefix partials' own layout
    const int hsq = q_hs ? q_hs : HD, vsq = q_vs ? q_vs : H * HD, hsd = dq_hs ? dq_hs : HD, vsd = dq_vs ? dq_vs : H * HD;
    const T* qg = qkv + row0 * ldqkv + (size_t)h * hsq;
    const int img = Lp * AT<T>::RS;
    char* q_lds = smem;
    char* k_lds = smem + img;
    char* v_lds = smem + 2 * img;
    char* do_lds = smem + 3 * img;
    float* lse_lds = reinterpret_cast<float*>(smem + 4 * img);
    float* dl_lds = lse_lds + Lp;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int g = lane >> 4;
    // everything the workgroup reads from HBM is requested before anything waits: this thread's lse value, the context rows of the wave's first query
    // blocks (delta), then the four images in one sweep
    const int li = threadIdx.x;
    const float lse_first = (li < L) ? lse[lse0 + li] : 0.f;
    Chunk oc0[NB][AT<T>::KS];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int qr = (wave * NB + j) * 16 + (lane & 15);
        load_row_chunks<T>(oc0[j], ctx + h * HD, row0 + qr, ldctx, g, qr < L);
    }
    stage_rows4<T, SV16>(k_lds, qkv + (size_t)h * hsq + vsq, v_lds, qkv + (size_t)h * hsq + 2 * (size_t)vsq, ldqkv, Lk, pb ? pb : 0, pb ? (long)row0 - pb : (long)row0, q_lds, qg, ldqkv, do_lds,
                         dctx + row0 * lddctx + h * HD, lddctx, L, Lp);
    for (int i = threadIdx.x; i < Lp; i += blockDim.x) {
        lse_lds[i] = i < L ? (i == li ? lse_first : lse[lse0 + i]) * LOG2E : INFINITY;
        if (pb) dl_lds[i] = 0.f;      // phase A covers the OWN rows' 32-row spans only; the image is as long as the KEYS
    }
    __syncthreads();

    const float c = SCALE * LOG2E;
    const char* const qp0 = q_lds + row_ptr_off<T>(lane);
    const char* const kp0 = k_lds + row_ptr_off<T>(lane);
    const char* const vp0 = v_lds + row_ptr_off<T>(lane);
    const char* const dp0 = do_lds + row_ptr_off<T>(lane);
    const char* const qt0 = q_lds + tr_ptr_off<T>(lane);
    const char* const kt0 = k_lds + tr_ptr_off<T>(lane);
    const char* const dt0 = do_lds + tr_ptr_off<T>(lane);

    // ---- phase A: a wave owns NB x 16 queries -> delta, dQ
    for (int q0 = wave * 16 * NB; q0 < L; q0 += nw * 16 * NB) {
        int qrow[NB];
        Chunk q[NB][AT<T>::KS], dO[NB][AT<T>::KS];
        float dls[NB], lq[NB];
        f32x4 dq[NB][4];
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            qrow[j] = q0 + 16 * j + (lane & 15);
            const bool valid = qrow[j] < L;
            lds_row_chunks<T>(q[j], q_lds, qrow[j], g);
            lds_row_chunks<T>(dO[j], do_lds, qrow[j], g);
            Chunk oc[AT<T>::KS];
            if (q0 == wave * 16 * NB) {      // fetched ahead of the staging
#pragma unroll
                for (int ks = 0; ks < AT<T>::KS; ++ks) oc[ks] = oc0[j][ks];
            } else {
                load_row_chunks<T>(oc, ctx + h * HD, row0 + qrow[j], ldctx, g, valid);
            }
            float dl = 0.f;
#pragma unroll
            for (int ks = 0; ks < AT<T>::KS; ++ks) {
                if constexpr (sizeof(T) == 4) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) dl += oc[ks].f[e] * dO[j][ks].f[e];
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) dl += (SV16 ? (float)oc[ks].hh[e] : (float)oc[ks].h[e]) * (float)dO[j][ks].h[e];
                }
            }
            dl = group_sum(dl);
            lq[j] = lse_lds[qrow[j]];                       // +inf for padded queries -> P = 0
            dls[j] = dl * SCALE;
            if (g == 0) {
                dl_lds[qrow[j]] = valid ? dls[j] : 0.f;     // delta * scale, for phase B
                if (valid) delta[lse0 + qrow[j]] = dl;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) dq[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        auto tile = [&](int kb, auto masked_tag) {
            constexpr bool MASKED = decltype(masked_tag)::value;
            f32x4 s0[NB], s1[NB], p0[NB], p1[NB];
            mma_lds_rows<T>(s0, kp0 + kb * AT<T>::RS, q);
            mma_lds_rows<T>(s1, kp0 + (kb + 16) * AT<T>::RS, q);
            mma_lds_rows<T>(p0, vp0 + kb * AT<T>::RS, dO);
            mma_lds_rows<T>(p1, vp0 + (kb + 16) * AT<T>::RS, dO);
#pragma unroll
            for (int j = 0; j < NB; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float e0 = fast_exp2(fmaf(s0[j][r], c, -lq[j]));
                    float e1 = fast_exp2(fmaf(s1[j][r], c, -lq[j]));
                    if constexpr (MASKED) {
                        const int k0 = kb + 4 * g + r, k1 = k0 + 16;
                        if (!(k0 < Lk && (!CAUSAL || k0 <= qrow[j] + pb))) e0 = 0.f;
                        if (!(k1 < Lk && (!CAUSAL || k1 <= qrow[j] + pb))) e1 = 0.f;
                    }
                    s0[j][r] = e0 * fmaf(p0[j][r], SCALE, -dls[j]);
                    s1[j][r] = e1 * fmaf(p1[j][r], SCALE, -dls[j]);
                }
            mma_transposed<T>(dq, kt0 + kb * AT<T>::RS, s0, s1);
        };
        if (q0 + pb >= rows_hi) continue;      // rows_hi: only dQ / dK / dV of token POSITIONS < rows_hi are wanted (delta above is needed for every row)
        const int qlast = q0 + pb + 16 * NB - 1;
        const int kend = CAUSAL ? min(Lp, (qlast / 32 + 1) * 32) : Lp;
        const int kfull = CAUSAL ? min((Lk / 32) * 32, ((q0 + pb) / 32) * 32) : (Lk / 32) * 32;
        int kb = 0;
        for (; kb < kfull; kb += 32) tile(kb, std::false_type{});
        for (; kb < kend; kb += 32) tile(kb, std::true_type{});
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            if constexpr (sizeof(T) == 2) {
                store_row_bf16_t(reinterpret_cast<bf16_t*>(dqkv) + (row0 + qrow[j]) * lddqkv + (size_t)h * hsd, dq[j], g, qrow[j] < L);
            } else if (qrow[j] < L) {
                T* dst = dqkv + (row0 + qrow[j]) * lddqkv + (size_t)h * hsd + 4 * g;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) Elem<T>::st4(dst + dt * 16, dq[j][dt]);
            }
        }
    }
    __syncthreads();   // every row < Lp of dl_lds was written: the waves' 32-row spans tile [0, Lp)

    // ---- phase B: a wave owns NB x 16 keys -> dK, dV
    for (int k0 = wave * 16 * NB; k0 < min(Lk, rows_hi); k0 += nw * 16 * NB) {
        int krow[NB];
        Chunk kk[NB][AT<T>::KS], vv[NB][AT<T>::KS];
        f32x4 dk[NB][4], dv[NB][4];
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            krow[j] = k0 + 16 * j + (lane & 15);
            lds_row_chunks<T>(kk[j], k_lds, krow[j], g);
            lds_row_chunks<T>(vv[j], v_lds, krow[j], g);
#pragma unroll
            for (int i = 0; i < 4; ++i) { dk[j][i] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[j][i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        }
        auto tile = [&](int qb, auto masked_tag) {
            constexpr bool MASKED = decltype(masked_tag)::value;
            f32x4 s0[NB], s1[NB], p0[NB], p1[NB], e0[NB], e1[NB];
            mma_lds_rows<T>(s0, qp0 + qb * AT<T>::RS, kk);
            mma_lds_rows<T>(s1, qp0 + (qb + 16) * AT<T>::RS, kk);
            mma_lds_rows<T>(p0, dp0 + qb * AT<T>::RS, vv);
            mma_lds_rows<T>(p1, dp0 + (qb + 16) * AT<T>::RS, vv);
            const f32x4 l0 = *reinterpret_cast<const f32x4*>(lse_lds + qb + 4 * g);
            const f32x4 l1 = *reinterpret_cast<const f32x4*>(lse_lds + qb + 16 + 4 * g);
            const f32x4 d0 = *reinterpret_cast<const f32x4*>(dl_lds + qb + 4 * g);
            const f32x4 d1 = *reinterpret_cast<const f32x4*>(dl_lds + qb + 16 + 4 * g);
#pragma unroll
            for (int j = 0; j < NB; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    e0[j][r] = fast_exp2(fmaf(s0[j][r], c, -l0[r]));
                    e1[j][r] = fast_exp2(fmaf(s1[j][r], c, -l1[r]));
                    if constexpr (MASKED) {
                        const int qa = qb + pb + 4 * g + r, qc = qa + 16;      // query positions
                        if (krow[j] > qa) e0[j][r] = 0.f;
                        if (krow[j] > qc) e1[j][r] = 0.f;
                    }
                    s0[j][r] = e0[j][r] * fmaf(p0[j][r], SCALE, -d0[r]);
                    s1[j][r] = e1[j][r] * fmaf(p1[j][r], SCALE, -d1[r]);
                }
            mma_transposed<T>(dv, dt0 + qb * AT<T>::RS, e0, e1);
            mma_transposed<T>(dk, qt0 + qb * AT<T>::RS, s0, s1);
        };
        // query tiles by OWN index; a key at position k is seen by the own queries i with i + pb >= k
        const int kq = k0 - pb;
        int qb = CAUSAL ? (kq > 0 ? (kq / 32) * 32 : 0) : 0;
        if constexpr (CAUSAL) {
            const int qdiag = (kq + 16 * NB - 1 >= 0) ? min(Lp, ((kq + 16 * NB - 1) / 32 + 1) * 32) : 0;
            for (; qb < qdiag; qb += 32) tile(qb, std::true_type{});
        }
        for (; qb < Lp; qb += 32) tile(qb, std::false_type{});
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            if (pb && krow[j] < pb) {      // a SHARED key: this sample's f32 partial (summed over the samples by lpi_shared_kv_reduce)
                float* dst = shared_dkv + ((size_t)b * pb + krow[j]) * 2 * dm + h * HD + 4 * g;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    *reinterpret_cast<f32x4*>(dst + dt * 16) = dk[j][dt];
                    *reinterpret_cast<f32x4*>(dst + dm + dt * 16) = dv[j][dt];
                }
                continue;
            }
            const size_t orow = row0 + (krow[j] - pb);      // own key
            if constexpr (sizeof(T) == 2) {
                bf16_t* dst = reinterpret_cast<bf16_t*>(dqkv) + orow * lddqkv + (size_t)h * hsd;
                store_row_bf16_t(dst + vsd, dk[j], g, krow[j] < Lk);
                store_row_bf16_t(dst + 2 * (size_t)vsd, dv[j], g, krow[j] < Lk);
            } else if (krow[j] < Lk) {
                T* dst = dqkv + orow * lddqkv + (size_t)h * hsd + 4 * g;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    Elem<T>::st4(dst + vsd + dt * 16, dk[j][dt]);
                    Elem<T>::st4(dst + 2 * (size_t)vsd + dt * 16, dv[j][dt]);
                }
            }
        }
    }
}

// waves per workgroup: each wave owns NB 16-row blocks per round; balance the rounds
inline int pick_waves(int L) {
    const int nqb = (L + 16 * NB - 1) / (16 * NB);
    const int rounds = (nqb + 7) / 8;
    return (nqb + rounds - 1) / rounds;
}
// ragged batches (the text tower: 2 000 workgroups of a few dozen rows, latency-bound): tuning key 9 = minimum number of waves per workgroup — the waves
// beyond the row blocks take part in the staging only (more loads in flight per workgroup)
inline int pick_waves_ragged(int L) {
    const int w = pick_waves(L), lo = g_lpi_tuning[9];
    return lo > w ? (lo < 8 ? lo : 8) : w;
}

// allow the full 160 KiB of a CU's LDS for a kernel: once per (kernel function, device), safe from any host thread
int set_lds(const void* kern, size_t bytes) {
    if (bytes > 160 * 1024) return LPI_EINVAL;
    static const void* fn[32];
    static LdsOnce once[32];
    static std::atomic<int> nfn{0};
    static std::atomic_flag lock = ATOMIC_FLAG_INIT;
    int slot = -1;
    const int n = nfn.load(std::memory_order_acquire);
    for (int i = 0; i < n; ++i)
        if (fn[i] == kern) { slot = i; break; }
    if (slot < 0) {
        while (lock.test_and_set(std::memory_order_acquire)) {}
        const int m = nfn.load(std::memory_order_relaxed);
        for (int i = 0; i < m; ++i)
            if (fn[i] == kern) { slot = i; break; }
        if (slot < 0 && m < 32) { fn[m] = kern; slot = m; nfn.store(m + 1, std::memory_order_release); }
        lock.clear(std::memory_order_release);
    }
    if (slot < 0) return (int)hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    return lpi_ensure_lds(once[slot], kern, 160 * 1024);
}

// the padded images of two workgroups do not fit a CU's 160 KiB but the unpadded ones do (tuning key 13 = 1: never; A/B switch)
static bool fwd_swizzled(int Lp) { return g_lpi_tuning[13] != 1 && (size_t)2 * 2 * Lp * 160 > (size_t)160 * 1024 && (size_t)2 * 2 * Lp * 128 <= (size_t)160 * 1024; }

template <typename T, bool CAUSAL>
int fwd_launch(int B, int L, int H, const void* qkv, int ldqkv, void* ctx, int ldctx, float* lse, hipStream_t s, const int* rs = nullptr, int pre = 0,
               int hs = 0, int vs = 0, int chs = 0) {
    // pre > 0 (causal ragged batches): B tail samples + the shared sequence as sample B (attn_fwd_body)
    const int nbh = (B + (pre > 0 ? 1 : 0)) * H, bsh = pre > 0 ? B : -1;
    const int Lp = (L + 31) / 32 * 32;
    const size_t lds = (size_t)2 * Lp * AT<T>::RS;
    const int thr = 64 * (rs ? pick_waves_ragged(L) : pick_waves(L));
    if constexpr (sizeof(T) == 2 && !CAUSAL) {
        if (fwd_swizzled(Lp) && !rs) {      // 257 .. 288 tokens: unpadded swizzled images, two workgroups per CU (stage_rows2)
            const size_t lsw = (size_t)2 * Lp * 128;
            if (int e = set_lds((const void*)attn_fwd_kernel<T, CAUSAL, true>, lsw)) return e;
            LPI_LAUNCH((attn_fwd_kernel<T, CAUSAL, true>), dim3(nbh), dim3(thr), lsw, s, L, Lp, rs, H, (const T*)qkv, ldqkv, (T*)ctx, ldctx, lse, pre, bsh, hs, vs, chs);
            LPI_CHECK_LAST();
            return 0;
        }
    }
    int e = set_lds((const void*)attn_fwd_kernel<T, CAUSAL>, lds);
    if (e) return e;
    LPI_LAUNCH((attn_fwd_kernel<T, CAUSAL>), dim3(nbh), dim3(thr), lds, s, L, Lp, rs, H, (const T*)qkv, ldqkv, (T*)ctx, ldctx, lse, pre, bsh, hs, vs, chs);
    LPI_CHECK_LAST();
    return 0;
}

template <typename T, bool CAUSAL, bool SV16 = false>
int bwd_launch(int B, int L, int H, const void* qkv, int ldqkv, const void* ctx, int ldctx, const void* dctx, int lddctx,
               const float* lse, float* delta, void* dqkv, int lddqkv, hipStream_t s, const int* rs = nullptr, int rows_hi = 1 << 30, int pre = 0,
               float* shared_dkv = nullptr, const int* lay = nullptr) {
    // lay: NULL = interleaved, else {qkv_hs, qkv_vs, dqkv_hs, dqkv_vs, ...} (lpi_attn_bwd_layout; ctx / dctx keep their 64-element head stride here)
    const int l0 = lay ? lay[0] : 0, l1 = lay ? lay[1] : 0, l2 = lay ? lay[2] : 0, l3 = lay ? lay[3] : 0;
    const int Lp = (L + 31) / 32 * 32;
    const size_t ldsA = (size_t)2 * Lp * AT<T>::RS;
    const size_t ldsB = ldsA + (size_t)2 * Lp * sizeof(float);
    const int thr = 64 * (rs ? pick_waves_ragged(L) : pick_waves(L));
    const size_t ldsF = (size_t)4 * Lp * AT<T>::RS + (size_t)2 * Lp * sizeof(float);
    // fused single pass for 2-byte operands (HBM bound: -7 % at L = 213, -23 % at L = 77); f32 is compute bound and faster with the
    // two-pass kernels at two workgroups per CU.  Tuning key 3 != 0 forces the two-pass kernels.
    if (sizeof(T) == 2 && ldsF <= 160 * 1024 && g_lpi_tuning[3] == 0) {
        int ef = set_lds((const void*)attn_bwd_fused_kernel<T, CAUSAL, SV16>, ldsF);
        if (ef) return ef;
        LPI_LAUNCH((attn_bwd_fused_kernel<T, CAUSAL, SV16>), dim3((B + (pre > 0 ? 1 : 0)) * H), dim3(thr), ldsF, s, L, Lp, rs, H, (const T*)qkv, ldqkv,
                   (const T*)ctx, ldctx, (const T*)dctx, lddctx, lse, delta, (T*)dqkv, lddqkv, rows_hi, pre, pre > 0 ? B : -1, shared_dkv, l0, l1, l2, l3);
        LPI_CHECK_LAST();
        return 0;
    }
    int e = set_lds((const void*)attn_bwd_dq_kernel<T, CAUSAL, SV16>, ldsA);
    if (e) return e;
    e = set_lds((const void*)attn_bwd_dkv_kernel<T, CAUSAL, SV16>, ldsB);
    if (e) return e;
    const int nbh = (B + (pre > 0 ? 1 : 0)) * H, bsh = pre > 0 ? B : -1;
    LPI_LAUNCH((attn_bwd_dq_kernel<T, CAUSAL, SV16>), dim3(nbh), dim3(thr), ldsA, s, L, Lp, rs, H, (const T*)qkv, ldqkv, (const T*)ctx, ldctx,
               (const T*)dctx, lddctx, lse, delta, (T*)dqkv, lddqkv, pre, bsh, l0, l1, l2, l3);
    LPI_CHECK_LAST();
    LPI_LAUNCH((attn_bwd_dkv_kernel<T, CAUSAL, SV16>), dim3(nbh), dim3(thr), ldsB, s, L, Lp, rs, H, (const T*)qkv, ldqkv, (const T*)dctx, lddctx,
               lse, delta, (T*)dqkv, lddqkv, pre, bsh, shared_dkv, l0, l1, l2, l3);
    LPI_CHECK_LAST();
    return 0;
}

inline bool bad_attn(int dtype, int B, int L, int H, int ld, int lmax = 288) {
    const int esz = dtype == LPI_F32 ? 4 : 2;
    return B <= 0 || H <= 0 || L <= 0 || L > lmax || ld < 3 * H * HD || (ld * esz) % 16;
}

}  // namespace

// attn_long.hip: 288 < L <= 1024, non-causal, uniform sequences — tiled over the keys with an online softmax (ViT-L/14@336px: 577 tokens + prompts)
bool lpi_attn_long_ok(int L, int causal, const void* row_start);
int lpi_attn_long_fwd(int dtype, int B, int L, int H, const void* qkv, int ldqkv, void* ctx, int ldctx, float* lse, hipStream_t s);
int lpi_attn_long_bwd(int dtype, int B, int L, int H, const void* qkv, int ldqkv, const void* ctx, int ldctx, const void* dctx, int lddctx, const float* lse,
                      float* delta, void* dqkv, int lddqkv, hipStream_t s);

// attention4.hip: the backward as ONE pass, 8 waves owning 16-32 keys each, query slices streamed through an LDS ring; non-causal, L <= 224;
// the default for L > 160 (tuning key 7 = 0), forced at every L it takes by key 7 = 5, never with key 7 = 1
bool lpi_attn4_bwd_ok(int L, int causal);
int lpi_attn4_bwd(int B, int L, int H, const void* qkv, int ldqkv, const void* ctx, int ldctx, const void* dctx, int lddctx, const float* lse,
                  float* delta, void* dqkv, int lddqkv, hipStream_t s, int saved_f16, int rows_hi, const int* lay = nullptr);
extern "C" int lpi_attn_fwd_varlen(int dtype, int B, int L, const int32_t* row_start, int H, const void* qkv, int ldqkv, void* ctx, int ldctx,
                                   float* lse, int causal, void* stream) {
    const int* rs = row_start;      // ragged batch: the one-head-per-workgroup kernels take a per-sample length
    const bool lng = lpi_attn_long_ok(L, causal, row_start);
    if (!qkv || !ctx || !lse || bad_attn(dtype, B, L, H, ldqkv, lng ? 1024 : 288) || ldctx < H * HD || (ldctx & 7)) return LPI_EINVAL;
    if (((uintptr_t)qkv | (uintptr_t)ctx) & 15) return LPI_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (lng) return lpi_attn_long_fwd(dtype, B, L, H, qkv, ldqkv, ctx, ldctx, lse, s);
    if (dtype == LPI_F32)
        return causal ? fwd_launch<float, true>(B, L, H, qkv, ldqkv, ctx, ldctx, lse, s, rs) : fwd_launch<float, false>(B, L, H, qkv, ldqkv, ctx, ldctx, lse, s, rs);
    if (dtype == LPI_F16)       // f16 operand mode: q, k, v and ctx are fp16 (the reference's own arithmetic type)
        return causal ? fwd_launch<f16_t, true>(B, L, H, qkv, ldqkv, ctx, ldctx, lse, s, rs) : fwd_launch<f16_t, false>(B, L, H, qkv, ldqkv, ctx, ldctx, lse, s, rs);
    if (dtype == LPI_BF16) {
        // (a persistent forward with K / V double-buffered by LDS-DMA measured SLOWER than two one-head workgroups per CU — 112.7 vs
        // 102.5 us at L = 213 —: tools/probe/attention2.hip, profiles/r02_gemm_experiments.md)
        return causal ? fwd_launch<bf16_t, true>(B, L, H, qkv, ldqkv, ctx, ldctx, lse, s, rs) : fwd_launch<bf16_t, false>(B, L, H, qkv, ldqkv, ctx, ldctx, lse, s, rs);
    }
    return LPI_EINVAL;
}

extern "C" int lpi_attn_fwd(int dtype, int B, int L, int H, const void* qkv, int ldqkv, void* ctx, int ldctx, float* lse, int causal,
                            void* stream) {
    return lpi_attn_fwd_varlen(dtype, B, L, nullptr, H, qkv, ldqkv, ctx, ldctx, lse, causal, stream);
}

// two forwards (2-byte operand types) in one launch; any other case runs as two launches
template <typename T>
static int fwd_pair_launch(const lpi_attn_fwd_desc* d, hipStream_t s) {
    AttnFwdP<T> p[2];
    size_t lds = 0;
    int thr = 0;
    for (int i = 0; i < 2; ++i) {
        const int Lp = (d[i].L + 31) / 32 * 32;
        const int pre = d[i].shared_rows;
        if (pre < 0 || (pre > 0 && (!d[i].causal || !d[i].row_start || pre >= d[i].L))) return LPI_EINVAL;
        p[i] = AttnFwdP<T>{d[i].L, Lp, d[i].H, d[i].ldqkv, d[i].ldctx, d[i].row_start, (const T*)d[i].qkv, (T*)d[i].ctx, d[i].lse, pre, pre > 0 ? d[i].B : -1,
                           d[i].qkv_hs, d[i].qkv_vs, d[i].ctx_hs};
        lds = std::max(lds, (size_t)2 * Lp * AT<T>::RS);
        thr = std::max(thr, 64 * (d[i].row_start ? pick_waves_ragged(d[i].L) : pick_waves(d[i].L)));
    }
    const int nb0 = (d[0].B + (d[0].shared_rows > 0)) * d[0].H, nb1 = (d[1].B + (d[1].shared_rows > 0)) * d[1].H;
    if (!d[0].causal && d[1].causal && !d[0].row_start && fwd_swizzled(p[0].Lp)) {      // a long vision sequence beside the text tower: problem 0 on the swizzled images
        const size_t lsw = std::max((size_t)2 * p[0].Lp * 128, (size_t)2 * p[1].Lp * AT<T>::RS);
        if (int e = set_lds((const void*)attn_fwd_pair_kernel<T, false, true, true>, lsw)) return e;
        LPI_LAUNCH((attn_fwd_pair_kernel<T, false, true, true>), dim3(nb0 + nb1), dim3(thr), lsw, s, p[0], p[1], nb0);
        LPI_CHECK_LAST();
        return 0;
    }
#define FP(C0, C1)                                                                                                              \
    do {                                                                                                                        \
        if (int e = set_lds((const void*)attn_fwd_pair_kernel<T, C0, C1>, lds)) return e;                                       \
        LPI_LAUNCH((attn_fwd_pair_kernel<T, C0, C1>), dim3(nb0 + nb1), dim3(thr), lds, s, p[0], p[1], nb0);                     \
    } while (0)
    if (d[0].causal && d[1].causal) FP(true, true);
    else if (d[0].causal) FP(true, false);
    else if (d[1].causal) FP(false, true);
    else FP(false, false);
#undef FP
    LPI_CHECK_LAST();
    return 0;
}
extern "C" int lpi_attn_fwd_varlen(int dtype, int B, int L, const int32_t* row_start, int H, const void* qkv, int ldqkv, void* ctx, int ldctx,
                                   float* lse, int causal, void* stream);
extern "C" int lpi_attn_fwd_pair(int dtype, const lpi_attn_fwd_desc* d, void* stream) {
    if (!d) return LPI_EINVAL;
    bool any_long = false;
    for (int i = 0; i < 2; ++i) {
        const bool lay = d[i].qkv_hs || d[i].qkv_vs || d[i].ctx_hs;      // an explicit layout: the strides are the caller's statement, only their alignment is checked
        if (lay && (dtype == LPI_F32 || d[i].row_start || d[i].shared_rows || !d[i].qkv_hs || !d[i].qkv_vs || !d[i].ctx_hs ||
                    ((d[i].qkv_hs | d[i].qkv_vs | d[i].ctx_hs | d[i].ldqkv | d[i].ldctx) & 7) || d[i].ldqkv < HD || d[i].ldctx < HD)) return LPI_EINVAL;
        const bool lng = !lay && !d[i].shared_rows && lpi_attn_long_ok(d[i].L, d[i].causal, d[i].row_start);      // a long sequence: its own launch (attn_long.hip)
        any_long |= lng;
        if (!d[i].qkv || !d[i].ctx || !d[i].lse || (!lay && (bad_attn(dtype, d[i].B, d[i].L, d[i].H, d[i].ldqkv, lng ? 1024 : 288) || d[i].ldctx < d[i].H * HD)) || (d[i].ldctx & 7))
            return LPI_EINVAL;
        if (lay && (d[i].B <= 0 || d[i].H <= 0 || d[i].L <= 0 || d[i].L > 288)) return LPI_EINVAL;
        if (((uintptr_t)d[i].qkv | (uintptr_t)d[i].ctx) & 15) return LPI_EINVAL;
    }
    if (dtype == LPI_BF16 && !any_long) return fwd_pair_launch<bf16_t>(d, (hipStream_t)stream);
    if (dtype == LPI_F16 && !any_long) return fwd_pair_launch<f16_t>(d, (hipStream_t)stream);
    for (int i = 0; i < 2; ++i) {
        if (d[i].shared_rows) {
            if (int e = lpi_attn_fwd_shared(dtype, d[i].B, d[i].L, d[i].row_start, d[i].shared_rows, d[i].H, d[i].qkv, d[i].ldqkv, d[i].ctx, d[i].ldctx, d[i].lse, stream)) return e;
            continue;
        }
        if (int e = lpi_attn_fwd_varlen(dtype, d[i].B, d[i].L, d[i].row_start, d[i].H, d[i].qkv, d[i].ldqkv, d[i].ctx, d[i].ldctx, d[i].lse, d[i].causal, stream))
            return e;
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------ shared prefix (see attn_fwd_body)
namespace {
// dqkv[key][K | V columns] (+)= sum over the B samples of their f32 partials part[b][key][dK (dm) | dV (dm)].  A workgroup owns 16 consecutive float4 outputs;
// its 16 x 16 threads each add up every 16th sample (b = slice, slice + 16, ...) and the 16 slice sums are added in slice order through LDS: a FIXED
// summation tree (bitwise reproducible), 272 workgroups at H = 8 instead of the 17 a thread-per-output kernel fills (measured 23.8 us -> see profiles/)
template <typename T>
__global__ __launch_bounds__(256) void shared_kv_reduce_kernel(int B, int pre, int dm, const float* __restrict__ part, T* __restrict__ dqkv, int ld, int accumulate) {
    __shared__ f32x4 red[16][16];
    const int per_row = 2 * dm / 4, n = pre * per_row;
    const int o = threadIdx.x & 15, slice = threadIdx.x >> 4;
    const int t = blockIdx.x * 16 + o;
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    if (t < n) {
        const int key = t / per_row, c = (t % per_row) * 4;
        const float* p = part + (size_t)key * 2 * dm + c;
        const size_t bstride = (size_t)pre * 2 * dm;
        f32x4 a1 = a;
        int b = slice;
        for (; b + 16 < B; b += 32) {      // two independent chains per thread
            a += *reinterpret_cast<const f32x4*>(p + (size_t)b * bstride);
            a1 += *reinterpret_cast<const f32x4*>(p + (size_t)(b + 16) * bstride);
        }
        if (b < B) a += *reinterpret_cast<const f32x4*>(p + (size_t)b * bstride);
        a += a1;
    }
    red[slice][o] = a;
    __syncthreads();
    if (slice == 0 && t < n) {
        f32x4 v = red[0][o];
#pragma unroll
        for (int i = 1; i < 16; ++i) v += red[i][o];
        const int key = t / per_row, c = (t % per_row) * 4;
        T* out = dqkv + (size_t)key * ld + dm + c;      // K columns at dm .., V columns at 2 dm .. : column dm + c for c in [0, 2 dm)
        if (accumulate) v += Elem<T>::ld4(out);
        Elem<T>::st4(out, v);
    }
}
}  // namespace

extern "C" int lpi_shared_kv_reduce(int dtype, int B, int shared_rows, int H, const float* partial, void* dqkv, int lddqkv, int accumulate, void* stream) {
    if (!partial || !dqkv || B <= 0 || shared_rows <= 0 || H <= 0 || lddqkv < 3 * H * HD || (lddqkv & 3) || ((uintptr_t)partial & 15) || ((uintptr_t)dqkv & (dtype == LPI_F32 ? 15 : 7))) return LPI_EINVAL;
    const int dm = H * HD, n = shared_rows * (2 * dm / 4);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == LPI_BF16 || dtype == LPI_F16)      // F16: "saved activations fp16, gradients bf16" (lpi_attn_bwd_prefix)
        LPI_LAUNCH((shared_kv_reduce_kernel<bf16_t>), dim3((n + 15) / 16), dim3(256), 0, s, B, shared_rows, dm, partial, (bf16_t*)dqkv, lddqkv, accumulate);
    else if (dtype == LPI_F32)
        LPI_LAUNCH((shared_kv_reduce_kernel<float>), dim3((n + 15) / 16), dim3(256), 0, s, B, shared_rows, dm, partial, (float*)dqkv, lddqkv, accumulate);
    else
        return LPI_ENOSYS;
    LPI_CHECK_LAST();
    return 0;
}

extern "C" int lpi_attn_fwd_shared(int dtype, int B, int L, const int32_t* row_start, int shared_rows, int H, const void* qkv, int ldqkv, void* ctx, int ldctx,
                                   float* lse, void* stream) {
    if (!qkv || !ctx || !lse || !row_start || shared_rows <= 0 || shared_rows >= L || bad_attn(dtype, B, L, H, ldqkv) || ldctx < H * HD || (ldctx & 7)) return LPI_EINVAL;
    if (((uintptr_t)qkv | (uintptr_t)ctx) & 15) return LPI_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == LPI_F16) return fwd_launch<f16_t, true>(B, L, H, qkv, ldqkv, ctx, ldctx, lse, s, row_start, shared_rows);
    if (dtype == LPI_BF16) return fwd_launch<bf16_t, true>(B, L, H, qkv, ldqkv, ctx, ldctx, lse, s, row_start, shared_rows);
    if (dtype == LPI_F32) return fwd_launch<float, true>(B, L, H, qkv, ldqkv, ctx, ldctx, lse, s, row_start, shared_rows);
    return LPI_ENOSYS;
}

extern "C" int lpi_attn_bwd_shared(int dtype, int B, int L, const int32_t* row_start, int shared_rows, int rows_needed, int H, const void* qkv, int ldqkv,
                                   const void* ctx, int ldctx, const void* dctx, int lddctx, const float* lse, float* delta, void* dqkv, int lddqkv,
                                   float* shared_dkv, void* stream) {
    if (!row_start || !shared_dkv || shared_rows <= 0 || shared_rows >= L || rows_needed < shared_rows) return LPI_EINVAL;
    const int rows_hi = rows_needed >= L ? (1 << 30) : rows_needed;
    if (!qkv || !ctx || !dctx || !lse || !delta || !dqkv || bad_attn(dtype, B, L, H, ldqkv) || bad_attn(dtype, B, L, H, lddqkv)) return LPI_EINVAL;
    const int esz = dtype == LPI_F32 ? 4 : 2;
    if (ldctx < H * HD || lddctx < H * HD || (ldctx * esz) % 16 || (lddctx * esz) % 16) return LPI_EINVAL;
    if (((uintptr_t)qkv | (uintptr_t)ctx | (uintptr_t)dctx | (uintptr_t)dqkv | (uintptr_t)shared_dkv) & 15) return LPI_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    int e;
    if (dtype == LPI_F32)      // the two-pass kernels (f32 is compute-bound); rows_needed is not used: everything is computed
        e = bwd_launch<float, true>(B, L, H, qkv, ldqkv, ctx, ldctx, dctx, lddctx, lse, delta, dqkv, lddqkv, s, row_start, rows_hi, shared_rows, shared_dkv);
    else if (dtype == LPI_F16) e = bwd_launch<bf16_t, true, true>(B, L, H, qkv, ldqkv, ctx, ldctx, dctx, lddctx, lse, delta, dqkv, lddqkv, s, row_start, rows_hi, shared_rows, shared_dkv);
    else if (dtype == LPI_BF16) e = bwd_launch<bf16_t, true>(B, L, H, qkv, ldqkv, ctx, ldctx, dctx, lddctx, lse, delta, dqkv, lddqkv, s, row_start, rows_hi, shared_rows, shared_dkv);
    else return LPI_ENOSYS;
    if (e) return e;
    // the shared sequence's own workgroups wrote dK / dV of the shared rows; the tail samples' partial sums are added on top
    return lpi_shared_kv_reduce(dtype, B, shared_rows, H, shared_dkv, dqkv, lddqkv, 1, stream);
}

extern "C" int lpi_attn_bwd_prefix(int dtype, int B, int L, const int32_t* row_start, int rows_needed, int H, const void* qkv, int ldqkv, const void* ctx,
                                   int ldctx, const void* dctx, int lddctx, const float* lse, float* delta, void* dqkv, int lddqkv, int causal,
                                   void* stream) {
    const int* rs = row_start;
    if (rows_needed <= 0) return LPI_EINVAL;
    const int rows_hi = rows_needed >= L ? (1 << 30) : rows_needed;      // the 2-byte kernels skip the 32-row blocks behind it; f32 computes all
    const bool lng = lpi_attn_long_ok(L, causal, row_start);
    const int lmax = lng ? 1024 : 288;
    if (!qkv || !ctx || !dctx || !lse || !delta || !dqkv || bad_attn(dtype, B, L, H, ldqkv, lmax) || bad_attn(dtype, B, L, H, lddqkv, lmax)) return LPI_EINVAL;
    const int esz = dtype == LPI_F32 ? 4 : 2;
    if (ldctx < H * HD || lddctx < H * HD || (ldctx * esz) % 16 || (lddctx * esz) % 16) return LPI_EINVAL;
    if (((uintptr_t)qkv | (uintptr_t)ctx | (uintptr_t)dctx | (uintptr_t)dqkv) & 15) return LPI_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (lng) return lpi_attn_long_bwd(dtype, B, L, H, qkv, ldqkv, ctx, ldctx, dctx, lddctx, lse, delta, dqkv, lddqkv, s);      // every row (rows_needed is a saving, not a contract)
    if (dtype == LPI_F32)
        return causal ? bwd_launch<float, true>(B, L, H, qkv, ldqkv, ctx, ldctx, dctx, lddctx, lse, delta, dqkv, lddqkv, s, rs, rows_hi)
                      : bwd_launch<float, false>(B, L, H, qkv, ldqkv, ctx, ldctx, dctx, lddctx, lse, delta, dqkv, lddqkv, s, rs, rows_hi);
    if (dtype == LPI_F16) {     // saved qkv / ctx are fp16 (f16-mode forward); dctx and dqkv are bf16, and so are the MFMA operands:
                                // q, k, v are converted on their way into LDS / registers (attention4.hip: in registers after the LDS read)
        if (!rs && g_lpi_tuning[3] == 0 && lpi_attn4_bwd_ok(L, causal) && (g_lpi_tuning[7] == 5 || (g_lpi_tuning[7] == 0 && L > 160)))
            return lpi_attn4_bwd(B, L, H, qkv, ldqkv, ctx, ldctx, dctx, lddctx, lse, delta, dqkv, lddqkv, s, 1, rows_hi);
        return causal ? bwd_launch<bf16_t, true, true>(B, L, H, qkv, ldqkv, ctx, ldctx, dctx, lddctx, lse, delta, dqkv, lddqkv, s, rs, rows_hi)
                      : bwd_launch<bf16_t, false, true>(B, L, H, qkv, ldqkv, ctx, ldctx, dctx, lddctx, lse, delta, dqkv, lddqkv, s, rs, rows_hi);
    }
    if (dtype == LPI_BF16) {
        // the streamed single-pass backward where a head's matrices would fill a CU's LDS (L = 213: 165-175 us against 271 us for the fused
        // kernel below and 228 us for the two-phase persistent one, tools/probe/attention2.hip); at short L several one-head workgroups
        // per CU are faster
        if (!rs && g_lpi_tuning[3] == 0 && lpi_attn4_bwd_ok(L, causal) && (g_lpi_tuning[7] == 5 || (g_lpi_tuning[7] == 0 && L > 160)))
            return lpi_attn4_bwd(B, L, H, qkv, ldqkv, ctx, ldctx, dctx, lddctx, lse, delta, dqkv, lddqkv, s, 0, rows_hi);
        return causal ? bwd_launch<bf16_t, true>(B, L, H, qkv, ldqkv, ctx, ldctx, dctx, lddctx, lse, delta, dqkv, lddqkv, s, rs, rows_hi)
                      : bwd_launch<bf16_t, false>(B, L, H, qkv, ldqkv, ctx, ldctx, dctx, lddctx, lse, delta, dqkv, lddqkv, s, rs, rows_hi);
    }
    return LPI_EINVAL;
}

extern "C" int lpi_attn_bwd_varlen(int dtype, int B, int L, const int32_t* row_start, int H, const void* qkv, int ldqkv, const void* ctx, int ldctx,
                                   const void* dctx, int lddctx, const float* lse, float* delta, void* dqkv, int lddqkv, int causal, void* stream) {
    return lpi_attn_bwd_prefix(dtype, B, L, row_start, L, H, qkv, ldqkv, ctx, ldctx, dctx, lddctx, lse, delta, dqkv, lddqkv, causal, stream);
}

extern "C" int lpi_attn_bwd(int dtype, int B, int L, int H, const void* qkv, int ldqkv, const void* ctx, int ldctx, const void* dctx,
                            int lddctx, const float* lse, float* delta, void* dqkv, int lddqkv, int causal, void* stream) {
    return lpi_attn_bwd_varlen(dtype, B, L, nullptr, H, qkv, ldqkv, ctx, ldctx, dctx, lddctx, lse, delta, dqkv, lddqkv, causal, stream);
}

// The backward on an explicit LAYOUT (round 6; see lpi_attn_fwd_desc): lay = {qkv_hs, qkv_vs, dqkv_hs, dqkv_vs, ctx_hs, dctx_hs}, element strides, multiples
// of 8.  2-byte operand types, non-causal, uniform sequences.  Dispatch as lpi_attn_bwd_prefix: the streamed single-pass kernel (attention4.hip) where it is
// the default, else the one-head-per-workgroup kernels of this file — those keep ctx / dctx at their interleaved head stride (ctx_hs = dctx_hs = 64).
extern "C" int lpi_attn_bwd_layout(int dtype, int B, int L, int rows_needed, int H, const void* qkv, int ldqkv, const void* ctx, int ldctx, const void* dctx,
                                   int lddctx, const float* lse, float* delta, void* dqkv, int lddqkv, const int32_t* lay, void* stream) {
    if (!qkv || !ctx || !dctx || !lse || !delta || !dqkv || !lay || B <= 0 || H <= 0 || L <= 0 || L > 288) return LPI_EINVAL;
    if (dtype != LPI_BF16 && dtype != LPI_F16) return LPI_EINVAL;
    int bits = ldqkv | ldctx | lddctx | lddqkv;
    for (int i = 0; i < 6; ++i) { if (lay[i] <= 0) return LPI_EINVAL; bits |= lay[i]; }
    if ((bits & 7) || ldqkv < HD || ldctx < HD || lddctx < HD || lddqkv < HD) return LPI_EINVAL;
    if (((uintptr_t)qkv | (uintptr_t)ctx | (uintptr_t)dctx | (uintptr_t)dqkv) & 15) return LPI_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int rows_hi = rows_needed > 0 && rows_needed < L ? rows_needed : L;
    const int sv16 = dtype == LPI_F16 ? 1 : 0;
    if (g_lpi_tuning[3] == 0 && lpi_attn4_bwd_ok(L, 0) && (g_lpi_tuning[7] == 5 || (g_lpi_tuning[7] == 0 && L > 160)))
        return lpi_attn4_bwd(B, L, H, qkv, ldqkv, ctx, ldctx, dctx, lddctx, lse, delta, dqkv, lddqkv, s, sv16, rows_hi, lay);
    if (lay[4] != HD || lay[5] != HD) return LPI_EINVAL;
    return sv16 ? bwd_launch<bf16_t, false, true>(B, L, H, qkv, ldqkv, ctx, ldctx, dctx, lddctx, lse, delta, dqkv, lddqkv, s, nullptr, rows_hi, 0, nullptr, lay)
                : bwd_launch<bf16_t, false>(B, L, H, qkv, ldqkv, ctx, ldctx, dctx, lddctx, lse, delta, dqkv, lddqkv, s, nullptr, rows_hi, 0, nullptr, lay);
}

// ONE forward in descriptor form (the only single-problem form that takes the layout strides of lpi_attn_fwd_desc)
extern "C" int lpi_attn_fwd_one(int dtype, const lpi_attn_fwd_desc* d, void* stream) {
    if (!d) return LPI_EINVAL;
    const bool lay = d->qkv_hs || d->qkv_vs || d->ctx_hs;
    if (!lay) {
        if (d->shared_rows) return lpi_attn_fwd_shared(dtype, d->B, d->L, d->row_start, d->shared_rows, d->H, d->qkv, d->ldqkv, d->ctx, d->ldctx, d->lse, stream);
        return lpi_attn_fwd_varlen(dtype, d->B, d->L, d->row_start, d->H, d->qkv, d->ldqkv, d->ctx, d->ldctx, d->lse, d->causal, stream);
    }
    if ((dtype != LPI_BF16 && dtype != LPI_F16) || d->row_start || d->shared_rows || d->causal || !d->qkv_hs || !d->qkv_vs || !d->ctx_hs) return LPI_EINVAL;
    if (((d->qkv_hs | d->qkv_vs | d->ctx_hs | d->ldqkv | d->ldctx) & 7) || d->ldqkv < HD || d->ldctx < HD) return LPI_EINVAL;
    if (!d->qkv || !d->ctx || !d->lse || d->B <= 0 || d->H <= 0 || d->L <= 0 || d->L > 288 || (((uintptr_t)d->qkv | (uintptr_t)d->ctx) & 15)) return LPI_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == LPI_F16) return fwd_launch<f16_t, false>(d->B, d->L, d->H, d->qkv, d->ldqkv, d->ctx, d->ldctx, d->lse, s, nullptr, 0, d->qkv_hs, d->qkv_vs, d->ctx_hs);
    return fwd_launch<bf16_t, false>(d->B, d->L, d->H, d->qkv, d->ldqkv, d->ctx, d->ldctx, d->lse, s, nullptr, 0, d->qkv_hs, d->qkv_vs, d->ctx_hs);
}
