// The LAST block's attention WITHOUT its K and V (round 6).  replaces: models/clip/model.py:183-185 for the block whose output only the pooled token reads
// (model.py:255: ln_post(x[:, 0])).
//
// Only one query per (sample, head) is live in the last block — the pooled token's — which rounds 1-2 already used: Q, the softmax row, out_proj and the MLP run
// on B rows.  K and V were still computed for EVERY token (a [B L, 2 d] GEMM: 133 us of the step at ViT-B/16, 256 pairs) and their gradient carried back by a
// second one (126 us), with the single-query attention kernels reading and writing those 2 x 168 MB in between (57 + 90 us).  With one query, neither is needed:
//
//   score of token l, head h:  q_h . k_l / 8 = q_h . (W_k,h LN1(x_l) + b_k,h) / 8 = LN1(x_l) . qt_h + const      with qt_h = W_k,h^T q_h / 8   (a d-vector per (sample, head))
//   context of head h:         sum_l p_l (W_v,h LN1(x_l) + b_v,h) = W_v,h hbar_h + b_v,h                          with hbar_h = sum_l p_l LN1(x_l)   (sum_l p_l = 1)
//
// — constants over l drop out of the softmax.  So the block needs, per sample, H dot products of every residual-stream row with a d-vector and H weighted
// sums of the rows: TWO passes over the sample's [L, d] rows of x (LayerNorm applied on the fly from the row statistics the stream already carries:
// LN1(x_l) = r_l (x_l - mu_l) gamma + beta), 2 H L d multiply-adds each instead of the 2 L d^2 of the two projections (d / H = 64 times fewer), and no [B L, 2 d]
// tensor at all.  The backward is the same two passes: with dhbar_h = W_v,h^T dctx_h,
//   dp_l = LN1(x_l) . dhbar_h,  ds_l = p_l (dp_l - sum p dp),  d LN1(x_l) = sum_h (p_l dhbar_h + ds_l qt_h),  dqt_h = sum_l ds_l LN1(x_l),  dq_h = W_k,h dqt_h / 8.
// Exact algebra (the reference's numbers up to rounding order; in fact K and V are no longer rounded to 16 bits on the way).
//
// Kernels (2-byte operand modes with the fp16 residual stream; uniform sequences; H <= 16, d = 64 H <= 1024, L <= 288):
//   headw_t_kernel   out[b, h, :] (f32) = scale sum_c in[b, 64 h + c] W[row0 + 64 h + c, :]                  qt from q (rows of W_k), dhbar from dctx (rows of W_v)
//   headw_n_kernel   out[b, 64 h + c]   = scale sum_j W[row0 + 64 h + c, j] in[b, h, j] (+ bias)             ctx from hbar (W_v, b_v), dq from dqt (W_k)
//                    (both on the matrix pipe: 16 samples x one head per workgroup, operands straight from global memory as MFMA fragments)
//   spool_fwd_kernel one workgroup per sample: scores by MFMA (x rows straight from HBM as B fragments, gamma o qt from LDS), softmax, weighted row sums
//   spool_bwd_kernel the same passes for ds and dqt; d LN1(x) of every row as ONE 16x16x32 MFMA per output tile ([p | ds] x [dhbar ; qt], 32 = 2 x 16 head slots)
#include "common.h"
#include "../../include/lpi_hip.h"

namespace {

constexpr int HD = 64;
constexpr int HS = 16;            // head slots of the MFMA tile (H <= 16)
constexpr int NT = 512;           // threads of the stream kernels: 8 waves; a thread owns the column pair (2 tid, 2 tid + 1) in the row-sum phases
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// ------------------------------------------------------------------------------------------------ the small per-head products (MFMA)
// Fragments (common.h mma_chunk): lane l supplies row (l & 15) of each operand and the 8 consecutive k of chunk (l >> 4) + 4 ks; acc[i] = (row 4 (l >> 4) + i of
// operand A) x (row (l & 15) of operand B).  A = 16 samples throughout.

// out[b, h, j] = scale * sum_c in[b, 64 h + c] * W[row0 + 64 h + c, j]: B rows = j, k = c — read from the TRANSPOSED weight WT [d, ldwt] (row j, columns
// col0 + 64 h + c: a head's 64 output features are contiguous).  Grid (H, ceil(B / 16)), 4 waves; wave w takes the 16-column tiles w, w + 4, ...
template <typename T>
__global__ __launch_bounds__(256) void headw_t_kernel(int B, int H, int d, const T* __restrict__ in, int ldin, const T* __restrict__ WT, int ldwt, int col0, float scale,
                                                      float* __restrict__ out) {
    const int h = blockIdx.x, b0 = blockIdx.y * 16, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, r16 = lane & 15;
    Chunk a[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        a[ks].u = make_uint4(0, 0, 0, 0);
        if (b0 + r16 < B) a[ks].u = *reinterpret_cast<const uint4*>(in + (size_t)(b0 + r16) * ldin + HD * h + (g + 4 * ks) * 8);
    }
    const int ntile = d / 16;
    const T* wc = WT + col0 + HD * h;
    for (int t0 = wave; t0 < ntile; t0 += 4 * 6) {      // six tiles' loads in flight, then their MFMAs
        Chunk bf[6][2];
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            const int t = t0 + 4 * u;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf[u][ks].u = make_uint4(0, 0, 0, 0);
                if (t < ntile) bf[u][ks].u = *reinterpret_cast<const uint4*>(wc + (size_t)(t * 16 + r16) * ldwt + (g + 4 * ks) * 8);
            }
        }
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            const int t = t0 + 4 * u;
            if (t >= ntile) continue;
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
            mma_chunk<T>(acc, a[0], bf[u][0]);
            mma_chunk<T>(acc, a[1], bf[u][1]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (b0 + 4 * g + i < B) out[((size_t)(b0 + 4 * g + i) * H + h) * d + t * 16 + r16] = acc[i] * scale;
        }
    }
}

// out[b, 64 h + c] = scale * sum_j W[row0 + 64 h + c, j] * in[b, h, j] + bias[row0 + 64 h + c]: B rows = c (the weight's own rows), k = j; the f32 input is
// rounded to T on its way into the A fragments.  Grid (H, ceil(B / 16)), 4 waves = the head's four 16-feature tiles.
template <typename T, typename TO>
__global__ __launch_bounds__(256) void headw_n_kernel(int B, int H, int d, const float* __restrict__ in, const T* __restrict__ W, int ldw, int row0,
                                                      const float* __restrict__ bias, float scale, TO* __restrict__ out, int ldo) {
    const int h = blockIdx.x, b0 = blockIdx.y * 16, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, r16 = lane & 15;
    const bool va = b0 + r16 < B;
    const float* ia = in + ((size_t)(va ? b0 + r16 : 0) * H + h) * d;
    const T* wb = W + (size_t)(row0 + HD * h + 16 * wave + r16) * ldw;
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    const int KS = d / 32;
    for (int k0 = 0; k0 < KS; k0 += 8) {      // eight chunks of both operands in flight
        f32x4 lo[8], hi[8];
        Chunk bf[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int ks = k0 + u;
            lo[u] = hi[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            bf[u].u = make_uint4(0, 0, 0, 0);
            if (ks < KS) {
                if (va) {
                    lo[u] = *reinterpret_cast<const f32x4*>(ia + (g + 4 * ks) * 8);
                    hi[u] = *reinterpret_cast<const f32x4*>(ia + (g + 4 * ks) * 8 + 4);
                }
                bf[u].u = *reinterpret_cast<const uint4*>(wb + (g + 4 * ks) * 8);
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            Chunk a;
            a.u = make_uint4(pack2_t<T>(lo[u][0], lo[u][1]), pack2_t<T>(lo[u][2], lo[u][3]), pack2_t<T>(hi[u][0], hi[u][1]), pack2_t<T>(hi[u][2], hi[u][3]));
            mma_chunk<T>(acc, a, bf[u]);
        }
    }
    const int c = HD * h + 16 * wave + r16;
    const float bv = bias ? bias[row0 + c] : 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (b0 + 4 * g + i < B) Elem<TO>::st(out + (size_t)(b0 + 4 * g + i) * ldo + c, acc[i] * scale + bv);
}

// ------------------------------------------------------------------------------------------------ the two passes over a sample's rows
struct SpoolArgs {
    int B, L, H, d;
    const f16_t* x; int ldx;               // the fp16 residual stream (block input), rows b L + l
    const float* mean; const float* rstd;  // ln_1's row statistics of those rows
    const float* gamma; const float* beta;
    const float* qt;                       // [B, H, d]: W_k,h^T q_h / 8
    float* hbar;                           // fwd out [B, H, d]: sum_l p_l LN1(x_l)
    float* lse;                            // [B, H]: log sum exp of the scores (natural log; constants over l dropped — the backward uses the same convention)
    const float* dhbar;                    // bwd in [B, H, d]: W_v,h^T dctx_h
    bf16_t* dh; int lddh;                  // bwd out: d LN1(x_l), every row of the sample
    float* dqt;                            // bwd out [B, H, d]: sum_l ds_l LN1(x_l)
};

// LDS row of the gamma o vector images: d halves + 8 (rows 16 bytes apart modulo 256: the 16 rows a quarter wave reads hit 16 different 16-byte slots)
__device__ __forceinline__ int grow_bytes(int d) { return d * 2 + 16; }

// g[h][j] = fp16(gamma[j] v[h, j]) for h < H (zero rows behind), cg[h] = sum_j of the ROUNDED values (so that the mean term of the folded LayerNorm cancels
// exactly against the products).  A thread makes 8 consecutive halves at a time (all its loads first); the sums afterwards, two heads per wave, from LDS.
__device__ __forceinline__ void fill_gamma_image(char* img, float* cg, const float* __restrict__ v, const float* __restrict__ gamma, int H, int d) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, rb = grow_bytes(d), cpr = d / 8;
    for (int i0 = tid; i0 < HS * cpr; i0 += 4 * NT) {
        f32x4 gv[4][2], vv[4][2];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * NT, h = i / cpr, ch = i % cpr;
            const bool live = i < HS * cpr && h < H;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                gv[u][q] = vv[u][q] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (live) {
                    gv[u][q] = *reinterpret_cast<const f32x4*>(gamma + ch * 8 + 4 * q);
                    vv[u][q] = *reinterpret_cast<const f32x4*>(v + (size_t)h * d + ch * 8 + 4 * q);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * NT, h = i / cpr, ch = i % cpr;
            if (i < HS * cpr) {
                const f32x4 a = gv[u][0] * vv[u][0], b2 = gv[u][1] * vv[u][1];
                *reinterpret_cast<uint4*>(img + h * rb + ch * 16) = make_uint4(pack2_t<f16_t>(a[0], a[1]), pack2_t<f16_t>(a[2], a[3]), pack2_t<f16_t>(b2[0], b2[1]), pack2_t<f16_t>(b2[2], b2[3]));
            }
        }
    }
    __syncthreads();
    for (int h = wave; h < HS; h += NT / 64) {
        float s = 0.f;
        for (int j = lane * 2; j < d; j += 128) {
            const f16x2 q = *reinterpret_cast<const f16x2*>(img + h * rb + j * 2);
            s += (float)q[0] + (float)q[1];
        }
        s = wave_sum(s);
        if (lane == 0) cg[h] = s;
    }
    __syncthreads();
}

// the sample's row statistics -> LDS (every later phase reads them several times)
__device__ __forceinline__ void stage_row_stats(float* mu_l, float* rs_l, const float* __restrict__ mean, const float* __restrict__ rstd, size_t row0, int L, int Lp) {
    for (int l = threadIdx.x; l < Lp; l += NT) {
        mu_l[l] = l < L ? mean[row0 + l] : 0.f;
        rs_l[l] = l < L ? rstd[row0 + l] : 0.f;
    }
}

// One 16-row tile of the sample: the rows' d / 32 chunks straight from global memory (B fragments: lane = row (lane & 15), k-group lane >> 4), against one or two
// gamma o vector images (A fragments from LDS) -> acc[i] = dot of row (lane & 15) with the image row 4 (lane >> 4) + i.
template <int KS, bool TWO>
__device__ __forceinline__ void score_tile(const f16_t* __restrict__ xrow, bool valid, const char* img0, const char* img1, int rb, int lane, f32x4& a0, f32x4& a1) {
    const int g = lane >> 4, r16 = lane & 15;
    Chunk xb[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        xb[ks].u = make_uint4(0, 0, 0, 0);
        if (valid) xb[ks].u = *reinterpret_cast<const uint4*>(xrow + (g + 4 * ks) * 8);
    }
    a0 = f32x4{0.f, 0.f, 0.f, 0.f};
    a1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        Chunk ga;
        ga.u = *reinterpret_cast<const uint4*>(img0 + r16 * rb + (g + 4 * ks) * 16);
        mma_chunk<f16_t>(a0, ga, xb[ks]);
        if constexpr (TWO) {
            Chunk gb;
            gb.u = *reinterpret_cast<const uint4*>(img1 + r16 * rb + (g + 4 * ks) * 16);
            mma_chunk<f16_t>(a1, gb, xb[ks]);
        }
    }
}

// sum over the sample's rows of w[l][h] * x_l ON THE MATRIX CORES: sum_l w[l][h] x_l[j] = (W^T X)[h][j] with k = the sample's rows.  (The first version summed on the
// vector pipe, a thread per column pair over all rows in eight-row groups with one group of loads ahead: a chain of 25 round trips, 25-30 us of the forward
// kernel's 52; groups of 16 rows, a ring two groups ahead and both tiles of pass 1 requested up front had all measured slower.)  Here 32 rows at a time are staged in LDS by
// coalesced 16-byte loads (the next tile's in registers meanwhile) and every wave takes the 16-column tiles w, w + 8, ...: A = the tile read TRANSPOSED
// (ds_read_b64_tr_b16: rows of the operand = columns of x, k = rows of x), B = the weights of head (lane & 15) for rows 4 g .. 4 g + 3 of the tile's two
// 16-row halves, taken from an fp16 table WT [16][Lp32] the softmax phase leaves (the weights are probabilities x rstd: fp16 keeps 11 bits of each,
// accumulation is f32) -> acc[t][i] = sum for head (lane & 15), column 16 (wave + 8 t) + 4 g + i.  Rows behind L carry weight 0 and are loaded from row L - 1.
constexpr int XT_PAD = 32;      // bytes behind a tile row (rows 8 banks apart: the transposing reads of 16 rows x 32 bytes)
typedef __attribute__((ext_vector_type(4))) short short4s;
template <int NJ, int NC>       // NJ = column tiles per wave (d / 128 rounded up), NC = 16-byte chunks a thread stages per tile (d / 128 rounded up)
__device__ __forceinline__ void weighted_row_sums_mfma(f32x4 (&acc)[NJ], const f16_t* __restrict__ xs, int ldx, int L, int Lp, int d, const char* wt, char* xt) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, r16 = lane & 15;
    const int cpr = d / 8, rsx = d * 2 + XT_PAD, Lp32 = (Lp + 31) & ~31, ntile = d / 16;
    uint4 stg[NC];
    auto load_tile = [&](int l0) {
#pragma unroll
        for (int u = 0; u < NC; ++u) {
            const int c = tid + u * NT;
            stg[u] = make_uint4(0, 0, 0, 0);
            if (c < 32 * cpr) stg[u] = *reinterpret_cast<const uint4*>(xs + (size_t)min(l0 + c / cpr, L - 1) * ldx + (c % cpr) * 8);
        }
    };
    load_tile(0);
    for (int l0 = 0; l0 < Lp; l0 += 32) {
        __syncthreads();      // the previous tile's reads are done (first round: the weights table is complete)
#pragma unroll
        for (int u = 0; u < NC; ++u) {
            const int c = tid + u * NT;
            if (c < 32 * cpr) *reinterpret_cast<uint4*>(xt + (c / cpr) * rsx + (c % cpr) * 16) = stg[u];
        }
        __syncthreads();
        if (l0 + 32 < Lp) load_tile(l0 + 32);
        Chunk wb;      // B: head r16, rows l0 + 4 g .. + 3 and l0 + 16 + 4 g .. + 3
        {
            const uint2 lo = *reinterpret_cast<const uint2*>(wt + (r16 * Lp32 + l0 + 4 * g) * 2), hi = *reinterpret_cast<const uint2*>(wt + (r16 * Lp32 + l0 + 16 + 4 * g) * 2);
            wb.u = make_uint4(lo.x, lo.y, hi.x, hi.y);
        }
        const char* tp = xt + (4 * g + (r16 >> 2)) * rsx + (r16 & 3) * 8;
#pragma unroll
        for (int t = 0; t < NJ; ++t) {
            const int jt = wave + 8 * t;
            if (jt < ntile) {
                const short4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4s*)(tp + jt * 32));
                const short4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4s*)(tp + 16 * rsx + jt * 32));
                const uint2 lo2 = __builtin_bit_cast(uint2, lo), hi2 = __builtin_bit_cast(uint2, hi);
                Chunk xa;
                xa.u = make_uint4(lo2.x, lo2.y, hi2.x, hi2.y);
                mma_chunk<f16_t>(acc[t], xa, wb);
            }
        }
    }
}

// forward LDS map: image [16 rb] | S [Lp][16] f32 | mu [Lp] | rs [Lp] | cg [16] | zmu [16] | WT [16][Lp32] fp16 | x tile [32][2 d + 32]
template <int KS, int NH>
__global__ __launch_bounds__(NT) void spool_fwd_kernel(SpoolArgs A) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int L = A.L, H = A.H, d = A.d, Lp = (L + 15) & ~15, rb = grow_bytes(d);
    char* img = smem;
    float* S = reinterpret_cast<float*>(smem + HS * rb);
    float* mu_l = S + Lp * HS;
    float* rs_l = mu_l + Lp;
    float* cg = rs_l + Lp;
    float* zmu = cg + HS;
    const int Lp32 = (Lp + 31) & ~31;
    char* wt = reinterpret_cast<char*>(zmu + HS);
    char* xt = wt + HS * Lp32 * 2;
    const size_t row0 = (size_t)b * L;
    stage_row_stats(mu_l, rs_l, A.mean, A.rstd, row0, L, Lp);
    fill_gamma_image(img, cg, A.qt + (size_t)b * H * d, A.gamma, H, d);
    // ---- pass 1: scores s[l][h] = r_l (x_l . g_h - mu_l cg_h)
    for (int rt = wave; rt * 16 < Lp; rt += NT / 64) {
        const int r = rt * 16 + (lane & 15);
        const bool valid = r < L;
        f32x4 a0, a1;
        score_tile<KS, false>(A.x + (row0 + (valid ? r : 0)) * A.ldx, valid, img, img, rb, lane, a0, a1);
        const float mu = mu_l[r], rs = rs_l[r];
        const int h0 = 4 * (lane >> 4);
        f32x4 s;
#pragma unroll
        for (int i = 0; i < 4; ++i) s[i] = valid ? rs * (a0[i] - mu * cg[h0 + i]) : -INFINITY;
        *reinterpret_cast<f32x4*>(S + r * HS + h0) = s;
    }
    __syncthreads();
    // ---- softmax over the rows, two heads per wave; WT <- fp16 (p_l r_l) (the weight of row l in the sum of RAW rows), zmu = sum_l of the ROUNDED weights x mu_l
    // (the mean term of the folded LayerNorm then cancels against the products exactly, whatever the rows' offset)
    for (int h = wave; h < HS; h += NT / 64) {
        f16_t* wrow = reinterpret_cast<f16_t*>(wt) + h * Lp32;
        if (h >= H) {
            for (int l = lane; l < Lp32; l += 64) wrow[l] = (f16_t)0.f;
            continue;
        }
        float m = -INFINITY;
        for (int l = lane; l < L; l += 64) m = fmaxf(m, S[l * HS + h]);
        m = wave_max(m);
        float z = 0.f;
        for (int l = lane; l < L; l += 64) z += __expf(S[l * HS + h] - m);
        z = wave_sum(z);
        const float lse = m + __logf(z);
        float zm = 0.f;
        for (int l = lane; l < Lp32; l += 64) {
            f16_t w = (f16_t)0.f;
            if (l < L) {
                w = (f16_t)(__expf(S[l * HS + h] - lse) * rs_l[l]);
                zm += (float)w * mu_l[l];
            }
            wrow[l] = w;
        }
        zm = wave_sum(zm);
        if (lane == 0) { zmu[h] = zm; A.lse[(size_t)b * H + h] = lse; }
    }
    // ---- pass 2: hbar_h = gamma o (sum_l w_lh x_l - zmu_h) + beta on the matrix cores; a lane ends with head (lane & 15), four adjacent columns per tile
    {
        constexpr int NJ = KS / 4;
        f32x4 acc[NJ];
#pragma unroll
        for (int t = 0; t < NJ; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        weighted_row_sums_mfma<NJ, NJ>(acc, A.x + row0 * A.ldx, A.ldx, L, Lp, d, wt, xt);
        const int h = lane & 15, g = lane >> 4;
        if (h < H) {
            const float zm = zmu[h];
#pragma unroll
            for (int t = 0; t < NJ; ++t) {
                const int j = 16 * (wave + 8 * t) + 4 * g;
                if (j < d) {
                    const f32x4 gm = *reinterpret_cast<const f32x4*>(A.gamma + j), bt = *reinterpret_cast<const f32x4*>(A.beta + j);
                    *reinterpret_cast<f32x4*>(A.hbar + ((size_t)b * H + h) * d + j) = gm * (acc[t] - zm) + bt;
                }
            }
        }
    }
}

// backward LDS map: R0 = max(2 images, W2 [d][80 B], x tile [32][2 d + 32]) | P [Lp][16] f32 | D [Lp][16] f32 (dp) | PD [Lp][32] fp16 (p | ds) | mu [Lp] | rs [Lp] |
// cg cgd zmu | WT [16][Lp32] fp16 (e = ds r)
constexpr int W2B = 80;      // bytes of a W2 row: 32 halves (dhbar of the 16 head slots | qt of the 16 head slots) + 16: rows 80 bytes apart are conflict-free for the fragment reads
template <int KS, int NH>
__global__ __launch_bounds__(NT) void spool_bwd_kernel(SpoolArgs A) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int L = A.L, H = A.H, d = A.d, Lp = (L + 15) & ~15, rb = grow_bytes(d);
    const int r0b = max(max(2 * HS * rb, d * W2B), 32 * (2 * d + XT_PAD));
    char* img0 = smem;
    char* img1 = smem + HS * rb;
    char* W2 = smem;                                                // pass 2: over the images
    float* P = reinterpret_cast<float*>(smem + r0b);              // p_l
    float* D = P + Lp * HS;                                       // dp_l, then e_l = ds_l r_l
    char* PD = reinterpret_cast<char*>(D + Lp * HS);              // [Lp][32] fp16: p_l | ds_l
    float* mu_l = reinterpret_cast<float*>(PD + Lp * 64);
    float* rs_l = mu_l + Lp;
    float* cg = rs_l + Lp;
    float* cgd = cg + HS;
    float* zmu = cgd + HS;
    const int Lp32 = (Lp + 31) & ~31;
    char* wt = reinterpret_cast<char*>(zmu + HS);
    char* xt = smem;                                                // pass 2b: over W2
    const size_t row0 = (size_t)b * L;
    const float* qtb = A.qt + (size_t)b * H * d;
    const float* dhb = A.dhbar + (size_t)b * H * d;
    stage_row_stats(mu_l, rs_l, A.mean, A.rstd, row0, L, Lp);
    fill_gamma_image(img0, cg, qtb, A.gamma, H, d);
    fill_gamma_image(img1, cgd, dhb, A.gamma, H, d);
    // ---- pass 1: p_l = exp(s_l - lse), dp_l = r_l (x_l . gd_h - mu_l cgd_h)   (+ beta . dhbar, constant over l: it cancels in ds)
    for (int rt = wave; rt * 16 < Lp; rt += NT / 64) {
        const int r = rt * 16 + (lane & 15);
        const bool valid = r < L;
        f32x4 a0, a1;
        score_tile<KS, true>(A.x + (row0 + (valid ? r : 0)) * A.ldx, valid, img0, img1, rb, lane, a0, a1);
        const float mu = mu_l[r], rs = rs_l[r];
        const int h0 = 4 * (lane >> 4);
        f32x4 p, dp;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool live = valid && h0 + i < H;
            p[i] = live ? __expf(rs * (a0[i] - mu * cg[h0 + i]) - A.lse[(size_t)b * H + h0 + i]) : 0.f;
            dp[i] = live ? rs * (a1[i] - mu * cgd[h0 + i]) : 0.f;
        }
        *reinterpret_cast<f32x4*>(P + r * HS + h0) = p;
        *reinterpret_cast<f32x4*>(D + r * HS + h0) = dp;
    }
    __syncthreads();      // (the images are dead from here on: W2 goes over them)
    // ---- ds_l = p_l (dp_l - sum p dp); WT <- fp16 (e_l = ds_l r_l); PD <- fp16 (p_l | ds_l); zmu = sum_l of the ROUNDED e_l x mu_l.  Two heads per wave.
    for (int h = wave; h < HS; h += NT / 64) {
        f16_t* wrow = reinterpret_cast<f16_t*>(wt) + h * Lp32;
        float dl = 0.f;
        for (int l = lane; l < L; l += 64) dl += P[l * HS + h] * D[l * HS + h];
        dl = wave_sum(dl);
        float zm = 0.f;
        for (int l = lane; l < Lp32; l += 64) {
            f16_t e = (f16_t)0.f;
            if (l < Lp) {
                const float p = P[l * HS + h];
                float ds = 0.f;
                if (l < L && h < H) {
                    ds = p * (D[l * HS + h] - dl);
                    e = (f16_t)(ds * rs_l[l]);
                    zm += (float)e * mu_l[l];
                }
                *reinterpret_cast<f16_t*>(PD + l * 64 + h * 2) = (f16_t)p;
                *reinterpret_cast<f16_t*>(PD + l * 64 + 32 + h * 2) = (f16_t)ds;
            }
            wrow[l] = e;
        }
        zm = wave_sum(zm);
        if (lane == 0) zmu[h] = zm;
    }
    // W2[j] = fp16 (dhbar[0..15][j] | qt[0..15][j]) (zero behind H): a thread makes the two 16-byte halves of four columns per round
    for (int i = tid; i < d * 4; i += NT) {
        const int jj = i >> 2, part = i & 3;      // part: 0, 1 = dhbar heads 0-7, 8-15; 2, 3 = qt heads 0-7, 8-15
        const float* src = (part < 2 ? dhb : qtb) + jj;
        const int hb = (part & 1) * 8;
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = hb + q < H ? src[(size_t)(hb + q) * d] : 0.f;
        *reinterpret_cast<uint4*>(W2 + jj * W2B + part * 16) = make_uint4(pack2_t<f16_t>(v[0], v[1]), pack2_t<f16_t>(v[2], v[3]), pack2_t<f16_t>(v[4], v[5]), pack2_t<f16_t>(v[6], v[7]));
    }
    __syncthreads();
    // ---- pass 2a: d LN1(x_l)[j] = sum_h (p_lh dhbar_h[j] + ds_lh qt_h[j]) = ([p | ds] x W2^T)[l][j]: ONE MFMA (k = 32) per 16 x 16 tile.  A wave takes row tiles
    // w, w + 8, ... and walks the columns: acc[i] = column 16 jt + 4 g + i of row l = lane & 15 — four adjacent columns per lane, 8-byte stores
    {
        const int g = lane >> 4, r16 = lane & 15;
        for (int rt = wave; rt * 16 < Lp; rt += NT / 64) {
            const int l = rt * 16 + r16;
            Chunk pd;
            pd.u = *reinterpret_cast<const uint4*>(PD + l * 64 + g * 16);
            bf16_t* orow = A.dh + (row0 + l) * A.lddh + 4 * g;
            for (int jt = 0; jt < d / 16; jt += 4) {
                Chunk w2[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) w2[u].u = *reinterpret_cast<const uint4*>(W2 + ((jt + u) * 16 + r16) * W2B + g * 16);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
                    mma_chunk<f16_t>(acc, w2[u], pd);
                    if (l < L) *reinterpret_cast<uint2*>(orow + (jt + u) * 16) = make_uint2(pack2_t<bf16_t>(acc[0], acc[1]), pack2_t<bf16_t>(acc[2], acc[3]));
                }
            }
        }
    }
    // ---- pass 2b: dqt_h = gamma o (sum_l e_lh x_l - zmu_h) on the matrix cores (the x tiles go over W2: the function's first barrier is behind pass 2a)
    {
        constexpr int NJ = KS / 4;
        f32x4 acc[NJ];
#pragma unroll
        for (int t = 0; t < NJ; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        weighted_row_sums_mfma<NJ, NJ>(acc, A.x + row0 * A.ldx, A.ldx, L, Lp, d, wt, xt);
        const int h = lane & 15, g = lane >> 4;
        if (h < H) {
            const float zm = zmu[h];
#pragma unroll
            for (int t = 0; t < NJ; ++t) {
                const int j = 16 * (wave + 8 * t) + 4 * g;
                if (j < d) *reinterpret_cast<f32x4*>(A.dqt + ((size_t)b * H + h) * d + j) = *reinterpret_cast<const f32x4*>(A.gamma + j) * (acc[t] - zm);
            }
        }
    }
}

size_t spool_lds(int L, int d, bool bwd) {
    const size_t Lp = (size_t)((L + 15) & ~15), rb = (size_t)d * 2 + 16;
    const size_t Lp32 = (Lp + 31) & ~(size_t)31, wt = HS * Lp32 * 2, xt = 32 * ((size_t)d * 2 + XT_PAD);      // the weights table and the x tile of the MFMA row sums
    if (!bwd) return HS * rb + Lp * HS * 4 + 2 * Lp * 4 + 2 * HS * 4 + wt + xt;
    return std::max(std::max(2 * HS * rb, (size_t)d * W2B), xt) + 2 * Lp * HS * 4 + Lp * 64 + 2 * Lp * 4 + 3 * HS * 4 + wt;
}

bool spool_ok(int L, int H, int d) { return H >= 1 && H <= HS && d == HD * H && d % 32 == 0 && d <= 2 * NT && L >= 1 && L <= 288 && spool_lds(L, d, true) <= 160 * 1024; }

template <bool BWD>
int spool_launch(const SpoolArgs& A, hipStream_t s) {
    const int KS = A.d / 32, NH = (A.H + 3) & ~3;
    const size_t lds = spool_lds(A.L, A.d, BWD);
#define SP(K, N)                                                                                                                   \
    if (KS == K && NH == N) {                                                                                                      \
        static LdsOnce once;                                                                                                       \
        const void* fn = BWD ? (const void*)spool_bwd_kernel<K, N> : (const void*)spool_fwd_kernel<K, N>;                          \
        if (int e = lpi_ensure_lds(once, fn, 160 * 1024)) return e;                                                                \
        if (BWD) LPI_LAUNCH((spool_bwd_kernel<K, N>), dim3(A.B), dim3(NT), lds, s, A);                                              \
        else LPI_LAUNCH((spool_fwd_kernel<K, N>), dim3(A.B), dim3(NT), lds, s, A);                                                  \
        LPI_CHECK_LAST();                                                                                                          \
        return 0;                                                                                                                  \
    }
    SP(24, 12) SP(32, 16) SP(4, 4) SP(8, 4) SP(16, 8)      // ViT-B/16, ViT-L/14, the tiny configurations (d = 128, 256), d = 512
#undef SP
    return LPI_ENOSYS;
}

template <typename T>
int headw_t(int B, int H, int d, const void* in, int ldin, const void* WT, int ldwt, int col0, float scale, float* out, hipStream_t s) {
    LPI_LAUNCH((headw_t_kernel<T>), dim3(H, (B + 15) / 16), dim3(256), 0, s, B, H, d, (const T*)in, ldin, (const T*)WT, ldwt, col0, scale, out);
    LPI_CHECK_LAST();
    return 0;
}
template <typename T, typename TO>
int headw_n(int B, int H, int d, const float* in, const void* W, int ldw, int row0, const float* bias, float scale, void* out, int ldo, hipStream_t s) {
    LPI_LAUNCH((headw_n_kernel<T, TO>), dim3(H, (B + 15) / 16), dim3(256), 0, s, B, H, d, in, (const T*)W, ldw, row0, bias, scale, (TO*)out, ldo);
    LPI_CHECK_LAST();
    return 0;
}

}  // namespace

// 1 if lpi_spool_attn_fwd / _bwd take this shape (H <= 16 heads of 64, L <= 288 uniform rows per sample, the LDS of the backward fits)
extern "C" int lpi_spool_attn_supported(int L, int H, int d) { return spool_ok(L, H, d) ? 1 : 0; }

// Forward of the last block's attention for the pooled query only, from the residual stream itself (see the head of this file).
//   w_dtype: LPI_BF16 / LPI_F16 = type of q [B, ldq], of Wqkv [3 d, ldw] (in_proj: rows d .. 2d = W_k, 2d .. 3d = W_v), of its transpose WqkvT [d, ldwt] and of
//   ctx [B, ldctx]; bqkv f32 [3 d]
//   x: fp16 [B L, ldx] = the block's input rows; mean / rstd: ln_1's statistics of those rows (f32 [B L]); gamma / beta: ln_1's affine (f32 [d])
//   scratch: f32 [2 B H d] (qt | hbar: qt is read again by the backward); lse: f32 [B, H] out
extern "C" int lpi_spool_attn_fwd(int w_dtype, int B, int L, int H, const void* q, int ldq, const void* Wqkv, int ldw, const void* WqkvT, int ldwt, const float* bqkv,
                                  const void* x, int ldx, const float* mean, const float* rstd, const float* gamma, const float* beta, float* scratch, float* lse,
                                  void* ctx, int ldctx, void* stream) {
    const int d = HD * H;
    if (!q || !Wqkv || !WqkvT || !bqkv || !x || !mean || !rstd || !gamma || !beta || !scratch || !lse || !ctx || B <= 0) return LPI_EINVAL;
    if ((w_dtype != LPI_BF16 && w_dtype != LPI_F16) || !spool_ok(L, H, d) || ldq < d || ldw < d || ldwt < 3 * d || ldx < d || ldctx < d || ((ldx | ldw | ldwt | ldq) & 7) ||
        (((uintptr_t)x | (uintptr_t)Wqkv | (uintptr_t)WqkvT | (uintptr_t)q) & 15))
        return LPI_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    float* qt = scratch;
    float* hbar = scratch + (size_t)B * H * d;
    int e = w_dtype == LPI_BF16 ? headw_t<bf16_t>(B, H, d, q, ldq, WqkvT, ldwt, d, 0.125f, qt, s) : headw_t<f16_t>(B, H, d, q, ldq, WqkvT, ldwt, d, 0.125f, qt, s);
    if (e) return e;
    SpoolArgs A{B, L, H, d, (const f16_t*)x, ldx, mean, rstd, gamma, beta, qt, hbar, lse, nullptr, nullptr, 0, nullptr};
    if ((e = spool_launch<false>(A, s))) return e;
    return w_dtype == LPI_BF16 ? headw_n<bf16_t, bf16_t>(B, H, d, hbar, Wqkv, ldw, 2 * d, bqkv, 1.0f, ctx, ldctx, s)
                               : headw_n<f16_t, f16_t>(B, H, d, hbar, Wqkv, ldw, 2 * d, bqkv, 1.0f, ctx, ldctx, s);
}

// Backward: dctx bf16 [B, lddctx] -> dq bf16 [B, lddq] and dh bf16 [B L, lddh] = the gradient w.r.t. LN1(x_l) of EVERY row (what the in_proj dgrad of K and V
// used to deliver; the pooled rows' query path is added by the caller as before).  The backward's operands are bf16 (gradients do not fit fp16's range):
// Wqkv / WqkvT here are the BF16 weight and its transpose whatever the forward's type was.  scratch: the forward's (qt is read), + f32 [2 B H d] behind it.
extern "C" int lpi_spool_attn_bwd(int B, int L, int H, const void* Wqkv, int ldw, const void* WqkvT, int ldwt, const void* x, int ldx, const float* mean,
                                  const float* rstd, const float* gamma, float* scratch, const float* lse, const void* dctx, int lddctx, void* dq, int lddq, void* dh,
                                  int lddh, void* stream) {
    const int d = HD * H;
    if (!Wqkv || !WqkvT || !x || !mean || !rstd || !gamma || !scratch || !lse || !dctx || !dq || !dh || B <= 0) return LPI_EINVAL;
    if (!spool_ok(L, H, d) || ldw < d || ldwt < 3 * d || ldx < d || lddctx < d || lddq < d || lddh < d || ((ldx | ldw | ldwt | lddctx) & 7) || (lddh & 3) ||
        (((uintptr_t)x | (uintptr_t)Wqkv | (uintptr_t)WqkvT | (uintptr_t)dctx) & 15) || ((uintptr_t)dh & 7))
        return LPI_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const size_t n = (size_t)B * H * d;
    float* qt = scratch;
    float* dhbar = scratch + 2 * n;
    float* dqt = scratch + 3 * n;
    if (int e = headw_t<bf16_t>(B, H, d, dctx, lddctx, WqkvT, ldwt, 2 * d, 1.0f, dhbar, s)) return e;
    SpoolArgs A{B, L, H, d, (const f16_t*)x, ldx, mean, rstd, gamma, nullptr, qt, nullptr, const_cast<float*>(lse), dhbar, (bf16_t*)dh, lddh, dqt};
    if (int e = spool_launch<true>(A, s)) return e;
    return headw_n<bf16_t, bf16_t>(B, H, d, dqt, Wqkv, ldw, d, nullptr, 0.125f, dq, lddq, s);
}
