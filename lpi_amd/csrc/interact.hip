// Low-rank cross-modal interaction of the prompt rows (LPI's "InteractModule"), forward and backward, f32.
//
// replaces: InteractModule.forward (grounding/maskrcnn_benchmark/modeling/bert/modeling_bert.py:616-651; constructed :558-590) and its autograd
// backward.  Per layer l the module maps the visual prompt rows into the textual space and back with rank-r CP weights that include a bias row:
//     M_v2t[i, j] = mean_r d1v[l, r] d2v[i, r] d3v[j, r]          i in [0, Dv] (row Dv = bias), j in [0, Dt)
//     t_new = v M_v2t[:Dv] + M_v2t[Dv],   v_new = t M_t2v[:Dt] + M_t2v[Dt]          (both from the ORIGINAL v, t)
//     v_out = LayerNorm_v((1 - a) v + a v_new),   t_out = LayerNorm_t((1 - a) t + a t_new)          a = 0.1, eps 1e-5
// The reference materialises the [layer_num, D_in + 1, D_out, r] product tensor on every call (:617-621, :631-635); here the rank-r form is kept:
// z_r = x . d2[:, r] (r dot products per row), w_r = (d1[l, r] / r)(z_r + d2[D_in, r]), y = sum_r w_r d3[:, r] — 2 r (D_in + D_out) MACs per row instead
// of D_in D_out, nothing materialised.  Rows are the prompt rows of the batch ([bs, P, D] flattened: :780-790), a few thousand at most: the op is
// latency bound, so one workgroup walks a chunk of rows with every thread owning up to four columns of either width, sums in a fixed order (bitwise
// reproducible, no atomics); the backward keeps its parameter-gradient partials in registers over the chunk, writes them per workgroup and a second
// kernel adds the workgroups' partials in order.
#include "common.h"

namespace {

constexpr int TPB = 256;                 // threads per workgroup
constexpr int MAXC = 4;                  // columns per thread: widths up to 1024
constexpr int MAXR = 8;                  // CP rank

struct Dir {                             // one direction: `in` rows [N, Din] -> contribution to the `out` side [N, Dout]
    const float* d1; const float* d2; const float* d3;      // [Lyr, R], [Din + 1, R], [Dout, R]
    const float* gamma; const float* beta;                  // LayerNorm of the OUT side
    const float* xin; int ldin;                             // source rows
    const float* xout; int ldout;                           // the out side's own rows (mixed in with weight 1 - a)
    float* y; int ldy;                                      // result rows [N, Dout]
    float* mean; float* rstd;                               // [N] (saved for the backward)
    int Din, Dout;
};

// sum of R per-thread values over the workgroup, every thread gets the totals; red: [4][MAXR] floats of LDS.  Fixed order.
template <int R>
__device__ __forceinline__ void block_sum(float (&v)[R], float* red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int r = 0; r < R; ++r) v[r] = wave_sum(v[r]);
    __syncthreads();            // the previous use of red is over
    if (lane == 0) {
#pragma unroll
        for (int r = 0; r < R; ++r) red[wave * MAXR + r] = v[r];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < R; ++r) v[r] = (red[r] + red[MAXR + r]) + (red[2 * MAXR + r] + red[3 * MAXR + r]);
}

// one row, one direction, forward: returns u_hat (normalised mixed row) columns of this thread in uh[], writes y
template <int R>
__device__ __forceinline__ void dir_fwd_row(const Dir& D, int n, int layer, float mix, float eps, float* red, float (&w)[R], float (&zb)[R],
                                            float (&uh)[MAXC], float& rstd_out, bool store) {
    const int t = threadIdx.x;
    float z[R];
#pragma unroll
    for (int r = 0; r < R; ++r) z[r] = 0.f;
#pragma unroll
    for (int k = 0; k < MAXC; ++k) {
        const int i = t + k * TPB;
        if (i < D.Din) {
            const float x = D.xin[(size_t)n * D.ldin + i];
#pragma unroll
            for (int r = 0; r < R; ++r) z[r] = fmaf(x, D.d2[i * R + r], z[r]);
        }
    }
    block_sum<R>(z, red);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        zb[r] = z[r] + D.d2[D.Din * R + r];
        w[r] = (D.d1[layer * R + r] * (1.0f / R)) * zb[r];
    }
    float u[MAXC], s[1] = {0.f};
#pragma unroll
    for (int k = 0; k < MAXC; ++k) {
        const int j = t + k * TPB;
        u[k] = 0.f;
        if (j < D.Dout) {
            float y = 0.f;
#pragma unroll
            for (int r = 0; r < R; ++r) y = fmaf(w[r], D.d3[j * R + r], y);
            u[k] = (1.0f - mix) * D.xout[(size_t)n * D.ldout + j] + mix * y;
            s[0] += u[k];
        }
    }
    block_sum<1>(s, red);
    const float mean = s[0] / (float)D.Dout;
    float q[1] = {0.f};
#pragma unroll
    for (int k = 0; k < MAXC; ++k) {
        const int j = t + k * TPB;
        if (j < D.Dout) { const float c = u[k] - mean; q[0] = fmaf(c, c, q[0]); }
    }
    block_sum<1>(q, red);
    const float rstd = rsqrtf(q[0] / (float)D.Dout + eps);
    rstd_out = rstd;
#pragma unroll
    for (int k = 0; k < MAXC; ++k) {
        const int j = t + k * TPB;
        uh[k] = 0.f;
        if (j < D.Dout) {
            uh[k] = (u[k] - mean) * rstd;
            if (store) D.y[(size_t)n * D.ldy + j] = fmaf(uh[k], D.gamma[j], D.beta[j]);
        }
    }
    if (store && t == 0) { D.mean[n] = mean; D.rstd[n] = rstd; }
}

template <int R>
__global__ __launch_bounds__(TPB) void interact_fwd_kernel(int N, int rows_per_wg, int layer, float mix, float eps, Dir v2t, Dir t2v) {
    __shared__ float red[4 * MAXR];
    const int n0 = blockIdx.x * rows_per_wg;
    for (int n = n0; n < min(N, n0 + rows_per_wg); ++n) {
        float w[R], zb[R], uh[MAXC], rs;
        dir_fwd_row<R>(v2t, n, layer, mix, eps, red, w, zb, uh, rs, true);
        dir_fwd_row<R>(t2v, n, layer, mix, eps, red, w, zb, uh, rs, true);
    }
}

// Partial parameter gradients of one direction, per thread: columns t + 256 k
template <int R>
struct Part {
    float dd3[MAXC][R];      // d3 [Dout, R]
    float dd2[MAXC][R];      // d2 [Din, R] (rows < Din)
    float dgam[MAXC], dbet[MAXC];
    float dc[R], db[R];      // d1[layer] (x R) and the bias row d2[Din]: identical in every thread (block sums)
};

// layout of one direction's partial vector: dd3 [Dout R] | dd2 [(Din + 1) R] | dd1 [R] | dgamma [Dout] | dbeta [Dout]
__host__ __device__ inline int part_len(int Din, int Dout, int R) { return Dout * R + (Din + 1) * R + R + 2 * Dout; }

template <int R>
__device__ __forceinline__ void dir_bwd_row(const Dir& D, int n, int layer, float mix, float eps, float* red, const float* __restrict__ g, int ldg,
                                            float* __restrict__ dxin, int lddin, float* __restrict__ dxout, int lddout, Part<R>& P) {
    const int t = threadIdx.x;
    float w[R], zb[R], uh[MAXC], rstd;
    dir_fwd_row<R>(D, n, layer, mix, eps, red, w, zb, uh, rstd, false);      // recompute z -> w, u_hat (the forward saves nothing the backward needs)
    // LayerNorm backward: du = rstd (gg - mean(gg) - u_hat mean(gg u_hat)),  gg = g gamma
    float gg[MAXC], s[2] = {0.f, 0.f};
#pragma unroll
    for (int k = 0; k < MAXC; ++k) {
        const int j = t + k * TPB;
        gg[k] = 0.f;
        if (j < D.Dout) {
            const float gv = g[(size_t)n * ldg + j];
            P.dgam[k] = fmaf(gv, uh[k], P.dgam[k]);
            P.dbet[k] += gv;
            gg[k] = gv * D.gamma[j];
            s[0] += gg[k];
            s[1] = fmaf(gg[k], uh[k], s[1]);
        }
    }
    block_sum<2>(s, red);
    const float m0 = s[0] / (float)D.Dout, m1 = s[1] / (float)D.Dout;
    float dw[R];
#pragma unroll
    for (int r = 0; r < R; ++r) dw[r] = 0.f;
#pragma unroll
    for (int k = 0; k < MAXC; ++k) {
        const int j = t + k * TPB;
        if (j < D.Dout) {
            const float du = rstd * (gg[k] - m0 - uh[k] * m1);
            // the out side's own row: weight (1 - a); ACCUMULATED (the other direction's source-row gradient lands in the same buffer)
            dxout[(size_t)n * lddout + j] += (1.0f - mix) * du;
            const float dy = mix * du;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                dw[r] = fmaf(dy, D.d3[j * R + r], dw[r]);
                P.dd3[k][r] = fmaf(dy, w[r], P.dd3[k][r]);
            }
        }
    }
    block_sum<R>(dw, red);
    // w_r = c_r (z_r + b_r): c_r = d1[l, r] / R
    float dz[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const float c = D.d1[layer * R + r] * (1.0f / R);
        P.dc[r] = fmaf(dw[r], zb[r], P.dc[r]);
        dz[r] = dw[r] * c;
        P.db[r] += dz[r];
    }
#pragma unroll
    for (int k = 0; k < MAXC; ++k) {
        const int i = t + k * TPB;
        if (i < D.Din) {
            const float x = D.xin[(size_t)n * D.ldin + i];
            float dx = 0.f;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                dx = fmaf(dz[r], D.d2[i * R + r], dx);
                P.dd2[k][r] = fmaf(x, dz[r], P.dd2[k][r]);
            }
            dxin[(size_t)n * lddin + i] += dx;
        }
    }
}

template <int R>
__device__ __forceinline__ void part_zero(Part<R>& P) {
#pragma unroll
    for (int k = 0; k < MAXC; ++k) {
        P.dgam[k] = 0.f; P.dbet[k] = 0.f;
#pragma unroll
        for (int r = 0; r < R; ++r) { P.dd3[k][r] = 0.f; P.dd2[k][r] = 0.f; }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) { P.dc[r] = 0.f; P.db[r] = 0.f; }
}
template <int R>
__device__ __forceinline__ void part_store(const Part<R>& P, const Dir& D, float* out) {
    const int t = threadIdx.x;
    float* o3 = out;
    float* o2 = o3 + D.Dout * R;
    float* o1 = o2 + (D.Din + 1) * R;
    float* og = o1 + R;
    float* ob = og + D.Dout;
#pragma unroll
    for (int k = 0; k < MAXC; ++k) {
        const int j = t + k * TPB;
        if (j < D.Dout) {
            og[j] = P.dgam[k]; ob[j] = P.dbet[k];
#pragma unroll
            for (int r = 0; r < R; ++r) o3[j * R + r] = P.dd3[k][r];
        }
        if (j < D.Din) {
#pragma unroll
            for (int r = 0; r < R; ++r) o2[j * R + r] = P.dd2[k][r];
        }
    }
    if (t == 0) {
#pragma unroll
        for (int r = 0; r < R; ++r) { o2[D.Din * R + r] = P.db[r]; o1[r] = P.dc[r] * (1.0f / R); }
    }
}

// dxv / dxt must be ZERO on entry (lpi_interact_bwd clears them): each receives the (1 - a) path of its own LayerNorm and the source-row path of
// the other direction, added by the same thread in program order.
template <int R>
__global__ __launch_bounds__(TPB) void interact_bwd_kernel(int N, int rows_per_wg, int layer, float mix, float eps, Dir v2t, Dir t2v, const float* gv, int ldgv,
                                                          const float* gt, int ldgt, float* dxv, int lddv, float* dxt, int lddt, float* partial, int plen_v2t,
                                                          int plen_t2v) {
    __shared__ float red[4 * MAXR];
    Part<R> Pa, Pb;
    part_zero<R>(Pa);
    part_zero<R>(Pb);
    const int n0 = blockIdx.x * rows_per_wg;
    for (int n = n0; n < min(N, n0 + rows_per_wg); ++n) {
        // v2t: source rows = visual, out side = textual (gradient gt); t2v: the other way round.  The two directions touch a column of dxv / dxt
        // from DIFFERENT threads in general (widths differ), so a workgroup barrier separates them.
        dir_bwd_row<R>(v2t, n, layer, mix, eps, red, gt, ldgt, dxv, lddv, dxt, lddt, Pa);
        __syncthreads();
        dir_bwd_row<R>(t2v, n, layer, mix, eps, red, gv, ldgv, dxt, lddt, dxv, lddv, Pb);
        __syncthreads();
    }
    float* out = partial + (size_t)blockIdx.x * (plen_v2t + plen_t2v);
    part_store<R>(Pa, v2t, out);
    part_store<R>(Pb, t2v, out + plen_v2t);
}

// out[i] = sum over workgroups g (in order) of partial[g][i]
__global__ void interact_reduce_kernel(int G, int len, const float* __restrict__ partial, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= len) return;
    float s = 0.f;
    for (int g = 0; g < G; ++g) s += partial[(size_t)g * len + i];
    out[i] = s;
}

inline bool bad_dims(int N, int Dv, int Dt, int R, int Lyr, int layer) {
    return N <= 0 || Dv <= 0 || Dt <= 0 || Dv > MAXC * TPB || Dt > MAXC * TPB || R <= 0 || R > MAXR || Lyr <= 0 || layer < 0 || layer >= Lyr;
}
inline int rows_per_wg(int N) { return (N + 255) / 256 < 1 ? 1 : (N + 255) / 256; }

#define LPI_R_SWITCH(R, BODY)                   \
    switch (R) {                                \
        case 1: { constexpr int RR = 1; BODY; } break; \
        case 2: { constexpr int RR = 2; BODY; } break; \
        case 3: { constexpr int RR = 3; BODY; } break; \
        case 4: { constexpr int RR = 4; BODY; } break; \
        case 5: { constexpr int RR = 5; BODY; } break; \
        case 6: { constexpr int RR = 6; BODY; } break; \
        case 7: { constexpr int RR = 7; BODY; } break; \
        default: { constexpr int RR = 8; BODY; } break; \
    }

}  // namespace

extern "C" int lpi_interact_workspace_floats(int N, int Dv, int Dt, int R) {
    if (N <= 0 || Dv <= 0 || Dt <= 0 || R <= 0) return LPI_EINVAL;
    const int rp = rows_per_wg(N), G = (N + rp - 1) / rp;
    return G * (part_len(Dv, Dt, R) + part_len(Dt, Dv, R));
}

extern "C" int lpi_interact_fwd(int N, int Dv, int Dt, int R, int Lyr, int layer, const float* xv, int ldv, const float* xt, int ldt,
                                const float* d1_v2t, const float* d2_v2t, const float* d3_v2t, const float* d1_t2v, const float* d2_t2v,
                                const float* d3_t2v, const float* gamma_v, const float* beta_v, const float* gamma_t, const float* beta_t, float mix,
                                float eps, float* out_v, int ldov, float* out_t, int ldot, float* stat, void* stream) {
    if (bad_dims(N, Dv, Dt, R, Lyr, layer) || !xv || !xt || !d1_v2t || !d2_v2t || !d3_v2t || !d1_t2v || !d2_t2v || !d3_t2v || !gamma_v || !beta_v ||
        !gamma_t || !beta_t || !out_v || !out_t || !stat || ldv < Dv || ldt < Dt || ldov < Dv || ldot < Dt)
        return LPI_EINVAL;
    const Dir v2t{d1_v2t, d2_v2t, d3_v2t, gamma_t, beta_t, xv, ldv, xt, ldt, out_t, ldot, stat + 2 * (size_t)N, stat + 3 * (size_t)N, Dv, Dt};
    const Dir t2v{d1_t2v, d2_t2v, d3_t2v, gamma_v, beta_v, xt, ldt, xv, ldv, out_v, ldov, stat, stat + (size_t)N, Dt, Dv};
    const int rp = rows_per_wg(N), G = (N + rp - 1) / rp;
    LPI_R_SWITCH(R, LPI_LAUNCH((interact_fwd_kernel<RR>), dim3(G), dim3(TPB), 0, (hipStream_t)stream, N, rp, layer, mix, eps, v2t, t2v));
    LPI_CHECK_LAST();
    return 0;
}

// grads: one flat f32 vector laid out as  [v2t: dd3 [Dt R] | dd2 [(Dv + 1) R] | dd1 row `layer` [R] | dgamma_t [Dt] | dbeta_t [Dt]]
//                                         [t2v: dd3 [Dv R] | dd2 [(Dt + 1) R] | dd1 row `layer` [R] | dgamma_v [Dv] | dbeta_v [Dv]]
extern "C" int lpi_interact_bwd(int N, int Dv, int Dt, int R, int Lyr, int layer, const float* xv, int ldv, const float* xt, int ldt,
                                const float* d1_v2t, const float* d2_v2t, const float* d3_v2t, const float* d1_t2v, const float* d2_t2v,
                                const float* d3_t2v, const float* gamma_v, const float* beta_v, const float* gamma_t, const float* beta_t, float mix,
                                float eps, const float* g_out_v, int ldgv, const float* g_out_t, int ldgt, float* dxv, int lddv, float* dxt, int lddt,
                                float* grads, float* workspace, void* stream) {
    if (bad_dims(N, Dv, Dt, R, Lyr, layer) || !xv || !xt || !d1_v2t || !d2_v2t || !d3_v2t || !d1_t2v || !d2_t2v || !d3_t2v || !gamma_v || !beta_v ||
        !gamma_t || !beta_t || !g_out_v || !g_out_t || !dxv || !dxt || !grads || !workspace || ldv < Dv || ldt < Dt || ldgv < Dv || ldgt < Dt ||
        lddv < Dv || lddt < Dt)
        return LPI_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    // the row gradients are accumulated by two passes per row: start from zero (row-strided buffers: one memset per matrix when dense)
    if (lddv == Dv) { if (hipMemsetAsync(dxv, 0, (size_t)N * Dv * sizeof(float), s) != hipSuccess) return LPI_EINVAL; }
    else if (hipMemset2DAsync(dxv, (size_t)lddv * sizeof(float), 0, (size_t)Dv * sizeof(float), N, s) != hipSuccess) return LPI_EINVAL;
    if (lddt == Dt) { if (hipMemsetAsync(dxt, 0, (size_t)N * Dt * sizeof(float), s) != hipSuccess) return LPI_EINVAL; }
    else if (hipMemset2DAsync(dxt, (size_t)lddt * sizeof(float), 0, (size_t)Dt * sizeof(float), N, s) != hipSuccess) return LPI_EINVAL;
    const Dir v2t{d1_v2t, d2_v2t, d3_v2t, gamma_t, beta_t, xv, ldv, xt, ldt, nullptr, 0, nullptr, nullptr, Dv, Dt};
    const Dir t2v{d1_t2v, d2_t2v, d3_t2v, gamma_v, beta_v, xt, ldt, xv, ldv, nullptr, 0, nullptr, nullptr, Dt, Dv};
    const int rp = rows_per_wg(N), G = (N + rp - 1) / rp;
    const int pa = part_len(Dv, Dt, R), pb = part_len(Dt, Dv, R);
    LPI_R_SWITCH(R, LPI_LAUNCH((interact_bwd_kernel<RR>), dim3(G), dim3(TPB), 0, s, N, rp, layer, mix, eps, v2t, t2v, g_out_v, ldgv, g_out_t, ldgt, dxv,
                               lddv, dxt, lddt, workspace, pa, pb));
    LPI_CHECK_LAST();
    LPI_LAUNCH(interact_reduce_kernel, dim3((pa + pb + 255) / 256), dim3(256), 0, s, G, pa + pb, workspace, grads);
    LPI_CHECK_LAST();
    return 0;
}
