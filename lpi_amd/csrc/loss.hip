// Contrastive loss, DecomposedPrompt (rank-r CP reconstruction) and retrieval ranking kernels.
//
// replaces (reference, retrieval/): loss/loss.py:75-87 (ClipLoss.forward: symmetric cross entropy),
// models/prompts/prompts.py:38-57 (DecomposedPrompt.forward) and its autograd backward,
// methods/sprompt.py:558-599 (itm_eval's per-row argsort rank search).
// All f32; all reductions run in a fixed order (no atomics) so results are bitwise reproducible.
#include "common.h"

namespace {

// ---------------------------------------------------------------------------------------------- ClipLoss
// row_lse[i] = logsumexp_j logits[i,j]  (one wave per row)
__global__ __launch_bounds__(256) void row_lse_kernel(int n, const float* __restrict__ x, int ld, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const float* r = x + (size_t)row * ld;
    float m = -INFINITY;
    for (int j = lane; j < n; j += 64) m = fmaxf(m, r[j]);
    m = wave_max(m);
    float s = 0.f;
    for (int j = lane; j < n; j += 64) s += expf(r[j] - m);
    s = wave_sum(s);
    if (lane == 0) out[row] = m + logf(s);
}

// col_lse[j] = logsumexp_i logits[i,j].  Block = 64 columns x 16 waves: wave w streams rows w, w+16, ... (coalesced across the 64
// columns) keeping an online (max, sum) per lane; the 16 partials are merged in a fixed order.  At n = 2048 (8-GPU global
// matrix) a one-thread-per-column loop would take ~0.7 ms; this takes ~30 us.
__global__ __launch_bounds__(1024) void col_lse_kernel(int n, const float* __restrict__ x, int ld, float* __restrict__ out) {
    __shared__ float pm[16][64], ps[16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + lane;
    float m = -INFINITY, s = 0.f;
    if (j < n)
        for (int i = wave; i < n; i += 16) {
            const float v = x[(size_t)i * ld + j];
            const float mn = fmaxf(m, v);
            s = s * expf(m - mn) + expf(v - mn);      // exp(-inf - finite) = 0 on the first row
            m = mn;
        }
    pm[wave][lane] = m;
    ps[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && j < n) {
        float M = pm[0][lane];
        for (int w = 1; w < 16; ++w) M = fmaxf(M, pm[w][lane]);
        float S = 0.f;
        for (int w = 0; w < 16; ++w) S += ps[w][lane] * expf(pm[w][lane] - M);   // empty partials: 0 * exp(-inf) = 0
        out[j] = M + logf(S);
    }
}

// row_lse and col_lse in ONE launch (1024-thread blocks): blocks [0, nbr) take 16 rows each (a wave per row: row_lse_kernel's arithmetic), the rest 64
// columns each (col_lse_kernel's body) — the two are independent, and alone each is a few-microsecond launch
__device__ __forceinline__ void col_lse_body(int blk, int n, const float* __restrict__ x, int ld, float* __restrict__ out, float (*pm)[64], float (*ps)[64]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = blk * 64 + lane;
    float m = -INFINITY, s = 0.f;
    if (j < n)
        for (int i = wave; i < n; i += 16) {
            const float v = x[(size_t)i * ld + j];
            const float mn = fmaxf(m, v);
            s = s * expf(m - mn) + expf(v - mn);
            m = mn;
        }
    pm[wave][lane] = m;
    ps[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && j < n) {
        float M = pm[0][lane];
        for (int w = 1; w < 16; ++w) M = fmaxf(M, pm[w][lane]);
        float S = 0.f;
        for (int w = 0; w < 16; ++w) S += ps[w][lane] * expf(pm[w][lane] - M);
        out[j] = M + logf(S);
    }
}
__global__ __launch_bounds__(1024) void lse_rows_cols_kernel(int n, int nbr, const float* __restrict__ x, int ld, float* __restrict__ rl, float* __restrict__ cl) {
    __shared__ float pm[16][64], ps[16][64];
    if ((int)blockIdx.x < nbr) {
        const int lane = threadIdx.x & 63;
        const int row = blockIdx.x * 16 + (threadIdx.x >> 6);
        if (row >= n) return;
        const float* r = x + (size_t)row * ld;
        float m = -INFINITY;
        for (int j = lane; j < n; j += 64) m = fmaxf(m, r[j]);
        m = wave_max(m);
        float s = 0.f;
        for (int j = lane; j < n; j += 64) s += expf(r[j] - m);
        s = wave_sum(s);
        if (lane == 0) rl[row] = m + logf(s);
    } else {
        col_lse_body(blockIdx.x - nbr, n, x, ld, cl, pm, ps);
    }
}

// loss = mean_i( (row_lse[i] + col_lse[i]) / 2 - logits[i,i] )   (single block, fixed-order tree)
__device__ __forceinline__ void clip_loss_reduce_body(int n, const float* __restrict__ x, int ld, const float* __restrict__ rl, const float* __restrict__ cl,
                                                      float* __restrict__ loss, float* part) {
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += 0.5f * (rl[i] + cl[i]) - x[(size_t)i * ld + i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) loss[0] = ((part[0] + part[1]) + (part[2] + part[3])) / (float)n;
}
__global__ __launch_bounds__(256) void clip_loss_reduce_kernel(int n, const float* __restrict__ x, int ld, const float* __restrict__ rl,
                                                              const float* __restrict__ cl, float* __restrict__ loss) {
    __shared__ float part[4];
    clip_loss_reduce_body(n, x, ld, rl, cl, loss, part);
}

// dlogits[i,j] = up/(2n) * (softmax_row[i,j] + softmax_col[i,j] - 2*delta_ij)
__global__ __launch_bounds__(256) void clip_loss_grad_kernel(int n, const float* __restrict__ x, int ld, const float* __restrict__ rl,
                                                            const float* __restrict__ cl, float up, float* __restrict__ dl, int lddl) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (j >= n) return;
    const float v = x[(size_t)i * ld + j];
    float g = expf(v - rl[i]) + expf(v - cl[j]);
    if (i == j) g -= 2.f;
    dl[(size_t)i * lddl + j] = g * (up / (2.f * (float)n));
}

// The LOCAL rows of dlogits and of dlogits^T (data-parallel step: a rank back-propagates only through its own B rows of the
// global n x n matrix, sprompt.py:75-80).  g[i, j]  = dL/dlogits[r0 + i, j],  gt[i, j] = dL/dlogits[j, r0 + i],  i < nloc, j < n:
// both row-major [nloc, n], so that dI_local = scale * g . T and dT_local = scale * gt . I are plain NT GEMMs.
__global__ __launch_bounds__(256) void clip_loss_local_grad_kernel(int n, const float* __restrict__ x, int ld, const float* __restrict__ rl,
                                                                  const float* __restrict__ cl, float up, int r0, float* __restrict__ g,
                                                                  float* __restrict__ gt, int ldg, float* __restrict__ loss) {
    __shared__ float part[4];
    if (loss && blockIdx.x == 0 && blockIdx.y == 0) clip_loss_reduce_body(n, x, ld, rl, cl, loss, part);      // the loss value rides in the first block (lpi_clip_loss_local)
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (j >= n) return;
    const int gi = r0 + i;
    const float c = up / (2.f * (float)n);
    const float v = x[(size_t)gi * ld + j];
    g[(size_t)i * ldg + j] = (expf(v - rl[gi]) + expf(v - cl[j]) - (gi == j ? 2.f : 0.f)) * c;
    const float w = x[(size_t)j * ld + gi];       // strided read: n * nloc floats in all, served by L2
    gt[(size_t)i * ldg + j] = (expf(w - rl[j]) + expf(w - cl[gi]) - (gi == j ? 2.f : 0.f)) * c;
}

// dst[r, 0:cols] = src[r, 0:cols] (f32, arbitrary row strides): pads / packs small operands without the host framework's copy kernels
__global__ __launch_bounds__(256) void copy_rows_kernel(int rows, int cols, const float* __restrict__ src, long lds, float* __restrict__ dst, long ldd) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)rows * cols) return;
    const int r = (int)(t / cols), c = (int)(t % cols);
    dst[r * ldd + c] = src[r * lds + c];
}

// ---------------------------------------------------------------------------------------------- task-id selection (a11)
// sel[i] = argmin_t min_c sum_e |f[i,e] - key[t,c,e]|   (sprompt.py:336-368; first t among equal distances).  One wave per row.
__global__ __launch_bounds__(256) void l1_task_id_kernel(int n, int E, int T, int C, const float* __restrict__ f, int ldf,
                                                        const float* __restrict__ keys, int32_t* __restrict__ sel, float* __restrict__ dist) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const float* fr = f + (size_t)row * ldf;
    float best = INFINITY;
    int bt = 0;
    for (int t = 0; t < T; ++t) {
        float dmin = INFINITY;
        for (int c = 0; c < C; ++c) {
            const float* k = keys + ((size_t)t * C + c) * E;
            float s = 0.f;
            for (int e = lane; e < E; e += 64) s += fabsf(fr[e] - k[e]);
            s = wave_sum(s);
            dmin = fminf(dmin, s);
        }
        if (dist && lane == 0) dist[(size_t)row * T + t] = dmin;
        if (dmin < best) { best = dmin; bt = t; }
    }
    if (lane == 0) sel[row] = bt;
}

// p = p - lr * (momentum-buffered gradient with weight decay): torch.optim.SGD(momentum, weight_decay), sprompt.py:253, over one flat
// parameter vector.  first != 0: the momentum buffer is initialised with the gradient (torch's first step).
__global__ __launch_bounds__(256) void sgd_step_kernel(long n, float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf,
                                                      float lr, float momentum, float wd, int first) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float d = g[i] + wd * p[i];
    const float b = first ? d : momentum * buf[i] + d;
    buf[i] = b;
    p[i] = p[i] - lr * b;
}

// ---------------------------------------------------------------------------------------------- CP prompt
constexpr int MAXR = 16;

__global__ __launch_bounds__(256) void cp_fwd_kernel(int Lyr, int P, int D, int r, const float* __restrict__ d1, const float* __restrict__ d2,
                                                    const float* __restrict__ d3, float sc, float* __restrict__ out) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)Lyr * P * D) return;
    const int d = (int)(t % D), p = (int)((t / D) % P), l = (int)(t / ((long)D * P));
    float s = 0.f;
    for (int k = 0; k < r; ++k) s += d1[l * r + k] * d2[p * r + k] * d3[d * r + k];   // same product order as prompts.py:49
    out[t] = s * sc;
}

// both prompt stacks (visual and textual share dim_1_share, prompts.py:38-57) in one launch: elements [0, nv) are the visual stack's
__global__ __launch_bounds__(256) void cp_fwd2_kernel(int Lyr, int P, int Dv, int Dt, int r, const float* __restrict__ d1, const float* __restrict__ d2v,
                                                     const float* __restrict__ d2t, const float* __restrict__ d3v, const float* __restrict__ d3t, float sc,
                                                     float* __restrict__ outv, float* __restrict__ outt) {
    long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long nv = (long)Lyr * P * Dv, nt = (long)Lyr * P * Dt;
    const bool isv = t < nv;
    if (!isv) t -= nv;
    if (!isv && t >= nt) return;
    const int D = isv ? Dv : Dt;
    const float* d2 = isv ? d2v : d2t;
    const float* d3 = isv ? d3v : d3t;
    const int d = (int)(t % D), p = (int)((t / D) % P), l = (int)(t / ((long)D * P));
    float s = 0.f;
    for (int k = 0; k < r; ++k) s += d1[l * r + k] * d2[p * r + k] * d3[d * r + k];   // same product order as prompts.py:49
    (isv ? outv : outt)[t] = s * sc;
}

// g3[d,k] = sc * sum_{l,p} dout[l,p,d] * d1[l,k] * d2[p,k].  Block = 64 columns d; 16 waves each take the (l,p) rows
// w, w+16, ... in order, lanes = columns (coalesced); the 16 partials are added in order (deterministic).
__device__ __forceinline__ void cp_bwd_g3_body(int blk, int Lyr, int P, int D, int r, const float* __restrict__ d1, const float* __restrict__ d2,
                                               float sc, const float* __restrict__ dout, float* __restrict__ g3, float (*part)[64][MAXR + 1]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int d = blk * 64 + lane;
    float acc[MAXR];
#pragma unroll
    for (int k = 0; k < MAXR; ++k) acc[k] = 0.f;
    if (d < D)
        for (int row = wave; row < Lyr * P; row += 16) {
            const int l = row / P, p = row % P;
            const float g = dout[(size_t)row * D + d];
#pragma unroll
            for (int k = 0; k < MAXR; ++k)
                if (k < r) acc[k] += g * (d1[l * r + k] * d2[p * r + k]);
        }
#pragma unroll
    for (int k = 0; k < MAXR; ++k) part[wave][lane][k] = acc[k];
    __syncthreads();
    if (wave == 0 && d < D)
        for (int k = 0; k < r; ++k) {
            float t = part[0][lane][k];
            for (int w = 1; w < 16; ++w) t += part[w][lane][k];
            g3[d * r + k] = t * sc;
        }
}
__global__ __launch_bounds__(1024) void cp_bwd_g3_kernel(int Lyr, int P, int D, int r, const float* __restrict__ d1, const float* __restrict__ d2,
                                                        float sc, const float* __restrict__ dout, float* __restrict__ g3) {
    __shared__ float part[16][64][MAXR + 1];
    cp_bwd_g3_body(blockIdx.x, Lyr, P, D, r, d1, d2, sc, dout, g3, part);
}

// t[l,p,k] = sum_d dout[l,p,d] * d3[d,k]: one wave per (l,p) row, 4 rows per block -> scratch [Lyr*P*r]
__device__ __forceinline__ void cp_bwd_t_body(int row, int rows, int D, int r, const float* __restrict__ d3, const float* __restrict__ dout,
                                              float* __restrict__ t) {
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    float acc[MAXR];
#pragma unroll
    for (int k = 0; k < MAXR; ++k) acc[k] = 0.f;
    const float* g = dout + (size_t)row * D;
    for (int d = lane; d < D; d += 64) {
        const float v = g[d];
#pragma unroll
        for (int k = 0; k < MAXR; ++k)
            if (k < r) acc[k] += v * d3[d * r + k];
    }
#pragma unroll
    for (int k = 0; k < MAXR; ++k)
        if (k < r) {
            const float s = wave_sum(acc[k]);
            if (lane == 0) t[row * r + k] = s;
        }
}
__global__ __launch_bounds__(256) void cp_bwd_t_kernel(int rows, int D, int r, const float* __restrict__ d3, const float* __restrict__ dout,
                                                      float* __restrict__ t) {
    cp_bwd_t_body(blockIdx.x * 4 + (threadIdx.x >> 6), rows, D, r, d3, dout, t);
}
// both stacks' t rows in one launch: rows [0, rows) visual (-> tv), [rows, 2 rows) textual (-> tt)
__global__ __launch_bounds__(256) void cp_bwd_t2_kernel(int rows, int Dv, int Dt, int r, const float* __restrict__ d3v, const float* __restrict__ d3t,
                                                       const float* __restrict__ doutv, const float* __restrict__ doutt, float* __restrict__ tv,
                                                       float* __restrict__ tt) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row < rows) cp_bwd_t_body(row, rows, Dv, r, d3v, doutv, tv);
    else cp_bwd_t_body(row - rows, rows, Dt, r, d3t, doutt, tt);
}

// g1[l,k] = sc * sum_p t[l,p,k] d2[p,k];  g2[p,k] = sc * sum_l t[l,p,k] d1[l,k]     (tiny, single block)
__device__ __forceinline__ void cp_bwd_g12_body(int Lyr, int P, int r, const float* __restrict__ d1, const float* __restrict__ d2,
                                                float sc, const float* __restrict__ t, float* __restrict__ g1, float* __restrict__ g2,
                                                int accumulate_g1) {
    for (int i = threadIdx.x; i < Lyr * r; i += blockDim.x) {
        const int l = i / r, k = i % r;
        float s = 0.f;
        for (int p = 0; p < P; ++p) s += t[(l * P + p) * r + k] * d2[p * r + k];
        s *= sc;
        g1[i] = accumulate_g1 ? g1[i] + s : s;
    }
    for (int i = threadIdx.x; i < P * r; i += blockDim.x) {
        const int p = i / r, k = i % r;
        float s = 0.f;
        for (int l = 0; l < Lyr; ++l) s += t[(l * P + p) * r + k] * d1[l * r + k];
        g2[i] = s * sc;
    }
}
__global__ __launch_bounds__(256) void cp_bwd_g12_kernel(int Lyr, int P, int r, const float* __restrict__ d1, const float* __restrict__ d2,
                                                        float sc, const float* __restrict__ t, float* __restrict__ g1, float* __restrict__ g2,
                                                        int accumulate_g1) {
    cp_bwd_g12_body(Lyr, P, r, d1, d2, sc, t, g1, g2, accumulate_g1);
}
// the g3 blocks of both stacks and ONE block for g1 (visual, then textual added: the shared factor's two contributions in a fixed order) and the two g2
__global__ __launch_bounds__(1024) void cp_bwd_g2_kernel(int Lyr, int P, int Dv, int Dt, int r, const float* __restrict__ d1, const float* __restrict__ d2v,
                                                        const float* __restrict__ d2t, float sc, const float* __restrict__ doutv, const float* __restrict__ doutt,
                                                        const float* __restrict__ tv, const float* __restrict__ tt, float* __restrict__ g1,
                                                        float* __restrict__ g2v, float* __restrict__ g2t, float* __restrict__ g3v, float* __restrict__ g3t) {
    __shared__ float part[16][64][MAXR + 1];
    const int nbv = (Dv + 63) / 64, nbt = (Dt + 63) / 64;
    const int b = blockIdx.x;
    if (b < nbv) cp_bwd_g3_body(b, Lyr, P, Dv, r, d1, d2v, sc, doutv, g3v, part);
    else if (b < nbv + nbt) cp_bwd_g3_body(b - nbv, Lyr, P, Dt, r, d1, d2t, sc, doutt, g3t, part);
    else {
        cp_bwd_g12_body(Lyr, P, r, d1, d2v, sc, tv, g1, g2v, 0);
        __syncthreads();      // g1[i] is rewritten by the thread that wrote it (same loop bounds): the barrier is for clarity only
        cp_bwd_g12_body(Lyr, P, r, d1, d2t, sc, tt, g1, g2t, 1);
    }
}


// ---------------------------------------------------------------------------------------------- alignment loss
// slinet.py:143-158: v = mean_d(vis)/T, t = mean_d(txt)/T ([Lyr,P]); S = v t^T; loss = w * ClipLoss(S).  Single block.
__global__ __launch_bounds__(1024) void align_loss_kernel(int Lyr, int P, int Dv, int Dt, const float* __restrict__ vis,
                                                         const float* __restrict__ txt, float temp, float w, float* __restrict__ loss,
                                                         float* __restrict__ dvis, float* __restrict__ dtxt, int means_ready) {
    extern __shared__ float sm[];
    float* v = sm;                 // [Lyr*P]
    float* t = v + Lyr * P;        // [Lyr*P]
    float* Sm = t + Lyr * P;       // [Lyr*Lyr]
    float* dS = Sm + Lyr * Lyr;    // [Lyr*Lyr]
    float* rl = dS + Lyr * Lyr;    // [Lyr]
    float* cl = rl + Lyr;          // [Lyr]
    float* dv = cl + Lyr;          // [Lyr*P]
    float* dt = dv + Lyr * P;      // [Lyr*P]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int nr = Lyr * P;
    if (means_ready) {      // align_means_kernel (one workgroup per row) left each row's mean/T in the first element of its gradient row
        for (int i = threadIdx.x; i < nr; i += blockDim.x) {
            v[i] = dvis[(size_t)i * Dv];
            t[i] = dtxt[(size_t)i * Dt];
        }
    } else {
        for (int row = wave; row < 2 * nr; row += nw) {
            const bool isv = row < nr;
            const int rr = isv ? row : row - nr;
            const int D = isv ? Dv : Dt;
            const float* src = (isv ? vis : txt) + (size_t)rr * D;
            float s = 0.f;
            for (int d = lane; d < D; d += 64) s += src[d];
            s = wave_sum(s);
            if (lane == 0) (isv ? v : t)[rr] = s / (float)D / temp;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < Lyr * Lyr; i += blockDim.x) {
        const int a = i / Lyr, b = i % Lyr;
        float s = 0.f;
        for (int p = 0; p < P; ++p) s += v[a * P + p] * t[b * P + p];
        Sm[i] = s;
    }
    __syncthreads();
    if ((int)threadIdx.x < 2 * Lyr) {
        const bool row = (int)threadIdx.x < Lyr;
        const int i = row ? threadIdx.x : threadIdx.x - Lyr;
        float m = -INFINITY;
        for (int j = 0; j < Lyr; ++j) m = fmaxf(m, row ? Sm[i * Lyr + j] : Sm[j * Lyr + i]);
        float s = 0.f;
        for (int j = 0; j < Lyr; ++j) s += expf((row ? Sm[i * Lyr + j] : Sm[j * Lyr + i]) - m);
        (row ? rl : cl)[i] = m + logf(s);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int i = 0; i < Lyr; ++i) s += 0.5f * (rl[i] + cl[i]) - Sm[i * Lyr + i];
        loss[0] = w * s / (float)Lyr;
    }
    if (!dvis || !dtxt) return;
    for (int i = threadIdx.x; i < Lyr * Lyr; i += blockDim.x) {
        const int a = i / Lyr, b = i % Lyr;
        float g = expf(Sm[i] - rl[a]) + expf(Sm[i] - cl[b]);
        if (a == b) g -= 2.f;
        dS[i] = g * (w / (2.f * (float)Lyr));
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * nr; i += blockDim.x) {
        const bool isv = i < nr;
        const int rr = isv ? i : i - nr;
        const int a = rr / P, p = rr % P;
        float s = 0.f;
        if (isv) for (int b = 0; b < Lyr; ++b) s += dS[a * Lyr + b] * t[b * P + p];
        else for (int b = 0; b < Lyr; ++b) s += dS[b * Lyr + a] * v[b * P + p];
        (isv ? dv : dt)[rr] = s / ((float)(isv ? Dv : Dt) * temp);
    }
    __syncthreads();
    // d mean_d / d x[.., d] is the same for every d: leave each row's value in its first element; align_expand_kernel (one
    // workgroup per row) spreads it — one workgroup writing all Lyr*P*(Dv+Dt) floats took ~100 us
    for (int i = threadIdx.x; i < nr; i += blockDim.x) {
        dvis[(size_t)i * Dv] = dv[i];
        dtxt[(size_t)i * Dt] = dt[i];
    }
}

// The alignment loss in TWO launches (lpi_align_loss_fwd_bwd2): align_means2_kernel leaves the 2 Lyr P row means / T in a scratch vector; then EVERY
// workgroup of align_loss_expand_kernel (one per prompt row, as align_expand_kernel) recomputes the tiny Lyr x Lyr core from them — the same formulas as
// align_loss_kernel, value for value — and writes its own row of the gradient; workgroup 0 also writes the loss.
__global__ __launch_bounds__(256) void align_means2_kernel(int nr, int Dv, int Dt, const float* __restrict__ vis, const float* __restrict__ txt,
                                                          float temp, float* __restrict__ means) {
    __shared__ float part[4];
    const bool isv = (int)blockIdx.x < nr;
    const int row = isv ? blockIdx.x : blockIdx.x - nr;
    const int D = isv ? Dv : Dt;
    const float* src = (isv ? vis : txt) + (size_t)row * D;
    float s = 0.f;
    for (int d = threadIdx.x; d < D; d += 256) s += src[d];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) means[blockIdx.x] = ((part[0] + part[1]) + (part[2] + part[3])) / (float)D / temp;
}
__global__ __launch_bounds__(256) void align_loss_expand_kernel(int Lyr, int P, int Dv, int Dt, const float* __restrict__ means, float temp, float w,
                                                               float* __restrict__ loss, float* __restrict__ dvis, float* __restrict__ dtxt) {
    extern __shared__ float sm[];
    const int nr = Lyr * P;
    float* v = sm;                 // [Lyr*P]
    float* t = v + nr;             // [Lyr*P]
    float* Sm = t + nr;            // [Lyr*Lyr]
    float* dS = Sm + Lyr * Lyr;    // [Lyr*Lyr]
    float* rl = dS + Lyr * Lyr;    // [Lyr]
    float* cl = rl + Lyr;          // [Lyr]
    for (int i = threadIdx.x; i < nr; i += blockDim.x) { v[i] = means[i]; t[i] = means[nr + i]; }
    __syncthreads();
    for (int i = threadIdx.x; i < Lyr * Lyr; i += blockDim.x) {
        const int a = i / Lyr, b = i % Lyr;
        float s = 0.f;
        for (int p = 0; p < P; ++p) s += v[a * P + p] * t[b * P + p];
        Sm[i] = s;
    }
    __syncthreads();
    if ((int)threadIdx.x < 2 * Lyr) {
        const bool row = (int)threadIdx.x < Lyr;
        const int i = row ? threadIdx.x : threadIdx.x - Lyr;
        float m = -INFINITY;
        for (int j = 0; j < Lyr; ++j) m = fmaxf(m, row ? Sm[i * Lyr + j] : Sm[j * Lyr + i]);
        float s = 0.f;
        for (int j = 0; j < Lyr; ++j) s += expf((row ? Sm[i * Lyr + j] : Sm[j * Lyr + i]) - m);
        (row ? rl : cl)[i] = m + logf(s);
    }
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        float s = 0.f;
        for (int i = 0; i < Lyr; ++i) s += 0.5f * (rl[i] + cl[i]) - Sm[i * Lyr + i];
        loss[0] = w * s / (float)Lyr;
    }
    if (!dvis || !dtxt) return;
    for (int i = threadIdx.x; i < Lyr * Lyr; i += blockDim.x) {
        const int a = i / Lyr, b = i % Lyr;
        float g = expf(Sm[i] - rl[a]) + expf(Sm[i] - cl[b]);
        if (a == b) g -= 2.f;
        dS[i] = g * (w / (2.f * (float)Lyr));
    }
    __syncthreads();
    const bool isv = (int)blockIdx.x < nr;
    const int rr = isv ? blockIdx.x : blockIdx.x - nr;
    const int a = rr / P, p = rr % P;
    float s = 0.f;
    if (isv) for (int b = 0; b < Lyr; ++b) s += dS[a * Lyr + b] * t[b * P + p];
    else for (int b = 0; b < Lyr; ++b) s += dS[b * Lyr + a] * v[b * P + p];
    const int D = isv ? Dv : Dt;
    const float val = s / ((float)D * temp);
    float* o = (isv ? dvis : dtxt) + (size_t)rr * D;
    for (int d = threadIdx.x; d < D; d += 256) o[d] = val;
}

// row means / T, one workgroup per prompt row, parked in the first element of that row of the (not yet written) gradient buffers
__global__ __launch_bounds__(256) void align_means_kernel(int nr, int Dv, int Dt, const float* __restrict__ vis, const float* __restrict__ txt,
                                                         float temp, float* __restrict__ dvis, float* __restrict__ dtxt) {
    __shared__ float part[4];
    const bool isv = (int)blockIdx.x < nr;
    const int row = isv ? blockIdx.x : blockIdx.x - nr;
    const int D = isv ? Dv : Dt;
    const float* src = (isv ? vis : txt) + (size_t)row * D;
    float s = 0.f;
    for (int d = threadIdx.x; d < D; d += 256) s += src[d];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) (isv ? dvis : dtxt)[(size_t)row * D] = ((part[0] + part[1]) + (part[2] + part[3])) / (float)D / temp;
}

__global__ __launch_bounds__(256) void align_expand_kernel(int nr, int Dv, int Dt, float* __restrict__ dvis, float* __restrict__ dtxt) {
    const int row = blockIdx.x < nr ? blockIdx.x : blockIdx.x - nr;
    const int D = blockIdx.x < nr ? Dv : Dt;
    float* o = (blockIdx.x < nr ? dvis : dtxt) + (size_t)row * D;
    const float v = o[0];
    __syncthreads();                       // every thread has read element 0 before thread 0 rewrites it
    for (int d = threadIdx.x; d < D; d += 256) o[d] = v;
}

// ---------------------------------------------------------------------------------------------- task loss (nt_bxent)
// loss/loss.py:6-33 as written: cos = cosine_similarity rows of X [T,D]; diag -> +inf; z = sigmoid(cos / temp);
// l_ij = BCEWithLogits(z_ij, t_ij) = max(z,0) - z*t + log(1 + exp(-|z|)); per row mean over positives + mean over negatives; mean over rows.
// Three tiny kernels: Gram matrix (one workgroup per pair), loss + dL/dcos (one workgroup), dX (one workgroup per row chunk).
__global__ __launch_bounds__(256) void gram_kernel(int T, int D, const float* __restrict__ X, float* __restrict__ G) {
    __shared__ float part[4];
    const int i = blockIdx.x / T, j = blockIdx.x % T;
    if (j < i) return;                       // symmetric: compute the upper triangle, mirror below
    const float* a = X + (size_t)i * D;
    const float* b = X + (size_t)j * D;
    float s = 0.f;
    for (int d = threadIdx.x; d < D; d += 256) s += a[d] * b[d];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float g = (part[0] + part[1]) + (part[2] + part[3]);
        G[i * T + j] = g;
        G[j * T + i] = g;
    }
}

constexpr int MAXT = 32;
__global__ __launch_bounds__(64) void nt_bxent_kernel(int T, const float* __restrict__ G, const int32_t* __restrict__ target, float temp,
                                                     float weight, float* __restrict__ loss, float* __restrict__ dcos) {
    __shared__ float rowloss[MAXT];
    const int i = threadIdx.x;
    if (i < T) {
        const float ni = fmaxf(sqrtf(G[i * T + i]), 1e-8f);            // F.cosine_similarity eps
        float lp = 0.f, ln = 0.f, np_ = 0.f;
        for (int j = 0; j < T; ++j) np_ += (float)target[i * T + j];
        const float nn = (float)T - np_;
        for (int j = 0; j < T; ++j) {
            const float nj = fmaxf(sqrtf(G[j * T + j]), 1e-8f);
            const float cs = (i == j) ? INFINITY : G[i * T + j] / (ni * nj);
            const float z = (i == j) ? 1.f : 1.f / (1.f + expf(-cs / temp));
            const float t = (float)target[i * T + j];
            const float l = fmaxf(z, 0.f) - z * t + log1pf(expf(-fabsf(z)));
            // d l / d z = sigmoid(z) - t ; d z / d cos = z (1 - z) / temp ; row weights 1/(num_pos T) or 1/(num_neg T)
            const float wrow = (t != 0.f ? 1.f / np_ : 1.f / nn) / (float)T;
            if (t != 0.f) lp += l; else ln += l;
            dcos[i * T + j] = (i == j) ? 0.f : weight * wrow * (1.f / (1.f + expf(-z)) - t) * z * (1.f - z) / temp;
        }
        rowloss[i] = lp / np_ + ln / nn;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int k = 0; k < T; ++k) s += rowloss[k];
        loss[0] = weight * s / (float)T;
    }
}

// dX[r,:] = sum_j (dcos[r,j] + dcos[j,r]) * ( x_j / (|x_r||x_j|) - cos_rj * x_r / |x_r|^2 )   for one row r
__global__ __launch_bounds__(256) void nt_bxent_dx_kernel(int T, int D, int r, const float* __restrict__ X, const float* __restrict__ G,
                                                         const float* __restrict__ dcos, float* __restrict__ dx, int accumulate) {
    const int d = blockIdx.x * 256 + threadIdx.x;
    if (d >= D) return;
    const float nr = fmaxf(sqrtf(G[r * T + r]), 1e-8f);
    const float xr = X[(size_t)r * D + d];
    float acc = 0.f;
    for (int j = 0; j < T; ++j) {
        if (j == r) continue;
        const float nj = fmaxf(sqrtf(G[j * T + j]), 1e-8f);
        const float c = G[r * T + j] / (nr * nj);
        const float w = dcos[r * T + j] + dcos[j * T + r];
        acc += w * (X[(size_t)j * D + d] / (nr * nj) - c * xr / (nr * nr));
    }
    dx[d] = accumulate ? dx[d] + acc : acc;
}

// ---------------------------------------------------------------------------------------------- task keys: KMeans (methods/sprompt.py:370-397)
// The heavy parts of scikit-learn's KMeans fit (k-means++ seeding + Lloyd iterations; lpi_amd/kmeans.py drives them and keeps the random draws and the
// convergence logic on the host, on vectors of n floats at most): the features [n, E] never leave the device.  All sums run in a fixed order.
// out[c, i] = |x_i - cand_c|^2 (cand_c = row cand[c] of X): one wave per point, differences in f32
__global__ __launch_bounds__(256) void kmeans_sqdist_kernel(int n, int E, int nc, const float* __restrict__ X, int ldx, const int32_t* __restrict__ cand,
                                                           float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const float* x = X + (size_t)i * ldx;
    for (int c = 0; c < nc; ++c) {
        const float* y = X + (size_t)cand[c] * ldx;
        float s = 0.f;
        for (int e = lane; e < E; e += 64) { const float d = x[e] - y[e]; s = fmaf(d, d, s); }
        s = wave_sum(s);
        if (lane == 0) out[(size_t)c * n + i] = s;
    }
}
// labels[i] = argmin_c |x_i - C_c|^2 (first minimum); *changed = 1 if any label differs from the one stored before (labels is in / out)
__global__ __launch_bounds__(256) void kmeans_assign_kernel(int n, int E, int k, const float* __restrict__ X, int ldx, const float* __restrict__ C,
                                                           int32_t* __restrict__ labels, int32_t* __restrict__ changed, float* __restrict__ mindist) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const float* x = X + (size_t)i * ldx;
    float best = INFINITY;
    int bc = 0;
    for (int c = 0; c < k; ++c) {
        const float* y = C + (size_t)c * E;
        float s = 0.f;
        for (int e = lane; e < E; e += 64) { const float d = x[e] - y[e]; s = fmaf(d, d, s); }
        s = wave_sum(s);
        if (s < best) { best = s; bc = c; }
    }
    if (lane == 0) {
        if (labels[i] != bc) { labels[i] = bc; *changed = 1; }
        if (mindist) mindist[i] = best;
    }
}
// Cnew[c, cols] = mean of the points labelled c; counts[c] (written by the column chunk 0 blocks).  Block = (cluster, 64 columns) x 16 waves: wave w scans
// the points w, w + 16, ... in order, the 16 partial sums and counts are added in order.
__global__ __launch_bounds__(1024) void kmeans_update_kernel(int n, int E, int k, const float* __restrict__ X, int ldx, const int32_t* __restrict__ labels,
                                                            float* __restrict__ Cnew, float* __restrict__ counts) {
    __shared__ float part[16][64];
    __shared__ int pcnt[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x, col = blockIdx.y * 64 + lane;
    float s = 0.f;
    int cnt = 0;
    for (int i = wave; i < n; i += 16)
        if (labels[i] == c) {
            ++cnt;
            if (col < E) s += X[(size_t)i * ldx + col];
        }
    part[wave][lane] = s;
    if (lane == 0) pcnt[wave] = cnt;
    __syncthreads();
    if (wave == 0) {
        float t = part[0][lane];
        int m = pcnt[0];
        for (int w = 1; w < 16; ++w) { t += part[w][lane]; m += pcnt[w]; }
        if (col < E) Cnew[(size_t)c * E + col] = m > 0 ? t / (float)m : 0.f;
        if (blockIdx.y == 0 && lane == 0) counts[c] = (float)m;
    }
}
// colsum[e] = sum_i (x[i, e] - center[e]), colsq[e] = sum_i (x[i, e] - center[e])^2 (center NULL = 0; for the tolerance mean_e var_i x[i, e] * tol: a first
// call gives the column means, a second one with them the centred squares — no E[x^2] - mean^2 cancellation): block = 64 columns x 16 waves, fixed order
__global__ __launch_bounds__(1024) void kmeans_colstats_kernel(int n, int E, const float* __restrict__ X, int ldx, const float* __restrict__ center,
                                                              float* __restrict__ colsum, float* __restrict__ colsq) {
    __shared__ float ps[16][64], pq[16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + lane;
    float s = 0.f, q = 0.f;
    if (col < E) {
        const float c = center ? center[col] : 0.f;
        for (int i = wave; i < n; i += 16) { const float v = X[(size_t)i * ldx + col] - c; s += v; q = fmaf(v, v, q); }
    }
    ps[wave][lane] = s;
    pq[wave][lane] = q;
    __syncthreads();
    if (wave == 0 && col < E) {
        float a = ps[0][lane], b = pq[0][lane];
        for (int w = 1; w < 16; ++w) { a += ps[w][lane]; b += pq[w][lane]; }
        colsum[col] = a;
        colsq[col] = b;
    }
}

// ---------------------------------------------------------------------------------------------- retrieval
// rank of the best ground-truth column under np.argsort(score)[::-1] (later index first among equal scores)
__global__ __launch_bounds__(256) void retrieval_rank_kernel(int n_rows, int n_cols, const float* __restrict__ s, int ld,
                                                            const int32_t* __restrict__ gt, int gpr, int32_t* __restrict__ rank) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    const float* r = s + (size_t)row * ld;
    int best = 0x7fffffff;
    for (int t = 0; t < gpr; ++t) {
        const int gidx = gt[(size_t)row * gpr + t];
        if (gidx < 0) continue;
        const float gv = r[gidx];
        int cnt = 0;
        for (int j = lane; j < n_cols; j += 64) {
            const float v = r[j];
            cnt += (v > gv) || (v == gv && j > gidx);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
        best = min(best, cnt);
    }
    if (lane == 0) rank[row] = best;
}

// top-k per row by k rounds of (value desc, index desc) arg-max; k is small (<= 16)
__global__ __launch_bounds__(256) void topk_kernel(int n_rows, int n_cols, int k, const float* __restrict__ s, int ld,
                                                  int32_t* __restrict__ idx, float* __restrict__ val) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    const float* r = s + (size_t)row * ld;
    float pv = INFINITY;
    int pi = 0x7fffffff;
    for (int t = 0; t < k; ++t) {
        float bv = -INFINITY;
        int bi = -1;
        for (int j = lane; j < n_cols; j += 64) {
            const float v = r[j];
            const bool after_prev = (v < pv) || (v == pv && j < pi);       // strictly after the previous pick in the order
            const bool better = (v > bv) || (v == bv && j > bi);
            if (after_prev && better) { bv = v; bi = j; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if ((ov > bv) || (ov == bv && oi > bi)) { bv = ov; bi = oi; }
        }
        if (lane == 0) { idx[(size_t)row * k + t] = bi; if (val) val[(size_t)row * k + t] = bv; }
        pv = bv;
        pi = bi;
    }
}

}  // namespace

#define S(stream) ((hipStream_t)(stream))

extern "C" int lpi_clip_loss_fwd_bwd(int n, const float* logits, int ld, float upstream, float* loss, float* dlogits, int lddl,
                                     float* row_lse, float* col_lse, void* stream) {
    if (!logits || !loss || !row_lse || !col_lse || n <= 0 || ld < n) return LPI_EINVAL;
    LPI_LAUNCH(row_lse_kernel, dim3((n + 3) / 4), dim3(256), 0, S(stream), n, logits, ld, row_lse);
    LPI_LAUNCH(col_lse_kernel, dim3((n + 63) / 64), dim3(1024), 0, S(stream), n, logits, ld, col_lse);
    LPI_LAUNCH(clip_loss_reduce_kernel, dim3(1), dim3(256), 0, S(stream), n, logits, ld, row_lse, col_lse, loss);
    if (dlogits) {
        if (lddl < n) return LPI_EINVAL;
        LPI_LAUNCH(clip_loss_grad_kernel, dim3((n + 255) / 256, n), dim3(256), 0, S(stream), n, logits, ld, row_lse, col_lse, upstream,
                           dlogits, lddl);
    }
    LPI_CHECK_LAST();
    return 0;
}

extern "C" int lpi_clip_loss_local_grad(int n, const float* logits, int ld, const float* row_lse, const float* col_lse, float upstream,
                                        int r0, int nloc, float* g, float* gt, int ldg, void* stream) {
    if (!logits || !row_lse || !col_lse || !g || !gt || n <= 0 || ld < n || r0 < 0 || nloc <= 0 || r0 + nloc > n || ldg < n) return LPI_EINVAL;
    LPI_LAUNCH(clip_loss_local_grad_kernel, dim3((n + 255) / 256, nloc), dim3(256), 0, S(stream), n, logits, ld, row_lse, col_lse, upstream, r0, g, gt, ldg,
               (float*)nullptr);
    LPI_CHECK_LAST();
    return 0;
}

// lpi_clip_loss_fwd_bwd (no dlogits) + lpi_clip_loss_local_grad in TWO launches instead of four: the two log-sum-exp vectors in one, the loss value
// in the first block of the local-gradient kernel.  The same arithmetic per value (bit for bit the four-launch results).
extern "C" int lpi_clip_loss_local(int n, const float* logits, int ld, float upstream, int r0, int nloc, float* loss, float* row_lse, float* col_lse,
                                   float* g, float* gt, int ldg, void* stream) {
    if (!logits || !loss || !row_lse || !col_lse || n <= 0 || ld < n) return LPI_EINVAL;
    if (g || gt) {
        if (!g || !gt || r0 < 0 || nloc <= 0 || r0 + nloc > n || ldg < n) return LPI_EINVAL;
    }
    const int nbr = (n + 15) / 16;
    LPI_LAUNCH(lse_rows_cols_kernel, dim3(nbr + (n + 63) / 64), dim3(1024), 0, S(stream), n, nbr, logits, ld, row_lse, col_lse);
    if (g) LPI_LAUNCH(clip_loss_local_grad_kernel, dim3((n + 255) / 256, nloc), dim3(256), 0, S(stream), n, logits, ld, row_lse, col_lse, upstream, r0, g, gt, ldg, loss);
    else LPI_LAUNCH(clip_loss_reduce_kernel, dim3(1), dim3(256), 0, S(stream), n, logits, ld, row_lse, col_lse, loss);
    LPI_CHECK_LAST();
    return 0;
}

// Row-wise cross-entropy of a [rows, n] block of logits whose row i has label label0 + i — the `local_loss=True` form of the
// contrastive loss (sprompt.py:278-283: logits_per_image = local images x ALL texts, labels offset by rank * B, loss/loss.py:62-73).
// One wave per row: lse, the row's loss lse_i - x[i, label], and (dx != NULL) dx[i, :] = upstream * (softmax(x[i, :]) - onehot(label)).
// dx may be x itself (the gradient overwrites the logits).
__global__ __launch_bounds__(256) void ce_rows_kernel(int rows, int n, const float* x, int ld, int label0, float upstream,
                                                     float* __restrict__ loss_rows, float* dx, int lddx) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* r = x + (size_t)row * ld;
    float m = -INFINITY;
    for (int j = lane; j < n; j += 64) m = fmaxf(m, r[j]);
    m = wave_max(m);
    float s = 0.f;
    for (int j = lane; j < n; j += 64) s += expf(r[j] - m);
    s = wave_sum(s);
    const float lse = m + logf(s);
    const int lab = label0 + row;
    if (lane == 0) loss_rows[row] = lse - r[lab];
    if (dx) {
        float* d = dx + (size_t)row * lddx;
        for (int j = lane; j < n; j += 64) d[j] = upstream * (expf(r[j] - lse) - (j == lab ? 1.f : 0.f));
    }
}
// out[0] = scale * sum of v[0..n) in a fixed order (one workgroup)
__global__ __launch_bounds__(256) void sum_scaled_kernel(int n, const float* __restrict__ a, const float* __restrict__ b, float scale,
                                                        float* __restrict__ out) {
    __shared__ float part[256];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += a[i] + (b ? b[i] : 0.f);
    part[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = part[0] * scale;
}

extern "C" int lpi_ce_rows_fwd_bwd(int rows, int n, const float* logits, int ld, int label0, float upstream, float* loss_rows, float* dlogits,
                                   int lddl, void* stream) {
    if (!logits || !loss_rows || rows <= 0 || n <= 0 || ld < n || label0 < 0 || label0 + rows > n || (dlogits && lddl < n)) return LPI_EINVAL;
    LPI_LAUNCH(ce_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, S(stream), rows, n, logits, ld, label0, upstream, loss_rows, dlogits, lddl);
    LPI_CHECK_LAST();
    return 0;
}
extern "C" int lpi_sum_scaled(int n, const float* a, const float* b, float scale, float* out, void* stream) {
    if (!a || !out || n <= 0) return LPI_EINVAL;
    LPI_LAUNCH(sum_scaled_kernel, dim3(1), dim3(256), 0, S(stream), n, a, b, scale, out);
    LPI_CHECK_LAST();
    return 0;
}

extern "C" int lpi_copy_rows(int rows, int cols, const float* src, long lds, float* dst, long ldd, void* stream) {
    if (!src || !dst || rows <= 0 || cols <= 0 || lds < cols || ldd < cols) return LPI_EINVAL;
    const long n = (long)rows * cols;
    LPI_LAUNCH(copy_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, S(stream), rows, cols, src, lds, dst, ldd);
    LPI_CHECK_LAST();
    return 0;
}

extern "C" int lpi_l1_task_id(int n, int E, int T, int C, const float* feat, int ldf, const float* keys, int32_t* sel, float* dist,
                              void* stream) {
    if (!feat || !keys || !sel || n <= 0 || E <= 0 || T <= 0 || C <= 0 || ldf < E) return LPI_EINVAL;
    LPI_LAUNCH(l1_task_id_kernel, dim3((n + 3) / 4), dim3(256), 0, S(stream), n, E, T, C, feat, ldf, keys, sel, dist);
    LPI_CHECK_LAST();
    return 0;
}

extern "C" int lpi_sgd_step(long n, float* param, const float* grad, float* momentum_buf, float lr, float momentum, float weight_decay,
                            int first, void* stream) {
    if (!param || !grad || !momentum_buf || n <= 0) return LPI_EINVAL;
    LPI_LAUNCH(sgd_step_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, S(stream), n, param, grad, momentum_buf, lr, momentum, weight_decay, first);
    LPI_CHECK_LAST();
    return 0;
}

extern "C" int lpi_prompt_cp_fwd(int Lyr, int P, int D, int r, const float* d1, const float* d2, const float* d3, float scale, float* out,
                                 void* stream) {
    if (!d1 || !d2 || !d3 || !out || Lyr <= 0 || P <= 0 || D <= 0 || r <= 0 || r > MAXR) return LPI_EINVAL;
    const long n = (long)Lyr * P * D;
    LPI_LAUNCH(cp_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, S(stream), Lyr, P, D, r, d1, d2, d3, scale / (float)r, out);
    LPI_CHECK_LAST();
    return 0;
}

extern "C" int lpi_prompt_cp_fwd2(int Lyr, int P, int Dv, int Dt, int r, const float* d1, const float* d2v, const float* d2t, const float* d3v,
                                  const float* d3t, float scale, float* outv, float* outt, void* stream) {
    if (!d1 || !d2v || !d2t || !d3v || !d3t || !outv || !outt || Lyr <= 0 || P <= 0 || Dv <= 0 || Dt <= 0 || r <= 0 || r > MAXR) return LPI_EINVAL;
    const long n = (long)Lyr * P * (Dv + Dt);
    LPI_LAUNCH(cp_fwd2_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, S(stream), Lyr, P, Dv, Dt, r, d1, d2v, d2t, d3v, d3t, scale / (float)r, outv, outt);
    LPI_CHECK_LAST();
    return 0;
}

// lpi_prompt_cp_bwd for the visual stack (g1 overwritten) followed by the textual one (g1 accumulated) in TWO launches instead of six; the same
// arithmetic per value.  scratch: 2 * Lyr * P * r floats.
extern "C" int lpi_prompt_cp_bwd2(int Lyr, int P, int Dv, int Dt, int r, const float* d1, const float* d2v, const float* d2t, const float* d3v,
                                  const float* d3t, float scale, const float* doutv, const float* doutt, float* g1, float* g2v, float* g2t, float* g3v,
                                  float* g3t, float* scratch, void* stream) {
    if (!d1 || !d2v || !d2t || !d3v || !d3t || !doutv || !doutt || !g1 || !g2v || !g2t || !g3v || !g3t || !scratch || Lyr <= 0 || P <= 0 || Dv <= 0 ||
        Dt <= 0 || r <= 0 || r > MAXR)
        return LPI_EINVAL;
    const float sc = scale / (float)r;
    const int rows = Lyr * P;
    float* tv = scratch;
    float* tt = scratch + (size_t)rows * r;
    LPI_LAUNCH(cp_bwd_t2_kernel, dim3((2 * rows + 3) / 4), dim3(256), 0, S(stream), rows, Dv, Dt, r, d3v, d3t, doutv, doutt, tv, tt);
    LPI_LAUNCH(cp_bwd_g2_kernel, dim3((Dv + 63) / 64 + (Dt + 63) / 64 + 1), dim3(1024), 0, S(stream), Lyr, P, Dv, Dt, r, d1, d2v, d2t, sc, doutv, doutt, tv, tt,
               g1, g2v, g2t, g3v, g3t);
    LPI_CHECK_LAST();
    return 0;
}

extern "C" int lpi_prompt_cp_bwd(int Lyr, int P, int D, int r, const float* d1, const float* d2, const float* d3, float scale,
                                 const float* dout, float* g1, float* g2, float* g3, int accumulate_g1, float* scratch, void* stream) {
    if (!d1 || !d2 || !d3 || !dout || !g1 || !g2 || !g3 || !scratch || Lyr <= 0 || P <= 0 || D <= 0 || r <= 0 || r > MAXR) return LPI_EINVAL;
    const float sc = scale / (float)r;
    const int rows = Lyr * P;
    LPI_LAUNCH(cp_bwd_t_kernel, dim3((rows + 3) / 4), dim3(256), 0, S(stream), rows, D, r, d3, dout, scratch);
    LPI_LAUNCH(cp_bwd_g12_kernel, dim3(1), dim3(256), 0, S(stream), Lyr, P, r, d1, d2, sc, scratch, g1, g2, accumulate_g1);
    LPI_LAUNCH(cp_bwd_g3_kernel, dim3((D + 63) / 64), dim3(1024), 0, S(stream), Lyr, P, D, r, d1, d2, sc, dout, g3);
    LPI_CHECK_LAST();
    return 0;
}

extern "C" int lpi_align_loss_fwd_bwd(int Lyr, int P, int Dv, int Dt, const float* vis, const float* txt, float temp, float weight,
                                      float* loss, float* dvis, float* dtxt, void* stream) {
    if (!vis || !txt || !loss || Lyr <= 0 || P <= 0 || Dv <= 0 || Dt <= 0 || temp <= 0.f) return LPI_EINVAL;
    const size_t lds = ((size_t)4 * Lyr * P + 2 * Lyr * Lyr + 2 * Lyr) * sizeof(float);
    if (lds > 64 * 1024) return LPI_EINVAL;
    const int grads = dvis && dtxt;      // with gradient buffers the row means and the final broadcast run grid-wide around the one-workgroup core
    if (grads) LPI_LAUNCH(align_means_kernel, dim3(2 * Lyr * P), dim3(256), 0, S(stream), Lyr * P, Dv, Dt, vis, txt, temp, dvis, dtxt);
    LPI_LAUNCH(align_loss_kernel, dim3(1), dim3(1024), lds, S(stream), Lyr, P, Dv, Dt, vis, txt, temp, weight, loss, dvis, dtxt, grads);
    if (grads) LPI_LAUNCH(align_expand_kernel, dim3(2 * Lyr * P), dim3(256), 0, S(stream), Lyr * P, Dv, Dt, dvis, dtxt);
    LPI_CHECK_LAST();
    return 0;
}

extern "C" int lpi_align_loss_fwd_bwd2(int Lyr, int P, int Dv, int Dt, const float* vis, const float* txt, float temp, float weight,
                                       float* loss, float* dvis, float* dtxt, float* scratch, void* stream) {
    if (!vis || !txt || !loss || !scratch || Lyr <= 0 || P <= 0 || Dv <= 0 || Dt <= 0 || temp <= 0.f || ((dvis != nullptr) != (dtxt != nullptr))) return LPI_EINVAL;
    const size_t lds = ((size_t)2 * Lyr * P + 2 * Lyr * Lyr + 2 * Lyr) * sizeof(float);
    if (lds > 64 * 1024) return LPI_EINVAL;
    LPI_LAUNCH(align_means2_kernel, dim3(2 * Lyr * P), dim3(256), 0, S(stream), Lyr * P, Dv, Dt, vis, txt, temp, scratch);
    LPI_LAUNCH(align_loss_expand_kernel, dim3(dvis ? 2 * Lyr * P : 1), dim3(256), lds, S(stream), Lyr, P, Dv, Dt, scratch, temp, weight, loss, dvis, dtxt);
    LPI_CHECK_LAST();
    return 0;
}

extern "C" int lpi_nt_bxent_fwd_bwd(int T, int D, int row, const float* X, const int32_t* target, float temp, float weight, float* loss,
                                    float* dx_row, int accumulate, float* scratch, void* stream) {
    if (!X || !target || !loss || !scratch || T <= 0 || T > MAXT || D <= 0 || temp <= 0.f || row >= T) return LPI_EINVAL;
    float* G = scratch;            // [T*T]
    float* dcos = scratch + T * T; // [T*T]
    LPI_LAUNCH(gram_kernel, dim3(T * T), dim3(256), 0, S(stream), T, D, X, G);
    LPI_LAUNCH(nt_bxent_kernel, dim3(1), dim3(64), 0, S(stream), T, G, target, temp, weight, loss, dcos);
    if (dx_row && row >= 0) LPI_LAUNCH(nt_bxent_dx_kernel, dim3((D + 255) / 256), dim3(256), 0, S(stream), T, D, row, X, G, dcos, dx_row, accumulate);
    LPI_CHECK_LAST();
    return 0;
}

extern "C" int lpi_kmeans_sqdist(int n, int E, int nc, const float* X, int ldx, const int32_t* cand, float* out, void* stream) {
    if (!X || !cand || !out || n <= 0 || E <= 0 || nc <= 0 || ldx < E) return LPI_EINVAL;
    LPI_LAUNCH(kmeans_sqdist_kernel, dim3((n + 3) / 4), dim3(256), 0, S(stream), n, E, nc, X, ldx, cand, out);
    LPI_CHECK_LAST();
    return 0;
}
extern "C" int lpi_kmeans_assign(int n, int E, int k, const float* X, int ldx, const float* centers, int32_t* labels, int32_t* changed, float* mindist,
                                 void* stream) {
    if (!X || !centers || !labels || !changed || n <= 0 || E <= 0 || k <= 0 || ldx < E) return LPI_EINVAL;
    LPI_LAUNCH(kmeans_assign_kernel, dim3((n + 3) / 4), dim3(256), 0, S(stream), n, E, k, X, ldx, centers, labels, changed, mindist);
    LPI_CHECK_LAST();
    return 0;
}
extern "C" int lpi_kmeans_update(int n, int E, int k, const float* X, int ldx, const int32_t* labels, float* new_centers, float* counts, void* stream) {
    if (!X || !labels || !new_centers || !counts || n <= 0 || E <= 0 || k <= 0 || ldx < E) return LPI_EINVAL;
    LPI_LAUNCH(kmeans_update_kernel, dim3(k, (E + 63) / 64), dim3(1024), 0, S(stream), n, E, k, X, ldx, labels, new_centers, counts);
    LPI_CHECK_LAST();
    return 0;
}
extern "C" int lpi_kmeans_colstats(int n, int E, const float* X, int ldx, const float* center, float* colsum, float* colsq, void* stream) {
    if (!X || !colsum || !colsq || n <= 0 || E <= 0 || ldx < E) return LPI_EINVAL;
    LPI_LAUNCH(kmeans_colstats_kernel, dim3((E + 63) / 64), dim3(1024), 0, S(stream), n, E, X, ldx, center, colsum, colsq);
    LPI_CHECK_LAST();
    return 0;
}

extern "C" int lpi_retrieval_rank(int n_rows, int n_cols, const float* scores, int ld, const int32_t* gt, int gt_per_row, int32_t* rank,
                                  void* stream) {
    if (!scores || !gt || !rank || n_rows <= 0 || n_cols <= 0 || gt_per_row <= 0 || ld < n_cols) return LPI_EINVAL;
    LPI_LAUNCH(retrieval_rank_kernel, dim3((n_rows + 3) / 4), dim3(256), 0, S(stream), n_rows, n_cols, scores, ld, gt, gt_per_row, rank);
    LPI_CHECK_LAST();
    return 0;
}

extern "C" int lpi_topk(int n_rows, int n_cols, int k, const float* scores, int ld, int32_t* idx, float* val, void* stream) {
    if (!scores || !idx || n_rows <= 0 || n_cols <= 0 || k <= 0 || k > n_cols || ld < n_cols) return LPI_EINVAL;
    LPI_LAUNCH(topk_kernel, dim3((n_rows + 3) / 4), dim3(256), 0, S(stream), n_rows, n_cols, k, scores, ld, idx, val);
    LPI_CHECK_LAST();
    return 0;
}
