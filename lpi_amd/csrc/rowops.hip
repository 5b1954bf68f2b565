// Row-wise, HBM-bound kernels of the LPI hot path: LayerNorm forward/backward (fp32 statistics), the vision
// and text front ends (class token / positional embedding / prompt insertion / ln_pre), deep-prompt add, pooled
// heads (CLS / EOT gather + LayerNorm), L2 normalisation and small utilities.
//
// replaces (reference, retrieval/): models/clip/model.py:154-160 (LayerNorm), :227-251 (VisionTransformer
// front end), :189-193 (deep prompt add), :255 (ln_post on CLS); models/clip/prompt_learner.py:52-61,128-163
// (TextEncoder front end / EOT gather), models/slinet.py:122,133 (L2 normalise).
//
// One 64-lane wave owns one row of d floats and keeps it in registers (float4 per lane per 256 columns), so each
// row is read once and written once; reductions are wave shuffles; 4 rows per 256-thread block; every access is a
// 16-byte coalesced load/store.  These kernels are bounded by HBM bandwidth, not by the matrix cores.
#include <type_traits>
#include "common.h"

// the guard of the one-sweep row statistics (lpi_rowstat_guard, include/lpi_hip.h): per HOST THREAD, like the stream the caller launches on.
// counter: device memory; flag: the DEVICE alias of a word of pinned host memory (or of device memory), written once, by the row that takes the
// counter from 0 to 1 — the host reads its own word whenever it likes: no copy, no event, no kernel
struct RowstatGuard { int* counter; int* flag; };
static thread_local RowstatGuard t_rowstat_guard = {nullptr, nullptr};
extern "C" int lpi_rowstat_guard(int32_t* counter, int32_t* flag) {
    int* dflag = nullptr;
    if (flag) {
        if (!counter) return LPI_EINVAL;
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, flag) != hipSuccess || !at.devicePointer) {      // not memory the device can write: refuse, never fault
            (void)hipGetLastError();
            return LPI_EINVAL;
        }
        dflag = static_cast<int*>(at.devicePointer);
    }
    t_rowstat_guard = {counter, dflag};
    return 0;
}

namespace {

constexpr float LN_EPS = 1e-5f;
constexpr int MAXC = 8;  // float4 chunks per lane: d <= 2048

// NC = compile-time bound on the chunks a lane holds (ceil(d / 256)): the LayerNorm kernels are instantiated per NC so that a row
// costs NC float4 registers, not MAXC (with the run-time bound alone the d = 768 backward took 90-176 VGPRs: 2-5 waves per SIMD).
template <int NC> struct RowT {
    f32x4 v[NC];
};
typedef RowT<MAXC> Row;

template <int NC, typename F> __device__ __forceinline__ void for_chunks_n(int d, int lane, F f) {
    const int nch = d >> 2;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int c = lane + (i << 6);
        if (c < nch) f(i, c << 2);
    }
}
template <typename F> __device__ __forceinline__ void for_chunks(int d, int lane, F f) { for_chunks_n<MAXC>(d, lane, f); }

__device__ __forceinline__ float hsum(f32x4 v) { return (v[0] + v[1]) + (v[2] + v[3]); }

// mean / rstd of the row held in r (two-pass, like ATen's CPU LayerNorm in effect)
template <int NC>
__device__ __forceinline__ void row_stats(const RowT<NC>& r, int d, int lane, float& mu, float& rs) {
    float s = 0.f;
    for_chunks_n<NC>(d, lane, [&](int i, int) { s += hsum(r.v[i]); });
    mu = wave_sum(s) / (float)d;
    float ss = 0.f;
    for_chunks_n<NC>(d, lane, [&](int i, int) { f32x4 c = r.v[i] - mu; ss += hsum(c * c); });
    const float var = wave_sum(ss) / (float)d;
    rs = 1.0f / sqrtf(var + LN_EPS);
}

// mean / rstd of a row in ONE sweep (sum and sum of squares reduced side by side: their cross-lane chains overlap; variance as E[x^2] - mean^2, clamped):
// for the statistics a kernel leaves of the rows it WRITES, where a second dependent reduction behind the store lengthened every wave (the vision
// front end 53 -> 83 us with the two-sweep form)
// The form loses digits as (mean / std)^2 * 1e-7 (2e-3 in rstd at |mean| = 30 std).  `guard` (lpi_rowstat_guard; NULL = off) counts the rows whose
// mean^2 exceeds ROWSTAT_GUARD x their variance — 8 deviations: the error is still ~6e-6 there — so that the host can fall back to the two-sweep
// statistics pass long before the one-sweep form hurts (engine.DualEncoder.poll_rowstat_guard).
constexpr float ROWSTAT_GUARD = 64.f;
__device__ __forceinline__ void rowstat_guard_check(RowstatGuard guard, float mu, float var) {
    if (guard.counter && mu * mu > ROWSTAT_GUARD * fmaxf(var, 0.f)) {
        if (atomicAdd(guard.counter, 1) == 0 && guard.flag) *reinterpret_cast<volatile int*>(guard.flag) = 1;      // one write towards the host per reset
    }
}
template <int NC>
__device__ __forceinline__ void row_stats_1sweep(const RowT<NC>& r, int d, int lane, float& mu, float& rs, RowstatGuard guard = RowstatGuard{nullptr, nullptr}) {
    float s = 0.f, q = 0.f;
    for_chunks_n<NC>(d, lane, [&](int i, int) { s += hsum(r.v[i]); q += hsum(r.v[i] * r.v[i]); });
    s = wave_sum(s);
    q = wave_sum(q);
    mu = s / (float)d;
    const float var = q / (float)d - mu * mu;
    rs = 1.0f / sqrtf(fmaxf(var, 0.f) + LN_EPS);
    if (lane == 0) rowstat_guard_check(guard, mu, var);
}

// y = LN(r)
template <typename TY, int NC>
__device__ __forceinline__ void ln_apply_store(const RowT<NC>& r, int d, int lane, float mu, float rs, const float* gamma,
                                               const float* beta, TY* y) {
    for_chunks_n<NC>(d, lane, [&](int i, int col) {
        f32x4 g = *reinterpret_cast<const f32x4*>(gamma + col);
        f32x4 b = *reinterpret_cast<const f32x4*>(beta + col);
        Elem<TY>::st4(y + col, (r.v[i] - mu) * rs * g + b);
    });
}

// y = LN(r), stored, and left in r as stored (rounded to TY): for the statistics of the OUTPUT row
template <typename TY, int NC>
__device__ __forceinline__ void ln_apply_round_store(RowT<NC>& r, int d, int lane, float mu, float rs, const float* gamma, const float* beta, TY* y) {
    for_chunks_n<NC>(d, lane, [&](int i, int col) {
        f32x4 g = *reinterpret_cast<const f32x4*>(gamma + col);
        f32x4 b = *reinterpret_cast<const f32x4*>(beta + col);
        r.v[i] = rounded4<TY>((r.v[i] - mu) * rs * g + b);
        Elem<TY>::st4(y + col, r.v[i]);
    });
}

// dxn = rstd * (g*dy - mean(g*dy) - xhat * mean(g*dy*xhat)); x row in `x`, dy row in `dy`; result left in dy
template <int NC>
__device__ __forceinline__ void ln_bwd_row(RowT<NC>& dy, const RowT<NC>& x, int d, int lane, float mu, float rs, const float* gamma) {
    float s1 = 0.f, s2 = 0.f;
    for_chunks_n<NC>(d, lane, [&](int i, int col) {
        f32x4 g = *reinterpret_cast<const f32x4*>(gamma + col);
        f32x4 gdy = g * dy.v[i];
        f32x4 xh = (x.v[i] - mu) * rs;
        dy.v[i] = gdy;
        s1 += hsum(gdy);
        s2 += hsum(gdy * xh);
    });
    const float c1 = wave_sum(s1) / (float)d, c2 = wave_sum(s2) / (float)d;
    for_chunks_n<NC>(d, lane, [&](int i, int) {
        f32x4 xh = (x.v[i] - mu) * rs;
        dy.v[i] = (dy.v[i] - c1 - xh * c2) * rs;
    });
}

template <typename TX, typename TY, int NC>
__global__ __launch_bounds__(256) void ln_fwd_kernel(int rows, int d, const TX* __restrict__ x, int ldx,
                                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                                    TY* __restrict__ y, int ldy, float* __restrict__ mean, float* __restrict__ rstd) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    RowT<NC> r;
    const TX* xr = x + (size_t)row * ldx;
    for_chunks_n<NC>(d, lane, [&](int i, int col) { r.v[i] = Elem<TX>::ld4(xr + col); });
    float mu, rs;
    row_stats(r, d, lane, mu, rs);
    ln_apply_store<TY>(r, d, lane, mu, rs, gamma, beta, y + (size_t)row * ldy);
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
}

template <typename TX, typename TDY, typename TCAST, int NC>
__device__ __forceinline__ void ln_bwd_body(int blk, int rows, int d, const TDY* __restrict__ dy, int lddy,
                                            const TX* __restrict__ x, int ldx, const float* __restrict__ gamma,
                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                            float* __restrict__ dx, int lddx, TCAST* __restrict__ dx_cast, int ldcast, int accumulate,
                                            int mapP, int mapL, int map0, const int* __restrict__ map_rs) {
    const int lane = threadIdx.x & 63;
    const int crow = blk * 4 + (threadIdx.x >> 6);
    if (crow >= rows) return;
    // row map (mapP != 0): dy is COMPACT [B*P, d]; x / statistics / dx live at row (crow / P) * L + row0 + crow % P of the full stream
    const int row = mapP ? (map_rs ? map_rs[crow / mapP] : (crow / mapP) * mapL) + map0 + crow % mapP : crow;      // map_rs: ragged batch (row starts)
    RowT<NC> g, xr;
    const TDY* dyr = dy + (size_t)crow * lddy;
    const TX* xp = x + (size_t)row * ldx;
    for_chunks_n<NC>(d, lane, [&](int i, int col) {
        g.v[i] = Elem<TDY>::ld4(dyr + col);
        xr.v[i] = Elem<TX>::ld4(xp + col);
    });
    ln_bwd_row(g, xr, d, lane, mean[row], rstd[row], gamma);
    if (dx) {
        float* dxr = dx + (size_t)row * lddx;
        for_chunks_n<NC>(d, lane, [&](int i, int col) {
            f32x4 t = g.v[i];
            if (accumulate) t += *reinterpret_cast<const f32x4*>(dxr + col);
            *reinterpret_cast<f32x4*>(dxr + col) = t;
            if (dx_cast) Elem<TCAST>::st4(dx_cast + (size_t)row * ldcast + col, t);
        });
    } else {   // the `cast` copy IS the gradient stream (in/out), no f32 stream is kept
        TCAST* dxr = dx_cast + (size_t)row * ldcast;
        for_chunks_n<NC>(d, lane, [&](int i, int col) {
            f32x4 t = g.v[i];
            if (accumulate) t += Elem<TCAST>::ld4(dxr + col);
            Elem<TCAST>::st4(dxr + col, t);
        });
    }
}
template <typename TX, typename TDY, typename TCAST, int NC>
__global__ __launch_bounds__(256) void ln_bwd_kernel(int rows, int d, const TDY* __restrict__ dy, int lddy,
                                                    const TX* __restrict__ x, int ldx, const float* __restrict__ gamma,
                                                    const float* __restrict__ mean, const float* __restrict__ rstd,
                                                    float* __restrict__ dx, int lddx, TCAST* __restrict__ dx_cast, int ldcast, int accumulate,
                                                    int mapP, int mapL, int map0, const int* __restrict__ map_rs) {
    ln_bwd_body<TX, TDY, TCAST, NC>(blockIdx.x, rows, d, dy, lddy, x, ldx, gamma, mean, rstd, dx, lddx, dx_cast, ldcast, accumulate, mapP, mapL, map0, map_rs);
}

// ---------------------------------------------------------------------------------------------------------
// bf16 mode with the fp16 residual stream: every operand is a 2-byte type, so a lane takes 8 elements (16 bytes) per access and
// a 32-lane half wave owns a row — the same number of memory instructions as the 4-element kernels above moves two rows.
// (Measured: with 8-byte accesses these kernels are instruction/latency-bound and halving the bytes bought nothing.)
constexpr int MAXC8 = 8;   // 8-element chunks per lane of a half wave: d <= 2048

template <int NC> struct Row8 {
    float v[NC][8];
};

template <int NC, typename F> __device__ __forceinline__ void for_chunks8(int d, int hl, F f) {
    const int nch = d >> 3;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int c = hl + (i << 5);
        if (c < nch) f(i, c << 3);
    }
}

__device__ __forceinline__ void ld8(const f16_t* p, float (&o)[8]) {
    typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
    const f16x8 h = *reinterpret_cast<const f16x8*>(p);
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (float)h[j];
}
__device__ __forceinline__ void ld8(const bf16_t* p, float (&o)[8]) {
    const uint4 u = *reinterpret_cast<const uint4*>(p);
    const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        o[2 * j] = __uint_as_float(w[j] << 16);
        o[2 * j + 1] = __uint_as_float(w[j] & 0xFFFF0000u);
    }
}
__device__ __forceinline__ void ld8(const float* p, float (&o)[8]) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) { o[j] = a[j]; o[4 + j] = b[j]; }
}
__device__ __forceinline__ void st8(bf16_t* p, const float (&v)[8]) {
    bf16x8 b;
#pragma unroll
    for (int j = 0; j < 8; ++j) b[j] = (__bf16)v[j];
    *reinterpret_cast<bf16x8*>(p) = b;
}
__device__ __forceinline__ void st8(f16_t* p, const float (&v)[8]) {
    f16x8 h;
#pragma unroll
    for (int j = 0; j < 8; ++j) h[j] = (_Float16)v[j];
    *reinterpret_cast<f16x8*>(p) = h;
}

template <int NC, typename TY>
__device__ __forceinline__ void ln_fwd_h16_body(int blk, int rows, int d, const f16_t* __restrict__ x, int ldx, const float* __restrict__ gamma,
                                                const float* __restrict__ beta, TY* __restrict__ y, int ldy,
                                                float* __restrict__ mean, float* __restrict__ rstd) {
    const int hl = threadIdx.x & 31;
    const int row = blk * 8 + (threadIdx.x >> 5);
    if (row >= rows) return;
    Row8<NC> r;
    const f16_t* xr = x + (size_t)row * ldx;
    for_chunks8<NC>(d, hl, [&](int i, int col) { ld8(xr + col, r.v[i]); });
    float s = 0.f;
    for_chunks8<NC>(d, hl, [&](int i, int) {
#pragma unroll
        for (int j = 0; j < 8; ++j) s += r.v[i][j];
    });
    const float mu = half_sum(s) / (float)d;
    float ss = 0.f;
    for_chunks8<NC>(d, hl, [&](int i, int) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float c = r.v[i][j] - mu; ss += c * c; }
    });
    const float rs = 1.0f / sqrtf(half_sum(ss) / (float)d + LN_EPS);
    if constexpr (!std::is_void<TY>::value) {      // TY = void: the statistics only (the LayerNorm itself is folded into the next GEMM, LPI_EPI_LN)
        TY* yr = y + (size_t)row * ldy;
        for_chunks8<NC>(d, hl, [&](int i, int col) {
            float g[8], b[8], o[8];
            ld8(gamma + col, g);
            ld8(beta + col, b);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (r.v[i][j] - mu) * rs * g[j] + b[j];
            st8(yr + col, o);
        });
    }
    if (hl == 0) { mean[row] = mu; rstd[row] = rs; }
}
template <int NC, typename TY = bf16_t>
__global__ __launch_bounds__(256) void ln_fwd_h16_kernel(int rows, int d, const f16_t* __restrict__ x, int ldx, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, TY* __restrict__ y, int ldy,
                                                        float* __restrict__ mean, float* __restrict__ rstd) {
    ln_fwd_h16_body<NC, TY>(blockIdx.x, rows, d, x, ldx, gamma, beta, y, ldy, mean, rstd);
}
// TWO LayerNorms in one launch (the vision and the text tower's LayerNorm of the same layer: the text one alone is a 4-5 us kernel for
// 16 MB of traffic — mostly launch ramp): blocks [0, nb0) run problem 0, the rest problem 1.
struct LnFwdP {
    int rows, d, ldx, ldy;
    const f16_t* x; const float* gamma; const float* beta; void* y; float* mean; float* rstd;
};
template <int NC0, int NC1, typename TY>
__global__ __launch_bounds__(256) void ln_fwd_h16_pair_kernel(LnFwdP p0, LnFwdP p1, int nb0) {
    if ((int)blockIdx.x < nb0) ln_fwd_h16_body<NC0, TY>(blockIdx.x, p0.rows, p0.d, p0.x, p0.ldx, p0.gamma, p0.beta, (TY*)p0.y, p0.ldy, p0.mean, p0.rstd);
    else ln_fwd_h16_body<NC1, TY>(blockIdx.x - nb0, p1.rows, p1.d, p1.x, p1.ldx, p1.gamma, p1.beta, (TY*)p1.y, p1.ldy, p1.mean, p1.rstd);
}

// dx_stream (bf16, in/out) += LN'(dy): the bf16-mode backward with the fp16 saved input
template <int NC>
__device__ __forceinline__ void ln_bwd_h16_body(int blk, int rows, int d, const bf16_t* __restrict__ dy, int lddy, const f16_t* __restrict__ x, int ldx,
                                                const float* __restrict__ gamma, const float* __restrict__ mean,
                                                const float* __restrict__ rstd, bf16_t* __restrict__ dxs, int ldcast, int accumulate,
                                                int mapP, int mapL, int map0, const int* __restrict__ map_rs) {
    const int hl = threadIdx.x & 31;
    const int crow = blk * 8 + (threadIdx.x >> 5);
    if (crow >= rows) return;
    const int row = mapP ? (map_rs ? map_rs[crow / mapP] : (crow / mapP) * mapL) + map0 + crow % mapP : crow;      // see ln_bwd_kernel
    Row8<NC> g, xh;
    const bf16_t* dyr = dy + (size_t)crow * lddy;
    const f16_t* xr = x + (size_t)row * ldx;
    const float mu = mean[row], rs = rstd[row];
    float s1 = 0.f, s2 = 0.f;
    for_chunks8<NC>(d, hl, [&](int i, int col) {
        float gm[8];
        ld8(dyr + col, g.v[i]);
        ld8(xr + col, xh.v[i]);
        ld8(gamma + col, gm);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            g.v[i][j] *= gm[j];
            xh.v[i][j] = (xh.v[i][j] - mu) * rs;
            s1 += g.v[i][j];
            s2 += g.v[i][j] * xh.v[i][j];
        }
    });
    const float c1 = half_sum(s1) / (float)d, c2 = half_sum(s2) / (float)d;
    bf16_t* o = dxs + (size_t)row * ldcast;
    for_chunks8<NC>(d, hl, [&](int i, int col) {
        float acc[8];
        if (accumulate) {
            ld8(o + col, acc);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = 0.f;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += (g.v[i][j] - c1 - xh.v[i][j] * c2) * rs;
        st8(o + col, acc);
    });
}
template <int NC>
__global__ __launch_bounds__(256) void ln_bwd_h16_kernel(int rows, int d, const bf16_t* __restrict__ dy, int lddy, const f16_t* __restrict__ x, int ldx,
                                                        const float* __restrict__ gamma, const float* __restrict__ mean,
                                                        const float* __restrict__ rstd, bf16_t* __restrict__ dxs, int ldcast, int accumulate,
                                                        int mapP, int mapL, int map0, const int* __restrict__ map_rs) {
    ln_bwd_h16_body<NC>(blockIdx.x, rows, d, dy, lddy, x, ldx, gamma, mean, rstd, dxs, ldcast, accumulate, mapP, mapL, map0, map_rs);
}
struct LnBwdP {
    int rows, d, lddy, ldx, ldcast, accumulate;
    const bf16_t* dy; const f16_t* x; const float* gamma; const float* mean; const float* rstd; bf16_t* dxs;
};
template <int NC0, int NC1>
__global__ __launch_bounds__(256) void ln_bwd_h16_pair_kernel(LnBwdP p0, LnBwdP p1, int nb0) {
    if ((int)blockIdx.x < nb0)
        ln_bwd_h16_body<NC0>(blockIdx.x, p0.rows, p0.d, p0.dy, p0.lddy, p0.x, p0.ldx, p0.gamma, p0.mean, p0.rstd, p0.dxs, p0.ldcast, p0.accumulate, 0, 0, 0, nullptr);
    else
        ln_bwd_h16_body<NC1>(blockIdx.x - nb0, p1.rows, p1.d, p1.dy, p1.lddy, p1.x, p1.ldx, p1.gamma, p1.mean, p1.rstd, p1.dxs, p1.ldcast, p1.accumulate, 0, 0, 0,
                             nullptr);
}

// ---------------------------------------------------------------------------------------------------------
// vision front end
template <typename T, bool VEC4 = false>
__global__ __launch_bounds__(256) void patchify_kernel(int B, int R, int ps, int G, int Kp, const float* __restrict__ img,
                                                      T* __restrict__ cols, int ldcols) {
    // one thread per 4 consecutive k of one patch row; k = c*ps*ps + dy*ps + dx (conv weight [d,3,ps,ps] flattened)
    const int K = 3 * ps * ps;
    const long total = (long)B * G * G * (Kp >> 2);
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int k4 = (int)(t % (Kp >> 2)) << 2;
    const long patch = t / (Kp >> 2);
    const int b = (int)(patch / (G * G)), gy = (int)(patch % (G * G)) / G, gx = (int)(patch % G);
    f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (VEC4) {      // ps and R multiples of 4: the four k are four consecutive pixels of one patch row — one 16-byte load, one index split
        if (k4 < K) {
            const int c = k4 / (ps * ps), rem = k4 % (ps * ps), dy = rem / ps, dx = rem % ps;
            v = *reinterpret_cast<const f32x4*>(img + (((size_t)b * 3 + c) * R + gy * ps + dy) * R + gx * ps + dx);
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = k4 + j;
            if (k < K) {
                const int c = k / (ps * ps), rem = k % (ps * ps), dy = rem / ps, dx = rem % ps;
                v[j] = img[(((size_t)b * 3 + c) * R + gy * ps + dy) * R + gx * ps + dx];
            }
        }
    }
    Elem<T>::st4(cols + (size_t)patch * ldcols + k4, v);
}

// The same im2col from UINT8 pixels [B, 3, R, R] with ToTensor + Normalize folded in (utils/data.py:201-204: x / 255, then (x - mean) / std per channel):
// lut[c][v] holds that f32 value for channel c and byte v, computed by the HOST with the very torch operations the f32 pipeline applies, so the columns are
// bit for bit those of patchify_kernel on the normalised f32 image — at a quarter of the host-to-device bytes.  ps and R multiples of 4.
template <typename T>
__global__ __launch_bounds__(256) void patchify_u8_kernel(int B, int R, int ps, int G, int Kp, const uint8_t* __restrict__ img, const float* __restrict__ lut,
                                                         T* __restrict__ cols, int ldcols) {
    __shared__ float s_lut[3 * 256];
    for (int i = threadIdx.x; i < 3 * 256; i += 256) s_lut[i] = lut[i];
    __syncthreads();
    const int K = 3 * ps * ps;
    const long total = (long)B * G * G * (Kp >> 2);
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int k4 = (int)(t % (Kp >> 2)) << 2;
    const long patch = t / (Kp >> 2);
    const int b = (int)(patch / (G * G)), gy = (int)(patch % (G * G)) / G, gx = (int)(patch % G);
    f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
    if (k4 < K) {
        const int c = k4 / (ps * ps), rem = k4 % (ps * ps), dy = rem / ps, dx = rem % ps;
        const uint32_t px = *reinterpret_cast<const uint32_t*>(img + (((size_t)b * 3 + c) * R + gy * ps + dy) * R + gx * ps + dx);
        const float* l = s_lut + c * 256;
        v = f32x4{l[px & 255u], l[(px >> 8) & 255u], l[(px >> 16) & 255u], l[px >> 24]};
    }
    Elem<T>::st4(cols + (size_t)patch * ldcols + k4, v);
}

__device__ __forceinline__ void vis_pre_row(Row& r, int b, int l, int G2, int P, int d, int lane, const float* patch_emb,
                                            int ldpe, const float* cls, const float* pos, const float* prompt0, long pbs) {
    if (l == 0) {
        for_chunks(d, lane, [&](int i, int col) {
            r.v[i] = *reinterpret_cast<const f32x4*>(cls + col) + *reinterpret_cast<const f32x4*>(pos + col);
        });
    } else if (l <= P) {
        const float* p = prompt0 + (size_t)b * pbs + (size_t)(l - 1) * d;
        for_chunks(d, lane, [&](int i, int col) { r.v[i] = *reinterpret_cast<const f32x4*>(p + col); });
    } else {
        const int g = l - 1 - P;
        const float* pe = patch_emb + ((size_t)b * G2 + g) * ldpe;
        const float* pp = pos + (size_t)(g + 1) * d;
        for_chunks(d, lane, [&](int i, int col) {
            r.v[i] = *reinterpret_cast<const f32x4*>(pe + col) + *reinterpret_cast<const f32x4*>(pp + col);
        });
    }
}

template <typename TX>
__global__ __launch_bounds__(256) void vis_assemble_fwd_kernel(int B, int G2, int P, int d, const float* __restrict__ patch_emb,
                                                              int ldpe, const float* __restrict__ cls, const float* __restrict__ pos,
                                                              const float* __restrict__ prompt0, long pbs,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              TX* __restrict__ x0, float* __restrict__ mean, float* __restrict__ rstd,
                                                              float* __restrict__ omean, float* __restrict__ orstd, RowstatGuard guard) {
    const int L = 1 + P + G2;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B * L) return;
    const int b = row / L, l = row % L;
    Row r;
    vis_pre_row(r, b, l, G2, P, d, lane, patch_emb, ldpe, cls, pos, prompt0, pbs);
    float mu, rs;
    row_stats(r, d, lane, mu, rs);
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
    if (omean) {      // ... and the statistics of the row as STORED: the first block's ln_1 (folded into its in_proj GEMM) needs no pass over x0
        ln_apply_round_store<TX>(r, d, lane, mu, rs, gamma, beta, x0 + (size_t)row * d);
        row_stats_1sweep(r, d, lane, mu, rs, guard);
        if (lane == 0) { omean[row] = mu; orstd[row] = rs; }
    } else {
        ln_apply_store<TX>(r, d, lane, mu, rs, gamma, beta, x0 + (size_t)row * d);
    }
}

// LN' on the prompt rows only, in place on dx0 (the other rows' input gradients are not needed: frozen weights)
template <typename TS>
__device__ __forceinline__ void vis_prompt_rows_bwd_body(int blk, int B, int L, int P, int d, TS* __restrict__ dx0,
                                                         const float* __restrict__ prompt0, long pbs,
                                                         const float* __restrict__ gamma, const float* __restrict__ mean,
                                                         const float* __restrict__ rstd) {
    const int lane = threadIdx.x & 63;
    const int w = blk * 4 + (threadIdx.x >> 6);
    if (w >= B * P) return;
    const int b = w / P, p = w % P;
    const int row = b * L + 1 + p;
    Row g, x;
    TS* dxr = dx0 + (size_t)row * d;
    const float* pr = prompt0 + (size_t)b * pbs + (size_t)p * d;
    for_chunks(d, lane, [&](int i, int col) {
        g.v[i] = Elem<TS>::ld4(dxr + col);
        x.v[i] = *reinterpret_cast<const f32x4*>(pr + col);
    });
    ln_bwd_row(g, x, d, lane, mean[row], rstd[row], gamma);
    for_chunks(d, lane, [&](int i, int col) { Elem<TS>::st4(dxr + col, g.v[i]); });
}
template <typename TS>
__global__ __launch_bounds__(256) void vis_prompt_rows_bwd_kernel(int B, int L, int P, int d, TS* __restrict__ dx0,
                                                                 const float* __restrict__ prompt0, long pbs,
                                                                 const float* __restrict__ gamma, const float* __restrict__ mean,
                                                                 const float* __restrict__ rstd) {
    vis_prompt_rows_bwd_body<TS>(blockIdx.x, B, L, P, d, dx0, prompt0, pbs, gamma, mean, rstd);
}

// out[p, :] (+)= sum_b dx[(b*L + row0 + p), :]   — deterministic: 16 waves each sum a fixed subset of the batch in order, then
// the 16 partials are added in order.  Block = (prompt row p, 256-column chunk); lane = one float4 column.
template <typename TS>
__device__ __forceinline__ void rows_sum_body(int B, int L, const int* __restrict__ rs, int row0, int P, int d, const TS* __restrict__ dx,
                                              float* __restrict__ out, int accumulate, f32x4 (*part)[64]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int p = blockIdx.x, col = (blockIdx.y * 64 + lane) << 2;
    f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
    if (col < d) {
        const TS* src = dx + (size_t)(row0 + p) * d + col;
        for (int b = wave; b < B; b += 16) s += Elem<TS>::ld4(src + (rs ? (size_t)rs[b] : (size_t)b * L) * d);      // rs: ragged batch (row starts)
    }
    part[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && col < d) {
        f32x4 t = part[0][lane];
#pragma unroll
        for (int w = 1; w < 16; ++w) t += part[w][lane];
        float* o = out + (size_t)p * d + col;
        if (accumulate) t += *reinterpret_cast<const f32x4*>(o);
        *reinterpret_cast<f32x4*>(o) = t;
    }
}
template <typename TS>
__global__ __launch_bounds__(1024) void rows_sum_kernel(int B, int L, const int* __restrict__ rs, int row0, int P, int d, const TS* __restrict__ dx,
                                                       float* __restrict__ out, int accumulate) {
    __shared__ f32x4 part[16][64];
    rows_sum_body<TS>(B, L, rs, row0, P, d, dx, out, accumulate, part);
}
// the two towers' batch sums of one prompt layer in one launch (blockIdx.z = problem; grid.x / .y cover the larger P / d)
struct RowsSumP { int B, L, row0, P, d, accumulate; const int* rs; const void* dx; float* out; };
template <typename TS>
__global__ __launch_bounds__(1024) void rows_sum_pair_kernel(RowsSumP p0, RowsSumP p1) {
    __shared__ f32x4 part[16][64];
    const RowsSumP& q = blockIdx.z ? p1 : p0;
    if ((int)blockIdx.x >= q.P || (int)blockIdx.y * 256 >= q.d) return;
    rows_sum_body<TS>(q.B, q.L, q.rs, q.row0, q.P, q.d, (const TS*)q.dx, q.out, q.accumulate, part);
}

// dst[(b*P + p), 0:cols] = src[(b*L + row0 + p), 0:cols]  in 16-byte chunks (cols * sizeof(T) a multiple of 16)
__device__ __forceinline__ void gather_batch_rows_body(int blk, int B, int L, const int* __restrict__ rs, int row0, int P, int chunks, const uint4* __restrict__ src, long lds16,
                                                       uint4* __restrict__ dst, long ldd16) {
    const long t = (long)blk * blockDim.x + threadIdx.x;
    if (t >= (long)B * P * chunks) return;
    const int c = (int)(t % chunks);
    const long r = t / chunks;
    const long srow = (rs ? (long)rs[r / P] : (r / P) * L) + row0 + r % P;
    dst[r * ldd16 + c] = src[srow * lds16 + c];
}
__global__ __launch_bounds__(256) void gather_batch_rows_kernel(int B, int L, const int* __restrict__ rs, int row0, int P, int chunks, const uint4* __restrict__ src, long lds16,
                                                               uint4* __restrict__ dst, long ldd16) {
    gather_batch_rows_body(blockIdx.x, B, L, rs, row0, P, chunks, src, lds16, dst, ldd16);
}

// ---------------------------------------------------------------------------------------------------------
// text front end
template <typename TX>
__global__ __launch_bounds__(256) void txt_embed_kernel(int B, int L, const int* __restrict__ rs, int P, int d, const int64_t* __restrict__ ids,
                                                       const float* __restrict__ tok, const float* __restrict__ pos,
                                                       const float* __restrict__ ctx, long cbs, TX* __restrict__ x0,
                                                       float* __restrict__ omean, float* __restrict__ orstd, RowstatGuard guard, int pre) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);      // (b, l) over the [B, L] id matrix
    if (row >= B * L) return;
    const int b = row / L, l = row % L;
    size_t orow = row;
    if (rs && pre > 0) {                                       // shared prefix (attention.hip: attn_fwd_body): positions < pre are stored once, as rows 0 .. pre-1
        if (l < pre) {
            if (b != 0) return;
            orow = (size_t)l;
        } else {
            if (l - pre >= rs[b + 1] - rs[b]) return;
            orow = (size_t)rs[b] + (l - pre);
        }
    } else if (rs) {                                           // ragged batch: sample b owns rows rs[b] .. rs[b+1]-1; tokens behind them are not embedded
        if (l >= rs[b + 1] - rs[b]) return;
        orow = (size_t)rs[b] + l;
    }
    const float* src = (ctx && l >= 1 && l <= P) ? ctx + (size_t)b * cbs + (size_t)(l - 1) * d
                                                 : tok + (size_t)ids[row] * d;
    const float* pp = pos + (size_t)l * d;
    TX* o = x0 + orow * d;
    Row r;
    for_chunks(d, lane, [&](int i, int col) {
        r.v[i] = rounded4<TX>(*reinterpret_cast<const f32x4*>(src + col) + *reinterpret_cast<const f32x4*>(pp + col));
        Elem<TX>::st4(o + col, r.v[i]);
    });
    if (omean) {      // the statistics of the row as stored (the first block's folded ln_1)
        float mu, rs_;
        row_stats_1sweep(r, d, lane, mu, rs_, guard);
        if (lane == 0) { omean[orow] = mu; orstd[orow] = rs_; }
    }
}

template <typename TX>
__device__ __forceinline__ void prompt_add_body(int blk, int B, int L, const int* __restrict__ rs, int P, int d, TX* __restrict__ x,
                                                const float* __restrict__ pr, long pbs, float* __restrict__ omean, float* __restrict__ orstd, RowstatGuard guard) {
    const int lane = threadIdx.x & 63;
    const int w = blk * 4 + (threadIdx.x >> 6);
    if (w >= B * P) return;
    const int b = w / P, p = w % P;
    const size_t row = (rs ? (size_t)rs[b] : (size_t)b * L) + 1 + p;
    TX* xr = x + row * d;
    const float* s = pr + (size_t)b * pbs + (size_t)p * d;
    Row r;
    for_chunks(d, lane, [&](int i, int col) {
        r.v[i] = rounded4<TX>(Elem<TX>::ld4(xr + col) + *reinterpret_cast<const f32x4*>(s + col));
        Elem<TX>::st4(xr + col, r.v[i]);
    });
    if (omean) {      // the rewritten rows' statistics replace the ones the producing GEMM's epilogue left for them (LPI_EPI_RES_ROWSTATS)
        float mu, rs_;
        row_stats_1sweep(r, d, lane, mu, rs_, guard);
        if (lane == 0) { omean[row] = mu; orstd[row] = rs_; }
    }
}
template <typename TX>
__global__ __launch_bounds__(256) void prompt_add_kernel(int B, int L, const int* __restrict__ rs, int P, int d, TX* __restrict__ x,
                                                        const float* __restrict__ pr, long pbs, float* __restrict__ omean, float* __restrict__ orstd, RowstatGuard guard) {
    prompt_add_body<TX>(blockIdx.x, B, L, rs, P, d, x, pr, pbs, omean, orstd, guard);
}

// ---------------------------------------------------------------------------------------------------------
// pooled heads
template <typename TX, typename TY>
__device__ __forceinline__ void pool_ln_fwd_body(int blk, int B, int L, int d, const TX* __restrict__ x,
                                                 const int32_t* __restrict__ idx, const float* __restrict__ gamma,
                                                 const float* __restrict__ beta, TY* __restrict__ y, int ldy,
                                                 float* __restrict__ mean, float* __restrict__ rstd, float* __restrict__ raw) {
    const int lane = threadIdx.x & 63;
    const int b = blk * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    const size_t row = (size_t)b * L + (idx ? idx[b] : 0);
    Row r;
    for_chunks(d, lane, [&](int i, int col) { r.v[i] = Elem<TX>::ld4(x + row * d + col); });
    if (raw) for_chunks(d, lane, [&](int i, int col) { *reinterpret_cast<f32x4*>(raw + (size_t)b * d + col) = r.v[i]; });      // the row itself, f32: lpi_gather_rows
    float mu, rs;
    row_stats(r, d, lane, mu, rs);
    ln_apply_store<TY>(r, d, lane, mu, rs, gamma, beta, y + (size_t)b * ldy);
    if (lane == 0) { mean[b] = mu; rstd[b] = rs; }
}
template <typename TX, typename TY>
__global__ __launch_bounds__(256) void pool_ln_fwd_kernel(int B, int L, int d, const TX* __restrict__ x,
                                                         const int32_t* __restrict__ idx, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, TY* __restrict__ y, int ldy,
                                                         float* __restrict__ mean, float* __restrict__ rstd) {
    pool_ln_fwd_body<TX, TY>(blockIdx.x, B, L, d, x, idx, gamma, beta, y, ldy, mean, rstd, nullptr);
}

template <typename TCAST>
__device__ __forceinline__ void pool_ln_bwd_body(int blk, int B, int L, int d, const float* __restrict__ dy, int lddy,
                                                 const float* __restrict__ x, const int32_t* __restrict__ idx,
                                                 const float* __restrict__ gamma, const float* __restrict__ mean,
                                                 const float* __restrict__ rstd, float* __restrict__ dx,
                                                 TCAST* __restrict__ dx_cast) {
    const int lane = threadIdx.x & 63;
    const int b = blk * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    const size_t row = (size_t)b * L + (idx ? idx[b] : 0);
    Row g, xr;
    for_chunks(d, lane, [&](int i, int col) {
        g.v[i] = *reinterpret_cast<const f32x4*>(dy + (size_t)b * lddy + col);
        xr.v[i] = *reinterpret_cast<const f32x4*>(x + row * d + col);
    });
    ln_bwd_row(g, xr, d, lane, mean[b], rstd[b], gamma);
    for_chunks(d, lane, [&](int i, int col) {
        *reinterpret_cast<f32x4*>(dx + row * d + col) = g.v[i];
        if (dx_cast) Elem<TCAST>::st4(dx_cast + row * d + col, g.v[i]);
    });
}
template <typename TCAST>
__global__ __launch_bounds__(256) void pool_ln_bwd_kernel(int B, int L, int d, const float* __restrict__ dy, int lddy,
                                                         const float* __restrict__ x, const int32_t* __restrict__ idx,
                                                         const float* __restrict__ gamma, const float* __restrict__ mean,
                                                         const float* __restrict__ rstd, float* __restrict__ dx,
                                                         TCAST* __restrict__ dx_cast) {
    pool_ln_bwd_body<TCAST>(blockIdx.x, B, L, d, dy, lddy, x, idx, gamma, mean, rstd, dx, dx_cast);
}

// dst[b,:] (f32) = src[b*L + idx[b], :]  (idx NULL -> row 0)
template <typename TX>
__global__ __launch_bounds__(256) void gather_rows_kernel(int B, int L, int d, const TX* __restrict__ src, const int32_t* __restrict__ idx,
                                                         float* __restrict__ dst) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    const TX* s = src + ((size_t)b * L + (idx ? idx[b] : 0)) * d;
    float* o = dst + (size_t)b * d;
    for_chunks(d, lane, [&](int, int col) { *reinterpret_cast<f32x4*>(o + col) = Elem<TX>::ld4(s + col); });
}

// dst[b*L + idx[b], :] = src[b,:] (f32) and its cast copy; the rest of dst is the caller's (pre-zeroed)
template <typename TCAST>
__global__ __launch_bounds__(256) void scatter_rows_kernel(int B, int L, int d, const float* __restrict__ src, const int32_t* __restrict__ idx,
                                                          float* __restrict__ dst, TCAST* __restrict__ dst_cast) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    const size_t row = (size_t)b * L + (idx ? idx[b] : 0);
    const float* s = src + (size_t)b * d;
    for_chunks(d, lane, [&](int, int col) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(s + col);
        if (dst) *reinterpret_cast<f32x4*>(dst + row * d + col) = v;
        if (dst_cast) Elem<TCAST>::st4(dst_cast + row * d + col, v);
    });
}

__device__ __forceinline__ void l2norm_fwd_body(int blk, int B, int E, const float* __restrict__ x, int ldx,
                                                float* __restrict__ y, int ldy, float* __restrict__ inv_norm) {
    const int lane = threadIdx.x & 63;
    const int b = blk * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    Row r;
    float ss = 0.f;
    for_chunks(E, lane, [&](int i, int col) {
        r.v[i] = *reinterpret_cast<const f32x4*>(x + (size_t)b * ldx + col);
        ss += hsum(r.v[i] * r.v[i]);
    });
    const float inv = 1.0f / sqrtf(wave_sum(ss));
    for_chunks(E, lane, [&](int i, int col) { *reinterpret_cast<f32x4*>(y + (size_t)b * ldy + col) = r.v[i] * inv; });
    if (lane == 0) inv_norm[b] = inv;
}
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(int B, int E, const float* __restrict__ x, int ldx,
                                                        float* __restrict__ y, int ldy, float* __restrict__ inv_norm) {
    l2norm_fwd_body(blockIdx.x, B, E, x, ldx, y, ldy, inv_norm);
}

__device__ __forceinline__ void l2norm_bwd_body(int blk, int B, int E, const float* __restrict__ y, int ldy,
                                                const float* __restrict__ dy, int lddy, const float* __restrict__ inv_norm,
                                                float* __restrict__ dx, int lddx, bf16_t* __restrict__ dx16) {
    const int lane = threadIdx.x & 63;
    const int b = blk * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    Row yr, g;
    float dot = 0.f;
    for_chunks(E, lane, [&](int i, int col) {
        yr.v[i] = *reinterpret_cast<const f32x4*>(y + (size_t)b * ldy + col);
        g.v[i] = *reinterpret_cast<const f32x4*>(dy + (size_t)b * lddy + col);
        dot += hsum(yr.v[i] * g.v[i]);
    });
    dot = wave_sum(dot);
    const float inv = inv_norm[b];
    for_chunks(E, lane, [&](int i, int col) {
        const f32x4 o = (g.v[i] - yr.v[i] * dot) * inv;
        *reinterpret_cast<f32x4*>(dx + (size_t)b * lddx + col) = o;
        if (dx16) Elem<bf16_t>::st4(dx16 + (size_t)b * lddx + col, o);      // the bf16 operand of the head's dgrad GEMM (lpi_cast of dx)
    });
}
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(int B, int E, const float* __restrict__ y, int ldy,
                                                        const float* __restrict__ dy, int lddy, const float* __restrict__ inv_norm,
                                                        float* __restrict__ dx, int lddx) {
    l2norm_bwd_body(blockIdx.x, B, E, y, ldy, dy, lddy, inv_norm, dx, lddx, nullptr);
}

__global__ void eot_index_kernel(int B, int L, const int64_t* __restrict__ ids, int32_t* __restrict__ idx) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    int64_t best = ids[(size_t)b * L];
    int bi = 0;
    for (int l = 1; l < L; ++l) {
        const int64_t v = ids[(size_t)b * L + l];
        if (v > best) { best = v; bi = l; }   // first maximum, as torch.argmax
    }
    idx[b] = bi;
}

template <typename TS, typename TD>
__global__ __launch_bounds__(256) void cast_kernel(long n4, const TS* __restrict__ s, TD* __restrict__ dst) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride)
        Elem<TD>::st4(dst + (i << 2), Elem<TS>::ld4(s + (i << 2)));
}

template <typename T>
__global__ __launch_bounds__(256) void transpose_kernel(int rows, int cols, const T* __restrict__ s, int lds, T* __restrict__ dst, int ldd) {
    __shared__ T tile[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int j = ty; j < 32; j += 8) {
        const int r = by + j, c = bx + tx;
        if (r < rows && c < cols) tile[j][tx] = s[(size_t)r * lds + c];
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int r = bx + j, c = by + tx;  // dst is [cols, rows]
        if (r < cols && c < rows) dst[(size_t)r * ldd + c] = tile[tx][j];
    }
}

// two transposes in one launch (blockIdx.z = problem): the two feature matrices of the contrastive loss's gradient GEMMs
template <typename T>
__global__ __launch_bounds__(256) void transpose2_kernel(int rows0, int cols0, const T* __restrict__ s0, int lds0, T* __restrict__ d0, int ldd0,
                                                        int rows1, int cols1, const T* __restrict__ s1, int lds1, T* __restrict__ d1, int ldd1) {
    __shared__ T tile[32][33];
    const bool z = blockIdx.z != 0;
    const int rows = z ? rows1 : rows0, cols = z ? cols1 : cols0, lds = z ? lds1 : lds0, ldd = z ? ldd1 : ldd0;
    const T* s = z ? s1 : s0;
    T* dst = z ? d1 : d0;
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    if (bx >= cols || by >= rows) return;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int j = ty; j < 32; j += 8) {
        const int r = by + j, c = bx + tx;
        if (r < rows && c < cols) tile[j][tx] = s[(size_t)r * lds + c];
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int r = bx + j, c = by + tx;  // dst is [cols, rows]
        if (r < cols && c < rows) dst[(size_t)r * ldd + c] = tile[tx][j];
    }
}

// ---------------------------------------------------------------------------------------------------------
// lpi_row_jobs: up to LPI_ROW_JOBS_MAX small row kernels of one dependency level in ONE launch.  The tail of a training step (pooled
// heads, L2 norms, prompt rows: everything that works on B or B*P rows) is a chain of ~5 us launches, one per tower and op; the towers'
// launches of the same op (and independent ops of one tower) are independent.  Blocks are dealt to the jobs in order (nb[i] blocks of job i); each
// job runs the BODY of the single-op kernel above with its own block index, so every result is bit for bit the single launch's.
struct RowJobs { lpi_row_job j[LPI_ROW_JOBS_MAX]; int nb[LPI_ROW_JOBS_MAX]; int n; RowstatGuard guard; };

__device__ __forceinline__ void run_row_job(const lpi_row_job& q, int blk, RowstatGuard guard) {
    switch (q.op) {
    case LPI_ROWOP_POOL_LN_FWD: {
#define PLF(TX, TY) pool_ln_fwd_body<TX, TY>(blk, q.B, q.L, q.d, (const TX*)q.a, q.idx, q.gamma, q.beta, (TY*)q.out, q.ld_c, q.mean, q.rstd, (float*)q.out2)
        if (q.dt_a == LPI_F32 && q.dt_b == LPI_F32) PLF(float, float);
        else if (q.dt_a == LPI_F32 && q.dt_b == LPI_BF16) PLF(float, bf16_t);
        else if (q.dt_a == LPI_F16 && q.dt_b == LPI_BF16) PLF(f16_t, bf16_t);
        else if (q.dt_a == LPI_F32 && q.dt_b == LPI_F16) PLF(float, f16_t);
        else PLF(f16_t, f16_t);
#undef PLF
        break;
    }
    case LPI_ROWOP_L2NORM_FWD:
        l2norm_fwd_body(blk, q.B, q.d, (const float*)q.a, q.ld_a, (float*)q.out, q.ld_c, q.mean);
        break;
    case LPI_ROWOP_L2NORM_BWD:
        l2norm_bwd_body(blk, q.B, q.d, (const float*)q.a, q.ld_a, (const float*)q.b, q.ld_b, q.mean_in, (float*)q.out, q.ld_c, (bf16_t*)q.out2);
        break;
    case LPI_ROWOP_POOL_LN_BWD:
        if (q.dt_b == LPI_BF16) pool_ln_bwd_body<bf16_t>(blk, q.B, q.L, q.d, (const float*)q.a, q.ld_a, (const float*)q.b, q.idx, q.gamma, q.mean_in, q.rstd_in, (float*)q.out, (bf16_t*)q.out2);
        else pool_ln_bwd_body<float>(blk, q.B, q.L, q.d, (const float*)q.a, q.ld_a, (const float*)q.b, q.idx, q.gamma, q.mean_in, q.rstd_in, (float*)q.out, (float*)q.out2);
        break;
    case LPI_ROWOP_LN_BWD: {      // f32 x rows (the pooled rows of the last block); dy / cast in dt_a / dt_b
#define LNBJ(TDY, TC, NC) ln_bwd_body<float, TDY, TC, NC>(blk, q.B, q.d, (const TDY*)q.a, q.ld_a, (const float*)q.b, q.ld_b, q.gamma, q.mean_in, q.rstd_in, (float*)q.out, q.ld_c, (TC*)q.out2, q.ld_c, q.flag, 0, 0, 0, nullptr)
        const int nc = (q.d + 255) / 256;
        if (q.dt_a == LPI_BF16 && q.dt_b == LPI_BF16) { if (nc <= 1) LNBJ(bf16_t, bf16_t, 1); else if (nc == 2) LNBJ(bf16_t, bf16_t, 2); else if (nc == 3) LNBJ(bf16_t, bf16_t, 3); else if (nc == 4) LNBJ(bf16_t, bf16_t, 4); else LNBJ(bf16_t, bf16_t, 8); }
        else { if (nc <= 1) LNBJ(float, float, 1); else if (nc == 2) LNBJ(float, float, 2); else if (nc == 3) LNBJ(float, float, 3); else if (nc == 4) LNBJ(float, float, 4); else LNBJ(float, float, 8); }
#undef LNBJ
        break;
    }
    case LPI_ROWOP_SCATTER_ADD: {      // dst[b*L + idx[b], :] += src[b, :]: one block per sample (the arithmetic of scatter_add_rows_kernel, attn_pooled.hip)
        const int b = blk;
        if (b >= q.B) break;
        const int row = q.idx ? q.idx[b] : 0;
        auto body = [&](auto* o0, const auto* i0) {
            typedef typename std::remove_const<typename std::remove_pointer<decltype(o0)>::type>::type T;
            T* o = o0 + ((size_t)b * q.L + row) * q.ld_c;
            const T* i = i0 + (size_t)b * q.ld_a;
            for (int c = threadIdx.x * 4; c < q.d; c += 256 * 4) {
                f32x4 a = Elem<T>::ld4(o + c), v = Elem<T>::ld4(i + c);
                a[0] += v[0]; a[1] += v[1]; a[2] += v[2]; a[3] += v[3];
                Elem<T>::st4(o + c, a);
            }
        };
        if (q.dt_a == LPI_BF16) body((bf16_t*)q.out, (const bf16_t*)q.a);
        else body((float*)q.out, (const float*)q.a);
        break;
    }
    case LPI_ROWOP_GATHER_BATCH_ROWS:      // d = 16-byte chunks per row, ld_a / ld_c in 16-byte units
        gather_batch_rows_body(blk, q.B, q.L, q.row_start, q.row0, q.P, q.d, (const uint4*)q.a, q.ld_a, (uint4*)q.out, q.ld_c);
        break;
    case LPI_ROWOP_PROMPT_ADD:
        if (q.dt_a == LPI_F16) prompt_add_body<f16_t>(blk, q.B, q.L, q.row_start, q.P, q.d, (f16_t*)q.out, (const float*)q.a, q.bstride, q.mean, q.rstd, guard);
        else prompt_add_body<float>(blk, q.B, q.L, q.row_start, q.P, q.d, (float*)q.out, (const float*)q.a, q.bstride, q.mean, q.rstd, guard);
        break;
    case LPI_ROWOP_LN_BWD_ROWS_H16: {      // compact bf16 dy [B*P, d], fp16 x, bf16 gradient stream (in/out): ln_bwd_h16_kernel with the row map
#define LNH(NC) ln_bwd_h16_body<NC>(blk, q.B * q.P, q.d, (const bf16_t*)q.a, q.ld_a, (const f16_t*)q.b, q.ld_b, q.gamma, q.mean_in, q.rstd_in, (bf16_t*)q.out2, q.ld_c, q.flag, q.P, q.L, q.row0, q.row_start)
        const int nc = (q.d + 255) / 256;
        if (nc <= 1) LNH(1); else if (nc == 2) LNH(2); else if (nc == 3) LNH(3); else LNH(4);
#undef LNH
        break;
    }
    case LPI_ROWOP_VIS_PROMPT_ROWS_BWD:
        if (q.dt_a == LPI_BF16) vis_prompt_rows_bwd_body<bf16_t>(blk, q.B, q.L, q.P, q.d, (bf16_t*)q.out, (const float*)q.a, q.bstride, q.gamma, q.mean_in, q.rstd_in);
        else vis_prompt_rows_bwd_body<float>(blk, q.B, q.L, q.P, q.d, (float*)q.out, (const float*)q.a, q.bstride, q.gamma, q.mean_in, q.rstd_in);
        break;
    default: break;
    }
}

__global__ __launch_bounds__(256) void row_jobs_kernel(const RowJobs js) {
    int b = blockIdx.x;
#pragma unroll
    for (int i = 0; i < LPI_ROW_JOBS_MAX; ++i) {
        if (i >= js.n) return;
        if (b < js.nb[i]) { run_row_job(js.j[i], b, js.guard); return; }
        b -= js.nb[i];
    }
}

inline bool bad_row_dim(int d) { return d <= 0 || (d & 3) || d > MAXC * 256; }
inline int rows_grid(long rows) { return (int)((rows + 3) / 4); }

}  // namespace

#define S(stream) ((hipStream_t)(stream))

// smallest instantiated chunk bound that covers d: ceil(d / 256) rounded up to one of 1, 2, 3, 4, 8
#define LN_NC_SWITCH(d, CALL)                      \
    do {                                           \
        const int nc__ = ((d) + 255) / 256;        \
        if (nc__ <= 1) { CALL(1); }                \
        else if (nc__ == 2) { CALL(2); }           \
        else if (nc__ == 3) { CALL(3); }           \
        else if (nc__ == 4) { CALL(4); }           \
        else { CALL(8); }                          \
    } while (0)

static int nc_of(int d) { const int n = (d + 255) / 256; return n <= 4 ? n : 0; }
static bool ln_h16_ok(int d, int ld0, int ld1, const void* p0, const void* p1) {
    return !(d & 7) && !(ld0 & 7) && !(ld1 & 7) && !(((uintptr_t)p0 | (uintptr_t)p1) & 15) && nc_of(d) != 0;
}
extern "C" int lpi_layernorm_fwd(int dtype, int x_dtype, int rows, int d, const void* x, int ldx, const float* gamma, const float* beta,
                                 void* y, int ldy, float* mean, float* rstd, void* stream) {
    if (!y) {      // statistics only (fp16 stream): mean / rstd of every row, nothing else written
        if (!x || !mean || !rstd || rows <= 0 || bad_row_dim(d) || x_dtype != LPI_F16 || !ln_h16_ok(d, ldx, 0, x, nullptr)) return LPI_EINVAL;
#define LNS(NC) LPI_LAUNCH((ln_fwd_h16_kernel<NC, void>), dim3((rows + 7) / 8), dim3(256), 0, S(stream), rows, d, (const f16_t*)x, ldx, gamma, beta, (void*)nullptr, 0, mean, rstd)
        LN_NC_SWITCH(d, LNS);
#undef LNS
        LPI_CHECK_LAST();
        return 0;
    }
    if (!x || !gamma || !beta || !y || !mean || !rstd || rows <= 0 || bad_row_dim(d) || (ldx & 3) || (ldy & 3)) return LPI_EINVAL;
    dim3 g(rows_grid(rows)), b(256);
#define LNF(TX, TY, NC) LPI_LAUNCH((ln_fwd_kernel<TX, TY, NC>), g, b, 0, S(stream), rows, d, (const TX*)x, ldx, gamma, beta, (TY*)y, ldy, mean, rstd)
#define LNF_FF(NC) LNF(float, float, NC)
#define LNF_FB(NC) LNF(float, bf16_t, NC)
#define LNF_HB(NC) LNF(f16_t, bf16_t, NC)
#define LNF_H16(NC) LPI_LAUNCH(ln_fwd_h16_kernel<NC>, dim3((rows + 7) / 8), b, 0, S(stream), rows, d, (const f16_t*)x, ldx, gamma, beta, (bf16_t*)y, ldy, mean, rstd)
#define LNF_H16H(NC) LPI_LAUNCH((ln_fwd_h16_kernel<NC, f16_t>), dim3((rows + 7) / 8), b, 0, S(stream), rows, d, (const f16_t*)x, ldx, gamma, beta, (f16_t*)y, ldy, mean, rstd)
#define LNF_HH(NC) LNF(f16_t, f16_t, NC)
#define LNF_FH(NC) LNF(float, f16_t, NC)
    if (dtype == LPI_F32 && x_dtype == LPI_F32) LN_NC_SWITCH(d, LNF_FF);
    else if (dtype == LPI_F16 && x_dtype == LPI_F16 && !(d & 7) && !(ldx & 7) && !(ldy & 7) && !(((uintptr_t)x | (uintptr_t)y) & 15))
        LN_NC_SWITCH(d, LNF_H16H);          // f16 operand mode: fp16 stream in, fp16 operand out
    else if (dtype == LPI_F16 && x_dtype == LPI_F16) LN_NC_SWITCH(d, LNF_HH);
    else if (dtype == LPI_F16 && x_dtype == LPI_F32) LN_NC_SWITCH(d, LNF_FH);
    else if (dtype == LPI_BF16 && x_dtype == LPI_F32) LN_NC_SWITCH(d, LNF_FB);
    else if (dtype == LPI_BF16 && x_dtype == LPI_F16 && !(d & 7) && !(ldx & 7) && !(ldy & 7) && !(((uintptr_t)x | (uintptr_t)y) & 15))
        LN_NC_SWITCH(d, LNF_H16);
    else if (dtype == LPI_BF16 && x_dtype == LPI_F16) LN_NC_SWITCH(d, LNF_HB);
    else return LPI_EINVAL;
#undef LNF
#undef LNF_FF
#undef LNF_FB
#undef LNF_HB
#undef LNF_H16
#undef LNF_H16H
#undef LNF_HH
#undef LNF_FH
    LPI_CHECK_LAST();
    return 0;
}

static int ln_bwd_impl(int dy_dtype, int cast_dtype, int x_dtype, int rows, int d, const void* dy, int lddy, const void* x, int ldx,
                       const float* gamma, const float* mean, const float* rstd, float* dx, int lddx, void* dx_cast,
                       int ldcast, int accumulate, int mapP, int mapL, int map0, const int* map_rs, void* stream) {
    if (!dy || !x || !gamma || !mean || !rstd || (!dx && !dx_cast) || rows <= 0 || bad_row_dim(d) || (lddy & 3) || (ldx & 3) || (lddx & 3) || (ldcast & 3))
        return LPI_EINVAL;
    dim3 g(rows_grid(rows)), b(256);
#define LNB(TX, TDY, TC, NC) LPI_LAUNCH((ln_bwd_kernel<TX, TDY, TC, NC>), g, b, 0, S(stream), rows, d, (const TDY*)dy, lddy, (const TX*)x, ldx, gamma, mean, rstd, dx, lddx, (TC*)dx_cast, ldcast, accumulate, mapP, mapL, map0, map_rs)
#define LNB_HBB(NC) LNB(f16_t, bf16_t, bf16_t, NC)
#define LNB_FFF(NC) LNB(float, float, float, NC)
#define LNB_FFB(NC) LNB(float, float, bf16_t, NC)
#define LNB_FBB(NC) LNB(float, bf16_t, bf16_t, NC)
#define LNB_FBF(NC) LNB(float, bf16_t, float, NC)
#define LNB_H16(NC) LPI_LAUNCH(ln_bwd_h16_kernel<NC>, dim3((rows + 7) / 8), b, 0, S(stream), rows, d, (const bf16_t*)dy, lddy, (const f16_t*)x, ldx, gamma, mean, rstd, (bf16_t*)dx_cast, ldcast, accumulate, mapP, mapL, map0, map_rs)
    if (x_dtype == LPI_F16) {
        if (dy_dtype == LPI_BF16 && cast_dtype == LPI_BF16 && !dx && !(d & 7) && !(lddy & 7) && !(ldx & 7) && !(ldcast & 7) &&
            !(((uintptr_t)dy | (uintptr_t)x | (uintptr_t)dx_cast) & 15))
            LN_NC_SWITCH(d, LNB_H16);
        else if (dy_dtype == LPI_BF16 && cast_dtype == LPI_BF16) LN_NC_SWITCH(d, LNB_HBB);
        else return LPI_EINVAL;
    } else if (x_dtype != LPI_F32) return LPI_EINVAL;
    else if (dy_dtype == LPI_F32 && cast_dtype == LPI_F32) LN_NC_SWITCH(d, LNB_FFF);
    else if (dy_dtype == LPI_F32 && cast_dtype == LPI_BF16) LN_NC_SWITCH(d, LNB_FFB);
    else if (dy_dtype == LPI_BF16 && cast_dtype == LPI_BF16) LN_NC_SWITCH(d, LNB_FBB);
    else if (dy_dtype == LPI_BF16 && cast_dtype == LPI_F32) LN_NC_SWITCH(d, LNB_FBF);
    else return LPI_EINVAL;
#undef LNB
#undef LNB_HBB
#undef LNB_FFF
#undef LNB_FFB
#undef LNB_FBB
#undef LNB_FBF
#undef LNB_H16
    LPI_CHECK_LAST();
    return 0;
}

extern "C" int lpi_layernorm_bwd(int dy_dtype, int cast_dtype, int x_dtype, int rows, int d, const void* dy, int lddy, const void* x, int ldx,
                                 const float* gamma, const float* mean, const float* rstd, float* dx, int lddx, void* dx_cast,
                                 int ldcast, int accumulate, void* stream) {
    return ln_bwd_impl(dy_dtype, cast_dtype, x_dtype, rows, d, dy, lddy, x, ldx, gamma, mean, rstd, dx, lddx, dx_cast, ldcast, accumulate, 0, 0, 0, nullptr, stream);
}

// LayerNorm backward of P rows per sample only: dy is compact [B*P, d]; x, mean / rstd and the gradient stream are the full [B*L, .]
// arrays, touched at rows b*L + row0 + p.  The first block's backward needs the input gradient at the prompt rows alone (nothing
// upstream of the prompt slots is trainable: sprompt.py:230-237).
extern "C" int lpi_layernorm_bwd_rows_varlen(int dy_dtype, int cast_dtype, int x_dtype, int B, int L, const int32_t* row_start, int row0, int P, int d,
                                             const void* dy, int lddy, const void* x, int ldx, const float* gamma, const float* mean, const float* rstd,
                                             float* dx, int lddx, void* dx_cast, int ldcast, int accumulate, void* stream) {
    if (B <= 0 || L <= 0 || P <= 0 || row0 < 0 || row0 + P > L) return LPI_EINVAL;
    return ln_bwd_impl(dy_dtype, cast_dtype, x_dtype, B * P, d, dy, lddy, x, ldx, gamma, mean, rstd, dx, lddx, dx_cast, ldcast, accumulate, P, L, row0,
                       row_start, stream);
}
extern "C" int lpi_layernorm_bwd_rows(int dy_dtype, int cast_dtype, int x_dtype, int B, int L, int row0, int P, int d, const void* dy, int lddy,
                                      const void* x, int ldx, const float* gamma, const float* mean, const float* rstd, float* dx, int lddx,
                                      void* dx_cast, int ldcast, int accumulate, void* stream) {
    return lpi_layernorm_bwd_rows_varlen(dy_dtype, cast_dtype, x_dtype, B, L, nullptr, row0, P, d, dy, lddy, x, ldx, gamma, mean, rstd, dx, lddx,
                                         dx_cast, ldcast, accumulate, stream);
}

// ---- two LayerNorms in one launch (lpi_layernorm_fwd_pair / _bwd_pair): the half-wave 16-byte kernels only; anything else runs as two launches
template <typename TY>
static int ln_fwd_pair_launch(const LnFwdP& a, const LnFwdP& b, hipStream_t s) {
    const int nb0 = (a.rows + 7) / 8, nb1 = (b.rows + 7) / 8;
    const int n0 = nc_of(a.d), n1 = nc_of(b.d);
#define LNP(A, B) LPI_LAUNCH((ln_fwd_h16_pair_kernel<A, B, TY>), dim3(nb0 + nb1), dim3(256), 0, s, a, b, nb0); LPI_CHECK_LAST(); return 0
    switch (n0 * 8 + n1) {
    case 1 * 8 + 1: LNP(1, 1);
    case 2 * 8 + 1: LNP(2, 1);
    case 2 * 8 + 2: LNP(2, 2);
    case 3 * 8 + 1: LNP(3, 1);
    case 3 * 8 + 2: LNP(3, 2);
    case 3 * 8 + 3: LNP(3, 3);
    case 4 * 8 + 1: LNP(4, 1);
    case 4 * 8 + 2: LNP(4, 2);
    case 4 * 8 + 3: LNP(4, 3);
    case 4 * 8 + 4: LNP(4, 4);
    }
#undef LNP
    return LPI_ENOSYS;
}
extern "C" int lpi_layernorm_fwd_pair(int dtype, int x_dtype, const lpi_ln_fwd_desc* d, void* stream) {
    if (!d) return LPI_EINVAL;
    if (!d[0].y && !d[1].y) {      // statistics only, both problems (y = NULL): see lpi_layernorm_fwd
        for (int i = 0; i < 2; ++i)
            if (!d[i].x || !d[i].mean || !d[i].rstd || d[i].rows <= 0 || bad_row_dim(d[i].d) || x_dtype != LPI_F16 || !ln_h16_ok(d[i].d, d[i].ldx, 0, d[i].x, nullptr))
                return LPI_EINVAL;
        const int o = nc_of(d[0].d) >= nc_of(d[1].d) ? 0 : 1;
        LnFwdP p[2];
        for (int i = 0; i < 2; ++i) {
            const lpi_ln_fwd_desc& q = d[i ^ o];
            p[i] = LnFwdP{q.rows, q.d, q.ldx, 0, (const f16_t*)q.x, q.gamma, q.beta, nullptr, q.mean, q.rstd};
        }
        const int rc = ln_fwd_pair_launch<void>(p[0], p[1], S(stream));
        if (rc != LPI_ENOSYS) return rc;
        for (int i = 0; i < 2; ++i)
            if (int e = lpi_layernorm_fwd(dtype, x_dtype, d[i].rows, d[i].d, d[i].x, d[i].ldx, d[i].gamma, d[i].beta, nullptr, 0, d[i].mean, d[i].rstd, stream)) return e;
        return 0;
    }
    for (int i = 0; i < 2; ++i)
        if (!d[i].x || !d[i].gamma || !d[i].beta || !d[i].y || !d[i].mean || !d[i].rstd || d[i].rows <= 0 || bad_row_dim(d[i].d)) return LPI_EINVAL;
    const bool fast = x_dtype == LPI_F16 && (dtype == LPI_BF16 || dtype == LPI_F16) && ln_h16_ok(d[0].d, d[0].ldx, d[0].ldy, d[0].x, d[0].y) &&
                      ln_h16_ok(d[1].d, d[1].ldx, d[1].ldy, d[1].x, d[1].y);
    if (fast) {
        const int o = nc_of(d[0].d) >= nc_of(d[1].d) ? 0 : 1;      // the wider row first (the instantiated pairs have NC0 >= NC1)
        LnFwdP p[2];
        for (int i = 0; i < 2; ++i) {
            const lpi_ln_fwd_desc& q = d[i ^ o];
            p[i] = LnFwdP{q.rows, q.d, q.ldx, q.ldy, (const f16_t*)q.x, q.gamma, q.beta, q.y, q.mean, q.rstd};
        }
        const int rc = dtype == LPI_BF16 ? ln_fwd_pair_launch<bf16_t>(p[0], p[1], S(stream)) : ln_fwd_pair_launch<f16_t>(p[0], p[1], S(stream));
        if (rc != LPI_ENOSYS) return rc;
    }
    for (int i = 0; i < 2; ++i)
        if (int e = lpi_layernorm_fwd(dtype, x_dtype, d[i].rows, d[i].d, d[i].x, d[i].ldx, d[i].gamma, d[i].beta, d[i].y, d[i].ldy, d[i].mean, d[i].rstd, stream))
            return e;
    return 0;
}
static int ln_bwd_pair_launch(const LnBwdP& a, const LnBwdP& b, hipStream_t s) {
    const int nb0 = (a.rows + 7) / 8, nb1 = (b.rows + 7) / 8;
    const int n0 = nc_of(a.d), n1 = nc_of(b.d);
#define LNP(A, B) LPI_LAUNCH((ln_bwd_h16_pair_kernel<A, B>), dim3(nb0 + nb1), dim3(256), 0, s, a, b, nb0); LPI_CHECK_LAST(); return 0
    switch (n0 * 8 + n1) {
    case 1 * 8 + 1: LNP(1, 1);
    case 2 * 8 + 1: LNP(2, 1);
    case 2 * 8 + 2: LNP(2, 2);
    case 3 * 8 + 1: LNP(3, 1);
    case 3 * 8 + 2: LNP(3, 2);
    case 3 * 8 + 3: LNP(3, 3);
    case 4 * 8 + 1: LNP(4, 1);
    case 4 * 8 + 2: LNP(4, 2);
    case 4 * 8 + 3: LNP(4, 3);
    case 4 * 8 + 4: LNP(4, 4);
    }
#undef LNP
    return LPI_ENOSYS;
}
extern "C" int lpi_layernorm_bwd_pair(int dy_dtype, int cast_dtype, int x_dtype, const lpi_ln_bwd_desc* d, void* stream) {
    if (!d) return LPI_EINVAL;
    for (int i = 0; i < 2; ++i)
        if (!d[i].dy || !d[i].x || !d[i].gamma || !d[i].mean || !d[i].rstd || (!d[i].dx && !d[i].dx_cast) || d[i].rows <= 0 || bad_row_dim(d[i].d))
            return LPI_EINVAL;
    bool fast = x_dtype == LPI_F16 && dy_dtype == LPI_BF16 && cast_dtype == LPI_BF16;
    for (int i = 0; i < 2 && fast; ++i)
        fast = !d[i].dx && d[i].dx_cast && ln_h16_ok(d[i].d, d[i].lddy, d[i].ldx, d[i].dy, d[i].x) && !(d[i].ldcast & 7) && !(((uintptr_t)d[i].dx_cast) & 15);
    if (fast) {
        const int o = nc_of(d[0].d) >= nc_of(d[1].d) ? 0 : 1;
        LnBwdP p[2];
        for (int i = 0; i < 2; ++i) {
            const lpi_ln_bwd_desc& q = d[i ^ o];
            p[i] = LnBwdP{q.rows, q.d, q.lddy, q.ldx, q.ldcast, q.accumulate, (const bf16_t*)q.dy, (const f16_t*)q.x, q.gamma, q.mean, q.rstd, (bf16_t*)q.dx_cast};
        }
        const int rc = ln_bwd_pair_launch(p[0], p[1], S(stream));
        if (rc != LPI_ENOSYS) return rc;
    }
    for (int i = 0; i < 2; ++i)
        if (int e = lpi_layernorm_bwd(dy_dtype, cast_dtype, x_dtype, d[i].rows, d[i].d, d[i].dy, d[i].lddy, d[i].x, d[i].ldx, d[i].gamma, d[i].mean, d[i].rstd,
                                      d[i].dx, d[i].lddx, d[i].dx_cast, d[i].ldcast, d[i].accumulate, stream))
            return e;
    return 0;
}

extern "C" int lpi_patchify(int dtype, int B, int R, int ps, const float* image, void* cols, int ldcols, void* stream) {
    if (!image || !cols || B <= 0 || ps <= 0 || R % ps) return LPI_EINVAL;
    const int G = R / ps, K = 3 * ps * ps;
    const int esz = dtype == LPI_F32 ? 4 : 2;
    const int bk = 128 / esz;
    const int Kp = (K + bk - 1) / bk * bk;
    if (ldcols < Kp || (ldcols & 3)) return LPI_EINVAL;
    const long total = (long)B * G * G * (Kp >> 2);
    dim3 g((unsigned)((total + 255) / 256)), b(256);
    const bool vec4 = (ps & 3) == 0 && (R & 3) == 0 && ((uintptr_t)image & 15) == 0;
#define PATCHIFY(T) do { if (vec4) LPI_LAUNCH((patchify_kernel<T, true>), g, b, 0, S(stream), B, R, ps, G, Kp, image, (T*)cols, ldcols); \
                         else LPI_LAUNCH((patchify_kernel<T, false>), g, b, 0, S(stream), B, R, ps, G, Kp, image, (T*)cols, ldcols); } while (0)
    if (dtype == LPI_F32) PATCHIFY(float);
    else if (dtype == LPI_BF16) PATCHIFY(bf16_t);
    else if (dtype == LPI_F16) PATCHIFY(f16_t);
    else return LPI_EINVAL;
#undef PATCHIFY
    LPI_CHECK_LAST();
    return 0;
}

extern "C" int lpi_patchify_u8(int dtype, int B, int R, int ps, const uint8_t* image, const float* lut, void* cols, int ldcols, void* stream) {
    if (!image || !lut || !cols || B <= 0 || ps <= 0 || R % ps || (ps & 3) || (R & 3) || ((uintptr_t)image & 3)) return LPI_EINVAL;
    const int G = R / ps, K = 3 * ps * ps;
    const int esz = dtype == LPI_F32 ? 4 : 2;
    const int bk = 128 / esz;
    const int Kp = (K + bk - 1) / bk * bk;
    if (ldcols < Kp || (ldcols & 3)) return LPI_EINVAL;
    const long total = (long)B * G * G * (Kp >> 2);
    dim3 g((unsigned)((total + 255) / 256)), b(256);
    if (dtype == LPI_F32) LPI_LAUNCH(patchify_u8_kernel<float>, g, b, 0, S(stream), B, R, ps, G, Kp, image, lut, (float*)cols, ldcols);
    else if (dtype == LPI_BF16) LPI_LAUNCH(patchify_u8_kernel<bf16_t>, g, b, 0, S(stream), B, R, ps, G, Kp, image, lut, (bf16_t*)cols, ldcols);
    else if (dtype == LPI_F16) LPI_LAUNCH(patchify_u8_kernel<f16_t>, g, b, 0, S(stream), B, R, ps, G, Kp, image, lut, (f16_t*)cols, ldcols);
    else return LPI_EINVAL;
    LPI_CHECK_LAST();
    return 0;
}

extern "C" int lpi_vis_assemble_fwd(int x_dtype, int B, int G2, int P, int d, const float* patch_emb, int ldpe, const float* cls, const float* pos,
                                    const float* prompt0, long prompt_bstride, const float* gamma, const float* beta, void* x0,
                                    float* mean, float* rstd, float* out_mean, float* out_rstd, void* stream) {
    if ((out_mean != nullptr) != (out_rstd != nullptr)) return LPI_EINVAL;
    if (!patch_emb || !cls || !pos || !gamma || !beta || !x0 || !mean || !rstd || B <= 0 || G2 <= 0 || P < 0 || bad_row_dim(d) || (ldpe & 3))
        return LPI_EINVAL;
    if (P > 0 && (!prompt0 || (prompt_bstride & 3))) return LPI_EINVAL;
    const long rows = (long)B * (1 + P + G2);
    if (x_dtype == LPI_F32)
        LPI_LAUNCH(vis_assemble_fwd_kernel<float>, dim3(rows_grid(rows)), dim3(256), 0, S(stream), B, G2, P, d, patch_emb, ldpe, cls, pos,
                   prompt0, prompt_bstride, gamma, beta, (float*)x0, mean, rstd, out_mean, out_rstd, t_rowstat_guard);
    else if (x_dtype == LPI_F16)
        LPI_LAUNCH(vis_assemble_fwd_kernel<f16_t>, dim3(rows_grid(rows)), dim3(256), 0, S(stream), B, G2, P, d, patch_emb, ldpe, cls, pos,
                   prompt0, prompt_bstride, gamma, beta, (f16_t*)x0, mean, rstd, out_mean, out_rstd, t_rowstat_guard);
    else
        return LPI_EINVAL;
    LPI_CHECK_LAST();
    return 0;
}

extern "C" int lpi_gather_batch_rows_varlen(int dtype, int B, int L, const int32_t* row_start, int row0, int P, int cols, const void* src, int ld_src,
                                            void* dst, int ld_dst, void* stream) {
    const int esz = dtype == LPI_F32 ? 4 : 2;
    if (!src || !dst || B <= 0 || P <= 0 || row0 < 0 || row0 + P > L || cols <= 0 || ld_src < cols || ld_dst < cols) return LPI_EINVAL;
    if ((cols * esz) % 16 || (ld_src * esz) % 16 || (ld_dst * esz) % 16 || (((uintptr_t)src | (uintptr_t)dst) & 15)) return LPI_EINVAL;
    const int chunks = cols * esz / 16;
    const long n = (long)B * P * chunks;
    LPI_LAUNCH(gather_batch_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, S(stream), B, L, row_start, row0, P, chunks, (const uint4*)src,
               (long)ld_src * esz / 16, (uint4*)dst, (long)ld_dst * esz / 16);
    LPI_CHECK_LAST();
    return 0;
}
extern "C" int lpi_gather_batch_rows(int dtype, int B, int L, int row0, int P, int cols, const void* src, int ld_src, void* dst, int ld_dst,
                                     void* stream) {
    return lpi_gather_batch_rows_varlen(dtype, B, L, nullptr, row0, P, cols, src, ld_src, dst, ld_dst, stream);
}

extern "C" int lpi_rows_sum_over_batch_varlen(int dtype, int B, int L, const int32_t* row_start, int row0, int P, int d, const void* dx, float* out,
                                              int accumulate, void* stream) {
    if (!dx || !out || B <= 0 || P <= 0 || row0 < 0 || row0 + P > L || bad_row_dim(d)) return LPI_EINVAL;
    if (dtype == LPI_F32) LPI_LAUNCH(rows_sum_kernel<float>, dim3(P, (d + 255) / 256), dim3(1024), 0, S(stream), B, L, row_start, row0, P, d, (const float*)dx, out, accumulate);
    else if (dtype == LPI_BF16) LPI_LAUNCH(rows_sum_kernel<bf16_t>, dim3(P, (d + 255) / 256), dim3(1024), 0, S(stream), B, L, row_start, row0, P, d, (const bf16_t*)dx, out, accumulate);
    else return LPI_EINVAL;
    LPI_CHECK_LAST();
    return 0;
}
extern "C" int lpi_rows_sum_over_batch(int dtype, int B, int L, int row0, int P, int d, const void* dx, float* out, int accumulate, void* stream) {
    return lpi_rows_sum_over_batch_varlen(dtype, B, L, nullptr, row0, P, d, dx, out, accumulate, stream);
}

extern "C" int lpi_vis_assemble_bwd(int dtype, int B, int G2, int P, int d, void* dx0, const float* prompt0, long prompt_bstride,
                                    const float* gamma, const float* mean, const float* rstd, float* dprompt, void* stream) {
    if (!dx0 || !prompt0 || !gamma || !mean || !rstd || !dprompt || B <= 0 || P <= 0 || bad_row_dim(d) || (prompt_bstride & 3)) return LPI_EINVAL;
    const int L = 1 + P + G2;
    dim3 g(rows_grid((long)B * P)), bl(256);
    if (dtype == LPI_F32) LPI_LAUNCH(vis_prompt_rows_bwd_kernel<float>, g, bl, 0, S(stream), B, L, P, d, (float*)dx0, prompt0, prompt_bstride, gamma, mean, rstd);
    else if (dtype == LPI_BF16) LPI_LAUNCH(vis_prompt_rows_bwd_kernel<bf16_t>, g, bl, 0, S(stream), B, L, P, d, (bf16_t*)dx0, prompt0, prompt_bstride, gamma, mean, rstd);
    else return LPI_EINVAL;
    LPI_CHECK_LAST();
    return lpi_rows_sum_over_batch(dtype, B, L, 1, P, d, dx0, dprompt, 0, stream);
}

static int txt_embed_launch(int x_dtype, int B, int L, const int32_t* row_start, int pre, int P, int d, const int64_t* ids, const float* tok_emb, const float* pos,
                            const float* ctx, long ctx_bstride, void* x0, float* out_mean, float* out_rstd, void* stream);
extern "C" int lpi_txt_embed_fwd_shared(int x_dtype, int B, int L, const int32_t* row_start, int shared_rows, int P, int d, const int64_t* ids, const float* tok_emb,
                                        const float* pos, const float* ctx, void* x0, float* out_mean, float* out_rstd, void* stream) {
    // the shared rows hold SOT and the P context slots: they are the same for every sample only if the context is broadcast (ctx_bstride = 0) and spliced in
    if (!row_start || !ctx || shared_rows != 1 + P || shared_rows >= L) return LPI_EINVAL;
    return txt_embed_launch(x_dtype, B, L, row_start, shared_rows, P, d, ids, tok_emb, pos, ctx, 0, x0, out_mean, out_rstd, stream);
}
static int txt_embed_launch(int x_dtype, int B, int L, const int32_t* row_start, int pre, int P, int d, const int64_t* ids, const float* tok_emb, const float* pos,
                            const float* ctx, long ctx_bstride, void* x0, float* out_mean, float* out_rstd, void* stream) {
    if ((out_mean != nullptr) != (out_rstd != nullptr)) return LPI_EINVAL;
    if (!ids || !tok_emb || !pos || !x0 || B <= 0 || L <= 0 || P < 0 || P + 1 > L || bad_row_dim(d) || (ctx_bstride & 3)) return LPI_EINVAL;
    if (x_dtype == LPI_F32)
        LPI_LAUNCH(txt_embed_kernel<float>, dim3(rows_grid((long)B * L)), dim3(256), 0, S(stream), B, L, row_start, P, d, ids, tok_emb, pos, ctx, ctx_bstride, (float*)x0, out_mean, out_rstd, t_rowstat_guard, pre);
    else if (x_dtype == LPI_F16)
        LPI_LAUNCH(txt_embed_kernel<f16_t>, dim3(rows_grid((long)B * L)), dim3(256), 0, S(stream), B, L, row_start, P, d, ids, tok_emb, pos, ctx, ctx_bstride, (f16_t*)x0, out_mean, out_rstd, t_rowstat_guard, pre);
    else
        return LPI_EINVAL;
    LPI_CHECK_LAST();
    return 0;
}
extern "C" int lpi_txt_embed_fwd_varlen(int x_dtype, int B, int L, const int32_t* row_start, int P, int d, const int64_t* ids, const float* tok_emb,
                                        const float* pos, const float* ctx, long ctx_bstride, void* x0, float* out_mean, float* out_rstd, void* stream) {
    return txt_embed_launch(x_dtype, B, L, row_start, 0, P, d, ids, tok_emb, pos, ctx, ctx_bstride, x0, out_mean, out_rstd, stream);
}
extern "C" int lpi_txt_embed_fwd(int x_dtype, int B, int L, int P, int d, const int64_t* ids, const float* tok_emb, const float* pos, const float* ctx,
                                 long ctx_bstride, void* x0, float* out_mean, float* out_rstd, void* stream) {
    return lpi_txt_embed_fwd_varlen(x_dtype, B, L, nullptr, P, d, ids, tok_emb, pos, ctx, ctx_bstride, x0, out_mean, out_rstd, stream);
}

extern "C" int lpi_prompt_add_varlen(int x_dtype, int B, int L, const int32_t* row_start, int P, int d, void* x, const float* prompt_l, long prompt_bstride,
                                     float* out_mean, float* out_rstd, void* stream) {
    if ((out_mean != nullptr) != (out_rstd != nullptr)) return LPI_EINVAL;
    if (!x || !prompt_l || B <= 0 || P <= 0 || P + 1 > L || bad_row_dim(d) || (prompt_bstride & 3)) return LPI_EINVAL;
    if (x_dtype == LPI_F32)
        LPI_LAUNCH(prompt_add_kernel<float>, dim3(rows_grid((long)B * P)), dim3(256), 0, S(stream), B, L, row_start, P, d, (float*)x, prompt_l, prompt_bstride, out_mean, out_rstd, t_rowstat_guard);
    else if (x_dtype == LPI_F16)
        LPI_LAUNCH(prompt_add_kernel<f16_t>, dim3(rows_grid((long)B * P)), dim3(256), 0, S(stream), B, L, row_start, P, d, (f16_t*)x, prompt_l, prompt_bstride, out_mean, out_rstd, t_rowstat_guard);
    else
        return LPI_EINVAL;
    LPI_CHECK_LAST();
    return 0;
}
extern "C" int lpi_prompt_add(int x_dtype, int B, int L, int P, int d, void* x, const float* prompt_l, long prompt_bstride, float* out_mean,
                              float* out_rstd, void* stream) {
    return lpi_prompt_add_varlen(x_dtype, B, L, nullptr, P, d, x, prompt_l, prompt_bstride, out_mean, out_rstd, stream);
}

extern "C" int lpi_pool_ln_fwd(int dtype, int x_dtype, int B, int L, int d, const void* x, const int32_t* idx, const float* gamma, const float* beta,
                               void* y, int ldy, float* mean, float* rstd, void* stream) {
    if (!x || !gamma || !beta || !y || !mean || !rstd || B <= 0 || L < 0 || (L == 0 && !idx) || bad_row_dim(d) || (ldy & 3)) return LPI_EINVAL;      // L == 0: idx holds absolute rows
    dim3 g(rows_grid(B)), b(256);
    if (dtype == LPI_F32 && x_dtype == LPI_F32)
        LPI_LAUNCH((pool_ln_fwd_kernel<float, float>), g, b, 0, S(stream), B, L, d, (const float*)x, idx, gamma, beta, (float*)y, ldy, mean, rstd);
    else if (dtype == LPI_BF16 && x_dtype == LPI_F32)
        LPI_LAUNCH((pool_ln_fwd_kernel<float, bf16_t>), g, b, 0, S(stream), B, L, d, (const float*)x, idx, gamma, beta, (bf16_t*)y, ldy, mean, rstd);
    else if (dtype == LPI_BF16 && x_dtype == LPI_F16)
        LPI_LAUNCH((pool_ln_fwd_kernel<f16_t, bf16_t>), g, b, 0, S(stream), B, L, d, (const f16_t*)x, idx, gamma, beta, (bf16_t*)y, ldy, mean, rstd);
    else if (dtype == LPI_F16 && x_dtype == LPI_F32)
        LPI_LAUNCH((pool_ln_fwd_kernel<float, f16_t>), g, b, 0, S(stream), B, L, d, (const float*)x, idx, gamma, beta, (f16_t*)y, ldy, mean, rstd);
    else if (dtype == LPI_F16 && x_dtype == LPI_F16)
        LPI_LAUNCH((pool_ln_fwd_kernel<f16_t, f16_t>), g, b, 0, S(stream), B, L, d, (const f16_t*)x, idx, gamma, beta, (f16_t*)y, ldy, mean, rstd);
    else return LPI_EINVAL;
    LPI_CHECK_LAST();
    return 0;
}

extern "C" int lpi_pool_ln_bwd(int cast_dtype, int B, int L, int d, const float* dy, int lddy, const float* x, const int32_t* idx,
                               const float* gamma, const float* mean, const float* rstd, float* dx, void* dx_cast, void* stream) {
    if (!dy || !x || !gamma || !mean || !rstd || !dx || B <= 0 || bad_row_dim(d) || (lddy & 3)) return LPI_EINVAL;
    dim3 g(rows_grid(B)), b(256);
    if (cast_dtype == LPI_F32) LPI_LAUNCH(pool_ln_bwd_kernel<float>, g, b, 0, S(stream), B, L, d, dy, lddy, x, idx, gamma, mean, rstd, dx, (float*)dx_cast);
    else if (cast_dtype == LPI_BF16) LPI_LAUNCH(pool_ln_bwd_kernel<bf16_t>, g, b, 0, S(stream), B, L, d, dy, lddy, x, idx, gamma, mean, rstd, dx, (bf16_t*)dx_cast);
    else return LPI_EINVAL;
    LPI_CHECK_LAST();
    return 0;
}

extern "C" int lpi_gather_rows(int x_dtype, int B, int L, int d, const void* src, const int32_t* idx, float* dst, void* stream) {
    if (!src || !dst || B <= 0 || L < 0 || (L == 0 && !idx) || bad_row_dim(d)) return LPI_EINVAL;      // L == 0: idx holds absolute rows
    if (x_dtype == LPI_F32) LPI_LAUNCH(gather_rows_kernel<float>, dim3(rows_grid(B)), dim3(256), 0, S(stream), B, L, d, (const float*)src, idx, dst);
    else if (x_dtype == LPI_F16) LPI_LAUNCH(gather_rows_kernel<f16_t>, dim3(rows_grid(B)), dim3(256), 0, S(stream), B, L, d, (const f16_t*)src, idx, dst);
    else return LPI_EINVAL;
    LPI_CHECK_LAST();
    return 0;
}

extern "C" int lpi_scatter_rows(int cast_dtype, int B, int L, int d, const float* src, const int32_t* idx, float* dst, void* dst_cast,
                                void* stream) {
    if (!src || (!dst && !dst_cast) || B <= 0 || L < 0 || (L == 0 && !idx) || bad_row_dim(d)) return LPI_EINVAL;      // L == 0: idx holds absolute rows
    dim3 g(rows_grid(B)), b(256);
    if (cast_dtype == LPI_F32) LPI_LAUNCH(scatter_rows_kernel<float>, g, b, 0, S(stream), B, L, d, src, idx, dst, (float*)dst_cast);
    else if (cast_dtype == LPI_BF16) LPI_LAUNCH(scatter_rows_kernel<bf16_t>, g, b, 0, S(stream), B, L, d, src, idx, dst, (bf16_t*)dst_cast);
    else return LPI_EINVAL;
    LPI_CHECK_LAST();
    return 0;
}

extern "C" int lpi_l2norm_fwd(int B, int E, const float* x, int ldx, float* y, int ldy, float* inv_norm, void* stream) {
    if (!x || !y || !inv_norm || B <= 0 || bad_row_dim(E) || (ldx & 3) || (ldy & 3)) return LPI_EINVAL;
    LPI_LAUNCH(l2norm_fwd_kernel, dim3(rows_grid(B)), dim3(256), 0, S(stream), B, E, x, ldx, y, ldy, inv_norm);
    LPI_CHECK_LAST();
    return 0;
}

extern "C" int lpi_l2norm_bwd(int B, int E, const float* y, int ldy, const float* dy, int lddy, const float* inv_norm, float* dx, int lddx,
                              void* stream) {
    if (!y || !dy || !inv_norm || !dx || B <= 0 || bad_row_dim(E) || (ldy & 3) || (lddy & 3) || (lddx & 3)) return LPI_EINVAL;
    LPI_LAUNCH(l2norm_bwd_kernel, dim3(rows_grid(B)), dim3(256), 0, S(stream), B, E, y, ldy, dy, lddy, inv_norm, dx, lddx);
    LPI_CHECK_LAST();
    return 0;
}

// ---- statistics of the fp16 residual stream from the slot sums an LPI_EPI_RES_ROWSTATS GEMM epilogue left (include/lpi_hip.h): one thread per row
// adds the d / 128 slots in order; the variance is E[x^2] - mean^2 in f32 (the stream's rows have |mean| well below their deviation: the relative
// error of the difference is ~1e-7 (1 + mean^2 / var)), clamped at zero.
struct StatFinP { int rows, nslot; const float* part; int ld; float* mean; float* rstd; float inv_d; RowstatGuard guard; };
__device__ __forceinline__ void stat_fin_body(const StatFinP& p, int row, float eps) {
    if (row >= p.rows) return;
    float s = 0.f, q = 0.f;
    for (int j = 0; j < p.nslot; ++j) {
        s += p.part[(size_t)(2 * j) * p.ld + row];
        q += p.part[(size_t)(2 * j + 1) * p.ld + row];
    }
    const float mu = s * p.inv_d;
    const float var = q * p.inv_d - mu * mu;
    p.mean[row] = mu;
    p.rstd[row] = 1.0f / sqrtf(fmaxf(var, 0.f) + eps);
    rowstat_guard_check(p.guard, mu, var);
}
__global__ __launch_bounds__(256) void ln_stats_finalize_kernel(StatFinP p0, StatFinP p1, int nb0, float eps) {
    if ((int)blockIdx.x < nb0) stat_fin_body(p0, blockIdx.x * 256 + threadIdx.x, eps);
    else stat_fin_body(p1, (blockIdx.x - nb0) * 256 + threadIdx.x, eps);
}
static bool stat_fin_ok(int rows, int d, const float* part, int ld, const float* mean, const float* rstd) {
    return rows > 0 && d > 0 && !(d & 127) && part && mean && rstd && ld >= rows;
}
extern "C" int lpi_ln_stats_finalize_pair(int rows0, int d0, const float* part0, int ld0, float* mean0, float* rstd0,
                                          int rows1, int d1, const float* part1, int ld1, float* mean1, float* rstd1, float eps, void* stream) {
    if (!stat_fin_ok(rows0, d0, part0, ld0, mean0, rstd0) || (rows1 > 0 && !stat_fin_ok(rows1, d1, part1, ld1, mean1, rstd1))) return LPI_EINVAL;
    const StatFinP p0 = {rows0, d0 / 128, part0, ld0, mean0, rstd0, 1.0f / (float)d0, t_rowstat_guard};
    const StatFinP p1 = {rows1 > 0 ? rows1 : 0, rows1 > 0 ? d1 / 128 : 0, part1, ld1, mean1, rstd1, rows1 > 0 ? 1.0f / (float)d1 : 0.f, t_rowstat_guard};
    const int nb0 = (rows0 + 255) / 256, nb1 = rows1 > 0 ? (rows1 + 255) / 256 : 0;
    LPI_LAUNCH(ln_stats_finalize_kernel, dim3(nb0 + nb1), dim3(256), 0, S(stream), p0, p1, nb0, eps);
    LPI_CHECK_LAST();
    return 0;
}
extern "C" int lpi_ln_stats_finalize(int rows, int d, const float* part, int ld, float eps, float* mean, float* rstd, void* stream) {
    return lpi_ln_stats_finalize_pair(rows, d, part, ld, mean, rstd, 0, 0, nullptr, 0, nullptr, nullptr, eps, stream);
}

// ---- lpi_row_jobs (see row_jobs_kernel): validation per op = the single-op entry point's
static int row_job_blocks(const lpi_row_job& q) {
    auto al16 = [](const void* p) { return !((uintptr_t)p & 15); };
    if (q.B <= 0) return -1;
    switch (q.op) {
    case LPI_ROWOP_POOL_LN_FWD:
        if (!q.a || !q.gamma || !q.beta || !q.out || !q.mean || !q.rstd || q.L < 0 || (q.L == 0 && !q.idx) || bad_row_dim(q.d) || (q.ld_c & 3)) return -1;
        if (!((q.dt_a == LPI_F32 && (q.dt_b == LPI_F32 || q.dt_b == LPI_BF16 || q.dt_b == LPI_F16)) || (q.dt_a == LPI_F16 && (q.dt_b == LPI_BF16 || q.dt_b == LPI_F16)))) return -1;
        return rows_grid(q.B);
    case LPI_ROWOP_L2NORM_FWD:
        if (!q.a || !q.out || !q.mean || bad_row_dim(q.d) || (q.ld_a & 3) || (q.ld_c & 3)) return -1;
        return rows_grid(q.B);
    case LPI_ROWOP_L2NORM_BWD:
        if (!q.a || !q.b || !q.mean_in || !q.out || bad_row_dim(q.d) || (q.ld_a & 3) || (q.ld_b & 3) || (q.ld_c & 3) || (q.out2 && q.dt_b != LPI_BF16)) return -1;
        return rows_grid(q.B);
    case LPI_ROWOP_POOL_LN_BWD:
        if (!q.a || !q.b || !q.gamma || !q.mean_in || !q.rstd_in || !q.out || bad_row_dim(q.d) || (q.ld_a & 3) || (q.dt_b != LPI_BF16 && q.dt_b != LPI_F32)) return -1;
        return rows_grid(q.B);
    case LPI_ROWOP_LN_BWD:
        if (!q.a || !q.b || !q.gamma || !q.mean_in || !q.rstd_in || (!q.out && !q.out2) || bad_row_dim(q.d) || (q.ld_a & 3) || (q.ld_b & 3) || (q.ld_c & 3)) return -1;
        if (!((q.dt_a == LPI_BF16 && q.dt_b == LPI_BF16) || (q.dt_a == LPI_F32 && q.dt_b == LPI_F32))) return -1;
        return rows_grid(q.B);
    case LPI_ROWOP_SCATTER_ADD: {
        if (!q.a || !q.out || q.L < 0 || (q.L == 0 && !q.idx) || q.d <= 0 || (q.d & 3) || q.ld_a < q.d || q.ld_c < q.d || (q.dt_a != LPI_BF16 && q.dt_a != LPI_F32)) return -1;
        const int esz = q.dt_a == LPI_F32 ? 4 : 2;
        if ((q.ld_a * esz) % 8 || (q.ld_c * esz) % 8 || (((uintptr_t)q.a | (uintptr_t)q.out) & 7)) return -1;
        return q.B;
    }
    case LPI_ROWOP_GATHER_BATCH_ROWS:
        if (!q.a || !q.out || q.P <= 0 || q.row0 < 0 || q.row0 + q.P > q.L || q.d <= 0 || q.ld_a < q.d || q.ld_c < q.d || !al16(q.a) || !al16(q.out)) return -1;
        return (int)(((long)q.B * q.P * q.d + 255) / 256);
    case LPI_ROWOP_PROMPT_ADD:
        if (!q.out || !q.a || q.P <= 0 || q.P + 1 > q.L || bad_row_dim(q.d) || (q.bstride & 3) || ((q.mean != nullptr) != (q.rstd != nullptr)) || (q.dt_a != LPI_F16 && q.dt_a != LPI_F32)) return -1;
        return rows_grid((long)q.B * q.P);
    case LPI_ROWOP_LN_BWD_ROWS_H16:
        if (!q.a || !q.b || !q.gamma || !q.mean_in || !q.rstd_in || !q.out2 || q.P <= 0 || q.row0 < 0 || q.row0 + q.P > q.L || bad_row_dim(q.d)) return -1;
        if (!ln_h16_ok(q.d, q.ld_a, q.ld_b, q.a, q.b) || (q.ld_c & 7) || !al16(q.out2)) return -1;
        return (q.B * q.P + 7) / 8;
    case LPI_ROWOP_VIS_PROMPT_ROWS_BWD:
        if (!q.out || !q.a || !q.gamma || !q.mean_in || !q.rstd_in || q.P <= 0 || bad_row_dim(q.d) || (q.bstride & 3) || (q.dt_a != LPI_BF16 && q.dt_a != LPI_F32)) return -1;
        return rows_grid((long)q.B * q.P);
    }
    return -1;
}
extern "C" int lpi_row_jobs(int n, const lpi_row_job* jobs, void* stream) {
    if (!jobs || n <= 0 || n > LPI_ROW_JOBS_MAX) return LPI_EINVAL;
    RowJobs js = {};
    js.n = n;
    js.guard = t_rowstat_guard;
    long total = 0;
    for (int i = 0; i < n; ++i) {
        const int nb = row_job_blocks(jobs[i]);
        if (nb <= 0) return LPI_EINVAL;
        js.j[i] = jobs[i];
        js.nb[i] = nb;
        total += nb;
    }
    LPI_LAUNCH(row_jobs_kernel, dim3((unsigned)total), dim3(256), 0, S(stream), js);
    LPI_CHECK_LAST();
    return 0;
}

// the two towers' lpi_rows_sum_over_batch_varlen of one prompt layer in ONE launch (the same sums, bit for bit)
extern "C" int lpi_rows_sum_over_batch_pair(int dtype, const lpi_rows_sum_desc* d, void* stream) {
    if (!d) return LPI_EINVAL;
    RowsSumP p[2];
    int gx = 0, gy = 0;
    for (int i = 0; i < 2; ++i) {
        const lpi_rows_sum_desc& q = d[i];
        if (!q.dx || !q.out || q.B <= 0 || q.P <= 0 || q.row0 < 0 || q.row0 + q.P > q.L || bad_row_dim(q.d)) return LPI_EINVAL;
        p[i] = RowsSumP{q.B, q.L, q.row0, q.P, q.d, q.accumulate, q.row_start, q.dx, q.out};
        gx = std::max(gx, q.P);
        gy = std::max(gy, (q.d + 255) / 256);
    }
    if (dtype == LPI_F32) LPI_LAUNCH(rows_sum_pair_kernel<float>, dim3(gx, gy, 2), dim3(1024), 0, S(stream), p[0], p[1]);
    else if (dtype == LPI_BF16) LPI_LAUNCH(rows_sum_pair_kernel<bf16_t>, dim3(gx, gy, 2), dim3(1024), 0, S(stream), p[0], p[1]);
    else return LPI_EINVAL;
    LPI_CHECK_LAST();
    return 0;
}

extern "C" int lpi_eot_index(int B, int L, const int64_t* ids, int32_t* idx, void* stream) {
    if (!ids || !idx || B <= 0 || L <= 0) return LPI_EINVAL;
    LPI_LAUNCH(eot_index_kernel, dim3((B + 255) / 256), dim3(256), 0, S(stream), B, L, ids, idx);
    LPI_CHECK_LAST();
    return 0;
}

extern "C" int lpi_cast(int src_dtype, int dst_dtype, long n, const void* src, void* dst, void* stream) {
    if (!src || !dst || n <= 0 || (n & 3)) return LPI_EINVAL;
    const long n4 = n >> 2;
    dim3 g((unsigned)((n4 + 255) / 256 > 4096 ? 4096 : (n4 + 255) / 256)), b(256);
    if (src_dtype == LPI_F32 && dst_dtype == LPI_BF16) LPI_LAUNCH((cast_kernel<float, bf16_t>), g, b, 0, S(stream), n4, (const float*)src, (bf16_t*)dst);
    else if (src_dtype == LPI_BF16 && dst_dtype == LPI_F32) LPI_LAUNCH((cast_kernel<bf16_t, float>), g, b, 0, S(stream), n4, (const bf16_t*)src, (float*)dst);
    else if (src_dtype == LPI_F32 && dst_dtype == LPI_F32) LPI_LAUNCH((cast_kernel<float, float>), g, b, 0, S(stream), n4, (const float*)src, (float*)dst);
    else return LPI_EINVAL;
    LPI_CHECK_LAST();
    return 0;
}

extern "C" int lpi_transpose2(int dtype, int rows0, int cols0, const void* src0, int lds0, void* dst0, int ldd0,
                              int rows1, int cols1, const void* src1, int lds1, void* dst1, int ldd1, void* stream) {
    if (!src0 || !dst0 || !src1 || !dst1 || rows0 <= 0 || cols0 <= 0 || rows1 <= 0 || cols1 <= 0 || lds0 < cols0 || ldd0 < rows0 || lds1 < cols1 || ldd1 < rows1)
        return LPI_EINVAL;
    dim3 g((std::max(cols0, cols1) + 31) / 32, (std::max(rows0, rows1) + 31) / 32, 2), b(256);
    if (dtype == LPI_F32)
        LPI_LAUNCH(transpose2_kernel<float>, g, b, 0, S(stream), rows0, cols0, (const float*)src0, lds0, (float*)dst0, ldd0, rows1, cols1, (const float*)src1, lds1,
                   (float*)dst1, ldd1);
    else return LPI_EINVAL;
    LPI_CHECK_LAST();
    return 0;
}

extern "C" int lpi_transpose(int dtype, int rows, int cols, const void* src, int lds, void* dst, int ldd, void* stream) {
    if (!src || !dst || rows <= 0 || cols <= 0 || lds < cols || ldd < rows) return LPI_EINVAL;
    dim3 g((cols + 31) / 32, (rows + 31) / 32), b(256);
    if (dtype == LPI_F32) LPI_LAUNCH(transpose_kernel<float>, g, b, 0, S(stream), rows, cols, (const float*)src, lds, (float*)dst, ldd);
    else if (dtype == LPI_BF16) LPI_LAUNCH(transpose_kernel<bf16_t>, g, b, 0, S(stream), rows, cols, (const bf16_t*)src, lds, (bf16_t*)dst, ldd);
    else return LPI_EINVAL;
    LPI_CHECK_LAST();
    return 0;
}
