// Prompted multi-head attention, bf16 operands, second generation: PERSISTENT workgroups that stream heads through LDS with LDS-DMA.
//
// replaces: nn.MultiheadAttention as called from ResidualAttentionBlock.attention (retrieval/models/clip/model.py:183-185) and its
// autograd backward, like attention.hip (whose tile arithmetic — transposed scores, lane-local online softmax, P from the
// accumulators straight into the next MFMA, fixed summation order, no atomics — is kept bit for bit; its kernels remain the f32 path
// and the A/B reference, tuning key 7).
//
// Why a second generation (profiles/r02_pmc.json, rocprofv3 --pmc on the round-1 kernels): the one-head-per-workgroup kernels load,
// compute and store strictly in sequence — waves parked on s_waitcnt / barriers 50 % (backward) and 32 % (forward) of their cycles,
// MFMA busy 18 % / 14 % — and 43 % of their LDS cycles were bank conflicts (144-byte row stride on the 64-bank LDS).  Here:
//   * a workgroup owns a CU-sized slice of LDS for the whole launch and walks heads bh = blockIdx.x, + gridDim.x, ...;
//   * the matrices of the NEXT phase arrive by global_load_lds_dwordx4 (no registers, issued from inline asm so that the compiler
//     does not drain them before every ds_read) while the current phase computes:
//       forward : K,V double-buffered (2 x 2 images); next head's K,V + the wave's next Q rows land during this head's tiles;
//       backward: one image each of Q, K, V, dO.  Phase A (dQ: needs all of K,V + the wave's own Q/dO/O rows, prefetched into
//                 registers) runs while Q,dO of the same head land; phase B (dK,dV: needs all of Q,dO + the wave's own K,V rows,
//                 copied to registers first) runs while K,V of the NEXT head land in the images it has just freed;
//   * every vmcnt(0) sits right BEFORE the phase's result stores, so a store's latency is never waited for;
//   * LDS images are unpadded 128-byte rows (what LDS-DMA writes: 1 KiB = 8 rows per wave instruction) with the 16-byte chunk index
//     XOR-swizzled by (row & 6): conflict-free for both read patterns on the 64-bank LDS (ds_read_b128 row fragments in lane groups
//     {0-3,12-15,20-27},.. and ds_read_b64_tr_b16 in halves of 32 lanes; checked exhaustively, tools/lds_swizzle_check.py).
// Rows [L, Lp) of every image are zeroed once per launch and never written again (the DMA is EXEC-masked there).
#include <type_traits>
#include "common.h"
#include "attn_softmax.h"

extern int g_lpi_tuning[16];

namespace {

typedef bf16_t T;
constexpr int HD = 64;
constexpr int NB = 2;
constexpr int RB = 128;          // bytes per LDS row (64 bf16, no padding)
constexpr int KS = 2;            // 32-wide k-steps per 64-element row
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;
constexpr float SCALE = 0.125f;

typedef __attribute__((ext_vector_type(2))) float f32x2_;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_;
__device__ __forceinline__ uint32_t pack2(float a, float b) { return __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2_){a, b}, bf16x2_)); }
__device__ __forceinline__ float exp2_fast(float x) { return __builtin_amdgcn_exp2f(x); }

__device__ __forceinline__ float grp_max(float v) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float grp_sum(float v) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// one LDS-DMA instruction: 64 lanes x 16 B -> 1 KiB of LDS starting at lds_addr (wave-uniform), lane i at +16 i
__device__ __forceinline__ void glds16(const T* src, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(lds_addr) : "memory");
}

// rows [0, L) of a [L, 64] matrix (row stride ld elements) -> swizzled image at LDS byte address img (wave-uniform).  Wave w issues the
// 8-row blocks w, w + nw, ...; lane i of a block moves physical chunk i & 7 of row 8 blk + (i >> 3) = logical chunk (i & 7) ^ (row & 6).
__device__ __forceinline__ void stage_dma(unsigned img, const T* g, int ld, int L, int wave, int nw, int lane) {
    const int r8 = lane >> 3, pc = lane & 7;
    for (int blk = wave; blk * 8 < L; blk += nw) {
        const int row = blk * 8 + r8;
#if defined(LPI_ABL_ATTN_NOLOAD) || defined(LPI_ABL_ATTN_NODMA)      /* ablation build: no global reads (images keep their zeros) */
        if (row < L && ld == 12345) glds16(g + (size_t)row * ld + ((pc ^ (row & 6)) << 3), img + blk * 1024);
#else
        if (row < L) glds16(g + (size_t)row * ld + ((pc ^ (row & 6)) << 3), img + blk * 1024);
#endif
    }
}

// per-lane byte offsets inside a 16-row (rc) / 32-row (tr) window of an image
struct RdOff {
    int rc[KS];   // row fragment: row (lane & 15), logical chunk (lane >> 4) + 4 ks
    int tr[4];    // ds_read_b64_tr_b16: row 4 (lane >> 4) + ((lane & 15) >> 2), logical chunk 2 dt + ((lane & 3) >> 1), half (lane & 1)
};
__device__ __forceinline__ RdOff make_offsets(int lane) {
    RdOff o;
    const int r = lane & 15, g = lane >> 4;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) o.rc[ks] = r * RB + (((g + 4 * ks) ^ (r & 6)) << 4);
    const int rr = 4 * g + (r >> 2), p = lane & 3;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o.tr[dt] = rr * RB + (((2 * dt + (p >> 1)) ^ (rr & 6)) << 4) + (p & 1) * 8;
    return o;
}

// this lane's KS chunks of image row row0 + (lane & 15)   (row0 a multiple of 16)
__device__ __forceinline__ void lds_rows(Chunk (&q)[KS], const char* img, int row0, const RdOff& o) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) q[ks].u = *reinterpret_cast<const uint4*>(img + row0 * RB + o.rc[ks]);
}
// this lane's KS row chunks (chunk g + 4 ks) of row `row` of a global matrix; zeros if !valid
__device__ __forceinline__ void glb_rows(Chunk (&q)[KS], const T* g, size_t row, int ld, int grp, bool valid) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        q[ks].u = make_uint4(0, 0, 0, 0);
#if defined(LPI_ABL_ATTN_NOLOAD) || defined(LPI_ABL_ATTN_NOOWN)
        if (valid && ld == 12345) q[ks].u = *reinterpret_cast<const uint4*>(g + row * ld + (grp + 4 * ks) * 8);
#else
        if (valid) q[ks].u = *reinterpret_cast<const uint4*>(g + row * ld + (grp + 4 * ks) * 8);
#endif
    }
}

// acc[j] = (16 image rows r0.. r0+15) x (register operand j)^T, the image chunk read once for all j
__device__ __forceinline__ void mma_rows(f32x4 (&acc)[NB], const char* win, const RdOff& o, const Chunk (&b)[NB][KS]) {
#pragma unroll
    for (int j = 0; j < NB; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        Chunk a;
        a.u = *reinterpret_cast<const uint4*>(win + o.rc[ks]);
#pragma unroll
        for (int j = 0; j < NB; ++j) mma_chunk<T>(acc[j], a, b[j][ks]);
    }
}
// acc[j][dt] += X^T[d = 16 dt + .., k] . P_j[k][col] over the 32 image rows of the window; p0/p1 = the two 16-row accumulator tiles
__device__ __forceinline__ void mma_tr(f32x4 (&acc)[NB][4], const char* win, const RdOff& o, const f32x4 (&p0)[NB], const f32x4 (&p1)[NB]) {
    Chunk b[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j)
        b[j].u = make_uint4(pack2(p0[j][0], p0[j][1]), pack2(p0[j][2], p0[j][3]), pack2(p1[j][0], p1[j][1]), pack2(p1[j][2], p1[j][3]));
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
        short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(win + o.tr[dt]));
        short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(win + 16 * RB + o.tr[dt]));
        Chunk a;
        const uint2 lo2 = __builtin_bit_cast(uint2, lo), hi2 = __builtin_bit_cast(uint2, hi);
        a.u = make_uint4(lo2.x, lo2.y, hi2.x, hi2.y);
#pragma unroll
        for (int j = 0; j < NB; ++j) mma_chunk<T>(acc[j][dt], a, b[j]);
    }
}

// rows [L, Lp) of n images of Lp rows each, starting at smem: zero (once per launch)
__device__ __forceinline__ void zero_pad_rows(char* smem, int n_img, int L, int Lp) {
    const int per = (Lp - L) * (RB / 16);
    for (int i = threadIdx.x; i < n_img * per; i += blockDim.x) {
        const int im = i / per, r = i % per;
        *reinterpret_cast<uint4*>(smem + (size_t)im * Lp * RB + (size_t)L * RB + r * 16) = make_uint4(0, 0, 0, 0);
    }
}

// rows [0, L) of an image: fp16 -> bf16 in place (the XOR swizzle permutes whole 16-byte chunks, so the layout is irrelevant here)
__device__ __forceinline__ void convert_image_f16_to_bf16(char* img, int L) {
    for (int i = threadIdx.x; i < L * (RB / 16); i += blockDim.x) {
        Chunk c;
        c.u = *reinterpret_cast<const uint4*>(img + i * 16);
        chunk_f16_to_bf16(c);
        *reinterpret_cast<uint4*>(img + i * 16) = c.u;
    }
}

#define LPI_WAIT_VM0() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define LPI_BARRIER()                                              \
    do {                                                           \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         \
        __builtin_amdgcn_s_barrier();                              \
        asm volatile("" ::: "memory");                             \
    } while (0)

// ------------------------------------------------------------------------------------------------ forward
template <bool CAUSAL>
__global__ __launch_bounds__(512) void attn_fwd2_kernel(int L, int Lp, int H, int total, const T* __restrict__ qkv, int ldqkv,
                                                       T* __restrict__ ctx, int ldctx, float* __restrict__ lse) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4;
    const int dm = H * HD;
    const int img = Lp * RB;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem);
    const RdOff off = make_offsets(lane);
    const float c = SCALE * LOG2E;

    zero_pad_rows(smem, 4, L, Lp);
    auto head_ptr = [&](int bh) { return qkv + (size_t)(bh / H) * L * ldqkv + (bh % H) * HD; };
    auto stage_kv = [&](int bh, int buf) {
        const T* qg = head_ptr(bh);
        stage_dma(lds0 + (2 * buf) * img, qg + dm, ldqkv, L, wave, nw, lane);
        stage_dma(lds0 + (2 * buf + 1) * img, qg + 2 * dm, ldqkv, L, wave, nw, lane);
    };
    auto load_q = [&](Chunk (&q)[NB][KS], int bh, int q0) {
        const T* qg = head_ptr(bh);
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int qr = q0 + 16 * j + (lane & 15);
            glb_rows(q[j], qg, qr, ldqkv, g, qr < L);
        }
    };

    int bh = blockIdx.x;
    Chunk q[NB][KS];
    load_q(q, bh, wave * 16 * NB);
    stage_kv(bh, 0);
    LPI_WAIT_VM0();
    __syncthreads();          // pad rows zeroed, first K/V images landed for every wave

    for (int buf = 0; bh < total; bh += gridDim.x, buf ^= 1) {
        const int nbh = bh + gridDim.x;
        Chunk qn[NB][KS];
        if (nbh < total) {        // next head's K,V -> the other pair of images; this wave's next Q rows -> registers
            stage_kv(nbh, buf ^ 1);
            load_q(qn, nbh, wave * 16 * NB);
        }
        __builtin_amdgcn_sched_barrier(0);
        const char* k_lds = smem + (2 * buf) * img;
        const char* v_lds = k_lds + img;
        const int b = bh / H, h = bh % H;
        for (int q0 = wave * 16 * NB; q0 < L; q0 += nw * 16 * NB) {
            int qrow[NB];
#pragma unroll
            for (int j = 0; j < NB; ++j) qrow[j] = q0 + 16 * j + (lane & 15);
            if (q0 != wave * 16 * NB) load_q(q, bh, q0);       // L > 32 * waves only: later blocks are not prefetched
            float m[NB], lsum[NB];
            f32x4 o[NB][4];
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                m[j] = -INFINITY;
                lsum[j] = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) o[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            auto tile = [&](int kb, auto masked_tag) {
                constexpr bool MASKED = decltype(masked_tag)::value;
                f32x4 s0[NB], s1[NB];
                mma_rows(s0, k_lds + kb * RB, off, q);
                mma_rows(s1, k_lds + (kb + 16) * RB, off, q);
                attn_softmax_tile<NB, MASKED, CAUSAL>(s0, s1, m, lsum, o, kb, g, L, qrow, c);
                mma_tr(o, v_lds + kb * RB, off, s0, s1);
            };
            const int qlast = q0 + 16 * NB - 1;
            const int kend = CAUSAL ? min(Lp, (qlast / 32 + 1) * 32) : Lp;
            const int kfull = CAUSAL ? min((L / 32) * 32, (q0 / 32) * 32) : (L / 32) * 32;
            int kb = 0;
            for (; kb < kfull; kb += 32) tile(kb, std::false_type{});
            for (; kb < kend; kb += 32) tile(kb, std::true_type{});
            const bool last_block = q0 + nw * 16 * NB >= L;
            if (last_block) LPI_WAIT_VM0();        // next head's images + Q rows have had this head's tiles to land; BEFORE the stores
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const float ltot = grp_sum(lsum[j]);
                const float inv = 1.0f / ltot;
                f32x4 os[4];
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) os[dt] = o[j][dt] * inv;
                store_row_bf16_t(ctx + ((size_t)b * L + qrow[j]) * ldctx + h * HD, os, g, qrow[j] < L);
                if (qrow[j] < L && g == 0) lse[((size_t)b * H + h) * L + qrow[j]] = (m[j] + log2f(ltot)) * LN2;
            }
        }
        if (wave * 16 * NB >= L) LPI_WAIT_VM0();   // a wave without a query block still owns DMA pieces
        LPI_BARRIER();            // every wave's pieces of the next images landed; this head's images may be overwritten
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) q[j][ks] = qn[j][ks];
    }
}

// ------------------------------------------------------------------------------------------------ backward
// SV16: the saved qkv / ctx are fp16 (f16-mode forward); dctx, dqkv and the MFMA operands are bf16.  LDS-DMA cannot convert, so the K, V
// and Q images are converted IN PLACE once they have landed (each thread its own 16-byte chunks: no hazard), the wave's own Q rows in
// registers; O is only dotted with dO.
// OIMG: the head's O rows (for delta) come through a fifth LDS image, by LDS-DMA with the K, V images, instead of a register prefetch: with
// them the prefetch state pushed phase B over the 256-register budget, and the reloads of the spilled lane addresses made the compiler wait
// vmcnt(0) — i.e. for the next head's whole K, V DMA — in front of phase B (ablation: profiles/r02_gemm_experiments.md, attention section).
template <bool CAUSAL, bool SV16, bool OIMG>
__global__ __launch_bounds__(512) void attn_bwd2_kernel(int L, int Lp, int H, int total, const T* __restrict__ qkv, int ldqkv,
                                                       const T* __restrict__ ctx, int ldctx, const T* __restrict__ dctx, int lddctx,
                                                       const float* __restrict__ lse, float* __restrict__ delta,
                                                       T* __restrict__ dqkv, int lddqkv, int rows_hi) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4;
    const int dm = H * HD;
    const int img = Lp * RB;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem);
    char* const q_lds = smem;
    char* const k_lds = smem + img;
    char* const v_lds = smem + 2 * img;
    char* const do_lds = smem + 3 * img;
    constexpr int NIMG = OIMG ? 5 : 4;
    char* const o_lds = smem + 4 * img;       // OIMG only
    float* const lse_lds = reinterpret_cast<float*>(smem + NIMG * img);
    float* const dl_lds = lse_lds + Lp;
    const RdOff off = make_offsets(lane);
    const float c = SCALE * LOG2E;
    const int q0 = wave * 16 * NB;          // this wave's 32 rows (queries in phase A, keys in phase B): Lp <= 32 * waves
    const bool has_rows = q0 < L;
    // rows_hi: only dQ / dK / dV of token rows < rows_hi are wanted (the first block's backward needs the prompt rows alone); a wave
    // whose 32 rows lie behind them computes delta for its queries (phase B needs every row's) and skips both tile loops and stores
    const bool work = has_rows && q0 < rows_hi;
#ifdef LPI_ABL_ATTN_NOSTORE     /* ablation build: no result stores */
    const bool ABL_ST = lddqkv == 12345;
#else
    constexpr bool ABL_ST = true;
#endif
#ifdef LPI_ABL_ATTN_NOCOMPUTE   /* ablation build: loads, barriers and stores only */
    const bool comp = work && lddqkv == 12345;
#else
    const bool comp = work;
#endif

    zero_pad_rows(smem, NIMG, L, Lp);
    auto head_ptr = [&](int bh) { return qkv + (size_t)(bh / H) * L * ldqkv + (bh % H) * HD; };

    // the wave's own rows of Q, dO, O (phase A operands) and their log-sum-exp, fetched one head ahead
    struct Own {
        Chunk q[NB][KS], d[NB][KS], o[OIMG ? 1 : NB][KS];
        float lq[NB], lse_t;
    };
    auto prefetch_own = [&](Own& w, int bh) {
        const int b = bh / H, h = bh % H;
        const T* qg = head_ptr(bh);
        // the lane's row / chunk indices are re-derived from an opaque copy of the lane id at every call: hoisted to kernel entry (they are
        // loop invariant) the 64-bit lane addresses are SPILLED around the tile loops, and a spill reload waits vmcnt(0), i.e. for the
        // next head's whole K / V DMA, in front of phase B
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int gl = ln >> 4;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int qr = q0 + 16 * j + (ln & 15);
            const bool valid = qr < L;
            const size_t grow = (size_t)b * L + qr;
            glb_rows(w.q[j], qg, qr, ldqkv, gl, valid);
            glb_rows(w.d[j], dctx + h * HD, grow, lddctx, gl, valid);
            if constexpr (!OIMG) glb_rows(w.o[j], ctx + h * HD, grow, ldctx, gl, valid);
            w.lq[j] = valid ? lse[((size_t)b * H + h) * L + qr] : INFINITY;       // raw (natural-log) value; padded queries -> P = 0
        }
        const int i = wave * 64 + ln;
        w.lse_t = (i < L) ? lse[((size_t)b * H + h) * L + i] : INFINITY;          // thread i carries row i of the head's lse vector
    };

    int bh = blockIdx.x;
    Own own;
    prefetch_own(own, bh);
    {
        const T* qg = head_ptr(bh);
        stage_dma(lds0 + img, qg + dm, ldqkv, L, wave, nw, lane);
        stage_dma(lds0 + 2 * img, qg + 2 * dm, ldqkv, L, wave, nw, lane);
        if constexpr (OIMG) stage_dma(lds0 + 4 * img, ctx + (size_t)(bh / H) * L * ldctx + (bh % H) * HD, ldctx, L, wave, nw, lane);
    }
    LPI_WAIT_VM0();
    __syncthreads();

    for (; bh < total; bh += gridDim.x) {
        const int b = bh / H, h = bh % H;
        const int nbh = bh + gridDim.x;
        const T* qg = head_ptr(bh);
        if constexpr (SV16) {       // this head's K, V images have landed (and are visible: barrier B3 / the prologue's): fp16 -> bf16
            convert_image_f16_to_bf16(k_lds, L);
            convert_image_f16_to_bf16(v_lds, L);
#pragma unroll
            for (int j = 0; j < NB; ++j)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) chunk_f16_to_bf16(own.q[j][ks]);
            LPI_BARRIER();
        }
        // The compiler waits for the prefetched own rows (vmcnt(0): it cannot count the LDS-DMA issued from inline asm) at their first use.
        // Make that first use HERE, before this head's Q / dO DMA is issued — where the wait is already satisfied by the vmcnt(0) that ended
        // the last phase B — and not at the top of phase A, where it waited for the whole DMA that phase A is meant to run under.
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                asm volatile("" ::"v"(own.q[j][ks].u.x), "v"(own.q[j][ks].u.y), "v"(own.q[j][ks].u.z), "v"(own.q[j][ks].u.w),
                             "v"(own.d[j][ks].u.x), "v"(own.d[j][ks].u.y), "v"(own.d[j][ks].u.z), "v"(own.d[j][ks].u.w));
                if constexpr (!OIMG) asm volatile("" ::"v"(own.o[OIMG ? 0 : j][ks].u.x), "v"(own.o[OIMG ? 0 : j][ks].u.y), "v"(own.o[OIMG ? 0 : j][ks].u.z), "v"(own.o[OIMG ? 0 : j][ks].u.w));
            }
        asm volatile("" ::"v"(own.lq[0]), "v"(own.lq[1]), "v"(own.lse_t));
        // Q, dO of THIS head -> their images (free since the previous head's phase B), landing under phase A
        stage_dma(lds0, qg, ldqkv, L, wave, nw, lane);
        stage_dma(lds0 + 3 * img, dctx + (size_t)b * L * lddctx + h * HD, lddctx, L, wave, nw, lane);
        if ((int)threadIdx.x < Lp) lse_lds[threadIdx.x] = own.lse_t * LOG2E;      // +inf for padded queries
        __builtin_amdgcn_sched_barrier(0);

        // ---- phase A: this wave's 2 x 16 queries -> delta, dQ
        int qrow[NB];
        float dls[NB], lq[NB], dlt[NB];
        f32x4 dq[NB][4];
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            qrow[j] = q0 + 16 * j + (lane & 15);
            float dl = 0.f;
            Chunk oc[KS];
            if constexpr (OIMG) lds_rows(oc, o_lds, q0 + 16 * j, off);
            else {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) oc[ks] = own.o[OIMG ? 0 : j][ks];
            }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int e = 0; e < 8; ++e) dl += (SV16 ? (float)oc[ks].hh[e] : (float)oc[ks].h[e]) * (float)own.d[j][ks].h[e];
            dl = grp_sum(dl);
            dlt[j] = dl;
            lq[j] = own.lq[j] * LOG2E;
            dls[j] = dl * SCALE;
#pragma unroll
            for (int i = 0; i < 4; ++i) dq[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (comp) {
            auto tile = [&](int kb, auto masked_tag) {
                constexpr bool MASKED = decltype(masked_tag)::value;
                f32x4 s0[NB], s1[NB], p0[NB], p1[NB];
                mma_rows(s0, k_lds + kb * RB, off, own.q);
                mma_rows(s1, k_lds + (kb + 16) * RB, off, own.q);
                mma_rows(p0, v_lds + kb * RB, off, own.d);
                mma_rows(p1, v_lds + (kb + 16) * RB, off, own.d);
#pragma unroll
                for (int j = 0; j < NB; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float e0 = exp2_fast(fmaf(s0[j][r], c, -lq[j]));
                        float e1 = exp2_fast(fmaf(s1[j][r], c, -lq[j]));
                        if constexpr (MASKED) {
                            const int k0 = kb + 4 * g + r, k1 = k0 + 16;
                            if (!(k0 < L && (!CAUSAL || k0 <= qrow[j]))) e0 = 0.f;
                            if (!(k1 < L && (!CAUSAL || k1 <= qrow[j]))) e1 = 0.f;
                        }
                        s0[j][r] = e0 * fmaf(p0[j][r], SCALE, -dls[j]);
                        s1[j][r] = e1 * fmaf(p1[j][r], SCALE, -dls[j]);
                    }
                mma_tr(dq, k_lds + kb * RB, off, s0, s1);
            };
            const int qlast = q0 + 16 * NB - 1;
            const int kend = CAUSAL ? min(Lp, (qlast / 32 + 1) * 32) : Lp;
            const int kfull = CAUSAL ? min((L / 32) * 32, (q0 / 32) * 32) : (L / 32) * 32;
            int kb = 0;
            for (; kb < kfull; kb += 32) tile(kb, std::false_type{});
            for (; kb < kend; kb += 32) tile(kb, std::true_type{});
        }
        LPI_WAIT_VM0();           // Q, dO images landed (they had phase A); BEFORE the dQ stores
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const bool valid = qrow[j] < L;
            if (g == 0 && qrow[j] < Lp) {
                dl_lds[qrow[j]] = valid ? dls[j] : 0.f;     // delta * scale, for phase B
                if (valid) delta[((size_t)b * H + h) * L + qrow[j]] = dlt[j];
            }
            store_row_bf16_t(dqkv + ((size_t)b * L + qrow[j]) * lddqkv + h * HD, dq[j], g, valid && work && ABL_ST);
        }
        LPI_BARRIER();            // B1: Q, dO, lse, delta images complete and visible
        if constexpr (SV16) {
            convert_image_f16_to_bf16(q_lds, L);
            LPI_BARRIER();
        }

        // ---- phase B: this wave's 2 x 16 keys -> dK, dV.  Its own K, V rows move to registers first: the K, V images are then free.
        Chunk kk[NB][KS], vv[NB][KS];
        f32x4 dk[NB][4], dv[NB][4];
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            lds_rows(kk[j], k_lds, q0 + 16 * j, off);
            lds_rows(vv[j], v_lds, q0 + 16 * j, off);
#pragma unroll
            for (int i = 0; i < 4; ++i) { dk[j][i] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[j][i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        }
        LPI_BARRIER();            // B2: every wave holds its K, V rows
        if (nbh < total) {        // next head: K, V -> the freed images; own Q / dO / O rows -> registers; all landing under phase B
            const T* ng = head_ptr(nbh);
            stage_dma(lds0 + img, ng + dm, ldqkv, L, wave, nw, lane);
            stage_dma(lds0 + 2 * img, ng + 2 * dm, ldqkv, L, wave, nw, lane);
            if constexpr (OIMG) stage_dma(lds0 + 4 * img, ctx + (size_t)(nbh / H) * L * ldctx + (nbh % H) * HD, ldctx, L, wave, nw, lane);
            prefetch_own(own, nbh);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (comp) {
            const int k0w = q0;
            int krow[NB];
#pragma unroll
            for (int j = 0; j < NB; ++j) krow[j] = k0w + 16 * j + (lane & 15);
            auto tile = [&](int qb, auto masked_tag) {
                constexpr bool MASKED = decltype(masked_tag)::value;
                f32x4 s0[NB], s1[NB], p0[NB], p1[NB], e0[NB], e1[NB];
                mma_rows(s0, q_lds + qb * RB, off, kk);
                mma_rows(s1, q_lds + (qb + 16) * RB, off, kk);
                mma_rows(p0, do_lds + qb * RB, off, vv);
                mma_rows(p1, do_lds + (qb + 16) * RB, off, vv);
                const f32x4 l0 = *reinterpret_cast<const f32x4*>(lse_lds + qb + 4 * g);
                const f32x4 l1 = *reinterpret_cast<const f32x4*>(lse_lds + qb + 16 + 4 * g);
                const f32x4 d0 = *reinterpret_cast<const f32x4*>(dl_lds + qb + 4 * g);
                const f32x4 d1 = *reinterpret_cast<const f32x4*>(dl_lds + qb + 16 + 4 * g);
#pragma unroll
                for (int j = 0; j < NB; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        e0[j][r] = exp2_fast(fmaf(s0[j][r], c, -l0[r]));
                        e1[j][r] = exp2_fast(fmaf(s1[j][r], c, -l1[r]));
                        if constexpr (MASKED) {
                            const int qa = qb + 4 * g + r, qc = qa + 16;
                            if (krow[j] > qa) e0[j][r] = 0.f;
                            if (krow[j] > qc) e1[j][r] = 0.f;
                        }
                        s0[j][r] = e0[j][r] * fmaf(p0[j][r], SCALE, -d0[r]);
                        s1[j][r] = e1[j][r] * fmaf(p1[j][r], SCALE, -d1[r]);
                    }
                mma_tr(dv, do_lds + qb * RB, off, e0, e1);
                mma_tr(dk, q_lds + qb * RB, off, s0, s1);
            };
            int qb = CAUSAL ? (k0w / 32) * 32 : 0;
            if constexpr (CAUSAL) {
                const int qdiag = min(Lp, ((k0w + 16 * NB - 1) / 32 + 1) * 32);
                for (; qb < qdiag; qb += 32) tile(qb, std::true_type{});
            }
            for (; qb < Lp; qb += 32) tile(qb, std::false_type{});
        }
        LPI_WAIT_VM0();           // next head's K, V images and own rows landed (they had phase B); BEFORE the dK / dV stores
        if (work) {
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int kr = q0 + 16 * j + (lane & 15);
                T* dst = dqkv + ((size_t)b * L + kr) * lddqkv + h * HD;
                store_row_bf16_t(dst + dm, dk[j], g, kr < L && ABL_ST);
                store_row_bf16_t(dst + 2 * dm, dv[j], g, kr < L && ABL_ST);
            }
        }
        LPI_BARRIER();            // B3: phase B's reads of the Q / dO / lse / delta images are done; the next K, V images are visible
    }
}

// ------------------------------------------------------------------------------------------------ backward, third generation
// ONE pass per head instead of two: the two-phase backward recomputes S, P, dP and dS once per orientation (phase A: scores with the
// query as column, to contract dS over KEYS for dQ; phase B: with the key as column, to contract over QUERIES for dK, dV) — 56 MFMAs and
// two exponentials per score.  Here a wave owns 32 keys (K, V rows in registers, K^T fragments too) and walks the query tiles once:
// the phase-B arithmetic gives dK, dV, and the same dS tile, transposed through a 2 KiB per-wave LDS scratch (written as [key][query]
// bf16 rows, read back with ds_read_b64_tr_b16 — the transposing read — as the B operand), gives this wave's dQ contribution
// K_own^T . dS^T for that query tile.  The contributions of the seven waves to a query tile are added in a FIXED order in an LDS
// buffer: at step t wave w works on tile (w + t) mod n, so no two waves touch a tile in the same step and the order of additions is
// the same in every run (no atomics: bitwise reproducible).  40 MFMAs and one exponential per score (-29 % / -50 %).
// LDS: Q, dO, K images (3 x Lp x 128 B) + dQ accumulators (Lp x 64 f32) + 2 KiB scratch per wave + lse / delta vectors = 155.8 KiB at
// L = 213: one workgroup per CU; the next head's images are issued as soon as the last step is done (under the dK / dV stores and the
// dQ write-out).  Non-causal only (the vision tower: the text tower's short, causal heads stay on the first generation).
template <bool SV16>
__global__ __launch_bounds__(512) void attn_bwd3_kernel(int L, int Lp, int H, int total, const T* __restrict__ qkv, int ldqkv,
                                                       const T* __restrict__ ctx, int ldctx, const T* __restrict__ dctx, int lddctx,
                                                       const float* __restrict__ lse, float* __restrict__ delta,
                                                       T* __restrict__ dqkv, int lddqkv) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, nw = blockDim.x >> 6;      // nw == Lp / 32: one 32-row block per wave, as keys AND as dQ tile owner
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4;
    const int dm = H * HD;
    const int img = Lp * RB;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem);
    char* const q_lds = smem;
    char* const do_lds = smem + img;
    char* const k_lds = smem + 2 * img;
    float* const dq_lds = reinterpret_cast<float*>(smem + 3 * img);                   // [tile][qsub][dt][lane] f32x4
    char* const scr = smem + 3 * img + Lp * HD * 4 + wave * 2048;                     // this wave's [32 keys][32 queries] bf16, swizzled
    float* const lse_lds = reinterpret_cast<float*>(smem + 3 * img + Lp * HD * 4 + nw * 2048);
    float* const dl_lds = lse_lds + Lp;
    const RdOff off = make_offsets(lane);
    const float c = SCALE * LOG2E;
    const int k0 = wave * 32;                 // this wave's keys / its rows in the delta pre-pass
    // scratch addressing: 64-byte rows (32 queries x bf16), 8-byte slot index XOR ((row >> 2 & 1) << 2 | (row >> 3 & 1) << 1):
    // conflict-free for the transposing reads (tools/lds_swizzle_check.py)
    auto sw = [](int row) { return (((row >> 2) & 1) << 2) | (((row >> 3) & 1) << 1); };
    int scr_w[NB][2];                         // write: row = 16 j + (lane & 15), slot = 4 tile + g
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int row = 16 * j + (lane & 15);
            scr_w[j][t] = row * 64 + (((4 * t + g) ^ sw(row)) << 3);
        }
    int scr_r[2][2];                          // transposing read: row = 16 hi + 4 g + ((lane & 15) >> 2), slot = 4 qsub + (lane & 3)
#pragma unroll
    for (int qs = 0; qs < 2; ++qs)
#pragma unroll
        for (int hi = 0; hi < 2; ++hi) {
            const int row = 16 * hi + 4 * g + ((lane & 15) >> 2);
            scr_r[qs][hi] = row * 64 + (((4 * qs + (lane & 3)) ^ sw(row)) << 3);
        }

    zero_pad_rows(smem, 3, L, Lp);
    auto head_ptr = [&](int bh) { return qkv + (size_t)(bh / H) * L * ldqkv + (bh % H) * HD; };
    auto stage_head = [&](int bh) {
        const int b = bh / H, h = bh % H;
        const T* qg = head_ptr(bh);
        stage_dma(lds0 + 2 * img, qg + dm, ldqkv, L, wave, nw, lane);
        stage_dma(lds0, qg, ldqkv, L, wave, nw, lane);
        stage_dma(lds0 + img, dctx + (size_t)b * L * lddctx + h * HD, lddctx, L, wave, nw, lane);
    };

    int bh = blockIdx.x;
    stage_head(bh);
    for (; bh < total; bh += gridDim.x) {
        const int b = bh / H, h = bh % H;
        const T* qg = head_ptr(bh);
        // this wave's V rows and O rows (delta) straight from global memory, its lse row values; then everything has landed
        Chunk vv[NB][KS], oc[NB][KS];
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int r = k0 + 16 * j + (lane & 15);
            glb_rows(vv[j], qg + 2 * dm, r, ldqkv, g, r < L);
            glb_rows(oc[j], ctx + h * HD, (size_t)b * L + r, ldctx, g, r < L);
        }
        const int ti = threadIdx.x;
        const float lse_t = (ti < L) ? lse[((size_t)b * H + h) * L + ti] : INFINITY;      // padded queries -> P = 0
        LPI_WAIT_VM0();
        if (ti < Lp) lse_lds[ti] = lse_t * LOG2E;
        LPI_BARRIER();
        if constexpr (SV16) {
            convert_image_f16_to_bf16(k_lds, L);
            convert_image_f16_to_bf16(q_lds, L);
#pragma unroll
            for (int j = 0; j < NB; ++j)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) chunk_f16_to_bf16(vv[j][ks]);
            LPI_BARRIER();
        }
        // delta = rowsum(dO * O) of this wave's 32 rows -> dl_lds (x scale) and global; K rows and K^T fragments of its 32 keys
        Chunk kk[NB][KS], kT[4];
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            Chunk dr[KS];
            lds_rows(dr, do_lds, k0 + 16 * j, off);
            float dl = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int e = 0; e < 8; ++e) dl += (SV16 ? (float)oc[j][ks].hh[e] : (float)oc[j][ks].h[e]) * (float)dr[ks].h[e];
            dl = grp_sum(dl);
            const int r = k0 + 16 * j + (lane & 15);
            if (g == 0) {
                dl_lds[r] = r < L ? dl * SCALE : 0.f;
                if (r < L) delta[((size_t)b * H + h) * L + r] = dl;
            }
            lds_rows(kk[j], k_lds, k0 + 16 * j, off);
        }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const char* win = k_lds + k0 * RB;
            short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(win + off.tr[dt]));
            short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(win + 16 * RB + off.tr[dt]));
            const uint2 lo2 = __builtin_bit_cast(uint2, lo), hi2 = __builtin_bit_cast(uint2, hi);
            kT[dt].u = make_uint4(lo2.x, lo2.y, hi2.x, hi2.y);
        }
        f32x4 dk[NB][4], dv[NB][4];
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) { dk[j][i] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[j][i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        LPI_BARRIER();            // lse / delta vectors complete

        for (int t = 0; t < nw; ++t) {
            int tile = wave + t;
            if (tile >= nw) tile -= nw;
            const int qb = tile * 32;
            f32x4 s0[NB], s1[NB], p0[NB], p1[NB], e0[NB], e1[NB];
            mma_rows(s0, q_lds + qb * RB, off, kk);
            mma_rows(s1, q_lds + (qb + 16) * RB, off, kk);
            mma_rows(p0, do_lds + qb * RB, off, vv);
            mma_rows(p1, do_lds + (qb + 16) * RB, off, vv);
            const f32x4 l0 = *reinterpret_cast<const f32x4*>(lse_lds + qb + 4 * g);
            const f32x4 l1 = *reinterpret_cast<const f32x4*>(lse_lds + qb + 16 + 4 * g);
            const f32x4 d0 = *reinterpret_cast<const f32x4*>(dl_lds + qb + 4 * g);
            const f32x4 d1 = *reinterpret_cast<const f32x4*>(dl_lds + qb + 16 + 4 * g);
#pragma unroll
            for (int j = 0; j < NB; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    e0[j][r] = exp2_fast(fmaf(s0[j][r], c, -l0[r]));
                    e1[j][r] = exp2_fast(fmaf(s1[j][r], c, -l1[r]));
                    s0[j][r] = e0[j][r] * fmaf(p0[j][r], SCALE, -d0[r]);
                    s1[j][r] = e1[j][r] * fmaf(p1[j][r], SCALE, -d1[r]);
                }
            // dS tile -> this wave's scratch as [key][query] bf16 (one 8-byte store per accumulator tile)
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                *reinterpret_cast<uint2*>(scr + scr_w[j][0]) = make_uint2(pack2(s0[j][0], s0[j][1]), pack2(s0[j][2], s0[j][3]));
                *reinterpret_cast<uint2*>(scr + scr_w[j][1]) = make_uint2(pack2(s1[j][0], s1[j][1]), pack2(s1[j][2], s1[j][3]));
            }
            mma_tr(dv, do_lds + qb * RB, off, e0, e1);
            mma_tr(dk, q_lds + qb * RB, off, s0, s1);
            // dQ^T[d, q] contribution of this wave's keys for the tile's two 16-query blocks, added into the LDS accumulators
#pragma unroll
            for (int qs = 0; qs < 2; ++qs) {
                short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(scr + scr_r[qs][0]));
                short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(scr + scr_r[qs][1]));
                Chunk bq;
                const uint2 lo2 = __builtin_bit_cast(uint2, lo), hi2 = __builtin_bit_cast(uint2, hi);
                bq.u = make_uint4(lo2.x, lo2.y, hi2.x, hi2.y);
                float* acc_p = dq_lds + ((size_t)(tile * 2 + qs) * 4 * 64 + lane) * 4;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
                    mma_chunk<T>(a, kT[dt], bq);
                    if (t != 0) a += *reinterpret_cast<const f32x4*>(acc_p + dt * 256);
                    *reinterpret_cast<f32x4*>(acc_p + dt * 256) = a;
                }
            }
            LPI_BARRIER();        // step t done by every wave: tile (w + t + 1) mod n is free for wave w
        }
        // the images are free: the next head's are issued now and land under the stores below
        const int nbh = bh + gridDim.x;
        if (nbh < total) stage_head(nbh);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int kr = k0 + 16 * j + (lane & 15);
            T* dst = dqkv + ((size_t)b * L + kr) * lddqkv + h * HD;
            store_row16_t<T>(dst + dm, dk[j], g, kr < L);
            store_row16_t<T>(dst + 2 * dm, dv[j], g, kr < L);
        }
        // dQ write-out: wave w converts and stores tile w (the accumulator layout is the transposed-tile layout of store_row16_t)
#pragma unroll
        for (int qs = 0; qs < 2; ++qs) {
            const float* acc_p = dq_lds + ((size_t)(wave * 2 + qs) * 4 * 64 + lane) * 4;
            f32x4 o[4];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) o[dt] = *reinterpret_cast<const f32x4*>(acc_p + dt * 256);
            const int qr = k0 + 16 * qs + (lane & 15);
            store_row16_t<T>(dqkv + ((size_t)b * L + qr) * lddqkv + h * HD, o, g, qr < L);
        }
    }
}

int set_lds2(LdsOnce& once, const void* kern) { return lpi_ensure_lds(once, kern, 160 * 1024); }

int cu_count() {
    static int n = 0;
    if (n == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        n = v;
    }
    return n;
}

// waves per workgroup: each wave owns NB 16-row blocks per round; balance the rounds (as attention.hip)
inline int pick_waves2(int L) {
    const int nqb = (L + 16 * NB - 1) / (16 * NB);
    const int rounds = (nqb + 7) / 8;
    return (nqb + rounds - 1) / rounds;
}

}  // namespace

// true if the second-generation kernels take this shape (bf16, every wave owns at most one 32-row block in the backward)
bool lpi_attn2_fwd_ok(int L) { return L >= 1 && (size_t)4 * ((L + 31) / 32 * 32) * RB <= 160 * 1024; }
bool lpi_attn2_bwd_ok(int L) {
    const int Lp = (L + 31) / 32 * 32;
    return L >= 1 && Lp <= 256 && (size_t)4 * Lp * RB + 2 * Lp * sizeof(float) <= 160 * 1024;
}

int lpi_attn2_fwd(int B, int L, int H, const void* qkv, int ldqkv, void* ctx, int ldctx, float* lse, int causal, hipStream_t s) {
    const int Lp = (L + 31) / 32 * 32;
    const size_t lds = (size_t)4 * Lp * RB;
    const int thr = 64 * pick_waves2(L);
    const int total = B * H;
    const int per_cu = (int)std::min<size_t>(4, std::max<size_t>(1, (160 * 1024) / lds));
    const int grid = std::min(total, cu_count() * per_cu);
    static LdsOnce o0, o1;
    if (causal) {
        if (int e = set_lds2(o1, (const void*)attn_fwd2_kernel<true>)) return e;
        LPI_LAUNCH((attn_fwd2_kernel<true>), dim3(grid), dim3(thr), lds, s, L, Lp, H, total, (const T*)qkv, ldqkv, (T*)ctx, ldctx, lse);
    } else {
        if (int e = set_lds2(o0, (const void*)attn_fwd2_kernel<false>)) return e;
        LPI_LAUNCH((attn_fwd2_kernel<false>), dim3(grid), dim3(thr), lds, s, L, Lp, H, total, (const T*)qkv, ldqkv, (T*)ctx, ldctx, lse);
    }
    LPI_CHECK_LAST();
    return 0;
}

// third generation (single pass): non-causal, one workgroup per CU
bool lpi_attn3_bwd_ok(int L, int causal) {
    const int Lp = (L + 31) / 32 * 32;
    return !causal && L >= 1 && Lp <= 256 && (size_t)3 * Lp * RB + (size_t)Lp * HD * 4 + (size_t)(Lp / 32) * 2048 + 2 * Lp * sizeof(float) <= 160 * 1024;
}
int lpi_attn3_bwd(int B, int L, int H, const void* qkv, int ldqkv, const void* ctx, int ldctx, const void* dctx, int lddctx,
                  const float* lse, float* delta, void* dqkv, int lddqkv, hipStream_t s, int saved_f16) {
    const int Lp = (L + 31) / 32 * 32;
    const size_t lds = (size_t)3 * Lp * RB + (size_t)Lp * HD * 4 + (size_t)(Lp / 32) * 2048 + 2 * Lp * sizeof(float);
    const int thr = 64 * (Lp / 32);
    const int total = B * H;
    const int per_cu = (int)std::min<size_t>(4, std::max<size_t>(1, (160 * 1024) / lds));
    const int grid = std::min(total, cu_count() * per_cu);
    static LdsOnce o0, o1;
    if (saved_f16) {
        if (int e = set_lds2(o1, (const void*)attn_bwd3_kernel<true>)) return e;
        LPI_LAUNCH((attn_bwd3_kernel<true>), dim3(grid), dim3(thr), lds, s, L, Lp, H, total, (const T*)qkv, ldqkv, (const T*)ctx, ldctx,
                   (const T*)dctx, lddctx, lse, delta, (T*)dqkv, lddqkv);
    } else {
        if (int e = set_lds2(o0, (const void*)attn_bwd3_kernel<false>)) return e;
        LPI_LAUNCH((attn_bwd3_kernel<false>), dim3(grid), dim3(thr), lds, s, L, Lp, H, total, (const T*)qkv, ldqkv, (const T*)ctx, ldctx,
                   (const T*)dctx, lddctx, lse, delta, (T*)dqkv, lddqkv);
    }
    LPI_CHECK_LAST();
    return 0;
}

int lpi_attn2_bwd(int B, int L, int H, const void* qkv, int ldqkv, const void* ctx, int ldctx, const void* dctx, int lddctx,
                  const float* lse, float* delta, void* dqkv, int lddqkv, int causal, hipStream_t s, int saved_f16, int rows_hi) {
    const int Lp = (L + 31) / 32 * 32;
    const bool oimg = (size_t)5 * Lp * RB + (size_t)2 * Lp * sizeof(float) <= 160 * 1024 && g_lpi_tuning[10] == 0;      // key 10 = 1: O rows by register prefetch (A/B)
    const size_t lds = (size_t)(oimg ? 5 : 4) * Lp * RB + (size_t)2 * Lp * sizeof(float);
    const int thr = 64 * ((Lp + 31) / 32);       // one 32-row block per wave
    const int total = B * H;
    const int per_cu = (int)std::min<size_t>(4, std::max<size_t>(1, (160 * 1024) / lds));
    const int grid = std::min(total, cu_count() * per_cu);
    static LdsOnce o0, o1, o2, o3, o4, o5, o6, o7;
#define BWD2(C, S, I, O)                                                                                                                    \
    do {                                                                                                                                    \
        if (int e = set_lds2(O, (const void*)attn_bwd2_kernel<C, S, I>)) return e;                                                          \
        LPI_LAUNCH((attn_bwd2_kernel<C, S, I>), dim3(grid), dim3(thr), lds, s, L, Lp, H, total, (const T*)qkv, ldqkv, (const T*)ctx, ldctx, \
                   (const T*)dctx, lddctx, lse, delta, (T*)dqkv, lddqkv, rows_hi);                                                          \
    } while (0)
    if (oimg) {
        if (causal && saved_f16) BWD2(true, true, true, o4);
        else if (causal) BWD2(true, false, true, o5);
        else if (saved_f16) BWD2(false, true, true, o6);
        else BWD2(false, false, true, o7);
    } else if (causal && saved_f16) BWD2(true, true, false, o3);
    else if (causal) BWD2(true, false, false, o1);
    else if (saved_f16) BWD2(false, true, false, o2);
    else BWD2(false, false, false, o0);
#undef BWD2
    LPI_CHECK_LAST();
    return 0;
}
