// One 256x128 output tile (8 waves of 64x64, 3-stage LDS ring, two staggered phases per K-tile; see gemm256x128.hip), as a device
// function shared by the stand-alone kernel and the tail rounds of gemm256.hip's hybrid kernel.
#pragma once
#include "common.h"
#include "gemm_epilogue.h"

namespace t128 {

constexpr int TM = 256, TN = 128;
constexpr int ROWB = 128;
constexpr int STAGE_BYTES = (TM + TN) * ROWB;   // 48 KiB
constexpr int NSTAGE = 3;
constexpr int NTHR = 512;
constexpr int ERS = TN * 4 + 16;                // epilogue staging row: 128 f32 + 16 B pad (conflict-free ds_write_b128)
constexpr int LDS_BYTES = NSTAGE * STAGE_BYTES; // 144 KiB >= 128 rows x 528 B of epilogue staging


template <typename T, typename TC, int EPI, bool RES, bool SAVE_U>
__device__ __forceinline__ void tile(int m0, int n0, int K, const T* __restrict__ A, int lda, const T* __restrict__ B, int ldb,
                                     TC* __restrict__ C, int ldc, const float* __restrict__ bias, const float* __restrict__ residual, int ldr,
                                     typename AuxT<T>::type* __restrict__ aux, int ldaux, float alpha, char* smem)
{
    constexpr int EPC = Elem<T>::EPC;
    constexpr int BK = ROWB / (int)sizeof(T);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int grp = wave >> 2;      // waves 4-7 = SIMD partners of waves 0-3

    // staging: thread t, instr i -> stage byte i*8192 + t*16: row = i*64 + t/8 (rows 0..255 = A, 256..383 = B), swizzled chunk
    const int srow = tid >> 3;
    const int schunk = (tid & 7) ^ ((tid >> 4) & 7);
    const T* a_src = A + (size_t)(m0 + srow) * lda + schunk * EPC;
    const T* b_src = B + (size_t)(n0 + srow) * ldb + schunk * EPC;
    const size_t a_i = (size_t)64 * lda, b_i = (size_t)64 * ldb;
    const unsigned lds_w = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + wave * 1024);
    auto glds16 = [&](const T* src, unsigned lds_addr) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(lds_addr) : "memory");
    };
    auto stage = [&](int kt, int st) {
        const unsigned base = lds_w + st * STAGE_BYTES;
        const T* ap = a_src + (size_t)kt * BK;
        const T* bp = b_src + (size_t)kt * BK;
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16(ap + i * a_i, base + i * 8192);
#pragma unroll
        for (int i = 0; i < 2; ++i) glds16(bp + i * b_i, base + (4 + i) * 8192);
    };

    const int frow = lane & 15, fg = lane >> 4, fsw = frow >> 1;
    int foff[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) foff[ks] = frow * ROWB + (((ks << 2) | fg) ^ fsw) * 16;
    const int a_base = (wm * 64) * ROWB;
    const int b_base = TM * ROWB + (wn * 64) * ROWB;

    f32x4 acc[4][4];   // [ni][mi]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    Chunk fa[4], fb[4];
    auto read_frags = [&](const char* st, int ks) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            fb[i].u = *reinterpret_cast<const uint4*>(st + b_base + i * 16 * ROWB + foff[ks]);
            fa[i].u = *reinterpret_cast<const uint4*>(st + a_base + i * 16 * ROWB + foff[ks]);
        }
    };
    auto mma_all = [&]() {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) mma_chunk<T>(acc[ni][mi], fb[ni], fa[mi]);
        __builtin_amdgcn_s_setprio(0);
    };
#define SYNC_IN()                                         \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    \
    __builtin_amdgcn_s_barrier();                         \
    __builtin_amdgcn_sched_barrier(0)
#define SYNC_OUT()                                        \
    __builtin_amdgcn_sched_barrier(0);                    \
    __builtin_amdgcn_s_barrier();                         \
    asm volatile("" ::: "memory")

    const int nk = K / BK;   // >= 2 (host checked)
    stage(0, 0);
    stage(1, 1);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");     // K-tile 0 landed, K-tile 1 in flight
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();

    int st = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const char* sp = smem + st * STAGE_BYTES;
        const int st2 = st == 0 ? 2 : st - 1;      // (kt + 2) % 3 == (kt - 1) % 3
        // phase 1: k-step 0; restage the ring slot K-tile kt-1 vacated
        read_frags(sp, 0);
        if (kt + 2 < nk) stage(kt + 2, st2);
        SYNC_IN();
        mma_all();
        SYNC_OUT();
        // phase 2: k-step 1; make sure K-tile kt+1 has landed before anyone reads it next iteration
        read_frags(sp, 1);
        if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        SYNC_IN();
        mma_all();
        SYNC_OUT();
        st = st == 2 ? 0 : st + 1;
    }
#undef SYNC_IN
#undef SYNC_OUT
    if (grp == 0) __builtin_amdgcn_s_barrier();

    // ---- epilogue through LDS: two passes of 128 rows; every global access is a whole 128-column tile row ------------------
    const int lrow = lane & 15, lcol = (lane >> 4) << 2;
    const int ecol = n0 + (lane & 31) * 4;
    f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
    if (bias) bv = *reinterpret_cast<const f32x4*>(bias + ecol);
    f32x4 c1v = f32x4{0.f, 0.f, 0.f, 0.f};      // LayerNorm-fold epilogues: c1 sits behind the two statistics vectors (include/lpi_hip.h)
    if constexpr (EPI == LPI_EPI_LN || EPI == LPI_EPI_LN_QUICKGELU) c1v = *reinterpret_cast<const f32x4*>(residual + 2 * (size_t)ldr + ecol);
#pragma unroll
    for (int ph = 0; ph < 2; ++ph) {
        if (ph) __builtin_amdgcn_s_barrier();
        if ((wm >> 1) == ph) {          // waves whose 64 rows fall in this 128-row pass
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
                    *reinterpret_cast<f32x4*>(smem + ((wm & 1) * 64 + mi * 16 + lrow) * ERS + (wn * 64 + ni * 16 + lcol) * 4) = acc[ni][mi];
        }
        __syncthreads();
        const int r0 = wave * 16 + (lane >> 5);      // this lane's first row within the pass; a wave instruction covers 2 rows
        // LPI_EPI_RES_ROWSTATS (the fp16 residual epilogue with SAVE_U set, as in gemm256p.hip): a 32-lane half owns one 128-column row per rr, i.e.
        // the whole slot n0 / 128 of eight rows per pass — the same halving exchange, one lane of each quad stores the sums
        constexpr bool STATS = EPI == LPI_EPI_NONE && RES && SAVE_U && __is_same(TC, f16_t);
        if constexpr (STATS) {
            float st_s[8], st_q[8];
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) {
                const int r = r0 + 2 * rr;
                const f32x4 v = *reinterpret_cast<const f32x4*>(smem + r * ERS + (lane & 31) * 16);
                const f32x4 o = gemm_epilogue_store<T, TC, EPI, RES, false>(v, m0 + ph * 128 + r, ecol, C, ldc, bv, alpha, residual, ldr, nullptr, 0);
                f16x4_sum_sumsq(pack2_t<f16_t>(o[0], o[1]), pack2_t<f16_t>(o[2], o[3]), st_s[rr], st_q[rr]);      // of the values as stored
            }
            float so, qo;
            rowstats8_half_reduce(st_s, st_q, lane, so, qo);
            const int r = r0 + 2 * ((lane >> 2) & 7);
            float* sp = reinterpret_cast<float*>(aux) + (size_t)(2 * (n0 >> 7)) * ldaux + (m0 + ph * 128 + r);
            if ((lane & 3) == 0) { sp[0] = so; sp[ldaux] = qo; }
        } else {
#pragma unroll 4
            for (int rr = 0; rr < 8; ++rr) {
                const int r = r0 + 2 * rr;
                const f32x4 v = *reinterpret_cast<const f32x4*>(smem + r * ERS + (lane & 31) * 16);
                float mu = 0.f, rs = 1.f;
                if constexpr (EPI == LPI_EPI_LN || EPI == LPI_EPI_LN_QUICKGELU) { mu = residual[m0 + ph * 128 + r]; rs = residual[(size_t)ldr + m0 + ph * 128 + r]; }
                gemm_epilogue_store<T, TC, EPI, RES, SAVE_U>(v, m0 + ph * 128 + r, ecol, C, ldc, bv, alpha, residual, ldr, aux, ldaux, c1v, mu, rs);
            }
        }
    }
}

}  // namespace t128
