// One 256x256 output tile of the phased NT GEMM (see gemm256.hip for the schedule), as a device function so that the plain
// kernel and the tail-splitting kernel share it.
#pragma once
#ifndef LPI_TWO_PHASE
#define LPI_TWO_PHASE 1
#endif
#include "common.h"
#include "gemm_epilogue.h"

namespace t256 {

#ifndef EPI_UNROLL
#define EPI_UNROLL 4
#endif
constexpr int T256 = 256;
constexpr int ROWB = 128;                 // bytes per staged row
constexpr int HALF_BYTES = 128 * ROWB;    // 16 KiB
constexpr int BUF_BYTES = 4 * HALF_BYTES; // 64 KiB per K-tile
constexpr int NTHR = 512;
constexpr int LDS_BYTES = 128 * 1040;      // max(2 K-tile buffers = 131072, epilogue staging 128 rows x 1040 B = 133120)
constexpr int OFF_A0 = 0, OFF_A1 = HALF_BYTES, OFF_B0 = 2 * HALF_BYTES, OFF_B1 = 3 * HALF_BYTES;


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
    const int wm = wave >> 2, wn = wave & 3;

    // ---- staging: a half tile = 128 rows x 128 B = 2 LDS-DMA instructions of 512 lanes x 16 B ---------------------
    // thread t, instr i -> LDS byte i*8192 + t*16: row = i*64 + t/8, phys chunk = t%8, logical = phys ^ ((row>>1)&7)
    const int srow = tid >> 3;
    const int schunk = (tid & 7) ^ ((tid >> 4) & 7);
    const T* a_src = A + (size_t)(m0 + srow) * lda + schunk * EPC;
    const T* b_src = B + (size_t)(n0 + srow) * ldb + schunk * EPC;
    const size_t a_i = (size_t)64 * lda, b_i = (size_t)64 * ldb, a_h = (size_t)128 * lda, b_h = (size_t)128 * ldb;

    // LDS-DMA issued from inline asm: hipcc models the builtin as an LDS write and would put s_waitcnt vmcnt(0) in front of
    // every later ds_read, draining the queue each phase; in asm only the counted vmcnt below orders it (guide section 5.7).
    const unsigned lds_w = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + wave * 1024);
    auto glds16 = [&](const T* src, unsigned lds_addr) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(lds_addr) : "memory");
    };
    auto stage_half = [&](const T* src, size_t istep, int lds_off) {
        glds16(src, lds_w + lds_off);
        glds16(src + istep, lds_w + lds_off + 8192);
    };
    auto stage_A = [&](int kt, int h, int buf) { stage_half(a_src + (size_t)kt * BK + h * a_h, a_i, buf * BUF_BYTES + (h ? OFF_A1 : OFF_A0)); };
    auto stage_B = [&](int kt, int h, int buf) { stage_half(b_src + (size_t)kt * BK + h * b_h, b_i, buf * BUF_BYTES + (h ? OFF_B1 : OFF_B0)); };

    // ---- fragment offsets within a half tile ---------------------------------------------------------------------
    const int frow = lane & 15, fg = lane >> 4, fsw = frow >> 1;
    int foff[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) foff[ks] = frow * ROWB + (((ks << 2) | fg) ^ fsw) * 16;
    const int a_base = (wm * 64) * ROWB;   // + mi*16 rows
    const int b_base = (wn * 32) * ROWB;   // + ni*16 rows

    Chunk fa[4][2], fb0[2][2], fb1[2][2];
    f32x4 acc[2][2][2][4];   // [nh][ni][mh][mi]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int d = 0; d < 4; ++d) acc[a][b][c][d] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto read_A = [&](const char* half) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) fa[mi][ks].u = *reinterpret_cast<const uint4*>(half + a_base + mi * 16 * ROWB + foff[ks]);
    };
    auto read_B = [&](Chunk (&fb)[2][2], const char* half) {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) fb[ni][ks].u = *reinterpret_cast<const uint4*>(half + b_base + ni * 16 * ROWB + foff[ks]);
    };
    auto mma_quadrant = [&](f32x4 (&c)[2][2][2][4], int nh, int mh, const Chunk (&fb)[2][2]) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) mma_chunk<T>(c[nh][ni][mh][mi], fb[ni][ks], fa[mi][ks]);
        __builtin_amdgcn_s_setprio(0);
    };
// lgkmcnt(0) BEFORE the barrier: with the two wave groups staggered by one barrier (below), the other group restages a half
// tile right after this barrier, so this group's ds_reads of it must already have completed (WAR).
#define PHASE_SYNC_IN()                                   \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    \
    __builtin_amdgcn_s_barrier();                         \
    __builtin_amdgcn_sched_barrier(0)
#define PHASE_SYNC_OUT()                                  \
    __builtin_amdgcn_sched_barrier(0);                    \
    __builtin_amdgcn_s_barrier();                         \
    asm volatile("" ::: "memory")

    const int nk = K / BK;   // even, >= 2 (checked on the host)
#if LPI_TWO_PHASE
    // Two phases per K-tile (32 MFMAs each, 4 barriers per K-tile instead of 8; four half tiles in flight instead of three).
    // X: quadrants (n0,m0),(n1,m0) from B0, B1, A0;   Y: quadrants (n1,m1),(n0,m1) from A1 (B fragments stay in registers).
    // LDS-DMA: A1(kt+1) is issued in X(kt) (its slot was last read in Y(kt-1)); A0, B0, B1 of kt+2 in Y(kt) (read in X(kt)).
    // Each wave waits for its own DMAs with a counted vmcnt BEFORE the phase's first barrier, so that with the one-barrier stagger
    // both groups' data is visible when the reading phase starts: X waits for A1(kt) (8 younger instructions may stay in flight),
    // Y for A0, B0, B1 of kt+1.
    stage_A(0, 0, 0); stage_B(0, 0, 0); stage_B(0, 1, 0); stage_A(0, 1, 0);
    stage_A(1, 0, 1); stage_B(1, 0, 1); stage_B(1, 1, 1);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wm == 1) __builtin_amdgcn_s_barrier();
    auto ktile = [&](int kt, const int BUF) {
        const char* buf = smem + BUF * BUF_BYTES;
        const bool more1 = kt + 1 < nk, more2 = kt + 2 < nk;
        // X
        read_B(fb0, buf + OFF_B0);
        read_B(fb1, buf + OFF_B1);
        __builtin_amdgcn_sched_barrier(0);
        read_A(buf + OFF_A0);
        if (more1) {
            stage_A(kt + 1, 1, BUF ^ 1);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        PHASE_SYNC_IN();
        mma_quadrant(acc, 0, 0, fb0);
        mma_quadrant(acc, 1, 0, fb1);
        PHASE_SYNC_OUT();
        // Y
        read_A(buf + OFF_A1);
        if (more2) {
            stage_A(kt + 2, 0, BUF); stage_B(kt + 2, 0, BUF); stage_B(kt + 2, 1, BUF);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else if (more1) {
            asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        PHASE_SYNC_IN();
        mma_quadrant(acc, 1, 1, fb1);
        mma_quadrant(acc, 0, 1, fb0);
        PHASE_SYNC_OUT();
    };
#else
    // prologue: K-tile 0 (4 halves) -> buffer 0, first three halves of K-tile 1 -> buffer 1
    stage_A(0, 0, 0); stage_B(0, 0, 0); stage_B(0, 1, 0); stage_A(0, 1, 0);
    stage_A(1, 0, 1); stage_B(1, 0, 1); stage_B(1, 1, 1);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // Stagger: waves 4-7 (the SIMD partners of waves 0-3) run one barrier behind, so that while one group issues its
    // ds_reads / LDS-DMA the other group's MFMAs own the matrix pipe (MI355X_MICROARCH.md, two waves per SIMD, item 9).
    // Every wave must execute the same number of barriers: group 0 pays its extra one after the loop.
    if (wm == 1) __builtin_amdgcn_s_barrier();

    // one K-tile = 4 phases; BUF is a compile-time constant so every LDS address folds to base + immediate
    auto ktile = [&](int kt, const int BUF) {
        const char* buf = smem + BUF * BUF_BYTES;
        const bool more1 = kt + 1 < nk, more2 = kt + 2 < nk;
        // P1
        read_B(fb0, buf + OFF_B0);
        __builtin_amdgcn_sched_barrier(0);
        read_A(buf + OFF_A0);
        if (more1) stage_A(kt + 1, 1, BUF ^ 1);
        PHASE_SYNC_IN();
        mma_quadrant(acc, 0, 0, fb0);
        PHASE_SYNC_OUT();
        // P2
        read_B(fb1, buf + OFF_B1);
        if (more2) stage_A(kt + 2, 0, BUF);
        PHASE_SYNC_IN();
        mma_quadrant(acc, 1, 0, fb1);
        PHASE_SYNC_OUT();
        // P3
        read_A(buf + OFF_A1);
        if (more2) stage_B(kt + 2, 0, BUF);
        PHASE_SYNC_IN();
        mma_quadrant(acc, 1, 1, fb1);
        PHASE_SYNC_OUT();
        // P4
        if (more2) {
            stage_B(kt + 2, 1, BUF);
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        PHASE_SYNC_IN();
        mma_quadrant(acc, 0, 1, fb0);
        PHASE_SYNC_OUT();
    };
#endif
    for (int kt = 0; kt < nk; kt += 2) {
        ktile(kt, 0);
        ktile(kt + 1, 1);
    }
#undef PHASE_SYNC_IN
#undef PHASE_SYNC_OUT
    if (wm == 0) __builtin_amdgcn_s_barrier();

    // ---- epilogue: through LDS, so that every global access is a whole contiguous tile row ----------------------------
    // Straight from the accumulators a store instruction would touch 16 rows x 64 B (half cache lines, 16 lines per
    // instruction): measured ~3 B/clk/CU, several times slower than the main loop for K = 768.  Instead the tile goes through
    // LDS in two passes of 128 rows (f32, row stride 1040 B = conflict-free ds_write_b128); each wave then owns 16 whole rows
    // per pass: one ds_read_b128 + one 1 KiB-contiguous residual/aux load + one contiguous store per row, bias held in registers.
    constexpr int ERS = 1040;   // epilogue LDS row stride in bytes: 256 f32 + 16 B pad
    const int lrow = lane & 15, lcol = (lane >> 4) << 2;
    const int ecol = n0 + lane * 4;
    f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
    if (bias) bv = *reinterpret_cast<const f32x4*>(bias + ecol);
#pragma unroll
    for (int mh = 0; mh < 2; ++mh) {
        if (mh) __builtin_amdgcn_s_barrier();   // pass 0's reads are done before pass 1 overwrites the staging area
#pragma unroll
        for (int nh = 0; nh < 2; ++nh)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
                    *reinterpret_cast<f32x4*>(smem + (wm * 64 + mi * 16 + lrow) * ERS + (nh * 128 + wn * 32 + ni * 16 + lcol) * 4) = acc[nh][ni][mh][mi];
        __syncthreads();
        const int r0 = wave * 16;
#pragma unroll EPI_UNROLL
        for (int rr = 0; rr < 16; ++rr) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(smem + (r0 + rr) * ERS + lane * 16);
            gemm_epilogue_store<T, TC, EPI, RES, SAVE_U>(v, m0 + mh * 128 + r0 + rr, ecol, C, ldc, bv, alpha, residual, ldr, aux, ldaux);
        }
    }
}

}  // namespace t256
