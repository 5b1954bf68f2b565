// 256x256-tile NT GEMM, 8 waves, phased schedule with LDS-DMA kept in flight across barriers (counted vmcnt).
//
// Same contract and epilogues as gemm.hip (C = epi(alpha * A . B^T + bias) + residual); selected by lpi_gemm_nt for
// shapes with enough 256x256 tiles.  Why a second kernel: a 128x128 tile needs 64 KB of L2->LDS traffic per CU per 1024
// MFMA cycles (56+ B/clk/CU), i.e. it is L2-bandwidth bound at about half the bf16 MFMA rate; a 256x256 tile halves the
// bytes per FLOP (128 FLOP/B) and, at one 512-thread workgroup per CU, leaves room for a deep staging pipeline.
//
// Geometry: BK = 128 bytes of K per tile row (bf16 64, f32 32).  LDS = 2 K-tile buffers x {A half 0, A half 1, B half 0,
// B half 1} x 16 KiB = 128 KiB.  Wave (wm in 0..1, wn in 0..3) owns rows {h*128 + wm*64 + 0..63 : h = 0,1} and columns
// {h*128 + wn*32 + 0..31 : h = 0,1} of the tile, so each of its four 64x32 output quadrants reads ONE A half and ONE B half.
// Per K-tile two phases of two quadrants (32 MFMA 16x16x32 per wave in bf16), each between two raw barriers:
//     X: read B(0), B(1), A(0)   compute (0,0), (1,0)   stage A-half1 of K-tile kt+1          ; s_waitcnt vmcnt(8)
//     Y: read A(1)               compute (1,1), (0,1)   stage A-half0, B-half0, B-half1 of kt+2 ; s_waitcnt vmcnt(8)
// (the B fragments stay in registers across X and Y).  A half tile is restaged only after the phase whose lgkmcnt(0)+barrier
// retired its last ds_read (WAR), and read only after the counted vmcnt that retires it plus a barrier every wave has passed
// (RAW).  vmcnt(8) leaves the four newest half tiles (2 LDS-DMA instructions each) in flight.  Raw s_barrier only — a
// __syncthreads() would drain the LDS-DMA queue (cdna_hip_programming.md, "Pipelining across barriers").  The first version had
// four phases of one quadrant (8 barriers per K-tile, three half tiles in flight): an empty loop of those barriers alone costs
// 0.64 us per K-tile against 0.87 us of MFMA work (profiles/r01_gemm_ablation.md).
// 1 (default): two phases of 32 MFMAs per K-tile; 0: the earlier four phases of 16 (kept for A/B: same results bit for bit,
// 2.9 % slower over the eight vision-layer GEMMs, 11 % on the qkv shape)
#ifndef LPI_TWO_PHASE
#define LPI_TWO_PHASE 1
#endif
#include "common.h"
#include "gemm_epilogue.h"

extern int g_lpi_tuning[8];

namespace {

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
__global__ __launch_bounds__(NTHR, 2) void gemm256_kernel(
    int M, int N, int K, const T* __restrict__ A, int lda, const T* __restrict__ B, int ldb,
    TC* __restrict__ C, int ldc, const float* __restrict__ bias, const float* __restrict__ residual, int ldr,
    T* __restrict__ aux, int ldaux, float alpha, int tiles_m, int tiles_n, int stagger, int group_m)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int EPC = Elem<T>::EPC;
    constexpr int BK = ROWB / (int)sizeof(T);

    // One workgroup per CU and identical tiles keep the CUs in lockstep: all stream HBM (epilogue) together, then all run the
    // matrix cores together.  Delaying every other first-round workgroup once makes the two halves of the chip alternate, so one
    // half's epilogue gets the whole HBM bandwidth while the other half computes (later workgroups inherit their CU's phase).
    if (stagger > 0 && blockIdx.x < 256 && ((blockIdx.x >> 3) & 1)) {
        for (int i = 0; i < stagger; ++i) __builtin_amdgcn_s_sleep(127);
    }
    const int nwg = tiles_m * tiles_n;
    int bid = blockIdx.x;
    {   // bijective XCD remap: blocks that share an XCD (bid % 8) get a contiguous run of tiles
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int GROUP_M = group_m;
    const int group = bid / (GROUP_M * tiles_n);
    const int first_m = group * GROUP_M;
    const int gsz = min(tiles_m - first_m, GROUP_M);
    const int in_group = bid - group * GROUP_M * tiles_n;
    const int tm = first_m + in_group % gsz;
    const int tn = in_group / gsz;
    const int m0 = tm * T256, n0 = tn * T256;

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

template <typename T, typename TC, int EPI, bool RES, bool SAVE_U>
int launch256_impl(int M, int N, int K, const void* A, int lda, const void* B, int ldb, void* C, int ldc, const float* bias,
              const float* residual, int ldr, void* aux, int ldaux, float alpha, hipStream_t s)
{
    const int tm = M / T256, tn = N / T256;
    auto kern = gemm256_kernel<T, TC, EPI, RES, SAVE_U>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    LPI_LAUNCH(kern, dim3(tm * tn), dim3(NTHR), LDS_BYTES, s, M, N, K, (const T*)A, lda, (const T*)B, ldb, (TC*)C, ldc, bias, residual,
               ldr, (T*)aux, ldaux, alpha, tm, tn, g_lpi_tuning[2], g_lpi_tuning[4] > 0 ? g_lpi_tuning[4] : 8);
    LPI_CHECK_LAST();
    return 0;
}

// run-time pointer presence -> compile-time epilogue flags (residual only with EPI_NONE, save-u only with QUICKGELU)
template <typename T, typename TC, int EPI>
int launch256(int M, int N, int K, const void* A, int lda, const void* B, int ldb, void* C, int ldc, const float* bias,
           const float* residual, int ldr, void* aux, int ldaux, float alpha, hipStream_t s)
{
    if constexpr (EPI == LPI_EPI_NONE) {
        if (residual) return launch256_impl<T, TC, EPI, true, false>(M, N, K, A, lda, B, ldb, C, ldc, bias, residual, ldr, aux, ldaux, alpha, s);
        return launch256_impl<T, TC, EPI, false, false>(M, N, K, A, lda, B, ldb, C, ldc, bias, residual, ldr, aux, ldaux, alpha, s);
    } else {
        if (residual) return LPI_ENOSYS;
        if constexpr (EPI == LPI_EPI_QUICKGELU) {
            if (aux) return launch256_impl<T, TC, EPI, false, true>(M, N, K, A, lda, B, ldb, C, ldc, bias, residual, ldr, aux, ldaux, alpha, s);
            return launch256_impl<T, TC, EPI, false, false>(M, N, K, A, lda, B, ldb, C, ldc, bias, residual, ldr, aux, ldaux, alpha, s);
        } else {
            if (!aux) return LPI_EINVAL;
            return launch256_impl<T, TC, EPI, false, false>(M, N, K, A, lda, B, ldb, C, ldc, bias, residual, ldr, aux, ldaux, alpha, s);
        }
    }
}

template <typename T, typename TC>
int dispatch256(int epi, int M, int N, int K, const void* A, int lda, const void* B, int ldb, void* C, int ldc, const float* bias,
                const float* residual, int ldr, void* aux, int ldaux, float alpha, hipStream_t s)
{
    switch (epi) {
    case LPI_EPI_NONE: return launch256<T, TC, LPI_EPI_NONE>(M, N, K, A, lda, B, ldb, C, ldc, bias, residual, ldr, aux, ldaux, alpha, s);
    case LPI_EPI_QUICKGELU: return launch256<T, TC, LPI_EPI_QUICKGELU>(M, N, K, A, lda, B, ldb, C, ldc, bias, residual, ldr, aux, ldaux, alpha, s);
    case LPI_EPI_DQUICKGELU: return launch256<T, TC, LPI_EPI_DQUICKGELU>(M, N, K, A, lda, B, ldb, C, ldc, bias, residual, ldr, aux, ldaux, alpha, s);
    }
    return LPI_EINVAL;
}

}  // namespace

// true if the 256x256 kernel can take this shape
bool lpi_gemm256_eligible(int dtype, int M, int N, int K) {
    const int bk = ROWB / (dtype == LPI_F32 ? 4 : 2);
    if (M % T256 || N % T256 || K % bk) return false;
    const int nk = K / bk;
    return nk >= 2 && (nk % 2) == 0;
}

int lpi_gemm256_launch(int dtype, int c_dtype, int M, int N, int K, const void* A, int lda, const void* B, int ldb, void* C, int ldc,
                       const float* bias, const float* residual, int ldr, int epilogue, void* aux, int ldaux, float alpha, hipStream_t s)
{
    if (dtype == LPI_F32 && c_dtype == LPI_F32)
        return dispatch256<float, float>(epilogue, M, N, K, A, lda, B, ldb, C, ldc, bias, residual, ldr, aux, ldaux, alpha, s);
    if (dtype == LPI_BF16 && c_dtype == LPI_BF16)
        return dispatch256<bf16_t, bf16_t>(epilogue, M, N, K, A, lda, B, ldb, C, ldc, bias, residual, ldr, aux, ldaux, alpha, s);
    if (dtype == LPI_BF16 && c_dtype == LPI_F32)
        return dispatch256<bf16_t, float>(epilogue, M, N, K, A, lda, B, ldb, C, ldc, bias, residual, ldr, aux, ldaux, alpha, s);
    if (dtype == LPI_BF16 && c_dtype == LPI_F16 && epilogue == LPI_EPI_NONE && residual)
        return launch256_impl<bf16_t, f16_t, LPI_EPI_NONE, true, false>(M, N, K, A, lda, B, ldb, C, ldc, bias, residual, ldr, aux, ldaux, alpha, s);
    return LPI_ENOSYS;
}
