// 256x256-tile NT GEMM, FOUR waves (2x2), each wave a 128x128 sub-tile: 256 accumulator registers per lane (one wave per SIMD has
// the whole 512-entry register file), 32 ds_read_b128 per 128 MFMAs per K-tile (0.25 per MFMA; the 8-wave kernel needs 0.375), and
// ONE workgroup barrier per K-tile.  EXPERIMENT, NOT BUILT INTO liblpi_hip.so (kept for the next round; results bit-identical to
// gemm256.hip on every epilogue).  Measured on the eight vision-layer GEMMs: 1 990 us vs 1 625 us for the 8-wave kernel
// (+7..16 % on light epilogues, +40 % on heavy ones).  Why it loses: with whole-K-tile double buffering the LDS-DMA window of a
// K-tile is exactly one K-tile (its buffer frees at mid K-tile and must be full by the next mid) while a CU needs ~1.07 us of
// transfer (64 KB at ~60 GB/s) plus ~0.6 us of latency; the 8-wave kernel's half-tile slots keep 3-4 half tiles in flight all the
// time.  The fix is a ring of k-step stages (64-byte rows, 4 x 32 KB); the 4-wave epilogue also needs more stores in flight.
// Lessons already paid for: pin the accumulators with "+a" inline-asm MFMAs (hipcc otherwise moves them through v_accvgpr_* around
// every MFMA); keep ONE loop body (a separately compiled tail re-maps all 256 AGPRs right behind asm MFMAs without the
// MFMA -> v_accvgpr_read wait states: wrong results); interleave every LDS-DMA / ds_read between MFMAs by hand.
//
// Software pipeline inside the single wave of a SIMD: the fragments of k-step s+1 are read from LDS while the 64 MFMAs of k-step s
// run (two fragment sets, 64 VGPRs each).  Per K-tile:
//     read F1 <- (buf, k-step 1)             | 64 MFMA on F0
//     lgkmcnt(0), vmcnt(0), s_barrier        : every wave has finished reading `buf`; K-tile kt+1 has landed in the other buffer
//     LDS-DMA K-tile kt+2 -> buf (16 instr)  | read F0 <- (buf^1, k-step 0) | 64 MFMA on F1
#include <type_traits>
#include "common.h"
#include "gemm_epilogue.h"

extern int g_lpi_tuning[8];

namespace {

constexpr int T256 = 256;
constexpr int ROWB = 128;
constexpr int HALF_BYTES = 128 * ROWB;       // 16 KiB
constexpr int BUF_BYTES = 4 * HALF_BYTES;    // A (256 rows) | B (256 rows)
constexpr int LDS_BYTES = 128 * 1040;        // >= 2 * BUF_BYTES (128 KiB); the epilogue stages 128 rows x 1040 B in it
constexpr int NTHR = 256;
constexpr int OFF_A = 0, OFF_B = 2 * HALF_BYTES;

template <typename T, typename TC, int EPI, bool RES, bool SAVE_U>
__global__ __launch_bounds__(NTHR, 1) void gemm256w4_kernel(
    int M, int N, int K, const T* __restrict__ A, int lda, const T* __restrict__ B, int ldb,
    TC* __restrict__ C, int ldc, const float* __restrict__ bias, const float* __restrict__ residual, int ldr,
    T* __restrict__ aux, int ldaux, float alpha, int tiles_m, int tiles_n, int group_m)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int EPC = Elem<T>::EPC;
    constexpr int BK = ROWB / (int)sizeof(T);

    const int nwg = tiles_m * tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int group = bid / (group_m * tiles_n);
    const int first_m = group * group_m;
    const int gsz = min(tiles_m - first_m, group_m);
    const int in_group = bid - group * group_m * tiles_n;
    const int tm = first_m + in_group % gsz;
    const int tn = in_group / gsz;
    const int m0 = tm * T256, n0 = tn * T256;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    // staging: a quarter-tile instruction = 256 threads x 16 B = 32 rows x 128 B; thread t, instr i -> row i*32 + t/8, swizzled chunk
    const int srow = tid >> 3;
    const int schunk = (tid & 7) ^ ((tid >> 4) & 7);
    const T* a_src = A + (size_t)(m0 + srow) * lda + schunk * EPC;
    const T* b_src = B + (size_t)(n0 + srow) * ldb + schunk * EPC;
    const size_t a_i = (size_t)32 * lda, b_i = (size_t)32 * ldb;
    const unsigned lds_w = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + wave * 1024);
    auto glds16 = [&](const T* src, unsigned lds_addr) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(lds_addr) : "memory");
    };
    auto stage = [&](int kt, int buf) {   // 16 LDS-DMA instructions: A rows 0..255, B rows 0..255 of K-tile kt
        const unsigned base = lds_w + buf * BUF_BYTES;
        const T* ap = a_src + (size_t)kt * BK;
        const T* bp = b_src + (size_t)kt * BK;
#pragma unroll
        for (int i = 0; i < 8; ++i) glds16(ap + i * a_i, base + OFF_A + i * 4096);
#pragma unroll
        for (int i = 0; i < 8; ++i) glds16(bp + i * b_i, base + OFF_B + i * 4096);
    };

    const int frow = lane & 15, fg = lane >> 4, fsw = frow >> 1;
    // LDS byte addresses of this lane's fragment rows: [buffer][k-step]; sub-tile i adds the immediate i * 2048
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    unsigned aaddr[2][2], baddr[2][2];
#pragma unroll
    for (int bf = 0; bf < 2; ++bf)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const unsigned fo = frow * ROWB + (((ks << 2) | fg) ^ fsw) * 16;
            aaddr[bf][ks] = lds0 + bf * BUF_BYTES + OFF_A + (wm * 128) * ROWB + fo;
            baddr[bf][ks] = lds0 + bf * BUF_BYTES + OFF_B + (wn * 128) * ROWB + fo;
        }

    f32x4 acc[8][8];   // [ni][mi], pinned to the 256 AGPRs by the "+a" constraints below
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Hand-scheduled main loop: with ONE wave per SIMD nothing else covers an instruction this wave issues outside the MFMA stream,
    // so every LDS-DMA and ds_read is placed BETWEEN MFMAs (the matrix pipe executes 16 cycles per MFMA while the wave issues the
    // next few instructions).  Everything in the loop is volatile asm in program order: hipcc neither reorders it nor adds waits.
    bf16x8 fa0[8], fb0[8], fa1[8], fb1[8];
#define MFMA_(ACC, B_, A_) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(ACC) : "v"(B_), "v"(A_))
#define DSRD_(DST, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(OFF))
    auto dma = [&](const T* ap, const T* bp, unsigned base, int j) {      // instruction j of the 16 that stage one K-tile
        if (j < 8) glds16(ap + j * a_i, base + OFF_A + j * 4096);
        else glds16(bp + (j - 8) * b_i, base + OFF_B + (j - 8) * 4096);
    };

    const int nk = K / BK;   // even, >= 2
    stage(0, 0);
    stage(1, 1);
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        DSRD_(fa0[i], aaddr[0][0], i * 2048);
        DSRD_(fb0[i], baddr[0][0], i * 2048);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    // BUF is compile-time (LDS addresses fold); "K-tile kt+1 / kt+2 exists" are wave-uniform run-time flags: ONE loop body, so the
    // accumulators keep one AGPR assignment (a separately compiled tail made hipcc shuffle all 256 through v_accvgpr_mov at the
    // loop exit, right behind inline-asm MFMAs whose read-after-write wait states it cannot know about).
    auto ktile = [&](int kt, auto buf_c) {
        constexpr int BUF = decltype(buf_c)::value;
        const bool more1 = kt + 1 < nk, more2 = kt + 2 < nk;
        // half A: 64 MFMAs on F0 (k-step 0 of K-tile kt); the 16 reads of F1 (k-step 1) go one after every 2nd MFMA of the first 32
#pragma unroll
        for (int ni = 0; ni < 8; ++ni)
#pragma unroll
            for (int mi = 0; mi < 8; ++mi) {
                MFMA_(acc[ni][mi], fb0[ni], fa0[mi]);
                const int q = ni * 8 + mi;
                if (q < 32 && (q & 1)) {
                    const int r = q >> 1;      // 0..15: fa1[0..7] first (every MFMA row needs them), then fb1[0..7]
                    if (r < 8) DSRD_(fa1[r], aaddr[BUF][1], r * 2048);
                    else DSRD_(fb1[r - 8], baddr[BUF][1], (r - 8) * 2048);
                }
            }
        // every wave has read all of `buf` once F1 is in registers; K-tile kt+1 (LDS-DMA issued during half B of kt-1) has landed
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        // half B: 64 MFMAs on F1; K-tile kt+2 -> buf (16 LDS-DMA) and F0 <- k-step 0 of K-tile kt+1 (16 reads) between the first 32
        const unsigned dbase = lds_w + BUF * BUF_BYTES;
        const T* ap = a_src + (size_t)(kt + 2) * BK;
        const T* bp = b_src + (size_t)(kt + 2) * BK;
#pragma unroll
        for (int ni = 0; ni < 8; ++ni)
#pragma unroll
            for (int mi = 0; mi < 8; ++mi) {
                MFMA_(acc[ni][mi], fb1[ni], fa1[mi]);
                const int q = ni * 8 + mi;
                if (q < 32) {
                    const int r = q >> 1;
                    if (q & 1) {
                        if (more1) {
                            if (r < 8) DSRD_(fa0[r], aaddr[BUF ^ 1][0], r * 2048);
                            else DSRD_(fb0[r - 8], baddr[BUF ^ 1][0], (r - 8) * 2048);
                        }
                    } else if (more2) {
                        dma(ap, bp, dbase, r);
                    }
                }
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    for (int kt = 0; kt < nk; kt += 2) {
        ktile(kt, I0{});
        ktile(kt + 1, I1{});
    }
#undef MFMA_
#undef DSRD_
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // inline-asm MFMAs: the compiler does not pad MFMA -> v_accvgpr_read for us
    __syncthreads();

    // ---- epilogue through LDS: two passes of 128 rows (the wm = pass waves hold them), every global access a whole tile row ----
    constexpr int ERS = 1040;
    const int lrow = lane & 15, lcol = (lane >> 4) << 2;
    const int ecol = n0 + lane * 4;
    f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
    if (bias) bv = *reinterpret_cast<const f32x4*>(bias + ecol);
#pragma unroll
    for (int ph = 0; ph < 2; ++ph) {
        if (ph) __syncthreads();
        if (wm == ph) {
#pragma unroll
            for (int ni = 0; ni < 8; ++ni)
#pragma unroll
                for (int mi = 0; mi < 8; ++mi)
                    *reinterpret_cast<f32x4*>(smem + (mi * 16 + lrow) * ERS + (wn * 128 + ni * 16 + lcol) * 4) = acc[ni][mi];
        }
        __syncthreads();
        const int r0 = wave * 32;
#pragma unroll 4
        for (int rr = 0; rr < 32; ++rr) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(smem + (r0 + rr) * ERS + lane * 16);
            gemm_epilogue_store<T, TC, EPI, RES, SAVE_U>(v, m0 + ph * 128 + r0 + rr, ecol, C, ldc, bv, alpha, residual, ldr, aux, ldaux);
        }
    }
}

template <typename T, typename TC, int EPI, bool RES, bool SAVE_U>
int launch_impl(int M, int N, int K, const void* A, int lda, const void* B, int ldb, void* C, int ldc, const float* bias,
                const float* residual, int ldr, void* aux, int ldaux, float alpha, hipStream_t s)
{
    const int tm = M / T256, tn = N / T256;
    auto kern = gemm256w4_kernel<T, TC, EPI, RES, SAVE_U>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    LPI_LAUNCH(kern, dim3(tm * tn), dim3(NTHR), LDS_BYTES, s, M, N, K, (const T*)A, lda, (const T*)B, ldb, (TC*)C, ldc, bias, residual,
               ldr, (T*)aux, ldaux, alpha, tm, tn, g_lpi_tuning[4] > 0 ? g_lpi_tuning[4] : 8);
    LPI_CHECK_LAST();
    return 0;
}

}  // namespace

// bf16 only while experimental: plain / +fp16 residual / QuickGELU(+u) / gelu' epilogues
int lpi_gemm256w4_launch(int dtype, int c_dtype, int M, int N, int K, const void* A, int lda, const void* B, int ldb, void* C, int ldc,
                         const float* bias, const float* residual, int ldr, int epilogue, void* aux, int ldaux, float alpha, hipStream_t s)
{
    if (dtype != LPI_BF16) return LPI_ENOSYS;
#define W4(TC, EPI, RES, SU) return launch_impl<bf16_t, TC, EPI, RES, SU>(M, N, K, A, lda, B, ldb, C, ldc, bias, residual, ldr, aux, ldaux, alpha, s)
    if (c_dtype == LPI_BF16 && epilogue == LPI_EPI_NONE && !residual) W4(bf16_t, LPI_EPI_NONE, false, false);
    if (c_dtype == LPI_F16 && epilogue == LPI_EPI_NONE && residual) W4(f16_t, LPI_EPI_NONE, true, false);
    if (c_dtype == LPI_BF16 && epilogue == LPI_EPI_QUICKGELU && aux && !residual) W4(bf16_t, LPI_EPI_QUICKGELU, false, true);
    if (c_dtype == LPI_BF16 && epilogue == LPI_EPI_DQUICKGELU && aux && !residual) W4(bf16_t, LPI_EPI_DQUICKGELU, false, false);
#undef W4
    return LPI_ENOSYS;
}
