// TWO workgroups per CU: 256x128-tile NT GEMM with 4 waves of 128x64 (the wave tile, fragment reads and MFMA order of the 256x256
// kernel, gemm256_tile.h) and an 80 KiB LDS ring, so that two independent workgroups are resident on a CU and one's epilogue
// (HBM stores, QuickGELU / gelu' vector work) runs under the other's main loop.
//
// Why: the 256x256 kernels own a CU (128 accumulator VGPRs x 8 waves, 128 KiB of LDS), so a tile's epilogue and main loop are serial on
// it, and the epilogue's stores share the wave's in-order vmcnt queue with the LDS-DMA of the next tile (gemm256p.hip) — for the
// store-heavy GEMMs (c_fc + QuickGELU saving u, d c_proj * gelu'(u): 750 MB per launch) HBM time and MFMA time ADD UP
// (profiles/r01_gemm_ablation.md).  The earlier two-tiles-per-CU attempt (gemm256x128.hip: 8 waves of 64x64) lost in the main loop
// because a 64x64 wave tile needs 0.5 ds_read_b128 per MFMA; here the wave tile stays 128x64 (0.375), only the workgroup shrinks to 4
// waves, and the price is 1.5x the L2->LDS bytes per FLOP (48 KiB per 256x128x64 K-tile), which the LDS-DMA path has room for
// (tools/probe/dma_rows_probe.hip: 0.76 us per 64 KiB at full chip against 1.35 us of main loop).
//
// LDS: 5 units of 16 KiB (128 rows x 128 B, chunk XOR-swizzled by (row >> 1) & 7 as everywhere): 2 for B (the whole 128-row operand of
// a K-tile: rows 0-63 = column half 0, rows 64-127 = column half 1), 3 for the A halves (A0 = tile rows 0-127, A1 = rows 128-255).
// K-tile k reads B from unit b[k & 1], A0 from a[2k % 3], A1 from a[(2k+1) % 3]; one s_barrier per phase:
//   X_k: read B_k, A0_k fragments | LDS-DMA A0_{k+1} -> a[(2k+2) % 3] (the unit A1_{k-1} left)  | vmcnt(8) | barrier | 32 MFMAs
//   Y_k: read A1_k fragments      | LDS-DMA A1_{k+1} -> a[2k % 3], B_{k+2} -> b[k & 1]          | vmcnt(8) | barrier | 32 MFMAs
// (every unit is refilled one phase after its last read; every fill is issued >= one K-tile before its first read).
// The epilogue stages 64 rows x 512 B per pass through the first 32 KiB and stores whole rows; same arithmetic, same bits as the other
// GEMM kernels (per-element K order and the f32 epilogue are the same).
#include <algorithm>
#include "common.h"
#include "gemm_epilogue.h"

extern int g_lpi_tuning[16];

namespace {

constexpr int ROWB = 128;              // bytes per staged row
constexpr int UNIT = 128 * ROWB;       // 16 KiB
constexpr int NTHR = 256;
constexpr int LDS_DUO = 5 * UNIT;      // 80 KiB: two workgroups fill a CU's 160 KiB

template <typename T, typename TC, int EPI, bool RES, bool SAVE_U>
__global__ __launch_bounds__(NTHR, 2) void gemm_duo_kernel(
    int K, const T* __restrict__ A, int lda, const T* __restrict__ B, int ldb, TC* __restrict__ C, int ldc, const float* __restrict__ bias,
    const float* __restrict__ residual, int ldr, typename AuxT<T>::type* __restrict__ aux, int ldaux, float alpha, int tiles_m, int tiles_n,
    int group_m)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int EPC = Elem<T>::EPC;
    constexpr int BK = ROWB / (int)sizeof(T);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    // tile of this workgroup: the XCD-aware order of gemm256_kernel (ids that share blockIdx & 7 share an XCD), 128-column tiles
    int m0, n0;
    {
        const int nwg = tiles_m * tiles_n, vb = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = vb & 7, idx = vb >> 3;
        const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        const int group = t / (group_m * tiles_n);
        const int first_m = group * group_m;
        const int gsz = min(tiles_m - first_m, group_m);
        const int in_group = t - group * group_m * tiles_n;
        m0 = __builtin_amdgcn_readfirstlane((first_m + in_group % gsz) * 256);
        n0 = __builtin_amdgcn_readfirstlane((in_group / gsz) * 128);
    }

    // ---- staging: a unit = 128 rows x 128 B = 4 LDS-DMA instructions of 256 lanes x 16 B (SGPR base + VGPR offset, gemm256p.hip)
    const int srow = tid >> 3;                                  // 0..31
    const int schunk = (tid & 7) ^ ((tid >> 4) & 7);            // logical chunk of physical chunk tid & 7 in row srow (+ 32 i)
    const unsigned a_off = (unsigned)(((size_t)srow * lda + schunk * EPC) * sizeof(T));
    const unsigned b_off = (unsigned)(((size_t)srow * ldb + schunk * EPC) * sizeof(T));
    const unsigned lds_w = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + wave * 1024);
    auto glds16 = [&](const T* sbase, unsigned voff, unsigned lds_addr) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
    };
    auto stage_A = [&](int kt, int h, int unit_off) {
        const T* sb = A + (size_t)(m0 + h * 128) * lda + (size_t)kt * BK;
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16(sb + (size_t)(32 * i) * lda, a_off, lds_w + unit_off + i * 4096);
    };
    auto stage_B = [&](int kt, int unit_off) {
        const T* sb = B + (size_t)n0 * ldb + (size_t)kt * BK;
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16(sb + (size_t)(32 * i) * ldb, b_off, lds_w + unit_off + i * 4096);
    };
    auto off_b = [](int k) { return (k & 1) * UNIT; };
    auto off_a = [](int j) { return (2 + j % 3) * UNIT; };      // A0_k: j = 2k, A1_k: j = 2k + 1

    // ---- fragment offsets within a unit
    const int frow = lane & 15, fg = lane >> 4, fsw = frow >> 1;
    int foff[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) foff[ks] = frow * ROWB + (((ks << 2) | fg) ^ fsw) * 16;
    const int a_base = (wm * 64) * ROWB;
    const int b_base = (wn * 32) * ROWB;

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
    auto read_A = [&](const char* unit) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) fa[mi][ks].u = *reinterpret_cast<const uint4*>(unit + a_base + mi * 16 * ROWB + foff[ks]);
    };
    auto read_B = [&](Chunk (&fb)[2][2], const char* half) {      // half = unit + nh * 64 rows
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) fb[ni][ks].u = *reinterpret_cast<const uint4*>(half + b_base + ni * 16 * ROWB + foff[ks]);
    };
    auto mma_quadrant = [&](int nh, int mh, const Chunk (&fb)[2][2]) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) mma_chunk<T>(acc[nh][ni][mh][mi], fb[ni][ks], fa[mi][ks]);
        __builtin_amdgcn_s_setprio(0);
    };
#define PHASE_SYNC()                                      \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    \
    __builtin_amdgcn_s_barrier();                         \
    __builtin_amdgcn_sched_barrier(0)

    const int nk = K / BK;      // >= 2 (checked on the host)
    // prologue: K-tile 0 (B_0, A0_0, A1_0) and B_1; A0_1 follows in phase X_0
    stage_B(0, off_b(0));
    stage_A(0, 0, off_a(0));
    stage_A(0, 1, off_a(1));
    stage_B(1, off_b(1));
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");          // B_0 and A0_0 landed
    __builtin_amdgcn_s_barrier();

    for (int k = 0; k < nk; ++k) {
        const char* ub = smem + off_b(k);
        const bool more1 = k + 1 < nk, more2 = k + 2 < nk;
        // ---- X_k
        read_B(fb0, ub);
        read_B(fb1, ub + 64 * ROWB);
        __builtin_amdgcn_sched_barrier(0);
        read_A(smem + off_a(2 * k));
        if (more1) {
            stage_A(k + 1, 0, off_a(2 * k + 2));
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      // A1_k landed (younger: B_{k+1}, A0_{k+1})
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        PHASE_SYNC();
        mma_quadrant(0, 0, fb0);
        mma_quadrant(1, 0, fb1);
        __builtin_amdgcn_sched_barrier(0);
        // ---- Y_k
        read_A(smem + off_a(2 * k + 1));
        if (more1) {
            stage_A(k + 1, 1, off_a(2 * k + 3));
            if (more2) {
                stage_B(k + 2, off_b(k));
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // B_{k+1}, A0_{k+1} landed (younger: A1_{k+1}, B_{k+2})
            } else {
                asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            }
        }
        PHASE_SYNC();
        mma_quadrant(1, 1, fb1);
        mma_quadrant(0, 1, fb0);
        __builtin_amdgcn_sched_barrier(0);
    }
#undef PHASE_SYNC

    // ---- epilogue: four passes of 64 rows x 512 B through the first 32 KiB (pass p: mh = p >> 1, mi in {2 (p & 1), 2 (p & 1) + 1}).
    // Staging row s = wm * 32 + (mi & 1) * 16 + lrow holds tile row mh * 128 + wm * 64 + mi * 16 + lrow as 32 16-byte chunks (chunk c =
    // output columns 4c .. 4c+3) at physical chunk c ^ (s & 7); a wave instruction reads two whole rows.
    const int lrow = lane & 15, lslot = lane >> 4;
    const int c4 = lane & 31;
    const int ecol = n0 + c4 * 4;
    f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
    if (bias) bv = *reinterpret_cast<const f32x4*>(bias + ecol);
    char* const stg = smem;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                  // every wave is past its last fragment reads
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int mh = p >> 1;
        if (p) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
#pragma unroll
        for (int mi2 = 0; mi2 < 2; ++mi2) {
            const int s_row = wm * 32 + mi2 * 16 + lrow;
#pragma unroll
            for (int nh = 0; nh < 2; ++nh)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    const int chunk = nh * 16 + wn * 8 + ni * 4 + lslot;
                    *reinterpret_cast<f32x4*>(stg + s_row * 512 + ((chunk ^ (s_row & 7)) << 4)) = acc[nh][ni][mh][(p & 1) * 2 + mi2];
                }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
            const int s_row = wave * 16 + rr * 2 + (lane >> 5);
            const int trow = mh * 128 + (s_row >> 5) * 64 + ((p & 1) * 2 + ((s_row >> 4) & 1)) * 16 + (s_row & 15);
            const f32x4 v = *reinterpret_cast<const f32x4*>(stg + s_row * 512 + ((c4 ^ (s_row & 7)) << 4));
            gemm_epilogue_store<T, TC, EPI, RES, SAVE_U>(v, m0 + trow, ecol, C, ldc, bv, alpha, residual, ldr, aux, ldaux);
        }
    }
}

template <typename T, typename TC, int EPI, bool RES, bool SAVE_U>
int launch_impl(int M, int N, int K, const void* A, int lda, const void* B, int ldb, void* C, int ldc, const float* bias,
                const float* residual, int ldr, void* aux, int ldaux, float alpha, hipStream_t s)
{
    const int tm = M / 256, tn = N / 128;
    auto kern = gemm_duo_kernel<T, TC, EPI, RES, SAVE_U>;
    static LdsOnce once;
    if (int e = lpi_ensure_lds(once, (const void*)kern, LDS_DUO)) return e;
    lpi_note_gemm_kernel(LPI_GEMM_K_DUO);
    LPI_LAUNCH(kern, dim3(tm * tn), dim3(NTHR), LDS_DUO, s, K, (const T*)A, lda, (const T*)B, ldb, (TC*)C, ldc, bias, residual, ldr,
               (typename AuxT<T>::type*)aux, ldaux, alpha, tm, tn, g_lpi_tuning[4] > 0 ? g_lpi_tuning[4] : 8);
    LPI_CHECK_LAST();
    return 0;
}

template <typename T, typename TC>
int dispatch(int epi, int M, int N, int K, const void* A, int lda, const void* B, int ldb, void* C, int ldc, const float* bias,
             const float* residual, int ldr, void* aux, int ldaux, float alpha, hipStream_t s)
{
#define GO(EPI, RES, SU) return launch_impl<T, TC, EPI, RES, SU>(M, N, K, A, lda, B, ldb, C, ldc, bias, residual, ldr, aux, ldaux, alpha, s)
    switch (epi) {
    case LPI_EPI_NONE:
        if (residual) GO(LPI_EPI_NONE, true, false);
        GO(LPI_EPI_NONE, false, false);
    case LPI_EPI_QUICKGELU:
        if (residual) return LPI_ENOSYS;
        if (aux) GO(LPI_EPI_QUICKGELU, false, true);
        GO(LPI_EPI_QUICKGELU, false, false);
    case LPI_EPI_DQUICKGELU:
        if (residual) return LPI_ENOSYS;
        if (!aux) return LPI_EINVAL;
        GO(LPI_EPI_DQUICKGELU, false, false);
    }
#undef GO
    return LPI_EINVAL;
}

}  // namespace

// shapes the two-workgroups-per-CU kernel takes: whole 256x128 tiles, whole and at least two 128-byte K-tiles, 2-byte operands
bool lpi_gemm_duo_eligible(int dtype, int M, int N, int K) {
    if (dtype == LPI_F32) return false;
    return M % 256 == 0 && N % 128 == 0 && K % 64 == 0 && K / 64 >= 2;
}

int lpi_gemm_duo_launch(int dtype, int c_dtype, int M, int N, int K, const void* A, int lda, const void* B, int ldb, void* C, int ldc,
                        const float* bias, const float* residual, int ldr, int epilogue, void* aux, int ldaux, float alpha, hipStream_t s)
{
    if (dtype == LPI_BF16 && c_dtype == LPI_BF16) return dispatch<bf16_t, bf16_t>(epilogue, M, N, K, A, lda, B, ldb, C, ldc, bias, residual, ldr, aux, ldaux, alpha, s);
    if (dtype == LPI_BF16 && c_dtype == LPI_F32) return dispatch<bf16_t, float>(epilogue, M, N, K, A, lda, B, ldb, C, ldc, bias, residual, ldr, aux, ldaux, alpha, s);
    if (dtype == LPI_BF16 && c_dtype == LPI_F16 && epilogue == LPI_EPI_NONE && residual)
        return launch_impl<bf16_t, f16_t, LPI_EPI_NONE, true, false>(M, N, K, A, lda, B, ldb, C, ldc, bias, residual, ldr, aux, ldaux, alpha, s);
    if (dtype == LPI_F16 && c_dtype == LPI_F16) return dispatch<f16_t, f16_t>(epilogue, M, N, K, A, lda, B, ldb, C, ldc, bias, residual, ldr, aux, ldaux, alpha, s);
    if (dtype == LPI_F16 && c_dtype == LPI_F32) return dispatch<f16_t, float>(epilogue, M, N, K, A, lda, B, ldb, C, ldc, bias, residual, ldr, aux, ldaux, alpha, s);
    return LPI_ENOSYS;
}
