// NT GEMM on the gfx950 matrix cores:  C[M,N] = epi(alpha * A[M,K] . B[N,K]^T + bias) + residual
//
// replaces: nn.Linear / nn.MultiheadAttention projections / conv1-as-matmul / x @ proj of the reference
// (retrieval/models/clip/model.py:172-177,185,215,228,257; prompt_learner.py:61) and all their dgrads
// (the backbone is frozen, so backward is dgrad only and B is then the pre-transposed weight).
//
// Design (MI355X first):
//  * block tile 128x128, 4 waves as 2x2, each wave 64x64 = 4x4 MFMA 16x16 tiles (64 accumulator VGPRs);
//  * both operands are K-contiguous, so one LDS tile row = 128 bytes = 8 x 16-byte chunks for either element
//    type (bf16: BK=64, f32: BK=32); the same staging/fragment code serves both, only mma_chunk differs;
//  * global -> LDS by `global_load_lds_dwordx4` (16 B/lane, no VGPR round trip), double buffered, one barrier
//    per K tile; the LDS image is lane-linear so the bank swizzle chunk ^= (row>>1)&7 is applied to the
//    per-lane SOURCE address and again on the ds_read_b128 (conflict-free for the 16x16 fragment pattern);
//  * operands are swapped in the MFMA (weights = "A", activations = "B") so that each lane ends up holding 4
//    CONSECUTIVE columns of one output row: the epilogue (bias, QuickGELU, gelu', residual) runs on float4s and
//    stores 16 B (f32) / 8 B (bf16) per lane;
//  * blockIdx is remapped so that the blocks sharing one XCD's L2 walk neighbouring M tiles of the same N panel.
#include "common.h"
#include "gemm_epilogue.h"

namespace {

constexpr int BM = 128, BN = 128;
constexpr int ROW_BYTES = 128;              // one staged tile row (BK elements)
constexpr int TILE_BYTES = BM * ROW_BYTES;  // 16 KiB per operand per stage
constexpr int STAGE_BYTES = 2 * TILE_BYTES;
constexpr int NTHREADS = 256;

template <typename T, typename TC, int EPI, bool RES, bool SAVE_U>
__device__ __forceinline__ void gemm_nt_tile(
    int bx, char* smem, int M, int N, int K, const T* __restrict__ A, int lda, const T* __restrict__ B, int ldb,
    TC* __restrict__ C, int ldc, const float* __restrict__ bias, const float* __restrict__ residual, int ldr,
    typename AuxT<T>::type* __restrict__ aux, int ldaux, float alpha, int tiles_m, int tiles_n)
{
    constexpr int EPC = Elem<T>::EPC;
    constexpr int BK = ROW_BYTES / (int)sizeof(T);

    // XCD-aware tile order: consecutive block ids round-robin over the 8 XCDs; give each XCD a contiguous
    // run of the (n-panel major, m minor) tile list so the B panel and neighbouring A tiles stay in its L2.
    const int nwg = tiles_m * tiles_n;
    int bid = bx;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    // group GROUP_M m-tiles per n-tile sweep so one XCD reuses a B panel against several A tiles
    constexpr int GROUP_M = 8;
    const int group = bid / (GROUP_M * tiles_n);
    const int first_m = group * GROUP_M;
    const int gsz = min(tiles_m - first_m, GROUP_M);
    const int in_group = bid - group * GROUP_M * tiles_n;
    const int tm = first_m + in_group % gsz;
    const int tn = in_group / gsz;
    const int m0 = tm * BM, n0 = tn * BN;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;   // wave's 64x64 quadrant

    // ---- staging addresses: thread t, instruction i writes LDS byte i*4096 + t*16 of the tile ----------
    // row = i*32 + t/8, physical chunk = t%8, logical chunk = phys ^ ((row>>1)&7) = phys ^ (4*(wave&1) + lane/16)
    const int srow = tid >> 3;
    const int schunk = (tid & 7) ^ (((wave & 1) << 2) | (lane >> 4));
    const T* a_src = A + (size_t)(m0 + srow) * lda + schunk * EPC;
    const T* b_src = B + (size_t)(n0 + srow) * ldb + schunk * EPC;
    const size_t a_step = (size_t)32 * lda, b_step = (size_t)32 * ldb;

    auto stage = [&](int kt, int buf) {
        char* base = smem + buf * STAGE_BYTES + wave * 1024;
        const T* ap = a_src + (size_t)kt * BK;
        const T* bp = b_src + (size_t)kt * BK;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ap + i * a_step),
                                             (__attribute__((address_space(3))) void*)(base + i * 4096), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(bp + i * b_step),
                                             (__attribute__((address_space(3))) void*)(base + TILE_BYTES + i * 4096), 16, 0, 0);
        }
    };

    // ---- fragment read offsets: lane reads row (l&15) of a 16-row sub tile, logical chunk 4*ks + (l>>4) ----
    const int frow = lane & 15;
    const int fsw = frow >> 1;                 // (row>>1)&7, sub-tile bases are multiples of 16
    const int fg = lane >> 4;
    int foff[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) foff[ks] = frow * ROW_BYTES + (((ks << 2) | fg) ^ fsw) * 16;
    const int a_frag_base = (wm * 64) * ROW_BYTES;               // activations: rows of the A tile
    const int b_frag_base = TILE_BYTES + (wn * 64) * ROW_BYTES;  // weights: rows of the B tile

    f32x4 acc[4][4];  // [n sub tile][m sub tile]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = K / BK;
    stage(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's LDS-DMA of tile kt has landed
        __syncthreads();                                  // ... and everyone's; buffer (kt+1)&1 is free again
        if (kt + 1 < nk) stage(kt + 1, (kt + 1) & 1);
        const char* buf = smem + (kt & 1) * STAGE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            Chunk fa[4], fb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                fa[i].u = *reinterpret_cast<const uint4*>(buf + a_frag_base + i * 16 * ROW_BYTES + foff[ks]);
                fb[i].u = *reinterpret_cast<const uint4*>(buf + b_frag_base + i * 16 * ROW_BYTES + foff[ks]);
            }
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) mma_chunk<T>(acc[ni][mi], fb[ni], fa[mi]);
        }
    }

    // ---- epilogue: lane holds C[m = .. + (l&15)][n = .. + 4*(l>>4) + 0..3] ------------------------------
    const int row_base = m0 + wm * 64 + (lane & 15);
    const int col_base = n0 + wn * 64 + ((lane >> 4) << 2);
    f32x4 bvs[4];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) bvs[ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (bias) {   // ONE branch for all bias loads
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) bvs[ni] = *reinterpret_cast<const f32x4*>(bias + col_base + ni * 16);
    }
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
        const int col = col_base + ni * 16;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
            gemm_epilogue_store<T, TC, EPI, RES, SAVE_U>(acc[ni][mi], row_base + mi * 16, col, C, ldc, bvs[ni], alpha, residual, ldr, aux, ldaux);
    }
}

template <typename T, typename TC, int EPI, bool RES, bool SAVE_U>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_nt_kernel(
    int M, int N, int K, const T* __restrict__ A, int lda, const T* __restrict__ B, int ldb,
    TC* __restrict__ C, int ldc, const float* __restrict__ bias, const float* __restrict__ residual, int ldr,
    typename AuxT<T>::type* __restrict__ aux, int ldaux, float alpha, int tiles_m, int tiles_n)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    gemm_nt_tile<T, TC, EPI, RES, SAVE_U>(blockIdx.x, smem, M, N, K, A, lda, B, ldb, C, ldc, bias, residual, ldr, aux, ldaux, alpha, tiles_m, tiles_n);
}

template <typename T, typename TC, int EPI, bool RES, bool SAVE_U>
int launch_impl(int M, int N, int K, const void* A, int lda, const void* B, int ldb, void* C, int ldc, const float* bias,
           const float* residual, int ldr, void* aux, int ldaux, float alpha, hipStream_t s)
{
    const int tm = M / BM, tn = N / BN;
    auto kern = gemm_nt_kernel<T, TC, EPI, RES, SAVE_U>;
    static LdsOnce once;
    if (int e = lpi_ensure_lds(once, (const void*)kern, 2 * STAGE_BYTES)) return e;
    lpi_note_gemm_kernel(LPI_GEMM_K_128);
    LPI_LAUNCH(kern, dim3(tm * tn), dim3(NTHREADS), 2 * STAGE_BYTES, s, M, N, K, (const T*)A, lda, (const T*)B, ldb,
                       (TC*)C, ldc, bias, residual, ldr, (typename AuxT<T>::type*)aux, ldaux, alpha, tm, tn);
    LPI_CHECK_LAST();
    return 0;
}

// run-time pointer presence -> compile-time epilogue flags (residual only with EPI_NONE, save-u only with QUICKGELU)
template <typename T, typename TC, int EPI>
int launch(int M, int N, int K, const void* A, int lda, const void* B, int ldb, void* C, int ldc, const float* bias,
           const float* residual, int ldr, void* aux, int ldaux, float alpha, hipStream_t s)
{
    if constexpr (EPI == LPI_EPI_NONE) {
        if (residual) return launch_impl<T, TC, EPI, true, false>(M, N, K, A, lda, B, ldb, C, ldc, bias, residual, ldr, aux, ldaux, alpha, s);
        return launch_impl<T, TC, EPI, false, false>(M, N, K, A, lda, B, ldb, C, ldc, bias, residual, ldr, aux, ldaux, alpha, s);
    } else {
        if (residual) return LPI_ENOSYS;
        if constexpr (EPI == LPI_EPI_QUICKGELU) {
            if (aux) return launch_impl<T, TC, EPI, false, true>(M, N, K, A, lda, B, ldb, C, ldc, bias, residual, ldr, aux, ldaux, alpha, s);
            return launch_impl<T, TC, EPI, false, false>(M, N, K, A, lda, B, ldb, C, ldc, bias, residual, ldr, aux, ldaux, alpha, s);
        } else {
            if (!aux) return LPI_EINVAL;
            return launch_impl<T, TC, EPI, false, false>(M, N, K, A, lda, B, ldb, C, ldc, bias, residual, ldr, aux, ldaux, alpha, s);
        }
    }
}

template <typename T, typename TC>
int dispatch_epi(int epi, int M, int N, int K, const void* A, int lda, const void* B, int ldb, void* C, int ldc,
                 const float* bias, const float* residual, int ldr, void* aux, int ldaux, float alpha, hipStream_t s)
{
    switch (epi) {
    case LPI_EPI_NONE: return launch<T, TC, LPI_EPI_NONE>(M, N, K, A, lda, B, ldb, C, ldc, bias, residual, ldr, aux, ldaux, alpha, s);
    case LPI_EPI_QUICKGELU: return launch<T, TC, LPI_EPI_QUICKGELU>(M, N, K, A, lda, B, ldb, C, ldc, bias, residual, ldr, aux, ldaux, alpha, s);
    case LPI_EPI_DQUICKGELU:
        if (!aux) return LPI_EINVAL;
        return launch<T, TC, LPI_EPI_DQUICKGELU>(M, N, K, A, lda, B, ldb, C, ldc, bias, residual, ldr, aux, ldaux, alpha, s);
    }
    return LPI_EINVAL;
}

}  // namespace

bool lpi_gemm256_eligible(int dtype, int M, int N, int K);
int lpi_gemm256p_launch(int dtype, int c_dtype, int M, int N, int K, const void* A, int lda, const void* B, int ldb, void* C, int ldc,
                        const float* bias, const float* residual, int ldr, int epilogue, void* aux, int ldaux, float alpha, hipStream_t s);
int lpi_gemm256_launch(int dtype, int c_dtype, int M, int N, int K, const void* A, int lda, const void* B, int ldb, void* C, int ldc,
                       const float* bias, const float* residual, int ldr, int epilogue, void* aux, int ldaux, float alpha, hipStream_t s);
bool lpi_gemm256x128_eligible(int dtype, int M, int N, int K);
int lpi_gemm256x128_launch(int dtype, int c_dtype, int M, int N, int K, const void* A, int lda, const void* B, int ldb, void* C, int ldc,
                           const float* bias, const float* residual, int ldr, int epilogue, void* aux, int ldaux, float alpha, hipStream_t s,
                           int group_m);
extern int g_lpi_tuning[16];

// argument checks shared by lpi_gemm_nt and lpi_gemm_nt_grouped
static int gemm_nt_check(int dtype, int c_dtype, int M, int N, int K, const void* A, int lda, const void* B, int ldb, void* C, int ldc,
                         const float* bias, const float* residual, int ldr, int epilogue, void* aux, int ldaux)
{

    const int esz = dtype == LPI_F32 ? 4 : 2;
    const int csz = c_dtype == LPI_F32 ? 4 : 2;
    if (epilogue == LPI_EPI_RES_ROWSTATS) {      // the fp16 residual epilogue + the row statistics of its output: aux = f32 slots, N/128 x 2 x ldaux
        if (c_dtype != LPI_F16 || dtype == LPI_F32) return LPI_ENOSYS;
        if (!residual || !aux || ldaux < M || (ldaux & 3) || ((uintptr_t)aux & 15)) return LPI_EINVAL;
        aux = nullptr;      // the checks below are for an operand-typed [M, N] aux tile
    }
    if (c_dtype == LPI_F16 && dtype == LPI_BF16 && ((epilogue != LPI_EPI_NONE && epilogue != LPI_EPI_RES_ROWSTATS) || !residual)) return LPI_ENOSYS;
    if (c_dtype == LPI_F16 && dtype == LPI_F32) return LPI_ENOSYS;
    const bool ln = epilogue == LPI_EPI_LN || epilogue == LPI_EPI_LN_QUICKGELU;
    if (c_dtype == LPI_BF16 && dtype == LPI_F16 && !ln) return LPI_ENOSYS;      // f16 operands write f16 or f32 (bf16: the LN-fold GEMMs of bf16 mode)
    if (ln) {      // LayerNorm folded into the GEMM: residual = mean[ldr] | rstd[ldr] | c1[N] (f32)
        if (dtype == LPI_F32 || c_dtype == LPI_F32) return LPI_ENOSYS;
        if (!residual || !bias || ldr < M || (ldr & 3) || ((uintptr_t)residual & 15)) return LPI_EINVAL;
        if (epilogue == LPI_EPI_LN && aux) return LPI_EINVAL;
    }
    const int bk = ROW_BYTES / esz;
    if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0) return LPI_EINVAL;
    if (M % BM || N % BN || K % bk) return LPI_EINVAL;
    if ((lda * esz) % 16 || (ldb * esz) % 16 || (ldc * csz) % 8 || lda < K || ldb < K || ldc < N) return LPI_EINVAL;
    if (((uintptr_t)A | (uintptr_t)B | (uintptr_t)C) & 15) return LPI_EINVAL;
    if (residual && !ln && (ldr < N || (ldr & 3) || ((uintptr_t)residual & 15))) return LPI_EINVAL;
    if (bias && ((uintptr_t)bias & 15)) return LPI_EINVAL;
    if (aux && (ldaux < N || ((uintptr_t)aux & 7) || (ldaux * esz) % 8)) return LPI_EINVAL;
    if (epilogue == LPI_EPI_DQUICKGELU && !aux) return LPI_EINVAL;
    return 0;
}

extern "C" int lpi_gemm_nt(int dtype, int c_dtype, int M, int N, int K, const void* A, int lda, const void* B, int ldb,
                           void* C, int ldc, const float* bias, const void* residual_, int ldr, int epilogue, void* aux,
                           int ldaux, float alpha, void* stream)
{
    const float* residual = (const float*)residual_;      // fp16 when c_dtype == LPI_F16 (re-typed in the epilogue)
    if (int e = gemm_nt_check(dtype, c_dtype, M, N, K, A, lda, B, ldb, C, ldc, bias, residual, ldr, epilogue, aux, ldaux)) return e;
    if ((epilogue == LPI_EPI_LN || epilogue == LPI_EPI_LN_QUICKGELU) && alpha != 1.0f) return LPI_EINVAL;      // LN(x) W^T + b has no scale (the epilogue does not multiply by one)
    hipStream_t s = (hipStream_t)stream;
    if (epilogue == LPI_EPI_LN || epilogue == LPI_EPI_LN_QUICKGELU || epilogue == LPI_EPI_RES_ROWSTATS) {
        // only the persistent 256x256 kernel has the LN-fold and the row-statistics epilogues
        if (!lpi_gemm256_eligible(dtype, M, N, K)) return LPI_ENOSYS;
        return lpi_gemm256p_launch(dtype, c_dtype, M, N, K, A, lda, B, ldb, C, ldc, bias, residual, ldr, epilogue, aux, ldaux, alpha, s);
    }
    // Half-empty launches: fewer than tuning key 5 (default 160) 256x256 tiles -> 256x128 tiles, twice the workgroups (bf16 only:
    // the f32 path is MFMA-bound at any tile size).  Key 5 = 0 disables it.
    if (dtype != LPI_F32 && g_lpi_tuning[5] > 0 && lpi_gemm256_eligible(dtype, M, N, K) && (M / 256) * (N / 256) < g_lpi_tuning[5] &&
        (M / 256) * (N / 256) >= 16 && lpi_gemm256x128_eligible(dtype, M, N, K)) {
        const int rc = lpi_gemm256x128_launch(dtype, c_dtype, M, N, K, A, lda, B, ldb, C, ldc, bias, residual, ldr, epilogue, aux, ldaux, alpha, s,
                                              g_lpi_tuning[4] > 0 ? g_lpi_tuning[4] : 8);
        if (rc != LPI_ENOSYS) return rc;
    }
    // 256x256 8-phase kernel when the shape gives it enough tiles to fill the chip (tuning keys 0 / 1 = minimum tile count for bf16 / f32)
    if (lpi_gemm256_eligible(dtype, M, N, K) && (M / 256) * (N / 256) >= g_lpi_tuning[dtype == LPI_F32 ? 1 : 0])
        return lpi_gemm256_launch(dtype, c_dtype, M, N, K, A, lda, B, ldb, C, ldc, bias, residual, ldr, epilogue, aux, ldaux, alpha, s);
    if (dtype == LPI_F32 && c_dtype == LPI_F32)
        return dispatch_epi<float, float>(epilogue, M, N, K, A, lda, B, ldb, C, ldc, bias, residual, ldr, aux, ldaux, alpha, s);
    if (dtype == LPI_BF16 && c_dtype == LPI_BF16)
        return dispatch_epi<bf16_t, bf16_t>(epilogue, M, N, K, A, lda, B, ldb, C, ldc, bias, residual, ldr, aux, ldaux, alpha, s);
    if (dtype == LPI_BF16 && c_dtype == LPI_F32)
        return dispatch_epi<bf16_t, float>(epilogue, M, N, K, A, lda, B, ldb, C, ldc, bias, residual, ldr, aux, ldaux, alpha, s);
    if (dtype == LPI_BF16 && c_dtype == LPI_F16)      // fp16 residual stream: x_out = x_in + (A.B^T + bias), both fp16
        return launch_impl<bf16_t, f16_t, LPI_EPI_NONE, true, false>(M, N, K, A, lda, B, ldb, C, ldc, bias, residual, ldr, aux, ldaux, alpha, s);
    if (dtype == LPI_F16 && c_dtype == LPI_F16)
        return dispatch_epi<f16_t, f16_t>(epilogue, M, N, K, A, lda, B, ldb, C, ldc, bias, residual, ldr, aux, ldaux, alpha, s);
    if (dtype == LPI_F16 && c_dtype == LPI_F32)
        return dispatch_epi<f16_t, float>(epilogue, M, N, K, A, lda, B, ldb, C, ldc, bias, residual, ldr, aux, ldaux, alpha, s);
    return LPI_ENOSYS;
}

// 1 if lpi_gemm_nt takes LPI_EPI_LN / LPI_EPI_LN_QUICKGELU for this operand type and shape (the persistent 256x256 kernel's shapes), else 0
extern "C" int lpi_gemm_ln_supported(int dtype, int M, int N, int K)
{
    return (dtype == LPI_F16 || dtype == LPI_BF16) && M > 0 && N > 0 && K > 0 && lpi_gemm256_eligible(dtype, M, N, K) ? 1 : 0;
}

int lpi_gemm256p_launch2(int dtype, int c_dtype, int epilogue, float alpha, const lpi_gemm_desc* d, hipStream_t s);
static thread_local int t_last_grouped = 0;
extern "C" int lpi_gemm_last_grouped(void) { return t_last_grouped; }

extern "C" int lpi_gemm_nt_grouped(int dtype, int c_dtype, int epilogue, float alpha, int count, const lpi_gemm_desc* d, void* stream)
{
    if ((epilogue == LPI_EPI_LN || epilogue == LPI_EPI_LN_QUICKGELU) && alpha != 1.0f) return LPI_EINVAL;
    if (count <= 0 || !d) return LPI_EINVAL;
    t_last_grouped = 0;
    for (int i = 0; i < count; ++i)
        if (int e = gemm_nt_check(dtype, c_dtype, d[i].M, d[i].N, d[i].K, d[i].A, d[i].lda, d[i].B, d[i].ldb, d[i].C, d[i].ldc, d[i].bias,
                                  (const float*)d[i].residual, d[i].ldr, epilogue, d[i].aux, d[i].ldaux)) return e;
    // one persistent launch: two bf16 / f16 problems that the 256x256 kernel takes, with an epilogue the persistent kernel has (store-only,
    // or a 2-byte side tile), together at least a round of tiles (tuning key 8 != 0 = never group: A/B switch)
    bool group = count == 2 && dtype != LPI_F32 && g_lpi_tuning[2] >= 0 && g_lpi_tuning[8] == 0;
    if (group) {
        int tiles = 0;
        for (int i = 0; i < 2 && group; ++i) {
            const bool res = d[i].residual != nullptr && epilogue != LPI_EPI_LN && epilogue != LPI_EPI_LN_QUICKGELU;      // LN block: not a tile
            const bool side16 = (res && c_dtype == LPI_F16) || epilogue == LPI_EPI_DQUICKGELU;      // LPI_EPI_RES_ROWSTATS: res and fp16 by its checks
            const bool loads = res || epilogue == LPI_EPI_DQUICKGELU;
            group = lpi_gemm256_eligible(dtype, d[i].M, d[i].N, d[i].K) && (!loads || (side16 && g_lpi_tuning[2] != 2));
            tiles += (d[i].M / 256) * (d[i].N / 256);
        }
        group = group && tiles >= 256 && (d[0].residual != nullptr) == (d[1].residual != nullptr) && (d[0].aux != nullptr) == (d[1].aux != nullptr);
    }
    if (group) {
        const int rc = lpi_gemm256p_launch2(dtype, c_dtype, epilogue, alpha, d, (hipStream_t)stream);
        if (rc == 0) t_last_grouped = 1;
        if (rc != LPI_ENOSYS) return rc;
    }
    for (int i = 0; i < count; ++i)
        if (int e = lpi_gemm_nt(dtype, c_dtype, d[i].M, d[i].N, d[i].K, d[i].A, d[i].lda, d[i].B, d[i].ldb, d[i].C, d[i].ldc, d[i].bias, d[i].residual,
                                d[i].ldr, epilogue, d[i].aux, d[i].ldaux, alpha, stream)) return e;
    return 0;
}
