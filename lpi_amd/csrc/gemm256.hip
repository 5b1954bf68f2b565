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
#include "gemm256_tile.h"
#include "gemm256x128_tile.h"

extern int g_lpi_tuning[16];

namespace {

using namespace t256;

template <typename T, typename TC, int EPI, bool RES, bool SAVE_U>
__global__ __launch_bounds__(NTHR, 2) void gemm256_kernel(
    int M, int N, int K, const T* __restrict__ A, int lda, const T* __restrict__ B, int ldb,
    TC* __restrict__ C, int ldc, const float* __restrict__ bias, const float* __restrict__ residual, int ldr,
    typename AuxT<T>::type* __restrict__ aux, int ldaux, float alpha, int tiles_m, int tiles_n, int stagger, int group_m)
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

    t256::tile<T, TC, EPI, RES, SAVE_U>(m0, n0, K, A, lda, B, ldb, C, ldc, bias, residual, ldr, aux, ldaux, alpha, smem);
}

// Hybrid launch for tile counts that leave a short last round: the first n_full (a multiple of 256) tiles run as 256x256 tiles,
// the remaining rem <= 128 tiles as 2*rem tiles of 256x128 — one round of half tiles instead of one round of a half-empty chip
// (at M = 54 528 every N = 768 GEMM has 639 = 512 + 127 tiles).  Tile ids follow the same XCD-aware order as the plain kernel;
// the two halves of a tail tile sit 8 workgroup ids apart, i.e. on the same XCD, so they share the A rows in its L2.
template <typename T, typename TC, int EPI, bool RES, bool SAVE_U>
__global__ __launch_bounds__(NTHR, 2) void gemm256_tail_kernel(
    int M, int N, int K, const T* __restrict__ A, int lda, const T* __restrict__ B, int ldb,
    TC* __restrict__ C, int ldc, const float* __restrict__ bias, const float* __restrict__ residual, int ldr,
    typename AuxT<T>::type* __restrict__ aux, int ldaux, float alpha, int tiles_m, int tiles_n, int n_full, int group_m)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nwg = tiles_m * tiles_n;
    int tile_id, half = -1;
    if ((int)blockIdx.x < n_full) {
        const int bid = blockIdx.x;       // n_full % 8 == 0: the XCD remap of the first n_full ids over the FULL tile list
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        tile_id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    } else {
        const int j = blockIdx.x - n_full;
        tile_id = -1 - ((j >> 4) * 8 + (j & 7));      // k-th tile NOT taken by the first n_full ids (resolved below)
        half = (j >> 3) & 1;
    }
    // The first n_full workgroup ids take, per XCD x, that XCD's first n_full/8 tiles; the tail takes the rest in the same order.
    if (tile_id < 0) {
        const int k = -1 - tile_id;                   // 0 .. rem-1 (padded to a multiple of 8: guard below)
        const int q = nwg >> 3, r = nwg & 7, per = n_full >> 3;
        const int xcd = k & 7, idx = per + (k >> 3);  // the (k>>3)-th leftover tile of XCD `xcd`
        const int count = q + (xcd < r ? 1 : 0);
        if (idx >= count) return;
        tile_id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int group = tile_id / (group_m * tiles_n);
    const int first_m = group * group_m;
    const int gsz = min(tiles_m - first_m, group_m);
    const int in_group = tile_id - group * group_m * tiles_n;
    const int tm = first_m + in_group % gsz;
    const int tn = in_group / gsz;
    if (half < 0) t256::tile<T, TC, EPI, RES, SAVE_U>(tm * 256, tn * 256, K, A, lda, B, ldb, C, ldc, bias, residual, ldr, aux, ldaux, alpha, smem);
    else t128::tile<T, TC, EPI, RES, SAVE_U>(tm * 256, tn * 256 + half * 128, K, A, lda, B, ldb, C, ldc, bias, residual, ldr, aux, ldaux, alpha, smem);
}

template <typename T, typename TC, int EPI, bool RES, bool SAVE_U>
int launch256_impl(int M, int N, int K, const void* A, int lda, const void* B, int ldb, void* C, int ldc, const float* bias,
              const float* residual, int ldr, void* aux, int ldaux, float alpha, hipStream_t s)
{
    const int tm = M / T256, tn = N / T256;
    const int nwg = tm * tn, rem = nwg % 256, n_full = nwg - rem;
    const int gm = g_lpi_tuning[4] > 0 ? g_lpi_tuning[4] : 8;
    // a short last round (<= 128 of 256 CUs busy) runs as 256x128 half tiles inside the same launch (tuning key 6, default on;
    // bf16 operands only: the f32 path is MFMA-bound and its K-tile count per 128-byte row differs)
    if (sizeof(T) == 2 && g_lpi_tuning[6] != 0 && n_full >= 256 && rem > 0 && rem <= 128 && K / (ROWB / (int)sizeof(T)) >= 2) {
        auto tk = gemm256_tail_kernel<T, TC, EPI, RES, SAVE_U>;
        constexpr int LDS_TAIL = t128::LDS_BYTES > LDS_BYTES ? t128::LDS_BYTES : LDS_BYTES;
        static LdsOnce tail_once;
        if (int e = lpi_ensure_lds(tail_once, (const void*)tk, LDS_TAIL)) return e;
        lpi_note_gemm_kernel(LPI_GEMM_K_256_TAIL);
        const int q = nwg >> 3, r = nwg & 7;
        const int max_left = q + (r ? 1 : 0) - (n_full >> 3);      // leftover tiles of the fullest XCD
        LPI_LAUNCH(tk, dim3(n_full + 16 * max_left), dim3(NTHR), LDS_TAIL, s, M, N, K, (const T*)A, lda, (const T*)B, ldb, (TC*)C, ldc, bias, residual,
                   ldr, (typename AuxT<T>::type*)aux, ldaux, alpha, tm, tn, n_full, gm);
        LPI_CHECK_LAST();
        return 0;
    }
    auto kern = gemm256_kernel<T, TC, EPI, RES, SAVE_U>;
    static LdsOnce once;
    if (int e = lpi_ensure_lds(once, (const void*)kern, LDS_BYTES)) return e;
    lpi_note_gemm_kernel(LPI_GEMM_K_256);
    LPI_LAUNCH(kern, dim3(tm * tn), dim3(NTHR), LDS_BYTES, s, M, N, K, (const T*)A, lda, (const T*)B, ldb, (TC*)C, ldc, bias, residual,
               ldr, (typename AuxT<T>::type*)aux, ldaux, alpha, tm, tn, 0, g_lpi_tuning[4] > 0 ? g_lpi_tuning[4] : 8);
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

int lpi_gemm256p_launch(int dtype, int c_dtype, int M, int N, int K, const void* A, int lda, const void* B, int ldb, void* C, int ldc,
                        const float* bias, const float* residual, int ldr, int epilogue, void* aux, int ldaux, float alpha, hipStream_t s);

int lpi_gemm256_launch(int dtype, int c_dtype, int M, int N, int K, const void* A, int lda, const void* B, int ldb, void* C, int ldc,
                       const float* bias, const float* residual, int ldr, int epilogue, void* aux, int ldaux, float alpha, hipStream_t s)
{
    // bf16 operands: the persistent kernel (gemm256p.hip: the K-tile ring runs across a workgroup's tiles, the next tile's first K-tile
    // lands under the epilogue); tuning key 2 = -1 keeps one tile per workgroup (A/B switch; same results bit for bit)
    // Store-only epilogues: the next tile's first K-tile lands under the epilogue (qkv 184 -> 162 us, fc+gelu 297 -> 275, dout 61 -> 56).
    // Epilogues that LOAD a 2-byte tile (the fp16 residual stream, the bf16 u of gelu'(u)): that tile comes to LDS by LDS-DMA two
    // passes ahead, so the epilogue issues no vector-memory load (with plain loads they queued behind the in-flight DMA in the in-order
    // vmcnt queue and the persistent kernel LOST 8-30 % on them).  f32 residuals (few-row GEMMs only) stay on the one-tile kernels.
    // Tuning key 2: -1 one tile per workgroup everywhere; 2 persistent for store-only epilogues only (A/B).
    const bool loads_in_epilogue = residual != nullptr || epilogue == LPI_EPI_DQUICKGELU;
    const bool side16 = (residual != nullptr && c_dtype == LPI_F16) || epilogue == LPI_EPI_DQUICKGELU;
    if (dtype != LPI_F32 && g_lpi_tuning[2] >= 0 && (!loads_in_epilogue || (side16 && g_lpi_tuning[2] != 2))) {
        const int rc = lpi_gemm256p_launch(dtype, c_dtype, M, N, K, A, lda, B, ldb, C, ldc, bias, residual, ldr, epilogue, aux, ldaux, alpha, s);
        if (rc != LPI_ENOSYS) return rc;
    }
    if (dtype == LPI_F32 && c_dtype == LPI_F32)
        return dispatch256<float, float>(epilogue, M, N, K, A, lda, B, ldb, C, ldc, bias, residual, ldr, aux, ldaux, alpha, s);
    if (dtype == LPI_BF16 && c_dtype == LPI_BF16)
        return dispatch256<bf16_t, bf16_t>(epilogue, M, N, K, A, lda, B, ldb, C, ldc, bias, residual, ldr, aux, ldaux, alpha, s);
    if (dtype == LPI_BF16 && c_dtype == LPI_F32)
        return dispatch256<bf16_t, float>(epilogue, M, N, K, A, lda, B, ldb, C, ldc, bias, residual, ldr, aux, ldaux, alpha, s);
    if (dtype == LPI_BF16 && c_dtype == LPI_F16 && epilogue == LPI_EPI_NONE && residual)
        return launch256_impl<bf16_t, f16_t, LPI_EPI_NONE, true, false>(M, N, K, A, lda, B, ldb, C, ldc, bias, residual, ldr, aux, ldaux, alpha, s);
    if (dtype == LPI_F16 && c_dtype == LPI_F16)
        return dispatch256<f16_t, f16_t>(epilogue, M, N, K, A, lda, B, ldb, C, ldc, bias, residual, ldr, aux, ldaux, alpha, s);
    if (dtype == LPI_F16 && c_dtype == LPI_F32)
        return dispatch256<f16_t, float>(epilogue, M, N, K, A, lda, B, ldb, C, ldc, bias, residual, ldr, aux, ldaux, alpha, s);
    return LPI_ENOSYS;
}
