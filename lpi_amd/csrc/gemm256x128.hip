// 256x128-tile NT GEMM, 8 waves (4x2, each 64x64), 3-stage LDS ring, two phases per K-tile, staggered wave groups.
//
// Same contract and epilogues as gemm256.hip, same results bit for bit.  Used by lpi_gemm_nt for launches that would leave the
// chip half empty with 256x256 tiles: the text tower's N = 512 GEMMs at B = 256 have 118 such tiles for 256 CUs; with 256x128
// tiles they have 236.  (On shapes with plenty of tiles it loses to the 256x256 kernel — 1.5x the L2->LDS bytes per FLOP and 0.5
// ds_read_b128 per MFMA instead of 0.375 — which is why it is not used there: profiles/r01_gemm_ablation.md.)
//
// Pipeline: a K-tile (128 B of K per row; 256 A rows + 128 B rows = 48 KiB) is staged by 6 LDS-DMA instructions per thread into a
// ring of 3 stages (144 KiB), two K-tiles ahead: at K-tile kt, phase 1 issues the DMA of kt+2 into the stage K-tile kt-1 just
// vacated, phase 2 waits with a counted vmcnt(6) (kt+2 stays in flight) so that kt+1 has landed before the next iteration.
// Each phase = 8 ds_read_b128 + 16 MFMA per wave between two raw barriers; waves 4-7 run one barrier behind waves 0-3 so one
// group's LDS reads overlap the other's MFMAs (lgkmcnt(0) before the barrier keeps the restage WAR-safe, as in gemm256.hip).
#include "gemm256x128_tile.h"

namespace {

using namespace t128;

template <typename T, typename TC, int EPI, bool RES, bool SAVE_U>
__global__ __launch_bounds__(NTHR, 2) void gemm256x128_kernel(
    int M, int N, int K, const T* __restrict__ A, int lda, const T* __restrict__ B, int ldb,
    TC* __restrict__ C, int ldc, const float* __restrict__ bias, const float* __restrict__ residual, int ldr,
    typename AuxT<T>::type* __restrict__ aux, int ldaux, float alpha, int tiles_m, int tiles_n, int group_m)
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
    const int m0 = tm * TM, n0 = tn * TN;

    t128::tile<T, TC, EPI, RES, SAVE_U>(m0, n0, K, A, lda, B, ldb, C, ldc, bias, residual, ldr, aux, ldaux, alpha, smem);
}

template <typename T, typename TC, int EPI, bool RES, bool SAVE_U>
int launch_impl(int M, int N, int K, const void* A, int lda, const void* B, int ldb, void* C, int ldc, const float* bias,
                const float* residual, int ldr, void* aux, int ldaux, float alpha, hipStream_t s, int group_m)
{
    const int tm = M / TM, tn = N / TN;
    auto kern = gemm256x128_kernel<T, TC, EPI, RES, SAVE_U>;
    static LdsOnce once;
    if (int e = lpi_ensure_lds(once, (const void*)kern, LDS_BYTES)) return e;
    lpi_note_gemm_kernel(LPI_GEMM_K_256X128);
    LPI_LAUNCH(kern, dim3(tm * tn), dim3(NTHR), LDS_BYTES, s, M, N, K, (const T*)A, lda, (const T*)B, ldb, (TC*)C, ldc, bias, residual,
               ldr, (typename AuxT<T>::type*)aux, ldaux, alpha, tm, tn, group_m);
    LPI_CHECK_LAST();
    return 0;
}

template <typename T, typename TC, int EPI>
int launch(int M, int N, int K, const void* A, int lda, const void* B, int ldb, void* C, int ldc, const float* bias,
           const float* residual, int ldr, void* aux, int ldaux, float alpha, hipStream_t s, int gm)
{
    if constexpr (EPI == LPI_EPI_NONE) {
        if (residual) return launch_impl<T, TC, EPI, true, false>(M, N, K, A, lda, B, ldb, C, ldc, bias, residual, ldr, aux, ldaux, alpha, s, gm);
        return launch_impl<T, TC, EPI, false, false>(M, N, K, A, lda, B, ldb, C, ldc, bias, residual, ldr, aux, ldaux, alpha, s, gm);
    } else {
        if (residual) return LPI_ENOSYS;
        if constexpr (EPI == LPI_EPI_QUICKGELU) {
            if (aux) return launch_impl<T, TC, EPI, false, true>(M, N, K, A, lda, B, ldb, C, ldc, bias, residual, ldr, aux, ldaux, alpha, s, gm);
            return launch_impl<T, TC, EPI, false, false>(M, N, K, A, lda, B, ldb, C, ldc, bias, residual, ldr, aux, ldaux, alpha, s, gm);
        } else {
            if (!aux) return LPI_EINVAL;
            return launch_impl<T, TC, EPI, false, false>(M, N, K, A, lda, B, ldb, C, ldc, bias, residual, ldr, aux, ldaux, alpha, s, gm);
        }
    }
}

template <typename T, typename TC>
int dispatch(int epi, int M, int N, int K, const void* A, int lda, const void* B, int ldb, void* C, int ldc, const float* bias,
             const float* residual, int ldr, void* aux, int ldaux, float alpha, hipStream_t s, int gm)
{
    switch (epi) {
    case LPI_EPI_NONE: return launch<T, TC, LPI_EPI_NONE>(M, N, K, A, lda, B, ldb, C, ldc, bias, residual, ldr, aux, ldaux, alpha, s, gm);
    case LPI_EPI_QUICKGELU: return launch<T, TC, LPI_EPI_QUICKGELU>(M, N, K, A, lda, B, ldb, C, ldc, bias, residual, ldr, aux, ldaux, alpha, s, gm);
    case LPI_EPI_DQUICKGELU: return launch<T, TC, LPI_EPI_DQUICKGELU>(M, N, K, A, lda, B, ldb, C, ldc, bias, residual, ldr, aux, ldaux, alpha, s, gm);
    }
    return LPI_EINVAL;
}

}  // namespace

bool lpi_gemm256x128_eligible(int dtype, int M, int N, int K) {
    const int bk = ROWB / (dtype == LPI_F32 ? 4 : 2);
    return M % TM == 0 && N % TN == 0 && K % bk == 0 && K / bk >= 2;
}

int lpi_gemm256x128_launch(int dtype, int c_dtype, int M, int N, int K, const void* A, int lda, const void* B, int ldb, void* C, int ldc,
                           const float* bias, const float* residual, int ldr, int epilogue, void* aux, int ldaux, float alpha, hipStream_t s,
                           int group_m)
{
    if (dtype == LPI_F32 && c_dtype == LPI_F32)
        return dispatch<float, float>(epilogue, M, N, K, A, lda, B, ldb, C, ldc, bias, residual, ldr, aux, ldaux, alpha, s, group_m);
    if (dtype == LPI_BF16 && c_dtype == LPI_BF16)
        return dispatch<bf16_t, bf16_t>(epilogue, M, N, K, A, lda, B, ldb, C, ldc, bias, residual, ldr, aux, ldaux, alpha, s, group_m);
    if (dtype == LPI_BF16 && c_dtype == LPI_F32)
        return dispatch<bf16_t, float>(epilogue, M, N, K, A, lda, B, ldb, C, ldc, bias, residual, ldr, aux, ldaux, alpha, s, group_m);
    if (dtype == LPI_BF16 && c_dtype == LPI_F16 && epilogue == LPI_EPI_NONE && residual)
        return launch_impl<bf16_t, f16_t, LPI_EPI_NONE, true, false>(M, N, K, A, lda, B, ldb, C, ldc, bias, residual, ldr, aux, ldaux, alpha, s, group_m);
    if (dtype == LPI_F16 && c_dtype == LPI_F16)
        return dispatch<f16_t, f16_t>(epilogue, M, N, K, A, lda, B, ldb, C, ldc, bias, residual, ldr, aux, ldaux, alpha, s, group_m);
    if (dtype == LPI_F16 && c_dtype == LPI_F32)
        return dispatch<f16_t, float>(epilogue, M, N, K, A, lda, B, ldb, C, ldc, bias, residual, ldr, aux, ldaux, alpha, s, group_m);
    return LPI_ENOSYS;
}
