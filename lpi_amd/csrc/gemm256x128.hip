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
#include "common.h"
#include "gemm_epilogue.h"

namespace {

constexpr int TM = 256, TN = 128;
constexpr int ROWB = 128;
constexpr int STAGE_BYTES = (TM + TN) * ROWB;   // 48 KiB
constexpr int NSTAGE = 3;
constexpr int NTHR = 512;
constexpr int ERS = TN * 4 + 16;                // epilogue staging row: 128 f32 + 16 B pad (conflict-free ds_write_b128)
constexpr int LDS_BYTES = NSTAGE * STAGE_BYTES; // 144 KiB >= 128 rows x 528 B of epilogue staging

template <typename T, typename TC, int EPI, bool RES, bool SAVE_U>
__global__ __launch_bounds__(NTHR, 2) void gemm256x128_kernel(
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
    const int m0 = tm * TM, n0 = tn * TN;

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
#pragma unroll 4
        for (int rr = 0; rr < 8; ++rr) {
            const int r = r0 + 2 * rr;
            const f32x4 v = *reinterpret_cast<const f32x4*>(smem + r * ERS + (lane & 31) * 16);
            gemm_epilogue_store<T, TC, EPI, RES, SAVE_U>(v, m0 + ph * 128 + r, ecol, C, ldc, bv, alpha, residual, ldr, aux, ldaux);
        }
    }
}

template <typename T, typename TC, int EPI, bool RES, bool SAVE_U>
int launch_impl(int M, int N, int K, const void* A, int lda, const void* B, int ldb, void* C, int ldc, const float* bias,
                const float* residual, int ldr, void* aux, int ldaux, float alpha, hipStream_t s, int group_m)
{
    const int tm = M / TM, tn = N / TN;
    auto kern = gemm256x128_kernel<T, TC, EPI, RES, SAVE_U>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    LPI_LAUNCH(kern, dim3(tm * tn), dim3(NTHR), LDS_BYTES, s, M, N, K, (const T*)A, lda, (const T*)B, ldb, (TC*)C, ldc, bias, residual,
               ldr, (T*)aux, ldaux, alpha, tm, tn, group_m);
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
    return LPI_ENOSYS;
}
