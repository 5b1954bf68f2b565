// Few-row NT GEMM in ONE launch:  C[M,N] = epi(alpha * A[M,K] . B[N,K]^T + bias) + residual  for M <= a few hundred rows
//
// replaces: the nn.Linear / MultiheadAttention projections of the reference on the POOLED rows of the last block and the two heads
// (retrieval/models/clip/model.py:172-177,185 on the class-token / EOT rows that model.py:257 and prompt_learner.py:61 keep;
// `x @ self.proj`, model.py:257; prompt_learner.py:61) and their dgrads — B rows per tower instead of B x L.
//
// Until round 6 these ran as split-K partial GEMM + reduction (gemm.hip: two launches, K slices x M x N f32 partials through HBM, 15-35 us per op for a
// 0.1-0.8 GFLOP problem).  Here a workgroup owns a 32 x 32 output tile for the WHOLE K range:
//  * 8 waves share K in 128-byte blocks (block b -> wave b mod 8); every wave accumulates the 32 x 32 tile over its blocks on the matrix cores (2 x 2 MFMA
//    16x16 tiles).  A wave loads a block COALESCED (8 lanes = one 128-byte row segment, 8 rows per instruction: a fragment-order load — 16 rows x 16 bytes per
//    quarter wave — moves 16 B per clock through the CU's address path, this 64) into registers, passes it through a wave-private 8 KiB LDS image (swizzled
//    as in gemm.hip; no barrier: LDS operations of one wave execute in order) and reads it back as MFMA fragments; two blocks (16 loads of 16 B per lane) are
//    in flight per wave, re-issued as soon as their registers have been written to LDS;
//  * the eight partial tiles meet in LDS (over the images), waves 0-3 sum them in wave order and run the usual fused epilogue (gemm_epilogue.h) on four
//    consecutive columns per lane: deterministic, no atomics, no scratch;
//  * (M/32) x (N/32) workgroups, two resident per CU: 192 for the 256 x 768 outputs, 768 for 256 x 3072 — what limits a launch is the CU's L2 -> L1 path over
//    (32 + 32) x K operand bytes (393 KB at K = 3072: ~3 us), not the K loop of a 128 x 128 tile on a handful of CUs;
//  * two problems (the two towers' GEMM of the same op) in one launch, as everywhere in the lock-stepped step; an XCD gets a contiguous run of the n-major tile
//    list, so a weight panel is fetched by one XCD.
#include "common.h"
#include "gemm_epilogue.h"

namespace {

constexpr int RT = 32;                    // output tile edge
constexpr int RWAVES = 8;
constexpr int RTHREADS = RWAVES * 64;
constexpr int RED_LD = RT + 4;            // floats per row of a partial tile in LDS (16-byte aligned, rows 144 B apart)
constexpr int BLK_BYTES = 128;            // one K block of a row
constexpr int IMG_BYTES = RT * BLK_BYTES; // one operand's block image: 4 KiB
constexpr int WAVE_LDS = 2 * IMG_BYTES;   // A and B images of the block a wave is working on
static_assert(RWAVES * WAVE_LDS >= RWAVES * RT * RED_LD * 4, "the partial tiles overlay the block images");

template <typename T>
struct RowsP {
    int M, N, K, lda, ldb, ldc, ldr, ldaux, tiles_m, nwg;
    const T* A; const T* B; void* C; const float* bias; const float* residual; void* aux;
};

template <typename T, typename TC, int EPI, bool RES, bool SAVE_U>
__device__ __forceinline__ void gemm_rows_tile(const RowsP<T>& p, int bid, float alpha, char* smem)
{
    constexpr int EPC = Elem<T>::EPC;
    constexpr int BLK = BLK_BYTES / (int)sizeof(T);      // elements per K block
    // XCD-aware order: block ids round-robin over the 8 XCDs; each XCD takes a contiguous run of the (n major, m minor) tile list
    {
        const int q = p.nwg >> 3, r = p.nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tn = bid / p.tiles_m, tm = bid - tn * p.tiles_m;
    const int m0 = tm * RT, n0 = tn * RT;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // scalar: the block count below and the branches on it are wave-uniform

    // this wave's K blocks: wave, wave + 8, ...
    const int nblk = p.K / BLK;
    const int cnt = wave < nblk ? (nblk - wave + RWAVES - 1) / RWAVES : 0;

    // staging: instruction i of a block moves rows 8 i + lane / 8, chunk lane % 8; the image is row-major, chunk ^= (row >> 1) & 7
    const int srow = lane >> 3, sc = lane & 7;
    const T* a_src = p.A + (size_t)(m0 + srow) * p.lda + sc * EPC + (size_t)wave * BLK;
    const T* b_src = p.B + (size_t)(n0 + srow) * p.ldb + sc * EPC + (size_t)wave * BLK;
    const size_t a_step = (size_t)8 * p.lda, b_step = (size_t)8 * p.ldb;
    char* img = smem + wave * WAVE_LDS;
    int woff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = 8 * i + srow;
        woff[i] = row * BLK_BYTES + ((sc ^ ((row >> 1) & 7)) << 4);
    }
    // fragments: lane reads row (l & 15) of a 16-row sub tile, logical chunk 4 ks + (l >> 4)
    const int frow = lane & 15, fg = lane >> 4;
    int foff[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) foff[ks] = frow * BLK_BYTES + ((((ks << 2) | fg) ^ (frow >> 1)) << 4);

    f32x4 acc[2][2];      // [n sub tile][m sub tile]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // (named members, values returned: with arrays passed by reference hipcc kept the two block sets in scratch memory)
    struct BlkRegs { uint4 a0, a1, a2, a3, b0, b1, b2, b3; };
    auto load = [&](int t) -> BlkRegs {      // this wave's t-th block
        const T* ap = a_src + (size_t)t * (RWAVES * BLK);
        const T* bp = b_src + (size_t)t * (RWAVES * BLK);
        BlkRegs r;
        r.a0 = *reinterpret_cast<const uint4*>(ap);
        r.a1 = *reinterpret_cast<const uint4*>(ap + a_step);
        r.a2 = *reinterpret_cast<const uint4*>(ap + 2 * a_step);
        r.a3 = *reinterpret_cast<const uint4*>(ap + 3 * a_step);
        r.b0 = *reinterpret_cast<const uint4*>(bp);
        r.b1 = *reinterpret_cast<const uint4*>(bp + b_step);
        r.b2 = *reinterpret_cast<const uint4*>(bp + 2 * b_step);
        r.b3 = *reinterpret_cast<const uint4*>(bp + 3 * b_step);
        return r;
    };
    auto put = [&](const BlkRegs& r) {
        *reinterpret_cast<uint4*>(img + woff[0]) = r.a0;
        *reinterpret_cast<uint4*>(img + woff[1]) = r.a1;
        *reinterpret_cast<uint4*>(img + woff[2]) = r.a2;
        *reinterpret_cast<uint4*>(img + woff[3]) = r.a3;
        *reinterpret_cast<uint4*>(img + IMG_BYTES + woff[0]) = r.b0;
        *reinterpret_cast<uint4*>(img + IMG_BYTES + woff[1]) = r.b1;
        *reinterpret_cast<uint4*>(img + IMG_BYTES + woff[2]) = r.b2;
        *reinterpret_cast<uint4*>(img + IMG_BYTES + woff[3]) = r.b3;
    };
    auto mul = [&]() {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            Chunk fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                fa[i].u = *reinterpret_cast<const uint4*>(img + i * 16 * BLK_BYTES + foff[ks]);
                fb[i].u = *reinterpret_cast<const uint4*>(img + IMG_BYTES + i * 16 * BLK_BYTES + foff[ks]);
            }
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) mma_chunk<T>(acc[ni][mi], fb[ni], fa[mi]);
        }
    };
    // Two blocks in flight per wave.  The steady loop is free of conditional loads — hipcc counts vmcnt exactly only along straight-line code: with a load under a
    // branch it waits for EVERYTHING in flight before the next ds_write, which serialises the round trips — and the short cases have their own arm.
    if (cnt >= 3) {
        BlkRegs r0 = load(0), r1 = load(1);
        int t = 0;
        while (t + 3 < cnt) {
            put(r0);
            r0 = load(t + 2);
            mul();
            put(r1);
            r1 = load(t + 3);
            mul();
            t += 2;
        }
        const bool three = cnt - t == 3;      // 2 or 3 blocks left: t and t + 1 are in flight
        put(r0);
        if (three) r0 = load(t + 2);
        mul();
        put(r1);
        mul();
        if (three) {
            put(r0);
            mul();
        }
    } else if (cnt == 2) {
        BlkRegs r0 = load(0), r1 = load(1);
        put(r0);
        mul();
        put(r1);
        mul();
    } else if (cnt == 1) {
        BlkRegs r0 = load(0);
        put(r0);
        mul();
    }
    __syncthreads();      // every wave is done with its images: the partial tiles go over them

    // ---- the eight partial tiles through LDS: lane holds C[m = 16 mi + (l & 15)][n = 16 ni + 4 (l >> 4) + 0..3] -----------------------
    float* red = reinterpret_cast<float*>(smem);
    float* mine = red + wave * (RT * RED_LD);
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
            *reinterpret_cast<f32x4*>(mine + (mi * 16 + frow) * RED_LD + ni * 16 + (fg << 2)) = acc[ni][mi];
    __syncthreads();
    if (tid < RT * RT / 4) {
        const int r = tid >> 3, c = (tid & 7) << 2;
        const float* src = red + r * RED_LD + c;
        f32x4 v = *reinterpret_cast<const f32x4*>(src);
#pragma unroll
        for (int w = 1; w < RWAVES; ++w) v += *reinterpret_cast<const f32x4*>(src + w * (RT * RED_LD));
        const int row = m0 + r, col = n0 + c;
        f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
        if (p.bias) bv = *reinterpret_cast<const f32x4*>(p.bias + col);
        gemm_epilogue_store<T, TC, EPI, RES, SAVE_U>(v, row, col, (TC*)p.C, p.ldc, bv, alpha, p.residual, p.ldr, (typename AuxT<T>::type*)p.aux, p.ldaux);
    }
}

template <typename T, typename TC, int EPI, bool RES, bool SAVE_U>
__global__ __launch_bounds__(RTHREADS, 2) void gemm_rows_kernel(RowsP<T> p0, RowsP<T> p1, float alpha)
{
    __shared__ __attribute__((aligned(16))) char smem[RWAVES * WAVE_LDS];
    if ((int)blockIdx.x < p0.nwg) gemm_rows_tile<T, TC, EPI, RES, SAVE_U>(p0, blockIdx.x, alpha, smem);
    else gemm_rows_tile<T, TC, EPI, RES, SAVE_U>(p1, blockIdx.x - p0.nwg, alpha, smem);
}

template <typename T, typename TC, int EPI, bool RES, bool SAVE_U>
int rows_impl(int count, const lpi_gemm_desc* d, float alpha, hipStream_t s)
{
    RowsP<T> g[2];
    for (int i = 0; i < 2; ++i) {
        const lpi_gemm_desc& q = d[i < count ? i : 0];
        const int tm = q.M / RT, tn = q.N / RT;
        g[i] = RowsP<T>{q.M, q.N, q.K, q.lda, q.ldb, q.ldc, q.ldr, q.ldaux, tm, i < count ? tm * tn : 0,
                        (const T*)q.A, (const T*)q.B, q.C, q.bias, (const float*)q.residual, q.aux};
    }
    lpi_note_gemm_kernel(LPI_GEMM_K_ROWS);
    LPI_LAUNCH((gemm_rows_kernel<T, TC, EPI, RES, SAVE_U>), dim3(g[0].nwg + g[1].nwg), dim3(RTHREADS), 0, s, g[0], g[1], alpha);
    LPI_CHECK_LAST();
    return 0;
}

template <typename T, typename TC>
int rows_dispatch(int epi, int count, const lpi_gemm_desc* d, float alpha, hipStream_t s)
{
    const bool res = d[0].residual != nullptr, ax = d[0].aux != nullptr;
    if (count == 2 && ((d[1].residual != nullptr) != res || (d[1].aux != nullptr) != ax)) return LPI_EINVAL;
#define RW(EPI, RES, SU) return rows_impl<T, TC, EPI, RES, SU>(count, d, alpha, s)
    if (epi == LPI_EPI_NONE) { if (res) RW(LPI_EPI_NONE, true, false); RW(LPI_EPI_NONE, false, false); }
    if (res) return LPI_ENOSYS;
    if (epi == LPI_EPI_QUICKGELU) { if (ax) RW(LPI_EPI_QUICKGELU, false, true); RW(LPI_EPI_QUICKGELU, false, false); }
    if (epi == LPI_EPI_DQUICKGELU) { if (!ax) return LPI_EINVAL; RW(LPI_EPI_DQUICKGELU, false, false); }
#undef RW
    return LPI_EINVAL;
}

}  // namespace

extern "C" int lpi_gemm_nt_rows_supported(int dtype, int M, int N, int K)
{
    const int blk = dtype == LPI_F32 ? 32 : 64;      // whole 128-byte K blocks
    return (M > 0 && N > 0 && K > 0 && M % RT == 0 && N % RT == 0 && K % blk == 0 && (long)(M / RT) * (N / RT) <= (1L << 20)) ? 1 : 0;
}

extern "C" int lpi_gemm_nt_rows(int dtype, int c_dtype, int epilogue, float alpha, int count, const lpi_gemm_desc* d, void* stream)
{
    if (!d || count < 1 || count > 2) return LPI_EINVAL;
    const int esz = dtype == LPI_F32 ? 4 : 2;
    const int csz = c_dtype == LPI_F32 ? 4 : 2;
    for (int i = 0; i < count; ++i) {
        const lpi_gemm_desc& q = d[i];
        if (!q.A || !q.B || !q.C) return LPI_EINVAL;
        if (!lpi_gemm_nt_rows_supported(dtype, q.M, q.N, q.K)) return LPI_EINVAL;
        if ((q.lda * esz) % 16 || (q.ldb * esz) % 16 || (q.ldc * csz) % 8 || q.lda < q.K || q.ldb < q.K || q.ldc < q.N) return LPI_EINVAL;
        if (((uintptr_t)q.A | (uintptr_t)q.B | (uintptr_t)q.C) & 15) return LPI_EINVAL;
        if (q.residual && (q.ldr < q.N || (q.ldr & 3) || ((uintptr_t)q.residual & 15))) return LPI_EINVAL;
        if (q.bias && ((uintptr_t)q.bias & 15)) return LPI_EINVAL;
        if (q.aux && (q.ldaux < q.N || ((uintptr_t)q.aux & 7) || (q.ldaux * esz) % 8)) return LPI_EINVAL;
    }
    if (c_dtype == LPI_F16 && dtype != LPI_F16) return LPI_ENOSYS;     // bf16 mode: the fp16 residual stream never has this few rows
    hipStream_t s = (hipStream_t)stream;
    if (dtype == LPI_F32 && c_dtype == LPI_F32) return rows_dispatch<float, float>(epilogue, count, d, alpha, s);
    if (dtype == LPI_BF16 && c_dtype == LPI_BF16) return rows_dispatch<bf16_t, bf16_t>(epilogue, count, d, alpha, s);
    if (dtype == LPI_BF16 && c_dtype == LPI_F32) return rows_dispatch<bf16_t, float>(epilogue, count, d, alpha, s);
    if (dtype == LPI_F16 && c_dtype == LPI_F16) return rows_dispatch<f16_t, f16_t>(epilogue, count, d, alpha, s);
    if (dtype == LPI_F16 && c_dtype == LPI_F32) return rows_dispatch<f16_t, float>(epilogue, count, d, alpha, s);
    return LPI_ENOSYS;
}
