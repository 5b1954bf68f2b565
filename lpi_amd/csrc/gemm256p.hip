// PERSISTENT 256x256-tile NT GEMM: the phased kernel of gemm256.hip with the K-tile ring kept running ACROSS tiles.
//
// Same contract, epilogues and arithmetic as gemm256.hip (same bits: the per-tile MFMA sequence and the f32 epilogue are unchanged);
// selected by lpi_gemm256_launch for bf16 operands (tuning key 0 >= 0; a negative key 0 keeps the one-tile-per-workgroup kernels).
//
// Why: with one tile per workgroup every tile pays a ~3 us prologue (workgroup launch, address set-up, the first 14 LDS-DMA
// instructions' round trip) with the matrix pipe idle, on top of 16-25 us of main loop + epilogue (profiles/r01_gemm_ablation.md) —
// and the epilogue needs 133 KB of LDS, so nothing of the next tile could be in flight under it.  Here a workgroup owns its CU for
// the whole launch (grid = min(#CUs, tiles)) and walks the tiles w, w + G, w + 2G, ... of the same XCD-aware order:
//   * the LAST two K-tiles of tile t already issue the LDS-DMA of tile t+1's K-tile 0 into the ring slot they vacate (the slot of
//     the even K-tiles), exactly where the schedule would have staged K-tile kt+2 of the same tile: the ring does not drain;
//   * the epilogue runs in FOUR passes of 64 rows through the OTHER slot alone (64 rows x 1024 B = the 64 KiB slot of the odd
//     K-tiles; 16-byte chunks XOR-swizzled by row & 7 instead of padded rows), so tile t+1's K-tile 0 lands under the epilogue of
//     tile t; after two passes every wave waits for its own pieces of it (vmcnt(16): the 16+ stores of those passes may stay in
//     flight), and the remaining two passes' barriers make it visible to all;
//   * after the last pass the three early halves of K-tile 1 are issued and the main loop resumes at K-tile 0 without any wait
//     (its first wait, in phase Y, is the schedule's usual vmcnt(8)).
// LDS: 128 KiB (two ring slots), the epilogue staging is the odd slot; the hybrid short last round (256x128 half tiles, tile body
// of gemm256x128_tile.h) runs after a workgroup's full tiles with its own ring.
#include "gemm256x128_tile.h"

extern int g_lpi_tuning[16];

namespace {

constexpr int T256 = 256;
constexpr int ROWB = 128;                 // bytes per staged row
constexpr int HALF_BYTES = 128 * ROWB;    // 16 KiB
constexpr int BUF_BYTES = 4 * HALF_BYTES; // 64 KiB per K-tile
constexpr int NTHR = 512;
constexpr int OFF_A0 = 0, OFF_A1 = HALF_BYTES, OFF_B0 = 2 * HALF_BYTES, OFF_B1 = 3 * HALF_BYTES;
constexpr int STG = BUF_BYTES;            // epilogue staging = ring slot 1 (the odd K-tiles'), 64 rows x 1024 B
constexpr int LDS_P = 2 * BUF_BYTES;      // 128 KiB
// Internal epilogue kind (never in the C ABI): LPI_EPI_NONE with NO bias, alpha = 1, a 2-byte output and no residual / aux — the dgrad GEMMs of the
// backward (d c_fc, d out_proj, d in_proj, d K/V: a fifth of the step).  The stamps of the diagnostic build (tools/gemm_stamps.py,
// profiles/r04_gemm_shapes.json) show where a store-only epilogue's 7 600 cycles per tile go: 48 % is the f32 staging WRITE (256 KiB per tile through the
// CU's ~79 B/clk ds_write_b128 path: 8 waves wait for it at the barrier behind the writes) and 14 % the barrier in front of it (the staging buffer is
// single: a pass may not overwrite what the pass before is still reading).  With nothing to add per column or per row the accumulators can be rounded to
// the output type BEFORE the staging: half the bytes (ds_write_b64, 32 KiB per pass), so two passes' buffers fit the staging slot (double buffer: ONE
// barrier per pass), whole rows leave as 16-byte stores (half the store instructions).  The same roundings of the same values: bit for bit the generic
// epilogue (acc * 1 + 0 is kept as acc + 0, which turns a -0 into the +0 the generic form stores).
constexpr int EPI_PLAIN16 = 100;
// (the same treatment of LPI_EPI_LN — EPI_LN16, rounds 4-5, opt-in — measured no faster on the whole step: 22.44-22.49 ms against 22.44-22.46; removed in round 6)
// Weight slices (round 5, see launchp_impl) are compiled into every instantiation but the plain QuickGELU + aux one (c_fc WITHOUT the LayerNorm fold:
// LPI_LN_FOLD < 2, an A/B path): its register allocation sits on the edge, and the two compares of the slice fold-back made it spill 56 bytes per lane
// (tests/test_no_spills.py) — a scratch reload there drains the LDS-DMA queue every tile.
template <int EPI, bool SAVE_U> constexpr bool slices_ok() { return !(EPI == LPI_EPI_QUICKGELU && SAVE_U); }

// One GEMM of a launch.  A launch takes one or two of them (a GROUPED launch: the same operand types and epilogue kind, e.g. the
// vision and the text tower's in_proj of the same layer): the second problem's tiles follow the first's in the virtual workgroup order
// (from a multiple of 8, so that an id's XCD is still id & 7), i.e. they fill the first problem's last partial round of CUs and run at
// the persistent kernel's rate instead of as a small launch of their own.
struct PProb {
    const void* A; const void* B; void* C; const float* bias; const void* residual; void* aux;
    int N, K, lda, ldb, ldc, ldr, ldaux, tiles_m, tiles_n;
    int vb0;        // first virtual workgroup id
    int bias_off;   // offset (floats) of its bias vector in the LDS copy (SIDE16 epilogues)
    int group_m;    // order of its tiles inside an XCD's share: low 16 bits = group_m row panels at a time, rows fastest (1 = columns fastest); high 16 bits
                    // (round 5, weight SLICES, see launchp_impl): 0, or the REAL number of row panels while tiles_m / tiles_n describe the virtual problem
                    // "S slices of tiles_n column tiles stacked below each other" (tiles_m = S x real): a tile whose virtual row panel lies in slice g is the
                    // tile (row panel - g x real, column + g x tiles_n) of the real problem
};
struct PGroup {
    PProb p[2];
    int nprob;
    int n_full;        // virtual ids below this run as 256x256 tiles in the persistent loop (a multiple of the grid size, or all)
    int tail_blocks;   // work items of the hybrid short last round (256x128 half tiles), 0 = none
    int reserved;      // keeps the kernel-argument layout: without this word the QuickGELU + aux instantiations allocate differently and SPILL
                       // 56 bytes per lane, with scratch reloads inside the tile loop (checked in the ISA: keep `private_seg_size` of every
                       // gemm256p_kernel instantiation at 0 when this struct changes)
    float alpha;
};

// Diagnostic build only (-DLPI_GEMM_STAMPS, tools/gemm_stamps.py; no stamp executes in the product build): per workgroup the shader cycles
// (s_memtime, wave 0) spent in the K loops, in the epilogues and in the hand-over to the next tile, and the tile count — where a tile's time goes per
// instantiation (MI355X_MICROARCH.md, 'DVFS give-back' item 6: stamps go to memory of their own, no output depends on them)
#ifdef LPI_GEMM_STAMPS
__device__ unsigned long long g_gemm_stamps[1024][8];      // K loop, epilogue, hand-over, tiles | epilogue split: pre-barrier, staging writes, post-write barrier, read + store
#define GS_NOW() __builtin_amdgcn_s_memtime()
#endif

template <typename T, typename TC, int EPI, bool RES, bool SAVE_U>
__global__ __launch_bounds__(NTHR, 2) void gemm256p_kernel(const PGroup grp)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int EPC = Elem<T>::EPC;
    constexpr int BK = ROWB / (int)sizeof(T);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int G = gridDim.x;
    int group_m = 1;      // of the current problem (bind)
    const float alpha = grp.alpha;
    typedef typename AuxT<T>::type TA;
    typedef float f32x8 __attribute__((ext_vector_type(8)));
    constexpr bool LNE = EPI == LPI_EPI_LN || EPI == LPI_EPI_LN_QUICKGELU;
    constexpr bool SLICES = slices_ok<EPI, SAVE_U>();
    constexpr bool P16 = EPI == EPI_PLAIN16;
    constexpr int EPIX = P16 ? LPI_EPI_NONE : EPI;      // the epilogue kind the half-tile body sees
    // the CURRENT problem's operands and geometry (wave-uniform; re-bound by bind() when the workgroup moves on to the next problem)
    const T* A = nullptr; const T* B = nullptr; TC* C = nullptr;
    const float* bias = nullptr; const float* residual = nullptr; TA* aux = nullptr;
    int K = 0, lda = 0, ldb = 0, ldc = 0, ldr = 0, ldaux = 0, tiles_m = 1, tiles_n = 1, nwg = 0, vb0 = 0, bias_off = 0;
    unsigned a_off = 0, b_off = 0;
    const int srow = tid >> 3;
    const int schunk = (tid & 7) ^ ((tid >> 4) & 7);
    auto bind = [&](int pi) {
        const PProb& P = grp.p[pi];
        A = (const T*)P.A; B = (const T*)P.B; C = (TC*)P.C; bias = P.bias; residual = (const float*)P.residual; aux = (TA*)P.aux;
        K = P.K; lda = P.lda; ldb = P.ldb; ldc = P.ldc; ldr = P.ldr; ldaux = P.ldaux; tiles_m = P.tiles_m; tiles_n = P.tiles_n;
        nwg = tiles_m * tiles_n; vb0 = P.vb0; bias_off = P.bias_off; group_m = P.group_m;
        a_off = (unsigned)(((size_t)srow * lda + schunk * EPC) * sizeof(T));
        b_off = (unsigned)(((size_t)srow * ldb + schunk * EPC) * sizeof(T));
    };

    // tile (tm, tn) of the current problem's workgroup id vb (= virtual id - vb0): the same XCD-aware order as gemm256_kernel (ids that
    // share vb & 7 share an XCD)
    auto coords = [&](int vb, int& m0, int& n0) {
        const int q = nwg >> 3, r = nwg & 7, xcd = vb & 7, idx = vb >> 3;
        const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        const int gm = SLICES ? (group_m & 0xffff) : group_m;
        const int group = t / (gm * tiles_n);
        const int first_m = group * gm;
        const int gsz = min(tiles_m - first_m, gm);
        const int in_group = t - group * gm * tiles_n;
        m0 = (first_m + in_group % gsz) * T256;
        n0 = (in_group / gsz) * T256;
        // weight slices (at most three): the virtual row panel folds back into the real one, the column moves on by a slice — two compares, no division
        // (a third integer division at this point made the QuickGELU + aux instantiations spill)
        if constexpr (SLICES) {
            const int wrap = (group_m >> 16) * T256;
            if (wrap) {
                if (m0 >= wrap) { m0 -= wrap; n0 += tiles_n * T256; }
                if (m0 >= wrap) { m0 -= wrap; n0 += tiles_n * T256; }
            }
        }
    };

    // ---- staging: a half tile = 128 rows x 128 B = 2 LDS-DMA instructions of 512 lanes x 16 B (as gemm256_tile.h), but addressed as
    // SGPR base + 32-bit VGPR offset: the base of (tile, K-tile, half, instruction) is scalar arithmetic and a thread keeps two offsets
    // for the whole launch — with 64-bit per-lane pointers re-formed for every tile the kernel spilled, and a scratch reload is a
    // vector-memory operation that sits in the same in-order vmcnt queue as the LDS-DMA.
    const unsigned lds_w = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + wave * 1024);
    auto glds16 = [&](const T* sbase, unsigned voff, unsigned lds_addr) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
    };
    // the A rows' LDS-DMA with its own cache policy (LPI_A_POLICY: "" plain, " nt", " sc1", ...): A/B switch, see profiles/r02_gemm_experiments.md
#ifndef LPI_A_POLICY
#define LPI_A_POLICY ""
#endif
    auto glds16a = [&](const T* sbase, unsigned voff, unsigned lds_addr) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" LPI_A_POLICY "\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
    };
    auto stage_A = [&](int m0, int kt, int h, int buf) {
        const T* sb = A + (size_t)(m0 + h * 128) * lda + (size_t)kt * BK;
        const int lo = buf * BUF_BYTES + (h ? OFF_A1 : OFF_A0);
        glds16a(sb, a_off, lds_w + lo);
        glds16a(sb + (size_t)64 * lda, a_off, lds_w + lo + 8192);
    };
    auto stage_B = [&](int n0, int kt, int h, int buf) {
        const T* sb = B + (size_t)(n0 + h * 128) * ldb + (size_t)kt * BK;
        const int lo = buf * BUF_BYTES + (h ? OFF_B1 : OFF_B0);
        glds16(sb, b_off, lds_w + lo);
        glds16(sb + (size_t)64 * ldb, b_off, lds_w + lo + 8192);
    };

    // ---- fragment offsets within a half tile
    const int frow = lane & 15, fg = lane >> 4, fsw = frow >> 1;
    int foff[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) foff[ks] = frow * ROWB + (((ks << 2) | fg) ^ fsw) * 16;
    const int a_base = (wm * 64) * ROWB;
    const int b_base = (wn * 32) * ROWB;

    Chunk fa[4][2], fb0[2][2], fb1[2][2];
    f32x4 acc[2][2][2][4];   // [nh][ni][mh][mi]
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
#define PHASE_SYNC_IN()                                   \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    \
    __builtin_amdgcn_s_barrier();                         \
    __builtin_amdgcn_sched_barrier(0)
#define PHASE_SYNC_OUT()                                  \
    __builtin_amdgcn_sched_barrier(0);                    \
    __builtin_amdgcn_s_barrier();                         \
    asm volatile("" ::: "memory")

    int nk = 0;              // K / BK of the current problem: even, >= 2 (checked on the host)
    // Epilogues that LOAD (residual, gelu'(u)): their loads share the in-order vmcnt queue with the LDS-DMA, so a next-tile K-tile 0
    // issued from the main loop makes the first epilogue load wait for it.  For them the eight DMA instructions of the next tile's
    // K-tile 0 are issued after pass 1 of the epilogue instead (only passes 2 and 3 queue behind them).
    constexpr bool LATE = RES || EPI == LPI_EPI_DQUICKGELU;
    // SIDE16: ... and when that loaded operand is a 2-byte tile (the fp16 residual stream, or the bf16 pre-activation u of gelu'(u)) it is
    // brought to LDS by LDS-DMA instead: the 64 rows x 512 B of each epilogue pass into one half of ring slot 0 (free once the main loop
    // is in its last K-tile), two passes ahead.  The epilogue then issues no vector-memory LOAD at all: nothing waits in the in-order
    // vmcnt queue behind a DMA, and the load latency of each pass (exposed four times per tile in the one-tile kernel) is hidden.
    // The bias vector goes to LDS once per launch for the same reason.  No next-tile K prefetch for these (slot 0 is taken).
    constexpr bool SIDE16 = (RES && sizeof(TC) == 2) || EPI == LPI_EPI_DQUICKGELU;
    // STATS (LPI_EPI_RES_ROWSTATS = the residual epilogue with SAVE_U set): the row sums / sums of squares of the stored fp16 values leave with the
    // tile — `aux` is the f32 slot buffer (include/lpi_hip.h), two more 4-byte stores per wave and pass (NST below counts them in the vmcnt windows)
    constexpr bool STATS = EPI == LPI_EPI_NONE && RES && SAVE_U && __is_same(TC, f16_t);
    constexpr int NST = 8 + (STATS ? 2 : 0);
    const char* side_base = nullptr;
    int side_ld = 0;
    float* const bias_lds = reinterpret_cast<float*>(smem + LDS_P);
    if constexpr (SIDE16) {
        for (int pi = 0; pi < grp.nprob; ++pi) {
            const float* bp = grp.p[pi].bias;
            float* dst = bias_lds + grp.p[pi].bias_off;
            for (int i = tid * 4; i < grp.p[pi].N; i += NTHR * 4)
                *reinterpret_cast<f32x4*>(dst + i) = bp ? *reinterpret_cast<const f32x4*>(bp + i) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        __syncthreads();
    }
    // pass p of the tile at (m0, n0): staging row s (0..63) = tile row mh*128 + (s>>5)*64 + ((p&1)*2 + ((s>>4)&1))*16 + (s&15); its 256
    // 2-byte elements -> LDS slot-0 half (p & 1), row s at byte s*512 (linear: what a wave reads back is one whole row).
    // Wave w issues instructions i = 0..3: LDS bytes [(4w+i)*1024, +1024) of the half = rows 2(4w+i), 2(4w+i)+1.
    auto stage_side = [&](int m0, int n0, int p) {
        const char* sb = side_base + ((size_t)m0 * side_ld + n0) * 2;
        int l2 = lane;
        asm volatile("" : "+v"(l2));          // keep these address computations where they are used (see the epilogue's note on LICM)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int sr = 2 * (4 * wave + i) + (l2 >> 5);
            const int trow = (p >> 1) * 128 + (sr >> 5) * 64 + ((p & 1) * 2 + ((sr >> 4) & 1)) * 16 + (sr & 15);
            const unsigned voff = (unsigned)trow * (unsigned)side_ld * 2u + (unsigned)(l2 & 31) * 16u;
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(voff), "s"(sb), "s"(lds_w + (p & 1) * 32768 + i * 1024 + wave * 3072) : "memory");
        }
    };

#ifdef LPI_GEMM_STAMPS
    unsigned long long gs_main = 0, gs_epi = 0, gs_next = 0, gs_tiles = 0, gs_t0 = 0, gs_t1 = 0, gs_t2 = 0;
    unsigned long long gs_e[4] = {0, 0, 0, 0}, gs_a = 0, gs_b = 0;
#endif
    const int n_full = grp.n_full;
    int vbv = blockIdx.x;          // virtual id of the workgroup's next tile: blockIdx.x, + G, ...
    bool lds_used = false;
    for (int pi = 0; pi < grp.nprob; ++pi) {
        bind(pi);
        nk = K / BK;
        if constexpr (SIDE16) {
            side_base = RES ? reinterpret_cast<const char*>(residual) : reinterpret_cast<const char*>(aux);
            side_ld = RES ? ldr : ldaux;
        }
        // this problem's 256x256 tiles of the persistent loop: virtual ids [vb0, vend); ids between two problems are padding
        const int vend = min(vb0 + nwg, n_full);
        if (vbv < vb0) vbv += (vb0 - vbv + G - 1) / G * G;
        if (vbv >= vend) continue;
        if (lds_used) {              // the previous problem's last epilogue pass may still be reading the ring slots
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
        lds_used = true;
        int vb = vbv - vb0;          // id within the problem
        const int lend = vend - vb0;
        int m0, n0;
        coords(vb, m0, n0);
        m0 = __builtin_amdgcn_readfirstlane(m0);
        n0 = __builtin_amdgcn_readfirstlane(n0);
        // prologue of the FIRST tile only: K-tile 0 (4 halves) -> slot 0, first three halves of K-tile 1 -> slot 1
        stage_A(m0, 0, 0, 0); stage_B(n0, 0, 0, 0); stage_B(n0, 0, 1, 0); stage_A(m0, 0, 1, 0);
        stage_A(m0, 1, 0, 1); stage_B(n0, 1, 0, 1); stage_B(n0, 1, 1, 1);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        bool first = true;
        for (;;) {
            const int nvb = vb + G;
            const bool more_tiles = nvb < lend;
            const bool has_next = more_tiles && !LATE;       // next tile's K-tile 0 staged from the main loop
            int nm0 = 0, nn0 = 0;
            if (more_tiles) coords(nvb, nm0, nn0);
            nm0 = __builtin_amdgcn_readfirstlane(nm0);      // wave-uniform: the next tile's staging pointers are formed where they are used
            nn0 = __builtin_amdgcn_readfirstlane(nn0);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int c = 0; c < 2; ++c)
#pragma unroll
                        for (int d = 0; d < 4; ++d) acc[a][b][c][d] = f32x4{0.f, 0.f, 0.f, 0.f};
            // Stagger: waves 4-7 (the SIMD partners of waves 0-3) run one barrier behind (gemm256_tile.h); balanced after the loop
            if (wm == 1) __builtin_amdgcn_s_barrier();

            // one K-tile = phases X and Y of gemm256_tile.h; `skip_wait`: K-tile 0 of a later tile (landed under the previous epilogue)
            auto ktile = [&](int kt, const int BUF, bool skip_wait) {
                const char* buf = smem + BUF * BUF_BYTES;
                const bool more1 = kt + 1 < nk, more2 = kt + 2 < nk;
                // X
                read_B(fb0, buf + OFF_B0);
                read_B(fb1, buf + OFF_B1);
                __builtin_amdgcn_sched_barrier(0);
                read_A(buf + OFF_A0);
                if (more1) {
                    stage_A(m0, kt + 1, 1, BUF ^ 1);
                    if (!skip_wait) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                } else if (has_next) {
                    stage_A(nm0, 0, 1, BUF ^ 1);         // next tile's K-tile 0, half A1 -> slot 0 (this is K-tile nk-1 in slot 1)
                    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if constexpr (SIDE16) {      // last K-tile (slot 1): slot 0 is free -> the side tiles of epilogue passes 0 and 1
                        stage_side(m0, n0, 0);
                        stage_side(m0, n0, 1);
                    }
                }
                PHASE_SYNC_IN();
                mma_quadrant(acc, 0, 0, fb0);
                mma_quadrant(acc, 1, 0, fb1);
                PHASE_SYNC_OUT();
                // Y
                read_A(buf + OFF_A1);
                if (more2) {
                    stage_A(m0, kt + 2, 0, BUF); stage_B(n0, kt + 2, 0, BUF); stage_B(n0, kt + 2, 1, BUF);
                    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                } else if (more1) {
                    if (has_next) {                      // K-tile nk-2 (slot 0): next tile's K-tile 0 takes the slot's three early halves
                        stage_A(nm0, 0, 0, BUF); stage_B(nn0, 0, 0, BUF); stage_B(nn0, 0, 1, BUF);
                        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                    } else {
                        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                    }
                } else if (!has_next && !SIDE16) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                PHASE_SYNC_IN();
                mma_quadrant(acc, 1, 1, fb1);
                mma_quadrant(acc, 0, 1, fb0);
                PHASE_SYNC_OUT();
            };
#ifdef LPI_GEMM_STAMPS
            gs_t0 = GS_NOW();
            if (gs_tiles) gs_next += gs_t0 - gs_t2;
#endif
            ktile(0, 0, !first);
            ktile(1, 1, false);
            for (int kt = 2; kt < nk; kt += 2) {
                ktile(kt, 0, false);
                ktile(kt + 1, 1, false);
            }
            if (wm == 0) __builtin_amdgcn_s_barrier();
#ifdef LPI_GEMM_STAMPS
            gs_t1 = GS_NOW();
            gs_main += gs_t1 - gs_t0;
#endif

            // ---- epilogue: four passes of 64 rows through slot 1 (pass p: mh = p >> 1, mi in {2 (p & 1), 2 (p & 1) + 1}, both wm).
            // Staging row s = wm * 32 + (mi & 1) * 16 + lrow holds tile row mh * 128 + wm * 64 + mi * 16 + lrow as 64 16-byte chunks
            // (chunk c = output columns 4c .. 4c+3) at physical chunk c ^ (s & 7): the 8 lanes of a ds_write_b128 group (8 rows, one
            // column group) spread over 8 chunks = all 32 banks of a 128-byte span; a whole-row read is any permutation.
            // every epilogue address is derived from a lane / wave id laundered through an empty asm: they are invariant across the tile
            // loop, and hoisted out of it they stayed live through the main loop — 38 spilled VGPRs whose scratch reloads (vector-memory
            // operations, counted in vmcnt) drained the LDS-DMA queue in every phase
            int lane_e = lane, wave_e = wave;
            asm volatile("" : "+v"(lane_e));
            asm volatile("" : "+s"(wave_e));
            const int lrow = lane_e & 15, lslot = lane_e >> 4;
            const int wm_e = wave_e >> 2, wn_e = wave_e & 3;
            if constexpr (P16) {
                // four passes of 64 rows x 512 B through the two halves of slot 1, alternating (see EPI_PLAIN16).  Staging row s = wm 32 + mi2 16 + lrow;
                // its 64 8-byte chunks (chunk c8 = output columns 4 c8 .. 4 c8 + 3) sit at c8 ^ 2 (s & 15): the 16 lanes of a ds_write_b64 group (16 rows, one
                // chunk) spread over 16 bank pairs, a row's 16-byte chunks stay whole (the XOR is even) and a read group's 16 chunks cover all banks once
                typedef __attribute__((ext_vector_type(4))) unsigned u32x4_;
                const int l5 = lane_e & 31, lh = lane_e >> 5;
                TC* const cbase = C + (size_t)m0 * ldc + n0 + l5 * 8;
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const int mh = p >> 1;
                    char* const sb = smem + STG + (p & 1) * 32768;
#pragma unroll
                    for (int mi2 = 0; mi2 < 2; ++mi2) {
                        const int s_row = wm_e * 32 + mi2 * 16 + lrow;
#pragma unroll
                        for (int nh = 0; nh < 2; ++nh)
#pragma unroll
                            for (int ni = 0; ni < 2; ++ni) {
                                const int c8 = nh * 32 + wn_e * 8 + ni * 4 + lslot;
                                const f32x4 a = acc[nh][ni][mh][(p & 1) * 2 + mi2] + 0.0f;
                                *reinterpret_cast<uint2*>(sb + s_row * 512 + ((c8 ^ (2 * lrow)) << 3)) = uint2{pack2_t<TC>(a[0], a[1]), pack2_t<TC>(a[2], a[3])};
                            }
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int i = 0; i < 4; ++i) {      // two rows per wave instruction: lanes 0-31 / 32-63, 16 bytes each
                        const int s_row = wave_e * 8 + 2 * i + lh;
                        const int trow = mh * 128 + (s_row >> 5) * 64 + ((p & 1) * 2 + ((s_row >> 4) & 1)) * 16 + (s_row & 15);
                        const u32x4_ v = *reinterpret_cast<const u32x4_*>(sb + s_row * 512 + (((2 * l5) ^ (2 * (s_row & 15))) << 3));
                        if constexpr (LPI_NTC_DEFAULT) __builtin_nontemporal_store(v, reinterpret_cast<u32x4_*>(cbase + (size_t)trow * ldc));
                        else *reinterpret_cast<u32x4_*>(cbase + (size_t)trow * ldc) = v;
                    }
                    // after two passes (8 stores of this wave since then) the next tile's K-tile 0, issued before them, must have landed
                    if (p == 1 && has_next) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                }
            } else {
            const int ecol = n0 + lane_e * 4;
            f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (SIDE16) bv = *reinterpret_cast<const f32x4*>(bias_lds + bias_off + ecol);
            else if (bias) bv = *reinterpret_cast<const f32x4*>(bias + ecol);
            f32x4 c1v = f32x4{0.f, 0.f, 0.f, 0.f};      // LayerNorm-fold epilogues: c1 sits behind the two statistics vectors
            if constexpr (EPI == LPI_EPI_LN || EPI == LPI_EPI_LN_QUICKGELU) c1v = *reinterpret_cast<const f32x4*>(residual + 2 * (size_t)ldr + ecol);
            char* const stg = smem + STG;
            // (eight double-buffered passes of 32 rows for the store-only epilogues with arithmetic were built and measured slower on the whole step: 22.78-22.82 ms
            // against 22.66-22.71 with these four — twice the barriers for half the work per pass; profiles/r04_experiments.md)
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int mh = p >> 1;
#ifdef LPI_GEMM_STAMPS
                gs_a = GS_NOW();
#endif
                if (p) {          // the previous pass's staging reads are done before this pass overwrites them
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                }
#ifdef LPI_GEMM_STAMPS
                gs_b = GS_NOW(); gs_e[0] += gs_b - gs_a;
#endif
                if constexpr (SIDE16) {      // pass p+1's side tile -> the half pass p-1 has just finished reading (passes 0, 1: main loop)
                    if (p >= 1 && p <= 2) stage_side(m0, n0, p + 1);
                    // pass 2 was the last reader of half 0 = the A0 | A1 halves of ring slot 0: the next tile's K-tile 0 A rows (the ones
                    // that come from HBM; the B rows are weights and sit in L2) get a pass of head start
                    if (p == 3 && more_tiles) { stage_A(nm0, 0, 0, 0); stage_A(nm0, 0, 1, 0); }
                }
                // LayerNorm-fold epilogues: mean and 1/std of this wave's 8 rows of the pass (consecutive rows) by two scalar loads, issued
                // here and waited for with the staging writes below.  (Left to the compiler they were per-row VECTOR loads of a uniform
                // address, each followed by vmcnt(0): a memory round trip per row that also drains the LDS-DMA queue.)
                f32x8 mu8, rs8;
                if constexpr (LNE) {
                    const int sr0 = wave_e * 8;
                    const int trow0 = mh * 128 + (sr0 >> 5) * 64 + ((p & 1) * 2 + ((sr0 >> 4) & 1)) * 16 + (sr0 & 15);
                    const float* pm = residual + (m0 + trow0);
                    const float* pr = pm + ldr;
                    asm volatile("s_load_dwordx8 %0, %2, 0x0\n\ts_load_dwordx8 %1, %3, 0x0" : "=&s"(mu8), "=&s"(rs8) : "s"(pm), "s"(pr));
                }
#pragma unroll
                for (int mi2 = 0; mi2 < 2; ++mi2) {
                    const int s_row = wm_e * 32 + mi2 * 16 + lrow;
#pragma unroll
                    for (int nh = 0; nh < 2; ++nh)
#pragma unroll
                        for (int ni = 0; ni < 2; ++ni) {
                            const int chunk = nh * 32 + wn_e * 8 + ni * 4 + lslot;
                            *reinterpret_cast<f32x4*>(stg + s_row * 1024 + ((chunk ^ (s_row & 7)) << 4)) = acc[nh][ni][mh][(p & 1) * 2 + mi2];
                        }
                }
                if constexpr (SIDE16) {
                    // this wave's pieces of pass p's side tile have landed once at most the YOUNGER operations are outstanding: the 4 DMA
                    // instructions of the next side tile and the 8 stores of the previous pass, in issue order
                    //   side(0) side(1) | stores(0) side(2) | stores(1) side(3) | stores(2) | stores(3)
                    if (p == 0) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                    else if (p == 3 && !more_tiles) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NST) : "memory");
                    else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NST + 4) : "memory");      // p == 3 with a next tile: stores(2) + its 4 A-row DMA instructions
                }
                if constexpr (LNE) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(mu8), "+s"(rs8) : : "memory");      // the statistics are used after this wait
                else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifdef LPI_GEMM_STAMPS
                gs_a = GS_NOW(); gs_e[1] += gs_a - gs_b;
#endif
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
#ifdef LPI_GEMM_STAMPS
                gs_b = GS_NOW(); gs_e[2] += gs_b - gs_a;
#endif
                float st_s[8], st_q[8];      // STATS: this lane's partial sums of the pass's eight rows
#pragma unroll
                for (int rr = 0; rr < 8; ++rr) {
                    const int s_row = wave_e * 8 + rr;
                    const int trow = mh * 128 + (s_row >> 5) * 64 + ((p & 1) * 2 + ((s_row >> 4) & 1)) * 16 + (s_row & 15);
                    const f32x4 v = *reinterpret_cast<const f32x4*>(stg + s_row * 1024 + ((lane_e ^ (s_row & 7)) << 4));
                    if constexpr (SIDE16) {
                        // the arithmetic of gemm_epilogue_store, with the residual / u row read from LDS
                        f32x4 o = v * alpha + bv;
                        const char* sp = smem + (p & 1) * 32768 + s_row * 512 + lane_e * 8;
                        if constexpr (EPI == LPI_EPI_DQUICKGELU) {
                            const f32x4 u = Elem<TA>::ld4(reinterpret_cast<const TA*>(sp));
                            o *= u;      // the side tile holds gelu'(u) (gemm_epilogue.h)
                        } else {
                            o += Elem<TC>::ld4(reinterpret_cast<const TC*>(sp));
                        }
                        // streaming store, as gemm_epilogue_store (a RUN-TIME choice per problem was tried: the duplicated store
                        // loops spilled 60 bytes per lane in the QuickGELU instantiations and the step lost 7 %)
                        if constexpr (STATS) {      // the two packed dwords serve the store and the sums (of the values as stored)
                            const uint32_t w0 = pack2_t<f16_t>(o[0], o[1]), w1 = pack2_t<f16_t>(o[2], o[3]);
                            *reinterpret_cast<uint2*>(C + (size_t)(m0 + trow) * ldc + ecol) = uint2{w0, w1};
                            f16x4_sum_sumsq(w0, w1, st_s[rr], st_q[rr]);
                        } else {
                            // The fp16 residual stream (84 MB for the vision tower at 256 pairs) is stored with PLAIN stores: the next GEMM reads it
                            // as its A operand and finds it in the Infinity Cache.  Round 2 measured no difference (24.40 / 24.46 against 24.42 / 24.39 ms)
                            // because a statistics pass over the stream sat between the two GEMMs and warmed the cache anyway; with the statistics
                            // coming out of this epilogue (LPI_EPI_RES_ROWSTATS) the pass is gone and the policy decides: 22.81 against 22.95-23.00 ms
                            // per step (profiles/r03_experiments.md).  d c_proj x gelu' (335 MB, read by one GEMM) stays a streaming store.
                            // -DLPI_NT_SIDE16C = streaming stores here too (A/B).
                            if constexpr (LPI_NTC_DEFAULT && EPI == LPI_EPI_DQUICKGELU) st4_nt<TC>(C + (size_t)(m0 + trow) * ldc + ecol, o);
                            else Elem<TC>::st4(C + (size_t)(m0 + trow) * ldc + ecol, o);
                        }
                    } else {
                        gemm_epilogue_store<T, TC, EPI, RES, SAVE_U>(v, m0 + trow, ecol, C, ldc, bv, alpha, residual, ldr, aux, ldaux, c1v, LNE ? mu8[rr] : 0.f,
                                                                     LNE ? rs8[rr] : 1.f);
                    }
                }
                if constexpr (STATS) {
                    // lanes 0-31 hold columns n0 .. n0+127 (slot n0/128), lanes 32-63 the next slot; after the halving exchange every lane has the
                    // sums of row (lane >> 2) & 7 of the wave's eight: one lane of each quad stores them
                    float so, qo;
                    rowstats8_half_reduce(st_s, st_q, lane_e, so, qo);
                    const int s_row = wave_e * 8 + ((lane_e >> 2) & 7);
                    const int trow = mh * 128 + (s_row >> 5) * 64 + ((p & 1) * 2 + ((s_row >> 4) & 1)) * 16 + (s_row & 15);
                    float* sp = reinterpret_cast<float*>(aux) + (size_t)(2 * ((n0 >> 7) + (lane_e >> 5))) * ldaux + (m0 + trow);
                    if ((lane_e & 3) == 0) { sp[0] = so; sp[ldaux] = qo; }
                }
#ifdef LPI_GEMM_STAMPS
                gs_a = GS_NOW(); gs_e[3] += gs_a - gs_b;
#endif
                // after two passes (>= 16 vector-memory instructions of this wave since then) the next tile's K-tile 0 must have landed:
                // all but the 16 youngest operations done.  Passes 2 and 3's barriers then publish it to every wave.
                if (p == 1 && has_next) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                if constexpr (LATE && !SIDE16) {
                    if (p == 1 && more_tiles) {      // slot 0 is free since the main loop ended; passes 2 and 3 run behind these
                        stage_A(nm0, 0, 0, 0); stage_B(nn0, 0, 0, 0); stage_B(nn0, 0, 1, 0); stage_A(nm0, 0, 1, 0);
                    }
                    // passes 2 and 3 issued >= 32 vector-memory operations after them: all but the 16 youngest done => K-tile 0 landed
                    if (p == 3 && more_tiles) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                }
            }
            }      // generic epilogue (else of P16)
#ifdef LPI_GEMM_STAMPS
            gs_t2 = GS_NOW();
            gs_epi += gs_t2 - gs_t1;
            ++gs_tiles;
#endif
            if (!more_tiles) { vbv = nvb + vb0; break; }
            // the staging slot is free again once every wave has read its rows: K-tile 1's three early halves of the next tile
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            vb = nvb; m0 = nm0; n0 = nn0;
            if constexpr (SIDE16) {      // both slots were the epilogue's: the rest of a full prologue (K-tile 0's A halves are in flight)
                stage_B(n0, 0, 0, 0); stage_B(n0, 0, 1, 0);
                stage_A(m0, 1, 0, 1); stage_B(n0, 1, 0, 1); stage_B(n0, 1, 1, 1);
                asm volatile("s_waitcnt vmcnt(6)" ::: "memory");      // all of K-tile 0 (and the epilogue's stores, older) landed
                __builtin_amdgcn_s_barrier();
            } else {
                stage_A(m0, 1, 0, 1); stage_B(n0, 1, 0, 1); stage_B(n0, 1, 1, 1);
                first = false;
            }
        }
    }
#undef PHASE_SYNC_IN
#undef PHASE_SYNC_OUT

#ifdef LPI_GEMM_STAMPS
    if (tid == 0 && blockIdx.x < 1024) {
        g_gemm_stamps[blockIdx.x][0] = gs_main; g_gemm_stamps[blockIdx.x][1] = gs_epi; g_gemm_stamps[blockIdx.x][2] = gs_next; g_gemm_stamps[blockIdx.x][3] = gs_tiles;
        for (int i = 0; i < 4; ++i) g_gemm_stamps[blockIdx.x][4 + i] = gs_e[i];
    }
#endif
    // ---- hybrid short last round: the leftover virtual ids [n_full, ...) as two 256x128 half tiles each (mapping of gemm256_tail_kernel:
    // work item j -> leftover id k = (j >> 4) * 8 + (j & 7), half (j >> 3) & 1, so both halves of a tile stay on one XCD)
    if (grp.tail_blocks > 0) {
        for (int j = blockIdx.x; j < grp.tail_blocks; j += G) {
            const int k = (j >> 4) * 8 + (j & 7), half = (j >> 3) & 1;
            const int v = n_full + k;
            const int pi = (grp.nprob > 1 && v >= grp.p[1].vb0) ? 1 : 0;
            bind(pi);
            const int lv = v - vb0;
            if (lv < 0 || lv >= nwg) continue;          // padding between the problems / past the last tile
            __syncthreads();      // LDS hand-over between tile bodies
            int m0, n0;
            coords(lv, m0, n0);
            t128::tile<T, TC, EPIX, RES, SAVE_U>(m0, n0 + half * 128, K, A, lda, B, ldb, C, ldc, bias, residual, ldr, aux, ldaux, alpha, smem);
        }
    }
}

int cu_count_p() { return lpi_cu_count(); }

// ---- host side: one launch for a group of 1 or 2 problems (same T / TC / epilogue kind)
struct HostProb {
    int M, N, K;
    const void* A; int lda; const void* B; int ldb; void* C; int ldc; const float* bias; const float* residual; int ldr; void* aux; int ldaux;
};

template <typename T, typename TC, int EPI, bool RES, bool SAVE_U>
int launchp_impl(const HostProb* hp, int np, float alpha, hipStream_t s)
{
    constexpr bool SIDE16 = (RES && sizeof(TC) == 2) || EPI == LPI_EPI_DQUICKGELU;
    static_assert(EPI != EPI_PLAIN16 || (!RES && !SAVE_U && sizeof(TC) == 2), "EPI_PLAIN16: store-only, 2-byte output");
    const int ncu = cu_count_p();
    PGroup g = {};
    g.nprob = np;
    g.alpha = alpha;
    int v = 0, nsum = 0;
    for (int i = 0; i < np; ++i) {
        PProb& P = g.p[i];
        const HostProb& h = hp[i];
        P.A = h.A; P.B = h.B; P.C = h.C; P.bias = h.bias; P.residual = h.residual; P.aux = h.aux;
        P.N = h.N; P.K = h.K; P.lda = h.lda; P.ldb = h.ldb; P.ldc = h.ldc; P.ldr = h.ldr; P.ldaux = h.ldaux;
        P.tiles_m = h.M / T256; P.tiles_n = h.N / T256;
        // A problem's ids follow the previous problem's directly.  (They used to start at a multiple of 8 "so that id & 7 stays the XCD"; that is
        // not needed: coords() gives the ids that share (id - vb0) & 7 a contiguous run of tiles, and those ids share an XCD whatever vb0 is — the
        // XCDs are merely rotated by vb0 & 7.  The padding ids counted as tiles: the grouped in_proj launch of ViT-B/16 at 256 pairs had 2 178 ids =
        // 8 rounds + 130, two more than the hybrid half-tile round takes, so its last round ran 130 full tiles on 256 CUs; 2 175 ids end in a round of
        // 254 half tiles instead.)  LPI_GROUP_PAD8 keeps the padding (A/B switch).
        P.vb0 = v;
        v += P.tiles_m * P.tiles_n;
        P.bias_off = nsum;
        // Tile order inside an XCD's share.  1 = columns fastest: the N-tiles of a row panel run together, so its A rows are fetched into the
        // XCD's 4 MB L2 once and only the weight panels have to stay resident from round to round — right while the WHOLE weight matrix fits
        // (ViT-B/16: 1.2-4.7 MB; whole step 24.16 -> 24.05 ms against 8).  Larger weights (ViT-L/14: 6-8 MB) would be re-read every round:
        // there 8 row panels at a time, rows fastest, is better (99.7 against 100.7 ms per step).  Tuning key 4 > 0 overrides.
        P.group_m = g_lpi_tuning[4] > 0 ? g_lpi_tuning[4] : ((size_t)h.N * h.K * sizeof(T) <= ((size_t)5 << 20) ? 1 : 8);
        // Round 5 — weight SLICES.  With columns fastest every round of an XCD (32 tiles = 2.7 row panels x all N-tiles) touches the WHOLE weight matrix,
        // and a 3.5-4.7 MB matrix does not survive a round in a 4 MB L2 next to the A panels and the C stream: rocprofv3 FETCH_SIZE showed the wide-N
        // GEMMs fetching 3.8x (in_proj) / 5.5x (c_fc) / 1.9x (d c_proj) their algorithmic bytes = the weights re-read by 8 XCDs x 7-10 rounds
        // (profiles/r03_gemm_shapes_pmc.json).  Ordering the tiles slice-major — the N-tiles cut into S slices whose weights (<= ~2.4 MB) DO stay resident,
        // every XCD walking one slice after the other down the row panels — trades that for reading the A panels once per slice (S x 84 MB instead of
        // 80 x 4.7 MB).  Measured (tools/shape_pmc_r05.sh, bytes fetched from beyond L2 per launch): c_fc 489 -> 362 MB, d c_proj 817 -> 719 MB, in_proj
        // with three slices 331 -> 341 MB (hence two slices only); -1.0 % / -0.4 % on the two GEMMs alone (tools/slice_ab.py), nothing measurable on the
        // whole step (22.75 ms either way, four interleaved pairs).  The chip is power-limited under these kernels (1 371 W of a 1 400 W cap over the whole step, tools/power_poll.py) and bytes from
        // beyond L2 are the first thing that costs clock (cdna_hip_programming.md rule 28).  Tuning key 15: 0 = automatic (weights above 3 MB with
        // an even number of N-tiles: two slices), 2 / 3 = that many slices where N divides, -1 = off.  Same tiles, same bits: only the order changes.
        if (slices_ok<EPI, SAVE_U>() && P.group_m == 1 && g_lpi_tuning[15] >= 0 && P.tiles_m < 32768) {
            int S = 0;
            const size_t wbytes = (size_t)h.N * h.K * sizeof(T);
            if (g_lpi_tuning[15] > 0) S = g_lpi_tuning[15];
            else if (wbytes > ((size_t)3 << 20) && P.tiles_n % 2 == 0) S = 2;      // three slices (in_proj: 9 N-tiles) measured no fewer fetched bytes
            if ((S == 2 || S == 3) && P.tiles_n % S == 0 && P.tiles_n / S >= 2) {
                P.group_m |= P.tiles_m << 16;      // the real row-panel count
                P.tiles_n /= S;
                P.tiles_m *= S;
            }
        }
        nsum += h.N;
    }
    if (SIDE16 && nsum > 8192) return LPI_ENOSYS;       // the bias vectors must fit behind the ring (the caller falls back to the one-tile kernel)
    // hybrid short last round as in gemm256.hip: rem <= ncu/2 leftover ids run as half tiles (tuning key 6)
    const int rem = v % ncu;
    g.n_full = v;
    if (g_lpi_tuning[6] != 0 && v - rem >= ncu && rem > 0 && rem <= ncu / 2 && (ncu % 8) == 0) {
        g.n_full = v - rem;
        g.tail_blocks = 16 * ((rem + 7) >> 3);
    }
    auto kern = gemm256p_kernel<T, TC, EPI, RES, SAVE_U>;
    const int LDS = std::max<int>(t128::LDS_BYTES, LDS_P + (SIDE16 ? nsum * 4 : 0));
    static LdsOnce once;
    if (int e = lpi_ensure_lds(once, (const void*)kern, 160 * 1024)) return e;
    lpi_note_gemm_kernel(g.tail_blocks ? LPI_GEMM_K_256_TAIL : LPI_GEMM_K_256);
    const int grid = std::min(ncu, std::max(g.n_full, g.tail_blocks));
    LPI_LAUNCH(kern, dim3(grid), dim3(NTHR), LDS, s, g);
    LPI_CHECK_LAST();
    return 0;
}

// run-time pointer presence -> compile-time epilogue flags; every problem of a group must agree on them
template <typename T, typename TC>
int dispatchp(int epi, const HostProb* hp, int np, float alpha, hipStream_t s)
{
    const bool res = hp[0].residual != nullptr, ax = hp[0].aux != nullptr;
    for (int i = 1; i < np; ++i)
        if ((hp[i].residual != nullptr) != res || (hp[i].aux != nullptr) != ax) return LPI_EINVAL;
    switch (epi) {
    case LPI_EPI_NONE:
        if (res) return launchp_impl<T, TC, LPI_EPI_NONE, true, false>(hp, np, alpha, s);
        if constexpr (sizeof(TC) == 2) {      // nothing to add per column or row: the half-width staging (EPI_PLAIN16; tuning key 14 = 1: the generic epilogue, A/B)
            bool plain = alpha == 1.0f && !ax && g_lpi_tuning[14] != 1;      // key 14 = 1 (a TEST hook): the generic epilogue, the bit-for-bit reference of tests/test_round4_gpu.py
            for (int i = 0; i < np; ++i) plain = plain && hp[i].bias == nullptr;
            if (plain) return launchp_impl<T, TC, EPI_PLAIN16, false, false>(hp, np, alpha, s);
        }
        return launchp_impl<T, TC, LPI_EPI_NONE, false, false>(hp, np, alpha, s);
    case LPI_EPI_RES_ROWSTATS:
        if constexpr (sizeof(TC) == 2 && !__is_same(TC, bf16_t)) {
            if (!res || !ax) return LPI_EINVAL;
            return launchp_impl<T, TC, LPI_EPI_NONE, true, true>(hp, np, alpha, s);
        }
        return LPI_ENOSYS;
    case LPI_EPI_QUICKGELU:
        if (res) return LPI_ENOSYS;
        if (ax) return launchp_impl<T, TC, LPI_EPI_QUICKGELU, false, true>(hp, np, alpha, s);
        return launchp_impl<T, TC, LPI_EPI_QUICKGELU, false, false>(hp, np, alpha, s);
    case LPI_EPI_DQUICKGELU:
        if (res) return LPI_ENOSYS;
        if (!ax) return LPI_EINVAL;
        return launchp_impl<T, TC, LPI_EPI_DQUICKGELU, false, false>(hp, np, alpha, s);
    }
    return LPI_EINVAL;
}

// LayerNorm-fold epilogues (LPI_EPI_LN / LPI_EPI_LN_QUICKGELU): `residual` is the LN operand block, never a residual tile
template <typename T, typename TC>
int dispatchp_ln(int epi, const HostProb* hp, int np, float alpha, hipStream_t s)
{
    const bool ax = hp[0].aux != nullptr;
    for (int i = 0; i < np; ++i)
        if (!hp[i].residual || (hp[i].aux != nullptr) != ax) return LPI_EINVAL;
    if (epi == LPI_EPI_LN) {
        if (ax) return LPI_EINVAL;
        return launchp_impl<T, TC, LPI_EPI_LN, false, false>(hp, np, alpha, s);
    }
    if (ax) return launchp_impl<T, TC, LPI_EPI_LN_QUICKGELU, false, true>(hp, np, alpha, s);
    return launchp_impl<T, TC, LPI_EPI_LN_QUICKGELU, false, false>(hp, np, alpha, s);
}

int launch_group(int dtype, int c_dtype, int epilogue, const HostProb* hp, int np, float alpha, hipStream_t s)
{
    if (epilogue == LPI_EPI_LN || epilogue == LPI_EPI_LN_QUICKGELU) {
        // A = the fp16 residual stream (bf16 mode keeps it fp16 too), B = gamma o W in fp16; C in the mode's storage type
        if (dtype == LPI_F16 && c_dtype == LPI_BF16) return dispatchp_ln<f16_t, bf16_t>(epilogue, hp, np, alpha, s);
        if (dtype == LPI_F16 && c_dtype == LPI_F16) return dispatchp_ln<f16_t, f16_t>(epilogue, hp, np, alpha, s);
        if (dtype == LPI_BF16 && c_dtype == LPI_BF16) return dispatchp_ln<bf16_t, bf16_t>(epilogue, hp, np, alpha, s);
        return LPI_ENOSYS;
    }
    if (dtype == LPI_BF16 && c_dtype == LPI_BF16) return dispatchp<bf16_t, bf16_t>(epilogue, hp, np, alpha, s);
    if (dtype == LPI_BF16 && c_dtype == LPI_F32) return dispatchp<bf16_t, float>(epilogue, hp, np, alpha, s);
    if (dtype == LPI_BF16 && c_dtype == LPI_F16 && (epilogue == LPI_EPI_NONE || epilogue == LPI_EPI_RES_ROWSTATS) && hp[0].residual) {
        for (int i = 1; i < np; ++i) if (!hp[i].residual) return LPI_EINVAL;
        if (epilogue == LPI_EPI_RES_ROWSTATS) {
            for (int i = 0; i < np; ++i) if (!hp[i].aux) return LPI_EINVAL;
            return launchp_impl<bf16_t, f16_t, LPI_EPI_NONE, true, true>(hp, np, alpha, s);
        }
        return launchp_impl<bf16_t, f16_t, LPI_EPI_NONE, true, false>(hp, np, alpha, s);
    }
    if (dtype == LPI_F16 && c_dtype == LPI_F16) return dispatchp<f16_t, f16_t>(epilogue, hp, np, alpha, s);
    if (dtype == LPI_F16 && c_dtype == LPI_F32) return dispatchp<f16_t, float>(epilogue, hp, np, alpha, s);
    return LPI_ENOSYS;
}

}  // namespace

#ifdef LPI_GEMM_STAMPS
// diagnostic build: copy the last launch's stamps ([1024][4] u64: K-loop cycles, epilogue cycles, hand-over cycles, tiles) to the host
extern "C" int lpi_gemm_stamps_read(unsigned long long* dst) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_gemm_stamps), sizeof(unsigned long long) * 1024 * 8, 0, hipMemcpyDeviceToHost);
}
#endif

// bf16 / f16 operands only (the f32 path is MFMA-bound: its prologue share is small and its K-tile geometry differs)
int lpi_gemm256p_launch(int dtype, int c_dtype, int M, int N, int K, const void* A, int lda, const void* B, int ldb, void* C, int ldc,
                        const float* bias, const float* residual, int ldr, int epilogue, void* aux, int ldaux, float alpha, hipStream_t s)
{
    const HostProb hp = {M, N, K, A, lda, B, ldb, C, ldc, bias, residual, ldr, aux, ldaux};
    return launch_group(dtype, c_dtype, epilogue, &hp, 1, alpha, s);
}

// Two GEMMs of the same operand types and epilogue kind in ONE persistent launch (lpi_gemm_nt_grouped); both already validated and
// eligible for the 256x256 kernel.  LPI_ENOSYS: combination not built (the caller issues them one after the other).
int lpi_gemm256p_launch2(int dtype, int c_dtype, int epilogue, float alpha, const lpi_gemm_desc* d, hipStream_t s)
{
    HostProb hp[2];
    for (int i = 0; i < 2; ++i)
        hp[i] = HostProb{d[i].M, d[i].N, d[i].K, d[i].A, d[i].lda, d[i].B, d[i].ldb, d[i].C, d[i].ldc, d[i].bias, (const float*)d[i].residual, d[i].ldr,
                         d[i].aux, d[i].ldaux};
    return launch_group(dtype, c_dtype, epilogue, hp, 2, alpha, s);
}
