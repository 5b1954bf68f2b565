// Prompted multi-head attention backward, bf16 operands, fourth generation: ONE pass over the scores, 8 waves, query slices STREAMED.
//
// replaces: the autograd backward of nn.MultiheadAttention as called from ResidualAttentionBlock.attention
// (retrieval/models/clip/model.py:183-185) for the vision tower (non-causal; ViT-B/16: L = 1 + P + 196 = 213, ViT-L/14: L = 273).
//
// Why ("what bounds attention"): the two-phase persistent backward of round 2 evaluated S, P, dP and dS twice (56 MFMAs and two
// exponentials per score), kept a whole head's Q, K, V, dO (and O) in LDS and overlapped neither its loads with its arithmetic nor one wave's vector
// work with another's matrix work.  Here:
//   * a persistent workgroup per CU = 8 waves (two per SIMD, <= 248 VGPRs so that MFMA results land in VGPRs); wave w OWNS one or two 16-key "units"
//     of a head (14 units at Lp = 224: 2,2,2,2,2,2,1,1): their K and V rows (MFMA B operands, V pre-scaled by 1/8) and their dK^T, dV^T
//     accumulators stay in registers for the whole head;
//   * the head's queries stream through LDS in SLICES of 32 rows (Q, dO, O: 12 KiB) from a 6-slot ring filled by LDS-DMA four slices ahead,
//     across head boundaries — every global read is an LDS-DMA (scalar head base + 32-bit lane offset) issued long before its use; K / V of the
//     NEXT head arrive in their images spread over the head's first iterations;
//   * per slice: delta = rowsum(dO o O) (waves 0-3, eight lanes per row, kept in LDS); S^T = Q K_own^T and dP' = dO (V/8)^T - delta/8 (the
//     accumulators start from -delta/8) with the key on the lane; P = exp2(c S - lse'), dS = P dP' are, packed, the B operands of dV^T += dO^T P
//     and dK^T += Q^T dS; dS^T crosses LDS once ([key][query] bf16, double-buffered) and each wave contracts it over ALL keys with the K^T
//     fragments of its 16 head-dim columns: a piece of dQ^T is complete in one MFMA chain — no partial sums across waves, no atomics;
//   * one barrier per slice; waves 4-7 run a slice's matrix half (dV, dK) one iteration late, in front of the next slice's score half (stagger:
//     SIMD partners are in complementary phases); dQ pieces 1,1,3,3 over the SIMDs, DMA issue and the delta pass on the waves with barrier slack;
//   * vmcnt is ONE in-order counter over loads, stores and DMA: the waits name exactly the operations that may stay in flight.
// 40 MFMAs and one exponential per score instead of 56 and two.  Results are bitwise reproducible (fixed summation orders).
//
// Round 4 — sequences of 225 .. 288 tokens (ViT-L/14: 273 = 18 key units, two more than 8 waves x 2; three units per wave spill): the head's keys
// run as TWO launches over all queries (template WIN, Args4): keys 0 .. 223 on the tuned Lp = 224 configuration, which also leaves -delta/8 in the
// C ABI's delta scratch; then keys 224 .. L-1 on the generic configuration, which reads that vector instead of the O rows and ADDS its dQ share to
// the stored one — the stored dQ rows of a slice arrive by LDS-DMA as a fourth piece of the slice's ring slot.  lse and delta are per query, so the
// two windows are independent; dK / dV of a window are complete; two addends per dQ element in a fixed order: still bitwise reproducible.
#include <type_traits>
#include "common.h"

extern int g_lpi_tuning[16];

namespace {

typedef bf16_t T;
constexpr int HD = 64;
constexpr int RB = 128;          // bytes per LDS image row (64 x 2 B, unpadded: what LDS-DMA writes)
constexpr int KS = 2;            // 32-wide k-steps per 64-element row
constexpr int NWV = 8;           // waves per workgroup (two per SIMD)
constexpr int MAXU = 2;          // 16-key units per wave
constexpr int MAXKB = 7;         // 32-key blocks per head (Lp <= 224)
constexpr int NSLOT = 6;         // ring slots
constexpr int SLOT_BYTES = 3 * 32 * RB;      // Q slice + dO slice + O slice
constexpr int AHEAD = 4;         // slices in flight ahead of the slice being computed (a slot stays in use one iteration longer: STAG)
constexpr float LOG2E = 1.4426950408889634f;
constexpr float SCALE = 0.125f;

typedef __attribute__((ext_vector_type(2))) float f32x2_;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_;
__device__ __forceinline__ uint32_t pack2(float a, float b) { return __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2_){a, b}, bf16x2_)); }
__device__ __forceinline__ float exp2_fast(float x) { return __builtin_amdgcn_exp2f(x); }

// one LDS-DMA instruction: 64 lanes x 16 B -> 1 KiB of LDS starting at lds_addr (wave-uniform), lane i at +16 i
__device__ __forceinline__ void glds16(const void* src, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(__builtin_amdgcn_readfirstlane(lds_addr)) : "memory");
}
// 64 lanes x 4 B -> 256 B of LDS, lane i at +4 i
__device__ __forceinline__ void glds4(const void* src, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(__builtin_amdgcn_readfirstlane(lds_addr)) : "memory");
}

// the same with the address as a scalar base + a 32-bit per-lane byte offset: the base of a head is scalar arithmetic once per head, the offset two
// vector instructions per piece (per-lane 64-bit pointers cost a dozen vector instructions per piece: a sixth of an iteration on the issuing waves)
__device__ __forceinline__ void glds16s(const void* sbase, unsigned voff, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(__builtin_amdgcn_readfirstlane(lds_addr)) : "memory");
}
__device__ __forceinline__ void glds4s(const void* sbase, unsigned voff, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(__builtin_amdgcn_readfirstlane(lds_addr)) : "memory");
}
// a pointer as a wave-uniform (scalar) value
__device__ __forceinline__ const void* scalar_ptr(const void* p) {
    const unsigned long long a = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return (const void*)(((unsigned long long)hi << 32) | lo);
}

typedef __attribute__((address_space(3))) short4v* lds_s4p;
__device__ __forceinline__ uint2 tr_read(const char* p) {
    return __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4p)p));
}
// two transposing reads 16 rows apart -> one 16x16x32 operand chunk (k elements 4g + 0..3 of the first block, then of the second)
__device__ __forceinline__ Chunk tr_pair(const char* p, int second) {
    const uint2 lo = tr_read(p), hi = tr_read(p + second);
    Chunk c;
    c.u = make_uint4(lo.x, lo.y, hi.x, hi.y);
    return c;
}
// 8 bf16 values times a power of two (exact)
__device__ __forceinline__ void chunk_scale_bf16(Chunk& c, float f) {
    Chunk o;
    o.u = make_uint4(pack2((float)c.h[0] * f, (float)c.h[1] * f), pack2((float)c.h[2] * f, (float)c.h[3] * f), pack2((float)c.h[4] * f, (float)c.h[5] * f),
                     pack2((float)c.h[6] * f, (float)c.h[7] * f));
    c = o;
}
__device__ __forceinline__ void mma(f32x4& acc, const Chunk& a, const Chunk& b) { acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.h, b.h, acc, 0, 0, 0); }

#define LPI4_BARRIER()                                             \
    do {                                                           \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         \
        __builtin_amdgcn_s_barrier();                              \
        asm volatile("" ::: "memory");                             \
    } while (0)

// dS^T scratch: 64-byte rows (32 queries x bf16); the 8-byte slot index is XOR-ed with (row >> 2 & 1) << 2 | (row >> 3 & 1) << 1 | (row >> 1 & 1).
// Bits 2 and 1 keep the transposing reads conflict-free (2 x 32 lanes on 64 banks: rows r and r + 4 of a half land in different slot groups); bit 0
// (round 4) does the same for the STORES, which the LDS serves as 4 x 16 contiguous lanes on 32 banks: the 8 rows of one parity that a 16-lane group
// writes then use 8 distinct slots — without it they shared 4 (two-way: 2.25 M extra LDS cycles per launch, profiles/r03_pmc.json; the model:
// tools/lds_swizzle_check.py, dS^T section)
__device__ __forceinline__ int ds_sw(int row) { return (((row >> 2) & 1) << 2) | (((row >> 3) & 1) << 1) | ((row >> 1) & 1); }

struct Args4 {
    int L, Lp, H, total;
    const T* qkv; int ldqkv;
    const T* ctx; int ldctx;
    const T* dctx; int lddctx;
    const float* lse;
    float* delta;
    T* dqkv; int lddqkv;
    int rows_hi;      // only dQ / dK / dV of token rows < rows_hi are wanted (lpi_attn_bwd_prefix); >= L: all
    int flags;        // A/B switches (tuning key 12): 1 = K / V of the next head as one burst after the prologue
    // key WINDOW (round 4; template WIN): the launch covers keys kw0 .. kw0 + Lk - 1 of every head (Lkp = Lk rounded up to 32) against ALL L queries
    // (Lp = L rounded up to 32 = 32 x the query slices).  A sequence longer than 224 tokens (ViT-L/14: 273) runs as two launches — keys 0 .. 223 on
    // the Lp = 224 configuration, the rest (<= 64 keys) on the generic one with acc_dq: its dQ is ADDED to what the first launch stored.
    int kw0, Lk, Lkp, acc_dq;
    // LAYOUT (round 6, lpi_attn_bwd_layout): head stride and q -> k -> v stride (elements) of qkv and of dqkv, head stride of ctx and of dctx; the interleaved
    // default is hs = 64, vs = H 64; head-blocked planes [.][rows][64]: row stride 64, hs = rows 64, vs = H rows 64
    int q_hs, q_vs, dq_hs, dq_vs, c_hs, dc_hs;
};

// LDS map (byte offsets from the dynamic region; Lp <= 224: 161 792 B):
//   [0, Lp RB) K image | [Lp RB, 2 Lp RB) V image | ring: NSLOT x (Q slice, dO slice, O slice: 4 KiB each) | dS^T x 2 (Lp x 64 B each) |
//   delta[2][32] f32 | lse[2][Lp] f32
// NKB: the head's 32-key blocks at compile time (7: L in 193 .. 224, the vision towers' prompted and plain sequences), 0 = read Lp.
// DQN: pieces of dQ^T (16 head-dim columns x 16 queries of a slice, 7 MFMAs at Lp = 224) this wave computes per slice: 1 = piece
// (dq_dt, dq_qs), 2 = both query halves of head-dim block dq_dt, 0 = none.  At Lp = 224 the key units split 2,2,2,2,2,2,1,1 over the
// waves, i.e. 4,4,3,3 over the SIMDs (waves w and w + 4 share one): the eight dQ pieces go 1,1,3,3 so that every SIMD issues 69-71 MFMAs
// per slice — waves 0-3 one piece each, waves 4, 5 none, waves 6, 7 (one unit) two.
// STAG: this wave runs the matrix half of a slice (dV, dK) one iteration late, IN FRONT of the next slice's score half: the two waves of a
// SIMD (w and w + 4) run the same program between the same barriers, so unstaggered they want the matrix pipe together and the vector
// pipe together; with waves 4-7 half an iteration behind, one wave's MFMAs run beside the other's exponentials.
// NQS: the query slices at compile time (= NKB for the whole-sequence launches of Lp = 224), 0 = read A.Lp.  WIN: key-window launch (Args4): without it
// the window is the whole sequence and every window quantity folds into the old one (the same code as before the window existed).
// WIN: 0 = the whole sequence; 1 = first key window of a long sequence; 2 = a later window: its dQ share is added to what the first launch stored — the
// stored rows of a slice come in by LDS-DMA as a FOURTH piece of the slice's ring slot (waves 4, 5, which move no piece otherwise), are read there when the
// slice's dQ pieces are complete, and the sum leaves as a plain store (a no-return packed-bf16 atomic was tried first: its 8-byte segments in 16 rows per
// wave instruction ran the launch at 0.3 TB/s)
template <int NUW, bool SV16, int NKB, int DQN, bool STAG, int NQS = NKB, int WIN = 0>
__device__ __forceinline__ void bwd4_body(const Args4& A, char* smem, int wave, int ub, int dq_dt, int dq_qs) {
    const int lane = threadIdx.x & 63, g = lane >> 4, r16 = lane & 15;
    // Lp: padded KEY rows of the launch (images, units, dS^T rows); Lqp: padded query rows (slices, lse vector); L: the sequence (queries, row addressing)
    const int L = A.L, Lp = NKB ? 32 * NKB : (WIN ? A.Lkp : A.Lp), H = A.H, total = A.total;
    const int NSL = NQS ? NQS : (A.Lp >> 5);                       // query slices of a head
    const int NKBr = NKB ? NKB : (Lp >> 5);                        // 32-key blocks of the window
    const int Lqp = WIN ? 32 * NSL : Lp;
    const int kw0 = WIN ? A.kw0 : 0, Lk = WIN ? A.Lk : L;
    const T* const qkv = A.qkv; const int ldqkv = A.ldqkv;
    const T* const dctx = A.dctx; const int lddctx = A.lddctx;
    float* const delta = A.delta;
    T* const dqkv = A.dqkv; const int lddqkv = A.lddqkv;
    const int dm = A.q_vs;            // q -> k -> v stride of qkv (H * HD in the interleaved layout)
    const float c = SCALE * LOG2E;
    const int nheads = (total - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;      // heads of this workgroup
    const int nslices = nheads * NSL;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem);
    constexpr int SLOTB = (WIN == 2 ? 4 : 3) * 32 * RB;      // ring slot: Q, dO, O (and the stored dQ rows) of a slice
    const int o_ring = 2 * Lp * RB, o_dsb = o_ring + NSLOT * SLOTB, o_dl = o_dsb + 2 * Lp * 64, o_lse = o_dl + 2 * 32 * 4;      // lse: 2 x Lqp floats
    const int o_dlt = o_lse + 2 * Lqp * 4;                   // WIN == 2: -delta / 8 of the head's queries, 2 x Lqp floats (written by the first window's launch)
    char* const k_img = smem;
    char* const v_img = smem + Lp * RB;
    char* const ring = smem + o_ring;
    char* const dsb = smem + o_dsb;
    float* const dl_l = reinterpret_cast<float*>(smem + o_dl);
    float* const lse_l = reinterpret_cast<float*>(smem + o_lse);
    float* const dlt_l = reinterpret_cast<float*>(smem + o_dlt);
    // prefix mode: dQ of the slices that hold rows < rows_hi (every wave works on those: dQ sums over all keys), dK / dV of the units that
    // hold such rows (their waves work on every slice); delta for every row
    const int rows_hi = A.rows_hi;
    const bool own_wanted = NUW > 0 && kw0 + ub * 16 < rows_hi;
    // ablation builds (timing only, wrong results): a run-time condition that is never true keeps the code alive
    constexpr bool abl_comp = true;
    constexpr bool abl_dma = true;
#if defined(LPI_ABL4_STAMPS)
    const bool abl_st = true;
#else
    constexpr bool abl_st = true;
#endif
    // diagnostic build (tools/attn4_stamps.py): workgroup 0 writes s_memtime stamps of its phases into the delta scratch ([wave][n] u64,
    // 2048 per wave) instead of delta; never in the product build
#ifdef LPI_ABL4_STAMPS
    unsigned long long* const stamp_buf = reinterpret_cast<unsigned long long*>(A.delta) + wave * 2048;
    int stamp_n = 0;
    auto STAMP = [&]() {
        if (blockIdx.x == 0 && stamp_n < 2048) {
            const unsigned long long tk = __builtin_amdgcn_s_memtime();
            if (lane == 0) stamp_buf[stamp_n] = tk;
            ++stamp_n;
        }
    };
#define LPI4_STAMP() STAMP()
#else
#define LPI4_STAMP() do {} while (0)
#endif

    // per-lane byte offsets
    int rc[KS];                  // row fragment of a 16-row window: row (lane & 15), logical chunk (lane >> 4) + 4 ks, chunk index ^ (row & 6)
    int tr[4];                   // transposing read of a 16-row window, head-dim block dt
    int trw;                     // the same for dt = dq_dt (K^T fragments)
    int dsw[NUW > 0 ? NUW : 1][2];   // dS^T store: [key][query] rows of 64 B inside 32-key blocks of 2 KiB
    int dsr[DQN > 0 ? DQN : 1][2];   // dS^T transposing read of query half dq_qs (+ piece): [second 16 keys]
    int dlo;                     // delta pass (waves 0-3): 16 bytes of row 8 wave + (lane >> 3), chunk lane & 7
    {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) rc[ks] = r16 * RB + (((g + 4 * ks) ^ (r16 & 6)) << 4);
        const int rr = 4 * g + (r16 >> 2), p = lane & 3;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) tr[dt] = rr * RB + (((2 * dt + (p >> 1)) ^ (rr & 6)) << 4) + (p & 1) * 8;
        trw = rr * RB + (((2 * dq_dt + (p >> 1)) ^ (rr & 6)) << 4) + (p & 1) * 8;
#pragma unroll
        for (int u = 0; u < NUW; ++u) {
            const int ug = ub + u;
            const int rowb = 16 * (ug & 1) + r16;
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2) dsw[u][t2] = (ug >> 1) * 2048 + rowb * 64 + (((4 * t2 + g) ^ ds_sw(rowb)) << 3);
        }
#pragma unroll
        for (int pi = 0; pi < DQN; ++pi)
#pragma unroll
            for (int hi = 0; hi < 2; ++hi) {
                const int row = 16 * hi + 4 * g + (r16 >> 2);
                dsr[pi][hi] = row * 64 + (((4 * (dq_qs + pi) + (lane & 3)) ^ ds_sw(row)) << 3);
            }
        const int drow = 8 * (wave & 3) + (lane >> 3);
        dlo = drow * RB + (((lane & 7) ^ (drow & 6)) << 4);
    }

    // ---- LDS-DMA issue helpers (all wave-uniform control flow; lanes behind L are EXEC-masked or clamped)
    // A head cursor: (sample, head) of the workgroup's it-th head = blockIdx.x + it gridDim.x, advanced without divisions
    struct Head { int b, h; };
    const int G = (int)gridDim.x, Gb = G / H, Gh = G % H;
    auto next_head = [&](Head& x) {
        x.b += Gb; x.h += Gh;
        if (x.h >= H) { x.h -= H; ++x.b; }
    };
    // K and V rows of a head -> their images, in PARTS of 16 pieces of 8 rows (2 per wave): the images are free from a head's prologue
    // on and needed at the next one, so the parts go out over the first iterations of a head instead of as one 57 KB burst per CU beside
    // the dK / dV stores of the head before
    const int nblk = (Lk + 7) >> 3;
    const int kv_parts = (2 * nblk + 15) >> 4;
    // scalar base of a head's rows in a [B L, ld] matrix (+ a column offset)
    auto head_base = [&](const T* m, int ld, int hs, const Head& x, int col) {
        return (const T*)scalar_ptr(m + ((size_t)x.b * L * ld + (size_t)x.h * hs + col));
    };
    const T* kv_base = nullptr;           // K columns of the head whose K / V are being fetched
    auto issue_kv_part = [&](int part) {
        if (wave >= 6) return;                // waves 6, 7 move the O pieces of every slice instead
        const int r8 = lane >> 3, pc = lane & 7;
        const int np = wave < 4 ? 3 : 2, p0 = wave < 4 ? 3 * wave : 12 + 2 * (wave - 4);      // 16 pieces of a part: 3,3,3,3,2,2
        for (int j = 0; j < np; ++j) {
            const int pi = part * 16 + p0 + j;
            if (pi < 2 * nblk) {
                const int isv = pi >= nblk ? 1 : 0, blk = pi - isv * nblk;
                const int row = blk * 8 + r8;
                if (row < Lk && abl_dma) glds16s(kv_base, (unsigned)((kw0 + row) * ldqkv + isv * dm + ((pc ^ (row & 6)) << 3)) * 2u, lds0 + isv * Lp * RB + blk * 1024);
            }
        }
    };
    auto issue_lse = [&](const Head& x, int buf) {
        const int i = wave * 64 + lane;
        if (wave * 64 < L) {
            const float* lb = (const float*)scalar_ptr(A.lse + (size_t)(x.b * H + x.h) * L);
            if (i < L && abl_dma) glds4s(lb, (unsigned)i * 4u, lds0 + o_lse + (buf * Lqp + wave * 64) * 4);
        }
    };
    // WIN == 2: the head's -delta / 8 vector, which the first window's launch left in the C ABI's delta scratch ([B, H, L] f32): no O rows are read
    auto issue_dlt = [&](const Head& x, int buf) {
        if constexpr (WIN == 2) {
            const int i = wave * 64 + lane;
            if (wave * 64 < L) {
                const float* db = (const float*)scalar_ptr(delta + (size_t)(x.b * H + x.h) * L);
                if (i < L && abl_dma) glds4s(db, (unsigned)i * 4u, lds0 + o_dlt + (buf * Lqp + wave * 64) * 4);
            }
        }
    };
    // slice t of a head -> ring slot: 12 pieces of 8 rows (Q, dO, O x 4): wave w < 4 moves piece w of Q and of dO, waves 6, 7 two pieces of
    // O each, waves 4, 5 (the longest iteration: staggered, two key units) none (rows behind L: the last row again — finite values; their
    // lse is -inf, so P = 0 there)
    const bool mv_dq = WIN == 2 && wave >= 4 && wave < 6;      // WIN == 2: waves 4, 5 move the stored dQ rows of every slice (piece 3 of the slot)
    const T* const sl_m0 = wave < 4 ? qkv : (mv_dq ? (const T*)dqkv : A.ctx);             // this wave's first source matrix and row stride
    const int sl_ld0 = wave < 4 ? ldqkv : (mv_dq ? lddqkv : A.ldctx);
    const int sl_hs0 = wave < 4 ? A.q_hs : (mv_dq ? A.dq_hs : A.c_hs);
    const int sl_blk0 = wave < 4 ? wave : ((wave & 1) << 1);
    const unsigned sl_dst = lds0 + o_ring + (wave < 4 ? 0 : (mv_dq ? 3 : 2)) * 32 * RB + sl_blk0 * 1024;
    const int sl_rl = sl_blk0 * 8 + (lane >> 3);
    const int sl_ch = ((lane & 7) ^ (sl_rl & 6)) << 3;         // (row & 6) is the same for rows 8 apart
    const bool sl_mine = wave < 4 || (wave >= 6 && WIN != 2) || mv_dq;      // WIN == 2: nobody moves O (delta comes from the scratch)
    const T* sl_b0 = nullptr;             // scalar bases of the prefetch cursor's head: first matrix, and dO for waves 0-3
    const T* sl_b1 = nullptr;
    auto slice_bases = [&](const Head& x) {
        if (!sl_mine) return;
        sl_b0 = head_base(sl_m0, sl_ld0, sl_hs0, x, 0);
        if (wave < 4) sl_b1 = head_base(dctx, lddctx, A.dc_hs, x, 0);
    };
    auto issue_slice = [&](int t, int slot) {
        if (abl_dma && sl_mine) {
            const unsigned dst = sl_dst + slot * SLOTB;
            const int r0 = min(t * 32 + sl_rl, L - 1);
            glds16s(sl_b0, (unsigned)(r0 * sl_ld0 + sl_ch) * 2u, dst);
            if (wave < 4) glds16s(sl_b1, (unsigned)(r0 * lddctx + sl_ch) * 2u, dst + 32 * RB);
            else glds16s(sl_b0, (unsigned)(min(t * 32 + sl_rl + 8, L - 1) * sl_ld0 + sl_ch) * 2u, dst + 1024);
        }
    };
    // prefetch cursor over this workgroup's slice stream
    Head pf{(int)blockIdx.x / H, (int)blockIdx.x % H};
    int pf_t = 0, pf_gs = 0, pf_slot = 0;
    slice_bases(pf);
    auto issue_next = [&]() {
        if (pf_gs < nslices) {
            issue_slice(pf_t, pf_slot);
            ++pf_gs;
            if (++pf_slot == NSLOT) pf_slot = 0;
            if (++pf_t == NSL) { pf_t = 0; next_head(pf); slice_bases(pf); }
        }
    };

    // ---- per-head register state
    constexpr int NUA = NUW > 0 ? NUW : 1;
    Chunk kk[NUA][KS], vv[NUA][KS];      // own K, V rows (B operands of S^T, dP^T)
    Chunk kT[MAXKB];                     // K^T fragments of head-dim block dq_dt, all keys (A operands of dQ^T)
    f32x4 dk[NUA][4], dv[NUA][4];

    // delta of slice t's rows 8 wave .. 8 wave + 7 = rowsum(dO o O) -> dl[par] (/ 8) and the C ABI's delta scratch; waves 0-3 (they have the
    // slack: waves 4-7 run the longer iteration), 8 lanes of 8 elements per row
    Chunk dl_d, dl_o;
    float* dlt_g = nullptr;      // WIN == 1: this head's row of the delta scratch
    auto stage_delta_load = [&](int slot) {         // issued in front of the iteration's DMA: the LDS round trip runs under it
        if (wave >= 4 || WIN == 2) return;
        const char* ds_ = ring + slot * SLOTB + 32 * RB;
        dl_d.u = *reinterpret_cast<const uint4*>(ds_ + dlo);
        dl_o.u = *reinterpret_cast<const uint4*>(ds_ + 32 * RB + dlo);
    };
    auto stage_delta = [&](int par, int tq) {         // tq: the slice whose rows 8 wave .. 8 wave + 7 these are
        if (wave >= 4 || WIN == 2) return;
        float v = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) v = fmaf((float)dl_d.h[e], SV16 ? (float)dl_o.hh[e] : (float)dl_o.h[e], v);
        v += dpp_move<0xB1>(v);     // lane ^ 1
        v += dpp_move<0x4E>(v);     // lane ^ 2
        v += dpp_move<0x141>(v);    // the other quad of each 8 lanes
        if ((lane & 7) == 0) {
            dl_l[par * 32 + 8 * wave + (lane >> 3)] = -v * SCALE;          // the dP' accumulators start from it
            if constexpr (WIN == 1) {      // ... and the later windows' launches read it instead of the O rows
                const int q = tq * 32 + 8 * wave + (lane >> 3);
                if (q < L) dlt_g[q] = -v * SCALE;
            }
        }
    };

    Chunk bp[NUA], bs[NUA];              // P and dS of a slice as MFMA B operands: score half -> matrix half
    // score half of slice t: S^T, dP^T -> P, dS = P (dP - delta) / 8; dS^T -> dsb[par]
    auto stage_s = [&](int t, int slot, int lbuf, int par, bool with_delta) {
        const char* qs_ = ring + slot * SLOTB;
        const char* ds_ = qs_ + 32 * RB;
        char* dsp = dsb + par * (Lp * 64);
        uint32_t pw[NUA][2][2], sw[NUA][2][2];
        // every row-fragment read of the slice first (both 16-query halves: 8 x 16 B per lane in flight), so that the second half's
        // LDS round trip hides under the first half's arithmetic
        Chunk qa[2][KS], da[2][KS];
        f32x4 nl[2], dls[2];
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                qa[t2][ks].u = *reinterpret_cast<const uint4*>(qs_ + t2 * 16 * RB + rc[ks]);
                da[t2][ks].u = *reinterpret_cast<const uint4*>(ds_ + t2 * 16 * RB + rc[ks]);
            }
            nl[t2] = *reinterpret_cast<const f32x4*>(lse_l + lbuf * Lqp + t * 32 + t2 * 16 + 4 * g);      // -lse log2(e): scaled in place per head
            if constexpr (WIN == 2) dls[t2] = *reinterpret_cast<const f32x4*>(dlt_l + lbuf * Lqp + t * 32 + t2 * 16 + 4 * g);
            else dls[t2] = *reinterpret_cast<const f32x4*>(dl_l + par * 32 + t2 * 16 + 4 * g);                 // -delta / 8
        }
        // the NEXT slice's delta (waves 0-3; its rows were read in front of the DMA issue): a short dependent chain that runs under this
        // slice's LDS round trip and first MFMAs
        if (with_delta) stage_delta(par ^ 1, t + 1);
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
            if constexpr (SV16) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) chunk_f16_to_bf16(qa[t2][ks]);
            }
#pragma unroll
            for (int u = 0; u < NUW; ++u) {
                // dP' = dO (V / 8)^T - delta / 8: the own V rows carry the 1 / 8 (exact in bf16), the accumulators start from -delta / 8
                f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f}, dp = dls[t2];
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    mma(s, qa[t2][ks], kk[u][ks]);
                    mma(dp, da[t2][ks], vv[u][ks]);
                }
                f32x4 p, e;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    p[r] = exp2_fast(fmaf(s[r], c, nl[t2][r]));
                    e[r] = p[r] * dp[r];
                }
                pw[u][t2][0] = pack2(p[0], p[1]);
                pw[u][t2][1] = pack2(p[2], p[3]);
                sw[u][t2][0] = pack2(e[0], e[1]);
                sw[u][t2][1] = pack2(e[2], e[3]);
                *reinterpret_cast<uint2*>(dsp + dsw[u][t2]) = make_uint2(sw[u][t2][0], sw[u][t2][1]);
            }
        }
#pragma unroll
        for (int u = 0; u < NUW; ++u) {
            bp[u].u = make_uint4(pw[u][0][0], pw[u][0][1], pw[u][1][0], pw[u][1][1]);
            bs[u].u = make_uint4(sw[u][0][0], sw[u][0][1], sw[u][1][0], sw[u][1][1]);
        }
    };
    // matrix half of the slice in ring slot `slot`: dV^T += dO^T P, dK^T += Q^T dS
    auto stage_m = [&](int slot) {
        const char* qs_ = ring + slot * SLOTB;
        const char* ds_ = qs_ + 32 * RB;
        // the transposing reads of two head-dim blocks are in flight together
#pragma unroll
        for (int d0 = 0; d0 < 4; d0 += 2) {
            Chunk ado[2], aq[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                ado[j] = tr_pair(ds_ + tr[d0 + j], 16 * RB);
                aq[j] = tr_pair(qs_ + tr[d0 + j], 16 * RB);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if constexpr (SV16) chunk_f16_to_bf16(aq[j]);
#pragma unroll
                for (int u = 0; u < NUW; ++u) {
                    mma(dv[u][d0 + j], ado[j], bp[u]);
                    mma(dk[u][d0 + j], aq[j], bs[u]);
                }
            }
        }
    };

    // dQ^T[16 dq_dt .., 16 queries of half dq_qs (+ piece)] of slice t = K^T (all keys) . dS^T -> global
    auto stage_dq = [&](T* dqh, int t, int par, int slot_t) {      // slot_t: the ring slot of slice t (WIN == 2: its piece 3 holds the stored dQ rows)
        if constexpr (DQN > 0) {
            const char* dsp = dsb + par * (Lp * 64);
            f32x4 dq[DQN];
#pragma unroll
            for (int pi = 0; pi < DQN; ++pi) dq[pi] = f32x4{0.f, 0.f, 0.f, 0.f};
            // the transposing reads of a group of key blocks are in flight together (16 registers), then their MFMAs: one LDS round trip
            // per group instead of one per key block
            constexpr int GRP = DQN == 2 ? 2 : 4;
#pragma unroll
            for (int k0 = 0; k0 < (NKB ? NKB : MAXKB); k0 += GRP) {
                Chunk bq[GRP][DQN];
#pragma unroll
                for (int j = 0; j < GRP; ++j)
                    if (k0 + j < (NKB ? NKB : MAXKB) && (NKB || k0 + j < NKBr) && abl_comp) {
#pragma unroll
                        for (int pi = 0; pi < DQN; ++pi) {
                            const uint2 lo = tr_read(dsp + (k0 + j) * 2048 + dsr[pi][0]), hi = tr_read(dsp + (k0 + j) * 2048 + dsr[pi][1]);
                            bq[j][pi].u = make_uint4(lo.x, lo.y, hi.x, hi.y);
                        }
                    }
#pragma unroll
                for (int j = 0; j < GRP; ++j)
                    if (k0 + j < (NKB ? NKB : MAXKB) && (NKB || k0 + j < NKBr) && abl_comp) {
#pragma unroll
                        for (int pi = 0; pi < DQN; ++pi) mma(dq[pi], kT[k0 + j], bq[j][pi]);
                    }
                if (NKB) __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int pi = 0; pi < DQN; ++pi) {
                const int q = t * 32 + (dq_qs + pi) * 16 + r16;
                if (q < L && abl_st)
                {
                    if constexpr (WIN == 2) {
                        // this window's share + what the first launch stored (exactly two addends per element, the first in memory since the launch
                        // before: bitwise reproducible): the stored row came with the slice (piece 3 of its slot, the images' chunk swizzle)
                        const int rl = (dq_qs + pi) * 16 + r16;
                        const uint2 old = *reinterpret_cast<const uint2*>(ring + slot_t * SLOTB + 3 * 32 * RB + rl * RB + (((2 * dq_dt + (g >> 1)) ^ (rl & 6)) << 4) + (g & 1) * 8);
                        dq[pi][0] += __uint_as_float(old.x << 16); dq[pi][1] += __uint_as_float(old.x & 0xFFFF0000u);
                        dq[pi][2] += __uint_as_float(old.y << 16); dq[pi][3] += __uint_as_float(old.y & 0xFFFF0000u);
                    }
                    *reinterpret_cast<uint2*>(dqh + (unsigned)(q * lddqkv + 16 * dq_dt + 4 * g)) = make_uint2(pack2(dq[pi][0], dq[pi][1]), pack2(dq[pi][2], dq[pi][3]));
                }
            }
        }
    };

    // ---- launch prologue: first head's K, V, lse and the first AHEAD slices
    Head cur{(int)blockIdx.x / H, (int)blockIdx.x % H};
    kv_base = head_base(qkv, ldqkv, A.q_hs, cur, dm);
    for (int part = 0; part < kv_parts; ++part) issue_kv_part(part);
    issue_lse(cur, 0);
    issue_dlt(cur, 0);
    for (int j = 0; j < AHEAD; ++j) issue_next();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    int gs = 0;      // global index of the slice being computed
    int slot = 0;    // its ring slot
#pragma unroll 1
    for (int it = 0; it < nheads; ++it) {
        const int lbuf = it & 1;
        LPI4_STAMP();
        Head nxt = cur;
        next_head(nxt);
        T* const dqh = dqkv + (size_t)cur.b * L * lddqkv + (size_t)cur.h * A.dq_hs;
        if constexpr (WIN == 1) dlt_g = delta + (size_t)(cur.b * H + cur.h) * L;
        // own K, V rows and the K^T fragments out of the images (landed a head ago)
#pragma unroll
        for (int u = 0; u < NUW; ++u)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                kk[u][ks].u = *reinterpret_cast<const uint4*>(k_img + (ub + u) * 16 * RB + rc[ks]);
                vv[u][ks].u = *reinterpret_cast<const uint4*>(v_img + (ub + u) * 16 * RB + rc[ks]);
                if constexpr (SV16) { chunk_f16_to_bf16(kk[u][ks]); chunk_f16_to_bf16(vv[u][ks]); }
                chunk_scale_bf16(vv[u][ks], SCALE);
            }
#pragma unroll
        for (int kb = 0; kb < (DQN ? (NKB ? NKB : MAXKB) : 0); ++kb) {
            kT[kb].u = make_uint4(0, 0, 0, 0);
            if (NKB || kb < NKBr) {
                kT[kb] = tr_pair(k_img + kb * 32 * RB + trw, 16 * RB);
                if constexpr (SV16) chunk_f16_to_bf16(kT[kb]);
            }
        }
#pragma unroll
        for (int u = 0; u < NUW; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) { dk[u][i] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[u][i] = f32x4{0.f, 0.f, 0.f, 0.f}; }

        // this head's lse vector (landed a head ago): -lse log2(e) in place, one entry per thread (the entries behind L stay -inf)
        if ((int)threadIdx.x < L) lse_l[lbuf * Lqp + threadIdx.x] *= -LOG2E;
        stage_delta_load(slot);
        stage_delta(0, 0);
        LPI4_BARRIER();           // delta of slice 0 and the scaled lse complete; every wave has its K, V rows: the images are free
        const bool spread_kv = NSL >= 6 && kv_parts <= NSL - 2 && !(A.flags & 1);      // parts 0 .. over iterations 0 ..: landed well before the head ends
        if (it + 1 < nheads) {
            kv_base = head_base(qkv, ldqkv, A.q_hs, nxt, dm);
            issue_lse(nxt, lbuf ^ 1);
            issue_dlt(nxt, lbuf ^ 1);
            if (!spread_kv)
                for (int part = 0; part < kv_parts; ++part) issue_kv_part(part);
        }
        LPI4_STAMP();
        // the last dQ store of a head (slice NSL - 2, in iteration NSL - 1) is what the wait of that iteration may leave in flight
        const bool dq_last = DQN > 0 && (NSL - 2) * 32 < rows_hi;
        const bool dq_all = rows_hi >= L;
#pragma unroll 1
        for (int t = 0; t < NSL; ++t) {
            LPI4_STAMP();
            const int slot1 = slot + 1 == NSLOT ? 0 : slot + 1;
            if (abl_comp && t + 1 < NSL) stage_delta_load(slot1);
            issue_next();         // slice gs + AHEAD -> the slot slice gs - 2 has left (its staggered matrix half ran in the last iteration)
            if (spread_kv && t < kv_parts && it + 1 < nheads) issue_kv_part(t);
            LPI4_STAMP();
            const int par = t & 1;
            if (abl_comp) {
                const bool run = own_wanted || t * 32 < rows_hi;
                if (t + 1 < NSL && (STAG || !run)) stage_delta(par ^ 1, t + 1);
                LPI4_STAMP();
                if constexpr (STAG) {
                    if (t >= 1 && (own_wanted || (t - 1) * 32 < rows_hi)) stage_m(slot == 0 ? NSLOT - 1 : slot - 1);
                    LPI4_STAMP();
                    if (run) stage_s(t, slot, lbuf, par, false);
                } else if (run) {
                    stage_s(t, slot, lbuf, par, t + 1 < NSL);
                    LPI4_STAMP();
                    stage_m(slot);
                }
                LPI4_STAMP();
            }
            __builtin_amdgcn_sched_barrier(0);
            if (t >= 1 && (t - 1) * 32 < rows_hi) stage_dq(dqh, t - 1, par ^ 1, slot == 0 ? NSLOT - 1 : slot - 1);
            LPI4_STAMP();
            // End of an iteration: this wave's pieces of slice gs + 2 (the next iteration's delta pass reads it) have landed — all but the
            // pieces of the AHEAD - 2 younger slices (2 per slice from each of waves 0-3, 6, 7; stores and K / V pieces issued in between only
            // make the wait stricter).  vmcnt counts stores too, in order: a wait in the first iterations of a head would also wait for the dK / dV
            // stores of the head before (57 KB per CU, all CUs at once).  So the LAST iteration of a head waits for everything but the
            // youngest slice and the dQ store behind it (slices 0 .. 2 of the next head: issued 3+ iterations ago), and the first
            // iteration of a head does not wait at all.  At the end of the stream, or with few slices per head: everything.
            if (NSL >= 6 && gs + AHEAD < nslices) {
                if (sl_mine) {            // waves 4, 5 move no slice pieces (their K / V pieces are older than any wait that matters)
                    if (t == NSL - 1) {   // may stay in flight: this wave's 2 pieces of the youngest slice and its DQN dQ stores behind them
                        if (DQN >= 2 && dq_last) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                        else if (DQN >= 1 && dq_last) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                        else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                    } else if (t >= 3 && dq_all) {
                        // younger than the pieces of slice gs + 2 for sure: the 2 x 2 pieces of slices gs + 3, gs + 4 and the DQN dQ stores of each
                        // of the iterations t - 2, t - 1, t (every row wanted; slices before the last are whole)
                        if constexpr (DQN >= 2) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
                        else if constexpr (DQN == 1) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
                        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                    } else if (t >= 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                } else if (t == NSL - 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the next head's K / V pieces (issued 3+ iterations ago)
            } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            LPI4_STAMP();
            LPI4_BARRIER();
            LPI4_STAMP();
            ++gs;
            slot = slot1;
        }
        if constexpr (STAG) {
            if (abl_comp && (own_wanted || (NSL - 1) * 32 < rows_hi)) stage_m(slot == 0 ? NSLOT - 1 : slot - 1);      // the last slice's matrix half
        }
        if ((NSL - 1) * 32 < rows_hi) stage_dq(dqh, NSL - 1, (NSL - 1) & 1, slot == 0 ? NSLOT - 1 : slot - 1);
        // dK^T / dV^T tiles (16 keys x 64) -> global as WHOLE 128-byte rows: the accumulator layout gives a lane 4 x 8 bytes of its key row; staged
        // through 1 KiB per wave of the dS^T buffer that is idle now (8 rows at a time, 8-byte slots XOR-ed with 2 (row & 7): conflict-free),
        // read back as [row][16-byte chunk] — every store instruction writes 8 full rows (straight from the registers it was 16 rows x 64 B:
        // twice the lines touched per byte)
        {
            char* st = dsb + (NSL & 1) * (Lp * 64) + wave * 1024;
            const int wr = (r16 & 7) * 128, wsw = 2 * (r16 & 7);
            const int rrow = lane >> 3, rch = lane & 7;
            const int rd = rrow * 128 + ((rch ^ (rrow & 7)) << 4);
            auto store_tile = [&](T* tile, const f32x4 (&acc)[4], int row0) {
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    if ((r16 >> 3) == hf) {
#pragma unroll
                        for (int dt = 0; dt < 4; ++dt)
                            *reinterpret_cast<uint2*>(st + wr + (((4 * dt + g) ^ wsw) << 3)) = make_uint2(pack2(acc[dt][0], acc[dt][1]), pack2(acc[dt][2], acc[dt][3]));
                    }
                    const uint4 v = *reinterpret_cast<const uint4*>(st + rd);
                    const int kr = row0 + 8 * hf + rrow;      // key row inside the window
                    if (kr < Lk && abl_st)
                        *reinterpret_cast<uint4*>(tile + (unsigned)((8 * hf + rrow) * lddqkv + 8 * rch)) = v;
                }
            };
#pragma unroll
            for (int u = 0; u < NUW; ++u) {
                if (kw0 + (ub + u) * 16 >= rows_hi) continue;
                T* dst = dqh + (unsigned)((kw0 + (ub + u) * 16) * lddqkv);
                store_tile(dst + A.dq_vs, dk[u], (ub + u) * 16);
                store_tile(dst + 2 * (size_t)A.dq_vs, dv[u], (ub + u) * 16);
            }
        }
        cur = nxt;
    }
}

template <bool SV16, int NKB, int WIN = 0>
__global__ __launch_bounds__(512) void attn_bwd4_kernel(Args4 A) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // the KEY rows of the launch (whole sequence, or the window) decide the images and the units; the lse vectors cover the queries
    const int L = WIN ? A.Lk : A.L, Lp = WIN ? A.Lkp : A.Lp;
    const int Lq = A.L, Lqp = A.Lp;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int NU = Lp >> 4;
    const int base = NU / NWV, rem = NU % NWV;
    const int nu = base + (wave < rem ? 1 : 0);
    const int ub = wave * base + min(wave, rem);
    // once per launch: rows [L, Lp) of the K and V images are zero (the DMA is EXEC-masked there: padded keys then add nothing to dQ),
    // entries [L, Lp) of both lse vectors are -inf (they hold -lse log2(e): P = 0 for padded queries)
    {
        const int per = (Lp - L) * (RB / 16);
        for (int i = threadIdx.x; i < 2 * per; i += blockDim.x) {
            const int im = i / per, rch = i % per;
            *reinterpret_cast<uint4*>(smem + (size_t)im * Lp * RB + (size_t)L * RB + rch * 16) = make_uint4(0, 0, 0, 0);
        }
        float* lse_l = reinterpret_cast<float*>(smem + 2 * Lp * RB + NSLOT * ((WIN == 2 ? 4 : 3) * 32 * RB) + 2 * Lp * 64 + 2 * 32 * 4);
        for (int i = threadIdx.x; i < 2 * (Lqp - Lq); i += blockDim.x) lse_l[(i / (Lqp - Lq)) * Lqp + Lq + i % (Lqp - Lq)] = -INFINITY;
        if constexpr (WIN == 2) {      // the padded queries' delta entries: finite (P = 0 there)
            float* dlt_l = lse_l + 2 * Lqp;
            for (int i = threadIdx.x; i < 2 * (Lqp - Lq); i += blockDim.x) dlt_l[(i / (Lqp - Lq)) * Lqp + Lq + i % (Lqp - Lq)] = 0.f;
        }
    }
    constexpr int NQS = WIN ? 0 : NKB;      // a window launch reads its query slices from A.Lp
    if constexpr (NKB == 7) {       // Lp == 224: 14 units = 2 x 6 + 1 x 2
        if (wave < 4) bwd4_body<2, SV16, 7, 1, false, NQS, WIN>(A, smem, wave, ub, (wave >> 1) ^ 1, wave & 1);     // head-dim blocks 1, 1, 0, 0; query halves 0, 1, 0, 1
        else if (wave < 6) bwd4_body<2, SV16, 7, 0, true, NQS, WIN>(A, smem, wave, ub, 0, 0);
        else bwd4_body<1, SV16, 7, 2, true, NQS, WIN>(A, smem, wave, ub, wave - 4, 0);                             // head-dim blocks 2, 3: both query halves
    } else {
        switch (nu) {
            case 2: bwd4_body<2, SV16, 0, 1, false, 0, WIN>(A, smem, wave, ub, wave & 3, wave >> 2); break;
            case 1: bwd4_body<1, SV16, 0, 1, false, 0, WIN>(A, smem, wave, ub, wave & 3, wave >> 2); break;
            default: bwd4_body<0, SV16, 0, 1, false, 0, WIN>(A, smem, wave, ub, wave & 3, wave >> 2); break;
        }
    }
}

int cu_count4() { return lpi_cu_count(); }

// Lkp: padded key rows of the launch, Lqp: padded query rows
size_t lds_bytes4(int Lkp, int Lqp, int win = 0) {
    return (size_t)2 * Lkp * RB + (size_t)NSLOT * ((win == 2 ? 4 : 3) * 32 * RB) + (size_t)2 * Lkp * 64 + (size_t)2 * 32 * 4 + (size_t)(win == 2 ? 4 : 2) * Lqp * 4;
}
constexpr int WIN0 = 224;      // keys of the first window of a long sequence: the Lp = 224 configuration (14 key units over 8 waves)

}  // namespace

// true if the fourth-generation backward takes this shape (bf16 operands, non-causal, at most 8 x 2 key units)
bool lpi_attn4_bwd_ok(int L, int causal) {
    const int Lp = (L + 31) / 32 * 32;
    if (causal || L < 1) return false;
    if (Lp <= 32 * MAXKB) return Lp / 16 <= NWV * MAXU && lds_bytes4(Lp, Lp) <= 160 * 1024;
    // two key windows: 224 keys + the rest (<= 64) against all queries
    return L - WIN0 <= 64 && lds_bytes4(WIN0, Lp) <= 160 * 1024;
}

int lpi_attn4_bwd(int B, int L, int H, const void* qkv, int ldqkv, const void* ctx, int ldctx, const void* dctx, int lddctx, const float* lse,
                  float* delta, void* dqkv, int lddqkv, hipStream_t s, int saved_f16, int rows_hi, const int* lay) {
    // lay: NULL = the interleaved layout, else {q_hs, q_vs, dq_hs, dq_vs, c_hs, dc_hs} (Args4)
    const int l_qh = lay ? lay[0] : HD, l_qv = lay ? lay[1] : H * HD, l_dh = lay ? lay[2] : HD, l_dv = lay ? lay[3] : H * HD, l_ch = lay ? lay[4] : HD, l_dch = lay ? lay[5] : HD;
    const int Lp = (L + 31) / 32 * 32;
    const int total = B * H;
    int grid = std::min(total, cu_count4());
    if (g_lpi_tuning[11] > 0) grid = std::min(grid, g_lpi_tuning[11]);      // tests: several heads per workgroup at small B H
    static LdsOnce o0, o1, o2, o3, w0, w1, w2, w3;
    if (Lp > 32 * MAXKB) {
        // a sequence of more than 224 tokens (ViT-L/14: 273) as TWO launches over all queries: keys 0 .. 223 (the Lp = 224 configuration), then keys
        // 224 .. L - 1 (generic configuration, 4 units at most) whose dQ share is added to the first launch's.  lse and delta are per query, so the two
        // key windows are independent; dK / dV of a window are complete.  Every query slice streams through twice (the price of not fitting 18 key
        // units on 8 waves: three units per wave spill).
#define BWD4W(S, K, W, O, ARGS, LDS)                                                                       \
    do {                                                                                                   \
        if (int e = lpi_ensure_lds(O, (const void*)attn_bwd4_kernel<S, K, W>, 160 * 1024)) return e;       \
        LPI_LAUNCH((attn_bwd4_kernel<S, K, W>), dim3(grid), dim3(64 * NWV), LDS, s, ARGS);                 \
    } while (0)
        const int Lk1 = L - WIN0, Lkp1 = (Lk1 + 31) / 32 * 32;
        const Args4 A0{L, Lp, H, total, (const T*)qkv, ldqkv, (const T*)ctx, ldctx, (const T*)dctx, lddctx, lse, delta, (T*)dqkv, lddqkv, rows_hi, g_lpi_tuning[12],
                       0, WIN0, WIN0, 0, l_qh, l_qv, l_dh, l_dv, l_ch, l_dch};
        const Args4 A1{L, Lp, H, total, (const T*)qkv, ldqkv, (const T*)ctx, ldctx, (const T*)dctx, lddctx, lse, delta, (T*)dqkv, lddqkv, rows_hi, g_lpi_tuning[12],
                       WIN0, Lk1, Lkp1, 1, l_qh, l_qv, l_dh, l_dv, l_ch, l_dch};
        if (saved_f16) { BWD4W(true, 7, 1, w0, A0, lds_bytes4(WIN0, Lp)); BWD4W(true, 0, 2, w1, A1, lds_bytes4(Lkp1, Lp, 2)); }
        else { BWD4W(false, 7, 1, w2, A0, lds_bytes4(WIN0, Lp)); BWD4W(false, 0, 2, w3, A1, lds_bytes4(Lkp1, Lp, 2)); }
#undef BWD4W
        LPI_CHECK_LAST();
        return 0;
    }
    const size_t lds = lds_bytes4(Lp, Lp);
    const Args4 A{L, Lp, H, total, (const T*)qkv, ldqkv, (const T*)ctx, ldctx, (const T*)dctx, lddctx, lse, delta, (T*)dqkv, lddqkv, rows_hi, g_lpi_tuning[12],
                  0, L, Lp, 0, l_qh, l_qv, l_dh, l_dv, l_ch, l_dch};
#define BWD4(S, K, O)                                                                               \
    do {                                                                                            \
        if (int e = lpi_ensure_lds(O, (const void*)attn_bwd4_kernel<S, K>, 160 * 1024)) return e;   \
        LPI_LAUNCH((attn_bwd4_kernel<S, K>), dim3(grid), dim3(64 * NWV), lds, s, A);                \
    } while (0)
    if (Lp == 224) {
        if (saved_f16) BWD4(true, 7, o0);
        else BWD4(false, 7, o1);
    } else if (saved_f16) BWD4(true, 0, o2);
    else BWD4(false, 0, o3);
#undef BWD4
    LPI_CHECK_LAST();
    return 0;
}
