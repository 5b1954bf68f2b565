// Shared device helpers for the LPI gfx950 kernels (wave64, MFMA 16x16 fragments, 16-byte chunks).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/lpi_hip.h"

typedef unsigned short bf16_t;  // storage type of a bfloat16
typedef _Float16 f16_t;         // IEEE half: storage type of the residual stream in bf16 mode
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short short4v;

#define WAVE 64

extern "C" void lpi_count_launch();
// Launch + count.  The sticky HIP error state is cleared first so that LPI_CHECK_LAST reports only THIS launch's error
// (the host framework's own runtime calls can leave a stale code behind).
#define LPI_LAUNCH(...)                     \
    do {                                    \
        (void)hipGetLastError();            \
        hipLaunchKernelGGL(__VA_ARGS__);    \
        lpi_count_launch();                 \
    } while (0)

#define LPI_CHECK_LAST()                                      \
    do {                                                      \
        hipError_t e__ = hipGetLastError();                   \
        if (e__ != hipSuccess) return (int)e__;               \
    } while (0)

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE attribute and launches come from several host threads (forward on the
// caller's thread, backward on autograd's): one atomic bit per device, one LdsOnce per kernel function.
#include <atomic>
struct LdsOnce {
    std::atomic<uint64_t> devices{0};
};
inline int lpi_ensure_lds(LdsOnce& once, const void* kern, int bytes) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return LPI_EINVAL;
    const uint64_t bit = 1ull << (dev & 63);
    if (once.devices.load(std::memory_order_acquire) & bit) return 0;
    hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return (int)e;
    once.devices.fetch_or(bit, std::memory_order_release);
    return 0;
}
// which GEMM kernel the last lpi_gemm_nt / lpi_gemm_nt_rows call of this thread launched (LPI_GEMM_K_*; lpi_gemm_last_kernel)
void lpi_note_gemm_kernel(int which);
int lpi_cu_count();      // CUs of the current device (api.hip: cached per device)

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
    __bf16 b = (__bf16)f;  // v_cvt_pk_bf16_f32: round-to-nearest-even, NaN preserving
    return __builtin_bit_cast(bf16_t, b);
}

template <typename T> struct Elem;
template <> struct Elem<float> {
    static constexpr int DT = LPI_F32;
    static constexpr int EPC = 4;  // elements per 16-byte chunk
    __device__ static __forceinline__ float ld(const float* p) { return *p; }
    __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
    // load / store 4 consecutive elements as floats
    __device__ static __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
    __device__ static __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
};
template <> struct Elem<bf16_t> {
    static constexpr int DT = LPI_BF16;
    static constexpr int EPC = 8;
    __device__ static __forceinline__ float ld(const bf16_t* p) { return bf16_to_f32(*p); }
    __device__ static __forceinline__ void st(bf16_t* p, float v) { *p = f32_to_bf16(v); }
    __device__ static __forceinline__ f32x4 ld4(const bf16_t* p) {
        uint2 u = *reinterpret_cast<const uint2*>(p);
        f32x4 r;
        r[0] = __uint_as_float(u.x << 16);
        r[1] = __uint_as_float(u.x & 0xFFFF0000u);
        r[2] = __uint_as_float(u.y << 16);
        r[3] = __uint_as_float(u.y & 0xFFFF0000u);
        return r;
    }
    __device__ static __forceinline__ void st4(bf16_t* p, f32x4 v) {
        bf16x4 b;
        b[0] = (__bf16)v[0]; b[1] = (__bf16)v[1]; b[2] = (__bf16)v[2]; b[3] = (__bf16)v[3];
        *reinterpret_cast<bf16x4*>(p) = b;
    }
};

template <> struct Elem<f16_t> {
    static constexpr int DT = LPI_F16;
    static constexpr int EPC = 8;
    __device__ static __forceinline__ float ld(const f16_t* p) { return (float)*p; }
    __device__ static __forceinline__ void st(f16_t* p, float v) { *p = (f16_t)v; }
    __device__ static __forceinline__ f32x4 ld4(const f16_t* p) {
        f16x4 h = *reinterpret_cast<const f16x4*>(p);
        f32x4 r;
        r[0] = (float)h[0]; r[1] = (float)h[1]; r[2] = (float)h[2]; r[3] = (float)h[3];
        return r;
    }
    __device__ static __forceinline__ void st4(f16_t* p, f32x4 v) {
        f16x4 h;
        h[0] = (f16_t)v[0]; h[1] = (f16_t)v[1]; h[2] = (f16_t)v[2]; h[3] = (f16_t)v[3];
        *reinterpret_cast<f16x4*>(p) = h;
    }
};

// 4 consecutive elements of TT from 4 floats with a STREAMING (non-temporal) store: for tensors that are not read again soon
template <typename TT> __device__ __forceinline__ void st4_nt(TT* p, f32x4 v);
// LPI_ST_POLICY (A/B builds): 0 = `nt` (default), 1 = `sc1`, 2 = `sc0 sc1`, 3 = `nt sc1`, 4 = `sc0`  — the cache-policy bits of the streaming store
#ifndef LPI_ST_POLICY
#define LPI_ST_POLICY 0
#endif
#if LPI_ST_POLICY == 1
#define LPI_ST_BITS "sc1"
#elif LPI_ST_POLICY == 2
#define LPI_ST_BITS "sc0 sc1"
#elif LPI_ST_POLICY == 3
#define LPI_ST_BITS "nt sc1"
#elif LPI_ST_POLICY == 4
#define LPI_ST_BITS "sc0"
#endif
__device__ __forceinline__ void st_stream8(void* p, unsigned w0, unsigned w1) {
    typedef __attribute__((ext_vector_type(2))) unsigned u32x2_;
#if LPI_ST_POLICY == 0
    __builtin_nontemporal_store((u32x2_){w0, w1}, reinterpret_cast<u32x2_*>(p));
#else
    const u32x2_ w = {w0, w1};
    asm volatile("global_store_dwordx2 %0, %1, off " LPI_ST_BITS :: "v"(p), "v"(w) : "memory");
#endif
}
template <> __device__ __forceinline__ void st4_nt<float>(float* p, f32x4 v) {
#if LPI_ST_POLICY == 0
    __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p));
#else
    asm volatile("global_store_dwordx4 %0, %1, off " LPI_ST_BITS :: "v"(p), "v"(v) : "memory");
#endif
}
// 16-byte fragment chunk as it sits in a lane's registers
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
union Chunk {
    uint4 u;
    f32x4 f;
    bf16x8 h;
    f16x8 hh;      // the same 16 bytes as 8 IEEE halves (f16 operand mode)
};

// acc += A_chunk (x) B_chunk over the chunk's k values, 16x16 output tile.
// Lane l supplies row (l & 15) of each operand and k-group (l >> 4); see cdna_hip_programming.md section 3.
template <typename T> __device__ __forceinline__ void mma_chunk(f32x4& acc, const Chunk& a, const Chunk& b);
template <> __device__ __forceinline__ void mma_chunk<float>(f32x4& acc, const Chunk& a, const Chunk& b) {
    // k index of element s in k-group g is 4g+s for both operands: any consistent assignment sums the same products
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.f[0], b.f[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.f[1], b.f[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.f[2], b.f[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.f[3], b.f[3], acc, 0, 0, 0);
}
template <> __device__ __forceinline__ void mma_chunk<bf16_t>(f32x4& acc, const Chunk& a, const Chunk& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.h, b.h, acc, 0, 0, 0);
}
// fp16 operands (the reference's own arithmetic type, model.py:394-415): same rate, 3 more mantissa bits than bf16
template <> __device__ __forceinline__ void mma_chunk<f16_t>(f32x4& acc, const Chunk& a, const Chunk& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.hh, b.hh, acc, 0, 0, 0);
}

// two floats -> one dword of T (round to nearest even): v_cvt_pk_bf16_f32 / v_cvt_pkrtz... (f16: v_cvt_pk via two v_cvt_f16_f32)
template <typename T> __device__ __forceinline__ uint32_t pack2_t(float a, float b);
template <> __device__ __forceinline__ uint32_t pack2_t<bf16_t>(float a, float b) {
    typedef __attribute__((ext_vector_type(2))) float f2_;
    typedef __attribute__((ext_vector_type(2))) __bf16 b2_;
    return __builtin_bit_cast(uint32_t, __builtin_convertvector((f2_){a, b}, b2_));
}
template <> __device__ __forceinline__ uint32_t pack2_t<f16_t>(float a, float b) {
    typedef __attribute__((ext_vector_type(2))) float f2_;
    typedef __attribute__((ext_vector_type(2))) _Float16 h2_;
    return __builtin_bit_cast(uint32_t, __builtin_convertvector((f2_){a, b}, h2_));
}
template <> __device__ __forceinline__ void st4_nt<bf16_t>(bf16_t* p, f32x4 v) { st_stream8(p, pack2_t<bf16_t>(v[0], v[1]), pack2_t<bf16_t>(v[2], v[3])); }
template <> __device__ __forceinline__ void st4_nt<f16_t>(f16_t* p, f32x4 v) { st_stream8(p, pack2_t<f16_t>(v[0], v[1]), pack2_t<f16_t>(v[2], v[3])); }
// element e (0..7) of a chunk of T as float
template <typename T> __device__ __forceinline__ float chunk_elem(const Chunk& c, int e);
template <> __device__ __forceinline__ float chunk_elem<bf16_t>(const Chunk& c, int e) { return (float)c.h[e]; }
template <> __device__ __forceinline__ float chunk_elem<f16_t>(const Chunk& c, int e) { return (float)c.hh[e]; }
// a chunk of 8 halves -> the same 8 values as bf16 (the backward's operand type; gradients do not fit fp16's range)
__device__ __forceinline__ void chunk_f16_to_bf16(Chunk& c) {
    Chunk o;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t w = pack2_t<bf16_t>((float)c.hh[2 * i], (float)c.hh[2 * i + 1]);
        if (i == 0) o.u.x = w; else if (i == 1) o.u.y = w; else if (i == 2) o.u.z = w; else o.u.w = w;
    }
    c = o;
}

// Store one 64-element bf16 row of a TRANSPOSED 16x16-tile accumulator set: o[dt] (dt = 0..3) holds, for output row (lane & 15),
// the elements 16 dt + 4 g + 0..3 (g = lane >> 4) — the layout of O^T / dQ^T / dK^T / dV^T in the attention kernels.  Stored straight
// from that layout a lane writes 4 x 8 bytes and an instruction touches 16 rows x 32 B.  Two v_permlane16_swap per dt pair trade the
// g-odd lanes' piece of tile 2p for the g-even lanes' piece of tile 2p+1, after which every lane owns 8 CONSECUTIVE elements: 2 x 16-byte
// stores per row, 16 rows x 64 B per instruction (half the store instructions, the same bytes and bits).  Must be called with all
// lanes active; `valid` masks the store only.  row_ptr = &row[0] of this lane's output row.
template <typename T16>
__device__ __forceinline__ void store_row16_t(T16* row_ptr, const f32x4 (&o)[4], int g, bool valid) {
    auto pk = [](float a, float b) { return pack2_t<T16>(a, b); };
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const uint32_t x0 = pk(o[2 * p][0], o[2 * p][1]), x1 = pk(o[2 * p][2], o[2 * p][3]);
        const uint32_t y0 = pk(o[2 * p + 1][0], o[2 * p + 1][1]), y1 = pk(o[2 * p + 1][2], o[2 * p + 1][3]);
        const auto r0 = __builtin_amdgcn_permlane16_swap(x0, y0, false, false);      // {x rows 0,2 | y rows 0,2 -> x rows 1,3}, {x rows 1,3 -> y rows 0,2 | y rows 1,3}
        const auto r1 = __builtin_amdgcn_permlane16_swap(x1, y1, false, false);
        if (valid) *reinterpret_cast<uint4*>(row_ptr + 32 * p + (g & 1) * 16 + (g >> 1) * 8) = make_uint4(r0[0], r1[0], r0[1], r1[1]);
    }
}

__device__ __forceinline__ void store_row_bf16_t(bf16_t* row_ptr, const f32x4 (&o)[4], int g, bool valid) { store_row16_t<bf16_t>(row_ptr, o, g, valid); }


// Cross-lane reductions at VALU speed: DPP quad permutes and row mirrors inside each 16-lane row, then v_permlane16_swap /
// v_permlane32_swap across rows (with vdst = src = v one instruction leaves {own, partner} in its two results).  `__shfl_xor`
// compiles to ds_bpermute_b32 — an LDS-crossbar round trip per step, five or six dependent ones per reduction.  Every lane ends
// with the full result.  LPI_SHFL_REDUCE=1 keeps the shuffle form (A/B switch; measured: no step-time difference — the LayerNorm
// kernels are HBM-bound and their occupancy hid the crossbar latency — kept for the lower instruction count).
#ifndef LPI_SHFL_REDUCE
#define LPI_SHFL_REDUCE 0
#endif
template <int CTRL> __device__ __forceinline__ float dpp_move(float v) {
    return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_move<0xB1>(v);     // quad_perm [1,0,3,2]  (lane ^ 1)
    v += dpp_move<0x4E>(v);     // quad_perm [2,3,0,1]  (lane ^ 2)
    v += dpp_move<0x141>(v);    // row_half_mirror: the other quad of each 8 lanes
    v += dpp_move<0x140>(v);    // row_mirror: the other 8 lanes of the row
    return v;
}
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, dpp_move<0xB1>(v));
    v = fmaxf(v, dpp_move<0x4E>(v));
    v = fmaxf(v, dpp_move<0x141>(v));
    v = fmaxf(v, dpp_move<0x140>(v));
    return v;
}
// sum over each 32-lane half of the wave
__device__ __forceinline__ float half_sum(float v) {
#if LPI_SHFL_REDUCE
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
#else
    v = row16_sum(v);
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
#endif
}
__device__ __forceinline__ float wave_sum(float v) {
#if LPI_SHFL_REDUCE
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
#else
    v = half_sum(v);
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
#endif
}
__device__ __forceinline__ float wave_max(float v) {
#if LPI_SHFL_REDUCE
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
#else
    v = row16_max(v);
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
#endif
}

// Row statistics in a GEMM epilogue (LPI_EPI_RES_ROWSTATS): each 32-lane half of a wave holds, per lane, partial sums s[j] / q[j] of EIGHT rows
// (one row per j, the half's 32 lanes x 4 columns = 128 columns of it).  A halving exchange instead of eight full reductions: at each of three steps a
// lane keeps half of its rows and receives its partner's partials of those (v_permlane16_swap between the two 16-lane rows, then DPP row_mirror and
// row_half_mirror inside a row — any pairing that joins the two halves of the undecided lane bit serves, the lanes' sets stay disjoint), then two quad
// steps.  Every lane ends with the full 128-column sums of row j = (lane >> 2) & 7 of its half: 9 exchanges per quantity, not 40.  Fixed order.
__device__ __forceinline__ void rowstats8_half_reduce(const float (&s)[8], const float (&q)[8], int lane, float& so, float& qo) {
    float a[2][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(s[i]), __float_as_uint(s[4 + i]), false, false);
        a[0][i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);      // lanes 0-15 of a half: row i; lanes 16-31: row 4 + i
        r = __builtin_amdgcn_permlane16_swap(__float_as_uint(q[i]), __float_as_uint(q[4 + i]), false, false);
        a[1][i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    const bool b3 = lane & 8, b2 = lane & 4;
    float o[2];
#pragma unroll
    for (int w = 0; w < 2; ++w) {
        float b[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const float keep = b3 ? a[w][2 + k] : a[w][k], send = b3 ? a[w][k] : a[w][2 + k];
            b[k] = keep + dpp_move<0x140>(send);      // row_mirror: lane 15 - l, the other value of bit 3
        }
        const float keep = b2 ? b[1] : b[0], send = b2 ? b[0] : b[1];
        float c = keep + dpp_move<0x141>(send);       // row_half_mirror: lane 7 - l of the 8, the other value of bit 2
        c += dpp_move<0xB1>(c);
        c += dpp_move<0x4E>(c);
        o[w] = c;
    }
    so = o[0];
    qo = o[1];
}
// sum and sum of squares of four values AS fp16 ROUNDS THEM, from the two packed dwords the store writes: v_dot2_f32_f16 against (1, 1) and against
// itself — four instructions per row instead of eight conversions and seven f32 operations (exact products, f32 accumulation)
__device__ __forceinline__ void f16x4_sum_sumsq(uint32_t w0, uint32_t w1, float& s, float& q) {
    typedef __attribute__((ext_vector_type(2))) _Float16 h2_;
    const h2_ h0 = __builtin_bit_cast(h2_, w0), h1 = __builtin_bit_cast(h2_, w1);
    const h2_ one = {(_Float16)1.0f, (_Float16)1.0f};
    s = __builtin_amdgcn_fdot2(h1, one, __builtin_amdgcn_fdot2(h0, one, 0.f, false), false);
    q = __builtin_amdgcn_fdot2(h1, h1, __builtin_amdgcn_fdot2(h0, h0, 0.f, false), false);
}
// the four values as the store of TT rounds them (fp16 / bf16), back in f32
template <typename TT> __device__ __forceinline__ f32x4 rounded4(f32x4 v);
template <> __device__ __forceinline__ f32x4 rounded4<float>(f32x4 v) { return v; }
template <> __device__ __forceinline__ f32x4 rounded4<f16_t>(f32x4 v) {
    f32x4 r;
    r[0] = (float)(f16_t)v[0]; r[1] = (float)(f16_t)v[1]; r[2] = (float)(f16_t)v[2]; r[3] = (float)(f16_t)v[3];
    return r;
}
template <> __device__ __forceinline__ f32x4 rounded4<bf16_t>(f32x4 v) {
    f32x4 r;
    r[0] = (float)(__bf16)v[0]; r[1] = (float)(__bf16)v[1]; r[2] = (float)(__bf16)v[2]; r[3] = (float)(__bf16)v[3];
    return r;
}

// sigmoid through v_rcp_f32 (1 ulp): a plain `/` compiles to the IEEE sequence (2 v_div_scale + v_rcp + 4 fma + v_div_fmas +
// v_div_fixup per element), which doubled the VALU instructions of the QuickGELU / gelu' GEMM epilogues.  LPI_IEEE_DIV=1 keeps it
// (A/B switch).
#ifndef LPI_IEEE_DIV
#define LPI_IEEE_DIV 0
#endif
__device__ __forceinline__ float fast_sigmoid1702(float u) {
#if LPI_IEEE_DIV
    return 1.0f / (1.0f + __expf(-1.702f * u));
#else
    return __builtin_amdgcn_rcpf(1.0f + __expf(-1.702f * u));
#endif
}
// The same for four values with the multiplies and adds as packed f32 operations (v_pk_mul_f32 / v_pk_add_f32: two lanes of work per issue slot;
// the exponential and the reciprocal have no packed form) and -1.702 * log2(e) folded into one constant: the QuickGELU / gelu' epilogues of the
// K = 768 GEMMs are bound by exactly these vector instructions (128 elements per lane per 256x256 tile).
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
__device__ __forceinline__ f32x4 fast_sigmoid1702_x4(f32x4 u) {
    constexpr float K = -1.702f * 1.4426950408889634f;
    const f32x2_t t0 = (f32x2_t){u[0], u[1]} * K, t1 = (f32x2_t){u[2], u[3]} * K;
    const f32x2_t d0 = (f32x2_t){__builtin_amdgcn_exp2f(t0[0]), __builtin_amdgcn_exp2f(t0[1])} + 1.0f;
    const f32x2_t d1 = (f32x2_t){__builtin_amdgcn_exp2f(t1[0]), __builtin_amdgcn_exp2f(t1[1])} + 1.0f;
    return f32x4{__builtin_amdgcn_rcpf(d0[0]), __builtin_amdgcn_rcpf(d0[1]), __builtin_amdgcn_rcpf(d1[0]), __builtin_amdgcn_rcpf(d1[1])};
}
__device__ __forceinline__ f32x4 quick_gelu_x4(f32x4 u) { return u * fast_sigmoid1702_x4(u); }
// value and derivative from ONE sigmoid: g = u s, g' = s (1 + 1.702 u (1 - s)) = s + 1.702 (g - g s): with g in hand the derivative is two
// fused multiply-adds per element instead of three operations on u (the QuickGELU epilogue of the c_fc GEMM is bound by its vector
// instructions: tools/isa_count.py gives 18 packed operations + 8 transcendentals per four elements before this form, 16 + 8 with it)
// g - g s for two values as ONE packed multiply-add with the negation as an operand modifier (hipcc negates with a v_xor per element)
__device__ __forceinline__ f32x2_t pk_g_minus_gs(f32x2_t g, f32x2_t s) {
    f32x2_t d;
    asm("v_pk_fma_f32 %0, %1, %2, %1 neg_lo:[0,1,0] neg_hi:[0,1,0]" : "=v"(d) : "v"(g), "v"(s));
    return d;
}
__device__ __forceinline__ f32x4 quick_gelu_grad_from(f32x4 g, f32x4 s) {
    const f32x2_t d0 = pk_g_minus_gs((f32x2_t){g[0], g[1]}, (f32x2_t){s[0], s[1]}), d1 = pk_g_minus_gs((f32x2_t){g[2], g[3]}, (f32x2_t){s[2], s[3]});
    return 1.702f * f32x4{d0[0], d0[1], d1[0], d1[1]} + s;
}
__device__ __forceinline__ void quick_gelu_both_x4(f32x4 u, f32x4& g, f32x4& dg) {
    const f32x4 s = fast_sigmoid1702_x4(u);
    g = u * s;
    dg = quick_gelu_grad_from(g, s);
}
__device__ __forceinline__ f32x4 quick_gelu_grad_x4(f32x4 u) {
    const f32x4 s = fast_sigmoid1702_x4(u);
    return quick_gelu_grad_from(u * s, s);
}
__device__ __forceinline__ float quick_gelu(float u) { return u * fast_sigmoid1702(u); }
__device__ __forceinline__ float quick_gelu_grad(float u) {
    const float s = fast_sigmoid1702(u), g = u * s;
    return fmaf(1.702f, fmaf(g, -s, g), s);
}
