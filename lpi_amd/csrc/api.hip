// Library-level entry points of liblpi_hip.so.
#include <atomic>
#include "common.h"

static std::atomic<uint64_t> g_launches{0};

extern "C" void lpi_count_launch() { g_launches.fetch_add(1, std::memory_order_relaxed); }
extern "C" uint64_t lpi_launch_count(void) { return g_launches.load(std::memory_order_relaxed); }
extern "C" int lpi_version(void) { return 213; }

static thread_local int t_last_gemm_kernel = -1;
void lpi_note_gemm_kernel(int which) { t_last_gemm_kernel = which; }
extern "C" int lpi_gemm_last_kernel(void) { return t_last_gemm_kernel; }

// tuning knobs: [0] / [1] minimum number of 256x256 tiles for the phased 256x256 GEMM kernel, bf16 / f32 operands (INT_MAX disables it).
// f32 is MFMA-bound at either tile size, so the bigger tile only pays when its last partial round of tiles is short.
int g_lpi_tuning[16] = {1, 1500, 0, 0, 0, 160, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0};
extern "C" int lpi_get_tuning(int key) { return (key < 0 || key >= 16) ? LPI_EINVAL : g_lpi_tuning[key]; }
extern "C" int lpi_set_tuning(int key, int value) {
    if (key < 0 || key >= 16) return LPI_EINVAL;
    g_lpi_tuning[key] = value;
    return 0;
}

// zero-fill on the stream (non-default A/B paths only: the default backward writes its gradient stream instead of zero-filling it)
extern "C" int lpi_zero(void* ptr, long bytes, void* stream) {
    if (!ptr || bytes <= 0) return LPI_EINVAL;
    return (int)hipMemsetAsync(ptr, 0, (size_t)bytes, (hipStream_t)stream);
}

// ---- CU-partitioned lanes -----------------------------------------------------------------------------------------------------
// A HIP stream restricted to a subset of the 256 CUs (hipExtStreamCreateWithCUMask).  step.py runs two half-batch "lanes" on two
// disjoint halves of the chip so that one lane's HBM-bound phases (LayerNorm, attention, GEMM epilogues) overlap the other lane's
// matrix-core phases; without a mask the 256x256 GEMM workgroups (one per CU, all LDS) of one stream own the whole chip and
// nothing co-schedules.
extern "C" int lpi_stream_create_cu_mask(const uint32_t* mask, int words, void** stream_out) {
    if (!mask || words <= 0 || !stream_out) return LPI_EINVAL;
    hipStream_t s = nullptr;
    hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)words, mask);
    if (e != hipSuccess) return (int)e;
    *stream_out = (void*)s;
    return 0;
}
extern "C" int lpi_stream_destroy(void* stream) {
    if (!stream) return LPI_EINVAL;
    return (int)hipStreamDestroy((hipStream_t)stream);
}
extern "C" int lpi_device_cu_count(void) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess) return LPI_EINVAL;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return LPI_EINVAL;
    return n;
}

// where workgroups land: out[2*b] = HW_REG_XCC_ID, out[2*b+1] = HW_REG_HW_ID of workgroup b (diagnostic for the CU masks above)
__global__ void placement_kernel(uint32_t* out, int spin) {
    uint32_t xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(64);      // hold the CU so that a grid spreads over every enabled CU
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hw; }
}
extern "C" int lpi_probe_placement(int blocks, int threads, int lds_bytes, int spin, uint32_t* out, void* stream) {
    if (blocks <= 0 || threads <= 0 || threads > 1024 || lds_bytes < 0 || lds_bytes > 160 * 1024 || !out) return LPI_EINVAL;
    hipError_t e = hipFuncSetAttribute((const void*)placement_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    LPI_LAUNCH(placement_kernel, dim3(blocks), dim3(threads), lds_bytes, (hipStream_t)stream, out, spin);
    LPI_CHECK_LAST();
    return 0;
}
