// Library-level entry points of liblpi_hip.so.
#include <atomic>
#include "common.h"

static std::atomic<uint64_t> g_launches{0};

extern "C" void lpi_count_launch() { g_launches.fetch_add(1, std::memory_order_relaxed); }
extern "C" uint64_t lpi_launch_count(void) { return g_launches.load(std::memory_order_relaxed); }
// bump with every change of a signature or of what an argument means (lpi_amd/_lib.py EXPECTED_ABI).  Diagnostic / ablation builds (tools/build_variant.sh)
// report LPI_ABI_VERSION + 1 000 000: the binding loads such a library only when LPI_LIB names it, and says so on stderr.
#ifndef LPI_ABI_VERSION
#define LPI_ABI_VERSION 604
#endif
#ifdef LPI_VARIANT_BUILD
extern "C" int lpi_version(void) { return LPI_ABI_VERSION + 1000000; }
#else
extern "C" int lpi_version(void) { return LPI_ABI_VERSION; }
#endif

// CU count of the CURRENT device (the persistent kernels launch one workgroup per CU), looked up once per device; safe from any host thread (a cached
// value is written once, every writer writes the same one)
int lpi_cu_count() {
    static std::atomic<int> cache[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    int n = cache[dev].load(std::memory_order_relaxed);
    if (n == 0) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        cache[dev].store(v, std::memory_order_relaxed);
        n = v;
    }
    return n;
}

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
