// Library-level entry points of liblpi_hip.so.
#include <atomic>
#include "common.h"

static std::atomic<uint64_t> g_launches{0};

extern "C" void lpi_count_launch() { g_launches.fetch_add(1, std::memory_order_relaxed); }
extern "C" uint64_t lpi_launch_count(void) { return g_launches.load(std::memory_order_relaxed); }
extern "C" int lpi_version(void) { return 100; }

// tuning knobs: [0] / [1] minimum number of 256x256 tiles for the phased 256x256 GEMM kernel, bf16 / f32 operands (INT_MAX disables it).
// f32 is MFMA-bound at either tile size, so the bigger tile only pays when its last partial round of tiles is short.
int g_lpi_tuning[8] = {1, 1500, 0, 0, 0, 160, 1, 0};
extern "C" int lpi_set_tuning(int key, int value) {
    if (key < 0 || key >= 8) return LPI_EINVAL;
    g_lpi_tuning[key] = value;
    return 0;
}
