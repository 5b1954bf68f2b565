// Library-level entry points of liblpi_hip.so.
#include <atomic>
#include "common.h"

static std::atomic<uint64_t> g_launches{0};

extern "C" void lpi_count_launch() { g_launches.fetch_add(1, std::memory_order_relaxed); }
extern "C" uint64_t lpi_launch_count(void) { return g_launches.load(std::memory_order_relaxed); }
extern "C" int lpi_version(void) { return 100; }
