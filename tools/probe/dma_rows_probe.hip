// Diagnostic: LDS-DMA streaming rate of GEMM-shaped tile reads with 128-byte rows (one K-tile = 64 KB per workgroup step) versus
// 64-byte rows (one k-step = 32 KB per step, the same bytes in twice as many row segments), 256 threads per workgroup, one
// workgroup per CU, `depth` instructions kept in flight per thread.  No compute, no LDS reads.  Host-side event timing.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned short bf16_t;

template <int ROWB>   // bytes of K per staged row: 128 or 64
__global__ __launch_bounds__(256, 1) void stream(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, int K, int tiles_m, int tiles_n, int depth_steps)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];      // 128 KiB ring
    constexpr int RPI = 4096 / ROWB;                // rows per 256-thread instruction
    constexpr int TPR = ROWB / 16;                  // threads per row
    constexpr int INSTR = 512 / RPI;                // instructions per step (256 A rows + 256 B rows)
    constexpr int STEP_BYTES = 512 * ROWB;
    constexpr int SLOTS = 131072 / STEP_BYTES;
    const int nwg = tiles_m * tiles_n;
    int bid = blockIdx.x;
    { const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3; bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx; }
    const int group = bid / (8 * tiles_n), first_m = group * 8, gsz = min(tiles_m - first_m, 8);
    const int in_group = bid - group * 8 * tiles_n;
    const int tm = first_m + in_group % gsz, tn = in_group / gsz;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int srow = tid / TPR, schunk = tid % TPR;
    const bf16_t* a_src = A + (size_t)(tm * 256 + srow) * K + schunk * 8;
    const bf16_t* b_src = B + (size_t)(tn * 256 + srow) * K + schunk * 8;
    const unsigned lds_w = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + wave * 1024);
    auto glds16 = [&](const bf16_t* src, unsigned lds_addr) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(src), "s"(lds_addr) : "memory");
    };
    const int nsteps = K * 2 / ROWB;
    for (int s = 0; s < nsteps; ++s) {
        const unsigned base = lds_w + (s % SLOTS) * STEP_BYTES;
        const bf16_t* ap = a_src + (size_t)s * (ROWB / 2);
        const bf16_t* bp = b_src + (size_t)s * (ROWB / 2);
#pragma unroll
        for (int i = 0; i < INSTR / 2; ++i) glds16(ap + (size_t)i * RPI * K, base + i * 4096);
#pragma unroll
        for (int i = 0; i < INSTR / 2; ++i) glds16(bp + (size_t)i * RPI * K, base + STEP_BYTES / 2 + i * 4096);
        // keep `depth_steps` steps in flight
        if (depth_steps == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(INSTR * 1 > 63 ? 63 : INSTR * 1) : "memory");
        else if (depth_steps == 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(INSTR * 2 > 63 ? 63 : INSTR * 2) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(INSTR * 3 > 63 ? 63 : INSTR * 3) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

int main() {
    const int M = 54528;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int cfg = 0; cfg < 2; ++cfg) {
        const int N = cfg == 0 ? 2304 : 768, K = cfg == 0 ? 768 : 2304;
        bf16_t *A, *B;
        (void)hipMalloc(&A, (size_t)M * K * 2); (void)hipMalloc(&B, (size_t)N * K * 2);
        (void)hipMemset(A, 0, (size_t)M * K * 2); (void)hipMemset(B, 0, (size_t)N * K * 2);
        const int tm = M / 256, tn = N / 256;
        for (int rowb : {128, 64})
            for (int depth = 1; depth <= 3; ++depth) {
                float best = 1e9f;
                for (int rep = 0; rep < 4; ++rep) {
                    (void)hipEventRecord(e0, 0);
                    if (rowb == 128) hipLaunchKernelGGL(stream<128>, dim3(tm * tn), dim3(256), 131072, 0, A, B, K, tm, tn, depth);
                    else hipLaunchKernelGGL(stream<64>, dim3(tm * tn), dim3(256), 131072, 0, A, B, K, tm, tn, depth);
                    (void)hipEventRecord(e1, 0); (void)hipDeviceSynchronize();
                    float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
                }
                const double bytes = (double)tm * tn * (K / 64) * 65536.0;
                const int instr_per_step = rowb == 128 ? 16 : 8;
                printf("N=%4d K=%4d rows %3d B, %d steps (%2d instr = %3d KB) in flight: %7.1f us  %6.2f TB/s L2->LDS  = %.2f us per 64 KB K-tile per CU-round\n", N, K, rowb, depth,
                       instr_per_step * depth, instr_per_step * depth * 4, best * 1e3, bytes / (best * 1e-3) / 1e12, best * 1e3 / ((double)((tm * tn + 255) / 256) * (K / 64)));
            }
        (void)hipFree(A); (void)hipFree(B);
    }
    return 0;
}
