// Diagnostic (not part of the product): does the GEMM main loop's instruction mix — 64 MFMA 16x16x32 bf16 + 24 independent
// ds_read_b128 per wave per K-tile, 8 waves per CU — run slower per CU when all 256 CUs run it than when 8 do?  Host-side event
// timing (the in-kernel s_memtime / s_memrealtime counters are NOT the shader clock / 100 MHz on this part).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
union Chunk { uint4 u; bf16x8 h; };

template <int NREAD>
__global__ __launch_bounds__(512, 2) void loop(int iters, float* sink) {
    __shared__ __attribute__((aligned(16))) char lds[131072];
    for (int i = threadIdx.x; i < 131072 / 4; i += 512) ((float*)lds)[i] = 1.0f;
    __syncthreads();
    f32x4 acc[32];
    for (int i = 0; i < 32; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    Chunk fa[8], fb[4];
    for (int i = 0; i < 8; ++i) fa[i].u = uint4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    for (int i = 0; i < 4; ++i) fb[i].u = fa[0].u;
    const int lane_off = (threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 4096;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {      // one K-tile = 2 k-steps of (8 A + 4 B fragment reads, 32 MFMAs)
            if (NREAD) {
#pragma unroll
                for (int i = 0; i < 8; ++i) fa[i].u = *reinterpret_cast<const uint4*>(lds + ((lane_off + (i + 8 * ks) * 1024 + it * 16) & 131056));
#pragma unroll
                for (int i = 0; i < 4; ++i) fb[i].u = *reinterpret_cast<const uint4*>(lds + ((lane_off + 65536 + (i + 4 * ks) * 1024 + it * 16) & 131056));
            }
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int mi = 0; mi < 8; ++mi) acc[ni * 8 + mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[ni].h, fa[mi].h, acc[ni * 8 + mi], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 32; ++i) s += acc[i][0] + acc[i][3];
    if (s == 12345.678f) sink[0] = s;
}

int main() {
    float* sink;
    (void)hipMalloc(&sink, 8);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 20000;    // K-tiles per wave
    for (int nread = 0; nread < 2; ++nread)
        for (int G : {8, 64, 256}) {
            float ms = 0.f;
            for (int rep = 0; rep < 3; ++rep) {
                (void)hipEventRecord(e0, 0);
                if (nread) hipLaunchKernelGGL(loop<1>, dim3(G), dim3(512), 0, 0, iters, sink);
                else hipLaunchKernelGGL(loop<0>, dim3(G), dim3(512), 0, 0, iters, sink);
                (void)hipEventRecord(e1, 0);
                (void)hipDeviceSynchronize();
                (void)hipEventElapsedTime(&ms, e0, e1);
            }
            const double us_per_ktile = ms * 1e3 / iters;
            const double tf = (double)G * 8 * 64 * 16384.0 * iters / (ms * 1e-3) / 1e12;
            printf("ds_reads=%d workgroups=%3d: %.2f ms, %.3f us per K-tile (64 MFMA/wave), %.0f TFLOP/s = %.0f at 256 CUs\n", nread, G, ms, us_per_ktile, tf, tf * 256.0 / G);
        }
    return 0;
}
