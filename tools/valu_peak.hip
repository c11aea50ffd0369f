// Developer microbenchmark: peak rate of the integer VALU ops the scan kernel is made of
// (v_and_or_b32 / v_bcnt_u32_b32 / v_min_u32 / v_med3_u32 / v_lshl_or_b32) on gfx950.
// build+run: hipcc --offload-arch=gfx950 -O3 tools/valu_peak.hip -o /tmp/valu_peak && /tmp/valu_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int MODE>
__global__ __launch_bounds__(256) void k(uint32_t *out, uint32_t seed, int iters) {
    uint32_t a[8];
    for (int i = 0; i < 8; ++i) a[i] = seed * (threadIdx.x + 1 + i);
    uint32_t s0 = seed | 1, s1 = seed | 2;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "s"(s0), "v"(a[(i + 1) & 7]));
                if (MODE == 1) asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
                if (MODE == 2) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
                if (MODE == 3) asm volatile("v_med3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(a[(i + 1) & 7]), "v"(a[(i + 2) & 7]));
                if (MODE == 4) asm volatile("v_lshl_or_b32 %0, %0, 16, %1" : "+v"(a[i]) : "s"(s1));
                if (MODE == 5) asm volatile("v_and_b32 %0, %1, %0" : "+v"(a[i]) : "s"(s0));
                if (MODE == 6) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(a[(i + 1) & 7]), "v"(a[(i + 2) & 7]));
            }
    }
    uint32_t r = 0;
    for (int i = 0; i < 8; ++i) r ^= a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int MODE>
void run(const char *name) {
    const int blocks = 256 * 8, iters = 2000;
    uint32_t *d;
    hipMalloc(&d, blocks * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 12345u, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 12345u, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double ops = (double)blocks * 256 * iters * 16 * 8;
    printf("%-16s %8.2f T lane-ops/s  (%.3f ms)\n", name, ops / ms / 1e9, ms);
    hipFree(d);
}

int main() {
    run<0>("v_and_or_b32");
    run<1>("v_bcnt_u32_b32");
    run<2>("v_min_u32");
    run<3>("v_med3_u32");
    run<4>("v_lshl_or_b32");
    run<5>("v_and_b32");
    run<6>("v_fma_f32");
    return 0;
}
