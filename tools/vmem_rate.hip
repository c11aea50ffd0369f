// Developer microbenchmark: what ONE vector-memory instruction costs a CU (the per-CU address / data path every wave of
// the CU shares), by kind: 2-byte gathers over a 2 MiB table (all lanes, a quarter of the lanes exec-masked, a quarter
// with the others parked on one dummy address, all lanes on one address), coalesced dword / dwordx3 / dwordx4 loads
// that hit in L2, 16-byte gathers.  32 waves per CU, eight independent loads in flight per wave.
// build+run: hipcc --offload-arch=gfx950 -O3 tools/vmem_rate.hip -o /tmp/vmem_rate && /tmp/vmem_rate
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(4))) u32x3s { uint32_t x, y, z; };

template <int MODE>
__global__ __launch_bounds__(1024) void k(const uint8_t *tab, uint32_t *out, int iters, unsigned long long *cyc) {
    const uint32_t lane = threadIdx.x & 63u, gw = (blockIdx.x * 1024u + threadIdx.x) >> 6;
    uint32_t x = (blockIdx.x * 1024u + threadIdx.x) * 2654435761u + 12345u;
    uint32_t acc = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        uint32_t v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            x = x * 1664525u + 1013904223u;
            const uint32_t idx = (x >> 11) & 0xFFFFFu;   // 2^20 entries
            if (MODE == 0) v[u] = reinterpret_cast<const uint16_t *>(tab)[idx];
            if (MODE == 1) { v[u] = 0; if ((lane & 3u) == 0) v[u] = reinterpret_cast<const uint16_t *>(tab)[idx]; }
            if (MODE == 2) v[u] = reinterpret_cast<const uint16_t *>(tab)[(lane & 3u) == 0 ? idx : 0u];
            if (MODE == 3) v[u] = reinterpret_cast<const uint16_t *>(tab)[__builtin_amdgcn_readfirstlane(idx)];
            const uint32_t wu = __builtin_amdgcn_readfirstlane(idx);   // a wave-uniform random row of the table
            if (MODE == 4) v[u] = reinterpret_cast<const uint32_t *>(tab)[(wu & 0x3FFFu) * 64u + lane];
            if (MODE == 5) { const u32x4v q = reinterpret_cast<const u32x4v *>(tab)[(wu & 0xFFFu) * 64u + lane]; v[u] = q.x ^ q.y ^ q.z ^ q.w; }
            if (MODE == 6) { const u32x3s q = reinterpret_cast<const u32x3s *>(tab)[(wu & 0xFFFu) * 64u + lane]; v[u] = q.x ^ q.y ^ q.z; }
            if (MODE == 9) { const uint2 q = reinterpret_cast<const uint2 *>(tab)[(wu & 0x1FFFu) * 64u + lane]; v[u] = q.x ^ q.y; }
            if (MODE == 7) { const u32x4v q = reinterpret_cast<const u32x4v *>(tab)[idx >> 3]; v[u] = q.x ^ q.y ^ q.z ^ q.w; }
            if (MODE == 8) { v[u] = 0; if (lane == 0) v[u] = reinterpret_cast<const uint16_t *>(tab)[idx]; }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc ^= v[u];
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 1024u + threadIdx.x] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int MODE>
void run(const char *name, const uint8_t *tab, uint32_t *out, unsigned long long *cyc, int cus, int blocks = 0) {
    const int iters = 400;
    if (!blocks) blocks = cus * 2;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(1024), 0, 0, tab, out, 20, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(1024), 0, 0, tab, out, iters, cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c = 0;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double instr = 16.0 * blocks * iters * 8;   // wave-instructions
    printf("%-58s %4d workgroups  %7.3f ms  %7.2f G wave-instr/s on the chip  %6.1f M per workgroup\n", name, blocks, ms, instr / ms / 1e6, instr / blocks / ms / 1e3);
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    uint8_t *tab;
    uint32_t *out;
    unsigned long long *cyc;
    hipMalloc(&tab, 4u << 20);
    hipMemset(tab, 1, 4u << 20);
    hipMalloc(&out, (size_t)cus * 2 * 1024 * 4);
    hipMalloc(&cyc, 8);
    run<0>("2-byte gather, 2 MiB table, 64 lanes", tab, out, cyc, cus);
    run<1>("2-byte gather, 16 lanes (the rest exec-masked)", tab, out, cyc, cus);
    run<2>("2-byte gather, 16 lanes (the rest on one dummy address)", tab, out, cyc, cus);
    run<8>("2-byte gather, 1 lane (the rest exec-masked)", tab, out, cyc, cus);
    run<3>("2-byte load, 64 lanes on one address", tab, out, cyc, cus);
    run<4>("dword, coalesced, L2-resident", tab, out, cyc, cus);
    run<6>("dwordx3 at 12-byte stride, L2-resident", tab, out, cyc, cus);
    run<5>("dwordx4, coalesced, L2-resident", tab, out, cyc, cus);
    run<9>("dwordx2, coalesced, L2-resident", tab, out, cyc, cus);
    run<7>("16-byte gather, 2 MiB table, 64 lanes", tab, out, cyc, cus);
    for (int b : {cus, cus / 2, cus / 4, cus / 8, 8, 1}) run<0>("2-byte gather, 64 lanes", tab, out, cyc, cus, b);
    for (int b : {cus, cus / 8}) run<5>("dwordx4 coalesced", tab, out, cyc, cus, b);
    for (int b : {cus, cus / 8}) run<6>("dwordx3 at 12-byte stride", tab, out, cyc, cus, b);
    return 0;
}
