// Developer microbenchmark: the LDS-form kernel's loop SKELETON without the matcher -- one 16-byte non-temporal
// load and one 4-byte non-temporal store per lane, 1024-lane workgroups, one per CU, persistent grid-stride tiles --
// with a dial for the work between load and store (C rounds of ~12 VALU + one dependent LDS read) and the loop
// shape (plain / software-pipelined one tile deep, as lds_memo_kernel), and per-workgroup start/finish times
// (how uneven is the tail?).
// build+run: hipcc --offload-arch=gfx950 -O3 tools/stream_skeleton.hip -o /tmp/stream_skeleton && /tmp/stream_skeleton
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int C>
__device__ __forceinline__ uint32_t work(const u32x4 v, const uint32_t *lds, uint32_t mask) {
    uint32_t x = v.x ^ (v.y * 3u) ^ (v.z * 5u) ^ (v.w * 7u);
#pragma unroll
    for (int c = 0; c < C; ++c) {
        uint32_t h = x * 0x9E3779B1u;
        h ^= h >> 15; h *= 0x85EBCA6Bu; h ^= h >> 13; h += x; h ^= h << 3; h ^= h >> 11;
        x = h + lds[(h >> 5) & mask];
    }
    return x;
}

template <int C, bool PF>
__global__ __launch_bounds__(1024) void kskel(const uint8_t *in, uint8_t *out, uint64_t n, uint32_t lds_words, uint64_t *stamps) {
    extern __shared__ uint32_t lds[];
    const uint32_t tid = threadIdx.x;
    const uint64_t t0 = wall_clock64();
    for (uint32_t w = tid; w < lds_words; w += 1024) lds[w] = w * 2654435761u;
    __syncthreads();
    const uint32_t mask = lds_words - 1;
    const uint64_t nt = n / 1024;
    const u32x4 *src = reinterpret_cast<const u32x4 *>(in);
    uint32_t *dst = reinterpret_cast<uint32_t *>(out);
    if constexpr (!PF) {
        for (uint64_t t = blockIdx.x; t < nt; t += gridDim.x) {
            const u32x4 v = __builtin_nontemporal_load(src + t * 1024 + tid);
            __builtin_nontemporal_store(work<C>(v, lds, mask), dst + t * 1024 + tid);
        }
    } else {
        auto pin4 = [](u32x4 &v) { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w) : : "memory"); };
        auto pin1 = [](uint32_t &v) { asm volatile("" : "+v"(v) : : "memory"); };
        uint64_t t = blockIdx.x;
        if (t < nt) {
            const uint64_t last = nt - 1;
            u32x4 a = __builtin_nontemporal_load(src + t * 1024 + tid), b;
            pin4(a);
            b = __builtin_nontemporal_load(src + std::min<uint64_t>(t + gridDim.x, last) * 1024 + tid);
            uint32_t held = work<C>(a, lds, mask);
            pin1(held);
            uint64_t th = t;
            t += gridDim.x;
            while (t < nt) {
                pin4(b);
                a = __builtin_nontemporal_load(src + std::min<uint64_t>(t + gridDim.x, last) * 1024 + tid);
                __builtin_nontemporal_store(held, dst + th * 1024 + tid);
                held = work<C>(b, lds, mask);
                pin1(held);
                th = t;
                t += gridDim.x;
                if (t >= nt) break;
                pin4(a);
                b = __builtin_nontemporal_load(src + std::min<uint64_t>(t + gridDim.x, last) * 1024 + tid);
                __builtin_nontemporal_store(held, dst + th * 1024 + tid);
                held = work<C>(a, lds, mask);
                pin1(held);
                th = t;
                t += gridDim.x;
            }
            __builtin_nontemporal_store(held, dst + th * 1024 + tid);
        }
    }
    if (tid == 0) { stamps[2 * blockIdx.x] = t0; stamps[2 * blockIdx.x + 1] = wall_clock64(); }
}

// Wider stores: every lane loads R 16-byte rows (wave-contiguous: a wave's r-th load covers 1 KiB) and stores ONE
// vector of R dwords -- as if the R results had been transposed across the wave -- so a wave's store covers
// R x 256 contiguous bytes instead of 256.  No work; does the memory system like the wider write bursts?
template <int R>
__global__ __launch_bounds__(1024) void kwide(const uint8_t *in, uint8_t *out, uint64_t n) {
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const uint64_t tile = 1024ull * R, nt = n / tile;
    const u32x4 *src = reinterpret_cast<const u32x4 *>(in);
    for (uint64_t t = blockIdx.x; t < nt; t += gridDim.x) {
        const uint64_t base = t * tile + (uint64_t)wave * 64 * R;
        uint32_t x[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const u32x4 v = __builtin_nontemporal_load(src + base + r * 64 + lane);
            x[r] = v.x ^ v.y ^ v.z ^ v.w;
        }
        uint32_t *dst = reinterpret_cast<uint32_t *>(out) + base + (uint64_t)lane * R;
        if constexpr (R == 4) {
            u32x4 o = {x[0], x[1], x[2], x[3]};
            __builtin_nontemporal_store(o, reinterpret_cast<u32x4 *>(dst));
        } else if constexpr (R == 2) {
            typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
            u32x2 o = {x[0], x[1]};
            __builtin_nontemporal_store(o, reinterpret_cast<u32x2 *>(dst));
        } else {
            __builtin_nontemporal_store(x[0], dst);
        }
    }
}
template <int R>
void run_wide(const uint8_t *in, uint8_t *out, uint64_t n, int grid) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kwide<R>, dim3(grid), dim3(1024), 0, 0, in, out, n);
    hipEventRecord(a);
    const int reps = 10;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(kwide<R>, dim3(grid), dim3(1024), 0, 0, in, out, n);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    ms /= reps;
    printf("%d rows per lane, one %2d-byte store per lane, %d workgroups: %.3f ms  %.0f GB/s\n", R, 4 * R, grid, ms, n * 20.0 / ms / 1e6);
}

template <int C, bool PF>
void run(const uint8_t *in, uint8_t *out, uint64_t n, int grid, uint32_t lds_words, uint64_t *d_stamps) {
    auto kern = kskel<C, PF>;
    hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const size_t shmem = (size_t)lds_words * 4;
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kern, dim3(grid), dim3(1024), shmem, 0, in, out, n, lds_words, d_stamps);
    hipEventRecord(a);
    const int reps = 10;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(kern, dim3(grid), dim3(1024), shmem, 0, in, out, n, lds_words, d_stamps);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    ms /= reps;
    std::vector<uint64_t> st(2 * grid);
    hipMemcpy(st.data(), d_stamps, st.size() * 8, hipMemcpyDeviceToHost);
    uint64_t s0 = ~0ull, s1 = 0, e0 = ~0ull, e1 = 0;
    for (int g = 0; g < grid; ++g) { s0 = std::min(s0, st[2 * g]); s1 = std::max(s1, st[2 * g]); e0 = std::min(e0, st[2 * g + 1]); e1 = std::max(e1, st[2 * g + 1]); }
    printf("work rounds %d, %-9s LDS %3u KB: %.3f ms  %.0f GB/s | workgroups start within %.1f us, first done %.1f us before the last (of %.1f us)\n",
           C, PF ? "pipelined" : "plain", lds_words / 256, ms, n * 20.0 / ms / 1e6, (s1 - s0) / 100.0, (e1 - e0) / 100.0, (e1 - s0) / 100.0);
}

int main() {
    const uint64_t n = 400000000ull;
    uint8_t *in, *out;
    uint64_t *d_stamps;
    if (hipMalloc(&in, n * 16) != hipSuccess || hipMalloc(&out, n * 4) != hipSuccess || hipMalloc(&d_stamps, 16 * 4096) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(in, 0x41, n * 16);
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    const uint32_t big = 32768;   // 128 KB: one workgroup per CU, as cfg 3's table
    run<0, false>(in, out, n, cus, big, d_stamps);
    run<0, true>(in, out, n, cus, big, d_stamps);
    run<1, false>(in, out, n, cus, big, d_stamps);
    run<1, true>(in, out, n, cus, big, d_stamps);
    run<2, false>(in, out, n, cus, big, d_stamps);
    run<2, true>(in, out, n, cus, big, d_stamps);
    run<4, false>(in, out, n, cus, big, d_stamps);
    run<4, true>(in, out, n, cus, big, d_stamps);
    run<8, true>(in, out, n, cus, big, d_stamps);
    printf("-- wider stores (no LDS, no work)\n");
    run_wide<1>(in, out, n, cus);
    run_wide<2>(in, out, n, cus);
    run_wide<4>(in, out, n, cus);
    run_wide<1>(in, out, n, 2 * cus);
    run_wide<2>(in, out, n, 2 * cus);
    run_wide<4>(in, out, n, 2 * cus);
    printf("-- two workgroups per CU (64 KB each)\n");
    run<0, false>(in, out, n, 2 * cus, 16384, d_stamps);
    run<2, false>(in, out, n, 2 * cus, 16384, d_stamps);
    run<2, true>(in, out, n, 2 * cus, 16384, d_stamps);
    run<4, true>(in, out, n, 2 * cus, 16384, d_stamps);
    return 0;
}
