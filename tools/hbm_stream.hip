// Developer microbenchmark: what the MI355X memory system delivers to kernels shaped like the matcher
// (one 16-byte non-temporal load and one 4-byte non-temporal store per lane, persistent grid-stride
// tiles) -- the practical ceiling that roofline.frac in bench.py should be read against.
//   MODE 0: read 16 B/lane only        MODE 1: read 16 B + write 4 B (the matcher's pattern)
//   MODE 2: copy 16 B -> 16 B          MODE 3: read 8 B + write 4 B (8-base barcodes)
// build+run: hipcc --offload-arch=gfx950 -O3 tools/hbm_stream.hip -o /tmp/hbm_stream && /tmp/hbm_stream
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// LAYOUT 0: lane's r-th read = r * BLOCK + tid (block-strided); 1: wave-contiguous (a wave's R loads cover R KiB in a row);
// 2: R interleaved single-read sweeps (the r-th read belongs to tile t + r * gridDim, each sweep is the R = 1 pattern)
template <int MODE, int BLOCK, int R, int LAYOUT = 0>
__global__ __launch_bounds__(BLOCK) void k(const uint8_t *in, uint8_t *out, uint64_t n) {
    const uint64_t tile = (uint64_t)BLOCK * R;
    uint32_t acc = 0;
    const uint64_t nt1 = n / BLOCK;   // LAYOUT 2 counts single-read tiles
    for (uint64_t t = blockIdx.x; LAYOUT == 2 ? t + (uint64_t)(R - 1) * gridDim.x < nt1 : t < n / tile; t += LAYOUT == 2 ? (uint64_t)gridDim.x * R : gridDim.x) {
        u32x4 v[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint64_t i = LAYOUT == 0 ? t * tile + (uint64_t)r * BLOCK + threadIdx.x
                               : LAYOUT == 1 ? t * tile + (uint64_t)(threadIdx.x >> 6) * 64 * R + r * 64 + (threadIdx.x & 63)
                                             : (t + (uint64_t)r * gridDim.x) * BLOCK + threadIdx.x;
            if (MODE == 3) {
                const u32x2 w = __builtin_nontemporal_load(reinterpret_cast<const u32x2 *>(in) + i);
                v[r] = u32x4{w.x, w.y, 0u, 0u};
            } else {
                v[r] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(in) + i);
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint64_t i = LAYOUT == 0 ? t * tile + (uint64_t)r * BLOCK + threadIdx.x
                               : LAYOUT == 1 ? t * tile + (uint64_t)(threadIdx.x >> 6) * 64 * R + r * 64 + (threadIdx.x & 63)
                                             : (t + (uint64_t)r * gridDim.x) * BLOCK + threadIdx.x;
            const uint32_t x = v[r].x ^ v[r].y ^ v[r].z ^ v[r].w;
            if (MODE == 0) acc ^= x;
            if (MODE == 1 || MODE == 3) __builtin_nontemporal_store(x, reinterpret_cast<uint32_t *>(out) + i);
            if (MODE == 2) __builtin_nontemporal_store(v[r], reinterpret_cast<u32x4 *>(out) + i);
        }
    }
    if (MODE == 0 && acc == 0x12345u) out[0] = 1;
}

// Single-read tiles with a K-deep rotating register prefetch: every wave always has K loads in flight, but
// issues them one per iteration (the R = 1 stream shape) instead of K back to back.
template <int MODE, int BLOCK, int K>
__global__ __launch_bounds__(BLOCK) void kpipe(const uint8_t *in, uint8_t *out, uint64_t n) {
    const uint64_t nt = n / BLOCK;
    u32x4 v[K];
    uint64_t t = blockIdx.x;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const uint64_t tt = t + (uint64_t)k * gridDim.x;
        v[k] = tt < nt ? __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(in) + tt * BLOCK + threadIdx.x) : u32x4{0, 0, 0, 0};
    }
    for (; t < nt; t += (uint64_t)gridDim.x * K) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const uint64_t tt = t + (uint64_t)k * gridDim.x;
            if (tt >= nt) break;
            const uint32_t x = v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
            const uint64_t tn = tt + (uint64_t)K * gridDim.x;
            if (tn < nt) v[k] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(in) + tn * BLOCK + threadIdx.x);
            __builtin_nontemporal_store(x, reinterpret_cast<uint32_t *>(out) + tt * BLOCK + threadIdx.x);
        }
    }
}

template <int MODE, int BLOCK, int K>
void run_pipe(const char *name, const uint8_t *in, uint8_t *out, uint64_t n, int cus, int per_cu, double bytes_per_lane) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const int grid = cus * per_cu;
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((kpipe<MODE, BLOCK, K>), dim3(grid), dim3(BLOCK), 0, 0, in, out, n);
    hipEventRecord(a);
    const int reps = 10;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((kpipe<MODE, BLOCK, K>), dim3(grid), dim3(BLOCK), 0, 0, in, out, n);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    ms /= reps;
    printf("%-44s block %4d, prefetch depth %d, %d/CU: %.3f ms  %.0f GB/s\n", name, BLOCK, K, per_cu, ms, n * bytes_per_lane / ms / 1e6);
}

template <int MODE, int BLOCK, int R, int LAYOUT = 0>
void run(const char *name, const uint8_t *in, uint8_t *out, uint64_t n, int cus, int per_cu, double bytes_per_lane) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const int grid = cus * per_cu;
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k<MODE, BLOCK, R, LAYOUT>), dim3(grid), dim3(BLOCK), 0, 0, in, out, n);
    hipEventRecord(a);
    const int reps = 10;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((k<MODE, BLOCK, R, LAYOUT>), dim3(grid), dim3(BLOCK), 0, 0, in, out, n);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    ms /= reps;
    printf("%-44s block %4d x R %d, layout %d, %d/CU: %.3f ms  %.0f GB/s\n", name, BLOCK, R, LAYOUT, per_cu, ms, n * bytes_per_lane / ms / 1e6);
}

int main() {
    const uint64_t n = 400000000ull;   // lanes (= reads of cfg 3)
    uint8_t *in, *out;
    if (hipMalloc(&in, n * 16) != hipSuccess || hipMalloc(&out, n * 16) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(in, 0x41, n * 16);
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    run<0, 256, 4>("read 16 B", in, out, n, cus, 8, 16);
    run<1, 256, 1>("read 16 B + write 4 B", in, out, n, cus, 8, 20);
    run<1, 256, 4>("read 16 B + write 4 B", in, out, n, cus, 8, 20);
    run<1, 1024, 4>("read 16 B + write 4 B (the LDS kernel's shape)", in, out, n, cus, 1, 20);
    run<1, 1024, 4>("read 16 B + write 4 B", in, out, n, cus, 2, 20);
    run<2, 256, 4>("copy 16 B -> 16 B", in, out, n, cus, 8, 32);
    run<3, 1024, 4>("read 8 B + write 4 B (8-base barcodes)", in, out, n, cus, 2, 12);
    printf("-- shapes for a 1-workgroup-per-CU kernel (read 16 B + write 4 B)\n");
    run<1, 1024, 1>("1024 x 1", in, out, n, cus, 1, 20);
    run<1, 1024, 2>("1024 x 2", in, out, n, cus, 1, 20);
    run<1, 1024, 2, 1>("1024 x 2 wave-contiguous", in, out, n, cus, 1, 20);
    run<1, 1024, 4, 1>("1024 x 4 wave-contiguous", in, out, n, cus, 1, 20);
    run<1, 1024, 8, 1>("1024 x 8 wave-contiguous", in, out, n, cus, 1, 20);
    run<1, 1024, 8>("1024 x 8", in, out, n, cus, 1, 20);
    run<1, 1024, 2, 2>("1024 x 2 interleaved sweeps", in, out, n, cus, 1, 20);
    run<1, 1024, 4, 2>("1024 x 4 interleaved sweeps", in, out, n, cus, 1, 20);
    run_pipe<1, 1024, 2>("1024 x 1, rotating prefetch", in, out, n, cus, 1, 20);
    run_pipe<1, 1024, 4>("1024 x 1, rotating prefetch", in, out, n, cus, 1, 20);
    printf("-- 2 workgroups per CU\n");
    run_pipe<1, 1024, 4>("1024 x 1, rotating prefetch", in, out, n, cus, 2, 20);
    run<1, 1024, 4, 2>("1024 x 4 interleaved sweeps", in, out, n, cus, 2, 20);
    run<3, 1024, 4, 2>("8 B: 1024 x 4 interleaved sweeps", in, out, n, cus, 2, 12);
    run<3, 1024, 2, 2>("8 B: 1024 x 2 interleaved sweeps", in, out, n, cus, 2, 12);
    run<1, 1024, 1>("1024 x 1", in, out, n, cus, 2, 20);
    run<1, 1024, 2>("1024 x 2", in, out, n, cus, 2, 20);
    run<1, 1024, 4, 1>("1024 x 4 wave-contiguous", in, out, n, cus, 2, 20);
    run<3, 1024, 1>("8 B: 1024 x 1", in, out, n, cus, 2, 12);
    run<3, 1024, 2>("8 B: 1024 x 2", in, out, n, cus, 2, 12);
    run<3, 1024, 8>("8 B: 1024 x 8", in, out, n, cus, 2, 12);
    run<3, 1024, 4, 1>("8 B: 1024 x 4 wave-contiguous", in, out, n, cus, 2, 12);
    return 0;
}
