// alloc_probe.hip -- what device allocations cost on this box: hipMalloc by size, hipMalloc after hipFree, virtual-memory reservations mapped granule by granule.
// hipcc --offload-arch=gfx950 -O2 -o /tmp/alloc_probe tools/alloc_probe.hip && /tmp/alloc_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
    CK(hipSetDevice(0));
    void *w = nullptr;
    CK(hipMalloc(&w, 1 << 20));
    for (size_t mb : {64, 256, 1024, 4096, 1024, 256}) {
        void *p = nullptr;
        double t0 = now();
        CK(hipMalloc(&p, mb << 20));
        double t1 = now();
        CK(hipMemset(p, 1, mb << 20));
        CK(hipDeviceSynchronize());
        double t2 = now();
        CK(hipFree(p));
        double t3 = now();
        std::printf("hipMalloc %5zu MB: %.1f ms, first memset %.1f ms, hipFree %.1f ms\n", mb, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3);
    }
    {   // virtual memory: reserve 8 GB, map 256 MB granules one by one
        hipMemAllocationProp prop = {};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = 0;
        size_t gran = 0;
        hipError_t e = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended);
        std::printf("granularity: %s %zu\n", hipGetErrorString(e), gran);
        if (e == hipSuccess) {
            void *va = nullptr;
            double t0 = now();
            e = hipMemAddressReserve(&va, 8ull << 30, 0, nullptr, 0);
            std::printf("reserve 8 GB: %s %.2f ms\n", hipGetErrorString(e), (now() - t0) * 1e3);
            if (e == hipSuccess) {
                const size_t piece = 256ull << 20;
                for (int k = 0; k < 4; ++k) {
                    hipMemGenericAllocationHandle_t h;
                    double a = now();
                    CK(hipMemCreate(&h, piece, &prop, 0));
                    double b = now();
                    CK(hipMemMap((char *)va + k * piece, piece, 0, h, 0));
                    hipMemAccessDesc acc = {};
                    acc.location = prop.location;
                    acc.flags = hipMemAccessFlagsProtReadWrite;
                    CK(hipMemSetAccess((char *)va + k * piece, piece, &acc, 1));
                    double c = now();
                    CK(hipMemset((char *)va + k * piece, 2, piece));
                    CK(hipDeviceSynchronize());
                    std::printf("granule %d (256 MB): create %.1f ms, map + access %.1f ms, memset %.1f ms\n", k, (b - a) * 1e3, (c - b) * 1e3, (now() - c) * 1e3);
                }
            }
        }
    }
    {   // two threads allocating side by side
        double t0 = now();
        void *a = nullptr, *b = nullptr;
        CK(hipMalloc(&a, 1024ull << 20));
        CK(hipMalloc(&b, 1024ull << 20));
        std::printf("two x 1 GB back to back: %.1f ms\n", (now() - t0) * 1e3);
        void *pin = nullptr;
        t0 = now();
        CK(hipHostMalloc(&pin, 256ull << 20, hipHostMallocDefault));
        std::printf("hipHostMalloc 256 MB: %.1f ms\n", (now() - t0) * 1e3);
    }
    return 0;
}
