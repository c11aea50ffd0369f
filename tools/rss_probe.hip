#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <chrono>
#include <time.h>
static size_t rss(const char *key) { size_t kb = 0; FILE *f = fopen("/proc/self/status", "r"); char b[256]; while (fgets(b, 256, f)) if (!strncmp(b, key, strlen(key))) kb = strtoull(b + strlen(key), 0, 10); fclose(f); return kb >> 10; }
static double epoch() { timespec ts; clock_gettime(CLOCK_REALTIME, &ts); return ts.tv_sec + ts.tv_nsec * 1e-9; }
__global__ void k(int *p) { p[threadIdx.x] = 1; }
int main(int argc, char **argv) {
    const int n_streams = argc > 1 ? atoi(argv[1]) : 0, prio = argc > 2 ? atoi(argv[2]) : 0, vram_gb = argc > 3 ? atoi(argv[3]) : 0, pinned_mb = argc > 4 ? atoi(argv[4]) : 0;
    const double t0 = epoch();
    hipSetDevice(0);
    int *d; hipMalloc(&d, 4096);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d); hipDeviceSynchronize();
    hipStream_t s[64];
    int lo = 0, hi = 0; hipDeviceGetStreamPriorityRange(&lo, &hi);
    for (int i = 0; i < n_streams; ++i) {
        if (prio) hipStreamCreateWithPriority(&s[i], hipStreamNonBlocking, i % 3 == 0 ? hi : (i % 3 == 1 ? lo : 0));
        else hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s[i], d);
        hipStreamSynchronize(s[i]);
    }
    for (int g = 0; g < vram_gb; ++g) { void *p; hipMalloc(&p, 1ull << 30); hipMemset(p, 1, 1ull << 30); }
    if (pinned_mb) { void *p; hipHostMalloc(&p, (size_t)pinned_mb << 20, hipHostMallocDefault); memset(p, 1, (size_t)pinned_mb << 20); }
    hipDeviceSynchronize();
    printf("vram %d GB pinned %d MB; streams %d prio %d (range %d..%d): anon %zu MB, main took %.3f s, ends at epoch %.3f\n", vram_gb, pinned_mb, n_streams, prio, lo, hi, rss("RssAnon:"), epoch() - t0, epoch());
    fflush(stdout);
    _Exit(0);
}
