// synth.hip -- deterministic synthetic observed-barcode generator (bench / test harness, NOT part of
// the matcher ABI).  Counter-based: read i is a pure function of (seed, i), so the SAME bytes can be
// produced on the device (fqtk_synth_fill_device, for HBM-resident bench inputs) and on the host
// (fqtk_synth_fill_host, for the CPU oracle / cpu_baseline leg) without any transfer.
//
// Workload model: with probability p_sample the read is drawn from a sample
// barcode (sample popularity ~ Zipf, given as a 32-bit CDF), degenerate IUPAC positions resolved to
// a concrete base; otherwise a uniform random ACGT L-mer.  Then per base: no-call 'N' with p_n,
// substitution with p_sub, lower-casing with p_lower, '.' with p_dot, the IUPAC code 'R' with p_iupac.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstring>

struct SynthParams {
    const uint8_t *barcodes;  // [S][L] ASCII, upper-case IUPAC
    const uint32_t *cdf;      // [S] inclusive upper bounds scaled to 2^32 (last = 0xFFFFFFFF)
    uint32_t S, L, stride;
    uint64_t seed;
    uint32_t thr_sample, thr_n, thr_sub, thr_lower, thr_dot, thr_iupac;  // probabilities * 2^32
};

__host__ __device__ inline uint64_t splitmix64(uint64_t &x) {
    uint64_t z = (x += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

__host__ __device__ inline uint8_t iupac_mask(uint8_t b) {
    switch (b) {
        case 'A': return 1; case 'C': return 2; case 'G': return 4; case 'T': return 8; case 'U': return 8;
        case 'M': return 3; case 'R': return 5; case 'W': return 9; case 'S': return 6; case 'Y': return 10;
        case 'K': return 12; case 'V': return 7; case 'H': return 11; case 'D': return 13; case 'B': return 14;
        default: return 15;  // N n . and anything else: any base
    }
}

__host__ __device__ inline void synth_read(const SynthParams &P, uint64_t i, uint8_t *dst) {
    const char ACGT[4] = {'A', 'C', 'G', 'T'};
    uint64_t st = P.seed ^ (i * 0xD1342543DE82EF95ULL + 0x2545F4914F6CDD1DULL);
    const uint64_t r0 = splitmix64(st);
    const bool from_sample = (uint32_t)r0 < P.thr_sample;
    uint32_t s = 0;
    if (from_sample) {
        const uint32_t u = (uint32_t)(r0 >> 32);
        uint32_t lo = 0, hi = P.S - 1;  // first s with cdf[s] >= u
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (P.cdf[mid] >= u) hi = mid; else lo = mid + 1;
        }
        s = lo;
    }
    for (uint32_t k = 0; k < P.L; ++k) {
        const uint64_t r = splitmix64(st);
        const uint32_t pick = (uint32_t)(r >> 60) & 3u;          // 2 bits: base choice
        const uint32_t ev = (uint32_t)r;                         // 32 bits: event
        const uint32_t alt = (uint32_t)(r >> 32) & 0x0FFFFFFFu;  // 28 bits: secondary draws
        uint8_t base;
        if (from_sample) {
            const uint8_t m = iupac_mask(P.barcodes[(uint64_t)s * P.L + k]);
            // pick-th allowed base, cyclically
            const uint32_t cnt = (m & 1) + ((m >> 1) & 1) + ((m >> 2) & 1) + ((m >> 3) & 1);
            uint32_t want = pick % cnt, j = 0;
            for (;; ++j) {
                if ((m >> j) & 1) {
                    if (want == 0) break;
                    --want;
                }
            }
            base = (uint8_t)ACGT[j];
        } else {
            base = (uint8_t)ACGT[pick];
        }
        if (ev < P.thr_n) {
            base = 'N';
        } else if (ev - P.thr_n < P.thr_sub) {
            // substitute with one of the three OTHER bases
            uint32_t cur = base == 'A' ? 0 : base == 'C' ? 1 : base == 'G' ? 2 : 3;
            base = (uint8_t)ACGT[(cur + 1 + (alt % 3)) & 3];
        }
        const uint32_t ev2 = (alt * 2654435761u);
        if (ev2 < P.thr_dot) base = '.';
        else if (ev2 - P.thr_dot < P.thr_lower) base = (uint8_t)(base | 0x20);
        else if (ev2 - P.thr_dot - P.thr_lower < P.thr_iupac) base = 'R';   // an IUPAC code IN THE READ (cliff studies only)
        dst[k] = base;
    }
    for (uint32_t k = P.L; k < P.stride; ++k) dst[k] = 0;
}

__global__ __launch_bounds__(256) void synth_kernel(const SynthParams P, uint64_t start, uint64_t n,
                                                     uint8_t *out) {
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t step = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = tid; i < n; i += step) {
        uint8_t buf[128];
        SynthParams Q = P;
        Q.stride = P.L;  // pad written below
        synth_read(Q, start + i, buf);
        uint8_t *dst = out + i * (uint64_t)P.stride;
        for (uint32_t k = 0; k < P.L; ++k) dst[k] = buf[k];
        for (uint32_t k = P.L; k < P.stride; ++k) dst[k] = 0;
    }
}

extern "C" {

// Fills out[0 .. n*stride) with reads start .. start+n on the HOST.  barcodes: S*L bytes.
int fqtk_synth_fill_host(const uint8_t *barcodes, const uint32_t *cdf, uint32_t S, uint32_t L,
                         uint32_t stride, uint64_t seed, const uint32_t *thr6, uint64_t start,
                         uint64_t n, uint8_t *out) {
    if (L > 128 || stride < L || S == 0) return 1;
    SynthParams P{barcodes, cdf, S, L, stride, seed, thr6[0], thr6[1], thr6[2], thr6[3], thr6[4], thr6[5]};
    for (uint64_t i = 0; i < n; ++i) synth_read(P, start + i, out + i * (uint64_t)stride);
    return 0;
}

// Same bytes, written to DEVICE memory d_out on hip_stream.  Copies the (small) table to the device
// for the duration of the call; synchronises the stream before returning.
int fqtk_synth_fill_device(const uint8_t *barcodes, const uint32_t *cdf, uint32_t S, uint32_t L,
                           uint32_t stride, uint64_t seed, const uint32_t *thr6, uint64_t start,
                           uint64_t n, void *d_out, void *hip_stream) {
    if (L > 128 || stride < L || S == 0) return 1;
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    uint8_t *d_bc = nullptr;
    uint32_t *d_cdf = nullptr;
    if (hipMalloc(reinterpret_cast<void **>(&d_bc), (size_t)S * L) != hipSuccess) return 3;
    if (hipMalloc(reinterpret_cast<void **>(&d_cdf), (size_t)S * 4) != hipSuccess) return 3;
    int rc = 0;
    if (hipMemcpyAsync(d_bc, barcodes, (size_t)S * L, hipMemcpyHostToDevice, stream) != hipSuccess) rc = 3;
    if (hipMemcpyAsync(d_cdf, cdf, (size_t)S * 4, hipMemcpyHostToDevice, stream) != hipSuccess) rc = 3;
    if (rc == 0 && n > 0) {
        SynthParams P{d_bc, d_cdf, S, L, stride, seed, thr6[0], thr6[1], thr6[2], thr6[3], thr6[4], thr6[5]};
        const uint64_t want = (n + 255) / 256;
        const uint32_t grid = (uint32_t)(want < 16384 ? want : 16384);
        hipLaunchKernelGGL(synth_kernel, dim3(grid), dim3(256), 0, stream, P, start, n,
                           static_cast<uint8_t *>(d_out));
        if (hipGetLastError() != hipSuccess) rc = 3;
    }
    if (hipStreamSynchronize(stream) != hipSuccess) rc = 3;
    (void)hipFree(d_bc);
    (void)hipFree(d_cdf);
    return rc;
}

}  // extern "C"
