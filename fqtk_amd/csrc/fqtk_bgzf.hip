// fqtk_bgzf.hip -- device side and C ABI (include/fqtk_bgzf.h) of the BGZF block compressor.
// The algorithm lives in bgzf_deflate.hpp (phase functions shared with the CPU test-suite); this file runs
// the phases of one block on one 1024-lane workgroup with barriers in between.
#include <hip/hip_runtime.h>

#include <new>
#include <string>

#include "../../include/fqtk_bgzf.h"
#include "../../include/fqtk_match.h"
#include "bgzf_deflate.hpp"
#include "bgzf_internal.hpp"

namespace fqtk {
namespace bgzf {

#ifdef FQTK_BGZF_PHASE_TIMES
// Developer build (tools/bgzf_phases.sh): 100 MHz ticks spent in each phase, summed over all blocks by lane 0.
__device__ unsigned long long g_phase_ticks[12];   // [10] = the parallel part of the code construction
#define FQTK_PHASE_MARK(k) do { if (lane == 0) { const uint64_t now = wall_clock64(); atomicAdd(&g_phase_ticks[k], (unsigned long long)(now - t_mark)); t_mark = now; } } while (0)
#else
#define FQTK_PHASE_MARK(k) do { } while (0)
#endif

__global__ __launch_bounds__(kLanes) __attribute__((amdgpu_waves_per_eu(kLanes / 256, kLanes / 256)))   // one workgroup per CU (LDS): registers are free
void deflate_kernel(const fqtk_bgzf_block *blocks, uint32_t n_blocks, const uint32_t *n_blocks_dev,
                                                         uint32_t *out_len, uint32_t *crc_out, uint32_t *tok_all, uint32_t level) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem_raw[];
    Shared &S = *reinterpret_cast<Shared *>(smem_raw);
    const int lane = (int)threadIdx.x;
    uint32_t *tok = tok_all + (size_t)blockIdx.x * kTokensPerBlock;   // token scratch of this workgroup
#ifdef FQTK_BGZF_PHASE_TIMES
    uint64_t t_mark = wall_clock64();
#endif
    if (n_blocks_dev) n_blocks = *n_blocks_dev;   // (the record pipeline learns the count on the device)
    if (blockIdx.x >= n_blocks) return;
    if (lane == 0) S.effort = effort_of_level(level);
    if (crc_out) crc_tables(S, lane);
    __syncthreads();
    for (uint32_t j = blockIdx.x; j < n_blocks; j += gridDim.x) {
        const uint8_t *in = blocks[j].in;
        uint8_t *out = blocks[j].out;
        const uint32_t n = blocks[j].n_in;
        phase_load(S, lane, in, n);
        __syncthreads();
        if (crc_out) {
            phase_crc(S, lane, n);
            __syncthreads();
            if (lane < 64) {   // the first wave folds the lanes' values (phase_crc_fold is the one-lane form of the CPU tests)
                uint32_t c = 0;
                for (int k = 0; k < kLanes / 64; ++k) c ^= S.lane_bits[lane + 64 * k];
                for (int d = 32; d >= 1; d >>= 1) c ^= (uint32_t)__shfl_xor((int)c, d);
                if (lane == 0) crc_out[j] = c;
            }
        }
        FQTK_PHASE_MARK(0);
        if (level == 0) {   // --compression-level 0: stored blocks (RFC 1951 3.2.4), as libdeflate's level 0
            if (lane == 0) S.stored = 1u;
            __syncthreads();
            const uint32_t raw = phase_store(S, lane, in, n, out);
            if (lane == 0) out_len[j] = raw;
            __syncthreads();
            continue;
        }
        phase_count(S, lane, n);
        __syncthreads();
        phase_literal_costs(S, lane, n);
        __syncthreads();
        FQTK_PHASE_MARK(2);
        const uint64_t cheap_mask = phase_index(S, lane, n);
        __syncthreads();
        FQTK_PHASE_MARK(1);
        phase_lz(S, lane, n, tok, cheap_mask);
        __syncthreads();
        {
            uint32_t span;
            phase_reach(S, lane, tok, &span);
            __syncthreads();   // every lane has read its neighbours' ends
            S.span[lane] = span;
        }
        __syncthreads();
        FQTK_PHASE_MARK(3);
        phase_clear_out(S, lane);
        __syncthreads();
        FQTK_PHASE_MARK(4);
        phase_code_lengths(S, lane);        // two lanes: the serial part of the code construction
        __syncthreads();
        FQTK_PHASE_MARK(5);
        phase_codes(S, lane);
        __syncthreads();
        phase_cl_runs(S, lane);
        __syncthreads();
        phase_cl_emit(S, lane);
        __syncthreads();
        if (lane == 0) phase_cl_code(S);        // one lane builds the 19-symbol code ...
        phase_count_bits(S, lane, n, tok);      // ... while all lanes add up the bits of their tokens (needs the two big codes only)
        __syncthreads();
        FQTK_PHASE_MARK(6);
        phase_cl_bits(S, lane);
        __syncthreads();
        FQTK_PHASE_MARK(10);
        {   // exclusive prefix sum of the lanes' bit counts (phase_offsets is the one-lane form of the CPU tests)
            const uint32_t mine = S.lane_bits[lane];
            uint32_t incl = mine;
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t up = (uint32_t)__shfl_up((int)incl, d);
                if ((lane & 63) >= d) incl += up;
            }
            if ((lane & 63) == 63) S.wave_tot[lane >> 6] = incl;
            __syncthreads();
            uint32_t before = S.header_bits;
            for (int w = 0; w < (lane >> 6); ++w) before += S.wave_tot[w];
            S.lane_bits[lane] = before + incl - mine;
            if (lane == kLanes - 1) {
                S.total_bits = before + incl + S.len_ll[256];
                S.stored = ((S.total_bits + 7) >> 3) >= n + 5 ? 1u : 0u;
            }
        }
        __syncthreads();
        FQTK_PHASE_MARK(7);
        phase_emit(S, lane, n, tok);
        __syncthreads();
        FQTK_PHASE_MARK(8);
        const uint32_t bytes = phase_store(S, lane, in, n, out);
        if (lane == 0) out_len[j] = bytes;
        __syncthreads();   // S is reused by the next block
        FQTK_PHASE_MARK(9);
    }
}
#undef FQTK_PHASE_MARK

}  // namespace bgzf
}  // namespace fqtk

namespace fqtk {
namespace bgzf {
size_t deflate_token_bytes_per_group() { return (size_t)kTokensPerBlock * sizeof(uint32_t); }
hipError_t deflate_prepare() {
    return hipFuncSetAttribute(reinterpret_cast<const void *>(deflate_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(Shared));
}
hipError_t deflate_launch(hipStream_t stream, uint32_t groups, const fqtk_bgzf_block *blocks, const uint32_t *n_blocks_dev,
                          uint32_t *out_len, uint32_t *crc, uint32_t *tok, int level) {
    hipLaunchKernelGGL(deflate_kernel, dim3(groups), dim3(kLanes), sizeof(Shared), stream, blocks, 0u, n_blocks_dev, out_len, crc, tok,
                       level <= 0 ? 0u : (uint32_t)level);
    return hipGetLastError();
}
}  // namespace bgzf
}  // namespace fqtk

namespace {
thread_local std::string g_bgzf_error;
int bfail(int code, const std::string &msg) { g_bgzf_error = msg; return code; }
#define BGZF_TRY(expr)                                                                       \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) return bfail(FQTK_EHIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)
}  // namespace

struct fqtk_bgzf {
    int device = 0;
    int num_cus = 256;
    hipStream_t streams[FQTK_BGZF_SLOTS] = {};
    uint32_t *d_tok[FQTK_BGZF_SLOTS] = {};   // per slot: one token scratch per resident workgroup
    bool busy[FQTK_BGZF_SLOTS] = {};
};

extern "C" {

#ifdef FQTK_BGZF_PHASE_TIMES
int fqtk_bgzf_dev_phase_ticks(unsigned long long *out12) {
    return hipMemcpyFromSymbol(out12, HIP_SYMBOL(fqtk::bgzf::g_phase_ticks), 12 * sizeof(unsigned long long)) == hipSuccess ? 0 : 1;
}
int fqtk_bgzf_dev_lz_cycles(unsigned long long *out10) {
    return hipMemcpyFromSymbol(out10, HIP_SYMBOL(fqtk::bgzf::g_lz_cycles), 10 * sizeof(unsigned long long)) == hipSuccess ? 0 : 1;
}
#endif

const char *fqtk_bgzf_last_error(void) { return g_bgzf_error.c_str(); }

int fqtk_bgzf_create(int device, fqtk_bgzf **out) {
    if (!out) return bfail(FQTK_EINVAL, "out is NULL");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        (void)hipGetLastError();
        return bfail(FQTK_ENODEV, "no HIP device available (the BGZF compressor has no CPU fallback)");
    }
    if (device < 0 || device >= ndev) return bfail(FQTK_ENODEV, "device index out of range");
    BGZF_TRY(hipSetDevice(device));
    fqtk_bgzf *z = new (std::nothrow) fqtk_bgzf();
    if (!z) return bfail(FQTK_ENOMEM, "out of host memory");
    z->device = device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) z->num_cus = prop.multiProcessorCount;
    const void *fn = reinterpret_cast<const void *>(fqtk::bgzf::deflate_kernel);
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(fqtk::bgzf::Shared)) != hipSuccess) {
        delete z;
        return bfail(FQTK_EHIP, "cannot reserve LDS for the BGZF kernel");
    }
    for (int s = 0; s < FQTK_BGZF_SLOTS; ++s) {
        if (hipStreamCreateWithFlags(&z->streams[s], hipStreamNonBlocking) != hipSuccess ||
            hipMalloc(reinterpret_cast<void **>(&z->d_tok[s]), (size_t)z->num_cus * fqtk::bgzf::kTokensPerBlock * sizeof(uint32_t)) != hipSuccess) {
            fqtk_bgzf_destroy(z);
            return bfail(FQTK_EHIP, "cannot allocate the BGZF compressor's streams / scratch");
        }
    }
    *out = z;
    return FQTK_OK;
}

void fqtk_bgzf_destroy(fqtk_bgzf *z) {
    if (!z) return;
    (void)hipSetDevice(z->device);
    for (int s = 0; s < FQTK_BGZF_SLOTS; ++s) {
        if (z->streams[s]) { (void)hipStreamSynchronize(z->streams[s]); (void)hipStreamDestroy(z->streams[s]); }
        if (z->d_tok[s]) (void)hipFree(z->d_tok[s]);
    }
    delete z;
}

int fqtk_bgzf_deflate_enqueue(fqtk_bgzf *z, int slot, const fqtk_bgzf_block *blocks, uint32_t n, uint32_t *out_len) {
    if (!z) return bfail(FQTK_EINVAL, "compressor is NULL");
    if (slot < 0 || slot >= FQTK_BGZF_SLOTS) return bfail(FQTK_EINVAL, "slot out of range");
    if (z->busy[slot]) return bfail(FQTK_EINVAL, "slot is busy: call fqtk_bgzf_wait() first");
    if (n == 0) return FQTK_OK;
    if (!blocks || !out_len) return bfail(FQTK_EINVAL, "blocks / out_len is NULL");
    BGZF_TRY(hipSetDevice(z->device));
    const uint32_t grid = n < (uint32_t)z->num_cus ? n : (uint32_t)z->num_cus;   // one workgroup per CU (107 KiB of LDS each)
    hipLaunchKernelGGL(fqtk::bgzf::deflate_kernel, dim3(grid), dim3(fqtk::bgzf::kLanes), sizeof(fqtk::bgzf::Shared),
                       z->streams[slot], blocks, n, (const uint32_t *)nullptr, out_len, (uint32_t *)nullptr, z->d_tok[slot], 5u);
    BGZF_TRY(hipGetLastError());
    z->busy[slot] = true;
    return FQTK_OK;
}

int fqtk_bgzf_wait(fqtk_bgzf *z, int slot) {
    if (!z) return bfail(FQTK_EINVAL, "compressor is NULL");
    if (slot < 0 || slot >= FQTK_BGZF_SLOTS) return bfail(FQTK_EINVAL, "slot out of range");
    if (!z->busy[slot]) return FQTK_OK;
    BGZF_TRY(hipSetDevice(z->device));
    BGZF_TRY(hipStreamSynchronize(z->streams[slot]));
    z->busy[slot] = false;
    return FQTK_OK;
}

}  // extern "C"
