// fqtk_bgzf.hip -- device side and C ABI (include/fqtk_bgzf.h) of the BGZF block compressor.
// The algorithm lives in bgzf_deflate.hpp (phase functions shared with the CPU test-suite); this file runs
// the phases of one block on one 1024-lane workgroup with barriers in between.
#include <hip/hip_runtime.h>

#include <new>
#include <string>

#include "../../include/fqtk_bgzf.h"
#include "../../include/fqtk_match.h"
#ifdef FQTK_BGZF_PHASE_TIMES
// Developer build (tools/bgzf_phases.sh): 100 MHz ticks spent in each phase, summed over all blocks by lane 0.
namespace fqtk {
namespace bgzf {
__device__ unsigned long long g_phase_ticks[16];   // [10] = the parallel part of the code construction, [12] = the LZ phase without phase_reach, [13..15] = marks inside phases
__shared__ unsigned long long s_t_mark;            // (lane 0's last mark)
}
}
#define FQTK_PHASE_MARK(k) do { if (lane == 0) { const unsigned long long now = wall_clock64(); atomicAdd(&fqtk::bgzf::g_phase_ticks[k], now - fqtk::bgzf::s_t_mark); fqtk::bgzf::s_t_mark = now; } } while (0)
#define FQTK_BGZF_MID(k) FQTK_PHASE_MARK(k)   // (a mark inside a phase function of bgzf_deflate.hpp: lane 0's own way through it)
#else
#define FQTK_PHASE_MARK(k) do { } while (0)
#endif
#include "bgzf_deflate.hpp"
#include "bgzf_internal.hpp"

namespace fqtk {
namespace bgzf {

// The code lengths of the two big codes (huffman_lengths of bgzf_deflate.hpp, the one-lane form the CPU tests run), in four
// steps with the serial part cut down to what is serial: (a) the two-queue tree construction by one lane per code -- the
// four queue heads it can need are read together, one LDS round trip per merge instead of four or five; (b) every node's
// depth by a lane of its own, walking up the parents (zlib's length limit puts a node deeper than max_bits AT max_bits and
// its children see that: the depth is min(true depth, max_bits), and every node below the limit is one overflow);
// (c) the repair of an over-subscribed code by one lane; (d) every leaf's length by a lane of its own.
struct CodeJob { const uint16_t *sorted; uint32_t *weight; uint16_t *parent; uint32_t *bl_count; uint8_t *len; uint32_t m; uint32_t *overflow; };
__device__ inline CodeJob code_job(Shared &S, int which) {
    return which == 0 ? CodeJob{S.sorted, S.weight, S.parent, S.bl_count, S.len_ll, S.m_ll, &S.next_code[0]}
                      : CodeJob{S.sorted_d, S.weight_d, S.parent_d, S.bl_count_d, S.len_d, S.m_d, &S.next_code[1]};
}
__device__ void code_tree(const CodeJob &J) {   // (a): one lane
    for (int b = 0; b <= 16; ++b) J.bl_count[b] = 0;
    *J.overflow = 0;
    const uint32_t m = J.m;
    if (m == 0u) { J.len[0] = 1; J.len[1] = 1; J.bl_count[1] = 2; return; }
    if (m == 1u) { J.len[J.sorted[0]] = 1; J.len[J.sorted[0] == 0 ? 1 : 0] = 1; J.bl_count[1] = 2; return; }
    constexpr uint32_t kInf = 0xFFFFFFFFu;
    uint32_t li = 0, ii = m, made = m;
    for (uint32_t k = 0; k + 1u < m; ++k) {
        // the two smallest of the leaf queue's and the internal queue's first two (a leaf wins a tie, as in huffman_lengths)
        // (four UNCONDITIONAL reads -- the array has room past both queues' ends -- and the selection behind them: `li < m ? J.weight[li] : kInf`
        //  is a branch around a read with a wait of its own, and four of them were four dependent round trips per merge: 14 us of a block's 180
        //  on this one lane while 1022 waited)
        const uint32_t rl0 = J.weight[li], rl1 = J.weight[li + 1u], ri0 = J.weight[ii], ri1 = J.weight[ii + 1u];
        uint32_t wl0 = li < m ? rl0 : kInf, wl1 = li + 1u < m ? rl1 : kInf;
        uint32_t wi0 = ii < made ? ri0 : kInf, wi1 = ii + 1u < made ? ri1 : kInf;
        uint32_t pick[2], w[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if (li < m && wl0 <= wi0) { pick[t] = li++; w[t] = wl0; wl0 = wl1; }
            else { pick[t] = ii++; w[t] = wi0; wi0 = wi1; }
        }
        J.weight[made] = w[0] + w[1];
        J.parent[pick[0]] = (uint16_t)made;
        J.parent[pick[1]] = (uint16_t)made;
        ++made;
    }
}
__device__ void code_depth(const CodeJob &J, uint32_t v, uint32_t max_bits) {   // (b): node v < root
    const uint32_t m = J.m;
    if (m < 2u) return;
    const uint32_t root = 2u * m - 2u;
    if (v >= root) return;
    uint32_t depth = 0;
    for (uint32_t x = v; x != root; x = J.parent[x]) ++depth;
    if (depth > max_bits) atomicAdd(J.overflow, 1u);
    if (v < m) atomicAdd(&J.bl_count[depth > max_bits ? max_bits : depth], 1u);
}
__device__ void code_repair(const CodeJob &J, uint32_t max_bits) {   // (c): one lane
    if (J.m < 2u) return;
    int overflow = (int)*J.overflow;
    while (overflow > 0) {
        uint32_t bits = max_bits - 1u;
        while (J.bl_count[bits] == 0u) --bits;
        --J.bl_count[bits];
        J.bl_count[bits + 1u] += 2u;
        --J.bl_count[max_bits];
        overflow -= 2;
    }
}
__device__ void code_assign(const CodeJob &J, uint32_t r, uint32_t max_bits) {   // (d): the leaf of rank r (rarest first)
    if (J.m < 2u || r >= J.m) return;
    uint32_t before = 0, mine = 0;
    for (uint32_t bits = max_bits; bits >= 1u; --bits) {
        const uint32_t c = J.bl_count[bits];
        if (mine == 0u && r < before + c) mine = bits;
        before += c;
    }
    J.len[J.sorted[r]] = (uint8_t)mine;
}

// The 19-symbol code-length code and the fixed part of the block header, by the 64 lanes of ONE wavefront: phase_cl_code's
// algorithm (bgzf_deflate.hpp: huffman_lengths + canonical_codes, the one-lane form the CPU tests run) with the tree in
// registers -- node k's weight, parent and depth live in lane k of a VGPR and are read and written with v_readlane /
// v_writelane under wave-uniform control flow.  The one-lane form is a chain of ~600 dependent LDS round trips (insertion
// sort, two queues, depths, codes, header bits): 52 us of a block's 305 while 1023 lanes waited (tools/bgzf_phases.sh).
__device__ inline uint32_t rdl(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); }
__device__ inline uint32_t wrl(uint32_t val, uint32_t l, uint32_t old) { return __lane_id() == l ? val : old; }   // (v_writelane_b32 by a select: l is wave-uniform)
__device__ void phase_cl_code_wave(Shared &S, int lane) {   // lane = 0 .. 63 of the first wavefront
    constexpr uint32_t kMax = 7;
    const uint32_t cnt = lane < kNumCl ? S.freq_cl[lane] : 0u;
    const uint64_t used = __ballot(cnt != 0u);
    const uint32_t m = (uint32_t)__popcll(used);
    // rank among the used symbols by (count, symbol); sym_of = the symbol of this lane's RANK
    uint32_t rank = 0;
    for (uint32_t j = 0; j < (uint32_t)kNumCl; ++j) {
        const uint32_t cj = rdl(cnt, j);
        if (cj != 0u && (cj < cnt || (cj == cnt && j < (uint32_t)lane))) ++rank;
    }
    uint32_t bl = 0;         // lane b: number of codes of b bits
    uint32_t len = 0;        // lane s < 19: code length of symbol s
    if (m == 0u) {
        len = lane < 2 ? 1u : 0u;
        bl = lane == 1 ? 2u : 0u;
    } else if (m == 1u) {
        const uint32_t s0 = (uint32_t)__ffsll((unsigned long long)used) - 1u;
        len = ((uint32_t)lane == s0 || (uint32_t)lane == (s0 == 0u ? 1u : 0u)) ? 1u : 0u;
        bl = lane == 1 ? 2u : 0u;
    } else {
        // leaves 0 .. m-1 in ascending order: lane r gets the weight of the leaf of rank r (a push through LDS: ds_permute)
        const uint32_t to = cnt != 0u ? rank : 63u;   // (unused symbols all push to lane 63: never a node, 2 m - 1 <= 37)
        uint32_t wv = (uint32_t)__builtin_amdgcn_ds_permute((int)(to << 2), (int)cnt);
        uint32_t sym_of = (uint32_t)__builtin_amdgcn_ds_permute((int)(to << 2), lane);
        uint32_t pv = 0, dv = 0;
        uint32_t li = 0, ii = m, made = m;
        for (uint32_t k = 0; k + 1u < m; ++k) {
            uint32_t pick[2];
            for (int t = 0; t < 2; ++t) {
                const uint32_t wl = li < m ? rdl(wv, li) : 0u, wi = ii < made ? rdl(wv, ii) : 0u;
                if (li < m && (ii >= made || wl <= wi)) pick[t] = li++;
                else pick[t] = ii++;
            }
            wv = wrl(rdl(wv, pick[0]) + rdl(wv, pick[1]), made, wv);
            pv = wrl(made, pick[0], pv);
            pv = wrl(made, pick[1], pv);
            ++made;
        }
        const uint32_t root = made - 1u;
        uint32_t overflow = 0;
        for (uint32_t v = root; v-- > 0u;) {   // depths top-down with zlib's rule for the length limit (huffman_lengths)
            uint32_t bits = rdl(dv, rdl(pv, v)) + 1u;
            if (bits > kMax) { bits = kMax; ++overflow; }
            dv = wrl(bits, v, dv);
            if (v < m) bl = wrl(rdl(bl, bits) + 1u, bits, bl);
        }
        while ((int)overflow > 0) {
            uint32_t bits = kMax - 1u;
            while (rdl(bl, bits) == 0u) --bits;
            bl = wrl(rdl(bl, bits) - 1u, bits, bl);
            bl = wrl(rdl(bl, bits + 1u) + 2u, bits + 1u, bl);
            bl = wrl(rdl(bl, kMax) - 1u, kMax, bl);
            overflow -= 2u;
        }
        // the leaf of rank r (rarest first) gets the longest length still to be given out
        uint32_t mine = 0, before = 0;
        for (uint32_t bits = kMax; bits >= 1u; --bits) {
            const uint32_t c = rdl(bl, bits);
            if (mine == 0u && (uint32_t)lane < before + c) mine = bits;
            before += c;
        }
        if ((uint32_t)lane >= m) mine = 0;
        // back to symbol order: lane r pushes its length to lane sym_of
        len = (uint32_t)__builtin_amdgcn_ds_permute((int)(((uint32_t)lane < m ? sym_of : 63u) << 2), (int)mine);
        if (lane >= kNumCl) len = 0;
    }
    if (lane < kNumCl) S.len_cl[lane] = (uint8_t)len;
    if (lane <= 16) S.bl_count[lane] = lane == 0 ? 0u : bl;
    // canonical code of symbol `lane`: first code of its length + number of lower symbols of the same length
    uint32_t first_code = 0;
    {
        uint32_t c = 0;
        for (uint32_t b = 1; b <= kMax; ++b) {
            c = (c + (b > 1u ? rdl(bl, b - 1u) : 0u)) << 1;
            if (b == len) first_code = c;
        }
    }
    uint32_t lower = 0;
    for (uint32_t j = 0; j < (uint32_t)kNumCl; ++j) {
        const uint32_t lj = rdl(len, j);
        if (j < (uint32_t)lane && lj == len) ++lower;
    }
    if (lane < kNumCl) S.code_cl[lane] = len ? (uint16_t)reverse_bits(first_code + lower, (int)len) : (uint16_t)0;
    // header: BFINAL, BTYPE = 2, HLIT, HDIST, HCLEN, then the code lengths in RFC 1951's order, three bits each
    uint32_t my_sym = 0;
    switch (lane) {
        case 0: my_sym = 16; break; case 1: my_sym = 17; break; case 2: my_sym = 18; break; case 3: my_sym = 0; break;
        case 4: my_sym = 8; break; case 5: my_sym = 7; break; case 6: my_sym = 9; break; case 7: my_sym = 6; break;
        case 8: my_sym = 10; break; case 9: my_sym = 5; break; case 10: my_sym = 11; break; case 11: my_sym = 4; break;
        case 12: my_sym = 12; break; case 13: my_sym = 3; break; case 14: my_sym = 13; break; case 15: my_sym = 2; break;
        case 16: my_sym = 14; break; case 17: my_sym = 1; break; case 18: my_sym = 15; break; default: my_sym = 0; break;
    }
    const uint32_t my_len = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(my_sym << 2), (int)len);   // pull: len of symbol my_sym
    const uint64_t nz = __ballot(lane < kNumCl && my_len != 0u);
    uint32_t hclen = nz ? 64u - (uint32_t)__builtin_clzll((unsigned long long)nz) : 0u;
    if (hclen < 4u) hclen = 4u;
    if (lane == 0) {
        BitWriter w;
        w.start(out_image(S), 0);
        w.put(1, 1);
        w.put(2, 2);
        w.put(S.hlit - 257u, 5);
        w.put(S.hdist - 1u, 5);
        w.put(hclen - 4u, 4);
        w.finish();
        S.fixed_header_bits = 17u + 3u * hclen;
    }
    if ((uint32_t)lane < hclen) {
        BitWriter w;
        w.start(out_image(S), 17u + 3u * (uint32_t)lane);
        w.put(my_len, 3);
        w.finish();
    }
}

__global__ __launch_bounds__(kLanes) __attribute__((amdgpu_waves_per_eu(kLanes / 256, kLanes / 256)))   // one workgroup per CU (LDS): registers are free
void deflate_kernel(const fqtk_bgzf_block *blocks, uint32_t n_blocks, const uint32_t *n_blocks_dev,
                                                         uint32_t *out_len, uint32_t *crc_out, uint32_t *tok_all, uint32_t level, uint32_t *next_block) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem_raw[];
    Shared &S = *reinterpret_cast<Shared *>(smem_raw);
    const int lane = (int)threadIdx.x;
    uint32_t *tok = tok_all + (size_t)blockIdx.x * kTokensPerBlock;   // token scratch of this workgroup
#ifdef FQTK_BGZF_PHASE_TIMES
    if (lane == 0) s_t_mark = wall_clock64();
#endif
    if (n_blocks_dev) n_blocks = *n_blocks_dev;   // (the record pipeline learns the count on the device)
    if (blockIdx.x >= n_blocks) return;
    if (lane == 0) S.effort = effort_of_level(level);
    if (crc_out) crc_tables(S, lane);
    __syncthreads();
    // Which block a workgroup takes next: with a counter (next_block, zeroed by the caller) the workgroups that HAVE a CU share out
    // all the blocks -- a workgroup needs a CU's whole LDS, and when other kernels' wavefronts sit on some CUs (the input decoders
    // of fqtk_inflate.hip live for milliseconds to a tenth of a second) the workgroups queued for those CUs would otherwise keep
    // their 1 / gridDim.x of the blocks waiting; without one, round robin.
    __shared__ uint32_t next_j;
    for (uint32_t j = blockIdx.x;; ) {
        if (next_block) {
            if (lane == 0) next_j = atomicAdd(next_block, 1u);
            __syncthreads();
            j = next_j;
        }
        if (j >= n_blocks) break;
        const uint8_t *in = blocks[j].in;
        uint8_t *out = blocks[j].out;
        const uint32_t n = blocks[j].n_in;
        // (Fetching the next block into registers while this one is coded -- before the emit phase, or before the code
        //  construction, with barriers there that wait for LDS only -- moved the ~14 us of a block's read from this phase
        //  into that one three times out of three: the kernel ran no faster.)
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
            if (!next_block) j += gridDim.x;
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
        FQTK_PHASE_MARK(12);
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
        {   // the two big codes' lengths (the one-lane form: phase_code_lengths)
            constexpr int kDistLane0 = 640;   // distance-code nodes and leaves: lanes 640 .. 703
            static_assert(kLanes >= 704 && 2 * kNumLitLen - 2 <= kDistLane0 && 2 * kNumDist - 2 <= 64, "a lane per node of both trees");
            if (lane == 0) { S.freq_ll[256] = 1; code_tree(code_job(S, 0)); }
            else if (lane == 64) code_tree(code_job(S, 1));
            __syncthreads();
            if (lane < kDistLane0) code_depth(code_job(S, 0), (uint32_t)lane, 15u);
            else if (lane < kDistLane0 + 64) code_depth(code_job(S, 1), (uint32_t)(lane - kDistLane0), 15u);
            __syncthreads();
            if (lane == 0) code_repair(code_job(S, 0), 15u);
            else if (lane == 64) code_repair(code_job(S, 1), 15u);
            __syncthreads();
            if (lane < kDistLane0) code_assign(code_job(S, 0), (uint32_t)lane, 15u);
            else if (lane < kDistLane0 + 64) code_assign(code_job(S, 1), (uint32_t)(lane - kDistLane0), 15u);
        }
        __syncthreads();
        FQTK_PHASE_MARK(5);
        phase_codes(S, lane);
        __syncthreads();
        FQTK_PHASE_MARK(11);
        phase_cl_runs(S, lane);
        __syncthreads();
        FQTK_PHASE_MARK(14);
        phase_cl_emit(S, lane);
        __syncthreads();
        FQTK_PHASE_MARK(15);
        if (lane < 64) phase_cl_code_wave(S, lane);   // the first wavefront builds the 19-symbol code ...
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
        if (!next_block) j += gridDim.x;
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
                          uint32_t *out_len, uint32_t *crc, uint32_t *tok, int level, uint32_t *next_block) {
    if (next_block) {
        const hipError_t e = hipMemsetAsync(next_block, 0, sizeof(uint32_t), stream);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(deflate_kernel, dim3(groups), dim3(kLanes), sizeof(Shared), stream, blocks, 0u, n_blocks_dev, out_len, crc, tok,
                       level <= 0 ? 0u : (uint32_t)level, next_block);
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
int fqtk_bgzf_dev_phase_ticks(unsigned long long *out16) {
    return hipMemcpyFromSymbol(out16, HIP_SYMBOL(fqtk::bgzf::g_phase_ticks), 16 * sizeof(unsigned long long)) == hipSuccess ? 0 : 1;
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
                       z->streams[slot], blocks, n, (const uint32_t *)nullptr, out_len, (uint32_t *)nullptr, z->d_tok[slot], 5u, (uint32_t *)nullptr);
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
