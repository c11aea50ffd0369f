// memo_kernels.hip.h -- the "complete memo" path of the matcher (gfx950).
//
// The reference's hot loop is dominated by a memo cache: BarcodeMatcher::assign looks the raw read
// bytes up in an AHashMap and only runs the per-sample scan on a miss
// (/root/reference/src/lib/barcode_matching.rs:174-181).  The cache is result-neutral.  This file is
// the MI355X form of that idea, made COMPLETE and precomputed instead of lazily filled:
//
//   * canonical reads  = every base is one of A C G T N (any case; '.' = N; U = T).  At create time
//     the host enumerates EVERY canonical string within max_mismatches of some sample, runs the
//     exhaustive-scan kernel (match_kernels.hip.h) on them, and stores the ones whose result is
//     Some(idx,best,next) in an open-addressing hash table in HBM (L2/MALL resident, <= a few MB);
//   * per read the kernel packs the barcode to a 3-bit-per-base key (LDS byte LUT), hashes, probes:
//       hit  -> the stored (idx,best,next) -- computed by the scan kernel, so bit-identical to it;
//       miss -> the read is canonical and NOT within max_mismatches of any sample, therefore
//               best > max_mismatches and the reference returns None (barcode_matching.rs:150-153);
//   * non-canonical reads (IUPAC codes / unknown bytes in the READ -- rare) cannot use the table:
//     the wave handles them cooperatively in place -- the read's bit-planes are broadcast with
//     v_readlane, the 64 lanes stripe the samples, and a wavefront min / second-min butterfly
//     (DPP / ds_bpermute via __shfl_xor) folds the packed keys.  Same arithmetic as the scan kernel.
//
// Per read this is ~L byte-LUT lookups + one or two 8/16-byte probes instead of S x 8 VALU ops, which
// moves the kernel from VALU-bound (~3 % of HBM peak at S=384) towards the HBM roofline.
#pragma once
#include "match_kernels.hip.h"

namespace fqtk {

constexpr uint32_t kMemoMaxLen = 20;       // 3 bits/base, 10 bases per 32-bit half
constexpr uint32_t kMemoEmpty = 0xFFFFFFFFu;

struct MemoParams {
    MatchParams m;
    const void *slots;        // KEY64: uint4 {lo, hi, val, 0}; else uint2 {lo, val}
    const uint32_t *code_lut; // [64] dwords = 256 bytes: A0 C1 G2 T3 N4, anything else 8
    uint32_t mask;            // n_slots - 1
};

__host__ __device__ inline uint32_t memo_hash(uint32_t lo, uint32_t hi) {
    uint32_t h = lo * 0x9E3779B1u;
    h ^= hi * 0x85EBCA77u;
    h ^= h >> 15;
    h *= 0x2C1B3C6Du;
    h ^= h >> 13;
    return h;
}

// (best, second) packed keys -> result word (barcode_matching.rs:150-159).
__device__ __forceinline__ uint32_t decide(uint32_t best, uint32_t second, uint32_t max_mm, uint32_t delta) {
    const uint32_t bm = best >> 16, nm = second >> 16;
    const bool none = bm > max_mm || (nm - bm) < delta;
    return none ? 0xFFFFFFFFu : ((best & 0xFFFFu) | (bm << 16) | (nm << 24));
}

// Wave-cooperative exhaustive scan of ONE read (lane `src`'s planes): lanes stripe the samples, then
// an all-reduce butterfly of (best, second).  Returns the pair in every lane.
template <int NW>
__device__ __forceinline__ void wave_scan(const Planes<NW> &mine, int src, const MatchParams &P,
                                          uint32_t &best, uint32_t &second) {
    uint32_t pl[NW][4];
#pragma unroll
    for (int w = 0; w < NW; ++w)
#pragma unroll
        for (int j = 0; j < 4; ++j) pl[w][j] = __builtin_amdgcn_readlane(mine.p[w][j], src);
    const uint32_t lane = __lane_id();
    best = second = kKeyInit;
    const u32x4 *tab = reinterpret_cast<const u32x4 *>(P.table);
    for (uint32_t s = lane; s < P.S; s += 64) {
        uint32_t mm = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const u32x4 e = tab[(size_t)s * NW + w];
            const uint32_t m = (pl[w][0] & e.x) | (pl[w][1] & e.y) | (pl[w][2] & e.z) | (pl[w][3] & e.w);
            mm += __builtin_popcount(m);
        }
        const uint32_t key = (mm << 16) | s;
        second = med3_u32(best, second, key);
        best = min(best, key);
    }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {   // min / second-min butterfly across the wavefront
        const uint32_t ob = __shfl_xor(best, off);
        const uint32_t os = __shfl_xor(second, off);
        second = min(min(second, os), max(best, ob));
        best = min(best, ob);
    }
}

template <int VEC, bool KEY64, int R>
__global__ __launch_bounds__(kBlock) void memo_kernel(const MemoParams Q) {
    const MatchParams &P = Q.m;
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint32_t *lds_lut = smem;                                        // 256 x u32 spread LUT (fallback)
    const uint8_t *lds_code = reinterpret_cast<const uint8_t *>(smem + 256);   // 256 x u8 code LUT
    uint32_t *lds_hist = smem + 256 + 64;

    const uint32_t tid = threadIdx.x;
    lds_lut[tid] = P.lut[tid];
    if (tid < 64) smem[256 + tid] = Q.code_lut[tid];
    const uint32_t bins = P.S + 1;
    if (P.counts && P.lds_hist)
        for (uint32_t b = tid; b < bins; b += kBlock) lds_hist[b] = 0;
    __syncthreads();

    const uint32_t L = P.L;
    const uint32_t nwords = (L + 3u) >> 2;
    const uint64_t tile = (uint64_t)kBlock * R;
    const uint64_t ntiles = (P.n + tile - 1) / tile;

    for (uint64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        uint32_t words[R][8];
        uint32_t lo[R], hi[R], bad[R], res[R];
        bool live[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint64_t i = t * tile + (uint64_t)r * kBlock + tid;
            live[r] = i < P.n;
#pragma unroll
            for (int w = 0; w < 8; ++w) words[r][w] = 0x41414141u;   // dead lanes look like "AAAA"
            if (live[r]) load_words<1, VEC>(P, i, nwords, words[r]);
        }
        // ---- ASCII -> 3-bit codes, 10 bases per 32-bit half; bit 3 of any code = non-canonical ----
#pragma unroll
        for (int r = 0; r < R; ++r) {
            uint32_t l = 0, h = 0, b = 0;
#pragma unroll
            for (int k = 0; k < (int)kMemoMaxLen; ++k) {
                if ((uint32_t)k < L) {   // wave-uniform
                    const uint32_t byte = (words[r][k >> 2] >> (8 * (k & 3))) & 0xFFu;
                    const uint32_t c = lds_code[byte];
                    b |= c;
                    if (k < 10) l |= c << (3 * k); else h |= c << (3 * (k - 10));
                }
            }
            lo[r] = l; hi[r] = h; bad[r] = (b & 8u) && live[r];
        }
        // ---- probe (linear probing; EMPTY slots carry val = 0xFFFFFFFF = the None encoding) -------
#pragma unroll
        for (int r = 0; r < R; ++r) {
            uint32_t slot = memo_hash(lo[r], KEY64 ? hi[r] : 0u) & Q.mask;
            uint32_t v = kMemoEmpty;
            if (!bad[r]) {
                for (;;) {
                    if constexpr (KEY64) {
                        const uint4 e = reinterpret_cast<const uint4 *>(Q.slots)[slot];
                        if ((e.x == lo[r] && e.y == hi[r]) || e.z == kMemoEmpty) { v = e.z; break; }
                    } else {
                        const uint2 e = reinterpret_cast<const uint2 *>(Q.slots)[slot];
                        if (e.x == lo[r] || e.y == kMemoEmpty) { v = e.y; break; }
                    }
                    slot = (slot + 1) & Q.mask;
                }
            }
            res[r] = v;
        }
        // ---- rare: non-canonical reads -> wave-cooperative exhaustive scan ---------------------
#pragma unroll
        for (int r = 0; r < R; ++r) {
            uint64_t todo = __ballot(bad[r]);
            if (todo) {   // wave-uniform
                Planes<1> mine;
                encode_planes<1>(words[r], nwords, L, lds_lut, mine);
                while (todo) {
                    const int src = __ffsll((unsigned long long)todo) - 1;
                    todo &= todo - 1;
                    uint32_t b, s;
                    wave_scan<1>(mine, src, P, b, s);
                    if ((int)__lane_id() == src) res[r] = decide(b, s, P.max_mm, P.delta);
                }
            }
        }
        // ---- results + per-sample counts -------------------------------------------------------
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (!live[r]) continue;
            const uint64_t i = t * tile + (uint64_t)r * kBlock + tid;
            P.out[i] = res[r];
            if (P.counts) {
                const uint32_t idx = res[r] & 0xFFFFu;
                const uint32_t bin = idx == kNoMatch ? P.S : idx;
                if (P.lds_hist) atomicAdd(&lds_hist[bin], 1u);
                else atomicAdd(&P.counts[bin], 1ull);
            }
        }
    }

    if (P.counts && P.lds_hist) {
        __syncthreads();
        for (uint32_t b = tid; b < bins; b += kBlock) {
            const uint32_t c = lds_hist[b];
            if (c) atomicAdd(&P.counts[b], (unsigned long long)c);
        }
    }
}

}  // namespace fqtk
