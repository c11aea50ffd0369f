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
//     Some(idx,best,next) in a two-choice (cuckoo) hash table in HBM (L2/MALL resident, <= a few MB);
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

constexpr uint32_t kHotBytes = 16384;      // LDS budget of the hot table per workgroup

struct MemoParams {
    MatchParams m;
    const void *slots;        // KEY64: uint4 {lo, hi, val, spill}; else uint2 {lo | spill << 31, val}
    const uint32_t *code_lut; // [64] dwords = 256 bytes: A0 C1 G2 T3 N4, anything else 8
    const uint32_t *hot;      // hot subset (exact matches) in the same slot format, copied to LDS
    uint32_t mask;            // n_slots - 1
    uint32_t hot_mask;        // hot slots - 1 (0 = no hot table)
};

// Two-choice (cuckoo) placement: a key lives in slot h1 or slot h2, nowhere else, so a lookup is two
// INDEPENDENT loads issued back to back -- no probe loop, no divergence, one memory round trip.
// 24-bit multiplies only: v_mul_u32_u24 / v_mad_u32_u24 issue at the full VALU rate on gfx950, while
// v_mul_lo_u32 is quarter rate.  The 60-bit key is cut into three <=24-bit limbs.
__host__ __device__ inline uint32_t mul24(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul24(a, b);
#else
    return (a & 0xFFFFFFu) * (b & 0xFFFFFFu);
#endif
}
__host__ __device__ inline void memo_hash2(uint32_t lo, uint32_t hi, uint32_t mask, uint32_t &s1,
                                           uint32_t &s2) {
    const uint32_t a = lo;                       // mul24 reads bits 0..23: bases 0-7
    const uint32_t b = (lo >> 24) | (hi << 6);   // bases 8-9 and 10-15
    const uint32_t c = hi >> 18;                 // bases 16-19
    uint32_t h = mul24(a, 0x9E3779u) + mul24(b, 0x85EBCBu) + mul24(c, 0xC2B2AFu);
    h ^= h >> 15;
    h = mul24(h, 0x2C1B3Du) + (h >> 9);
    h ^= h >> 13;
    s1 = h & mask;
    uint32_t g = mul24(h >> 7, 0xD6E8FFu) + h;
    g ^= g >> 14;
    s2 = g & mask;
}

// acc | (v << SH) as one v_lshl_or_b32 (the optimiser otherwise builds a shift + v_or3 tree)
template <int SH>
__device__ __forceinline__ uint32_t lshl_or_imm(uint32_t v, uint32_t acc) {
    uint32_t d;
    asm("v_lshl_or_b32 %0, %1, %2, %3" : "=v"(d) : "v"(v), "n"(SH), "v"(acc));
    return d;
}

// Compile-time unrolled ASCII -> 3-bit-code packing, two bases per step: byte extract (1 VALU each),
// LDS byte LUT (address = byte value), one v_or3 for the non-canonical flag, one v_lshl_or per base.
template <int K, int NB, int ABL>
__device__ __forceinline__ void encode_codes(const uint32_t (&words)[8], const uint8_t *lds_code, uint32_t &l,
                                             uint32_t &h, uint32_t &b) {
    if constexpr (K < NB) {
        uint32_t c[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const uint32_t wv = words[(K + u) >> 2];
            constexpr int by0 = K & 3;
            const int by = by0 + u;
            const uint32_t byte = by == 0 ? (wv & 0xFFu) : (by == 3 ? (wv >> 24) : __builtin_amdgcn_ubfe(wv, 8 * by, 8));
            c[u] = (ABL & 2) ? (byte & 3u) : lds_code[byte];
        }
        b = b | c[0] | c[1];
        if constexpr (K < 10) l = lshl_or_imm<3 * K>(c[0], l); else h = lshl_or_imm<3 * (K - 10)>(c[0], h);
        if constexpr (K + 1 < 10) l = lshl_or_imm<3 * (K + 1)>(c[1], l); else h = lshl_or_imm<3 * (K + 1 - 10)>(c[1], h);
        encode_codes<K + 2, NB, ABL>(words, lds_code, l, h, b);
    }
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

// ABL: developer-only ablation mask (tools/ablate.sh builds with -DFQTK_DEV_ABLATE); 0 in the product.
//   1 = skip table probes, 2 = skip LDS code lookups, 4 = skip histogram, 8 = skip result store,
//   16 = skip the LDS hot table
#ifndef FQTK_MEMO_WAVES
#define FQTK_MEMO_WAVES 8
#endif
template <int VEC, bool KEY64, int R, int ABL>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(FQTK_MEMO_WAVES, 8)))
void memo_kernel(const MemoParams Q) {
    const MatchParams &P = Q.m;
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    // the byte-code LUT is a STATIC LDS object (compile-time address 0): a base's byte value is its
    // LDS address, so the 16 lookups per read need no address arithmetic at all
    __shared__ uint32_t s_code[64];                                      // 256 x u8 code LUT
    const uint8_t *lds_code = reinterpret_cast<const uint8_t *>(s_code);
    uint32_t *lds_lut = smem;                                            // 256 x u32 spread LUT (fallback)
    // hot table: the memo entries with 0 mismatches (a read that IS a sample barcode -- the bulk of
    // real data) live in LDS, so most lanes never touch the global table; the rest probe it with
    // the hit lanes masked off, which shrinks the gather traffic by the hit rate.
    const uint32_t hot_words = Q.hot_mask ? (Q.hot_mask + 1) * (KEY64 ? 4u : 2u) : 0u;
    uint32_t *lds_hot = smem + 256;
    uint32_t *lds_hist = lds_hot + hot_words;

    const uint32_t tid = threadIdx.x;
    lds_lut[tid] = P.lut[tid];
    if (tid < 64) s_code[tid] = Q.code_lut[tid];
    for (uint32_t w = tid; w < hot_words; w += kBlock) lds_hot[w] = Q.hot[w];
    const uint32_t bins = P.S + 1;
    if (P.counts && P.lds_hist)
        for (uint32_t b = tid; b < bins; b += kBlock) lds_hist[b] = 0;
    __syncthreads();

    const uint32_t L = P.L;
    const uint32_t nwords = (L + 3u) >> 2;
    // bases encoded per read: exactly the packed stride on the vector paths, the 20-base maximum else
    constexpr int NB = VEC >= 1 ? VEC * 4 : (int)kMemoMaxLen;
    uint32_t keep[NB / 4];   // byte masks of the real bases (< L) in each word; wave-uniform
#pragma unroll
    for (int w = 0; w < NB / 4; ++w) {
        const int rem = (int)L - 4 * w;
        keep[w] = rem >= 4 ? 0xFFFFFFFFu : (rem <= 0 ? 0u : ((1u << (8 * rem)) - 1u));
    }
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
        // Straight-line on purpose: NB is a compile-time constant and pad positions (>= L) are forced
        // to 'A' (= code 0 = "absent" in the key), so all NB LUT reads are in flight together instead
        // of one LDS round trip per base behind a wave-uniform `k < L` branch.
#pragma unroll
        for (int r = 0; r < R; ++r) {
#pragma unroll
            for (int w = 0; w < NB / 4; ++w) words[r][w] = (words[r][w] & keep[w]) | (0x41414141u & ~keep[w]);
            uint32_t l = 0, h = 0, b = 0;
            encode_codes<0, NB, ABL>(words[r], lds_code, l, h, b);
            lo[r] = l; hi[r] = h; bad[r] = (b & 8u) && live[r];
        }
        // ---- probe: both candidate slots of every read are loaded up front (2*R independent
        //      gathers in flight), then compared.  Empty slots carry key = ~0 and val = None. -------
        uint32_t s1[R], s2[R];
#pragma unroll
        for (int r = 0; r < R; ++r) memo_hash2(lo[r], KEY64 ? hi[r] : 0u, Q.mask, s1[r], s2[r]);
        bool hit[R];
#pragma unroll
        for (int r = 0; r < R; ++r) { hit[r] = false; res[r] = kMemoEmpty; }
        if (Q.hot_mask && !(ABL & 16)) {   // wave-uniform
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const uint32_t a1 = s1[r] & Q.hot_mask, a2 = s2[r] & Q.hot_mask;
                if constexpr (KEY64) {
                    const uint4 h1 = reinterpret_cast<const uint4 *>(lds_hot)[a1];
                    const uint4 h2 = reinterpret_cast<const uint4 *>(lds_hot)[a2];
                    const bool m1 = h1.x == lo[r] && h1.y == hi[r];
                    const bool m2 = h2.x == lo[r] && h2.y == hi[r];
                    hit[r] = m1 || m2;
                    res[r] = m1 ? h1.z : (m2 ? h2.z : kMemoEmpty);
                } else {
                    const uint2 h1 = reinterpret_cast<const uint2 *>(lds_hot)[a1];
                    const uint2 h2 = reinterpret_cast<const uint2 *>(lds_hot)[a2];
                    const bool m1 = h1.x == lo[r], m2 = h2.x == lo[r];
                    hit[r] = m1 || m2;
                    res[r] = m1 ? h1.y : (m2 ? h2.y : kMemoEmpty);
                }
            }
        }
        if constexpr (ABL & 1) {
#pragma unroll
            for (int r = 0; r < R; ++r) res[r] = (s1[r] ^ s2[r]) | 0xFFFFu;
        } else {
            // Global table, two-choice placement with a per-slot SPILL bit: the builder keeps a key in
            // its first slot whenever it can and marks a slot whose would-be owner lives in its second
            // slot.  So one gather settles ~90 % of the probing lanes (hit, or miss with spill = 0);
            // only the rest issue the second, dependent gather -- with almost every lane masked off.
            bool again[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                again[r] = false;
                if (!hit[r] && !bad[r]) {
                    if constexpr (KEY64) {
                        const uint4 e = reinterpret_cast<const uint4 *>(Q.slots)[s1[r]];
                        if (e.x == lo[r] && e.y == hi[r]) res[r] = e.z;
                        else again[r] = (e.w & 1u) != 0;
                    } else {
                        const uint2 e = reinterpret_cast<const uint2 *>(Q.slots)[s1[r]];
                        if ((e.x & 0x7FFFFFFFu) == lo[r]) res[r] = e.y;
                        else again[r] = (e.x >> 31) != 0;
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (again[r]) {
                    if constexpr (KEY64) {
                        const uint4 e = reinterpret_cast<const uint4 *>(Q.slots)[s2[r]];
                        if (e.x == lo[r] && e.y == hi[r]) res[r] = e.z;
                    } else {
                        const uint2 e = reinterpret_cast<const uint2 *>(Q.slots)[s2[r]];
                        if ((e.x & 0x7FFFFFFFu) == lo[r]) res[r] = e.y;
                    }
                }
            }
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
            if constexpr (ABL & 8) { if (res[r] == 0x12345u) P.out[i] = res[r]; } else
            FQTK_STREAM_STORE(res[r], &P.out[i]);
            if (P.counts && !(ABL & 4)) {
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
