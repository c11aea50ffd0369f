// match_kernels.hip.h -- gfx950 (CDNA4) device code of the sample-barcode matcher.
//
// What it computes (bit-exact with the reference):
//   reference src/lib/mod.rs:49-61 (encode), src/lib/bitenc.rs:432-459 (hamming = AND-NOT + count
//   of non-zero nibbles), src/lib/barcode_matching.rs:119-186 (best / next-best / decision).
//
// How (MI355X-first, nothing here is a translation of the Rust):
//   * one LANE per read; the expected-barcode table is WAVE-UNIFORM, so it is fetched with scalar
//     loads (s_load_dwordxN through the scalar cache) and used as SGPR operands of the VALU ops --
//     no LDS traffic, no cross-lane reduction, no divergence in the inner loop;
//   * observed barcodes are transposed once per read into 4 BIT-PLANES (A,C,G,T; bit i = base i),
//     the table holds the PRE-INVERTED planes, so a mismatch word for <=32 bases is
//         m = (oA & ~eA) | (oC & ~eC) | (oG & ~eG) | (oT & ~eT)    (1 v_and_b32 + 3 v_and_or_b32)
//     and the mismatch count is one v_bcnt_u32_b32 -- 5 VALU per (read, sample, 32 bases) instead of
//     the reference's 8-iteration nibble scan per u32 block;
//   * best/next-best are tracked as packed keys (mm << 16 | sample) so "lowest index wins" and
//     "next == best on ties" fall out of integer min / med3 (v_min_u32 + v_med3_u32);
//   * ASCII -> plane bits goes through a 256-entry LDS LUT (1 KiB) whose entries carry the 4 plane
//     bits in 4 separate bytes, so 8 bases accumulate with one v_lshl_or_b32 each;
//   * per-sample counts: LDS histogram per workgroup, flushed with one global atomic per non-empty
//     bin per workgroup.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fqtk {

constexpr int kBlock = 256;          // 4 waves; launch_bounds lets the allocator use <=64 VGPRs
constexpr uint32_t kNoMatch = 0xFFFFu;
constexpr uint32_t kKeyInit = 0x00FFFFFFu;  // mm = 255, idx = 0xFFFF: larger than any real key
constexpr uint32_t kMaxLdsHist = 8192;      // bins; above this counts go straight to global atomics

struct MatchParams {
    const uint8_t *obs;          // n x stride ASCII
    const uint32_t *lens;        // nullable
    uint32_t *out;               // n x fqtk_match_t
    unsigned long long *counts;  // nullable, S+1
    const uint32_t *table;       // [S][NW][4] pre-inverted planes (A,C,G,T), bits >= L cleared
    const uint32_t *lut;         // [256] spread LUT (bit0 A, bit8 C, bit16 G, bit24 T)
    unsigned long long *err;     // [0]: min offending read index (init ~0ull)
    uint64_t n;
    uint32_t stride;
    uint32_t S;
    uint32_t L;
    uint32_t max_mm;
    uint32_t delta;
    uint32_t nocall_limit;       // max_mismatches + max_ns_in_barcodes (barcode_matching.rs:171)
    uint32_t lds_hist;           // 1: histogram in LDS, 0: global atomics
    uint32_t scan_tab_lds;       // memo kernels: 1 = the wave scan of non-canonical reads finds the table in LDS
    uint32_t plain_samples;      // 1 = every base of every sample is one of A C G T: a read's ambiguity codes read as N (memo kernels)
    // Non-canonical reads (IUPAC / junk bytes in the READ) are not in the memo.  Every wave of a memo kernel owns
    // one segment of `work` and appends the indices of its such reads there -- no atomics: one counter on one
    // address serialises in L2 at ~8 ns per wave that has anything to add, 25 x the kernel's own time at 1 % of
    // reads -- and what does not fit is scanned in place by the wave.  A second pass, this scan kernel over the
    // listed reads only (one lane per read), resolves them and zeroes the fill counts again.
    uint32_t *work;              // [work_segs][work_cap] entries of 1 + work_rw dwords: the read's index, then its row
    uint32_t *work_n;            // [work_segs] entries filled; all zero between launches
    uint32_t work_cap;           // entries per segment
    uint32_t work_segs;          // 0: no list (every such read is scanned in place)
    // The wave that lists a read has its row in registers: the row travels with the index (work_rw dwords of it, the
    // whole row when rows are at most eight dwords; 0 = indices only), so that the second pass STREAMS its work list --
    // gathering one 16-byte row per 128-byte line made it run at ~1.7 TB/s of lines for 80 MB of useful bytes.
    uint32_t work_rw;
    // Launches carry the list (and are followed by the second pass) only once such a read has been seen: a
    // list-less memo kernel that meets one sets this word in page-locked host memory, and the host attaches
    // the list from its next launch on -- inputs without such bytes never pay for the second launch.
    uint32_t *seen;
};

// Developer-only phase clock (tools/phase_times.sh builds with -DFQTK_DEV_TIMING): where a wave's cycles go inside one
// iteration of a streaming loop.  mark(k) WAITS for everything the wave has issued (vector memory, LDS, scalar memory),
// reads the shader clock and books the cycles since the previous mark under phase k -- so a phase's number is its
// exposed latency plus its share of the SIMD, and the marks change the schedule (nothing overlaps across a mark): the
// sums say where time can be won, not what the product loop takes.  Wave-uniform: lives in SGPRs.  No-op in the product.
#ifdef FQTK_DEV_TIMING
extern __device__ unsigned long long g_dev_phase[16];
struct PhaseClock {
    unsigned long long t, acc[8];
    __device__ __forceinline__ void start() {
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = 0;
        t = __builtin_readcyclecounter();
    }
    __device__ __forceinline__ void mark(int k) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        const unsigned long long n = __builtin_readcyclecounter();
        acc[k] += n - t;
        t = n;
    }
    __device__ __forceinline__ void publish() {
        if (__lane_id() == 0) {
#pragma unroll
            for (int k = 0; k < 8; ++k) atomicAdd(&g_dev_phase[k], acc[k]);
            atomicAdd(&g_dev_phase[15], 1ull);
        }
    }
};
#else
struct PhaseClock {
    __device__ __forceinline__ void start() {}
    __device__ __forceinline__ void mark(int) {}
    __device__ __forceinline__ void publish() {}
};
#endif

__device__ __forceinline__ uint32_t med3_u32(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t d;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

// Observed barcode of one read as bit-planes: pl[w][j], w = 32-base word, j = A,C,G,T.
template <int NW>
struct Planes {
    uint32_t p[NW][4];
};

// Accumulate the 4 bases held in one dword of ASCII into `acc` (8 bases per accumulator).
// lut entries: plane bit j of the base in byte j, so (entry << k) lands base k of this group.
template <int BASE_IN_GROUP>
__device__ __forceinline__ void add4(uint32_t word, const uint32_t *lds_lut, uint32_t &acc) {
    uint32_t b0 = word & 0xFFu, b1 = (word >> 8) & 0xFFu, b2 = (word >> 16) & 0xFFu, b3 = word >> 24;
    acc = (lds_lut[b0] << (BASE_IN_GROUP + 0)) | acc;
    acc = (lds_lut[b1] << (BASE_IN_GROUP + 1)) | acc;
    acc = (lds_lut[b2] << (BASE_IN_GROUP + 2)) | acc;
    acc = (lds_lut[b3] << (BASE_IN_GROUP + 3)) | acc;
}

// Gather byte `J` of four 8-base accumulators into one 32-base plane word.
template <int J>
__device__ __forceinline__ uint32_t gather_plane(uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3) {
    // v_perm_b32: selector bytes 0-3 pick from the 2nd operand, 4-7 from the 1st, 0x0c = 0x00
    constexpr uint32_t lo_sel = 0x0c0c0000u | ((4u + J) << 8) | (uint32_t)J;
    constexpr uint32_t hi_sel = 0x00000c0cu | ((4u + J) << 24) | ((uint32_t)J << 16);
    return __builtin_amdgcn_perm(a1, a0, lo_sel) | __builtin_amdgcn_perm(a3, a2, hi_sel);
}

// words[] = the read's ASCII as dwords (little-endian), nwords = ceil(L/4) (wave-uniform).
template <int NW>
__device__ __forceinline__ void encode_planes(const uint32_t *words, uint32_t nwords, uint32_t L,
                                              const uint32_t *lds_lut, Planes<NW> &o) {
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        uint32_t a[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int dw = w * 8 + g * 2;
            if ((uint32_t)dw < nwords) add4<0>(words[dw], lds_lut, a[g]);
            if ((uint32_t)dw + 1 < nwords) add4<4>(words[dw + 1], lds_lut, a[g]);
        }
        // bases >= L inside the last dword are padding: clear them
        const int rem = (int)L - w * 32;
        const uint32_t keep = rem >= 32 ? 0xFFFFFFFFu : (rem <= 0 ? 0u : ((1u << rem) - 1u));
        o.p[w][0] = gather_plane<0>(a[0], a[1], a[2], a[3]) & keep;
        o.p[w][1] = gather_plane<1>(a[0], a[1], a[2], a[3]) & keep;
        o.p[w][2] = gather_plane<2>(a[0], a[1], a[2], a[3]) & keep;
        o.p[w][3] = gather_plane<3>(a[0], a[1], a[2], a[3]) & keep;
    }
}

// The barcode stream is read once and the result stream written once: mark both non-temporal so that
// they do not evict the (re-used) memo table from the XCD's L2.
typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2v __attribute__((ext_vector_type(2)));
#ifndef FQTK_NO_NT
#define FQTK_STREAM_LOAD(p) __builtin_nontemporal_load(p)
#define FQTK_STREAM_STORE(v, p) __builtin_nontemporal_store(v, p)
#else
#define FQTK_STREAM_LOAD(p) (*(p))
#define FQTK_STREAM_STORE(v, p) (*(p) = (v))
#endif

// Load the first ceil(L/4) dwords of read `i`.  VEC = stride in dwords when the fast, aligned vector
// path applies (1 .. 8), 0 = generic byte path (any stride / alignment).
template <int NW, int VEC, int N>
__device__ __forceinline__ void load_words(const MatchParams &P, uint64_t i, uint32_t nwords,
                                           uint32_t (&words)[N]) {
    static_assert(N >= NW * 8, "row buffer too small");
    const uint8_t *src = P.obs + i * (uint64_t)P.stride;
    if constexpr (VEC == 4) {
        const u32x4v v = FQTK_STREAM_LOAD(reinterpret_cast<const u32x4v *>(src));
        words[0] = v.x; words[1] = v.y; words[2] = v.z; words[3] = v.w;
    } else if constexpr (VEC == 2) {
        const u32x2v v = FQTK_STREAM_LOAD(reinterpret_cast<const u32x2v *>(src));
        words[0] = v.x; words[1] = v.y;
    } else if constexpr (VEC == 1) {
        words[0] = *reinterpret_cast<const uint32_t *>(src);
    } else if constexpr (VEC == 3) {
        const uint32_t *s32 = reinterpret_cast<const uint32_t *>(src);
        words[0] = s32[0]; words[1] = s32[1]; words[2] = s32[2];
    } else if constexpr (VEC == 5) {   // 20-byte reads (10+10 dual index): rows are only 4-byte aligned
        const uint32_t *s32 = reinterpret_cast<const uint32_t *>(src);
        words[0] = s32[0]; words[1] = s32[1]; words[2] = s32[2]; words[3] = s32[3]; words[4] = s32[4];
    } else if constexpr (VEC == 6) {   // 24-byte rows (12 + 12): 8-byte aligned
        const u32x2v *s64 = reinterpret_cast<const u32x2v *>(src);
        const u32x2v a = s64[0], b = s64[1], c = s64[2];
        words[0] = a.x; words[1] = a.y; words[2] = b.x; words[3] = b.y; words[4] = c.x; words[5] = c.y;
    } else if constexpr (VEC == 7) {
        const uint32_t *s32 = reinterpret_cast<const uint32_t *>(src);
#pragma unroll
        for (int w = 0; w < 7; ++w) words[w] = s32[w];
    } else if constexpr (VEC == 8) {   // 32-byte rows: 16-byte aligned
        const u32x4v a = reinterpret_cast<const u32x4v *>(src)[0], b = reinterpret_cast<const u32x4v *>(src)[1];
        words[0] = a.x; words[1] = a.y; words[2] = a.z; words[3] = a.w; words[4] = b.x; words[5] = b.y; words[6] = b.z; words[7] = b.w;
    } else if constexpr (VEC == -1) {   // stride % 4 == 0, base 4-aligned, any length
#pragma unroll
        for (int w = 0; w < NW * 8; ++w)
            if ((uint32_t)w < nwords) words[w] = reinterpret_cast<const uint32_t *>(src)[w];
    } else {
#pragma unroll
        for (int w = 0; w < NW * 8; ++w) {
            if ((uint32_t)w < nwords) {
                uint32_t x = 0;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const uint32_t k = (uint32_t)w * 4 + b;
                    if (k < P.L) x |= (uint32_t)src[k] << (8 * b);
                }
                words[w] = x;
            }
        }
    }
}

// Expected-barcode table viewed through the CONSTANT address space: loads with a wave-uniform
// address become s_load_dwordxN (scalar cache -> SGPRs) regardless of the stores the kernel makes.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) u32x4 *const_u32x4_ptr;

template <int NW, int G>
struct SampleGroup {
    u32x4 e[G][NW];   // [sample in group][32-base word] = {~A, ~C, ~G, ~T} planes
    __device__ __forceinline__ void load(const_u32x4_ptr tab, uint32_t s0) {
#pragma unroll
        for (int q = 0; q < G; ++q)
#pragma unroll
            for (int w = 0; w < NW; ++w) e[q][w] = tab[(size_t)(s0 + q) * NW + w];
    }
};

// One (read, sample) step: mismatch count over all words, then fold the packed key into the
// running (best, second) pair: 4*NW logic ops + NW v_bcnt + v_lshl_or + v_med3 + v_min.
// The logic chain and the key are written as single-instruction asm statements so that (a) the table
// words stay SGPR operands (one constant-bus read per VALU op) and (b) the OR of four ANDs is the
// 4-op v_and / v_and_or chain, not the 5-op tree the optimiser prefers.  They are not volatile: the
// scheduler still interleaves the R independent chains of a lane.
__device__ __forceinline__ uint32_t v_and_sv(uint32_t s, uint32_t v) {
    uint32_t d;
    asm("v_and_b32 %0, %1, %2" : "=v"(d) : "s"(s), "v"(v));
    return d;
}
__device__ __forceinline__ uint32_t v_and_or_vsv(uint32_t v, uint32_t s, uint32_t acc) {
    uint32_t d;
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(d) : "v"(v), "s"(s), "v"(acc));
    return d;
}
__device__ __forceinline__ uint32_t v_lshl_or_vs(uint32_t v, uint32_t s) {
    uint32_t d;
    asm("v_lshl_or_b32 %0, %1, 16, %2" : "=v"(d) : "v"(v), "s"(s));
    return d;
}

template <int NW>
__device__ __forceinline__ void update(const Planes<NW> &o, const u32x4 (&e)[NW], uint32_t s,
                                       uint32_t &best, uint32_t &second) {
    uint32_t mm = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        uint32_t m = v_and_sv(e[w].x, o.p[w][0]);
        m = v_and_or_vsv(o.p[w][1], e[w].y, m);
        m = v_and_or_vsv(o.p[w][2], e[w].z, m);
        m = v_and_or_vsv(o.p[w][3], e[w].w, m);
        mm += __builtin_popcount(m);
    }
    const uint32_t key = v_lshl_or_vs(mm, s);
    second = med3_u32(best, second, key);
    best = min(best, key);
}

__device__ __forceinline__ bool byte_is_nocall(uint8_t b) { return b == 'N' || b == 'n' || b == '.'; }

// A read LONGER than the expected barcodes (barcode_matching.rs:170-172, then :95-107): the reference counts
// the no-calls of the WHOLE read; above max_mismatches + max_ns_in_barcodes it returns None, otherwise it
// panics in count_mismatches -> the lowest such read index is latched in P.err (FQTK_ELEN).  Rare;
// `len` is clamped to the row so a bad length can never read past it.
__device__ __forceinline__ void overlong_read(const MatchParams &P, uint64_t i, uint32_t len) {
    const uint8_t *src = P.obs + i * (uint64_t)P.stride;
    const uint32_t lim = len < P.stride ? len : P.stride;
    uint32_t nc = 0;
    for (uint32_t k = 0; k < lim; ++k) nc += byte_is_nocall(src[k]) ? 1u : 0u;
    if (nc <= P.nocall_limit) atomicMin(P.err, (unsigned long long)i);
}

// Every read None: used when the rows are shorter than a barcode (stride < L, so every obs_len < L).
__global__ __launch_bounds__(256) void none_kernel(const MatchParams P) {
    const uint64_t step = (uint64_t)gridDim.x * 256;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < P.n; i += step) P.out[i] = 0xFFFFFFFFu;
    if (P.counts && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&P.counts[P.S], (unsigned long long)P.n);
}

// The kernel.  NW = 32-base words per plane (L <= 32*NW), R = reads per lane (amortises the scalar
// table loads), VEC = load path (see load_words).
// INDEXED: the second pass over the memo kernels' worklist (see MatchParams::work).
template <int NW, int R, int VEC, bool INDEXED = false>
__global__ __launch_bounds__(kBlock) void match_kernel(const MatchParams P) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint32_t *lds_lut = smem;          // 256 entries
    uint32_t *lds_hist = smem + 256;   // S+1 entries when P.lds_hist

    const uint32_t tid = threadIdx.x;
    lds_lut[tid] = P.lut[tid];
    const uint32_t bins = P.S + 1;
    if (P.counts && P.lds_hist)
        for (uint32_t b = tid; b < bins; b += kBlock) lds_hist[b] = 0;
    __syncthreads();

    const uint32_t nwords = (P.L + 3u) >> 2;
    const const_u32x4_ptr tab = (const_u32x4_ptr)(uintptr_t)P.table;
    const uint64_t tile = (uint64_t)kBlock * R;
    // INDEXED: a workgroup takes whole segments of the worklist, one after the other; else one pass over the batch
    for (uint32_t seg = INDEXED ? blockIdx.x : 0; seg < (INDEXED ? P.work_segs : 1u); seg += INDEXED ? gridDim.x : 1u) {
    uint64_t n_items = P.n;
    const uint32_t *list = nullptr;
    if constexpr (INDEXED) {
        const uint32_t c = P.work_n[seg];
        n_items = c < P.work_cap ? c : P.work_cap;
        list = P.work + (uint64_t)seg * P.work_cap * (1u + P.work_rw);
        if (n_items == 0) continue;   // workgroup-uniform
        __syncthreads();              // every lane has read the count ...
        if (tid == 0) P.work_n[seg] = 0;   // ... so it can go back to zero for the next launch
    }
    const uint64_t ntiles = (n_items + tile - 1) / tile;

    for (uint64_t t = INDEXED ? 0 : blockIdx.x; t < ntiles; t += INDEXED ? 1 : gridDim.x) {
        Planes<NW> o[R];
        bool live[R];
        uint64_t row[R];   // obs / out row of the r-th read of this lane
#pragma unroll
        for (int r = 0; r < R; ++r) {
            uint64_t i = t * tile + (uint64_t)r * kBlock + tid;
            live[r] = i < n_items;
            const uint32_t *entry = nullptr;
            if constexpr (INDEXED) {
                entry = list + i * (1u + P.work_rw);
                i = live[r] ? entry[0] : 0;
            }
            row[r] = i;
            uint32_t words[NW * 8];
#pragma unroll
            for (int w = 0; w < NW * 8; ++w) words[w] = 0;
            if (INDEXED && P.work_rw) {   // the row came with the index
                if (live[r]) {
#pragma unroll
                    for (int w = 0; w < 8; ++w)
                        if ((uint32_t)w < P.work_rw && (uint32_t)w < nwords) words[w] = entry[1 + w];
                }
            } else if (live[r]) load_words<NW, VEC, NW * 8>(P, i, nwords, words);
            encode_planes<NW>(words, nwords, P.L, lds_lut, o[r]);
        }

        uint32_t best[R], second[R];
#pragma unroll
        for (int r = 0; r < R; ++r) best[r] = second[r] = kKeyInit;

        // ---- hot loop: S samples x R reads, table operands are wave-uniform (SGPR) -------------
        // Samples go in groups of G (16 dwords = one s_load_dwordx16); the next group's scalar load
        // is issued before the current group's VALU work so SMEM latency hides under it.
        constexpr int G = NW == 1 ? 4 : (NW == 2 ? 2 : 1);
        const uint32_t ngroups = P.S / G;
        SampleGroup<NW, G> cur, nxt;
        if (ngroups) cur.load(tab, 0);
        for (uint32_t g = 0; g < ngroups; ++g) {
            if (g + 1 < ngroups) nxt.load(tab, (g + 1) * G);
#pragma unroll
            for (int q = 0; q < G; ++q) {
                const uint32_t s = g * G + q;
#pragma unroll
                for (int r = 0; r < R; ++r) update<NW>(o[r], cur.e[q], s, best[r], second[r]);
            }
            cur = nxt;
        }
        for (uint32_t s = ngroups * G; s < P.S; ++s) {   // S % G tail
            SampleGroup<NW, 1> one;
            one.load(tab, s);
#pragma unroll
            for (int r = 0; r < R; ++r) update<NW>(o[r], one.e[0], s, best[r], second[r]);
        }

        // ---- decision (barcode_matching.rs:150-159) + length rules (:165-172) ------------------
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (!live[r]) continue;
            const uint64_t i = row[r];
            const uint32_t bm = best[r] >> 16;
            const uint32_t nm = second[r] >> 16;
            bool none = bm > P.max_mm || (nm - bm) < P.delta;
            if (P.lens) {
                const uint32_t len = P.lens[i];
                if (len < P.L) {
                    none = true;
                } else if (len > P.L) {
                    none = true;
                    overlong_read(P, i, len);
                }
            }
            const uint32_t idx = none ? kNoMatch : (best[r] & 0xFFFFu);
            const uint32_t res = none ? 0xFFFFFFFFu : (idx | (bm << 16) | (nm << 24));
            FQTK_STREAM_STORE(res, &P.out[i]);
            if (P.counts) {
                const uint32_t bin = none ? P.S : idx;
                if (P.lds_hist) atomicAdd(&lds_hist[bin], 1u);
                else atomicAdd(&P.counts[bin], 1ull);
            }
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
