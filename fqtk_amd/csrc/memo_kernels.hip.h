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
//   * per read the kernel packs the barcode to a 4-bit-per-base key (SWAR on the packed ASCII words:
//     code = bits 1..2 of the byte, N = 7; validated with one v_perm_b32 per word), hashes, probes:
//       hit  -> the stored (idx,best,next) -- computed by the scan kernel, so bit-identical to it;
//       miss -> the read is canonical and NOT within max_mismatches of any sample, therefore
//               best > max_mismatches and the reference returns None (barcode_matching.rs:150-153);
//   * non-canonical reads (IUPAC codes / unknown bytes in the READ -- rare) cannot use the table:
//     the wave handles them cooperatively in place -- the read's bit-planes are broadcast with
//     v_readlane, the 64 lanes stripe the samples, and a wavefront min / second-min butterfly
//     (DPP / ds_bpermute via __shfl_xor) folds the packed keys.  Same arithmetic as the scan kernel.
//
// Per read this is ~7 VALU ops per 4 bases + one or two 8/16-byte probes instead of S x 8 VALU ops,
// which moves the kernel from VALU-bound (~3 % of HBM peak at S=384) towards the HBM roofline.
#pragma once
#include "match_kernels.hip.h"
#include "memo_hash.hpp"

namespace fqtk {

// A read's row in registers: up to eight dwords of barcode (32 bases) and, for a variable-length batch, its length.
constexpr int kRowWords = 9;
constexpr int kLenWord = 8;

struct MemoParams {
    MatchParams m;
    const void *slots;        // KW=1: uint2 {lo | spill << 31, val}.  KW=2: uint4 {lo, hi, val, spill}.  KW=3: uint4 {lo, hi, ext,
                              // val | spill << 31} (a result word never has bit 31: next <= 32).  KW=4: two uint4 {lo, hi, ext, ext2}
                              // {val, spill, -, -}, one 32-byte line
                              // direct form: BUCKETS of two one-word-key slots, uint4 {lo0 | spill << 31, val0, lo1, val1} (mask = buckets - 1)
    const uint32_t *hot;      // hot subset (exact matches) in the same slot format, copied to LDS
    uint32_t mask;            // n_slots - 1
    uint32_t hot_mask;        // hot slots - 1 (0 = no hot table)
    const uint32_t *filter;   // presence filter over ALL keys of the table (memo_hash.hpp), copied to LDS behind the hot table; NULL = none
    uint32_t filter_bits;     // log2 of its bits (0 = none).  With a filter the hot table is TWO-choice: slot h1 & hot_mask or h2 & hot_mask
    // direct-indexed form (short barcodes, memo_hash.hpp): `slots` then holds only the N-containing entries
    const void *direct;       // [memo_direct_entries(L)] uint16 (packed) or uint32 results, indexed by the read itself
    const uint32_t *hot2;     // [2 << hot2_bits] LDS cache of the exact-match entries (two-slot buckets), or NULL
    uint32_t hot2_bits;       // log2(buckets)
    uint32_t d_nbits;         // bits of the read's index (memo_direct_index_bits(L))
    uint32_t d_ib, d_bb;      // 16-bit entries: bits of idx and of best (memo_direct_unpack16)
};

// ASCII -> 4-bit codes, SWAR on the packed words (no LDS, no per-base work).  Per 4-base word:
//   c   = (w >> 1) & 0x07070707          the four codes, one per byte            (2 VALU)
//   e   = v_perm_b32(pool, c)            the bytes those codes stand for         (1)
//   bad |= (w ^ e) & 0xDFDFDFDF          any other byte (IUPAC, '.', junk) flags (1, v_bitop3)
// and ONE v_lshl_or_b32 per PAIR of words makes a key word: codes of the second word << 4 | codes of
// the first (memo_nibble_shift is the host's view of that order).  kc / kv are the two masks above cut
// down to the real bases of a word (pad positions encode as 'A' = 0 = absent); FULL says all words but
// the last are complete, so only the last one needs its masks from SGPRs.
// plain_samples (wave-uniform; every sample base is one of A C G T): the odd bytes of a read get their codes where they stand
// (memo_hash.hpp recode_flagged_bytes: ambiguity codes and '.' read as N, U as T) and `bad` is left set only by bytes of no
// IUPAC meaning -- behind one wave-uniform test per read slot and one per word in which some lane holds such a byte.
template <int NWD, bool FULL, bool FOLD>
__device__ __forceinline__ void encode_nibbles(const uint32_t (&words)[kRowWords], const uint32_t (&kc)[NWD],
                                               const uint32_t (&kv)[NWD], uint32_t (&key)[4], uint32_t &bad, uint32_t &lo_unf, uint32_t &c2,
                                               bool plain_samples = false) {
    static_assert(!FOLD || NWD == 3, "the fold is for the 9-10 base keys only");
    static_assert(NWD >= 1 && NWD <= 8, "at most 32 bases");
    uint32_t c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    bad = 0;
#pragma unroll
    for (int w = 0; w < NWD; ++w) {
        const bool full = FULL && w < NWD - 1;
        uint32_t x;
        encode_word(words[w], full ? 0x07070707u : kc[w], full ? 0xDFDFDFDFu : kv[w], c[w], x);
        bad |= x;
    }
    if (plain_samples && __builtin_amdgcn_uicmp(bad, 0u, 33 /* ne */)) {   // wave-uniform
        bad = 0;
#pragma unroll
        for (int w = 0; w < NWD; ++w) {
            const bool full = FULL && w < NWD - 1;
            uint32_t x = (words[w] ^ perm_b32(kCodePoolHi, kCodePoolLo, c[w])) & (full ? 0xDFDFDFDFu : kv[w]);   // (the byte-wise test again)
            if (__builtin_amdgcn_uicmp(x, 0u, 33)) x = recode_flagged_bytes(words[w], x, c[w], full ? 0x07070707u : kc[w]);
            bad |= x;
        }
    }
    key[0] = NWD >= 2 ? ((c[1] << 4) | c[0]) : c[0];
    lo_unf = key[0];    // bases 0-7 before the fold, and the codes of bases 8.. : what memo_direct_index reads
    c2 = c[2];
    if constexpr (FOLD) {   // L <= 10: kc[2] leaves only codes 8 and 9 in c[2]
        key[0] |= mul24(c[2], kFoldMul) & kFoldMask;
        key[1] = key[2] = key[3] = 0;
        return;
    }
    key[1] = NWD >= 4 ? ((c[3] << 4) | c[2]) : (NWD == 3 ? c[2] : 0u);
    key[2] = NWD >= 6 ? ((c[5] << 4) | c[4]) : (NWD == 5 ? c[4] : 0u);
    key[3] = NWD >= 8 ? ((c[7] << 4) | c[6]) : (NWD == 7 ? c[6] : 0u);
}

template <int NWD, bool FULL, bool FOLD>
__device__ __forceinline__ void encode_nibbles(const uint32_t (&words)[kRowWords], const uint32_t (&kc)[NWD],
                                               const uint32_t (&kv)[NWD], uint32_t (&key)[4], uint32_t &bad) {
    uint32_t lo_unf, c2;
    encode_nibbles<NWD, FULL, FOLD>(words, kc, kv, key, bad, lo_unf, c2, false);
}

// The same encode with the words' codes kept apart (the LDS forms: see recode_flagged_bytes): c[w] = the four codes of word w,
// bad = nonzero when some byte is not one of A C G T N (either case).  flagged_bytes_of_word: that test for ONE word again, byte
// by byte -- the rare wave that needs it recomputes it (two instructions a word) so that the others do not carry it.
template <int NWD, bool FULL>
__device__ __forceinline__ void encode_codes(const uint32_t (&words)[kRowWords], const uint32_t (&kc)[NWD], const uint32_t (&kv)[NWD],
                                             uint32_t (&c)[NWD], uint32_t &bad) {
    bad = 0;
#pragma unroll
    for (int w = 0; w < NWD; ++w) {
        const bool full = FULL && w < NWD - 1;
        uint32_t x;
        encode_word(words[w], full ? 0x07070707u : kc[w], full ? 0xDFDFDFDFu : kv[w], c[w], x);
        bad |= x;
    }
}
__device__ __forceinline__ uint32_t flagged_bytes_of_word(uint32_t w, uint32_t c, uint32_t byte_mask) {
    return (w ^ perm_b32(kCodePoolHi, kCodePoolLo, c)) & byte_mask;
}
template <int NWD>
__device__ __forceinline__ void assemble_key(const uint32_t (&c)[NWD], uint32_t (&key)[4]) {
    key[0] = NWD >= 2 ? ((c[NWD >= 2 ? 1 : 0] << 4) | c[0]) : c[0];
    key[1] = NWD >= 4 ? ((c[NWD >= 4 ? 3 : 0] << 4) | c[NWD >= 3 ? 2 : 0]) : (NWD == 3 ? c[NWD >= 3 ? 2 : 0] : 0u);
    key[2] = NWD >= 6 ? ((c[NWD >= 6 ? 5 : 0] << 4) | c[NWD >= 5 ? 4 : 0]) : (NWD == 5 ? c[NWD >= 5 ? 4 : 0] : 0u);
    key[3] = NWD >= 8 ? ((c[NWD >= 8 ? 7 : 0] << 4) | c[NWD >= 7 ? 6 : 0]) : (NWD == 7 ? c[NWD >= 7 ? 6 : 0] : 0u);
}

// (best, second) packed keys -> result word (barcode_matching.rs:150-159).
__device__ __forceinline__ uint32_t decide(uint32_t best, uint32_t second, uint32_t max_mm, uint32_t delta) {
    const uint32_t bm = best >> 16, nm = second >> 16;
    const bool none = bm > max_mm || (nm - bm) < delta;
    return none ? 0xFFFFFFFFu : ((best & 0xFFFFu) | (bm << 16) | (nm << 24));
}

// Wave-cooperative exhaustive scan of ONE read (lane `src`'s planes): lanes stripe the samples, then
// an all-reduce butterfly of (best, second).  Returns the pair in every lane.
// `lds_tab`: the table staged in LDS by the kernel (P.scan_tab_lds), or nullptr -> rows come from global
// memory.  A scan is latency: 6-24 rows per lane; from L2 that is ~1.8 us per scanned read (measured: 1 % of
// cfg 3 reads with an IUPAC byte halved the kernel's throughput), from LDS a few hundred cycles.
template <int NW>
__device__ __forceinline__ void wave_scan(const Planes<NW> &mine, int src, const MatchParams &P,
                                          uint32_t &best, uint32_t &second, const uint32_t *lds_tab = nullptr) {
    uint32_t pl[NW][4];
#pragma unroll
    for (int w = 0; w < NW; ++w)
#pragma unroll
        for (int j = 0; j < 4; ++j) pl[w][j] = __builtin_amdgcn_readlane(mine.p[w][j], src);
    const uint32_t lane = __lane_id();
    best = second = kKeyInit;
    const u32x4 *tab = reinterpret_cast<const u32x4 *>(P.table);
    typedef __attribute__((address_space(3))) const u32x4 lds_row;
    lds_row *ltab = reinterpret_cast<lds_row *>((uintptr_t)(__attribute__((address_space(3))) const uint32_t *)lds_tab);
    // four table rows per lane in flight per round (one memory round trip per 256 samples, not per 64)
    for (uint32_t s0 = lane; s0 < P.S; s0 += 256) {
        u32x4 e[4][NW];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t s = s0 + 64u * k;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                if (lds_tab) e[k][w] = s < P.S ? ltab[(size_t)s * NW + w] : u32x4{0u, 0u, 0u, 0u};   // wave-uniform choice
                else e[k][w] = s < P.S ? tab[(size_t)s * NW + w] : u32x4{0u, 0u, 0u, 0u};
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t s = s0 + 64u * k;
            uint32_t mm = 0;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                const uint32_t m = (pl[w][0] & e[k][w].x) | (pl[w][1] & e[k][w].y) | (pl[w][2] & e[k][w].z) |
                                   (pl[w][3] & e[k][w].w);
                mm += __builtin_popcount(m);
            }
            const uint32_t key = s < P.S ? ((mm << 16) | s) : kKeyInit;
            second = med3_u32(best, second, key);
            best = min(best, key);
        }
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
//   1 = skip table probes, 2 = skip the canonical-byte validation, 4 = skip histogram,
//   8 = skip result store, 16 = skip the LDS hot table, 32 = fold the global probes into 4 KB (L1 hits),
//   64 = skip the second (spill) probe, 128 = second probe goes to the first probe's neighbour slot,
//   256 = skip the cuckoo table (direct form: reads with an N come out None)
// Workgroup of the table-form kernel: 1024 lanes, two per CU, so that the LDS hot table can be 64 KiB
// instead of 16 KiB at the same 32 waves/CU.  Measured (tools/ab_memo_block.sh): 768 samples x 16 bases
// 114 -> 168 G reads/s, 1536 x 10 152 -> 213, cfg 5 82 -> 120; tables whose exact-match entries already
// fit 16 KiB (cfg 3) are unchanged.
#ifndef FQTK_MEMO_BLOCK
#define FQTK_MEMO_BLOCK 1024
#endif
constexpr int kMemoBlock = FQTK_MEMO_BLOCK;
#ifndef FQTK_MEMO_WAVES
#define FQTK_MEMO_WAVES 8
#endif
// The cheap per-tile test flags every byte that is not A/C/G/T/N in either case -- including '.', the
// legacy no-call, which encodes exactly like 'N' (mod.rs:85-87: 'N', 'n' and '.' are the no-calls) and has
// the same 4-bit key code.  So a read whose only offence is '.' was ALREADY looked up under the right key;
// this exact test (rare branch only) keeps such reads out of the wave-cooperative scan.
template <int NWD>
__device__ __forceinline__ uint32_t noncanonical_beyond_dots(const uint32_t (&words)[kRowWords], const uint32_t (&kc)[NWD],
                                                             const uint32_t (&kv)[NWD]) {
    uint32_t bad = 0;
#pragma unroll
    for (int w = 0; w < NWD; ++w) {
        const uint32_t t = words[w] ^ 0x2E2E2E2Eu;
        const uint32_t dot = ~(((t & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | t | 0x7F7F7F7Fu);   // 0x80 in every byte that is '.'
        const uint32_t ww = words[w] ^ ((dot >> 7) * 0x60u);                             // '.' (0x2E) -> 'N' (0x4E)
        const uint32_t c = (ww >> 1) & kc[w];
        const uint32_t e = __builtin_amdgcn_perm(kCodePoolHi, kCodePoolLo, c);
        bad |= (ww ^ e) & kv[w];
    }
    return bad;
}

// Hands the lanes in `flagged` (a wave-wide mask; `mine` = this lane is one of them) over to the second pass:
// each writes its read index into the wave's own segment of the worklist (`fill` = entries used so far,
// wave-uniform) and gets the placeholder result.  Returns the lanes that did NOT fit (segment full / no list):
// the caller scans those in place.
__device__ __forceinline__ uint64_t defer_to_second_pass(const MatchParams &P, uint32_t seg, uint32_t &fill, uint64_t flagged,
                                                         bool mine, uint64_t read_index, uint32_t &res, const uint32_t (&row)[kRowWords]) {
    if (!flagged) return flagged;
    if (seg >= P.work_segs) {   // launched without a list (no such read seen so far): tell the host, once per wave
        if (!fill) {
            fill = 1;
            if (P.seen && __lane_id() == 0) *P.seen = 1u;   // page-locked host memory
        }
        return flagged;
    }
    const uint32_t at = fill + (uint32_t)__popcll((unsigned long long)(flagged & ((1ull << __lane_id()) - 1ull)));
    const bool fits = mine && at < P.work_cap;
    if (fits) {
        uint32_t *entry = P.work + ((uint64_t)seg * P.work_cap + at) * (1u + P.work_rw);
        entry[0] = (uint32_t)read_index;
        // the row goes along (wave-uniform count; the words are in registers already): as few store instructions as
        // the width allows -- every one of them sits in the pipelined loop's vmcnt window
        if (P.work_rw == 4) {
            *reinterpret_cast<u32x4v *>(entry + 1) = u32x4v{row[0], row[1], row[2], row[3]};
        } else if (P.work_rw == 2) {
            *reinterpret_cast<u32x2v *>(entry + 1) = u32x2v{row[0], row[1]};
        } else {
#pragma unroll
            for (int w = 0; w < 8; ++w)
                if ((uint32_t)w < P.work_rw) entry[1 + w] = row[w];
        }
        res = kMemoDeferred;   // the second pass writes the result and counts it
    }
    fill = (uint32_t)__builtin_amdgcn_readfirstlane((int)min(fill + (uint32_t)__popcll((unsigned long long)flagged), P.work_cap));
    return __ballot(mine && !fits);
}
// The wave's fill count goes out once, at the end of the kernel.
__device__ __forceinline__ void publish_worklist_fill(const MatchParams &P, uint32_t seg, uint32_t fill) {
    if (fill && seg < P.work_segs && __lane_id() == 0) P.work_n[seg] = fill;
}

// LENS: the batch carries obs_len (variable-length '+B' structures): the memo serves the reads of length
// exactly L, the others follow the length rules of barcode_matching.rs:165-172.  A separate instantiation,
// so that the fixed-length kernels carry none of it.
// DIRECT (0 = off, 2 / 4 = bytes per entry): barcodes of <= 10 bases -- reads without a no-call index a flat
// array by their own 2-bit codes (memo_hash.hpp), exact matches are caught by a compact LDS cache of that
// array, and only reads with an N go to the cuckoo table.
// PF: the full-tile loop is software-pipelined one tile deep on both streams (see lds_memo_kernels.hip.h: on
// gfx950 a wait for loads is also a wait for every store issued since, so the next tile's loads and the
// previous tile's stores are issued together at the top of an iteration and fly during the look-up).
template <int VEC, int KW, int R, int ABL, bool LENS = false, int DIRECT = 0, bool PF = false>
__global__ __launch_bounds__(kMemoBlock) __attribute__((amdgpu_waves_per_eu(FQTK_MEMO_WAVES, 8)))
void memo_kernel(const MemoParams Q) {
    static_assert(DIRECT == 0 || KW == 1, "the direct index is for keys of <= 10 bases");
    const MatchParams &P = Q.m;
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint32_t *lds_lut = smem;                                            // 256 x u32 spread LUT (fallback)
    // hot table: the memo entries with 0 mismatches (a read that IS a sample barcode -- the bulk of
    // real data) live in LDS, so most lanes never touch the global table; the rest probe it with
    // the hit lanes masked off, which shrinks the gather traffic by the hit rate.
    const uint32_t hot_words = DIRECT ? (Q.hot2 ? (2u << Q.hot2_bits) : 0u)
                                      : (Q.hot_mask ? (Q.hot_mask + 1) * (KW == 4 ? 8u : (KW >= 2 ? 4u : 2u)) : 0u);
    const uint32_t filter_words = (!DIRECT && Q.filter_bits) ? 1u << (Q.filter_bits - 5u) : 0u;
    uint32_t *lds_hot = smem + 256;
    uint32_t *lds_filter = lds_hot + hot_words;
    uint32_t *lds_hist = lds_filter + filter_words;
    const uint32_t bins = P.S + 1;
    // [S][1][4] planes for the wave scan of non-canonical reads (L <= 20: one word), 16-byte aligned behind the histogram
    uint32_t *lds_tab = P.scan_tab_lds ? smem + ((256u + hot_words + filter_words + ((P.counts && P.lds_hist) ? bins : 0u) + 3u) & ~3u) : nullptr;

    const uint32_t tid = threadIdx.x;
    if (tid < 256) lds_lut[tid] = P.lut[tid];
    const uint32_t *hot_src = DIRECT ? Q.hot2 : Q.hot;
    for (uint32_t w = tid; w < hot_words; w += kMemoBlock) lds_hot[w] = hot_src[w];
    for (uint32_t w = tid; w < filter_words; w += kMemoBlock) lds_filter[w] = Q.filter[w];
    if (P.counts && P.lds_hist)
        for (uint32_t b = tid; b < bins; b += kMemoBlock) lds_hist[b] = 0;
    if (lds_tab)
        for (uint32_t w = tid; w < P.S * 4u; w += kMemoBlock) lds_tab[w] = P.table[w];
    __syncthreads();

    const uint32_t L = P.L;
    const uint32_t nwords = (L + 3u) >> 2;
    // words encoded per read: exactly the packed stride on the vector paths, the key's capacity else
    constexpr int NWD = VEC >= 1 ? VEC : (KW == 1 ? 3 : 2 * KW);
    constexpr bool FOLD = KW == 1 && NWD == 3;
    static_assert(NWD <= 2 * KW || FOLD, "key too narrow for the load width");
    uint32_t kc[NWD], kv[NWD];   // code / validation masks of the real bases (< L) per word; wave-uniform
#pragma unroll
    for (int w = 0; w < NWD; ++w) {
        const int rem = (int)L - 4 * w;
        const uint32_t keep = rem >= 4 ? 0xFFFFFFFFu : (rem <= 0 ? 0u : ((1u << (8 * rem)) - 1u));
        kc[w] = keep & 0x07070707u;
        kv[w] = (ABL & 2) ? 0u : (keep & 0xDFDFDFDFu);
    }
    const uint64_t tile = (uint64_t)kMemoBlock * R;
    const uint64_t ntiles = (P.n + tile - 1) / tile;
    const uint32_t hot2_mask = (1u << Q.hot2_bits) - 1u;
    // lane -> read inside a tile is WAVE-CONTIGUOUS (a wave's R loads cover 64 R consecutive reads), and a full
    // tile is addressed as "uniform 64-bit base + loop-invariant 32-bit lane offset" (as in the LDS form)
    // (ONE lane offset per stream: read r of a lane sits r * 64 rows behind read 0, which goes into the uniform part of
    // the address -- the compiler keeps every loop-invariant lane offset as a 64-bit register pair.)
    uint32_t local[R];
#pragma unroll
    for (int r = 0; r < R; ++r) local[r] = (tid >> 6) * (64u * R) + (uint32_t)r * 64u + (tid & 63u);
    const uint32_t in_off0 = local[0] * P.stride, out_off0 = local[0] * 4u;

    // The packed vector loads of one full tile (every read exists, the rows are VEC dwords).
    auto load_full = [&](uint64_t t, uint32_t (&words)[R][kRowWords]) {
        const uint8_t *tile_in = P.obs + t * tile * (uint64_t)P.stride;   // wave-uniform
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint8_t *src = (tile_in + (uint32_t)r * 64u * P.stride) + in_off0;
            if constexpr (VEC == 4) {
                const u32x4v v = FQTK_STREAM_LOAD(reinterpret_cast<const u32x4v *>(src));
                words[r][0] = v.x; words[r][1] = v.y; words[r][2] = v.z; words[r][3] = v.w;
            } else if constexpr (VEC == 3 || VEC == 5 || VEC == 7) {   // 12- / 20- / 28-byte rows are only 4-byte aligned: dword pieces,
                const uint32_t *s32 = reinterpret_cast<const uint32_t *>(src);   // non-temporal like the rest of the stream
#pragma unroll
                for (int w = 0; w < VEC; ++w) words[r][w] = FQTK_STREAM_LOAD(s32 + w);
            } else if constexpr (VEC == 2) {
                const u32x2v v = FQTK_STREAM_LOAD(reinterpret_cast<const u32x2v *>(src));
                words[r][0] = v.x; words[r][1] = v.y;
            } else if constexpr (VEC == 6) {   // 24-byte rows (12 + 12 dual index): three 8-byte pieces
#pragma unroll
                for (int w = 0; w < 3; ++w) {
                    const u32x2v v = FQTK_STREAM_LOAD(reinterpret_cast<const u32x2v *>(src) + w);
                    words[r][2 * w] = v.x; words[r][2 * w + 1] = v.y;
                }
            } else if constexpr (VEC == 8) {   // 32-byte rows: two 16-byte pieces
#pragma unroll
                for (int w = 0; w < 2; ++w) {
                    const u32x4v v = FQTK_STREAM_LOAD(reinterpret_cast<const u32x4v *>(src) + w);
                    words[r][4 * w] = v.x; words[r][4 * w + 1] = v.y; words[r][4 * w + 2] = v.z; words[r][4 * w + 3] = v.w;
                }
            } else {
                words[r][0] = FQTK_STREAM_LOAD(reinterpret_cast<const uint32_t *>(src));
            }
            // a variable-length batch: the read's length travels with its row, in the buffer's last word -- loaded in
            // the same group, so these batches run the same loops as the others
            if constexpr (LENS) words[r][kLenWord] = FQTK_STREAM_LOAD(P.lens + t * tile + local[r]);
        }
    };
    // Any tile through the generic path (ragged last tile, unaligned strides): bounds-checked loads.
    auto load_any = [&](uint64_t t, uint32_t (&words)[R][kRowWords], bool (&live)[R]) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint64_t i = t * tile + local[r];
            live[r] = i < P.n;
#pragma unroll
            for (int w = 0; w < 8; ++w) words[r][w] = 0x41414141u;   // dead lanes look like "AAAA"
            if (live[r]) load_words<1, VEC, kRowWords>(P, i, nwords, words[r]);
            if constexpr (LENS) words[r][kLenWord] = live[r] ? P.lens[i] : L;
        }
    };

    // this wave's segment of the second pass's worklist, and how much of it is used
    const uint32_t work_seg = blockIdx.x * (kMemoBlock / 64u) + (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));   // wave-uniform: SGPRs
    uint32_t work_fill = 0;
    PhaseClock clk;   // (developer builds only)
    clk.start();
    // Looks one tile up: res[r] = the result word of the tile's r-th read; also feeds the histogram.
    auto lookup = [&](uint64_t t, uint32_t (&words)[R][kRowWords], const bool (&live)[R], uint32_t (&res)[R]) {
        uint32_t key[R][4], didx[R];
        bool bad[R], has_n[R];
        // ---- ASCII -> 4-bit codes (SWAR, see encode_nibbles); `bad` = some base is not A C G T N ----
#pragma unroll
        for (int r = 0; r < R; ++r) {
            uint32_t b, lo_unf, c2;
            encode_nibbles<NWD, (VEC >= 1 && !(ABL & 2)), FOLD>(words[r], kc, kv, key[r], b, lo_unf, c2, P.plain_samples != 0u);
            bad[r] = b != 0 && live[r];
            if constexpr (LENS) bad[r] = bad[r] && words[r][kLenWord] == L;
            if constexpr (DIRECT) {
                didx[r] = memo_direct_index(lo_unf, c2);
                has_n[r] = memo_nocall_bits(lo_unf, c2) != 0;
            } else {
                didx[r] = 0;
                has_n[r] = true;   // every read takes the cuckoo table
            }
        }
        // Every load of a tile's probe phase is UNCONDITIONAL, and the loads of one phase are issued together and waited for
        // once (`arrived`): lanes that need nothing read the table's first bytes (one line for the whole wave).  A load
        // inside a lane-masked branch is followed by its own s_waitcnt before the next read's branch can start -- the probe
        // phase of a two-read tile paid up to six LDS / L2 round trips back to back (round 3, tools/ab_libs_cfg5.sh, same
        // box: cfg 3 with the table form pinned 172.9 -> 198.1 G reads/s).
        auto arrived2 = [&](u32x2v (&v)[R]) {
#pragma unroll
            for (int r = 0; r < R; ++r) { asm volatile("" : "+v"(v[r].x), "+v"(v[r].y) : : "memory"); }
        };
        auto arrived4 = [&](u32x4v (&v)[R]) {
#pragma unroll
            for (int r = 0; r < R; ++r) { asm volatile("" : "+v"(v[r].x), "+v"(v[r].y), "+v"(v[r].z), "+v"(v[r].w) : : "memory"); }
        };
        bool hit[R];
        if constexpr (DIRECT != 0) {
            // ======== direct-indexed form (L <= 10) ========================================================================
            // Per tile: the LDS cache of the exact spellings (both buckets of every read, one wait), then ONE batch of
            // global look-ups behind one wait -- the flat array at the read's own index (2 / 4 bytes) and, for a read with an
            // N, one 16-byte bucket of the N table.  Every load is unconditional: a lane that needs nothing reads entry 0 /
            // bucket 0 (one request for all such lanes of the wave).  Round 4, tools/phase_times.sh + tools/pmc_table2.sh:
            // the lane-masked shape paid up to 2 R + 2 R dependent L2 round trips per tile here; batching them took cfg 5
            // from 187 to 205 G reads/s and left the kernel bound by VALU issue (71 % busy), so this path is also written
            // for few instructions: cheap bucket arithmetic (two shifts, two masks), the 16-bit value picked first and
            // unpacked once, a four-operation hash for the N table.
#pragma unroll
            for (int r = 0; r < R; ++r) { hit[r] = false; res[r] = kMemoEmpty; }
            clk.mark(1);   // encode
            const uint32_t hb = Q.hot2_bits, hs = Q.d_nbits - Q.hot2_bits;   // wave-uniform
            const uint32_t nsh = 32u - (uint32_t)__builtin_popcount(Q.mask);   // memo_nbucket_shift
            const DirectLayout lay = memo_direct_layout(Q.d_ib, Q.d_bb);
            uint32_t e16[R];
#pragma unroll
            for (int r = 0; r < R; ++r) e16[r] = 0xFFFFu;
            if (Q.hot2 && !(ABL & 16)) {   // wave-uniform
                u32x2v b1[R], b2[R];
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    b1[r] = reinterpret_cast<const u32x2v *>(lds_hot)[didx[r] & hot2_mask];
                    b2[r] = reinterpret_cast<const u32x2v *>(lds_hot)[didx[r] >> hs];
                }
                arrived2(b1);
                arrived2(b2);
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const uint32_t w1 = didx[r] >> hb, w2 = (didx[r] & ((1u << hs) - 1u)) | 0x8000u;
                    const bool m0 = (b1[r].x >> 16) == w1, m1 = (b1[r].y >> 16) == w1, m2 = (b2[r].x >> 16) == w2, m3 = (b2[r].y >> 16) == w2;
                    e16[r] = m0 ? b1[r].x : (m1 ? b1[r].y : (m2 ? b2[r].x : b2[r].y));
                    hit[r] = (m0 || m1 || m2 || m3) && !has_n[r];   // an N aliases G in the 2-bit index: never trust it
                }
            }
            clk.mark(2);   // LDS cache
            uint32_t dv[R];
            u32x4v nb[R];
            const uint8_t *dbase = reinterpret_cast<const uint8_t *>(Q.direct), *nbase = reinterpret_cast<const uint8_t *>(Q.slots);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const uint32_t di = ((hit[r] || has_n[r]) && !(ABL & 1)) ? 0u : didx[r];
                if constexpr (DIRECT == 2) dv[r] = *reinterpret_cast<const uint16_t *>(dbase + (di << 1));   // 32-bit offsets from a uniform base
                else dv[r] = *reinterpret_cast<const uint32_t *>(dbase + (di << 2));
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const uint32_t bi = (has_n[r] && !(ABL & 256)) ? memo_nbucket1(key[r][0], nsh) : 0u;
                nb[r] = *reinterpret_cast<const u32x4v *>(nbase + (bi << 4));
            }
#pragma unroll
            for (int r = 0; r < R; ++r) { asm volatile("" : "+v"(dv[r]) : : "memory"); }
            arrived4(nb);
            clk.mark(3);   // the tile's gathers
            bool again[R], any_again = false;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const bool k0 = (nb[r].x & 0x7FFFFFFFu) == key[r][0], k1 = (nb[r].z & 0x7FFFFFFFu) == key[r][0];
                const uint32_t nres = k0 ? nb[r].y : (k1 ? nb[r].w : kMemoEmpty);
                uint32_t dres;
                if constexpr (DIRECT == 2) dres = memo_direct_unpack16(hit[r] ? (e16[r] & 0xFFFFu) : dv[r], lay);   // picked first, unpacked once
                else dres = dv[r];
                res[r] = has_n[r] ? ((ABL & 256) ? kMemoEmpty : nres) : dres;
                again[r] = has_n[r] && !k0 && !k1 && (nb[r].x >> 31) != 0 && !(ABL & (64 | 256));
                any_again |= again[r];
            }
            if (__ballot(any_again)) {   // wave-uniform, rare: some first bucket was full when the table was built
#pragma unroll
                for (int r = 0; r < R; ++r) nb[r] = *reinterpret_cast<const u32x4v *>(nbase + ((again[r] ? memo_nbucket2(key[r][0], nsh) : 0u) << 4));
                arrived4(nb);
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const bool k0 = (nb[r].x & 0x7FFFFFFFu) == key[r][0], k1 = (nb[r].z & 0x7FFFFFFFu) == key[r][0];
                    if (again[r] && (k0 || k1)) res[r] = k0 ? nb[r].y : nb[r].w;
                }
            }
        } else {
            // ======== hash-table form ======================================================================================
            // ---- probe: both candidate slots of every read are known up front.  Empty slots carry
            //      key = ~0 (no real key has a nibble's top bit set) and val = None. ---------------------
            // The FIRST slot only: the second comes from an independent hash of its own (memo_slot2) that only the rare wave
            // with a spilled key works out, and the LDS hot table is single-choice (the few exact-match entries that lose
            // their slot to another are served by the global table): a dozen VALU operations and one LDS read fewer per read.
            uint32_t s1[R], hf1[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                hf1[r] = memo_hash1_full(key[r][0], KW >= 2 ? key[r][1] : 0u, KW >= 3 ? key[r][2] : 0u, KW >= 4 ? key[r][3] : 0u);
                s1[r] = hf1[r] & Q.mask;
            }
            // With a presence filter (wave-uniform; tables whose exact-match entries leave LDS for one): the second hash of every read -- it
            // names the read's second hot slot and its second filter bit --, and `absent`: a read that is in no slot of the table.
            const bool filtered = Q.filter_bits != 0u && !(ABL & 16);
            uint32_t hf2[R];
            bool absent[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                absent[r] = false;
                hf2[r] = filtered ? memo_hash2_full(key[r][0], KW >= 2 ? key[r][1] : 0u, KW >= 3 ? key[r][2] : 0u, KW >= 4 ? key[r][3] : 0u) : 0u;
            }
            auto filter_probe = [&]() {   // (after the hot table: a read found there needs no filter)
                uint32_t w1[R], w2[R], p1[R], p2[R];
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    p1[r] = memo_filter_pos1(hf1[r], Q.filter_bits);
                    p2[r] = memo_filter_pos2(hf2[r], Q.filter_bits);
                    w1[r] = lds_filter[p1[r] >> 5];
                    w2[r] = lds_filter[p2[r] >> 5];
                }
#pragma unroll
                for (int r = 0; r < R; ++r) absent[r] = !hit[r] && !(((w1[r] >> (p1[r] & 31u)) & (w2[r] >> (p2[r] & 31u))) & 1u);
            };
            auto second_slot = [&](int r) {
                const uint32_t s2 = memo_slot2(key[r][0], KW >= 2 ? key[r][1] : 0u, KW >= 3 ? key[r][2] : 0u, KW >= 4 ? key[r][3] : 0u, Q.mask);
                return (ABL & 32) ? (s2 & 0xFFu) : ((ABL & 128) ? (s1[r] ^ 1u) : s2);
            };
#pragma unroll
            for (int r = 0; r < R; ++r) { hit[r] = false; res[r] = kMemoEmpty; }
            clk.mark(1);   // encode + hashes
            uint32_t g1[R];   // global-table slots (ABL 32: folded into a 4 KB corner = L1-resident)
#pragma unroll
            for (int r = 0; r < R; ++r) g1[r] = (ABL & 32) ? (s1[r] & 0xFFu) : s1[r];
            if constexpr (KW >= 2) {
                // A slot by key width (MemoParams::slots): k = the key quad (KW 2 / 3: the whole 16-byte slot), m = the
                // second quad of a four-word key's 32-byte slot.  Global slots are addressed as a 32-bit byte offset from
                // the table's (uniform) base: no 64-bit lane arithmetic (the builder keeps tables under 4 GiB).
                constexpr uint32_t kSlotShift = KW == 4 ? 5u : 4u;
                auto slot_hit = [&](const u32x4v &k, int r) {
                    return k.x == key[r][0] && k.y == key[r][1] && (KW < 3 || k.z == key[r][2]) && (KW < 4 || k.w == key[r][3]);
                };
                auto slot_val = [&](const u32x4v &k, const u32x4v &m) { return KW == 2 ? k.z : (KW == 3 ? (k.w & 0x7FFFFFFFu) : m.x); };
                auto slot_spill = [&](const u32x4v &k, const u32x4v &m) { return KW == 2 ? (k.w & 1u) != 0 : (KW == 3 ? (k.w >> 31) != 0 : (m.y & 1u) != 0); };
                const uint8_t *gbase = reinterpret_cast<const uint8_t *>(Q.slots);
                if (Q.hot_mask && !(ABL & 16)) {   // wave-uniform: the LDS hot table (0-mismatch entries), same slot format
                    const uint8_t *hbase = reinterpret_cast<const uint8_t *>(lds_hot);
                    u32x4v h1[R], n1[R];
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        h1[r] = *reinterpret_cast<const u32x4v *>(hbase + ((s1[r] & Q.hot_mask) << kSlotShift));
                        if constexpr (KW == 4) n1[r] = *reinterpret_cast<const u32x4v *>(hbase + ((s1[r] & Q.hot_mask) << kSlotShift) + 16);
                        else n1[r] = h1[r];
                    }
                    arrived4(h1);
                    if constexpr (KW == 4) arrived4(n1);
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        hit[r] = slot_hit(h1[r], r);
                        res[r] = hit[r] ? slot_val(h1[r], n1[r]) : kMemoEmpty;
                    }
                    if (filtered) {   // the second choice
#pragma unroll
                        for (int r = 0; r < R; ++r) {
                            h1[r] = *reinterpret_cast<const u32x4v *>(hbase + ((hf2[r] & Q.hot_mask) << kSlotShift));
                            if constexpr (KW == 4) n1[r] = *reinterpret_cast<const u32x4v *>(hbase + ((hf2[r] & Q.hot_mask) << kSlotShift) + 16);
                            else n1[r] = h1[r];
                        }
                        arrived4(h1);
                        if constexpr (KW == 4) arrived4(n1);
#pragma unroll
                        for (int r = 0; r < R; ++r)
                            if (!hit[r] && slot_hit(h1[r], r)) { hit[r] = true; res[r] = slot_val(h1[r], n1[r]); }
                    }
                }
                if (filtered) filter_probe();
                clk.mark(2);   // LDS hot table
                if constexpr (ABL & 1) {
#pragma unroll
                    for (int r = 0; r < R; ++r) res[r] = s1[r] | 0xFFFFu;
                } else {
                    // Global table, two-choice placement with a per-slot SPILL bit: the builder keeps a key in
                    // its first slot whenever it can and marks a slot whose would-be owner lives in its second
                    // slot.  So one gather settles ~90 % of the probing lanes (hit, or miss with spill = 0);
                    // only the rest issue the second, dependent gather -- with almost every lane reading slot 0.
                    bool again[R], any_again = false;
                    u32x4v e[R], f[R];
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const uint32_t off = ((hit[r] || absent[r]) ? 0u : g1[r]) << kSlotShift;
                        e[r] = *reinterpret_cast<const u32x4v *>(gbase + off);
                        if constexpr (KW == 4) f[r] = *reinterpret_cast<const u32x4v *>(gbase + off + 16); else f[r] = e[r];
                    }
                    arrived4(e);
                    if constexpr (KW == 4) arrived4(f);
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const bool k = slot_hit(e[r], r);
                        if (!hit[r] && !absent[r] && k) res[r] = slot_val(e[r], f[r]);
                        again[r] = !hit[r] && !absent[r] && !k && slot_spill(e[r], f[r]) && !(ABL & 64);
                        any_again |= again[r];
                    }
                    if (__ballot(any_again)) {   // wave-uniform
#pragma unroll
                        for (int r = 0; r < R; ++r) {
                            const uint32_t off = (again[r] ? second_slot(r) : 0u) << kSlotShift;
                            e[r] = *reinterpret_cast<const u32x4v *>(gbase + off);
                            if constexpr (KW == 4) f[r] = *reinterpret_cast<const u32x4v *>(gbase + off + 16); else f[r] = e[r];
                        }
                        arrived4(e);
                        if constexpr (KW == 4) arrived4(f);
#pragma unroll
                        for (int r = 0; r < R; ++r)
                            if (again[r] && slot_hit(e[r], r)) res[r] = slot_val(e[r], f[r]);
                    }
                }
            } else {
                if (Q.hot_mask && !(ABL & 16)) {   // wave-uniform
                    u32x2v h1[R];
#pragma unroll
                    for (int r = 0; r < R; ++r) h1[r] = reinterpret_cast<const u32x2v *>(lds_hot)[s1[r] & Q.hot_mask];
                    arrived2(h1);
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        hit[r] = h1[r].x == key[r][0];
                        res[r] = hit[r] ? h1[r].y : kMemoEmpty;
                    }
                    if (filtered) {   // the second choice
#pragma unroll
                        for (int r = 0; r < R; ++r) h1[r] = reinterpret_cast<const u32x2v *>(lds_hot)[hf2[r] & Q.hot_mask];
                        arrived2(h1);
#pragma unroll
                        for (int r = 0; r < R; ++r)
                            if (!hit[r] && h1[r].x == key[r][0]) { hit[r] = true; res[r] = h1[r].y; }
                    }
                }
                if (filtered) filter_probe();
                clk.mark(2);   // LDS hot table
                if constexpr (ABL & 1) {
#pragma unroll
                    for (int r = 0; r < R; ++r) res[r] = s1[r] | 0xFFFFu;
                } else {
                    bool again[R], any_again = false;
                    u32x2v e[R];
#pragma unroll
                    for (int r = 0; r < R; ++r) e[r] = reinterpret_cast<const u32x2v *>(Q.slots)[(hit[r] || absent[r]) ? 0u : g1[r]];
                    arrived2(e);
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const bool k = (e[r].x & 0x7FFFFFFFu) == key[r][0];
                        if (!hit[r] && !absent[r] && k) res[r] = e[r].y;
                        again[r] = !hit[r] && !absent[r] && !k && (e[r].x >> 31) != 0 && !(ABL & 64);
                        any_again |= again[r];
                    }
                    if (__ballot(any_again)) {   // wave-uniform
#pragma unroll
                        for (int r = 0; r < R; ++r) e[r] = reinterpret_cast<const u32x2v *>(Q.slots)[again[r] ? second_slot(r) : 0u];
                        arrived2(e);
#pragma unroll
                        for (int r = 0; r < R; ++r)
                            if (again[r] && (e[r].x & 0x7FFFFFFFu) == key[r][0]) res[r] = e[r].y;
                    }
                }
            }
        }
        clk.mark(4);   // cuckoo table (direct form: the reads with an N; hash form: every read that missed the hot table)
        // ---- rare: non-canonical reads -> wave-cooperative exhaustive scan.  Every lane was looked up
        //      above under its 4-bit key; '.' has N's code, so a read whose only non-canonical bytes are
        //      '.' no-calls already holds its answer -- only IUPAC / junk bytes need the scan. ----------
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (__ballot(bad[r])) {   // wave-uniform
                const bool really = bad[r] && noncanonical_beyond_dots<NWD>(words[r], kc, kv) != 0;
                uint64_t todo = __ballot(really);
                todo = defer_to_second_pass(P, work_seg, work_fill, todo, really, t * tile + local[r], res[r], words[r]);   // normally all of them
                if (todo) {
                    Planes<1> mine;
                    encode_planes<1>(words[r], nwords, L, lds_lut, mine);
                    while (todo) {
                        const int src = __ffsll((unsigned long long)todo) - 1;
                        todo &= todo - 1;
                        uint32_t b, s;
                        wave_scan<1>(mine, src, P, b, s, lds_tab);
                        if ((int)__lane_id() == src) res[r] = decide(b, s, P.max_mm, P.delta);
                    }
                }
            }
        }
        // ---- length rules of a variable-length batch, per-sample counts ---------------------------
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (!live[r]) continue;
            if constexpr (LENS) {   // the memo served the reads of length L
                const uint32_t len = words[r][kLenWord];
                if (len != L) {   // shorter -> None (barcode_matching.rs:167-169); longer -> None or the panic
                    res[r] = kMemoEmpty;
                    if (len > L) overlong_read(P, t * tile + local[r], len);
                }
            }
            if (P.counts && !(ABL & 4) && res[r] != kMemoDeferred) {
                const uint32_t idx = res[r] & 0xFFFFu;
                const uint32_t bin = idx == kNoMatch ? P.S : idx;
                if (P.lds_hist) atomicAdd(&lds_hist[bin], 1u);
                else atomicAdd(&P.counts[bin], 1ull);
            }
        }
        clk.mark(5);   // non-canonical reads, length rules, histogram
    };
    // The result stream of one tile.
    auto store_full = [&](uint64_t t, const uint32_t (&res)[R]) {
        if constexpr (ABL & 8) { if (res[0] == 0x12345u) P.out[t * tile + local[0]] = res[0]; return; }
        uint8_t *tile_out = reinterpret_cast<uint8_t *>(P.out + t * tile);   // wave-uniform
#pragma unroll
        for (int r = 0; r < R; ++r) FQTK_STREAM_STORE(res[r], reinterpret_cast<uint32_t *>((tile_out + r * 256) + out_off0));
    };
    auto store_any = [&](uint64_t t, const uint32_t (&res)[R], const bool (&live)[R]) {
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (live[r]) FQTK_STREAM_STORE(res[r], &P.out[t * tile + local[r]]);
    };

    const uint64_t full_tiles = (VEC >= 1) ? P.n / tile : 0;
    const uint32_t grid = gridDim.x;   // (read once: inside the loops hipcc reloaded it from the dispatch packet every tile)
    bool all_live[R];
#pragma unroll
    for (int r = 0; r < R; ++r) all_live[r] = true;
    if constexpr (PF && VEC >= 1) {
        // Software pipeline, one tile deep on both streams.  gfx950 counts loads AND stores in vmcnt and
        // they complete out of order with respect to each other, so "wait for my loads" also waits for every
        // store issued since: the plain loop (load, look up, store) pays a store acknowledgement plus a load
        // round trip, back to back, per tile.  Here a wave waits ONCE per tile, at the top, for operations it
        // issued a whole look-up phase earlier (the loads of this tile, the stores of the tile before), and only
        // then issues the next tile's loads and the previous tile's stores, which fly during the look-up.
        // Two word buffers used alternately (no register rotation); the prefetch is unconditional -- past the
        // end it re-reads the last full tile and the words are never used -- so that it cannot sit in a branch.
        uint32_t wa[R][kRowWords], wb[R][kRowWords], held[R];
        auto landed = [&](uint32_t (&w)[R][kRowWords]) {   // the words are in registers; nothing below moves above this point
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int k = 0; k < NWD; ++k) asm volatile("" : "+v"(w[r][k]) : : "memory");
            if constexpr (LENS) {
#pragma unroll
                for (int r = 0; r < R; ++r) asm volatile("" : "+v"(w[r][kLenWord]) : : "memory");
            }
        };
        auto computed = [&](uint32_t (&v)[R]) {     // the results exist now (their gathers / LDS reads were waited for HERE)
#pragma unroll
            for (int r = 0; r < R; ++r) asm volatile("" : "+v"(v[r]) : : "memory");
        };
        uint64_t t = blockIdx.x;
        if (t < full_tiles) {
            const uint64_t last = full_tiles - 1;
            uint32_t res[R];
            load_full(t, wa);
            landed(wa);
            load_full(min(t + gridDim.x, last), wb);
            lookup(t, wa, all_live, held);
            computed(held);
            uint64_t t_held = t;
            t += gridDim.x;
            while (t < full_tiles) {
                landed(wb);
                load_full(min(t + gridDim.x, last), wa);
                store_full(t_held, held);
                lookup(t, wb, all_live, res);
                computed(res);
#pragma unroll
                for (int r = 0; r < R; ++r) held[r] = res[r];
                t_held = t;
                t += gridDim.x;
                if (t >= full_tiles) break;
                landed(wa);
                load_full(min(t + gridDim.x, last), wb);
                store_full(t_held, held);
                lookup(t, wa, all_live, res);
                computed(res);
#pragma unroll
                for (int r = 0; r < R; ++r) held[r] = res[r];
                t_held = t;
                t += gridDim.x;
            }
            store_full(t_held, held);
        }
    } else {
        for (uint64_t t = blockIdx.x; t < full_tiles; t += grid) {
            uint32_t words[R][kRowWords], res[R];
            load_full(t, words);
            clk.mark(0);   // the row stream
            lookup(t, words, all_live, res);
            store_full(t, res);
            clk.mark(6);   // the result stream
        }
    }
    // whatever is left (the ragged last tile; every tile on the generic load paths)
    for (uint64_t t = full_tiles + (blockIdx.x + gridDim.x - full_tiles % gridDim.x) % gridDim.x; t < ntiles; t += gridDim.x) {
        uint32_t words[R][kRowWords], res[R];
        bool live[R];
        load_any(t, words, live);
        lookup(t, words, live, res);
        store_any(t, res, live);
    }
    publish_worklist_fill(P, work_seg, work_fill);
    clk.mark(7);
    clk.publish();

    if (P.counts && P.lds_hist) {
        __syncthreads();
        for (uint32_t b = tid; b < bins; b += kMemoBlock) {
            const uint32_t c = lds_hist[b];
            if (c) atomicAdd(&P.counts[b], (unsigned long long)c);
        }
    }
}

}  // namespace fqtk
