// bgzf_deflate.hpp -- DEFLATE block compressor for BGZF output, written for one 1024-lane workgroup per block.
//
// SURVEY.md section 8(f) row 3 (output side).  End to end `fqtk demux` is bound by BGZF compression on the
// host (the reference: pooled-writer -> bgzf -> libdeflater, /root/reference/src/bin/commands/demux.rs:755-798)
// while the GPU matcher idles.  This is the MI355X form of that stage: every <= 65 280-byte block becomes one
// dynamic-Huffman DEFLATE block (RFC 1951), produced by 1024 lanes (16 wavefronts: four per SIMD hide each other's LDS latencies):
//   P0  the block is copied into LDS, the LZ hash table and the histograms are cleared
//   P1a every lane counts the bytes of its 64-byte slice (-> estimated literal costs) and enters all its
//       positions into a shared table keyed by (16 KiB region of the block, hash of 4 bytes) holding the SMALLEST
//       and the LARGEST position seen -- min / max, so the table does not depend on how the lanes interleave and
//       the output is deterministic
//   P1b LZ77: every lane parses its slice greedily; candidates for a position are its own small table of recent
//       positions, the smallest entry of its own region and the largest entry of the region before it (always
//       behind it and inside the 32 KiB window); each is verified 16 bytes per round and must pay for itself
//       (estimated literal bits saved > bits of the match); tokens go to a global scratch, symbol counts to LDS
//   P2  all lanes rank the used literal/length symbols; one lane builds the two Huffman codes (two-queue
//       construction on the sorted counts, zlib's overflow rule for the 15-bit limit), the code-length code, and
//       writes the block header
//   P3  every lane adds up the bits of its tokens; exclusive prefix sum -> its bit offset; if the result would
//       not be smaller than the input, the block is emitted STORED instead
//   P4  every lane ORs its tokens' bits into the output image (LDS), the end-of-block code follows
//   P5  the image is copied out
// The compressed bytes are unpinned by the reference's tests (they compare decompressed content only,
// demux.rs:1069-1076); parity here = any inflate implementation returns the input bytes.
//
// Plain C++17 with a few device-only fast paths: compiles under hipcc for the device (fqtk_bgzf.hip runs the phases with barriers
// in between) and under g++ for the CPU test-suite, which runs the SAME phase functions lane by lane through
// libfqtk_host.so and inflates the result with zlib.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define FQTK_HD __host__ __device__
#else
#ifndef FQTK_HD
#define FQTK_HD
#endif
#endif

namespace fqtk {
namespace bgzf {

#ifndef FQTK_BGZF_MID
#define FQTK_BGZF_MID(k) do { } while (0)   // (tools/bgzf_phases.sh: a mark inside a phase)
#endif
#ifndef FQTK_BGZF_LANES
#define FQTK_BGZF_LANES 1024   // (overridable for studies: tools/ab_bgzf.sh, tools/bgzf_ratio.py -DFQTK_BGZF_LANES=512)
#endif
constexpr int kLanes = FQTK_BGZF_LANES;
constexpr uint32_t kMaxIn = 65280;        // uncompressed payload of a BGZF block (as the bgzf crate cuts them)
constexpr uint32_t kChunk = 65536 / kLanes;          // bytes parsed by one lane: 1020 lanes x 64 = 65 280
#ifndef FQTK_BGZF_REGION_SHIFT
#define FQTK_BGZF_REGION_SHIFT 11   // (tools/bgzf_ratio.py: 14 -> 11 takes 1.3 % off records with binned qualities, costs nothing)
#endif
constexpr uint32_t kRegionShift = FQTK_BGZF_REGION_SHIFT, kRegion = 1u << kRegionShift, kRegions = 65536u >> kRegionShift;
constexpr uint32_t kHashBits = kRegionShift - 3;        // per region: 32 regions x 256 buckets x {earliest, latest} = 32 KiB
constexpr uint32_t kNearSlots = 16384 / kLanes;       // per lane: direct-mapped table of its recent positions (local repeats)
static_assert(kRegionShift <= 11, "a position's offset in its region has 11 bits (+ 5 check bits = half a word)");
constexpr uint32_t kOutStride = 65536;    // bytes reserved per block in the output arena (stored worst case: n + 5)
constexpr uint32_t kMaxMatchesPerLane = kChunk / 4;     // kMinMatch bytes each at least
constexpr uint32_t kTokensPerBlock = kLanes * kMaxMatchesPerLane;   // match scratch (global, L2-resident), u32 each, [m][lane]
constexpr int kNumLitLen = 286, kNumDist = 30, kNumCl = 19;
constexpr int kCands = 3;                 // match candidates looked at per position
constexpr int kMinMatch = 4;              // shorter matches cost more bits than their literals on FASTQ

// The input block in LDS is SKEWED: logical word w lives at w + (w >> 4), one pad word per 64 bytes.  Every lane parses
// its own 64-byte slice, so at any moment the lanes of a wave touch addresses 64 bytes apart -- 16 words: two LDS
// banks for the whole wave, a 32-way conflict on every access (measured with rocprofv3: SQ_LDS_BANK_CONFLICT a
// third of the kernel's cycles).  With the pad word the stride is 17 words and consecutive lanes use consecutive banks.
constexpr uint32_t kBufWords = kOutStride / 4 + kOutStride / 64 + 16;
FQTK_HD inline uint32_t buf_word(uint32_t w) { return w + (w >> 4); }
// one byte of the (skewed) input block
FQTK_HD inline uint32_t buf_byte(const uint32_t *words, uint32_t pos) { return (words[buf_word(pos >> 2)] >> (8 * (pos & 3u))) & 0xFFu; }

// Everything a block's workgroup shares.  LDS on the device (158 KiB of the CU's 160: one workgroup per CU), heap in the CPU tests.
struct Shared {
    uint32_t buf[kBufWords];              // the input bytes, skewed (buf_word), from P0 to the end (literals are read from here)
    // P0-P1: the two match tables.  P2-P5: the same 64 KiB, as one array, hold the output bit stream (out_image()).
    uint32_t tminmax[(kRegions + 1u) << kHashBits];   // per (hash bucket, region; region_slot): the earliest position in the low half, the latest in the high half (region_entry)
    uint16_t near_tab[kNearSlots * kLanes];    // [slot][lane]: every lane's private table of recent positions (near_entry)
    uint32_t byte_cnt[256];                  // P1a: how often each byte value occurs in the block
    uint8_t lit_cost[256];                   // estimated cost of a literal, in half-bits (from byte_cnt)
    uint32_t lit_total;                      // sum of them over the block's bytes (for the average)
    uint32_t m_ll;                           // used literal/length symbols (S.sorted holds them in order after P2a)
    uint32_t freq_ll[288], freq_d[32];
    uint16_t code_ll[288], code_d[32];    // bit-reversed canonical codes (appended LSB first)
    uint8_t len_ll[288], len_d[32];
    uint32_t lane_bits[kLanes];           // bits of a lane's tokens, then their exclusive prefix sum
    uint16_t ntok[kLanes];                // matches of the lane's slice (the literals between them are not stored anywhere); << 8: the first that counts
    uint32_t span[kLanes];                // P1b: where the lane's last match ends; from P1c: the bytes the lane codes, [low half, high half)
    uint32_t header_bits, total_bits, stored;
    // scratch of the code builders: the literal/length code and the distance code are built side by side by two lanes
    // (of different wavefronts), each with its own scratch; the code-length code reuses the first set afterwards
    uint16_t sorted[288];                 // used symbols, ascending by (count, symbol)
    uint32_t weight[576];                 // leaves then internal nodes
    uint16_t parent[576];
    uint8_t depth[576];
    uint32_t bl_count[17], next_code[17]; // (indexed by run-time values: private arrays would live in scratch memory)
    uint16_t sorted_d[32];
    uint32_t weight_d[64];
    uint16_t parent_d[64];
    uint8_t depth_d[64];
    uint32_t bl_count_d[17];
    uint32_t m_d;                         // used distance symbols
    uint32_t hlit, hdist;                 // lengths transmitted: literal/length 257.., distance 1..
    // run-length coding of the hlit + hdist code lengths (RFC 1951 3.2.7), one lane per run
    uint8_t cl_sym[320], cl_extra[320];   // the coded sequence
    uint16_t run_syms[320];               // per position: symbols its run emits (0: not the start of a run)
    uint16_t run_len[320];
    uint32_t n_cl;                        // length of the coded sequence
    uint32_t fixed_header_bits;           // block header up to and including the code-length code's lengths
    uint32_t wave_tot[kLanes / 64 + 1];   // prefix sum of the lanes' bit counts: totals per wavefront
    uint32_t freq_cl[kNumCl];
    uint16_t code_cl[kNumCl];
    uint8_t len_cl[kNumCl];
    // CRC-32 of the block's bytes (the BGZF trailer): byte table, x^(8 * kChunk * j) and x^(8 * r) mod the CRC polynomial
    // (built once per workgroup by crc_tables(), they survive the blocks), the lanes' partial values
    uint32_t crc_tab[256];
    uint32_t crc_pow_chunk[kLanes];
    uint32_t crc_pow_byte[kChunk + 1];
    uint32_t crc;                         // (the lanes' partial values wait in lane_bits, which is free that early)
    uint32_t effort;                      // parse effort of the block (from --compression-level): 0 fast, 1 default
};

// The output bit stream lives where the match tables were: they are dead once the LZ phase is over, and the input stays
// intact to the end -- so a token stream need not be written out: a lane's tokens are its MATCHES (a few per slice, in a
// small global scratch) and, between them, the bytes of its slice.  (Tokens used to go to a 64 MB global scratch, one dword
// per token, written once and read twice: 9x the kernel's input in HBM traffic, rocprofv3 FETCH_SIZE / WRITE_SIZE.)
static_assert(offsetof(Shared, near_tab) == offsetof(Shared, tminmax) + sizeof(uint32_t) * ((kRegions + 1u) << kHashBits), "the two tables are one 64 KiB array");
static_assert(sizeof(uint32_t) * ((kRegions + 1u) << kHashBits) + sizeof(uint16_t) * kNearSlots * kLanes >= kOutStride, "the output image fits the tables' place");
static_assert(sizeof(Shared) <= 160 * 1024, "one workgroup's LDS");
static_assert(offsetof(Shared, run_syms) % 4 == 0 && offsetof(Shared, len_ll) % 4 == 0 && offsetof(Shared, len_d) % 4 == 0, "code lengths are read four at a time (canonical_code_of)");
FQTK_HD inline uint32_t *out_image(Shared &S) { return S.tminmax; }

// ---- symbol arithmetic (RFC 1951 3.2.5), computed rather than tabulated -------------------------------------
FQTK_HD inline int floor_log2(uint32_t x) { return 31 - __builtin_clz(x); }   // x >= 1
// length 3..258 -> literal/length symbol 257..285, number of extra bits and their value
FQTK_HD inline void length_symbol(uint32_t len, uint32_t &sym, uint32_t &nextra, uint32_t &extra) {
    const uint32_t l = len - 3;
    if (l < 8) { sym = 257 + l; nextra = 0; extra = 0; return; }
    if (len == 258) { sym = 285; nextra = 0; extra = 0; return; }
    const int m = floor_log2(l);              // 3..7
    sym = 265 + 4 * (uint32_t)(m - 3) + ((l >> (m - 2)) & 3u);
    nextra = (uint32_t)(m - 2);
    extra = l & ((1u << (m - 2)) - 1u);
}
// distance 1..32768 -> distance symbol 0..29, extra bits
FQTK_HD inline void dist_symbol(uint32_t dist, uint32_t &sym, uint32_t &nextra, uint32_t &extra) {
    const uint32_t d = dist - 1;
    if (d < 4) { sym = d; nextra = 0; extra = 0; return; }
    const int m = floor_log2(d);              // 2..14
    sym = 2 * (uint32_t)m + ((d >> (m - 1)) & 1u);
    nextra = (uint32_t)(m - 1);
    extra = d & ((1u << (m - 1)) - 1u);
}
FQTK_HD inline uint32_t reverse_bits(uint32_t code, int len) {   // len >= 1
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_bitreverse32(code) >> (32 - len);
#else
    uint32_t r = 0;
    for (int i = 0; i < len; ++i) { r = (r << 1) | (code & 1u); code >>= 1; }
    return r;
#endif
}

// a match of a lane: offset of its first byte from the start of the lane's slice (9 bits: low eight << 24, ninth << 15)
// | (len - 3) << 16 | (dist - 1).  The offset passes the slice's end when the match was cut at its front (phase_reach).
FQTK_HD inline uint32_t match_token(uint32_t len, uint32_t dist, uint32_t slice_off) {
    return ((slice_off & 0xFFu) << 24) | ((slice_off >> 8) << 15) | ((len - 3) << 16) | (dist - 1);
}
FQTK_HD inline uint32_t token_off(uint32_t t) { return (t >> 24) | ((t >> 7) & 0x100u); }
FQTK_HD inline uint32_t token_len(uint32_t t) { return ((t >> 16) & 0xFFu) + 3u; }
FQTK_HD inline uint32_t token_dist(uint32_t t) { return (t & 0x7FFFu) + 1u; }
static_assert(kChunk + 258 <= 512, "a match's offset from its slice's start has 9 bits");

// ---- append bits to the output image -------------------------------------------------------------------------
// The image is zero before P2; several lanes may touch one word, so words are ORed in (atomic on the device).
#if defined(__HIP_DEVICE_COMPILE__)
#define FQTK_BGZF_OR(ptr, v) atomicOr((ptr), (v))
#define FQTK_BGZF_ADD(ptr, v) atomicAdd((ptr), (v))
#define FQTK_BGZF_CAS(ptr, expect, v) atomicCAS((ptr), (expect), (v))
#define FQTK_BGZF_MAX(ptr, v) atomicMax((ptr), (v))
#else
#define FQTK_BGZF_MAX(ptr, v) (*(ptr) = *(ptr) > (v) ? *(ptr) : (v))
#define FQTK_BGZF_OR(ptr, v) (*(ptr) |= (v))
#define FQTK_BGZF_ADD(ptr, v) (*(ptr) += (v))
#define FQTK_BGZF_CAS(ptr, expect, v) (*(ptr) == (expect) ? (*(ptr) = (v), (expect)) : *(ptr))
#endif
struct BitWriter {
    uint32_t *words;
    uint64_t acc;      // pending bits, aligned to the 32-bit word `word`
    uint32_t nacc;     // number of valid low bits in acc (including the offset inside the first word)
    uint32_t word;
    FQTK_HD void start(uint32_t *w, uint32_t bitpos) { words = w; word = bitpos >> 5; nacc = bitpos & 31u; acc = 0; }
    FQTK_HD void put(uint32_t value, uint32_t nbits) {   // nbits <= 32 (31 pending bits + 32 fit the 64-bit accumulator)
        acc |= (uint64_t)value << nacc;
        nacc += nbits;
        if (nacc >= 32) {
            FQTK_BGZF_OR(&words[word], (uint32_t)acc);
            acc >>= 32;
            nacc -= 32;
            ++word;
        }
    }
    FQTK_HD void finish() { if (nacc) FQTK_BGZF_OR(&words[word], (uint32_t)acc); }
    FQTK_HD uint32_t bitpos() const { return (word << 5) + nacc; }
};

// ---- Huffman code lengths ---------------------------------------------------------------------------------------
// counts[0..n) -> len[0..n) (0 = unused symbol), no code longer than max_bits, Kraft sum exactly 1 (inflate
// implementations reject incomplete literal/length sets).  At least two symbols get a code (as zlib does).
// presorted_m >= 0: S.sorted[0..presorted_m) already holds the used symbols ascending by (count, symbol).
struct HuffScratch { uint16_t *sorted; uint32_t *weight; uint16_t *parent; uint8_t *depth; uint32_t *bl_count; };
FQTK_HD inline HuffScratch huff_scratch_ll(Shared &S) { return HuffScratch{S.sorted, S.weight, S.parent, S.depth, S.bl_count}; }
FQTK_HD inline HuffScratch huff_scratch_d(Shared &S) { return HuffScratch{S.sorted_d, S.weight_d, S.parent_d, S.depth_d, S.bl_count_d}; }
// len_cleared: len[0..n) is all zero already (the workgroup cleared it).  On return W.bl_count[b] = number of codes of b bits.
FQTK_HD inline void huffman_lengths(const HuffScratch &W, const uint32_t *counts, int n, int max_bits, uint8_t *len, int presorted_m = -1, bool len_cleared = false) {
    for (int i = 0; !len_cleared && i < n; ++i) len[i] = 0;
    for (int b = 0; b <= 16; ++b) W.bl_count[b] = 0;
    int m = presorted_m < 0 ? 0 : presorted_m;
    for (int i = 0; presorted_m < 0 && i < n; ++i) {   // insertion sort of the used symbols by (count, symbol)
        if (!counts[i]) continue;
        int j = m++;
        while (j > 0 && counts[W.sorted[j - 1]] > counts[i]) { W.sorted[j] = W.sorted[j - 1]; --j; }
        W.sorted[j] = (uint16_t)i;
    }
    if (m == 0) { len[0] = 1; len[1] = 1; W.bl_count[1] = 2; return; }
    if (m == 1) { len[W.sorted[0]] = 1; len[W.sorted[0] == 0 ? 1 : 0] = 1; W.bl_count[1] = 2; return; }
    // two-queue construction: leaves 0..m-1 in ascending order, internal nodes m.. in creation (= ascending) order
    for (int i = 0; i < m; ++i) W.weight[i] = counts[W.sorted[i]];
    int li = 0, ii = m, made = m;
    for (int k = 0; k < m - 1; ++k) {
        int pick[2];
        for (int t = 0; t < 2; ++t) {
            if (li < m && (ii >= made || W.weight[li] <= W.weight[ii])) pick[t] = li++;
            else pick[t] = ii++;
        }
        W.weight[made] = W.weight[pick[0]] + W.weight[pick[1]];
        W.parent[pick[0]] = (uint16_t)made;
        W.parent[pick[1]] = (uint16_t)made;
        ++made;
    }
    const int root = made - 1;
    // Depths top-down with zlib's rule for the length limit (trees.c gen_bitlen): a node deeper than max_bits is put
    // AT max_bits -- its children then see that depth -- and every such node, INTERNAL ones included, counts as one
    // overflow; the repair loop below then moves leaves down until the code is complete again.  (Counting only the
    // leaves beyond the limit, as this did before, leaves the code over-subscribed by one when an internal node
    // sits at the limit: inflate rejects the block.)
    uint32_t *bl_count = W.bl_count;
    int overflow = 0;
    W.depth[root] = 0;
    for (int v = root - 1; v >= 0; --v) {
        int bits = (int)W.depth[W.parent[v]] + 1;
        if (bits > max_bits) { bits = max_bits; ++overflow; }
        W.depth[v] = (uint8_t)bits;
        if (v < m) ++bl_count[bits];   // a leaf
    }
    while (overflow > 0) {
        int bits = max_bits - 1;
        while (bl_count[bits] == 0) --bits;
        --bl_count[bits];
        bl_count[bits + 1] += 2;
        --bl_count[max_bits];
        overflow -= 2;
    }
    int idx = 0;   // sorted[] ascends by count: rarest first
    for (int bits = max_bits; bits >= 1; --bits)
        for (uint32_t c = bl_count[bits]; c > 0; --c) len[W.sorted[idx++]] = (uint8_t)bits;
}

// canonical codes (RFC 1951 3.2.2), stored bit-reversed.  All lanes, one symbol each: the code of a symbol is the
// first code of its length (from the counts per length) plus the number of lower symbols of the same length.
FQTK_HD inline uint16_t canonical_code_of(const uint8_t *len, int sym, const uint32_t *bl_count, int max_bits) {
    const int l = len[sym];
    if (!l) return 0;
    uint32_t c = 0;
    for (int b = 1; b <= l && b <= max_bits; ++b) c = (c + (b > 1 ? bl_count[b - 1] : 0u)) << 1;
    // how many lower symbols have the same length: four lengths per read, compared as one word, eight reads in flight -- a byte per round trip made this loop, up to 285 rounds on one lane's
    // latency, a tenth of a block's time
    const uint32_t *words = reinterpret_cast<const uint32_t *>(len);   // (len_ll / len_d / len_cl begin on a word: static_assert below)
    const uint32_t same = (uint32_t)l * 0x01010101u;
    const int full = sym >> 2;
    uint32_t rank = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 8
#endif
    for (int q = 0; q < full; ++q) {
        const uint32_t y = words[q] ^ same;   // (every byte < 16: adding 0x7F sets a byte's top bit iff it is not zero, without a carry)
        rank += 4u - (uint32_t)__builtin_popcount((y + 0x7F7F7F7Fu) & 0x80808080u);
    }
    if (sym & 3) {
        const uint32_t y = (words[full] ^ same) | (0x01010101u << (8 * (sym & 3)));   // (the symbol itself and those above it do not count)
        rank += 4u - (uint32_t)__builtin_popcount((y + 0x7F7F7F7Fu) & 0x80808080u);
    }
    return (uint16_t)reverse_bits(c + rank, l);
}
// the same for a whole set by one lane (the 19-symbol code-length code)
FQTK_HD inline void canonical_codes(Shared &S, const uint8_t *len, int n, int max_bits, uint16_t *code) {   // one lane
    uint32_t *bl_count = S.bl_count, *next_code = S.next_code;
    for (int b = 0; b <= 16; ++b) bl_count[b] = 0;
    for (int i = 0; i < n; ++i) ++bl_count[len[i]];
    bl_count[0] = 0;
    uint32_t c = 0;
    for (int b = 1; b <= max_bits; ++b) { c = (c + bl_count[b - 1]) << 1; next_code[b] = c; }
    for (int i = 0; i < n; ++i) code[i] = len[i] ? (uint16_t)reverse_bits(next_code[len[i]]++, len[i]) : 0;
}

// ---- CRC-32 of the block (RFC 1952 8.; the reflected polynomial 0xEDB88320) ---------------------------------------------
// Every lane takes the CRC of its own slice; CRC(A || B) = CRC(A) * x^(8 |B|) + CRC(B) over GF(2) modulo the polynomial,
// so the slices' values, each multiplied by x^(8 * bytes behind it), XOR to the CRC of the block.  Polynomials are held
// the way the CRC register holds them: bit 31 is the coefficient of x^0.
FQTK_HD inline uint32_t crc_gf_mul(uint32_t a, uint32_t b) {
    uint32_t p = 0;
    for (int i = 0; i < 32; ++i) {
        if (a & 0x80000000u) p ^= b;
        a <<= 1;
        b = (b & 1u) ? (b >> 1) ^ 0xEDB88320u : (b >> 1);   // b * x
    }
    return p;
}
FQTK_HD inline uint32_t crc_x_pow(uint32_t nbits) {   // x^nbits
    uint32_t r = 0x80000000u, base = 0x40000000u;
    for (uint32_t e = nbits; e; e >>= 1) {
        if (e & 1u) r = crc_gf_mul(r, base);
        base = crc_gf_mul(base, base);
    }
    return r;
}
// once per workgroup (all lanes; a barrier must follow)
FQTK_HD inline void crc_tables(Shared &S, int lane) {
    if (lane < 256) {
        uint32_t c = (uint32_t)lane;
        for (int k = 0; k < 8; ++k) c = (c & 1u) ? (c >> 1) ^ 0xEDB88320u : (c >> 1);
        S.crc_tab[lane] = c;
    }
    S.crc_pow_chunk[lane] = crc_x_pow(8u * kChunk * (uint32_t)lane);
    for (uint32_t r = (uint32_t)lane; r <= kChunk; r += kLanes) S.crc_pow_byte[r] = crc_x_pow(8u * r);
}
// after phase_load (the block is in S.buf); a barrier, then phase_crc_fold by one lane
FQTK_HD inline void phase_crc(Shared &S, int lane, uint32_t n) {
    const uint32_t lo = (uint32_t)lane * kChunk;
    uint32_t part = 0;
    if (lo < n) {
        const uint32_t hi = lo + kChunk < n ? lo + kChunk : n;
        uint32_t c = 0xFFFFFFFFu;
        for (uint32_t p = lo; p < hi; ++p) c = S.crc_tab[(c ^ buf_byte(S.buf, p)) & 0xFFu] ^ (c >> 8);
        c = ~c;
        if (hi < n) {   // bytes behind this slice: whole slices of the lanes between it and the last one, then the last one's
            const uint32_t last = (n - 1u) / kChunk;
            c = crc_gf_mul(c, S.crc_pow_chunk[last - 1u - (uint32_t)lane]);
            c = crc_gf_mul(c, S.crc_pow_byte[n - last * kChunk]);
        }
        part = c;
    }
    S.lane_bits[lane] = part;
}
FQTK_HD inline void phase_crc_fold(Shared &S) {
    uint32_t c = 0;
    for (int l = 0; l < kLanes; ++l) c ^= S.lane_bits[l];
    S.crc = c;
}

// ---- the phases ----------------------------------------------------------------------------------------------------
FQTK_HD inline uint32_t load_le32(const uint8_t *p) {
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
// The hash of a position's gram: bucket = the top kHashBits bits, slot of the lane's own table = the bits below them,
// check = sixteen bits further down.  A CHEAP position -- its next four bytes cost at most kCheap4 half-bits as literals:
// sequence lines, a run of the frequent quality value -- is known by its next EIGHT bytes: a match of four to seven such
// bytes never pays for its distance code, yet nearly every position of a sequence line finds an earlier copy of its four
// bases, and reading and comparing those candidates was most of the LZ phase's time (tools/ab_bgzf.sh: 45 GB/s with the
// three candidates, 102 GB/s with none).  The check bits travel with every table entry, so a candidate's bytes are only
// read when its gram (almost certainly) is the position's own.
#ifndef FQTK_BGZF_CHEAP4
#define FQTK_BGZF_CHEAP4 37u
#endif
constexpr uint32_t kCheap4 = FQTK_BGZF_CHEAP4;
FQTK_HD inline uint32_t gram_hash(uint32_t w, uint32_t w4, bool cheap) {
    const uint32_t x = cheap ? (w ^ (w4 * 0x85EBCA6Bu) ^ 0x5BD1E995u) : w;
    return x * 2654435761u;
}
// Entries are half words: the region table's = offset in the 2 KiB region << 5 | 5 check bits (none: 0xFFFF as an earliest,
// 0 as a latest entry -- the two grams that would look like that are not entered), the lane's own table's = distance from
// 64 bytes before the lane's slice << 7 | 7 check bits (a lane's table holds the slice before its own, its own, and the
// 257 bytes a match may run on: 9 bits; none: 0xFFFF, a position no look-up can be behind).
static_assert(kHashBits + 4u + 7u <= 32u && kChunk + kChunk + 258u <= 512u, "bucket / slot / check bits; 9-bit distances");
FQTK_HD inline uint32_t bucket_of(uint32_t h) { return h >> (32 - kHashBits); }
FQTK_HD inline uint32_t near_of(uint32_t h) { return (h >> (32 - kHashBits - 4)) & (kNearSlots - 1u); }
FQTK_HD inline uint32_t region_entry(uint32_t p, uint32_t h) { return ((p & (kRegion - 1u)) << 5) | ((h >> 8) & 0x1Fu); }
FQTK_HD inline uint32_t near_entry(uint32_t p, int lane, uint32_t h) { return ((p + kChunk - (uint32_t)lane * kChunk) << 7) | ((h >> 8) & 0x7Fu); }
// Four bytes at any byte offset of the block buffer.  Device: two aligned LDS words and one v_alignbyte_b32
// instead of four byte reads (the word after the last payload byte exists: the buffer is 64 KiB, a block 65 280 B).
FQTK_HD inline uint32_t buf_le32(const uint32_t *words, uint32_t pos) {
    const uint32_t w = pos >> 2;
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbyte(words[buf_word(w + 1)], words[buf_word(w)], pos & 3u);
#else
    const uint64_t two = (uint64_t)words[buf_word(w)] | ((uint64_t)words[buf_word(w + 1)] << 32);
    return (uint32_t)(two >> (8 * (pos & 3u)));
#endif
}
// NW dwords from byte `pos` on: NW + 1 aligned words and a funnel shift each (two reads per dword the buf_le32 way; the LZ
// phase is bound by LDS reads whose banks collide -- the lanes stand at unrelated offsets of their slices)
template <int NW>
FQTK_HD inline void buf_run(const uint32_t *words, uint32_t pos, uint32_t (&out)[NW]) {
    const uint32_t w = pos >> 2, sh = pos & 3u;
    uint32_t a[NW + 1];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int i = 0; i <= NW; ++i) a[i] = words[buf_word(w + (uint32_t)i)];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
    for (int i = 0; i < NW; ++i) out[i] = __builtin_amdgcn_alignbyte(a[i + 1], a[i], sh);
#else
    for (int i = 0; i < NW; ++i) out[i] = (uint32_t)(((uint64_t)a[i] | ((uint64_t)a[i + 1] << 32)) >> (8 * sh));
#endif
}
// four bytes from byte `o` (0 .. 3) of the eight in (lo, hi)
FQTK_HD inline uint32_t bytes_at(uint32_t lo, uint32_t hi, uint32_t o) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbyte(hi, lo, o);
#else
    return (uint32_t)(((uint64_t)lo | ((uint64_t)hi << 32)) >> (8 * o));
#endif
}
FQTK_HD inline uint32_t ctz32(uint32_t x) {   // x != 0
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__builtin_ctz(x);
#else
    uint32_t k = 0;
    while (!(x & 1u)) { x >>= 1; ++k; }
    return k;
#endif
}

// --compression-level (demux.rs:641-643) -> parse effort: 1-3 every match is cut at its lane's slice (7 % faster, output
// +2-5 %), 4 and up a slice's last match runs on (phase_reach; the default is 5); level 0 does not come here (stored blocks).
FQTK_HD inline uint32_t effort_of_level(uint32_t level) { return level <= 3u ? 0u : 1u; }

// P0: clear the shared state, bring the block in.  `in` may be device or (pinned, device-visible) host memory.
FQTK_HD inline void phase_load(Shared &S, int lane, const uint8_t *in, uint32_t n) {
    for (uint32_t i = (uint32_t)lane; i < ((kRegions + 1u) << kHashBits); i += kLanes) S.tminmax[i] = 0x0000FFFFu;   // earliest = none (0xFFFF), latest = none (0)
    for (uint32_t i = 0; i < kNearSlots; ++i) S.near_tab[i * kLanes + (uint32_t)lane] = 0xFFFFu;
    for (uint32_t i = (uint32_t)lane; i < 288; i += kLanes) S.freq_ll[i] = 0;
    if (lane < 32) S.freq_d[lane] = 0;
    if (lane < 256) S.byte_cnt[lane] = 0;
    S.span[lane] = 0;   // (until the LZ phase: kCountCopies - 1 more sets of byte counters, phase_count)
    if (lane == 0) { S.lit_total = 0; S.m_ll = 0; S.m_d = 0; S.n_cl = 0; S.hlit = 257; S.hdist = 1; }
    if ((reinterpret_cast<uintptr_t>(in) & 3u) == 0) {
        const uint32_t nw = n >> 2;
        if ((reinterpret_cast<uintptr_t>(in) & 15u) == 0) {
            const uint32_t n16 = n >> 4;
            for (uint32_t i = (uint32_t)lane; i < n16; i += kLanes) {   // four words of one 64-byte group: one skew
                const uint32_t *src = reinterpret_cast<const uint32_t *>(in) + 4 * i;
                uint32_t *dst = S.buf + buf_word(4 * i);
                dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2]; dst[3] = src[3];
            }
            for (uint32_t i = (n16 << 2) + (uint32_t)lane; i < nw; i += kLanes) S.buf[buf_word(i)] = reinterpret_cast<const uint32_t *>(in)[i];
        } else {
            for (uint32_t i = (uint32_t)lane; i < nw; i += kLanes) S.buf[buf_word(i)] = reinterpret_cast<const uint32_t *>(in)[i];
        }
        if (lane == 0 && (n & 3u)) {   // the ragged last word
            uint32_t v = 0;
            for (uint32_t k = 0; k < (n & 3u); ++k) v |= (uint32_t)in[(nw << 2) + k] << (8 * k);
            S.buf[buf_word(nw)] = v;
        }
    } else {   // (a source that is not word-aligned: whole words assembled from bytes)
        const uint32_t nw = (n + 3u) >> 2;
        for (uint32_t i = (uint32_t)lane; i < nw; i += kLanes) {
            uint32_t v = 0;
            for (uint32_t k = 0; k < 4 && 4 * i + k < n; ++k) v |= (uint32_t)in[4 * i + k] << (8 * k);
            S.buf[buf_word(i)] = v;
        }
    }
}

// P1a: how often each byte value occurs in the block (-> what a literal costs, phase_literal_costs).  The slice is read
// once, as aligned words.  (The four bases are nearly half of FASTQ text and 64 lanes of a wave hammer their four counters
// -- yet counting them in a packed register per lane and adding once per slice was SLOWER, 36.6 against 38.4 GB/s,
// tools/ab_bgzf.sh: same-address LDS atomics are cheap, the extra selects are not.)
#ifndef FQTK_BGZF_COUNT_COPIES
#define FQTK_BGZF_COUNT_COPIES 4   // (1: every lane adds to the one set; tools/ab_bgzf.sh)
#endif
constexpr uint32_t kCountCopies = FQTK_BGZF_COUNT_COPIES;   // sets of byte counters: S.byte_cnt and, in S.span (unused until the LZ phase), kCountCopies - 1 more
static_assert(kCountCopies >= 1 && (kCountCopies & (kCountCopies - 1u)) == 0 && (kCountCopies - 1u) * 256u <= (uint32_t)kLanes, "the extra sets live in S.span");
FQTK_HD inline uint32_t *count_set(Shared &S, int lane) {   // neighbouring lanes stand in the same line of text and count the same few values: each its own set
    const uint32_t k = (uint32_t)lane & (kCountCopies - 1u);
    return k ? S.span + (k - 1u) * 256u : S.byte_cnt;
}
FQTK_HD inline void phase_count(Shared &S, int lane, uint32_t n) {
    uint32_t *cnt = count_set(S, lane);
    const uint32_t lo = (uint32_t)lane * kChunk;
    const uint32_t hi = lo + kChunk < n ? lo + kChunk : n;
    for (uint32_t g = lo; g < hi; g += 16u) {
        uint32_t v[4];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int i = 0; i < 4; ++i) v[i] = S.buf[buf_word((g >> 2) + (uint32_t)i)];
#ifndef FQTK_BGZF_NO_COUNT_RUNS
        // (sixteen equal bytes -- inside the run of a quality line -- are one addition: 27 lanes of a wavefront stand in quality lines, and
        //  where those hold one value they all add to ONE counter, sixteen times each)
        const uint32_t rep = (v[0] & 0xFFu) * 0x01010101u;
        if (g + 16u <= hi && ((v[0] ^ rep) | (v[1] ^ rep) | (v[2] ^ rep) | (v[3] ^ rep)) == 0u) {
            FQTK_BGZF_ADD(&cnt[v[0] & 0xFFu], 16u);
            continue;
        }
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (uint32_t k = 0; k < 16u; ++k)
            if (g + k < hi) FQTK_BGZF_ADD(&cnt[(v[k >> 2] >> (8 * (k & 3u))) & 0xFFu], 1u);
    }
}

// The grams of positions from .. from + count - 1 (count <= MAXK): fn(position, gram_hash, is it a cheap position).  The bytes come in as one run of
// aligned words, the literal costs that decide between the four-byte and the eight-byte gram one read per byte.
template <int MAXK, typename Fn>
FQTK_HD inline void for_grams(Shared &S, uint32_t n, uint32_t from, uint32_t count, Fn fn) {
    constexpr int NW = (MAXK + 7 + 3) / 4;
    uint32_t d[NW];
    buf_run<NW>(S.buf, from, d);
    uint32_t c[MAXK + 3];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int j = 0; j < MAXK + 3; ++j) c[j] = S.lit_cost[(d[j >> 2] >> (8 * (j & 3))) & 0xFFu];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = 0; k < MAXK; ++k) {
        const int k4 = k + 4;
        const uint32_t w = (k & 3) ? (d[k >> 2] >> (8 * (k & 3))) | (d[(k >> 2) + 1] << (32 - 8 * (k & 3))) : d[k >> 2];
        const uint32_t w4 = (k4 & 3) ? (d[k4 >> 2] >> (8 * (k4 & 3))) | (d[(k4 >> 2) + 1] << (32 - 8 * (k4 & 3))) : d[k4 >> 2];
        const uint32_t p = from + (uint32_t)k;
        if ((uint32_t)k < count && p + 4 <= n) {   // (no break: the loop must unroll for d[] and c[] to stay in registers)
            const bool cheap = c[k] + c[k + 1] + c[k + 2] + c[k + 3] <= kCheap4 && p + 8 <= n;
            fn(p, gram_hash(w, w4, cheap), cheap);
        }
    }
}

// P1b: every position of this lane's slice into the (region, hash bucket) table.  A bucket keeps the earliest and the latest
// of its positions, each with the check bits of its gram, position in the high half: one atomic min and one atomic max per
// position, and both are order-independent -- so the table, and with it the whole output, does not depend on how the lanes
// interleave.
// Word of (region of p, bucket): the regions of a bucket side by side, so that the two entries a look-up wants -- its own
// region's and the one before -- are neighbours (one ds_read2_b32); kRegions + 1 words per bucket, or every lane of a
// wavefront (they stand in one or two regions) would hit the same bank.
FQTK_HD inline uint32_t region_slot(uint32_t p, uint32_t bucket) { return bucket * (kRegions + 1u) + 1u + (p >> kRegionShift); }
// The same grams go into the NEXT lane's own table (its history begins with the slice before its own: this lane's), and
// which positions are cheap comes back as a bit mask -- the LZ phase then hashes a position without a look at the
// literal costs (one dependent LDS round trip less per step).
FQTK_HD inline uint64_t phase_index(Shared &S, int lane, uint32_t n) {
    const uint32_t lo = (uint32_t)lane * kChunk;
    const uint32_t hi = lo + kChunk < n ? lo + kChunk : n;
    uint64_t cheap_mask = 0;
    for (uint32_t g = lo; g < hi; g += 16u)
        for_grams<16>(S, n, g, hi - g < 16u ? hi - g : 16u, [&](uint32_t p, uint32_t h, bool cheap) {
            if (cheap) cheap_mask |= 1ull << (p - lo);
            if (lane + 1 < kLanes) S.near_tab[near_of(h) * kLanes + (uint32_t)lane + 1u] = (uint16_t)near_entry(p, lane + 1, h);
            const uint32_t v = region_entry(p, h);
            if (v == 0u || v == 0xFFFFu) return;   // (would read as "none")
            uint32_t *e = &S.tminmax[region_slot(p, bucket_of(h))];
            uint32_t old = *e;
            for (;;) {
                const uint32_t mn = (old & 0xFFFFu) < v ? (old & 0xFFFFu) : v;
                const uint32_t mx = (old >> 16) > v ? (old >> 16) : v;
                const uint32_t want = (mx << 16) | mn;
                if (want == old) break;
                const uint32_t seen = FQTK_BGZF_CAS(e, old, want);
                if (seen == old) break;
                old = seen;
            }
        });
    return cheap_mask;
}

// What a literal will cost after Huffman coding, roughly: -log2 of the byte's share of the block, in half-bits
// (integer arithmetic only: the CPU tests expect the device's bytes).  A match is taken only when the literals it
// replaces would cost more than the match itself -- in FASTQ text the bases cost ~2.3 bits each and a frequent
// quality value less than that, so short matches rarely pay.
FQTK_HD inline uint32_t log2_halfbits(uint32_t x) {   // ~ 2 * log2(x), x >= 1
    const int m = floor_log2(x);
    return 2u * (uint32_t)m + (m > 0 ? ((x >> (m - 1)) & 1u) : 0u);
}
FQTK_HD inline void phase_literal_costs(Shared &S, int lane, uint32_t n) {   // lane = byte value
    if (lane >= 256) return;
    uint32_t c = S.byte_cnt[lane];
    for (uint32_t k = 1; k < kCountCopies; ++k) c += S.span[(k - 1u) * 256u + (uint32_t)lane];
    uint32_t cost = 30;
    if (c) {
        const uint32_t a = log2_halfbits(n), b = log2_halfbits(c);
        cost = a > b ? a - b : 0u;
#ifndef FQTK_BGZF_MINLIT
#define FQTK_BGZF_MINLIT 4u   // two bits: the frequent quality value of binned records shares the short codes with the four bases (2u: +3 % output)
#endif
        if (cost < FQTK_BGZF_MINLIT) cost = FQTK_BGZF_MINLIT;
        if (cost > 30u) cost = 30u;
    }
    S.lit_cost[lane] = (uint8_t)cost;
    if (c) FQTK_BGZF_ADD(&S.lit_total, c * cost);
}
FQTK_HD inline uint32_t match_cost(uint32_t len, uint32_t dist) {   // half-bits: length code + distance code + extras
    uint32_t sym, ne_l, ne_d, ev;
    length_symbol(len, sym, ne_l, ev);
    dist_symbol(dist, sym, ne_d, ev);
    return 2u * (7u + 5u + ne_l + ne_d);
}

// P1b: greedy LZ77 over this lane's slice; tokens to tok[t * kLanes + lane].  Deterministic: reads the tables
// of P1a and the lane's own state only.
struct LzLane { uint32_t p, end, nt, avg16, effort, pending, miss, open; uint64_t cheap; };   // open: a match still being compared (lz_step), (candidate + 1) << 9 | bytes so far; miss: literals since the lane's last match (lz_step's skipping); avg16: the block's average literal cost, half-bits x 16; cheap: phase_index's mask
FQTK_HD inline void lz_begin(Shared &S, int lane, uint32_t n, LzLane &st, uint64_t cheap_mask) {
    st.cheap = cheap_mask;
    st.pending = 0;
    st.miss = 0;
    st.open = 0;
    st.p = (uint32_t)lane * kChunk;
    st.end = st.p + kChunk < n ? st.p + kChunk : n;
    st.nt = 0;
    st.effort = S.effort;
    st.avg16 = n ? (uint32_t)(((uint64_t)S.lit_total << 4) / n) : 0u;
#if defined(__HIP_DEVICE_COMPILE__)   // (the same for every lane: scalar registers)
    st.effort = __builtin_amdgcn_readfirstlane(st.effort);
    st.avg16 = __builtin_amdgcn_readfirstlane(st.avg16);
#endif
    // (the lane's own table already holds the slice before this one: phase_index of the lane before)
    (void)S;
}
// one token; false when the slice is done
#if defined(FQTK_BGZF_PHASE_TIMES) && defined(__HIPCC__)
__device__ unsigned long long g_lz_cycles[10];   // setup, candidate reads + literal costs, extension, token + inserts, steps
#endif
#if defined(FQTK_BGZF_LZ_TIMES) && defined(__HIP_DEVICE_COMPILE__)   // (marks inside the LZ loop: they slow it down a lot)
#define FQTK_LZ_MARK(k) do { const uint64_t now_ = __builtin_readcyclecounter(); lz_acc[k] += now_ - lz_t; lz_t = now_; } while (0)
#else
#define FQTK_LZ_MARK(k) do { } while (0)
#endif
#if defined(FQTK_BGZF_LZ_COUNTS) && defined(FQTK_BGZF_PHASE_TIMES) && defined(__HIP_DEVICE_COMPILE__)   // (events per WAVEFRONT: the first active lane counts; tools/bgzf_phases.sh with LZ_COUNTS=1)
#define FQTK_LZ_COUNT(k) do { if (__lane_id() == (unsigned)__ffsll((unsigned long long)__ballot(1)) - 1u) atomicAdd(&g_lz_cycles[k], 1ull); } while (0)
#else
#define FQTK_LZ_COUNT(k) do { } while (0)
#endif
#if defined(FQTK_BGZF_TRACE) && !defined(__HIP_DEVICE_COMPILE__)   // (CPU study, tools/bgzf_lockstep.py: what every lane did in every step, to stderr)
struct LzTraceAt { int lane = 0, step = 0; };
inline LzTraceAt g_lz_at;
#define FQTK_LZ_TRACE(fmt, ...) fprintf(stderr, "LZ %d %d " fmt "\n", g_lz_at.lane, g_lz_at.step, __VA_ARGS__)
#else
#define FQTK_LZ_TRACE(fmt, ...) do { } while (0)
#endif
#ifndef FQTK_BGZF_SKIP
#define FQTK_BGZF_SKIP 1   // unprobed literals deep inside cheap runs (lz_step); 0: every position is probed (tools/ab_bgzf.sh "" "-DFQTK_BGZF_SKIP=0")
#endif
#ifndef FQTK_BGZF_SKIP_AFTER
#define FQTK_BGZF_SKIP_AFTER 6u   // literals since the lane's last match before any goes unprobed (tools/ab_bgzf.sh, tools/bgzf_ratio.py; GB/s in / output on binned qualities: 6: 75.5 / +0.6 %, 8: 75.0 / +0.4 %, 12: 73.8 / +0.2 %; one more per four literals instead: 74.8 / +0.5 %; none: 69)
#endif
#ifndef FQTK_BGZF_ABL
#define FQTK_BGZF_ABL 0   // developer ablations of the LZ phase (tools/bgzf_phases.sh); 0 in the product
#endif
// A position's look-up comes in three parts, so that a step can have the look-ups of TWO positions in flight (lz_step):
// the table entries of its gram (lz_probe), the candidates they stand for (lz_candidates), and -- only when there is one --
// the comparison of the bytes (lz_first, lz_rest).
struct LzProbe { uint32_t near_slot, mine_near, mine, e_near, e_min, e_max; };
FQTK_HD inline void lz_probe(Shared &S, int lane, uint32_t p, const LzLane &st, uint32_t w, uint32_t w4, LzProbe &pr) {
    const bool cheap = ((st.cheap >> (p - (uint32_t)lane * kChunk)) & 1ull) != 0;   // (what for_grams found: literal costs and p + 8 <= n)
    const uint32_t h = gram_hash(w, w4, cheap);
    pr.near_slot = near_of(h) * kLanes + (uint32_t)lane;
    pr.mine_near = near_entry(p, lane, h);
    pr.mine = region_entry(p, h);
    pr.e_near = S.near_tab[pr.near_slot];
    const uint32_t *e = &S.tminmax[region_slot(p, bucket_of(h))];
    pr.e_min = e[0] & 0xFFFFu;
    pr.e_max = e[-1] >> 16;   // (the word before region 0's is never written: "none")
}
// Three candidates: the lane's own table, the region's earliest entry of the bucket, the previous region's latest -- each
// only if its check bits are the gram's.  Three more were tried and dropped (tools/bgzf_ratio.py): the previous match's
// distance and distance 1 bought nothing -- the lane's own table already holds the position a run or a repeat comes
// from -- and "the same place one record back" bought 0.2-0.7 % (records are rarely equally long).  What each of the
// three is worth on Illumina-style records: without the lane's table the output grows by 0.1-8 %, without the region's
// earliest occurrence by 6-9 %, without the previous region's latest by 1-2 %.
FQTK_HD inline bool lz_candidates(int lane, uint32_t p, const LzProbe &pr, uint32_t (&qpos)[kCands]) {
    uint32_t cand[kCands];
    // (position + 1 of an entry whose check bits are the gram's; "none" entries fail the q < p test below or are 0 here)
    cand[0] = ((pr.e_near ^ pr.mine_near) & 0x7Fu) == 0u ? (pr.e_near >> 7) + (uint32_t)lane * kChunk - kChunk + 1u : 0u;
    cand[1] = ((pr.e_min ^ pr.mine) & 0x1Fu) == 0u ? (p & ~(kRegion - 1u)) + (pr.e_min >> 5) + 1u : 0u;
    cand[2] = (pr.e_max != 0u && ((pr.e_max ^ pr.mine) & 0x1Fu) == 0u) ? ((p - kRegion) & ~(kRegion - 1u)) + (pr.e_max >> 5) + 1u : 0u;
#ifdef FQTK_BGZF_DROP   // developer study (tools/bgzf_ratio.py): candidates switched off by bit mask
    for (int c = 0; c < kCands; ++c) if ((FQTK_BGZF_DROP >> c) & 1) cand[c] = 0;
#endif
    bool any = false;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int c = 0; c < kCands; ++c) {
        const uint32_t q = cand[c] - 1u;                               // 0xFFFFFFFF for "none"
        const bool in_reach = cand[c] != 0u && q < p && p - q <= 32768u;
        qpos[c] = in_reach ? q : p;                                    // p itself: reads fine, never accepted
        any = any || in_reach;
    }
    return any;
}
// The candidates' first four bytes, all read before any is looked at (every dependent LDS round trip is paid in full, so the
// reads go out together); true when one of them is the position's own.  Five check bits let one entry in thirty through
// whose gram is another: among the 64 lanes of a wavefront there is nearly always one, and it is turned away here -- before
// the literal costs, the second four bytes and the arithmetic of lz_rest are paid for by the whole wavefront.
FQTK_HD inline bool lz_first(Shared &S, uint32_t p, uint32_t w, const uint32_t (&qpos)[kCands], uint32_t (&first)[kCands]) {
    bool real = false;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int c = 0; c < kCands; ++c) {
        uint32_t there[1];
        buf_run<1>(S.buf, qpos[c], there);
        first[c] = there[0];
        real = real || (qpos[c] != p && first[c] == w);
    }
    return real && !(FQTK_BGZF_ABL & 32);
}
// How far a match at p may run: past the end of the lane's slice (phase_reach; --compression-level 1-3: it may not, 7 % faster).
FQTK_HD inline uint32_t lz_max_len(uint32_t n, uint32_t p, const LzLane &st) {
#ifdef FQTK_BGZF_NO_REACH   // (study: matches cut at the slice's end, as before phase_reach existed)
    uint32_t maxl = st.end - p;
#else
    uint32_t maxl = st.effort ? n - p : st.end - p;
#endif
    maxl = maxl < 258u ? maxl : 258u;
    if ((FQTK_BGZF_ABL & 8) && maxl > 8u) maxl = 8u;
    return maxl;
}
// Sixteen bytes of a comparison, each side a run of aligned words: how many of them are equal (16: all).
FQTK_HD inline uint32_t lz_round(Shared &S, uint32_t q, uint32_t p) {
    FQTK_LZ_COUNT(3);
    uint32_t a[4], b[4];
    buf_run<4>(S.buf, q, a);
    buf_run<4>(S.buf, p, b);
    const uint32_t x0 = a[0] ^ b[0], x1 = a[1] ^ b[1], x2 = a[2] ^ b[2], x3 = a[3] ^ b[3];
    if (!(x0 | x1 | x2 | x3)) return 16u;
    return x0 ? ctz32(x0) >> 3 : (x1 ? 4 + (ctz32(x1) >> 3) : (x2 ? 8 + (ctz32(x2) >> 3) : 12 + (ctz32(x3) >> 3)));
}
#ifndef FQTK_BGZF_OPEN_ROUNDS
#define FQTK_BGZF_OPEN_ROUNDS 0u   // rounds of sixteen bytes a candidate is compared for inside the step that found it (lz_rest); 0xFFFFu: to its end, as until round 5.  (GB/s in on varied / binned / constant qualities, output on binned ones: 1: 96.0 / 81.5 / 102.2; 0: 99.8 / 91.8 / 107.7, +0.6 %)
#endif
#ifndef FQTK_BGZF_OPEN_STEP
#define FQTK_BGZF_OPEN_STEP 1u     // rounds per step of a match that is still being compared (lz_step)
#endif
// The best of the candidates at position p: length and distance (0 = none pays for itself).  open: the candidate was still equal
// after FQTK_BGZF_OPEN_ROUNDS rounds : it is the match, whatever the others are, and how long it is the lane's
// next steps find out (lz_step).  With no round at all inside the step (the default) that is the first candidate whose eight bytes are the position's.
FQTK_HD inline void lz_rest(Shared &S, uint32_t n, uint32_t p, const LzLane &st, uint32_t w, uint32_t w4, const uint32_t (&qpos)[kCands],
                            const uint32_t (&first)[kCands], uint32_t &mlen, uint32_t &mdist, bool &open) {
    uint32_t msave = 0;
    mlen = mdist = 0;
    open = false;
    const uint32_t maxl = lz_max_len(n, p, st);
    uint32_t second[kCands];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int c = 0; c < kCands; ++c) second[c] = buf_le32(S.buf, qpos[c] + 4u);
    // what the bytes cost as literals: exactly for the first eight, the block's average beyond (every long match pays for
    // itself many times over; the estimate only ranks long candidates among themselves)
    uint32_t lit8[9];
    lit8[0] = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = 0; k < 4; ++k) lit8[k + 1] = lit8[k] + S.lit_cost[(w >> (8 * k)) & 255u];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = 0; k < 4; ++k) lit8[k + 5] = lit8[k + 4] + S.lit_cost[(w4 >> (8 * k)) & 255u];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int c = 0; c < kCands; ++c) {
        const uint32_t q = qpos[c];
        if (q == p || first[c] != w) continue;
        // Bytes 4-7 were fetched with the first four (no loop, no further round trip); the loop below -- sixteen bytes per
        // round, each side a run of aligned words -- is left to the matches that reach eight.
        uint32_t l;
        const uint32_t x4 = second[c] ^ w4;
        if (x4) {
            l = 4u + (ctz32(x4) >> 3);
        } else {
            l = 8;
            uint32_t rounds = 0;
            while (l < maxl) {
                if (rounds == FQTK_BGZF_OPEN_ROUNDS) { open = true; break; }
                const uint32_t same = lz_round(S, q + l, p + l);
                l += same;
                ++rounds;
                if (same != 16u) break;
            }
            if (open) {
                FQTK_LZ_TRACE("C %d %u", c, l);
                mlen = l;
                mdist = p - q;
                return;
            }
        }
        if (l > maxl) l = maxl;
        FQTK_LZ_TRACE("C %d %u", c, l);
        if (l < (uint32_t)kMinMatch) continue;
        const uint32_t lit = l == 4 ? lit8[4] : (l == 5 ? lit8[5] : (l == 6 ? lit8[6] : (l == 7 ? lit8[7] : lit8[8] + (((l - 8) * st.avg16) >> 4))));
        const uint32_t cost = match_cost(l, p - q);
        if (lit > cost && lit - cost > msave) { msave = lit - cost; mlen = l; mdist = p - q; }
    }
}

// A match at p is taken: its token stored (phase_reach counts its symbols).  The last two positions it skips are recent
// history too (a long match is a run or a copied line: its end is where the next one starts from) -- they enter the lane's
// table at the start of the lane's next step, which reads their bytes anyway (tools/bgzf_ratio.py: no positions +1.4 % output
// on binned qualities, the last two +0.05 %, four at each end as much as sixteen and eight; hashed here, at once, they cost
// 7 % of the kernel's time: two more chains of dependent LDS round trips per match).
FQTK_HD inline void lz_take(Shared &S, int lane, uint32_t n, uint32_t *tok, LzLane &st, uint32_t p, uint32_t mlen, uint32_t mdist) {
    (void)S;
    (void)n;
    if (!(FQTK_BGZF_ABL & 2)) tok[st.nt * kLanes + (uint32_t)lane] = match_token(mlen, mdist, p - (uint32_t)lane * kChunk);
    ++st.nt;
    st.pending = (FQTK_BGZF_ABL & 4) ? 0u : 2u;   // (a match is at least four bytes long: three positions skipped)
    st.miss = 0;
    st.p = p + mlen;
}

// One step: the token at st.p and, when that is a literal, the token behind it -- the table look-ups of both positions are
// issued together (in a sequence line nearly every position is a literal, and a wave's step is a chain of dependent LDS
// round trips: two positions per chain instead of one).  The second position's look-up is the one it would have had in a
// step of its own: it sees the first position in the lane's table when both share a slot, and when the first position
// starts a match, the second is among the positions the match enters into the table anyway.  False when the slice is done.
FQTK_HD inline bool lz_step(Shared &S, int lane, uint32_t n, uint32_t *tok, LzLane &st
#if defined(FQTK_BGZF_LZ_TIMES) && defined(__HIP_DEVICE_COMPILE__)   // (marks inside the LZ loop: they slow it down a lot)
                            , uint64_t (&lz_acc)[8], uint64_t &lz_t
#endif
) {
    if (st.p >= st.end) return false;
    const uint32_t p = st.p;
    if (st.open) {
        // A match that was still equal when the step that found it ended: FQTK_BGZF_OPEN_STEP more rounds in every step, until the
        // bytes differ or the match is full.  (Compared to its end inside ONE step, a long match -- the run of a quality line, a
        // line copied whole -- held the other 63 lanes of the wavefront up for sixteen rounds; now they go on with their own steps
        // and pay for two rounds.  tools/bgzf_lockstep.py: the rounds a wavefront runs per step.)
        const uint32_t q = (st.open >> 9) - 1u, maxl = lz_max_len(n, p, st);
        uint32_t l = st.open & 511u;
        bool done = l >= maxl;
        for (uint32_t k = 0; k < FQTK_BGZF_OPEN_STEP && !done; ++k) {
            const uint32_t same = lz_round(S, q + l, p + l);
            l += same;
            done = same != 16u || l >= maxl;
        }
        FQTK_LZ_TRACE("O %u", (l - (st.open & 511u) + 15u) / 16u);
        if (done) {
            FQTK_LZ_COUNT(4);
            st.open = 0;
            lz_take(S, lane, n, tok, st, p, l < maxl ? l : maxl, p - q);
        } else {
            st.open = ((q + 1u) << 9) | l;
        }
        return true;
    }
    if (p + 4 > n) {   // the block's last three bytes: literals
        if (!(FQTK_BGZF_ABL & 1)) FQTK_BGZF_ADD(&S.freq_ll[buf_byte(S.buf, p)], 1u);
        st.p = p + 1;
        return true;
    }
    const bool two = p + 1 < st.end && p + 5 <= n;
    // twelve bytes from two positions back: the grams of p and p + 1, and of the two positions before p when the lane's last
    // match left them to be entered into its table (they lie in the lane's slice: their class is in the mask)
    const uint32_t back = p >= 2u ? 2u : 0u;
    uint32_t r[3];
    buf_run<3>(S.buf, p - back, r);
    const uint32_t wa = bytes_at(r[0], r[1], back), wa4 = bytes_at(r[1], r[2], back);
    const uint32_t wb = bytes_at(r[0], r[1], back + 1u), wb4 = bytes_at(r[1], r[2], back + 1u);
    FQTK_LZ_COUNT(0);
    if (st.pending) {
        FQTK_LZ_COUNT(5);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (uint32_t o = 0; o < 2u; ++o) {   // p - 2, then p - 1: the later one wins a shared slot
            const uint32_t q = p - 2u + o;
            const bool cheap = ((st.cheap >> (q - (uint32_t)lane * kChunk)) & 1ull) != 0;
            const uint32_t h = gram_hash(bytes_at(r[0], r[1], o), bytes_at(r[1], r[2], o), cheap);
            S.near_tab[near_of(h) * kLanes + (uint32_t)lane] = (uint16_t)near_entry(q, lane, h);
        }
        st.pending = 0;
    }
    LzProbe a, b;
    lz_probe(S, lane, p, st, wa, wa4, a);
    if (two) {
        lz_probe(S, lane, p + 1, st, wb, wb4, b);
        if (b.near_slot == a.near_slot) b.e_near = a.mine_near;   // (a run: the position before is the candidate)
    }
    S.near_tab[a.near_slot] = (uint16_t)a.mine_near;
    if (two) S.near_tab[b.near_slot] = (uint16_t)b.mine_near;
    // (A lazy step -- take the literal when the next position holds a longer match that saves more, zlib's levels 4-9 -- was
    //  measured on the CPU run of these phases: 0.0 % / -0.4 % of the output on varied / binned qualities.  Not kept.)
    // Both positions' candidates are fetched and their first four bytes compared; then ONE pass through the match code serves
    // whichever position the lane needs it for -- p when p has a real candidate, else p + 1.  (A wavefront runs that code whenever
    // one of its 64 lanes has a real candidate, and with a pass per position it ran it twice in nearly every step.)  The lane
    // whose candidate at p turns out not to pay for itself while p + 1 has one too takes a second pass: the output is the same
    // as with a pass per position.
    uint32_t qa[kCands], qb[kCands], fa[kCands], fb[kCands], mlen = 0, mdist = 0;
    bool real_a = false, real_b = false;
    if (lz_candidates(lane, p, a, qa)) {
        if (FQTK_BGZF_ABL & 16) st.avg16 += qa[0] + qa[1] + qa[2]; else   // (ablation: the candidates are worked out, nothing is compared)
        real_a = lz_first(S, p, wa, qa, fa);
    }
    if (two && lz_candidates(lane, p + 1, b, qb)) {
        if (FQTK_BGZF_ABL & 16) st.avg16 += qb[0] + qb[1] + qb[2]; else
        real_b = lz_first(S, p + 1, wb, qb, fb);
    }
    uint32_t mp = p;
    bool still_equal = false;
    if (real_a || real_b) {
        FQTK_LZ_COUNT(1);
        uint32_t qs[kCands], fs[kCands];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int c = 0; c < kCands; ++c) { qs[c] = real_a ? qa[c] : qb[c]; fs[c] = real_a ? fa[c] : fb[c]; }
        mp = real_a ? p : p + 1u;
        FQTK_LZ_TRACE("R %d", real_a ? 0 : 1);
        lz_rest(S, n, mp, st, real_a ? wa : wb, real_a ? wa4 : wb4, qs, fs, mlen, mdist, still_equal);
        if (real_a && real_b && !mlen) {
            FQTK_LZ_COUNT(2);
            mp = p + 1u;
            FQTK_LZ_TRACE("R %d", 2);
            lz_rest(S, n, mp, st, wb, wb4, qb, fb, mlen, mdist, still_equal);
        }
    }
    FQTK_LZ_MARK(2);
    if (!(mlen && mp == p) && !(FQTK_BGZF_ABL & 1)) FQTK_BGZF_ADD(&S.freq_ll[wa & 0xFFu], 1u);
    FQTK_LZ_TRACE("S %u %d", mlen, (int)(st.pending != 0));
    if (still_equal) {   // (the lane stays at the match's position: its next steps take the first branch above)
        st.p = mp;
        st.open = ((mp - mdist + 1u) << 9) | mlen;
        return true;
    }
    if (mlen) {
        FQTK_LZ_COUNT(4);
        lz_take(S, lane, n, tok, st, mp, mlen, mdist);
        FQTK_LZ_MARK(6);
        return true;
    }
    st.p = p + 1;
    if (!two) return true;
    if (!(FQTK_BGZF_ABL & 1)) FQTK_BGZF_ADD(&S.freq_ll[wb & 0xFFu], 1u);
    st.p = p + 2;
#if FQTK_BGZF_SKIP
    // Deep inside a run of literals that are cheap as literals -- a sequence line: 150 random bases, no repeat of them pays for its distance code
    // before twelve bases, and one of twelve is a once-in-five-hundred-positions accident -- the positions behind these two are taken as
    // literals UNPROBED: none while the run is young (six literals since the lane's last match), then two, then six per step (their bytes are
    // in the twelve this step has read).  A wave's step count is that of its slowest lanes, and those are the lanes in sequence lines: 32 steps
    // of a 64-byte slice become 10.  A true repeat that starts inside such a run is found a few bases late; the region tables hold every
    // position regardless (phase_index), so nothing becomes unfindable.  The first literals of a run are probed as before.
    // (Skipping in ANY run of literals -- varied qualities are one -- was measured too: output +2.2 % / +2.4 %.  Not taken.)
    st.miss += 2u;
    {
        uint32_t extra = st.miss < FQTK_BGZF_SKIP_AFTER ? 0u : st.miss - FQTK_BGZF_SKIP_AFTER + 2u;
        extra = extra < 6u ? extra : 6u;
        const uint32_t left = st.end - st.p;                              // (st.p <= st.end here)
        extra = extra < left ? extra : left;
        if (st.p + extra + 4u > n) extra = 0;                             // (the block's last bytes take the plain path)
        if (back != 2u) extra = 0;
        if (extra) {                                                      // (left > 0: the shift below stays under 64)
            const uint32_t cheap_bits = (uint32_t)(st.cheap >> (st.p - (uint32_t)lane * kChunk)) & 0x3Fu;
            const uint32_t run = ctz32(~cheap_bits | 0x40u);              // cheap positions in a row from st.p on (<= 6)
            extra = extra < run ? extra : run;
        }
        for (uint32_t k = 0; k < extra; ++k) {
            const uint32_t idx = 4u + k;                                  // byte of r: two positions back + p, p + 1 + k more
            const uint32_t wsel = idx < 8u ? r[1] : r[2];
            if (!(FQTK_BGZF_ABL & 1)) FQTK_BGZF_ADD(&S.freq_ll[(wsel >> (8u * (idx & 3u))) & 0xFFu], 1u);
        }
        st.p += extra;
        st.miss += extra;
    }
#endif
    FQTK_LZ_MARK(7);
    return true;
}
FQTK_HD inline void lz_end(Shared &S, int lane, const LzLane &st) {
    S.ntok[lane] = (uint16_t)st.nt;
    S.span[lane] = (FQTK_BGZF_ABL & 16) ? st.p + (st.avg16 & 0u) : st.p;   // >= the slice's end when the last match ran on (a lane without bytes: its slice's start)
}
FQTK_HD inline void phase_lz(Shared &S, int lane, uint32_t n, uint32_t *tok, uint64_t cheap_mask) {
    LzLane st;
#if defined(FQTK_BGZF_LZ_TIMES) && defined(__HIP_DEVICE_COMPILE__)   // (marks inside the LZ loop: they slow it down a lot)
    uint64_t lz_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, lz_t = __builtin_readcyclecounter();
    lz_begin(S, lane, n, st, cheap_mask);
    { const uint64_t now_ = __builtin_readcyclecounter(); lz_acc[4] += now_ - lz_t; lz_t = now_; }
    uint64_t steps = 0;
    while (lz_step(S, lane, n, tok, st, lz_acc, lz_t)) { ++steps; }
    if ((lane & 63) == 0) {
        for (int k = 0; k < 8; ++k) atomicAdd(&g_lz_cycles[k], (unsigned long long)lz_acc[k]);
        atomicAdd(&g_lz_cycles[8], (unsigned long long)steps);
        atomicAdd(&g_lz_cycles[9], 1ull);
    }
#else
    lz_begin(S, lane, n, st, cheap_mask);
    // (Token-major instead -- an inner loop over literals until the lane has a real candidate, then the match code run by all
    //  lanes that got that far together -- was measured: 40 instead of 55 GB/s.  The lanes' literal runs do not line up, and
    //  every round waits for the longest.)
#if defined(FQTK_BGZF_TRACE) && !defined(__HIP_DEVICE_COMPILE__)
    g_lz_at.lane = lane;
    g_lz_at.step = 0;
    while (lz_step(S, lane, n, tok, st)) { ++g_lz_at.step; }
#else
    while (lz_step(S, lane, n, tok, st)) {}
#endif
#endif
    lz_end(S, lane, st);
}

// P1c (all lanes): who codes which bytes.  Every lane parsed its whole slice from the slice's first byte on, and the last
// match of a slice was allowed to run on into the next ones (cutting every match at a 64-byte boundary cost 2-4 % of the
// output: a header line, a run of equal qualities became two or three matches).  So a lane's bytes begin where the
// matches of the lanes before it end -- the largest of at most five values, a match being at most 258 bytes long -- and
// what its own parse said about the bytes before that point is taken back: literals leave the symbol counts (matches are
// counted here in the first place), a match that straddles the point is cut at its front (shorter than three bytes: its
// rest become literals).  Nothing is parsed again, and no lane waits for another: the ends are those of the first parse, which
// stay valid however a match is cut at its front.
constexpr int kReachLanes = (int)((257u + kChunk - 1u) / kChunk);
FQTK_HD inline void count_match(Shared &S, uint32_t len, uint32_t dist, uint32_t by) {
    uint32_t sym, ne, ev;
    length_symbol(len, sym, ne, ev);
    FQTK_BGZF_ADD(&S.freq_ll[sym], by);
    dist_symbol(dist, sym, ne, ev);
    FQTK_BGZF_ADD(&S.freq_d[sym], by);
}
FQTK_HD inline void phase_reach(Shared &S, int lane, uint32_t *tok, uint32_t *span_out) {
    const uint32_t lo = (uint32_t)lane * kChunk;
    const uint32_t reach = S.span[lane];
    uint32_t start = lo;
    for (int i = 1; i <= kReachLanes; ++i)
        if (lane >= i) { const uint32_t r = S.span[lane - i]; start = r > start ? r : start; }
    const uint32_t nm = S.ntok[lane];
    uint32_t first = 0, cut = 0;
    if (start > lo) {
        const uint32_t stop = start < reach ? start : reach;
        uint32_t m = 0, next = tok[(uint32_t)lane];   // (read whether or not there is a token: a load behind a condition is a branch, and row 0 of the lane's column always exists)
        uint32_t next_pos = nm ? lo + token_off(next) : 0xFFFFFFFFu;
        for (uint32_t p = lo; p < stop;) {
            if (p == next_pos) {
                const uint32_t len = token_len(next), dist = token_dist(next);
                first = m + 1;
                if (p + len > start) {   // (then start < reach: this lane keeps the match's rest)
                    const uint32_t rest = p + len - start;
                    if (rest >= 3u) {
                        tok[m * kLanes + (uint32_t)lane] = match_token(rest, dist, start - lo);
                        count_match(S, rest, dist, 1u);
                        first = m;
                        cut = 1;
                    } else {
                        for (uint32_t k = 0; k < rest; ++k) FQTK_BGZF_ADD(&S.freq_ll[buf_byte(S.buf, start + k)], 1u);
                    }
                }
                p += len;
                ++m;
                next = tok[(m < nm ? m : 0u) * kLanes + (uint32_t)lane];
                next_pos = m < nm ? lo + token_off(next) : 0xFFFFFFFFu;
            } else {
                FQTK_BGZF_ADD(&S.freq_ll[buf_byte(S.buf, p)], 0xFFFFFFFFu);
                ++p;
            }
        }
    }
    // The symbols of the matches that stay are counted here, all lanes side by side, four tokens per round trip -- not where
    // a match is taken: there the other lanes of the wavefront, busy with literals, paid for the arithmetic every time.
    for (uint32_t m0 = first + cut; m0 < nm; m0 += 4u) {
        uint32_t t[4];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (uint32_t k = 0; k < 4u; ++k) t[k] = tok[(m0 + k < nm ? m0 + k : m0) * kLanes + (uint32_t)lane];   // (four unconditional reads in flight; the rows past the last are not looked at)
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (uint32_t k = 0; k < 4u; ++k)
            if (m0 + k < nm) count_match(S, token_len(t[k]), token_dist(t[k]), 1u);
    }
    *span_out = (start < reach ? start : reach) | (reach << 16);   // (nothing left: an empty range)
    S.ntok[lane] = (uint16_t)(nm | (first << 8));
}

// P2a (all lanes): the input copy is no longer needed: the same LDS becomes the (zeroed) output image
// ... and the used literal/length symbols are put in order for the code builder: every lane ranks one symbol
// among all of them (a one-lane insertion sort of ~80 symbols was a fifth of the kernel's time).
FQTK_HD inline void phase_clear_out(Shared &S, int lane) {
    uint32_t *image = out_image(S);
    for (uint32_t i = (uint32_t)lane; i < kOutStride / 4; i += kLanes) image[i] = 0;
    for (uint32_t i = (uint32_t)lane; i < 288; i += kLanes) S.len_ll[i] = 0;
    if (lane < 32) S.len_d[lane] = 0;
    if (lane < kNumCl) S.freq_cl[lane] = 0;
    if (lane == 0) S.freq_ll[256] = 1;   // end of block (read through count_of below: no barrier needed)
    FQTK_BGZF_MID(13);
    auto count_of = [&](int sym) -> uint32_t { return sym == 256 ? 1u : S.freq_ll[sym]; };
#if defined(__HIP_DEVICE_COMPILE__)
    // A symbol's rank = how many used symbols have a smaller (count, symbol).  The wavefronts that rank (five for the literal/length
    // symbols, a sixth for the distance symbols) hold ALL the keys in registers -- key = count << 9 | symbol, unused: all ones, lane l of
    // register k = symbol 64 k + l -- and every lane compares its own key with each of them, handed round by v_readlane: three
    // instructions per comparison and no memory.  (The loop over LDS that the CPU tests run -- 286 dependent rounds of read, compare,
    // add -- was 7 % of the kernel: 14 us of a block's 200.)
    static_assert(kLanes >= 384 && kNumLitLen <= 320 && kNumDist <= 64, "five wavefronts rank the literal/length symbols, the sixth the distance symbols");
    if (lane < 320) {
        uint32_t keys[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int sym = 64 * k + (lane & 63);
            const uint32_t c = sym < kNumLitLen ? count_of(sym) : 0u;
            keys[k] = c ? (c << 9) | (uint32_t)sym : 0xFFFFFFFFu;
        }
        const uint32_t c = lane < kNumLitLen ? count_of(lane) : 0u;
        const uint32_t key = c ? (c << 9) | (uint32_t)lane : 0u;   // (unused: smaller than every key, ranks nothing)
        uint32_t rank = 0;
#pragma unroll
        for (int k = 0; k < 5; ++k)
#pragma unroll   // (a lane number in a register costs v_readlane its wait states after every scalar add)
            for (int l = 0; l < 64; ++l) rank += (uint32_t)__builtin_amdgcn_readlane((int)keys[k], l) < key ? 1u : 0u;
        if (c) {
            S.sorted[rank] = (uint16_t)lane;
            S.weight[rank] = c;   // (the leaves of the code builder, in its order)
            FQTK_BGZF_ADD(&S.m_ll, 1u);
        }
    } else if (lane < 384) {
        const int d = lane - 320;
        const uint32_t c = d < kNumDist ? S.freq_d[d] : 0u;
        const uint32_t mine = c ? (c << 9) | (uint32_t)d : 0xFFFFFFFFu, key = c ? mine : 0u;
        uint32_t rank = 0;
#pragma unroll
        for (int l = 0; l < kNumDist; ++l) rank += (uint32_t)__builtin_amdgcn_readlane((int)mine, l) < key ? 1u : 0u;
        if (c) {
            S.sorted_d[rank] = (uint16_t)d;
            S.weight_d[rank] = c;
            FQTK_BGZF_ADD(&S.m_d, 1u);
        }
    }
#else
    for (int sym = lane; sym < kNumLitLen; sym += kLanes) {
        const uint32_t c = count_of(sym);
        if (!c) continue;
        uint32_t rank = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 8   // (eight reads in flight: a read per round trip made these 286 rounds 14 us of a block's 290)
#endif
        for (int j = 0; j < kNumLitLen; ++j) {
            const uint32_t cj = count_of(j);
            rank += (cj != 0u && (cj < c || (cj == c && j < sym))) ? 1u : 0u;
        }
        S.sorted[rank] = (uint16_t)sym;
        S.weight[rank] = c;   // (the leaves of the code builder, in its order)
        FQTK_BGZF_ADD(&S.m_ll, 1u);
    }
    // the distance symbols likewise, by lanes of another wavefront
    const int d = lane - (kLanes >= 128 ? 64 : 0);
    if (d >= 0 && d < kNumDist) {
        const uint32_t c = S.freq_d[d];
        if (c) {
            uint32_t rank = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 10
#endif
            for (int j = 0; j < kNumDist; ++j) {
                const uint32_t cj = S.freq_d[j];
                rank += (cj != 0u && (cj < c || (cj == c && j < d))) ? 1u : 0u;
            }
            S.sorted_d[rank] = (uint16_t)d;
            S.weight_d[rank] = c;
            FQTK_BGZF_ADD(&S.m_d, 1u);
        }
    }
#endif
}

// P2b (two lanes): the code lengths of both codes -- the serial part: the two-queue tree construction and the depth pass
FQTK_HD inline void phase_code_lengths(Shared &S, int lane) {
    if (lane == 0) {
        S.freq_ll[256] = 1;   // end of block
        huffman_lengths(huff_scratch_ll(S), S.freq_ll, kNumLitLen, 15, S.len_ll, (int)S.m_ll, true);
    } else if (lane == (kLanes >= 128 ? 64 : 1)) {
        huffman_lengths(huff_scratch_d(S), S.freq_d, kNumDist, 15, S.len_d, (int)S.m_d, true);
    }
}
// P2c (all lanes): canonical codes, one symbol per lane; how many lengths of each set are transmitted
FQTK_HD inline void phase_codes(Shared &S, int lane) {
    if (lane < kNumLitLen) {
        S.code_ll[lane] = canonical_code_of(S.len_ll, lane, S.bl_count, 15);
        S.freq_ll[lane] = (uint32_t)S.code_ll[lane] | ((uint32_t)S.len_ll[lane] << 16);   // (the counts are done with: code and length side by side, ONE read per token in phase_emit)
        if (lane >= 257 && S.len_ll[lane]) FQTK_BGZF_MAX(&S.hlit, (uint32_t)lane + 1u);
    }
    const int d = lane - (kLanes >= 512 ? 320 : 0);
    if (d >= 0 && d < kNumDist) {
        S.code_d[d] = canonical_code_of(S.len_d, d, S.bl_count_d, 15);
        if (S.len_d[d]) FQTK_BGZF_MAX(&S.hdist, (uint32_t)d + 1u);
    }
}
// Run-length coding of the hlit + hdist lengths (RFC 1951 3.2.7: 16 = repeat the previous length 3-6 times, 17 = 3-10
// zeros, 18 = 11-138 zeros).  A run of equal lengths is coded by the lane of its first position.
// symbols a run of r lengths of value v is coded with; out != nullptr: they are written (symbol, extra-bit value)
FQTK_HD inline uint32_t cl_code_run(uint32_t v, uint32_t r, uint8_t *sym, uint8_t *extra) {
    uint32_t n = 0, left = r;
    auto put = [&](uint32_t sy, uint32_t ex) { if (sym) { sym[n] = (uint8_t)sy; extra[n] = (uint8_t)ex; } ++n; };
    if (v == 0) {
        while (left >= 11) { const uint32_t t = left < 138 ? left : 138; put(18, t - 11); left -= t; }
        if (left >= 3) { put(17, left - 3); left = 0; }
        while (left-- > 0) put(0, 0);
    } else {
        put(v, 0);
        --left;     // the length itself, then repeats of it
        while (left >= 3) { const uint32_t t = left < 6 ? left : 6; put(16, t - 3); left -= t; }
        while (left-- > 0) put(v, 0);
    }
    return n;
}
// P2d (all lanes): the runs
FQTK_HD inline void phase_cl_runs(Shared &S, int lane) {
    // (hlit and total in registers: read through S at every use they cost a dependent LDS round trip each time -- the stores below may alias them
    //  as far as the compiler knows -- and this phase was 11 us of a block's 200)
    const uint32_t hlit = S.hlit, total = hlit + S.hdist;
    // (one UNCONDITIONAL read behind a selected address: `i < hlit ? S.len_ll[i] : S.len_d[i - hlit]` compiles to a branch around each read -- a
    //  load is not moved across a condition -- and eight of those in a row were sixteen dependent round trips per round of the loop below)
    auto length_at = [&](uint32_t i) -> uint32_t {
        const uint32_t j = i < total ? i : total - 1u;
        const uint8_t *at = j < hlit ? S.len_ll + j : S.len_d + (j - hlit);
        return *at;
    };
    for (uint32_t i = (uint32_t)lane; i < total; i += kLanes) {
        const uint32_t v = length_at(i);
        if (i && length_at(i - 1) == v) { S.run_syms[i] = 0; continue; }
        // the run's length, eight lengths per round trip (a run of unused symbols is up to 138+ long, and a lane that walks it
        // one read at a time holds up the whole workgroup)
        uint32_t r = 1;
        for (;;) {
            uint32_t vals[8];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
            for (uint32_t k = 0; k < 8u; ++k) vals[k] = length_at(i + r + k);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
            for (uint32_t k = 0; k < 8u; ++k) vals[k] = i + r + k < total ? vals[k] : 0xFFFFFFFFu;
            uint32_t k = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
            for (uint32_t q = 0; q < 8u; ++q) k += (k == q && vals[q] == v) ? 1u : 0u;
            r += k;
            if (k < 8u) break;
        }
        S.run_len[i] = (uint16_t)r;
        S.run_syms[i] = (uint16_t)cl_code_run(v, r, nullptr, nullptr);
    }
}
// P2e (all lanes): every run's symbols go to their place in the coded sequence; symbol counts for the code-length code
FQTK_HD inline void phase_cl_emit(Shared &S, int lane) {
    const uint32_t hlit = S.hlit, total = hlit + S.hdist;
    for (uint32_t i = (uint32_t)lane; i < total; i += kLanes) {
        const uint32_t n = S.run_syms[i];
        if (!n) continue;
        uint32_t off = 0;   // symbols of the runs before this one: two counts per read, eight reads in flight
        const uint32_t *pairs = reinterpret_cast<const uint32_t *>(S.run_syms);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 8
#endif
        for (uint32_t j = 0; j < (i >> 1); ++j) { const uint32_t two = pairs[j]; off += (two & 0xFFFFu) + (two >> 16); }
        if (i & 1u) off += S.run_syms[i - 1u];
        cl_code_run(i < hlit ? S.len_ll[i] : S.len_d[i - hlit], S.run_len[i], S.cl_sym + off, S.cl_extra + off);
        for (uint32_t k = 0; k < n; ++k) FQTK_BGZF_ADD(&S.freq_cl[S.cl_sym[off + k]], 1u);
        FQTK_BGZF_ADD(&S.n_cl, n);
    }
}
// P2f (one lane): the code-length code and the fixed part of the block header (BFINAL = 1, BTYPE = 2)
FQTK_HD inline void phase_cl_code(Shared &S) {
    huffman_lengths(huff_scratch_ll(S), S.freq_cl, kNumCl, 7, S.len_cl);
    canonical_codes(S, S.len_cl, kNumCl, 7, S.code_cl);
    const uint8_t order[kNumCl] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    int hclen = kNumCl;
    while (hclen > 4 && S.len_cl[order[hclen - 1]] == 0) --hclen;
    BitWriter w;
    w.start(out_image(S), 0);
    w.put(1, 1);                        // BFINAL
    w.put(2, 2);                        // BTYPE = dynamic Huffman
    w.put(S.hlit - 257u, 5);
    w.put(S.hdist - 1u, 5);
    w.put((uint32_t)(hclen - 4), 4);
    for (int k = 0; k < hclen; ++k) w.put(S.len_cl[order[k]], 3);
    w.finish();
    S.fixed_header_bits = w.bitpos();
}
// P2g (all lanes): the coded lengths, one symbol per lane
FQTK_HD inline void phase_cl_bits(Shared &S, int lane) {
    const uint32_t n = S.n_cl;
    if ((uint32_t)lane >= n) return;
    // what each of the 19 symbols costs, four bits apiece in three words (14 at most: a 7-bit code + 7 extra bits): the sum over the symbols
    // before a lane's own then needs ONE read per symbol -- the symbol -- instead of a second, dependent one for its length
    uint32_t cost[3] = {0, 0, 0};
    for (uint32_t sy = 0; sy < (uint32_t)kNumCl; ++sy)
        cost[sy >> 3] |= (S.len_cl[sy] + (sy == 16 ? 2u : (sy == 17 ? 3u : (sy == 18 ? 7u : 0u)))) << (4u * (sy & 7u));
    for (uint32_t k = (uint32_t)lane; k < n; k += kLanes) {
        uint32_t pos = S.fixed_header_bits;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 8
#endif
        for (uint32_t j = 0; j < k; ++j) {
            const uint32_t sy = S.cl_sym[j];
            pos += ((sy < 8u ? cost[0] : (sy < 16u ? cost[1] : cost[2])) >> (4u * (sy & 7u))) & 15u;
        }
        const uint32_t s = S.cl_sym[k];
        BitWriter w;
        w.start(out_image(S), pos);
        w.put(S.code_cl[s], S.len_cl[s]);
        if (s == 16) w.put(S.cl_extra[k], 2);
        else if (s == 17) w.put(S.cl_extra[k], 3);
        else if (s == 18) w.put(S.cl_extra[k], 7);
        w.finish();
        if (k + 1 == n) S.header_bits = w.bitpos();
    }
}

// A lane's tokens in order: the matches of its slice (global scratch, [m][lane]) and the literal bytes between them.
// visit(literal byte) / visit(len, dist).
template <typename OnLit, typename OnWord, typename OnMatch>
FQTK_HD inline void walk_tokens(Shared &S, int lane, uint32_t n, const uint32_t *tok, OnLit on_lit, OnWord on_word, OnMatch on_match) {
    const uint32_t lo = (uint32_t)lane * kChunk;
    const uint32_t span = S.span[lane], hi = span >> 16;
    const uint32_t nm = S.ntok[lane] & 0xFFu;
    uint32_t m = S.ntok[lane] >> 8, next = tok[(m < nm ? m : 0u) * kLanes + (uint32_t)lane];   // (unconditional reads: see phase_reach)
    uint32_t next_pos = m < nm ? lo + token_off(next) : 0xFFFFFFFFu;
    (void)n;
    uint32_t have = 0xFFFFFFFFu, word = 0;
    // Token-major: the literals up to the lane's next match in an inner loop of their own, then the match.  (With one loop
    // over the positions, a wavefront ran the match's arithmetic in nearly every round -- some lane is at a match -- while
    // most lanes only had a literal to do: the rounds of 64 lanes are as many as the slowest lane's, and each then costs
    // the literal path only.)
    uint32_t p = span & 0xFFFFu;
    for (;;) {
        const uint32_t lim = next_pos < hi ? next_pos : hi;
        for (; p < lim && (p & 3u); ++p) {   // up to a word boundary
            if ((p >> 2) != have) { have = p >> 2; word = S.buf[buf_word(have)]; }
            on_lit((word >> (8 * (p & 3u))) & 0xFFu);
        }
        for (; p + 4u <= lim; p += 4u) {     // whole words: the four table look-ups of a round go out together
            on_word(S.buf[buf_word(p >> 2)]);
        }
        for (; p < lim; ++p) {
            if ((p >> 2) != have) { have = p >> 2; word = S.buf[buf_word(have)]; }
            on_lit((word >> (8 * (p & 3u))) & 0xFFu);
        }
        if (p >= hi) break;
        const uint32_t len = token_len(next);
        on_match(len, token_dist(next));
        p += len;
        ++m;
        next = tok[(m < nm ? m : 0u) * kLanes + (uint32_t)lane];
        next_pos = m < nm ? lo + token_off(next) : 0xFFFFFFFFu;
    }
}

// P3a (all lanes): bits this lane's tokens will take
FQTK_HD inline void phase_count_bits(Shared &S, int lane, uint32_t n, const uint32_t *tok) {
    uint32_t bits = 0;
    walk_tokens(S, lane, n, tok,
                [&](uint32_t lit) { bits += S.len_ll[lit]; },
                [&](uint32_t four) { bits += S.len_ll[four & 0xFFu] + S.len_ll[(four >> 8) & 0xFFu] + S.len_ll[(four >> 16) & 0xFFu] + S.len_ll[four >> 24]; },
                [&](uint32_t len, uint32_t dist) {
                    uint32_t sym, ne, ev;
                    length_symbol(len, sym, ne, ev);
                    bits += S.len_ll[sym] + ne;
                    dist_symbol(dist, sym, ne, ev);
                    bits += S.len_d[sym] + ne;
                });
    S.lane_bits[lane] = bits;
}
// P3b (one lane): exclusive prefix sum, total size, stored-block decision
FQTK_HD inline void phase_offsets(Shared &S, uint32_t n) {
    uint32_t run = S.header_bits;
    for (int l = 0; l < kLanes; ++l) { const uint32_t b = S.lane_bits[l]; S.lane_bits[l] = run; run += b; }
    S.total_bits = run + S.len_ll[256];
    S.stored = ((S.total_bits + 7) >> 3) >= n + 5 ? 1u : 0u;
}

// P4 (all lanes): the tokens' bits; lane 0 also appends the end-of-block code
FQTK_HD inline void phase_emit(Shared &S, int lane, uint32_t n, const uint32_t *tok) {
    if (S.stored) return;
    BitWriter w;
    w.start(out_image(S), S.lane_bits[lane]);
    walk_tokens(S, lane, n, tok,
                [&](uint32_t lit) { const uint32_t e = S.freq_ll[lit]; w.put(e & 0xFFFFu, e >> 16); },
                [&](uint32_t four) {   // four literals: their codes joined two and two (30 bits at most) -- two appends, each with its test for a full word, instead of four
                    const uint32_t e0 = S.freq_ll[four & 0xFFu], e1 = S.freq_ll[(four >> 8) & 0xFFu], e2 = S.freq_ll[(four >> 16) & 0xFFu], e3 = S.freq_ll[four >> 24];
                    w.put((e0 & 0xFFFFu) | ((e1 & 0xFFFFu) << (e0 >> 16)), (e0 >> 16) + (e1 >> 16));
                    w.put((e2 & 0xFFFFu) | ((e3 & 0xFFFFu) << (e2 >> 16)), (e2 >> 16) + (e3 >> 16));
                },
                [&](uint32_t len, uint32_t dist) {
                    uint32_t sym, ne, ev;
                    length_symbol(len, sym, ne, ev);
                    const uint32_t e = S.freq_ll[sym];
                    w.put(e & 0xFFFFu, e >> 16);
                    if (ne) w.put(ev, ne);
                    dist_symbol(dist, sym, ne, ev);
                    w.put(S.code_d[sym], S.len_d[sym]);
                    if (ne) w.put(ev, ne);
                });
    w.finish();
    if (lane == 0) {
        BitWriter e;
        e.start(out_image(S), S.total_bits - S.len_ll[256]);
        e.put(S.code_ll[256], S.len_ll[256]);
        e.finish();
    }
}

// P5 (all lanes): the payload leaves LDS (or, stored: 5 header bytes + the raw input).  Returns the payload size.
FQTK_HD inline uint32_t phase_store(Shared &S, int lane, const uint8_t *in, uint32_t n, uint8_t *out) {
    if (S.stored) {
        if (lane == 0) {
            out[0] = 1;                                   // BFINAL = 1, BTYPE = 00, padding
            out[1] = (uint8_t)(n & 0xFF); out[2] = (uint8_t)(n >> 8);
            out[3] = (uint8_t)(~n & 0xFF); out[4] = (uint8_t)((~n >> 8) & 0xFF);
        }
        for (uint32_t i = (uint32_t)lane; i < n; i += kLanes) out[5 + i] = in[i];
        return n + 5;
    }
    const uint32_t bytes = (S.total_bits + 7) >> 3;
    const uint32_t words = (bytes + 3) >> 2;              // out is 4-byte aligned and kOutStride long
    uint32_t *o = reinterpret_cast<uint32_t *>(out);
    const uint32_t *image = out_image(S);
    for (uint32_t i = (uint32_t)lane; i < words; i += kLanes) o[i] = image[i];
    return bytes;
}

}  // namespace bgzf
}  // namespace fqtk
