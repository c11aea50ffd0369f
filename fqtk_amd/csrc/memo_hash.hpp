// memo_hash.hpp -- key layout, hashes and table constants shared by the memo kernels (device) and the
// memo builders (host).  Plain C++17: compiles under hipcc (host + device) and under g++ (host tools / tests).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define FQTK_HD __host__ __device__
#else
#define FQTK_HD
#endif

namespace fqtk {

constexpr uint32_t kMemoMaxLen = 32;       // 4 bits/base, eight bases per key word: lo = bases 0-7, hi = 8-15, ext = 16-23, ext2 = 24-31
constexpr int kMemoMaxKeyWords = 4;
// Inside a key word, byte k holds base k in its low nibble and base 4+k in its high nibble: the kernel
// packs two 4-base input words with ONE v_lshl_or_b32 (codes_hi << 4 | codes_lo).  The same in every word.
// (The reference's cache has no length limit at all -- barcode_matching.rs:174-181 keys an AHashMap by the raw bytes --
// 32 bases cover 16 + 16 dual indices; longer barcodes take the exhaustive scan.)
FQTK_HD constexpr uint32_t memo_nibble_shift(uint32_t base) {   // bit offset of `base` inside its key word (base >> 3)
    return 4u * (((base & 3u) << 1) | ((base & 7u) >> 2));
}
constexpr uint32_t kMemoEmpty = 0xFFFFFFFFu;
// Placeholder a memo kernel writes for a read it handed to the second pass (index 0xFFFE is no sample: at most
// 65534 of them); the second pass, next in the stream, overwrites it with the read's result.
constexpr uint32_t kMemoDeferred = 0xFFFFFFFEu;

#ifndef FQTK_HOT_BYTES
#define FQTK_HOT_BYTES 65536
#endif
constexpr uint32_t kHotBytes = FQTK_HOT_BYTES;   // LDS budget of the hot table per workgroup

// Key words: 1 (L <= 10: bases 8-9 are folded into the spare top bits of lo's nibbles, see kFoldMul),
// 2 (L <= 16), 3 (L <= 24), 4 (L <= 32).
FQTK_HD constexpr int memo_key_words(uint32_t L) { return L <= 10 ? 1 : (L <= 16 ? 2 : (L <= 24 ? 3 : 4)); }
// Fold of the third word's codes x = code8 | code9 << 8 into bits {3,7,19} / {11,15,27} of lo: the three
// shifted copies of x (<< 3, << 6, << 17) have disjoint supports, so one 24-bit multiply and one AND
// deposit the six bits with no carries; bit 31 stays free for the slot's SPILL flag.
constexpr uint32_t kFoldMul = (1u << 3) | (1u << 6) | (1u << 17);
constexpr uint32_t kFoldMask = 0x08088888u;

// The canonical code of a base is bits 1..2 of its ASCII byte -- A 0x41 -> 0, C 0x43 -> 1, T 0x54 -> 2,
// G 0x47 -> 3 -- and N 0x4E -> 7 with bit 3 included; lower case gives the same codes.  kCodePool maps a
// code back to the upper-case byte it must have come from (0xFF = no such base), which is how the
// kernel proves a byte canonical: ((byte ^ pool[code]) & 0xDF) == 0.
constexpr uint32_t kCodePoolLo = 0x47544341u;   // codes 0..3: 'A' 'C' 'T' 'G'
constexpr uint32_t kCodePoolHi = 0x4EFFFFFFu;   // codes 4..6: none, 7: 'N'
inline uint32_t memo_code_of(char ch) {   // host mirror (the builder only sees A C G T N)
    return ((uint32_t)(uint8_t)ch >> 1) & 7u;
}

// v_perm_b32: byte j of the result is byte sel[j] of the eight bytes {hi : lo} for sel[j] in 0..7, 0x00 for 12, 0xFF above 12
// (8..11, sign bytes, are not used here).  The host's copy is for the CPU tests of the code below.
FQTK_HD inline uint32_t perm_b32(uint32_t hi, uint32_t lo, uint32_t sel) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_perm(hi, lo, sel);
#else
    const uint64_t pool = ((uint64_t)hi << 32) | lo;
    uint32_t out = 0;
    for (int j = 0; j < 4; ++j) {
        const uint32_t s = (sel >> (8 * j)) & 0xFFu;
        const uint32_t byte = s <= 7 ? (uint32_t)(pool >> (8 * s)) & 0xFFu : (s > 12 ? 0xFFu : 0u);
        out |= byte << (8 * j);
    }
    return out;
#endif
}
// One word of a read (four bases) -> its codes c and x = nonzero in every byte that is not one of A C G T N, either case
// (code_mask / byte_mask: 0x07070707 / 0xDFDFDFDF cut down to the word's real bases).
FQTK_HD inline void encode_word(uint32_t w, uint32_t code_mask, uint32_t byte_mask, uint32_t &c, uint32_t &x) {
    c = (w >> 1) & code_mask;
    x = (w ^ perm_b32(kCodePoolHi, kCodePoolLo, c)) & byte_mask;
}

// Plain-A/C/G/T sample tables only (the LDS forms: no sample base covers two bases).  An observed base mismatches an expected
// one iff observed_mask & ~expected_mask != 0 (bitenc.rs:432-459), so against single-base samples EVERY observed code of two
// or more bases -- M R W S Y K V H D B as much as N -- mismatches every sample, and 'U' is 'T' (mod.rs:26-46).  The no-call
// prefilter (barcode_matching.rs:171) cannot tell them apart either: with no N in any sample it passes reads of <=
// max_mismatches no-calls, and a read of k such bases is k mismatches from every sample anyway (None, like the prefilter's
// answer).  So such a read has the memo entry of the read with N's code (7) in those places.  This gives the flagged bytes of
// ONE word (x: nonzero in them; w: the word; c: its codes) those codes where they stand, in the look-up's own pass: '.' has
// code 7 from the encode already; the ten ambiguity letters get 7, U / u keep T's 2, by a 32-entry table of four v_perm_b32
// pools indexed with the letter's low five bits (a selector byte above 12 -- any byte outside 0x40..0x7F -- reads 0xFF).
// Returns 0x80 in every byte that is STILL non-canonical: a byte of no IUPAC meaning (its mask is 0: it MATCHES everything),
// which only the scan resolves.  Rounds 2-5 listed every read with an ambiguity code for a second launch (which spelled the
// codes as N and looked the read up again) or scanned it in place: at 1 % of reads with such a byte that cost 24 % of the
// kernel's rate, at 10 % 49 %; now 5-7 % and 26-29 % (profiles/r06_cliff.jsonl).  Wave-uniform callers: only words in which SOME
// lane has a flagged byte come here (~10 instructions when those are all '.', ~35 otherwise).
FQTK_HD inline uint32_t recode_flagged_bytes(uint32_t w, uint32_t x, uint32_t &c, uint32_t code_mask) {
    const uint32_t flagged = ((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x;                       // bit 7 in every flagged byte
    const uint32_t t = w ^ 0x2E2E2E2Eu;
    const uint32_t rest = flagged & (((t & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | t) & 0x80808080u;   // ... that is not '.'
#if defined(__HIP_DEVICE_COMPILE__)
    if (!__builtin_amdgcn_uicmp(rest, 0u, 33 /* ne */)) return 0u;                        // wave-uniform: only no-call dots here
#else
    if (!rest) return 0u;
#endif
    // code by (byte & 0x1F); 0x80 = no IUPAC meaning:    @ A B C | D E F G     H I J K | L M N O     P Q R S | T U V W     X Y Z [ | \\ ] ^ _
    constexpr uint32_t p0lo = 0x01070080u, p0hi = 0x03808007u, p1lo = 0x07808007u, p1hi = 0x80070780u;
    constexpr uint32_t p2lo = 0x07078080u, p2hi = 0x07070202u, p3lo = 0x80800780u, p3hi = 0x80808080u;
    const uint32_t sel = (w & 0xC7C7C7C7u) ^ 0x40404040u;                                 // letters: byte & 7; anything else: > 12
    const uint32_t t0 = perm_b32(p0hi, p0lo, sel), t1 = perm_b32(p1hi, p1lo, sel);
    const uint32_t t2 = perm_b32(p2hi, p2lo, sel), t3 = perm_b32(p3hi, p3lo, sel);
    // byte j of the pool its bits 3 and 4 name: selector j picks the first word's byte j, j + 4 the second's
    const uint32_t by3 = 0x03020100u | ((w >> 1) & 0x04040404u), by4 = 0x03020100u | ((w >> 2) & 0x04040404u);
    const uint32_t t01 = perm_b32(t1, t0, by3), t23 = perm_b32(t3, t2, by3);
    const uint32_t lut = perm_b32(t23, t01, by4);
    // the table is right for the word's other bytes as well (A C G T N; '.' reads 0xFF: code 7): every code of the word from it
    c = lut & code_mask;
    return lut & rest;
}

// Two-choice (cuckoo) placement: a key lives in slot h1 or slot h2, nowhere else, so a lookup is two
// INDEPENDENT loads issued back to back -- no probe loop, no divergence, one memory round trip.
// 24-bit multiplies only: v_mul_u32_u24 / v_mad_u32_u24 issue at the full VALU rate on gfx950, while
// v_mul_lo_u32 is quarter rate.  The key is cut into <= 24-bit limbs, one multiply each (the limbs of absent key words
// are compile-time zeros in the kernels: their multiplies fold away).
FQTK_HD inline uint32_t mul24(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul24(a, b);
#else
    return (a & 0xFFFFFFu) * (b & 0xFFFFFFu);
#endif
}
FQTK_HD inline uint32_t memo_limb_sum(uint32_t lo, uint32_t hi, uint32_t ext, uint32_t ext2) {
    const uint32_t a = lo;                         // mul24 reads bits 0..23
    const uint32_t b = (lo >> 24) | (hi << 8);
    const uint32_t c = (hi >> 16) | (ext << 16);
    const uint32_t d = ext >> 8;
    const uint32_t e = ext2;
    const uint32_t f = ext2 >> 24;
    return mul24(a, 0x9E3779u) + mul24(b, 0x85EBCBu) + mul24(c, 0xC2B2AFu) + mul24(d, 0xA54FF5u) + mul24(e, 0xB5297Au) + mul24(f, 0x68E31Du);
}
// The same limbs under a second set of odd multipliers: the two slots of a key come from two INDEPENDENT 32-bit sums.
// (Until round 4 the second slot was a function of the first sum alone: keys whose sums collide then share BOTH slots,
// the sums being linear in the limbs such collisions are systematic among millions of one- and two-substitution
// neighbours, and three of them made every placement fail whatever the table's size -- 384 x 24 bases with two
// mismatches, 1.7 M keys.)
FQTK_HD inline uint32_t memo_limb_sum2(uint32_t lo, uint32_t hi, uint32_t ext, uint32_t ext2) {
    const uint32_t a = lo;
    const uint32_t b = (lo >> 24) | (hi << 8);
    const uint32_t c = (hi >> 16) | (ext << 16);
    const uint32_t d = ext >> 8;
    const uint32_t e = ext2;
    const uint32_t f = ext2 >> 24;
    return mul24(a, 0xD6E8FFu) + mul24(b, 0x2C1B3Du) + mul24(c, 0x7F4A7Du) + mul24(d, 0x51ED27u) + mul24(e, 0xC4CEB9u) + mul24(f, 0x3243F7u);
}
// (the whole 32 bits of the two hashes: a slot is their low bits; the LDS presence filter takes other bits of both)
FQTK_HD inline uint32_t memo_hash1_full(uint32_t lo, uint32_t hi, uint32_t ext, uint32_t ext2) {
    uint32_t h = memo_limb_sum(lo, hi, ext, ext2);
    h ^= h >> 15;
    h = mul24(h, 0x2C1B3Du) + (h >> 9);
    h ^= h >> 13;
    return h;
}
FQTK_HD inline uint32_t memo_hash2_full(uint32_t lo, uint32_t hi, uint32_t ext, uint32_t ext2) {
    uint32_t g = memo_limb_sum2(lo, hi, ext, ext2);
    g ^= g >> 14;
    g = mul24(g, 0x9E3779u) + (g >> 10);
    g ^= g >> 12;
    return g;
}
FQTK_HD inline uint32_t memo_slot1(uint32_t lo, uint32_t hi, uint32_t ext, uint32_t ext2, uint32_t mask) { return memo_hash1_full(lo, hi, ext, ext2) & mask; }
// (the kernels work this one out only in the rare wave that needs a second probe -- or for every read, when the table has a presence filter)
FQTK_HD inline uint32_t memo_slot2(uint32_t lo, uint32_t hi, uint32_t ext, uint32_t ext2, uint32_t mask) { return memo_hash2_full(lo, hi, ext, ext2) & mask; }

// ---- presence filter of the hash-table form (round 5) ------------------------------------------------------------------------
// Every key of the memo sets two bits of a bit array that the kernel keeps in LDS next to the hot table: bit filter_pos1(h1) and bit
// filter_pos2(h2) of the key's two full hashes.  A read whose two bits are not both set is in no slot of the table -- its result is
// None without a look at the table (no false negatives: the builder and the kernel compute the same two positions from the same
// functions).  What it is for: a gather into the table costs one 64-byte request per lane, and the reads that are NOT in the memo --
// one read in ten of a real run is no sample's barcode -- used to pay it only to learn that.  log2_bits <= 19 (64 KiB).
FQTK_HD inline uint32_t memo_filter_pos1(uint32_t h1_full, uint32_t log2_bits) { return (h1_full >> 11) & ((1u << log2_bits) - 1u); }
FQTK_HD inline uint32_t memo_filter_pos2(uint32_t h2_full, uint32_t log2_bits) { return (mul24(h2_full >> 7, 0x5BD1E9u) >> 4) & ((1u << log2_bits) - 1u); }
FQTK_HD inline void memo_hash2(uint32_t lo, uint32_t hi, uint32_t ext, uint32_t ext2, uint32_t mask,
                               uint32_t &s1, uint32_t &s2) {
    s1 = memo_slot1(lo, hi, ext, ext2, mask);
    s2 = memo_slot2(lo, hi, ext, ext2, mask);
}

// ---- direct-indexed memo for short barcodes (table form, L <= kDirectMaxLen) ----------------------------
// A read of L <= 10 plain A/C/G/T bases IS a 2L-bit number: the memo of those reads is a flat array in
// HBM (L2-resident: 4^10 x 2 B = 2 MiB) indexed by the read itself -- no hash, no key compare, no second
// probe, and "absent" is just the stored None.  Reads that carry a no-call keep using the cuckoo table
// (which then holds only the N-containing entries).
// Index = the 2-bit codes of the bases (A0 C1 T2 G3, memo_code_of & 3), taken straight from the 4-bit key
// BEFORE the fold: lo_unf = bases 0-7 in memo_nibble_shift order, c2 = codes of bases 8, 9 in bytes 0, 1.
// Bit layout (what three VALU ops give): base0 0-1, base2 2-3, base4 4-5, base6 6-7, base1 8-9, base3 10-11,
// base5 12-13, base7 14-15, base8 16-17, base9 18-19.
constexpr uint32_t kDirectMaxLen = 10;
FQTK_HD inline uint32_t memo_direct_index(uint32_t lo_unf, uint32_t c2) {
    const uint32_t x = lo_unf & 0x33333333u;
    const uint32_t i16 = (x & 0xFFFFu) | (x >> 14);            // nibbles 4-7 drop into the gaps of nibbles 0-3
    return (mul24(c2 & 0x0303u, 0x10400u) & 0xF0000u) | i16;   // codes 8, 9 -> bits 16-19 (one 24-bit multiply)
}
FQTK_HD inline uint32_t memo_nocall_bits(uint32_t lo_unf, uint32_t c2) {   // != 0 <=> some base is N (code 7)
    return (lo_unf & 0x44444444u) | (c2 & 0x0404u);
}
FQTK_HD constexpr uint32_t memo_direct_entries(uint32_t L) { return L <= 8 ? (1u << 16) : (L == 9 ? (1u << 18) : (1u << 20)); }
// 2-byte entries [idx : ib | best : bb | next : rest] when idx, best and next of every entry fit 16 bits with
// the all-ones pattern left over (the builder picks ib and bb; cfg 5: 11 + 1 + 4), else the 4-byte result
// word itself; None = all ones either way.
FQTK_HD inline uint32_t memo_direct_unpack16(uint32_t e, uint32_t ib, uint32_t bb) {
    const uint32_t idx = e & ((1u << ib) - 1u), best = (e >> ib) & ((1u << bb) - 1u), next = e >> (ib + bb);
    return e == 0xFFFFu ? kMemoEmpty : (idx | (best << 16) | (next << 24));
}
// The kernel's form: the layout as three wave-uniform words (mask of idx, shift | width of best, shift of next) and NO
// branch -- everything is computed and None is one compare + select at the end (hipcc turned the ?: above into a
// lane-masked branch with its scalar operands reloaded from spilled SGPRs inside).
struct DirectLayout { uint32_t idx_mask, ib, bb_mask, nsh; };
FQTK_HD inline DirectLayout memo_direct_layout(uint32_t ib, uint32_t bb) { return DirectLayout{(1u << ib) - 1u, ib, (1u << bb) - 1u, ib + bb}; }
FQTK_HD inline uint32_t memo_direct_unpack16(uint32_t e, const DirectLayout &d) {
    const uint32_t lowhalf = (e & d.idx_mask) | (((e >> d.ib) & d.bb_mask) << 16);
    const uint32_t v = ((e >> d.nsh) << 24) | lowhalf;
    return e == 0xFFFFu ? kMemoEmpty : v;
}
FQTK_HD inline uint32_t memo_direct_pack16(uint32_t val, uint32_t ib, uint32_t bb) {   // val = idx | best << 16 | next << 24
    return (val & 0xFFFFu) | (((val >> 16) & 0xFFu) << ib) | ((val >> 24) << (ib + bb));
}
// LDS cache of the direct table's EXACT-match entries (the bulk of real reads): two-choice cuckoo over
// buckets of two 4-byte slots, so one ds_read_b64 fetches a bucket.  A slot is [which : 1 | tag | val16]:
// the bucket index is a slice of the read's index and the tag is the REST of it, so a tag match is an EXACT
// key match without storing the key.  Choice 0 buckets by the low B bits of the index (tag = the bits above),
// choice 1 by its top B bits (tag = the bits below): two shifts and two masks per read, no rotation.
// nbits = bits of the index (memo_direct_index_bits(L)); 6 <= B <= nbits, nbits - B <= 14 (the tag sits below `which`).
FQTK_HD constexpr uint32_t memo_direct_index_bits(uint32_t L) { return L <= 8 ? 16u : (L == 9 ? 18u : 20u); }
FQTK_HD inline uint32_t memo_hot2_bucket(uint32_t didx, uint32_t B, uint32_t nbits, uint32_t which) {
    return which ? didx >> (nbits - B) : didx & ((1u << B) - 1u);
}
FQTK_HD inline uint32_t memo_hot2_want(uint32_t didx, uint32_t B, uint32_t nbits, uint32_t which) {
    return which ? ((didx & ((1u << (nbits - B)) - 1u)) | 0x8000u) : didx >> B;   // upper half-word of a matching slot
}
constexpr uint32_t kHot2MaxBucketBits = 13;           // 8192 buckets x 8 B = 64 KiB

// Direct form, the entries WITH a no-call (direct_memo_plan.hpp: buckets of two slots), keyed by the folded 4-bit key
// `lo` (which says WHERE the no-calls are; the 2-bit index alone does not -- an N reads as G there, and a string with k
// G's would put k keys on one index).  Four VALU operations for the first bucket; the second choice is only computed by
// the rare wave that needs it.
// `shift` = 32 - log2(buckets) (>= 6 bucket bits): the TOP bits of two 24-bit multiplies over the low and the high 24 bits
// of the key -- every base reaches them, whatever the table's size.
FQTK_HD inline uint32_t memo_nbucket1(uint32_t lo, uint32_t shift) { return (mul24(lo, 0x9E3779u) + mul24(lo >> 8, 0x85EBCBu)) >> shift; }
FQTK_HD inline uint32_t memo_nbucket2(uint32_t lo, uint32_t shift) { return (mul24(lo, 0xC2B2AFu) + mul24(lo >> 8, 0xA54FF5u) + 0x7F4A7C15u) >> shift; }
FQTK_HD inline uint32_t memo_nbucket_shift(uint32_t mask) {   // mask = buckets - 1
    uint32_t b = 0;
    while (mask >> b) ++b;
    return 32u - b;
}

// ---- LDS-resident compact memo (lds_memo_kernels.hip.h, lds_memo_plan.hpp) ---------------------------
constexpr uint32_t kLdsMemoMaxBytes = 160u * 1024u;   // LDS per CU = per workgroup limit on gfx950
// Entry dword.  The result bits sit where fqtk_match_t wants them, so "entry & res_mask" IS the result:
//   [0, IB) idx | [IB, 16) fingerprint | 16 best | 17-19 xnib | 20-23 pos (nibble index; bit 23 = hi word)
//   | 24-28 next | 29 pos bit 4 (ext / ext2 words; keys of 3 and 4 words only, else fingerprint) | 30-31 fingerprint
constexpr uint32_t kLdsMaxIdxBits = 12;               // leaves >= 6 fingerprint bits
FQTK_HD constexpr uint32_t lds_fp_mask(uint32_t idx_bits, int kw) {
    return (kw >= 3 ? 0xC0000000u : 0xE0000000u) | (0xFFFFu & ~((1u << idx_bits) - 1u));
}
FQTK_HD constexpr uint32_t lds_res_mask(uint32_t idx_bits) { return 0x1F010000u | ((1u << idx_bits) - 1u); }
FQTK_HD constexpr uint32_t lds_entry_fields(uint32_t idx, uint32_t best, uint32_t next, uint32_t xnib, uint32_t pos) {
    return idx | (best << 16) | (xnib << 17) | ((pos & 15u) << 20) | (next << 24) | ((pos >> 4) << 29);
}
FQTK_HD constexpr uint32_t lds_entry_pos(uint32_t e, int kw) { return ((e >> 20) & 15u) | (kw >= 3 ? ((e >> 29) & 1u) << 4 : 0u); }

// Slot hashes + fingerprint source of the LDS table: two mixed words.  The three slots are bits 2..
// of h, of g and of h >> 15 (so each masked value IS an LDS byte address; with at most 2^15 slots the
// first and third use disjoint bits of h).  The fingerprint must be independent of the slot bits --
// two keys that share a slot share those bits -- so it is taken from g[16..31], which no slot index
// reads (slot 2 uses g[2..16]): one v_perm_b32 moves g's upper half under the entry's fingerprint
// fields (bits IB..15 <- g[16+IB..31], bits 29..31 <- g[21..23]).
FQTK_HD inline void memo_hash3(uint32_t lo, uint32_t hi, uint32_t ext, uint32_t ext2, uint32_t salt,
                               uint32_t &h1, uint32_t &h2, uint32_t &h3, uint32_t &fp_src) {
    // one multiply-add per 24-bit limb, one xor-shift to bring the well-mixed top bits down, one more
    // multiply for the second word: 3-choice cuckoo at load 0.76 does not need more (the planner
    // verifies every placement and falls back to the table form if a build ever failed)
    uint32_t h = memo_limb_sum(lo, hi, ext, ext2) + salt;
    h ^= h >> 15;
    uint32_t g = mul24(h >> 7, 0xD6E8FFu) + h;
    g ^= g >> 14;
    h1 = h; h2 = g; h3 = h >> 15;   // (power-of-two tables; lds_slot_any below reads h and g directly)
#if defined(__HIP_DEVICE_COMPILE__)
    fp_src = __builtin_amdgcn_perm(g, g, 0x02000302u);   // bytes: g.b2, g.b3, g.b0, g.b2
#else
    fp_src = (g >> 16) | ((g & 0xFFu) << 16) | (((g >> 16) & 0xFFu) << 24);
#endif
}

// Slot byte addresses.  Power-of-two tables mask the hash words (memo_hash3's h1/h2/h3 & slot_mask_b).
// Any other slot count n (< 2^16; used when the next power of two would not fit LDS) maps a 16-bit
// hash piece x to floor(x * n / 2^16) with one 24-bit multiply: pieces h[0..15], g[0..15], h[16..31].
FQTK_HD inline uint32_t lds_slot_any(uint32_t x16, uint32_t n_slots) { return (mul24(x16, n_slots) >> 14) & ~3u; }
FQTK_HD inline void lds_slots(bool pow2, uint32_t h, uint32_t g, uint32_t slot_mask_b, uint32_t n_slots,
                              uint32_t &a1, uint32_t &a2, uint32_t &a3) {
    if (pow2) { a1 = h & slot_mask_b; a2 = g & slot_mask_b; a3 = (h >> 15) & slot_mask_b; }
    else { a1 = lds_slot_any(h & 0xFFFFu, n_slots); a2 = lds_slot_any(g & 0xFFFFu, n_slots); a3 = lds_slot_any(h >> 16, n_slots); }
}

// ---- LDS form with a minimal perfect hash: ONE probe, three-byte entries (round 6) ------------------------------------------------
// For plain-A/C/G/T tables whose entries do not fit one CU's LDS as four-byte cuckoo slots at a workable load -- 384 samples x 24 bases
// (a 12+12 dual index): 37 248 entries, 175 KB at load 0.85.  A hash-and-displace function built on the host (lds_memo_plan.hpp) sends
// every key of the memo to a slot of its own in a table 3 % larger than the key set, so an entry needs no fingerprint: the read's one
// candidate is verified against its sample's key exactly like a cuckoo candidate (a read that is NOT in the memo lands on some other
// key's entry, or an empty one, and fails that check).  Entry, 24 bits: [0, 9) idx | 9 best | 10-14 next | 15-17 xnib | 18-22 pos
// (the nibble that differs from the sample's, and where); stored as a 16-bit and an 8-bit array so that no read is unaligned.
constexpr uint32_t kMphIdxBits = 9;                    // S + 1 <= 512
constexpr uint32_t kMphMaxDisplacement = 65535;        // a bucket's displacement is a 16-bit word
FQTK_HD constexpr uint32_t mph_entry_fields(uint32_t idx, uint32_t best, uint32_t next, uint32_t xnib, uint32_t pos) {
    return idx | (best << 9) | (next << 10) | (xnib << 15) | (pos << 18);
}
FQTK_HD constexpr uint32_t mph_entry_result(uint32_t e) {   // the fqtk_match_t word: idx | best << 16 | next << 24
    return (e & 511u) | (((e >> 9) & 1u) << 16) | (((e >> 10) & 31u) << 24);
}
// two independent 32-bit hashes of the key (the limb sums of the hash-table form, a salt in each)
FQTK_HD inline void mph_hashes(uint32_t k0, uint32_t k1, uint32_t k2, uint32_t k3, uint32_t salt, uint32_t &a, uint32_t &b) {
    a = memo_limb_sum(k0, k1, k2, k3) + salt;
    a ^= a >> 15;
    a = mul24(a, 0x2C1B3Du) + (a >> 9);
    a ^= a >> 13;
    b = memo_limb_sum2(k0, k1, k2, k3) + (salt ^ 0xA54FF53Au);
    b ^= b >> 14;
    b = mul24(b, 0x9E3779u) + (b >> 10);
    b ^= b >> 12;
}
FQTK_HD inline uint32_t mph_mulhi(uint32_t x, uint32_t y) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umulhi(x, y);
#else
    return (uint32_t)(((uint64_t)x * y) >> 32);
#endif
}
// bucket = a & bucket_mask; the bucket's displacement d (16 bits) moves all its keys together: slot = floor((b + d * step) * n_slots / 2^32)
FQTK_HD inline uint32_t mph_slot(uint32_t a, uint32_t b, uint32_t d, uint32_t n_slots) {
    return mph_mulhi(b + mul24(d, (a >> 9) | 1u), n_slots);
}

}  // namespace fqtk
