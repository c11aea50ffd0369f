// bgzf_inflate.hpp -- DEFLATE decoder for BGZF input members, written for ONE 64-lane wavefront per member.
//
// SURVEY.md section 8(f) row 3 (input side).  The reference opens every input through a gz-aware reader
// (/root/reference/src/bin/commands/demux.rs:844-849: fgoxide `Io` -> flate2) and inflates on the host; real FASTQs are
// compressed, so the host's inflate rate bounds the whole run while PCIe carries four to five times the bytes it needs to.
// BGZF members (RFC 1952 members of <= 64 KiB with the 'BC' extra field; what bgzip, htslib and fqtk itself write) are
// independent DEFLATE streams: one wavefront takes one member, thousands are in flight.
//
// A DEFLATE stream is serial -- where a code starts is known only when the one before it has been decoded -- so one lane
// per stream would leave 63 idle and a wave would crawl through its member at one dependent LDS look-up per symbol.  The wave
// decodes SPECULATIVELY instead: lane i decodes the whole token (literal, or length + distance with their extra bits, or
// end-of-block) that WOULD start at bit `base + i`, all 64 at once, one table look-up each; the true tokens are then the
// chain 0 -> 0 + bits[0] -> ... walked with v_readlane on the scalar unit (~10 cycles a step, no memory), which also
// gives the position of the next window.  The lanes on the chain hold real tokens: a wave prefix sum of their output
// lengths places them, literal lanes store their byte, matches are copied by all 64 lanes together.  Codes longer than the
// table's index bits (rare symbols) are resolved canonically (puff-style: lengths' limits compared side by side, no
// second-level tables to build) only when the chain reaches one.
//
// The output goes straight to HBM: a match reads what earlier lanes wrote, which the CU's L1 / the L2 serve; a
// workgroup-scope fence is placed only where a match reaches into bytes whose stores may still be in flight.
//
// Parity: zlib inflates the same members to the same bytes (tests/test_bgzf_inflate.py: every level and strategy, stored /
// fixed / dynamic blocks, corrupted streams, CRC and ISIZE mismatches).  Written from RFC 1951 / RFC 1952.
//
// One source for two builds: the device (fqtk_bgzf.hip instantiates inflate_member<DeviceWave>) and the CPU test-suite,
// where host/wave_emu.hpp runs the SAME function as 64 cooperatively scheduled fibers whose cross-lane operations
// (ballot, readlane, scan, barrier) are real exchanges -- slow (a few MB/s) but the decoder's logic, not a restatement of it.
#pragma once
#include <stddef.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define FQTK_HD __host__ __device__
#else
#ifndef FQTK_HD
#define FQTK_HD
#endif
#endif

#if defined(__clang__)
#define FQTK_UNROLL _Pragma("unroll")
#else
#define FQTK_UNROLL
#endif

namespace fqtk {
namespace inflate {

#ifndef FQTK_INFLATE_LIT_BITS
#define FQTK_INFLATE_LIT_BITS 9    // index bits of the literal/length table: 2 KiB of LDS; longer codes are resolved canonically when the chain meets one.
                                   // (10 bits, 4 KiB, until round 6: 89 GB/s of text against 94 with 9 bits, 103 with 9 bits and six wavefronts a SIMD --
                                   //  24 576 members, zlib -6, tools/ab_inflate.sh "-DFQTK_INFLATE_LIT_BITS=10"; profiles/r06_ab_inflate.txt)
#endif
constexpr int kLitBits = FQTK_INFLATE_LIT_BITS, kDistBits = 8;
static_assert(kLitBits >= 8 && kLitBits <= 11, "the code-length code's 128-entry table and the member check's 256-entry CRC table share the literal table's room");
#ifndef FQTK_INFLATE_SETS
#define FQTK_INFLATE_SETS 2   // bit positions decoded per lane and window (1 or 2): tools/ab_inflate.sh "-DFQTK_INFLATE_SETS=1"
#endif
constexpr uint32_t kSets = FQTK_INFLATE_SETS;
static_assert(kSets == 1 || kSets == 2, "one or two sets of 64 bit positions");
constexpr uint32_t kRingWords = 256;   // 1 KiB of the compressed stream in LDS (dword d of the stream at ring[d & 255])

// status of a member (written to status[member]; 0 = fine)
enum : uint32_t {
    kOk = 0,
    kErrBlockType = 1,      // reserved block type 3
    kErrStoredLen = 2,      // LEN != ~NLEN
    kErrCodeLengths = 3,    // bad code-length code / repeat without a previous length / repeat past the end / no EOB code
    kErrOverSubscribed = 4, // a Huffman code that is over-subscribed, or incomplete where RFC 1951 / zlib do not allow it
    kErrBadCode = 5,        // a bit pattern that is no code of the block's Huffman code (or symbols 286, 287, 30, 31)
    kErrDistance = 6,       // distance reaches before the start of the member's output
    kErrOutput = 7,         // more output than ISIZE says
    kErrTruncated = 8,      // the stream runs past the member's payload
    kErrLength = 9,         // fewer bytes than ISIZE says
    kErrCrc = 10,           // CRC-32 of the output differs from the trailer's (set by the check kernel)
    kErrHeader = 11,        // not a gzip member / unsupported flags (set by whoever parses the header)
};

FQTK_HD inline uint32_t brev32(uint32_t x) {
#if defined(__clang__)
    return __builtin_bitreverse32(x);
#else
    x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
    x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
    x = ((x >> 4) & 0x0F0F0F0Fu) | ((x & 0x0F0F0F0Fu) << 4);
    return __builtin_bswap32(x);
#endif
}

// table entry: bits 0-3 code length (0: no such code), 4-7 extra bits, 8-9 kind, 16-31 literal byte / base value
enum : uint32_t { kLit = 0, kLen = 1, kEob = 2, kLong = 3 };
FQTK_HD inline uint32_t make_entry(uint32_t codelen, uint32_t extra, uint32_t kind, uint32_t value) {
    return codelen | (extra << 4) | (kind << 8) | (value << 16);
}
// literal/length symbol -> entry (RFC 1951 3.2.5, computed: no constant tables on the device); 0 for 286, 287
FQTK_HD inline uint32_t litlen_entry(uint32_t sym, uint32_t codelen) {
    if (sym < 256u) return make_entry(codelen, 0, kLit, sym);
    if (sym == 256u) return make_entry(codelen, 0, kEob, 0);
    if (sym >= 286u) return 0u;
    const uint32_t i = sym - 257u;
    if (i < 8u) return make_entry(codelen, 0, kLen, 3u + i);
    if (i == 28u) return make_entry(codelen, 0, kLen, 258u);
    const uint32_t e = (i >> 2) - 1u;
    return make_entry(codelen, e, kLen, 3u + ((4u + (i & 3u)) << e));
}
FQTK_HD inline uint32_t dist_entry(uint32_t sym, uint32_t codelen) {
    if (sym >= 30u) return 0u;
    if (sym < 4u) return make_entry(codelen, 0, kLit, 1u + sym);
    const uint32_t e = (sym >> 1) - 1u;
    return make_entry(codelen, e, kLit, 1u + ((2u + (sym & 1u)) << e));
}

// A canonical Huffman code (RFC 1951 3.2.2) the way the slow path and the table builder read it.  Code words are taken
// left-aligned in 15 bits (first bit of the code = bit 14): codes of increasing length are increasing numbers, so the
// length of the code in front of `rev` is 1 + the number of lengths whose (left-aligned, exclusive) limit rev has reached.
struct Canon {
    uint32_t limit[16];    // [l]: (first code of length l + number of codes of length l) << (15 - l); limit[0] = 0
    uint32_t first[16];    // [l]: first code of length l, left-aligned
    uint16_t offs[16];     // [l]: index in perm of the first symbol of length l
    uint16_t cnt[16];
};

// Everything a member's wavefront shares (LDS on the device: 7.6 KiB -- how many wavefronts a CU holds is this number, and the decoder
// lives on wavefronts in flight: every window is one long chain of dependent look-ups, a scalar walk and a round trip to memory).
// The code-length code's table (a block header's business) lives in the literal table's room, which is built after it is done with;
// so does the CRC table of the member check (fqtk_inflate.hip), which runs when the decoder is.
struct Shared {
    uint32_t ring[kRingWords];
    uint32_t lit[1u << kLitBits];
    uint32_t dist[1u << kDistBits];
    Canon cl, cd;              // literal/length code, distance code
    uint16_t perm_l[288 + 32]; // symbols ordered by (code length, symbol)
    uint16_t perm_d[32];
    uint8_t lens[320 + 8];     // the block's code lengths: hlit literal/length ones, then hdist distance ones
    uint32_t own[64];          // per byte of the 64 being written: lane + 1 of a token that starts there
};

struct MemberArgs {
    const uint32_t *in_words;  // 4-byte aligned address at or before the first byte of the DEFLATE payload
    uint32_t first_bit;        // bit offset of the payload from in_words (0, 8, 16 or 24)
    uint32_t payload_bits;     // length of the payload in bits (the trailer follows)
    uint32_t readable_words;   // whole dwords that may be read from in_words (the caller's buffer ends behind them; past it reads as 0)
    uint32_t tail_bytes;       // 0..3 bytes of the buffer behind those dwords (read one by one)
    uint8_t *out;              // isize bytes
    uint32_t isize;            // ISIZE of the member's trailer (stream mode: room in out_sym, in symbols)
    // stream mode (inflate_member<W, true>: a stretch of ONE serial gzip stream, started at a block boundary somewhere inside it)
    uint16_t *out_sym;         // symbols: < 256 a byte; 256 + j: byte j of the 32 KiB in front of the chunk, which it does not know
    uint32_t stop_bit;         // the chunk ends at the first block boundary at or behind this bit (or with the final block)
};
// What a chunk of a stream came to (stream mode).
// n_blocks: whole blocks decoded; n_sym / end_bit / final_block describe the LAST block boundary reached -- also when the chunk then failed
// (no room for more symbols, no more input): what came before that boundary is good, and a caller may take it.
// flags: kStreamHighLiterals -- some block up to that boundary gave a code to a literal >= 128 (text such as FASTQ never does, and the
// block-start search can then refuse candidates that do: fqtk_inflate.hip).
struct StreamEnd { uint32_t n_sym, end_bit, final_block, n_blocks, flags; };
enum : uint32_t { kStreamHighLiterals = 1 };
constexpr uint32_t kWindow = 32768;

// ---- the wave -------------------------------------------------------------------------------------------------------
// W provides: lane() 0..63; ballot(bool) -> uint64; readlane(uint32 v, uint32 l) (l the same in all lanes);
// uniform(uint32 v) (a value known to be the same in all lanes: the device keeps it in a scalar register); set_bit64(uint64 &m, bit) m |= 1 << bit, both wave-uniform;
// scan_incl(uint32 v) inclusive prefix sum over the lanes; scan_max_incl(uint32 v) inclusive prefix maximum;
// shuffle(uint32 v, uint32 l) lane l's v (l may differ between lanes); rcp(float) ~ 1 / x; barrier() (LDS written before is visible to
// all lanes after); fence_global() (global stores issued by any lane before are visible to the loads of all lanes after);
// atomic_inc_lds(uint16*/uint32*).

template <class W>
FQTK_HD inline uint32_t peek32(W &w, const Shared &S, uint32_t bit) {   // 32 bits of the stream from `bit`
    const uint32_t d = bit >> 5, s = bit & 31u;
    const uint64_t v = (uint64_t)S.ring[d & (kRingWords - 1u)] | ((uint64_t)S.ring[(d + 1u) & (kRingWords - 1u)] << 32);
    return (uint32_t)(v >> s);
}

// The ring: dwords [.., filled) of the stream are in LDS, every lane holds dword filled + lane in `pref` (requested one
// refill ahead, so its latency hides behind ~250 bytes of decoding).
template <class W>
struct Ring {
    uint32_t filled;   // uniform
    uint32_t pref;     // per lane
    FQTK_HD inline uint32_t load(W &w, const MemberArgs &a, uint32_t d) const {
        if (d < a.readable_words) return a.in_words[d];
        uint32_t v = 0;
        if (d == a.readable_words) {   // the buffer's last, partial dword
            const uint8_t *p = reinterpret_cast<const uint8_t *>(a.in_words + d);
            for (uint32_t k = 0; k < a.tail_bytes; ++k) v |= (uint32_t)p[k] << (8u * k);
        }
        return v;
    }
    FQTK_HD inline void reset(W &w, const MemberArgs &a, uint32_t bit) {
        filled = (bit >> 5) & ~63u;
        pref = load(w, a, filled + w.lane());
    }
    // after this, dwords up to (bit >> 5) + 7 are in the ring
    FQTK_HD inline void ensure(W &w, Shared &S, const MemberArgs &a, uint32_t bit) {
        const uint32_t need = (bit >> 5) + 8u;
        if (filled >= need) return;
        w.barrier();   // (reads of the ring issued before the overwrite)
        while (filled < need) {
            S.ring[(filled + w.lane()) & (kRingWords - 1u)] = pref;
            filled += 64u;
            pref = load(w, a, filled + w.lane());
        }
        w.barrier();
    }
};

// Builds the canonical description of a code from lens[0..n) and its fast table (2^P entries).  Returns an error code.
// kTables = false: only whether the lengths make a code the decoder accepts (the block-start search asks no more).
template <class W, bool kLitLen, bool kTables = true>
FQTK_HD inline uint32_t build_code(W &w, Shared &S, const uint8_t *lens, uint32_t n) {
    constexpr uint32_t P = kLitLen ? kLitBits : kDistBits;
    Canon &C = kLitLen ? S.cl : S.cd;
    uint16_t *perm = kLitLen ? S.perm_l : S.perm_d;
    uint32_t *tab = kLitLen ? S.lit : S.dist;
    const uint32_t lane = w.lane();
    // (1) how many codes of each length, and each symbol's rank among the symbols of its length (in symbol order)
    uint32_t total[16];
FQTK_UNROLL
    for (int l = 0; l < 16; ++l) total[l] = 0;
    uint32_t my_rank[5], my_len[5];
FQTK_UNROLL
    for (uint32_t p = 0; p < 5u; ++p) {
        const uint32_t s = p * 64u + lane;
        const uint32_t len = (kLitLen || p == 0u) && s < n ? (uint32_t)lens[s] : 0u;
        my_len[p] = len;
        my_rank[p] = 0;
        if (!kLitLen && p > 0u) continue;
FQTK_UNROLL
        for (uint32_t l = 1; l < 16u; ++l) {
            const uint64_t m = w.ballot(len == l);
            if (len == l) my_rank[p] = total[l] + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull));
            total[l] += (uint32_t)__builtin_popcountll(m);
        }
    }
    // (2) the canonical numbers (the same in every lane)
    uint32_t first = 0, offs = 0, used = 0, maxlen = 0;
    int32_t left = 1;
    bool over = false;
    uint32_t my_offs[5];
FQTK_UNROLL
    for (uint32_t p = 0; p < 5u; ++p) my_offs[p] = 0;
    if (lane == 0u) { C.limit[0] = 0; C.first[0] = 0; C.offs[0] = 0; C.cnt[0] = 0; }
FQTK_UNROLL
    for (uint32_t l = 1; l < 16u; ++l) {
        const uint32_t c = total[l];
        left = (left << 1) - (int32_t)c;
        if (left < 0) over = true;
        if (c) maxlen = l;
        used += c;
        if (lane == 0u) {
            C.first[l] = first << (15u - l);
            C.limit[l] = (first + c) << (15u - l);
            C.offs[l] = (uint16_t)offs;
            C.cnt[l] = (uint16_t)c;
        }
FQTK_UNROLL
        for (uint32_t p = 0; p < 5u; ++p)
            if (my_len[p] == l) my_offs[p] = offs;
        offs += c;
        first = (first + c) << 1;
    }
    if (over) return kErrOverSubscribed;
    // incomplete codes: legal only as "at most one code, of one bit" (a block with one distance code, or none)
    if (left > 0 && !(used <= 1u && maxlen <= 1u)) return kErrOverSubscribed;
    if (!kTables) return kOk;
FQTK_UNROLL
    for (uint32_t p = 0; p < 5u; ++p)
        if (my_len[p]) perm[my_offs[p] + my_rank[p]] = (uint16_t)(p * 64u + lane);
    w.barrier();
    // (3) the fast table, entry by entry: what code stands in front of these P bits?
    uint32_t lim[16];
FQTK_UNROLL
    for (int l = 1; l < 16; ++l) lim[l] = w.uniform(C.limit[l]);
    for (uint32_t idx = lane; idx < (1u << P); idx += 64u) {
        const uint32_t rev = brev32(idx) >> 17;   // the P index bits, first bit at bit 14; the bits behind them: 0
        uint32_t len = 1;
FQTK_UNROLL
        for (int l = 1; l < 15; ++l) len += rev >= lim[l] ? 1u : 0u;
        uint32_t e;
        if (rev >= lim[15]) {
            // past the last code: no code starts with these bits -- unless a longer code does (bits behind the index are 0 here,
            // so a code of more than P bits whose first P bits are these compares below its limit; reaching here means none does)
            e = 0u;
        } else if (len > P) {
            e = make_entry(0, 0, kLong, 0) | 1u;   // code length unknown yet (non-zero marks "a code exists")
        } else {
            const uint32_t sym = perm[C.offs[len] + ((rev - C.first[len]) >> (15u - len))];
            e = kLitLen ? litlen_entry(sym, len) : dist_entry(sym, len);
        }
        tab[idx] = e;
    }
    w.barrier();
    return kOk;
}

// The code in front of `rev` (15 stream bits, first bit at bit 14): symbol and length, canonically.  false: no such code.
FQTK_HD inline bool canon_decode(const Canon &C, const uint16_t *perm, uint32_t rev, uint32_t *sym, uint32_t *len_out) {
    uint32_t len = 1;
FQTK_UNROLL
    for (int l = 1; l < 15; ++l) len += rev >= C.limit[l] ? 1u : 0u;
    if (rev >= C.limit[15]) return false;
    *sym = perm[C.offs[len] + ((rev - C.first[len]) >> (15u - len))];
    *len_out = len;
    return true;
}

// One speculative token: what stands at the start of `bits` (64 stream bits)?
struct Token {
    uint32_t nbits;    // bits of the whole token
    uint32_t outlen;   // bytes it produces
    uint32_t value;    // literal byte, or the distance
    uint32_t flags;    // kTok*
};
enum : uint32_t { kTokMatch = 1, kTokEob = 2, kTokSlow = 4, kTokBad = 8 };

template <bool kSlow>
FQTK_HD inline Token decode_token(const Shared &S, uint64_t bits) {
    Token t;
    t.nbits = 1; t.outlen = 0; t.value = 0; t.flags = 0;
    uint32_t e = S.lit[(uint32_t)bits & ((1u << kLitBits) - 1u)];
    if (((e >> 8) & 3u) == kLong) {
        if (!kSlow) { t.flags = kTokSlow; return t; }
        uint32_t sym, len;
        if (!canon_decode(S.cl, S.perm_l, brev32((uint32_t)bits) >> 17, &sym, &len)) { t.flags = kTokBad; return t; }
        e = litlen_entry(sym, len);
    }
    const uint32_t cl = e & 15u;
    if (cl == 0u) { t.flags = kTokBad; return t; }
    const uint32_t kind = (e >> 8) & 3u;
    if (kind == kLit) { t.nbits = cl; t.outlen = 1; t.value = e >> 16; return t; }
    if (kind == kEob) { t.nbits = cl; t.flags = kTokEob; return t; }
    const uint32_t ne = (e >> 4) & 15u;
    const uint32_t len = (e >> 16) + ((uint32_t)(bits >> cl) & ((1u << ne) - 1u));
    const uint32_t u = cl + ne;   // <= 20
    const uint32_t dbits = (uint32_t)(bits >> u);
    uint32_t de = S.dist[dbits & ((1u << kDistBits) - 1u)];
    if (((de >> 8) & 3u) == kLong) {
        if (!kSlow) { t.flags = kTokSlow; return t; }
        uint32_t sym, dl;
        if (!canon_decode(S.cd, S.perm_d, brev32(dbits) >> 17, &sym, &dl)) { t.flags = kTokBad; return t; }
        de = dist_entry(sym, dl);
    }
    const uint32_t dcl = de & 15u;
    if (dcl == 0u) { t.flags = kTokBad; return t; }
    const uint32_t dne = (de >> 4) & 15u;
    t.value = (de >> 16) + ((uint32_t)(bits >> (u + dcl)) & ((1u << dne) - 1u));
    t.nbits = u + dcl + dne;   // <= 48
    t.outlen = len;
    t.flags = kTokMatch;
    return t;
}

// The common case in straight-line code: both look-ups are unconditional (a lane whose first code is a literal reads
// some distance entry it does not use) and the token's fields are selected at the end -- the 64 lanes decode 64 different
// bit positions, so every branch of a branchy version is taken by somebody and costs the wave its full length.
// 32 bits of a 64-bit window from bit s < 32 on: one v_alignbit_b32 (a 64-bit shift is two to four times a 32-bit instruction)
// A value the optimiser must take as it is HERE (device builds): work that hangs on it stays in the branch that asks for it.
FQTK_HD inline uint64_t opaque64(uint64_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    asm volatile("" : "+v"(lo), "+v"(hi));
    return (uint64_t)lo | ((uint64_t)hi << 32);
#else
    return v;
#endif
}
FQTK_HD inline uint32_t funnel32(uint32_t hi, uint32_t lo, uint32_t s) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbit(hi, lo, s);
#else
    return (uint32_t)((((uint64_t)hi << 32) | lo) >> (s & 31u));
#endif
}
FQTK_HD inline Token decode_token_fast(const Shared &S, uint32_t b0, uint32_t b1) {   // the 64 bits from the token's first one on, low word first
    const uint32_t e = S.lit[b0 & ((1u << kLitBits) - 1u)];
    const uint32_t cl = e & 15u, ne = (e >> 4) & 15u, kind = (e >> 8) & 3u, val = e >> 16;
    const uint32_t len = val + (funnel32(b1, b0, cl) & ((1u << ne) - 1u));
    const uint32_t u = cl + ne;   // <= 28
    const uint32_t dbits = funnel32(b1, b0, u);   // the distance code, its extra bits behind it: 15 + 13 of these 32
    const uint32_t de = S.dist[dbits & ((1u << kDistBits) - 1u)];
    const uint32_t dcl = de & 15u, dne = (de >> 4) & 15u, dkind = (de >> 8) & 3u;
    const uint32_t dist = (de >> 16) + ((dbits >> dcl) & ((1u << dne) - 1u));
    const bool is_len = kind == kLen;
    const bool slow = kind == kLong || (is_len && dkind == kLong);
    const bool bad = !slow && (cl == 0u || (is_len && dcl == 0u));
    Token t;
    t.nbits = is_len ? u + dcl + dne : cl;
    t.outlen = is_len ? len : (kind == kLit ? 1u : 0u);
    t.value = is_len ? dist : val;
    t.flags = slow ? kTokSlow : (bad ? kTokBad : (is_len ? kTokMatch : (kind == kEob ? kTokEob : 0u)));
    return t;
}

// The header of a dynamic-Huffman block behind its three type bits (RFC 1951 3.2.7): HLIT, HDIST, HCLEN, the code-length code and the
// run-length coded code lengths, which go to S.lens (hlit literal/length ones, then hdist distance ones).  `bit` moves behind them.
// Returns an error code (the same in all lanes); used by the decoder and by the block-start search.
template <class W>
FQTK_HD inline uint32_t read_dynamic_header(W &w, Shared &S, const MemberArgs &a, Ring<W> &ring, uint32_t &bit, uint32_t &hlit, uint32_t &hdist) {
    const uint32_t lane = w.lane();
    ring.ensure(w, S, a, bit);
    const uint32_t hh = w.uniform(peek32(w, S, bit));
    hlit = (hh & 31u) + 257u; hdist = ((hh >> 5) & 31u) + 1u;
    const uint32_t hclen = ((hh >> 10) & 15u) + 4u;
    bit += 14u;
    if (hlit > 286u || hdist > 30u) return kErrCodeLengths;
    // the code-length code's 19 lengths: lane k takes the k-th (3 bits each, in the order of RFC 1951 3.2.7)
    ring.ensure(w, S, a, bit + 64u);
    uint32_t my_cl_len = 0;   // lane s: length of code-length symbol s
    // position of symbol s in the transmitted order 16 17 18 0 8 7 9 6 10 5 11 4 12 3 13 2 14 1 15
    uint32_t pos_of = 19u;
    if (lane < 19u) {
        const uint32_t s = lane;
        if (s >= 16u) pos_of = s - 16u;
        else if (s == 0u) pos_of = 3u;
        else if (s >= 8u) pos_of = 4u + 2u * (s - 8u);          // 8 -> 4, 9 -> 6, 10 -> 8, ... 15 -> 18
        else pos_of = 5u + 2u * (7u - s);                      // 7 -> 5, 6 -> 7, ... 1 -> 17
    }
    if (pos_of < hclen) my_cl_len = peek32(w, S, bit + 3u * pos_of) & 7u;
    bit += 3u * hclen;
    // its canonical code (<= 7 bits, 19 symbols) and a 128-entry table, by the wave
    uint32_t ctot[8];
FQTK_UNROLL
    for (int l = 0; l < 8; ++l) ctot[l] = 0;
    uint32_t my_rank = 0;
FQTK_UNROLL
    for (uint32_t l = 1; l < 8u; ++l) {
        const uint64_t m = w.ballot(my_cl_len == l);
        if (my_cl_len == l) my_rank = (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull));
        ctot[l] = (uint32_t)__builtin_popcountll(m);
    }
    uint32_t code = 0, my_code = 0;
    int32_t left = 1;
    bool over = false;
FQTK_UNROLL
    for (uint32_t l = 1; l < 8u; ++l) {
        left = (left << 1) - (int32_t)ctot[l];
        if (left < 0) over = true;
        if (my_cl_len == l) my_code = code + my_rank;
        code = (code + ctot[l]) << 1;
    }
    if (over) return kErrCodeLengths;
    uint32_t *const cl_tab = S.lit;   // the code-length code: 7 index bits -> symbol << 16 | length (0: no such code); 128 entries of the literal table's room
    cl_tab[lane] = 0u;
    cl_tab[lane + 64u] = 0u;
    w.barrier();
    if (my_cl_len) {
        const uint32_t rev = brev32(my_code) >> (32u - my_cl_len);
        for (uint32_t k = rev; k < 128u; k += 1u << my_cl_len) cl_tab[k] = (lane << 16) | my_cl_len;
    }
    w.barrier();
    // the hlit + hdist code lengths, run-length coded: a serial chain, every lane follows it
    const uint32_t total = hlit + hdist;
    uint32_t i = 0, prev = 0;
    while (i < total) {
        ring.ensure(w, S, a, bit);
        const uint32_t b = w.uniform(peek32(w, S, bit));
        const uint32_t e = w.uniform(cl_tab[b & 127u]);
        const uint32_t l = e & 0xFFu;
        if (l == 0u) return kErrCodeLengths;
        const uint32_t sym = e >> 16;
        bit += l;
        const uint32_t x = b >> l;
        if (sym < 16u) {
            if (lane == 0u) S.lens[i] = (uint8_t)sym;
            prev = sym;
            ++i;
        } else {
            uint32_t rep, v = 0;
            if (sym == 16u) {
                if (i == 0u) return kErrCodeLengths;
                v = prev; rep = 3u + (x & 3u); bit += 2u;
            } else if (sym == 17u) {
                rep = 3u + (x & 7u); bit += 3u;
            } else {
                rep = 11u + (x & 127u); bit += 7u;
            }
            if (i + rep > total) return kErrCodeLengths;
            for (uint32_t k = lane; k < rep; k += 64u) S.lens[i + k] = (uint8_t)v;
            prev = v;
            i += rep;
        }
    }
    w.barrier();
    if (w.uniform(S.lens[256]) == 0u) return kErrCodeLengths;
    return kOk;
}

// The first bit in [from_bit, limit_bit) at which a non-final dynamic-Huffman block can start: BFINAL = 0, BTYPE = 2, HLIT and HDIST in
// range, a COMPLETE code-length code -- tested by lane i for bit t + i, 64 positions at a time, on bits fetched straight from memory -- and,
// for the few positions that pass (the wave takes them one at a time, in order), a header that parses into two codes the decoder accepts.
// That is host/parallel_gunzip.hpp's SpecInflate::find_block_start on a wavefront: a one-in-many-millions accident where no block starts,
// and nothing rests on it -- a chunk cut at such a place counts only if the chunk before it ends on exactly that bit.  Bits count from
// a.in_words; ~0: none.  (Stored and fixed blocks are not looked for: a chunk just runs on through them.)
// An accidental header still turns up about once per 100 MB of FASTQ.gz (measured: one in a 60 MB stream), and in stream mode a chunk that
// starts at one decodes garbage without an error (any bits decode under two complete codes, any distance reaches into the unknown window);
// the chain refuses it, at the price of the rest of its stretch.  low_literals_only makes such headers rarer by orders of magnitude.
// Kraft sums of four 3-bit code lengths at a time (the code-length code has seven index bits: a length l weighs 128 >> l, none weighs 0): entry
// g of 4096 is the weight of the four lengths in g's twelve bits, saturated at 255 (four lengths of 1 weigh 256: no complete code has them).
// The block-start search tests the code-length code of EVERY candidate position for completeness -- nineteen extract / shift / select / add
// steps per lane and window until round 6, five look-ups now.  Built by the wavefront that searches (64 entries per lane).
constexpr uint32_t kKraftLutBytes = 4096;
template <class W>
FQTK_HD inline void build_kraft_lut(W &w, uint8_t *lut) {
    for (uint32_t g = w.lane(); g < kKraftLutBytes; g += 64u) {
        uint32_t sum = 0;
FQTK_UNROLL
        for (uint32_t k = 0; k < 4u; ++k) { const uint32_t l = (g >> (3u * k)) & 7u; sum += l ? 128u >> l : 0u; }
        lut[g] = (uint8_t)(sum > 255u ? 255u : sum);
    }
    w.barrier();
}

template <class W>
FQTK_HD inline uint32_t find_block_start(W &w, Shared &S, const MemberArgs &a, uint32_t from_bit, uint32_t limit_bit, bool low_literals_only = false,
                                         const uint8_t *kraft_lut = nullptr) {
    const uint32_t lane = w.lane();
    const uint32_t end_bit = a.first_bit + a.payload_bits;
    // The bits come through the ring in LDS, 64 dwords a refill, like the decoder's (until round 6 every lane fetched its four dwords from
    // memory for every 64 positions: a round trip to the L2 per step, 5 ms of a 1023-chunk stretch's 20).
    Ring<W> ring;
    ring.reset(w, a, from_bit);
    for (uint32_t t0 = from_bit; t0 < limit_bit; t0 += 64u) {
        ring.ensure(w, S, a, t0);   // dwords up to (t0 >> 5) + 7: lane 63 reads (t0 + 63 >> 5) + 3
        const uint32_t t = t0 + lane, d = t >> 5, s = t & 31u;
        const uint64_t lo = (uint64_t)S.ring[d & (kRingWords - 1u)] | ((uint64_t)S.ring[(d + 1u) & (kRingWords - 1u)] << 32);
        const uint64_t hi = (uint64_t)S.ring[(d + 2u) & (kRingWords - 1u)] | ((uint64_t)S.ring[(d + 3u) & (kRingWords - 1u)] << 32);
        const uint64_t b = s ? (lo >> s) | (hi << (64u - s)) : lo;   // bits t .. t + 63
        const uint64_t c = hi >> s;                                   // bits t + 64 .. (32 of them at least)
        const uint32_t hclen = ((uint32_t)(b >> 13) & 15u) + 4u;
        bool pass = t < limit_bit && ((uint32_t)b & 7u) == 4u && ((uint32_t)(b >> 3) & 31u) <= 29u && ((uint32_t)(b >> 8) & 31u) <= 29u &&
                    t + 17u + 3u * hclen < end_bit;
        if (w.ballot(pass)) {
            // Kraft sum of the code-length code: 3-bit fields k < hclen from bit 17 on (fields 0-14 in b, 15-18 from bit 62 on)
            const uint64_t c0 = b >> 17, c1 = (b >> 62) | (c << 2);
            uint32_t kraft = 0;
            if (kraft_lut) {
                // the lengths that are there: 3 * hclen bits of c0 (fields 0-14) and c1 (fields 15-18); what lies behind them weighs nothing
                const uint32_t nb = 3u * hclen;                                   // 12 .. 57
                const uint64_t f0 = nb >= 45u ? c0 & ((1ull << 45) - 1ull) : c0 & ((1ull << nb) - 1ull);
                const uint32_t f1 = nb > 45u ? (uint32_t)c1 & ((1u << (nb - 45u)) - 1u) : 0u;
                kraft = (uint32_t)kraft_lut[(uint32_t)f0 & 0xFFFu] + (uint32_t)kraft_lut[(uint32_t)(f0 >> 12) & 0xFFFu] + (uint32_t)kraft_lut[(uint32_t)(f0 >> 24) & 0xFFFu] +
                        (uint32_t)kraft_lut[(uint32_t)(f0 >> 36) & 0x1FFu] + (uint32_t)kraft_lut[f1 & 0xFFFu];
            } else {
FQTK_UNROLL
                for (uint32_t k = 0; k < 19u; ++k) {
                    const uint32_t l = (uint32_t)(k < 15u ? c0 >> (3u * k) : c1 >> (3u * (k - 15u))) & 7u;
                    kraft += k < hclen && l ? 128u >> l : 0u;
                }
            }
            pass = pass && kraft == 128u;
        }
        uint64_t m = w.ballot(pass);
        while (m) {
            const uint32_t cand = (uint32_t)__builtin_ctzll(m);
            m &= m - 1ull;
            uint32_t bit = t0 + cand + 3u, hlit = 0, hdist = 0;
            // (the header is read through the SAME ring, which only ever moves forward: a header is 2 300 bits at most -- 72 dwords -- and the
            //  ring keeps 256, so the search's own place is still in it afterwards; a ring started afresh for every candidate, and again for
            //  the search, was two round trips to memory per candidate -- one in seven windows has one)
            uint32_t err = read_dynamic_header(w, S, a, ring, bit, hlit, hdist);
            if (!err && bit > end_bit) err = kErrTruncated;
            // (the caller has seen nothing but 7-bit text in this stream so far: a header that gives codes to literals >= 128 -- as nearly every
            //  accidental one does -- is not taken for a start; a true one that is refused only makes the chunk before it longer)
            if (!err && low_literals_only && w.ballot(S.lens[128u + lane] != 0 || S.lens[192u + lane] != 0)) err = kErrBadCode;
            if (!err) err = build_code<W, true, false>(w, S, S.lens, hlit);
            if (!err) err = build_code<W, false, false>(w, S, S.lens + hlit, hdist);
            if (!err) return t0 + cand;
        }
    }
    return 0xFFFFFFFFu;
}

// Decodes one member.  Returns its status (the same in all lanes).
// kStream: the same decoder on a piece of a SERIAL gzip stream (`gzip`, bcl2fastq: one member per file) -- it starts at a
// block boundary inside the stream without the 32 KiB of text before it, so it writes 16-bit symbols (a byte, or "byte j of the
// window I do not have"; a copy of such a symbol copies the reference), goes on from block to block and stops at the first
// block boundary at or behind a.stop_bit.  The windows are handed down the chain of chunks afterwards (fqtk_inflate.hip),
// the way host/parallel_gunzip.hpp does it on CPU threads.
template <class W, bool kStream = false>
FQTK_HD inline uint32_t inflate_member(W &w, Shared &S, const MemberArgs &a, StreamEnd *end = nullptr) {
    const uint32_t lane = w.lane();
    Ring<W> ring;
    uint32_t bit = a.first_bit;                    // uniform: next unread bit of the stream
    const uint32_t end_bit = a.first_bit + a.payload_bits;
    uint32_t out_pos = 0;                          // uniform
    uint32_t safe = 0;                             // output below this is visible to every lane's loads
    uint32_t own_tag = 0;                          // rounds of 64 output bytes so far (24 bits are plenty: <= 65 536 bytes a member)
    uint32_t stream_flags = 0;                     // uniform
    ring.reset(w, a, bit);
    S.own[lane] = 0u;   // (tag 0 is never a round's)
    w.barrier();
    for (;;) {
        // ---- block header (RFC 1951 3.2.3): the same work in every lane
        ring.ensure(w, S, a, bit);
        const uint32_t h = w.uniform(peek32(w, S, bit));
        const uint32_t final_block = h & 1u, type = (h >> 1) & 3u;
        bit += 3u;
        if (type == 3u) return kErrBlockType;
        if (type == 0u) {
            bit = (bit + 7u) & ~7u;
            ring.ensure(w, S, a, bit);
            const uint32_t ll = w.uniform(peek32(w, S, bit));
            const uint32_t len = ll & 0xFFFFu, nlen = ll >> 16;
            if ((len ^ 0xFFFFu) != nlen) return kErrStoredLen;
            bit += 32u;
            if (bit + 8u * len > end_bit) return kErrTruncated;
            if (out_pos + len > a.isize) return kErrOutput;
            const uint8_t *src = reinterpret_cast<const uint8_t *>(a.in_words) + (bit >> 3);
            for (uint32_t k = lane; k < len; k += 64u) {
                if (kStream) a.out_sym[out_pos + k] = src[k]; else a.out[out_pos + k] = src[k];
            }
            out_pos += len;
            bit += 8u * len;
            ring.reset(w, a, bit);
        } else {
            uint32_t hlit, hdist;
            if (type == 1u) {
                hlit = 288u; hdist = 32u;
                for (uint32_t s = lane; s < 320u; s += 64u) S.lens[s] = (uint8_t)(s < 144u ? 8u : s < 256u ? 9u : s < 280u ? 7u : s < 288u ? 8u : 5u);
                w.barrier();
            } else {
                const uint32_t herr = read_dynamic_header(w, S, a, ring, bit, hlit, hdist);
                if (herr) return herr;
            }
            if (bit > end_bit) return kErrTruncated;
            if (kStream && w.ballot(S.lens[128u + lane] != 0 || S.lens[192u + lane] != 0)) stream_flags |= kStreamHighLiterals;
            uint32_t err = build_code<W, true>(w, S, S.lens, hlit);
            if (err) return err;
            err = build_code<W, false>(w, S, S.lens + hlit, hdist);
            if (err) return err;

            // ---- the block's tokens, kSets x 64 bit positions at a time: lane i takes bit i of every set of 64
            for (;;) {
                ring.ensure(w, S, a, bit + 64u * kSets + 64u);
                const uint32_t b = bit + lane, d = b >> 5, s = b & 31u;
                uint32_t dw[2 * kSets + 1];
                FQTK_UNROLL
                for (uint32_t k = 0; k < 2u * kSets + 1u; ++k) dw[k] = S.ring[(d + k) & (kRingWords - 1u)];
                uint64_t bits[kSets];
                Token t[kSets];
                uint64_t slow[kSets];
                uint32_t meta[kSets];
                // A lane's step in the walk below is its token's bits; the rare ones (a long code to resolve, no code at all, end of
                // block) step 0x80 -- out of the window -- so that the walk has ONE exit test, and are looked at behind it.
                FQTK_UNROLL
                for (uint32_t q = 0; q < kSets; ++q) {
                    const uint32_t b0 = funnel32(dw[2 * q + 1], dw[2 * q], s), b1 = funnel32(dw[2 * q + 2], dw[2 * q + 1], s);   // (s < 32; s = 0: the words themselves)
                    bits[q] = (uint64_t)b0 | ((uint64_t)b1 << 32);
                    t[q] = decode_token_fast(S, b0, b1);
                    slow[q] = w.ballot((t[q].flags & kTokSlow) != 0u);
                    meta[q] = t[q].flags & (kTokSlow | kTokBad | kTokEob) ? 0x80u : t[q].nbits;
                }
                // the chain of real tokens: one scalar step per token, set after set
                uint64_t chain[kSets];
                FQTK_UNROLL
                for (uint32_t q = 0; q < kSets; ++q) chain[q] = 0;
                uint32_t cur = 0;
                bool eob = false;
                for (;;) {
                    uint32_t m = 0;
                    FQTK_UNROLL
                    for (uint32_t q = 0; q < kSets; ++q) {
                        if (cur >= 64u * q && cur < 64u * (q + 1u)) {
                            // (four steps per turn of the loop: a step is a v_readlane and four scalar instructions, and a TAKEN branch costs
                            //  as much again -- three of four are now branches that fall through)
                            for (;;) {
#define FQTK_WALK_STEP                                                      \
                                m = w.readlane(meta[q], cur - 64u * q);     \
                                w.set_bit64(chain[q], cur - 64u * q);       \
                                cur += m;                                   \
                                if (cur >= 64u * (q + 1u)) break;
                                FQTK_WALK_STEP FQTK_WALK_STEP FQTK_WALK_STEP FQTK_WALK_STEP
#undef FQTK_WALK_STEP
                            }
                        }
                    }
                    if (!(m & 0x80u)) break;
                    cur -= 0x80u;   // a rare one: back on its position
                    if (slow[0] | (kSets == 2 ? slow[kSets - 1] : 0ull)) {   // long codes: every lane that met one resolves it now (once per window at most)
                        FQTK_UNROLL
                        for (uint32_t q = 0; q < kSets; ++q) {
                            if ((slow[q] >> lane) & 1ull) {
                                // (opaque: the canonical look-up -- fourteen compares per code -- depends on nothing the walk changes, and the
                                //  compiler otherwise computes it for every lane of EVERY window in front of the walk: ~90 of a window's ~300
                                //  vector instructions for a path one window in hundreds takes)
                                t[q] = decode_token<true>(S, opaque64(bits[q]));
                                meta[q] = t[q].flags & (kTokBad | kTokEob) ? 0x80u : t[q].nbits;
                            }
                            slow[q] = 0;
                        }
                        continue;   // the walk goes on from here
                    }
                    const uint32_t cq = cur >> 6, cl = cur & 63u;
                    const uint32_t f = kSets == 2 && cq ? w.readlane(t[kSets - 1].flags, cl) : w.readlane(t[0].flags, cl);
                    if (f & kTokBad) return kErrBadCode;
                    cur += kSets == 2 && cq ? w.readlane(t[kSets - 1].nbits, cl) : w.readlane(t[0].nbits, cl);   // end of block
                    eob = true;
                    break;
                }
                bool mine[kSets];
                uint32_t my_out[kSets], my_pos[kSets];
                uint32_t produced = 0;
                FQTK_UNROLL
                for (uint32_t q = 0; q < kSets; ++q) {
                    mine[q] = ((chain[q] >> lane) & 1ull) != 0ull;
                    my_out[q] = mine[q] ? t[q].outlen : 0u;
                }
                if (kSets == 2) {
                    // both sets' prefix sums in ONE scan: a token makes 258 bytes at most, 64 of them 16 512 -- sixteen bits a set
                    const uint32_t incl = w.scan_incl(my_out[0] | (my_out[kSets - 1] << 16));
                    const uint32_t tot = w.readlane(incl, 63u), tot0 = tot & 0xFFFFu;
                    my_pos[0] = out_pos + (incl & 0xFFFFu) - my_out[0];
                    my_pos[kSets - 1] = out_pos + tot0 + (incl >> 16) - my_out[kSets - 1];
                    produced = tot0 + (tot >> 16);
                } else {
                    const uint32_t incl = w.scan_incl(my_out[0]);
                    my_pos[0] = out_pos + incl - my_out[0];
                    produced = w.readlane(incl, 63u);
                }
                if (out_pos + produced > a.isize) return kErrOutput;
                // ---- a window of literals only (most windows of text: bases, qualities): every token is one byte, the prefix sum says
                // where -- a lane per token stores it, and the machinery below (owners, sources, pointer doubling: ~70 vector
                // instructions per 64 bytes) is for the windows that hold a match
                {
                    bool match_here = false;
                    FQTK_UNROLL
                    for (uint32_t q = 0; q < kSets; ++q) match_here = match_here || (mine[q] && (t[q].flags & kTokMatch) != 0u);
                    if (!w.ballot(match_here)) {
                        FQTK_UNROLL
                        for (uint32_t q = 0; q < kSets; ++q) {
#ifndef FQTK_INFLATE_ABL_NOSTORE
                            if (my_out[q]) { if (kStream) a.out_sym[my_pos[q]] = (uint16_t)t[q].value; else a.out[my_pos[q]] = (uint8_t)t[q].value; }
#endif
                        }
                        out_pos += produced;
                        bit += cur;
                        if (eob) break;
                        if (bit > end_bit) return kErrTruncated;
                        continue;
                    }
                }
                // ---- the window's bytes, 64 at a time, one lane per BYTE (not per token): who owns the byte (the last token that
                // starts at or before it: a prefix maximum over the tokens' starting places), what it is (the owner's literal, or
                // the byte `distance` back -- in memory already, or another byte of these 64, reached by pointer doubling), one
                // coalesced store.  A match costs what a literal costs, and the wave waits for memory once per 64 bytes
                // instead of once per match.
                uint32_t rel_start[kSets], packed[kSets];
                bool has_out[kSets], too_far = false;
                FQTK_UNROLL
                for (uint32_t q = 0; q < kSets; ++q) {
                    const bool is_match = (t[q].flags & kTokMatch) != 0u;
                    too_far = too_far || (mine[q] && is_match && t[q].value > my_pos[q] + (kStream ? kWindow : 0u));
                    rel_start[q] = my_pos[q] - out_pos;                       // < 128 * 258: 16 bits
                    // kind | start << 1 | (distance - 1, or the literal) << 17
                    packed[q] = (is_match ? 1u : 0u) | (rel_start[q] << 1) | ((is_match ? t[q].value - 1u : t[q].value) << 17);
                    has_out[q] = mine[q] && my_out[q] != 0u;
                }
                if (w.ballot(too_far)) return kErrDistance;
                uint32_t carry = 0;                                           // 1 + position of the token that owns the byte before these 64
#ifdef FQTK_INFLATE_ABL_NOCOPY   // (developer ablation, tools/ab_inflate.sh: wrong output, same decoding)
                for (uint32_t base = produced; base < produced; base += 64u) {
#else
                for (uint32_t base = 0; base < produced; base += 64u) {
#endif
                    // (entries carry the round's tag: a stale one reads as "no token starts here", and nothing is cleared)
                    ++own_tag;
                    FQTK_UNROLL
                    for (uint32_t q = 0; q < kSets; ++q)
                        if (has_out[q] && rel_start[q] >= base && rel_start[q] < base + 64u) S.own[rel_start[q] - base] = (own_tag << 8) | (64u * q + lane + 1u);
                    w.barrier();
                    const uint32_t own_raw = S.own[lane];
                    w.barrier();   // (read before the next round's writes)
                    uint32_t own = w.scan_max_incl((own_raw >> 8) == own_tag ? (own_raw & 0xFFu) : 0u);
                    own = own > carry ? own : carry;
                    carry = w.readlane(own, 63u);
                    const uint32_t j = base + lane;
                    const bool live = j < produced;
                    uint32_t info = w.shuffle(packed[0], (own - 1u) & 63u);   // (byte 0 starts a token: own >= 1)
                    if (kSets == 2) {
                        const uint32_t info_b = w.shuffle(packed[kSets - 1], (own - 1u) & 63u);
                        info = own > 64u ? info_b : info;
                    }
                    const uint32_t start = (info >> 1) & 0xFFFFu, val = info >> 17;
                    uint32_t byte = val & 0xFFu;
                    uint32_t ptr = lane;                                      // a byte that is known points at itself
                    bool from_memory = false;
                    uint32_t src_pos = 0;
                    if (live && (info & 1u)) {
                        const uint32_t k = j - start, dist = val + 1u;
                        uint32_t off = k;
                        if (k >= dist) {                                      // an overlapping match repeats its `dist` bytes: k mod dist (k, dist < 259)
                            const uint32_t qq = (uint32_t)((float)k * w.rcp((float)dist));   // (approximate: corrected below)
                            int32_t r = (int32_t)k - (int32_t)(qq * dist);
                            if (r < 0) r += (int32_t)dist; else if (r >= (int32_t)dist) r -= (int32_t)dist;
                            off = (uint32_t)r;
                        }
                        const int32_t rel = (int32_t)(start + off) - (int32_t)dist;   // the source, counted from the window's first byte
                        if (rel < (int32_t)base) {
                            const int32_t sp = (int32_t)out_pos + rel;
                            if (kStream && sp < 0) byte = 256u + (uint32_t)((int32_t)kWindow + sp);   // in front of the chunk: a reference
                            else { from_memory = true; src_pos = (uint32_t)sp; }
                        } else ptr = (uint32_t)rel - base;
                    }
                    if (w.ballot(from_memory && src_pos >= safe)) {           // stores that may still be on their way
                        w.fence_global();
                        safe = out_pos + base;
                    }
                    if (from_memory) byte = kStream ? (uint32_t)a.out_sym[src_pos] : (uint32_t)a.out[src_pos];
                    for (;;) {                                                // ptr -> ptr of ptr until every byte points at a known one (<= 6 rounds)
                        const uint32_t pp = w.shuffle(ptr, ptr);
                        if (!w.ballot(pp != ptr)) break;
                        ptr = pp;
                    }
                    byte = w.shuffle(byte, ptr);
#ifndef FQTK_INFLATE_ABL_NOSTORE
                    if (live) { if (kStream) a.out_sym[out_pos + j] = (uint16_t)byte; else a.out[out_pos + j] = (uint8_t)byte; }
#endif
                }
                out_pos += produced;
                bit += cur;
                if (eob) break;
                if (bit > end_bit) return kErrTruncated;
            }
        }
        if (bit > end_bit) return kErrTruncated;
        if (kStream && end && lane == 0u) { end->n_sym = out_pos; end->end_bit = bit; end->final_block = final_block; end->n_blocks += 1u; end->flags = stream_flags; }
        if (final_block || (kStream && bit >= a.stop_bit)) break;
    }
    if (!kStream && out_pos != a.isize) return kErrLength;
    return kOk;
}

}  // namespace inflate
}  // namespace fqtk
