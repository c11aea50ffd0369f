// fast_inflate.hpp -- streaming DEFLATE / gzip decoder for single-stream .gz inputs.
//
// A gzip member is one serial bit stream: nothing about it can be split across threads, so a `.fastq.gz` written by
// gzip / bcl2fastq is read at the speed of ONE inflate loop.  zlib's (the reference reads through flate2's zlib
// binding, fgoxide `Io`, demux.rs:844-849) runs at ~450 MB/s of output here, which bounds `fqtk demux` at ~1.3 M
// templates/s for 150-base reads; libdeflate's decoder is 2-3x faster but has no streaming interface (it wants the
// whole member's output in one buffer).  This is a streaming decoder built the way the fast ones are:
//   * a 64-bit bit buffer refilled without a branch (one unaligned 8-byte load per refill);
//   * one table look-up per symbol: 11 index bits for literals/lengths, 8 for distances, second-level tables only
//     for longer codes; an entry holds everything the symbol needs (literal byte / base value, extra-bit count,
//     code length);
//   * two literals per look-up where both codes fit the 11 index bits, three look-ups per refill, matches copied
//     eight bytes at a time;
//   * no bounds checks inside the loop: the input is a memory-mapped file whose last bytes are decoded from a
//     zero-padded copy, and the output buffer keeps 258 + 16 bytes of slack behind the stop mark.
// The caller maps the file, calls next() for successive pieces of output (each piece continues where the last one
// stopped, mid-block if need be; the 32 KiB window is carried inside), and gets CRC32 / ISIZE verification of every
// member, concatenated members included (RFC 1952 2.2).  Written from RFC 1951 / RFC 1952; no code of zlib or
// libdeflate is used.  Parity: tests/test_host_components.py decodes zlib's output at every level and strategy,
// stored / fixed / dynamic blocks, multi-member files and corrupted streams against zlib itself.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

namespace fqtk_host {

// A byte vector whose resize() does not zero what it adds (buffers that are about to be overwritten anyway).
template <typename T>
struct DefaultInitAllocator : std::allocator<T> {
    template <typename U> struct rebind { using other = DefaultInitAllocator<U>; };
    using std::allocator<T>::allocator;
    template <typename U> void construct(U *p) noexcept(std::is_nothrow_default_constructible<U>::value) { ::new (static_cast<void *>(p)) U; }
    template <typename U, typename... A> void construct(U *p, A &&...a) { ::new (static_cast<void *>(p)) U(std::forward<A>(a)...); }
};
using ByteVec = std::vector<char, DefaultInitAllocator<char>>;

class FastInflate {
  public:
    // crc(seed, data, n) -> running CRC-32 (zlib's or libdeflate's)
    using CrcFn = uint32_t (*)(uint32_t, const void *, size_t);

    // `data` must stay valid (a mapping) until the last next().
    void open(const uint8_t *data, size_t n, CrcFn crc, bool with_output = true) {
        data_ = data;
        data_end_ = data + n;
        in_end_ = data + n;
        ip_ = data;
        crc_fn_ = crc;
        if (with_output) obuf_.assign(kWindow + kPiece + kSlack, 0);
        hist_ = 0;
        state_ = State::Header;
        bb_ = 0;
        bc_ = 0;
        tail_active_ = false;
        members_ = 0;
    }

    // Next piece of output (valid until the following call).  *n == 0 with true: clean end of the file.
    // false: corrupt / truncated stream, *err says what.
    bool next(const uint8_t **out, size_t *n, std::string *err) {
        for (;;) {
            // slide the window: the last 32 KiB of what exists stay in front of the new piece
            uint8_t *base = obuf_.data();
            if (hist_ > kWindow) {
                std::memmove(base, base + hist_ - kWindow, kWindow);
                hist_ = kWindow;
            }
            uint8_t *const piece = base + hist_;
            uint8_t *op = piece;
            uint8_t *const stop = piece + kPiece;   // a symbol that starts before it may run kSlack bytes past
            bool member_done = false;
            bool block_done = false;
            while (op < stop && state_ != State::Done && !member_done && !block_done) {
                const State before = state_;
                switch (state_) {
                    case State::Header:
                        if (!parse_header(err)) return false;
                        break;
                    case State::BlockStart:
                        if (!start_block(err)) return false;
                        break;
                    case State::Stored: {
                        const size_t room = (size_t)(stop - op);
                        const size_t take_n = stored_left_ < room ? stored_left_ : room;
                        if (in_end_ - ip_ < (ptrdiff_t)take_n) return fail(err, "stored block runs past the end of the file");
                        std::memcpy(op, ip_, take_n);
                        op += take_n;
                        ip_ += take_n;
                        stored_left_ -= take_n;
                        if (stored_left_ == 0) state_ = final_block_ ? State::Trailer : State::BlockStart;
                        break;
                    }
                    case State::Codes:
                        if (!decode(base, op, stop, err)) return false;
                        break;
                    case State::Trailer:
                        member_done = true;   // the member's CRC wants this piece first
                        break;
                    case State::Done:
                        break;
                }
                if (stop_at_block_end_ && (before == State::Codes || before == State::Stored) && state_ == State::BlockStart)
                    block_done = true;
            }
            const size_t got = (size_t)(op - piece);
            if (got) {
                crc_ = crc_fn_(crc_, piece, got);
                isize_ += (uint32_t)got;
            }
            hist_ += got;
            if (member_done) {
                if (!check_trailer(err)) return false;
                if (got == 0 && !stop_at_block_end_) continue;   // nothing in hand: on to the next member, or the end
            }
            if (got == 0 && block_done) { *out = piece; *n = 0; at_boundary_ = true; return true; }
            at_boundary_ = block_done || member_done;
            *out = piece;
            *n = got;
            return true;
        }
    }

    static constexpr size_t kPiece = 4u << 20;

  protected:   // (parallel_gunzip.hpp builds its speculative decoder on these)
    static constexpr size_t kWindow = 32768, kSlack = 258 + 64;
    static constexpr int kLitBits = 11, kDistBits = 8;
    // table entry: bits 0-7 code bits to consume | bits 8-12 extra-bit count (or second-level index bits)
    //              | flags | bits 16-31 literal / base value / second-level offset
    static constexpr uint32_t kLit = 1u << 15, kSub = 1u << 14, kEob = 1u << 13;

    enum class State { Header, BlockStart, Stored, Codes, Trailer, Done };

    static bool fail(std::string *err, const char *what) {
        *err = std::string("corrupt gzip stream: ") + what;
        return false;
    }

    // ---- bit input ------------------------------------------------------------------------------------------
    // The last bytes of the file are decoded from a zero-padded copy, so that the 8-byte refill never reads past
    // the mapping; `ip_` then points into tail_.
    // The bit buffer may still hold up to 7 bytes fetched before `ip_` (align_to_byte() hands them back), so the
    // copy keeps kTailBack bytes of history in front of the switch point.
    void guard_tail() {
        if (tail_active_ || in_end_ - ip_ >= (ptrdiff_t)kTailAt) return;
        const size_t left = (size_t)(in_end_ - ip_);
        const size_t have = (size_t)(ip_ - data_);
        const size_t back = have < kTailBack ? have : kTailBack;
        std::memset(tail_, 0, sizeof tail_);
        std::memcpy(tail_ + kTailBack - back, ip_ - back, left + back);
        tail_origin_off_ = (ptrdiff_t)(ip_ - data_) - (ptrdiff_t)kTailBack;   // where tail_[0] sits in the mapping
        ip_ = tail_ + kTailBack;
        in_end_ = tail_ + kTailBack + left;
        tail_active_ = true;
    }
    static uint64_t load64(const uint8_t *p) {
        uint64_t v;
        std::memcpy(&v, p, 8);
        return v;   // little-endian host (x86-64)
    }
    void refill() {
        bb_ |= load64(ip_) << bc_;
        ip_ += (63 - bc_) >> 3;
        bc_ |= 56;
    }
    uint32_t take(unsigned nbits) {
        const uint32_t v = (uint32_t)(bb_ & ((1ull << nbits) - 1));
        bb_ >>= nbits;
        bc_ -= nbits;
        return v;
    }
    // Back to whole bytes: drops the bits of the current byte and returns the look-ahead to the input.
    void align_to_byte() {
        const unsigned drop = bc_ & 7;
        bb_ >>= drop;
        bc_ -= drop;
        ip_ -= bc_ >> 3;
        bb_ = 0;
        bc_ = 0;
    }
    // the next unconsumed bit lies past the end of the file
    bool overrun() const { return ip_ - (bc_ >> 3) > in_end_; }
    // refill for the block headers: never runs off the padded copy of the file's last bytes
    bool safe_refill(std::string *err) {
        guard_tail();
        if (tail_active_ && ip_ > in_end_ + 8) return fail(err, "stream runs past the end of the file");
        refill();
        return true;
    }

    // Position of the next unconsumed bit, in bits from the start of the mapping.
    uint64_t bit_pos() const {
        const ptrdiff_t off = tail_active_ ? tail_origin_off_ + (ip_ - tail_) : ip_ - data_;
        return (uint64_t)off * 8u - bc_;
    }
    // Continue at bit `pos` of the mapping (a block boundary).
    bool seek_bit(uint64_t pos, std::string *err) {
        ip_ = data_ + (pos >> 3);
        in_end_ = data_end_;
        tail_active_ = false;
        bb_ = 0;
        bc_ = 0;
        if (ip_ > data_end_) return fail(err, "position past the end of the file");
        if (!safe_refill(err)) return false;
        take((unsigned)(pos & 7u));
        return true;
    }

    // ---- gzip framing (RFC 1952) ------------------------------------------------------------------------------
    bool parse_header(std::string *err) {
        guard_tail();
        if (ip_ == in_end_) {
            if (members_ == 0) return fail(err, "empty file");
            state_ = State::Done;
            return true;
        }
        const uint8_t *p = ip_;
        auto need = [&](size_t k) { return (size_t)(in_end_ - p) >= k; };
        if (!need(10)) return fail(err, "truncated header");
        if (p[0] != 0x1f || p[1] != 0x8b) {
            if (members_ > 0) {   // trailing garbage after the last member: zlib's gzread ignores it
                state_ = State::Done;
                return true;
            }
            return fail(err, "not a gzip file");
        }
        if (p[2] != 8) return fail(err, "unknown compression method");
        const uint8_t flg = p[3];
        if (flg & 0xE0) return fail(err, "reserved header flags set");
        p += 10;
        if (flg & 4) {
            if (!need(2)) return fail(err, "truncated header");
            const size_t xlen = p[0] | (p[1] << 8);
            p += 2;
            if (!need(xlen)) return fail(err, "truncated header");
            p += xlen;
        }
        for (int f = 8; f <= 16; f <<= 1) {   // FNAME, FCOMMENT: zero-terminated
            if (!(flg & f)) continue;
            const void *z = std::memchr(p, 0, (size_t)(in_end_ - p));
            if (!z) return fail(err, "truncated header");
            p = static_cast<const uint8_t *>(z) + 1;
        }
        if (flg & 2) {
            if (!need(2)) return fail(err, "truncated header");
            p += 2;
        }
        ip_ = p;
        ++members_;
        crc_ = 0;
        isize_ = 0;
        bb_ = 0;
        bc_ = 0;
        state_ = State::BlockStart;
        return true;
    }
    bool check_trailer(std::string *err) {
        align_to_byte();
        guard_tail();
        if (in_end_ - ip_ < 8) return fail(err, "truncated trailer");
        uint32_t want_crc, want_size;
        std::memcpy(&want_crc, ip_, 4);
        std::memcpy(&want_size, ip_ + 4, 4);
        ip_ += 8;
        if (want_crc != crc_) return fail(err, "CRC mismatch");
        if (want_size != isize_) return fail(err, "length mismatch");
        state_ = State::Header;
        return true;
    }

    // ---- blocks (RFC 1951 3.2.3) ------------------------------------------------------------------------------
    bool start_block(std::string *err) {
        if (!safe_refill(err)) return false;
        final_block_ = take(1) != 0;
        const uint32_t type = take(2);
        if (type == 0) {
            align_to_byte();
            guard_tail();
            if (in_end_ - ip_ < 4) return fail(err, "truncated stored block");
            const uint32_t len = ip_[0] | (ip_[1] << 8), nlen = ip_[2] | (ip_[3] << 8);
            if ((len ^ 0xFFFFu) != nlen) return fail(err, "stored block length check");
            ip_ += 4;
            stored_left_ = len;
            state_ = len ? State::Stored : (final_block_ ? State::Trailer : State::BlockStart);
            return true;
        }
        if (type == 1) {
            uint8_t lens[288 + 32];
            for (int i = 0; i < 144; ++i) lens[i] = 8;
            for (int i = 144; i < 256; ++i) lens[i] = 9;
            for (int i = 256; i < 280; ++i) lens[i] = 7;
            for (int i = 280; i < 288; ++i) lens[i] = 8;
            for (int i = 0; i < 32; ++i) lens[288 + i] = 5;
            if (!build(lens, 288, true, err) || !build(lens + 288, 32, false, err)) return false;
            state_ = State::Codes;
            return true;
        }
        if (type == 3) return fail(err, "reserved block type");
        // dynamic Huffman: code-length code, then the two length sets in one run
        const uint32_t hlit = take(5) + 257, hdist = take(5) + 1, hclen = take(4) + 4;
        if (hlit > 286 || hdist > 30) return fail(err, "too many length or distance symbols");
        static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        uint8_t cl[19] = {0};
        for (uint32_t i = 0; i < hclen; ++i) {
            if (bc_ < 3 && !safe_refill(err)) return false;
            cl[order[i]] = (uint8_t)take(3);
        }
        uint32_t pre[128];   // 7-bit table of the code-length code
        if (!build_small(cl, 19, pre, err)) return false;
        uint8_t lens[286 + 30 + 140];
        uint32_t i = 0;
        const uint32_t total = hlit + hdist;
        while (i < total) {
            if (!safe_refill(err)) return false;
            const uint32_t e = pre[bb_ & 127];
            if ((e & 0xFF) == 0) return fail(err, "invalid code-length code");
            take(e & 0xFF);
            const uint32_t sym = e >> 16;
            if (sym < 16) {
                lens[i++] = (uint8_t)sym;
            } else {
                uint32_t rep, v = 0;
                if (sym == 16) {
                    if (i == 0) return fail(err, "repeat with no previous length");
                    v = lens[i - 1];
                    rep = 3 + take(2);
                } else if (sym == 17) {
                    rep = 3 + take(3);
                } else {
                    rep = 11 + take(7);
                }
                if (i + rep > total) return fail(err, "length repeat overruns the set");
                std::memset(lens + i, (int)v, rep);
                i += rep;
            }
        }
        if (overrun()) return fail(err, "truncated block header");
        if (lens[256] == 0) return fail(err, "no end-of-block code");
        if (!build(lens, hlit, true, err) || !build(lens + hlit, hdist, false, err)) return false;
        state_ = State::Codes;
        return true;
    }

    // Canonical code -> one-level table of 2^7 entries (the code-length code: <= 7 bits, 19 symbols).
    static bool build_small(const uint8_t *lens, int n, uint32_t *tab, std::string *err) {
        int count[8] = {0};
        for (int s = 0; s < n; ++s) count[lens[s]]++;
        count[0] = 0;
        int left = 1;
        for (int l = 1; l <= 7; ++l) {
            left = (left << 1) - count[l];
            if (left < 0) return fail(err, "over-subscribed code-length code");
        }
        for (int k = 0; k < 128; ++k) tab[k] = 0;
        uint32_t code = 0;
        for (int l = 1; l <= 7; ++l) {
            for (int s = 0; s < n; ++s) {
                if (lens[s] != l) continue;
                uint32_t rev = 0;
                for (int b = 0; b < l; ++b) rev |= ((code >> b) & 1u) << (l - 1 - b);
                for (uint32_t k = rev; k < 128; k += 1u << l) tab[k] = ((uint32_t)s << 16) | (uint32_t)l;
                ++code;
            }
            code <<= 1;
        }
        return true;   // incomplete sets: unused entries stay 0 = invalid, caught by the reader
    }

    // Canonical code -> two-level table.  litlen: symbols become literal / end-of-block / length entries.
    bool build(const uint8_t *lens, int n, bool litlen, std::string *err) {
        static const uint16_t len_base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
        static const uint8_t len_extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
        static const uint16_t dist_base[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
        static const uint8_t dist_extra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
        const int P = litlen ? kLitBits : kDistBits;
        uint32_t *tab = litlen ? lit_ : dist_;
        const uint32_t cap = litlen ? kLitCap : kDistCap;
        int count[16] = {0};
        for (int s = 0; s < n; ++s) count[lens[s]]++;
        count[0] = 0;
        int left = 1, maxlen = 0, used = 0;
        for (int l = 1; l <= 15; ++l) {
            left = (left << 1) - count[l];
            if (left < 0) return fail(err, "over-subscribed Huffman code");
            if (count[l]) maxlen = l;
            used += count[l];
        }
        // incomplete sets: legal only as "one code of one bit" (a single distance, or none at all)
        if (left > 0 && !(used <= 1 && maxlen <= 1)) return fail(err, "incomplete Huffman code");
        for (uint32_t k = 0; k < (1u << P); ++k) tab[k] = 0;   // 0 bits to consume = invalid
        uint32_t next_code[16], code = 0;
        for (int l = 1; l <= 15; ++l) { code = (code + (uint32_t)count[l - 1]) << 1; next_code[l] = code; }
        auto entry_of = [&](int s, int bits) -> uint32_t {   // 0 = a code that must not occur (fixed code: 286, 287; 30, 31)
            if (litlen ? s >= 286 : s >= 30) return 0;
            if (!litlen) return ((uint32_t)dist_base[s] << 16) | ((uint32_t)dist_extra[s] << 8) | (uint32_t)bits;
            if (s < 256) return ((uint32_t)s << 16) | kLit | (1u << 8) | (uint32_t)bits;   // one literal
            if (s == 256) return kEob | (uint32_t)bits;
            return ((uint32_t)len_base[s - 257] << 16) | ((uint32_t)len_extra[s - 257] << 8) | (uint32_t)bits;
        };
        // second-level tables: how many index bits each first-level prefix needs
        uint8_t sub_bits[1 << kLitBits];
        std::memset(sub_bits, 0, (size_t)1 << P);
        struct Long { uint16_t rev; uint8_t len; uint16_t sym; };
        Long longs[288];
        int n_long = 0;
        for (int s = 0; s < n; ++s) {
            const int l = lens[s];
            if (!l) continue;
            const uint32_t c = next_code[l]++;
            uint32_t rev = 0;
            for (int b = 0; b < l; ++b) rev |= ((c >> b) & 1u) << (l - 1 - b);
            if (l <= P) {
                const uint32_t e = entry_of(s, l);
                for (uint32_t k = rev; k < (1u << P); k += 1u << l) tab[k] = e;
            } else {
                const uint32_t prefix = rev & ((1u << P) - 1);
                if (l - P > sub_bits[prefix]) sub_bits[prefix] = (uint8_t)(l - P);
                longs[n_long++] = Long{(uint16_t)rev, (uint8_t)l, (uint16_t)s};
            }
        }
        uint32_t top = 1u << P;
        for (uint32_t prefix = 0; prefix < (1u << P); ++prefix) {
            if (!sub_bits[prefix]) continue;
            const uint32_t size = 1u << sub_bits[prefix];
            if (top + size > cap) return fail(err, "Huffman table overflow");
            tab[prefix] = (top << 16) | kSub | ((uint32_t)sub_bits[prefix] << 8) | (uint32_t)P;
            for (uint32_t k = 0; k < size; ++k) tab[top + k] = 0;
            top += size;
        }
        if (litlen) {
            // Two literals per look-up where both codes fit the index: FASTQ bases have 2-3-bit codes, and the
            // loop's cost is the serial chain look-up -> shift -> look-up, not the bytes.
            uint32_t single[1 << kLitBits];
            std::memcpy(single, tab, sizeof single);
            for (uint32_t k = 0; k < (1u << P); ++k) {
                const uint32_t e1 = single[k];
                if (!(e1 & kLit)) continue;
                const uint32_t l1 = e1 & 0xFF;
                const uint32_t e2 = single[k >> l1];   // the index bits that follow, unknown high bits as zeros
                if (!(e2 & kLit) || (e2 & 0xFF) + l1 > (uint32_t)P) continue;   // only if those bits hold ALL of the code
                tab[k] = (e1 & 0x00FF0000u) | ((e2 & 0x00FF0000u) << 8) | kLit | (2u << 8) | (l1 + (e2 & 0xFF));
            }
        }
        for (int q = 0; q < n_long; ++q) {
            const Long &g = longs[q];
            const uint32_t prefix = g.rev & ((1u << P) - 1);
            const uint32_t off = tab[prefix] >> 16, sb = (tab[prefix] >> 8) & 31u;
            const uint32_t e = entry_of(g.sym, g.len - P);
            for (uint32_t k = (uint32_t)g.rev >> P; k < (1u << sb); k += 1u << (g.len - P)) tab[off + k] = e;
        }
        return true;
    }

    // ---- the loop ---------------------------------------------------------------------------------------------
    // Decodes symbols of the current block until `stop` is reached or the block ends.  Two copies of the loop: the
    // one for CPUs with BMI2 lets the compiler use shrx / shlx / bzhi for the variable shifts and masks (x86's
    // shift-by-CL costs three micro-ops); picked once per process.
    bool decode(uint8_t *base, uint8_t *&op_ref, uint8_t *stop, std::string *err) {
#if defined(__x86_64__) && defined(__GNUC__)
        static const bool bmi2 = __builtin_cpu_supports("bmi2") && !env_flag_off_bmi2();
        if (bmi2) return decode_bmi2(base, op_ref, stop, err);
#endif
        return decode_loop(base, op_ref, stop, err);
    }
    static bool env_flag_off_bmi2() { const char *v = std::getenv("FQTK_NO_BMI2"); return v && *v && !(v[0] == '0' && v[1] == 0); }
#if defined(__x86_64__) && defined(__GNUC__)
    __attribute__((target("bmi2"))) bool decode_bmi2(uint8_t *base, uint8_t *&op_ref, uint8_t *stop, std::string *err) {
        return decode_loop(base, op_ref, stop, err);
    }
#endif
    __attribute__((always_inline)) inline bool decode_loop(uint8_t *base, uint8_t *&op_ref, uint8_t *stop, std::string *err) {
        uint8_t *op = op_ref;
        uint64_t bb = bb_;
        unsigned bc = bc_;
        const uint8_t *ip = ip_;
        const uint32_t *const lit = lit_, *const dist = dist_;
        const char *bad = nullptr;
#define FQTK_REFILL() do { bb |= load64(ip) << bc; ip += (63 - bc) >> 3; bc |= 56; } while (0)
#define FQTK_DROP(n) do { bb >>= (n); bc -= (n); } while (0)
        // a literal entry: both bytes are stored, one or two of them count
#define FQTK_PUT(e) do { const uint16_t two = (uint16_t)((e) >> 16); std::memcpy(op, &two, 2); op += ((e) >> 8) & 3u; FQTK_DROP((e) & 0xFF); } while (0)
        // input guard: the mapping up to 64 bytes before its end, then the zero-padded copy of those bytes
        const uint8_t *in_limit = tail_active_ ? in_end_ + 8 : in_end_ - kTailAt;
        for (;;) {
            if (op >= stop) break;
            if (ip > in_limit) {
                if (tail_active_) { bad = "stream runs past the end of the file"; break; }
                ip_ = ip;
                guard_tail();
                ip = ip_;
                in_limit = in_end_ + 8;
            }
            FQTK_REFILL();
            uint32_t e = lit[bb & ((1u << kLitBits) - 1)];
            if (e & kLit) {   // up to three look-ups of one or two literals on one refill (<= 45 of >= 56 bits)
                FQTK_PUT(e);
                e = lit[bb & ((1u << kLitBits) - 1)];
                if (e & kLit) {
                    FQTK_PUT(e);
                    e = lit[bb & ((1u << kLitBits) - 1)];
                    if (e & kLit) {
                        FQTK_PUT(e);
                        continue;
                    }
                }
                FQTK_REFILL();
            }
            if (e & kSub) {
                FQTK_DROP(kLitBits);
                e = lit[(e >> 16) + (bb & ((1u << ((e >> 8) & 31u)) - 1))];
                if (e & kLit) {
                    FQTK_PUT(e);
                    continue;
                }
            }
            if ((e & 0xFF) == 0) { bad = "invalid literal/length code"; break; }
            FQTK_DROP(e & 0xFF);
            if (e & kEob) {
                state_ = final_block_ ? State::Trailer : State::BlockStart;
                break;
            }
            // a match: length = base + extra bits, then the distance code (<= 48 bits in all: one refill covers it)
            const unsigned lx = (e >> 8) & 31u;
            const uint32_t len = (e >> 16) + (uint32_t)(bb & ((1ull << lx) - 1));
            FQTK_DROP(lx);
            uint32_t d = dist[bb & ((1u << kDistBits) - 1)];
            if (d & kSub) {
                FQTK_DROP(kDistBits);
                d = dist[(d >> 16) + (bb & ((1u << ((d >> 8) & 31u)) - 1))];
            }
            if ((d & 0xFF) == 0) { bad = "invalid distance code"; break; }
            FQTK_DROP(d & 0xFF);
            const unsigned dx = (d >> 8) & 31u;
            const uint32_t distance = (d >> 16) + (uint32_t)(bb & ((1ull << dx) - 1));
            FQTK_DROP(dx);
            if (distance > (size_t)(op - base)) { bad = "distance reaches before the start of the data"; break; }
            const uint8_t *src = op - distance;
            uint8_t *const end = op + len;
            if (distance >= 8) {
                // sixteen bytes without a branch (most matches of sequence data are shorter), the rest in a loop
                std::memcpy(op, src, 8);
                std::memcpy(op + 8, src + 8, 8);
                if (len > 16) {
                    op += 16;
                    src += 16;
                    do {
                        std::memcpy(op, src, 8);
                        std::memcpy(op + 8, src + 8, 8);
                        op += 16;
                        src += 16;
                    } while (op < end);
                }
            } else if (distance == 1) {
                std::memset(op, *src, len);
            } else {
                do { *op++ = *src++; } while (op < end);
            }
            op = end;
        }
#undef FQTK_REFILL
#undef FQTK_DROP
#undef FQTK_PUT
        op_ref = op;
        bb_ = bb;
        bc_ = bc;
        ip_ = ip;
        if (bad) return fail(err, bad);
        if (overrun()) return fail(err, "stream runs past the end of the file");
        return true;
    }

    static constexpr size_t kTailAt = 64, kTailBack = 8;
    static constexpr uint32_t kLitCap = (1u << kLitBits) + 2048, kDistCap = (1u << kDistBits) + 1024;

    const uint8_t *in_end_ = nullptr, *ip_ = nullptr;
    const uint8_t *data_ = nullptr, *data_end_ = nullptr;   // the whole mapping
    ptrdiff_t tail_origin_off_ = 0;                          // offset of tail_[0] in the mapping (may be negative)
    bool at_boundary_ = false;                               // the last next() ended exactly at a block / member end
    bool stop_at_block_end_ = false;                         // next() hands back control after every block
    CrcFn crc_fn_ = nullptr;
    std::vector<uint8_t> obuf_;
    size_t hist_ = 0;          // bytes of obuf_ that hold output already handed out (<= window after the slide)
    State state_ = State::Done;
    uint64_t bb_ = 0;
    unsigned bc_ = 0;
    bool final_block_ = false, tail_active_ = false;
    size_t stored_left_ = 0;
    uint32_t crc_ = 0, isize_ = 0;
    uint64_t members_ = 0;
    uint8_t tail_[kTailBack + kTailAt + 64];
    uint32_t lit_[kLitCap], dist_[kDistCap];
};

}  // namespace fqtk_host
