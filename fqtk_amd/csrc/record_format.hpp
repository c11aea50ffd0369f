// record_format.hpp -- the demultiplexed FASTQ record, as a list of pieces.
//
// SURVEY.md section 8(f) row 2.  The reference formats every output record on the host, one at a time:
// ReadSet::write_header_internal (/root/reference/src/bin/commands/demux.rs:171-267) rewrites the header and
// SampleWriters::write (:396-415) appends "\n<bases>\n+\n<quals>\n".  On the MI355X path the record is formatted
// where the matcher's result already is -- in HBM -- so this file states the record ONCE, as an ordered list of
// pieces (a span of some input's text, or up to eight literal bytes), for both users:
//   * the device: a sizing pass adds the piece lengths up (the prefix sums over them place every record in its
//     output file), the formatting pass copies the pieces with all 64 lanes of a wavefront (demux_kernels.hip.h);
//   * the CPU test-suite, which runs the same functions through libfqtk_host.so against host/header.hpp and the
//     reference's own header vectors (demux.rs:2084-2196).
// Plain C++17, no allocation, no library calls: compiles under hipcc for the device and under g++.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define FQTK_HD __host__ __device__
#else
#ifndef FQTK_HD
#define FQTK_HD
#endif
#endif

namespace fqtk {
namespace fmt {

// Why a record cannot be formatted (the reference returns an error / panics at the same places).
enum HeaderError : uint8_t {
    kHeaderOk = 0,
    kTooManyNameSegments = 1,   // "Can't handle read name with more than 8 segments"   demux.rs:197-199
    kEmptyComment = 2,          // comment present but empty: the reference unwraps chars.last()    :225
    kCommentNot4Segments = 3,   // more than three ':' in the comment                                 :234
    kMalformedComment = 4,      // nothing left of the comment after its first field                 :252
};

// What write_header_internal decides from the input header alone (the header of the FIRST input, :126-139).
struct HeaderPlan {
    uint32_t name_len;    // bytes before the first space (the whole header when there is none)
    uint32_t copy_off;    // part of the comment that is copied: all of it (kind 1) or what follows its first ':' (kind 2)
    uint32_t copy_len;
    uint8_t kind;         // 0: no comment -> "<n>:N:0:"   1: fewer than 3 colons -> comment as it is   2: "<n>:" + rest
    uint8_t tail;         // byte appended after the copied part (':' or '+'), 0 = none
    uint8_t msep;         // what joins the name and the first molecular barcode: ':' or '+' (8th name field = a UMI)
    uint8_t err;          // HeaderError
};

FQTK_HD inline HeaderPlan plan_header(const uint8_t *h, uint32_t len, bool have_molecular) {
    HeaderPlan p;
    p.name_len = len;
    p.copy_off = 0;
    p.copy_len = 0;
    p.kind = 0;
    p.tail = 0;
    p.msep = ':';
    p.err = kHeaderOk;
    // The line's first space, the colons in front of it, the colons behind it and where the first of those stands.
    uint32_t sp = len, name_colons = 0, colons = 0, first_at = len;
#if defined(__HIP_DEVICE_COMPILE__)
    // (Sixteen bytes per round, four unaligned dword reads in flight -- every text buffer has 64 bytes of slack behind it --, the bytes looked at
    //  in order by selection.  Byte by byte with an early exit, each byte was a dependent round trip to the L2: some fifty-five of them per
    //  template, and k_plan_rank's 150 us per chunk were mostly this.)
    for (uint32_t base = 0; base < len; base += 16u) {
        uint32_t w[4];
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k) w[k] = *reinterpret_cast<const uint32_t *>(h + base + 4u * k);
#pragma unroll
        for (uint32_t j = 0; j < 16u; ++j) {
            const uint32_t c = (w[j >> 2] >> (8u * (j & 3u))) & 0xFFu, pos = base + j;
            const bool in = pos < len, before_space = sp == len;
            const bool colon = in && c == ':';
            name_colons += colon && before_space ? 1u : 0u;
            first_at = colon && !before_space && first_at == len ? pos : first_at;
            colons += colon && !before_space ? 1u : 0u;
            sp = in && before_space && c == ' ' ? pos : sp;
        }
    }
#else
    for (uint32_t i = 0; i < len; ++i)
        if (h[i] == ' ') { sp = i; break; }
    for (uint32_t i = 0; i < sp; ++i) name_colons += h[i] == ':';
    for (uint32_t i = sp + 1; i < len; ++i)
        if (h[i] == ':') { if (first_at == len) first_at = i; ++colons; }
#endif
    p.name_len = sp;
    if (have_molecular) {   // demux.rs:188-211
        if (name_colons > 7) { p.err = kTooManyNameSegments; return p; }
        p.msep = name_colons == 7 ? '+' : ':';
    }
    if (sp == len) return p;   // no comment: "<n>:N:0:"
    const uint32_t c0 = sp + 1, clen = len - c0;
    if (clen == 0) { p.err = kEmptyComment; return p; }
    const uint32_t first = first_at == len ? clen : first_at - c0;
    const uint8_t last = h[len - 1];
    if (colons < 3) {   // demux.rs:227-232
        p.kind = 1;
        p.copy_off = c0;
        p.copy_len = clen;
        p.tail = last != ':' ? ':' : 0;
        return p;
    }
    if (colons != 3) { p.err = kCommentNot4Segments; return p; }
    // Illumina can place a "0" in the index position of unmatched FASTQs: a trailing digit goes (demux.rs:241-246)
    const uint32_t drop = (last >= '0' && last <= '9') ? 1u : 0u;
    p.kind = 2;
    p.copy_off = c0 + first + 1;
    p.copy_len = clen - (first + 1) - drop;
    if (p.copy_len == 0) { p.err = kMalformedComment; return p; }
    p.tail = h[p.copy_off + p.copy_len - 1] != ':' ? '+' : 0;
    return p;
}

// A span of an input's text.
struct Span { uint32_t input, off, len; };

// What one output file takes from a template: which segment, and the read number its header carries.
struct FileSeg { uint32_t input; uint32_t offset; int32_t length; uint32_t read_num; };
// A segment of a read structure by position: [offset, offset + length), length < 0 = to the end of the read.
struct SegPos { uint32_t input; uint32_t offset; int32_t length; };

FQTK_HD inline void segment_span(uint32_t offset, int32_t length, uint32_t read_len, uint32_t *lo, uint32_t *hi) {
    uint32_t e = length >= 0 ? offset + (uint32_t)length : read_len;
    if (e > read_len) e = read_len;
    *hi = e;
    *lo = offset > e ? e : offset;
}

FQTK_HD inline uint32_t decimal_digits(uint32_t v, uint8_t *out /* >= 10 bytes */) {
    uint8_t tmp[10];
    uint32_t n = 0;
    do { tmp[n++] = (uint8_t)('0' + v % 10u); v /= 10u; } while (v);
    for (uint32_t i = 0; i < n; ++i) out[i] = tmp[n - 1 - i];
    return n;
}

// Emits the record of one output file as pieces, in order.  Sink: lit(byte), span(input, off, len).
// `head`: the first input's header (input 0, bytes [head_off, head_off + head_len) of its text, without '@').
template <typename Sink>
FQTK_HD inline void emit_record(Sink &s, const HeaderPlan &p, uint32_t head_off, uint32_t read_num,
                                const Span *bsegs, uint32_t nb, const Span *msegs, uint32_t nm,
                                const Span &bases, const Span &quals) {
    s.lit('@');
    s.span(0, head_off, p.name_len);
    if (nm) {
        s.lit(p.msep);
        for (uint32_t i = 0; i < nm; ++i) {
            if (i) s.lit('+');
            s.span(msegs[i].input, msegs[i].off, msegs[i].len);
        }
    }
    s.lit(' ');
    if (p.kind != 1) {
        uint8_t d[10];
        const uint32_t nd = decimal_digits(read_num, d);
        for (uint32_t i = 0; i < nd; ++i) s.lit(d[i]);
        s.lit(':');
        if (p.kind == 0) { s.lit('N'); s.lit(':'); s.lit('0'); s.lit(':'); }
    }
    if (p.kind != 0) {
        s.span(0, head_off + p.copy_off, p.copy_len);
        if (p.tail) s.lit(p.tail);
    }
    for (uint32_t i = 0; i < nb; ++i) {
        if (i) s.lit('+');
        s.span(bsegs[i].input, bsegs[i].off, bsegs[i].len);
    }
    s.lit('\n');
    s.span(bases.input, bases.off, bases.len);
    s.lit('\n');
    s.lit('+');
    s.lit('\n');
    s.span(quals.input, quals.off, quals.len);
    s.lit('\n');
}

// The same record as a TABLE OF SLOTS in fixed positions (empty ones have length 0), so that slot s can be worked out on
// its own -- by lane s of a wavefront, all slots at once (k_format) -- where emit_record walks the record from its start:
//   0 '@' | 1 name | 2 msep | 3 + 2i ['+'] , 4 + 2i molecular segment i | b1 ' ' | b1+1 "<n>:" or "<n>:N:0:" | b1+2 copied
//   part of the comment | b1+3 tail | b2 + 2i ['+'] , b2 + 2i + 1 sample segment i | b3 '\n' | bases | "\n+\n" | quals | '\n'
// with b1 = 3 + 2 nm, b2 = b1 + 4, b3 = b2 + 2 nb.  A literal slot holds up to four bytes in `lit`; the number slot (kNumber)
// refers to the caller's digits (they depend on the file, not on the template).  Checked against emit_record byte for byte
// in the CPU tests (fqtk_host_format_record).
enum SlotKind : uint32_t { kSpan = 0, kLiteral = 1, kNumber = 2 };
struct Slot { uint32_t len, kind, input, off, lit; };
FQTK_HD inline uint32_t record_slots(uint32_t nb, uint32_t nm) { return 12u + 2u * (nb + nm); }
// "<n>:" (kind 2) / "<n>:N:0:" (kind 0) / nothing (kind 1) as up to 16 bytes in four words; returns the length
FQTK_HD inline uint32_t number_literal(uint32_t read_num, uint32_t header_kind, uint32_t (&w)[4]) {
    w[0] = w[1] = w[2] = w[3] = 0;
    if (header_kind == 1) return 0;
    uint8_t d[16];
    uint32_t n = decimal_digits(read_num, d);
    d[n++] = ':';
    if (header_kind == 0) { d[n++] = 'N'; d[n++] = ':'; d[n++] = '0'; d[n++] = ':'; }
    for (uint32_t i = 0; i < n; ++i) w[i >> 2] |= (uint32_t)d[i] << (8 * (i & 3u));
    return n;
}
// (bseg_of(i) / mseg_of(i): the span of sample / molecular barcode segment i -- from an array, or worked out where it is asked for:
//  k_format's lanes each take one slot of one of sixteen records and have no room for sixteen arrays of spans)
template <typename BFn, typename MFn>
FQTK_HD inline Slot record_slot_with(uint32_t s, const HeaderPlan &p, uint32_t head_off, uint32_t number_len, BFn bseg_of, uint32_t nb, MFn mseg_of, uint32_t nm,
                                     const Span &bases, const Span &quals) {
    Slot z;
    z.len = 0; z.kind = kLiteral; z.input = 0; z.off = 0; z.lit = 0;
    auto lit1 = [&](uint32_t byte, bool on) { z.kind = kLiteral; z.lit = byte; z.len = on ? 1u : 0u; };
    auto span = [&](uint32_t input, uint32_t off, uint32_t len) { z.kind = kSpan; z.input = input; z.off = off; z.len = len; };
    const uint32_t b1 = 3u + 2u * nm, b2 = b1 + 4u, b3 = b2 + 2u * nb;
    if (s == 0) lit1('@', true);
    else if (s == 1) span(0, head_off, p.name_len);
    else if (s == 2) lit1(p.msep, nm != 0);
    else if (s < b1) {
        const uint32_t i = (s - 3u) >> 1;
        if (((s - 3u) & 1u) == 0) lit1('+', i != 0); else { const Span m = mseg_of(i); span(m.input, m.off, m.len); }
    } else if (s == b1) lit1(' ', true);
    else if (s == b1 + 1u) { z.kind = kNumber; z.len = p.kind != 1 ? number_len : 0u; }
    else if (s == b1 + 2u) span(0, head_off + p.copy_off, p.kind != 0 ? p.copy_len : 0u);
    else if (s == b1 + 3u) lit1(p.tail, p.kind != 0 && p.tail != 0);
    else if (s < b3) {
        const uint32_t i = (s - b2) >> 1;
        if (((s - b2) & 1u) == 0) lit1('+', i != 0); else { const Span b = bseg_of(i); span(b.input, b.off, b.len); }
    } else if (s == b3) lit1('\n', true);
    else if (s == b3 + 1u) span(bases.input, bases.off, bases.len);
    else if (s == b3 + 2u) { z.kind = kLiteral; z.lit = (uint32_t)'\n' | ((uint32_t)'+' << 8) | ((uint32_t)'\n' << 16); z.len = 3; }
    else if (s == b3 + 3u) span(quals.input, quals.off, quals.len);
    else if (s == b3 + 4u) lit1('\n', true);
    return z;
}
FQTK_HD inline Slot record_slot(uint32_t s, const HeaderPlan &p, uint32_t head_off, uint32_t number_len,
                                const Span *bsegs, uint32_t nb, const Span *msegs, uint32_t nm, const Span &bases, const Span &quals) {
    return record_slot_with(s, p, head_off, number_len, [&](uint32_t i) { return bsegs[i]; }, nb, [&](uint32_t i) { return msegs[i]; }, nm, bases, quals);
}

// Length of that record from sums alone (the placement kernel sizes every record of a chunk before any is written):
// `digits` of the read number, `bl` / `ml` bases in the nb sample / nm molecular barcode segments, `seg` bases in the
// file's own segment.  Checked against the sizing sink in the CPU tests.
FQTK_HD inline uint32_t record_len(const HeaderPlan &p, uint32_t digits, uint32_t bl, uint32_t nb, uint32_t ml, uint32_t nm, uint32_t seg) {
    uint32_t len = 1u + p.name_len + 1u;                                    // '@' name ' '
    if (nm) len += 1u + ml + (nm - 1u);                                     // sep M1+M2..
    if (p.kind == 0) len += digits + 5u;                                    // "<n>:N:0:"
    else if (p.kind == 1) len += p.copy_len + (p.tail ? 1u : 0u);           // the comment as it is [:]
    else len += digits + 1u + p.copy_len + (p.tail ? 1u : 0u);              // "<n>:" rest [+]
    if (nb) len += bl + (nb - 1u);                                          // B1+B2..
    return len + 5u + 2u * seg;                                             // "\n" bases "\n+\n" quals "\n"
}

// Sizing sink.
struct LenSink {
    uint32_t n = 0;
    FQTK_HD void lit(uint8_t) { ++n; }
    FQTK_HD void span(uint32_t, uint32_t, uint32_t len) { n += len; }
};

// Piece list: spans as they are, runs of literals packed eight to a piece.
constexpr int kMaxPieces = 64;
struct Piece {
    uint64_t lit;        // literal bytes, first byte lowest (is_lit)
    uint32_t off, len;   // span: offset in its input's text
    uint16_t input;
    uint16_t is_lit;
};
struct PieceSink {
    Piece *pc;
    uint32_t n = 0;
    FQTK_HD explicit PieceSink(Piece *p) : pc(p) {}
    FQTK_HD void lit(uint8_t b) {
        if (n && pc[n - 1].is_lit && pc[n - 1].len < 8) {
            pc[n - 1].lit |= (uint64_t)b << (8 * pc[n - 1].len);
            ++pc[n - 1].len;
            return;
        }
        pc[n].lit = b;
        pc[n].off = 0;
        pc[n].len = 1;
        pc[n].input = 0;
        pc[n].is_lit = 1;
        ++n;
    }
    FQTK_HD void span(uint32_t input, uint32_t off, uint32_t len) {
        if (!len) return;
        pc[n].lit = 0;
        pc[n].off = off;
        pc[n].len = len;
        pc[n].input = (uint16_t)input;
        pc[n].is_lit = 0;
        ++n;
    }
};
// Upper bound of the pieces of one record (every literal its own run at worst between spans).
FQTK_HD inline uint32_t max_pieces(uint32_t nb, uint32_t nm) { return 2u * (nb + nm) + 14u; }

}  // namespace fmt
}  // namespace fqtk
