// host_capi.cpp -- tiny C shim over the host-side components so the CPU test-suite can exercise them
// through ctypes (no GPU needed): read structures, header rewriting, FASTQ parsing, BGZF, metrics, and
// the host-side planners of the LDS-resident memo (csrc/lds_memo_plan.hpp) and of the direct-indexed memo
// (csrc/direct_memo_plan.hpp).
#include <chrono>
#include <cstring>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

#include "bgzf.hpp"
#include "bgzf_walk.hpp"
#include "chunk_dispatch.hpp"
#include "chunk_schedule.hpp"
#include "fastq_io.hpp"
#include "parallel_gunzip.hpp"
#include "header.hpp"
#include "metrics.hpp"
#include "read_structure.hpp"
#include "samples.hpp"
#include "../lds_memo_plan.hpp"
#include "../direct_memo_plan.hpp"
#include "../bgzf_deflate.hpp"
#include "../bgzf_inflate.hpp"
#include "wave_emu.hpp"
#include "region_inflate.hpp"
#include "parallel_gunzip.hpp"
#include "../record_format.hpp"

using namespace fqtk_host;

static int put(const std::string &s, char *out, size_t cap) {
    if (s.size() + 1 > cap) return -2;
    std::memcpy(out, s.c_str(), s.size() + 1);
    return 0;
}

extern "C" {

// Parses `text`; writes the canonical string, min length and per-segment (offset, length|-1, kind).
int fqtk_host_read_structure(const char *text, char *canon, size_t cap, uint64_t *min_len, int64_t *segs,
                             size_t max_segs, size_t *n_segs, char *err, size_t errcap) {
    ReadStructure rs;
    std::string e;
    if (!ReadStructure::parse(text, &rs, &e)) { put(e, err, errcap); return 1; }
    *min_len = rs.min_length();
    *n_segs = rs.segments.size();
    for (size_t i = 0; i < rs.segments.size() && i < max_segs; ++i) {
        segs[3 * i] = (int64_t)rs.segments[i].offset;
        segs[3 * i + 1] = rs.segments[i].length;
        segs[3 * i + 2] = (int64_t)(char)rs.segments[i].kind;
    }
    return put(rs.to_string(), canon, cap);
}

// [begin,end) of every segment for a read of read_len bases.
int fqtk_host_segment_spans(const char *text, uint64_t read_len, uint64_t *spans, size_t max_segs) {
    ReadStructure rs;
    std::string e;
    if (!ReadStructure::parse(text, &rs, &e)) return 1;
    for (size_t i = 0; i < rs.segments.size() && i < max_segs; ++i) {
        size_t lo, hi;
        segment_span(rs.segments[i], (size_t)read_len, &lo, &hi);
        spans[2 * i] = lo;
        spans[2 * i + 1] = hi;
    }
    return 0;
}

int fqtk_host_write_header(uint64_t read_num, const char *header, const char *const *bsegs, size_t nb,
                           const char *const *msegs, size_t nm, char *out, size_t cap, char *err, size_t errcap) {
    std::vector<std::string_view> b, m;
    for (size_t i = 0; i < nb; ++i) b.emplace_back(bsegs[i]);
    for (size_t i = 0; i < nm; ++i) m.emplace_back(msegs[i]);
    std::string o, e;
    if (!write_header(o, (size_t)read_num, header, b, m, &e)) { put(e, err, errcap); return 1; }
    return put(o, out, cap);
}

// Parses a FASTQ (plain or gz) and writes "head\tseq\tqual\n" per record.  Returns #records or -1.
int64_t fqtk_host_parse_fastq(const char *path, uint64_t batch, char *out, size_t cap, char *err, size_t errcap) {
    FastqSource src;
    std::string e, text;
    if (!src.open(path, &e)) { put(e, err, errcap); return -1; }
    int64_t n = 0;
    for (;;) {
        RecBatch b;
        if (!src.next_batch((size_t)batch, &b, &e)) { put(e, err, errcap); return -1; }
        if (b.recs.empty()) break;
        for (size_t i = 0; i < b.recs.size(); ++i) {
            text.append(b.head(i), b.recs[i].head_len);
            text.push_back('\t');
            text.append(b.seq(i), b.recs[i].seq_len);
            text.push_back('\t');
            text.append(b.qual(i), b.recs[i].seq_len);
            text.push_back('\n');
            ++n;
        }
    }
    if (put(text, out, cap) != 0) { put("output buffer too small", err, errcap); return -1; }
    return n;
}

// Parses a whole file and folds every record (head, seq, qual) into a 64-bit FNV-1a digest: lets the tests
// compare large plain / gzip / BGZF inputs without shipping the text back.  kind: 0 plain, 1 gzip, 2 BGZF.
int64_t fqtk_host_fastq_digest(const char *path, uint64_t batch, uint32_t inflate_helpers, uint64_t *digest, int *kind,
                               char *err, size_t errcap) {
    FastqSource src;
    std::string e;
    if (!src.open(path, &e, inflate_helpers)) { put(e, err, errcap); return -1; }
    *kind = (int)src.kind();
    uint64_t h = 1469598103934665603ull;
    auto fold = [&](const char *p, size_t n) {
        for (size_t i = 0; i < n; ++i) { h ^= (uint8_t)p[i]; h *= 1099511628211ull; }
        h ^= 0xFF; h *= 1099511628211ull;
    };
    int64_t n = 0;
    for (;;) {
        RecBatch b;
        if (!src.next_batch((size_t)batch, &b, &e)) { put(e, err, errcap); return -1; }
        if (b.recs.empty()) break;
        for (size_t i = 0; i < b.recs.size(); ++i) {
            fold(b.head(i), b.recs[i].head_len);
            fold(b.seq(i), b.recs[i].seq_len);
            fold(b.qual(i), b.recs[i].seq_len);
            ++n;
        }
    }
    *digest = h;
    return n;
}

// Decodes a gzip file held in memory with the streaming decoder of fast_inflate.hpp (tests: against zlib).
// Returns the decoded length (also when it exceeds cap: then only cap bytes were stored), -1 on a corrupt stream.
int64_t fqtk_host_gunzip(const uint8_t *in, size_t n, uint8_t *out, size_t cap, char *err, size_t errcap) {
    std::unique_ptr<FastInflate> z(new FastInflate());
    z->open(in, n, [](uint32_t seed, const void *p, size_t k) -> uint32_t {
        return (uint32_t)::crc32(seed, static_cast<const Bytef *>(p), (uInt)k);
    });
    int64_t total = 0;
    std::string e;
    for (;;) {
        const uint8_t *p = nullptr;
        size_t k = 0;
        if (!z->next(&p, &k, &e)) { put(e, err, errcap); return -1; }
        if (k == 0) break;
        if ((size_t)total < cap) std::memcpy(out + total, p, std::min(k, cap - (size_t)total));
        total += (int64_t)k;
    }
    return total;
}

// The same through parallel_gunzip.hpp (`threads` speculative decoders); stats[0] = stretches decoded in parallel,
// stats[1] = stretches that fell back (partly) to the sequential decoder.
int64_t fqtk_host_gunzip_parallel(const uint8_t *in, size_t n, uint8_t *out, size_t cap, unsigned threads, size_t chunk, uint64_t *stats,
                                  char *err, size_t errcap) {
    std::unique_ptr<ParallelGunzip> z(new ParallelGunzip());
    z->open(in, n, [](uint32_t seed, const void *p, size_t k) -> uint32_t {
        return (uint32_t)::crc32(seed, static_cast<const Bytef *>(p), (uInt)k);
    }, threads, chunk ? chunk : ParallelGunzip::kChunk);
    int64_t total = 0;
    std::string e;
    for (;;) {
        const uint8_t *p = nullptr;
        size_t k = 0;
        if (!z->next(&p, &k, &e)) { put(e, err, errcap); return -1; }
        if (k == 0) break;
        if ((size_t)total < cap) std::memcpy(out + total, p, std::min(k, cap - (size_t)total));
        total += (int64_t)k;
    }
    if (stats) { stats[0] = z->rounds(); stats[1] = z->fallbacks(); }
    return total;
}

// Reads a FASTQ file through the reader (decode + parse, nothing else) and returns the number of records; *bytes =
// sequence + quality + header bytes seen.  For timing the input side alone (tools/reader_bench.py).
int64_t fqtk_host_fastq_count(const char *path, uint64_t batch, uint32_t inflate_helpers, uint32_t gz_threads, uint64_t *bytes,
                              char *err, size_t errcap) {
    FastqSource src;
    std::string e;
    if (!src.open(path, &e, inflate_helpers, gz_threads)) { put(e, err, errcap); return -1; }
    int64_t n = 0;
    uint64_t b = 0;
    for (;;) {
        RecBatch rb;
        if (!src.next_batch((size_t)batch, &rb, &e)) { put(e, err, errcap); return -1; }
        if (rb.recs.empty()) break;
        n += (int64_t)rb.recs.size();
        for (const FastqRec &r : rb.recs) b += r.head_len + 2u * r.seq_len;
    }
    if (bytes) *bytes = b;
    return n;
}

// BGZF-compresses a whole buffer (blocks of kBgzfBlockSize + EOF marker).
int fqtk_host_bgzf(const uint8_t *in, size_t n, int level, uint8_t *out, size_t cap, size_t *out_len) {
    std::vector<uint8_t> o;
    std::string e;
    for (size_t off = 0; off < n; off += kBgzfBlockSize)
        if (!bgzf_compress_block(in + off, std::min(kBgzfBlockSize, n - off), level, o, &e)) return 1;
    o.insert(o.end(), kBgzfEof, kBgzfEof + sizeof kBgzfEof);
    if (o.size() > cap) return 2;
    std::memcpy(out, o.data(), o.size());
    *out_len = o.size();
    return 0;
}

int fqtk_host_format_f64(double v, char *out, size_t cap) { return put(format_f64(v), out, cap); }

// counts: S sample counts followed by the unmatched count.  Writes the three derived columns.
void fqtk_host_metrics(const uint64_t *counts, size_t S, double *frac, double *to_mean, double *to_best) {
    std::vector<DemuxMetric> rows(S);
    for (size_t i = 0; i < S; ++i) rows[i].templates = counts[i];
    DemuxMetric un;
    un.templates = counts[S];
    update_metrics(rows, un);
    rows.push_back(un);
    for (size_t i = 0; i <= S; ++i) {
        frac[i] = rows[i].frac_templates;
        to_mean[i] = rows[i].ratio_to_mean;
        to_best[i] = rows[i].ratio_to_best;
    }
}

int64_t fqtk_host_load_samples(const char *path, char *err, size_t errcap) {
    std::vector<Sample> s;
    std::string e;
    if (!load_samples(path, &s, &e)) { put(e, err, errcap); return -1; }
    return (int64_t)s.size();
}


// Plans the LDS-resident memo from `n_ents` (key[4], val) entries and the S x L encoded sample barcodes.
// meta (16 words) = {ok, n_slots, slot_mask_b, idx_bits, skey_off_b, salt, kw, key_stride, pow2, image_words,
//                    mph, t8_off_b, aux_off_b, bucket_mask, max_displacement, 0}.
// salt_trials < 0: the minimal-perfect-hash form (plan_lds_memo_mph) instead of the cuckoo form.
// Returns 0 (also when the memo is not of the LDS shape: meta[0] = 0), -2 if `image` is too small.
int fqtk_host_plan_lds_memo(uint32_t S, uint32_t L, const uint8_t *enc, uint64_t n_ents, const uint32_t *keys,
                            const uint32_t *vals, uint32_t *image, uint64_t cap_words, uint32_t *meta,
                            uint32_t salt_offset, int salt_trials) {
    std::vector<std::vector<uint8_t>> e(S, std::vector<uint8_t>(L));
    for (uint32_t s = 0; s < S; ++s) std::memcpy(e[s].data(), enc + (size_t)s * L, L);
    std::vector<fqtk::LdsEntry> ents(n_ents);
    for (uint64_t i = 0; i < n_ents; ++i) {
        for (int w = 0; w < 4; ++w) ents[i].k[w] = keys[4 * i + w];
        ents[i].val = vals[i];
    }
    const fqtk::LdsMemoPlan p = salt_trials < 0 ? fqtk::plan_lds_memo_mph(S, L, ents, e, salt_offset)
                                                : fqtk::plan_lds_memo(S, L, ents, e, salt_offset, salt_trials);
    const uint32_t m[16] = {p.ok ? 1u : 0u, p.n_slots, p.slot_mask_b, p.idx_bits, p.skey_off_b, p.salt,
                            (uint32_t)p.kw, (uint32_t)p.key_stride, p.pow2 ? 1u : 0u, (uint32_t)p.image.size(),
                            p.mph ? 1u : 0u, p.t8_off_b, p.aux_off_b, p.bucket_mask, p.max_displacement, 0u};
    std::memcpy(meta, m, sizeof m);
    if (!p.ok) return 0;
    if (p.image.size() > cap_words) return -2;
    std::memcpy(image, p.image.data(), p.image.size() * 4);
    return 0;
}

// The kernel's lookup (csrc/lds_memo_plan.hpp: lds_memo_lookup) over n keys.
void fqtk_host_lds_memo_lookup(const uint32_t *image, const uint32_t *meta, uint64_t n, const uint32_t *keys,
                               uint32_t *out) {
    fqtk::LdsMemoPlan p;
    p.image.assign(image, image + meta[9]);
    p.n_slots = meta[1]; p.slot_mask_b = meta[2]; p.idx_bits = meta[3]; p.skey_off_b = meta[4];
    p.salt = meta[5]; p.kw = (int)meta[6]; p.key_stride = (int)meta[7]; p.pow2 = meta[8] != 0;
    p.mph = meta[10] != 0; p.t8_off_b = meta[11]; p.aux_off_b = meta[12]; p.bucket_mask = meta[13];
    for (uint64_t i = 0; i < n; ++i) out[i] = fqtk::lds_memo_lookup(p, keys + 4 * i);
}

// The LDS forms' encode of n read words (four bases each; csrc/memo_hash.hpp: encode_word, then recode_flagged_bytes for a word
// with a byte that is not A C G T N): codes[i] = the word's four codes, junk[i] = 0x80 in every byte of no IUPAC meaning.
void fqtk_host_recode_words(const uint32_t *words, uint64_t n, uint32_t code_mask, uint32_t byte_mask, uint32_t *codes, uint32_t *junk) {
    for (uint64_t i = 0; i < n; ++i) {
        uint32_t c, x;
        fqtk::encode_word(words[i], code_mask, byte_mask, c, x);
        junk[i] = x ? fqtk::recode_flagged_bytes(words[i], x, c, code_mask) : 0u;
        codes[i] = c;
    }
}

// Plans the direct-indexed memo (csrc/direct_memo_plan.hpp) from n_ents no-call-free entries given as
// unfolded key words (lo = bases 0-7, hi = bases 8-9) + result words, then replays the kernel's lookup for
// n_q query keys: out[i] = result word or 0xFFFFFFFF, cached[i] = 1 when the LDS cache answered.
// meta = {entry_bytes, ib, bb, hot2_bits, hot2_slots, hot2_wanted, hot2_placed, table_entries}.
int fqtk_host_direct_memo(uint32_t S, uint32_t L, uint64_t n_ents, const uint32_t *keys, const uint32_t *vals,
                          uint64_t n_q, const uint32_t *qkeys, uint32_t *out, uint8_t *cached, uint32_t *meta) {
    std::vector<fqtk::DirectEntry> ents(n_ents);
    for (uint64_t i = 0; i < n_ents; ++i) ents[i] = fqtk::DirectEntry{keys[2 * i], keys[2 * i + 1], vals[i]};
    const fqtk::DirectMemoPlan p = fqtk::plan_direct_memo(S, L, ents);
    const uint32_t m[8] = {(uint32_t)p.entry_bytes, p.ib, p.bb, p.hot2_bits, (uint32_t)p.hot2.size(), (uint32_t)p.hot2_wanted,
                           (uint32_t)p.hot2_placed, (uint32_t)(p.entry_bytes == 2 ? p.table16.size() : p.table32.size())};
    std::memcpy(meta, m, sizeof m);
    if (!p.entry_bytes) return 0;
    for (uint64_t i = 0; i < n_q; ++i) {
        bool c = false;
        out[i] = fqtk::direct_memo_lookup(p, qkeys[2 * i], qkeys[2 * i + 1], &c);
        cached[i] = c ? 1 : 0;
    }
    return 0;
}

// The direct form's table of the entries WITH a no-call (csrc/direct_memo_plan.hpp: plan_nbuckets): planned from n
// (folded key, result) pairs, then the kernel's lookup replayed for n_q keys.  meta = {ok, buckets, keys in their second bucket}.
int fqtk_host_nbuckets(uint64_t n, const uint32_t *lo, const uint32_t *vals, uint64_t n_q, const uint32_t *q, uint32_t *out, uint32_t *meta) {
    std::vector<fqtk::NKey> keys(n);
    for (uint64_t i = 0; i < n; ++i) keys[i] = fqtk::NKey{lo[i], vals[i]};
    const fqtk::NBucketPlan p = fqtk::plan_nbuckets(keys);
    meta[0] = p.ok ? 1u : 0u;
    meta[1] = p.ok ? p.mask + 1u : 0u;
    meta[2] = (uint32_t)p.second;
    if (!p.ok) return 0;
    for (uint64_t i = 0; i < n_q; ++i) out[i] = fqtk::nbucket_lookup(p, q[i]);
    return 0;
}

// The GPU BGZF compressor's phase functions (csrc/bgzf_deflate.hpp) run lane by lane on the CPU -- the same
// code the HIP kernel runs with barriers in between.  Test infrastructure: lets the CPU suite inflate what
// the algorithm produces (zlib) without a GPU.  Returns the DEFLATE payload size, or -1 on a bad argument.
// lockstep != 0: the LZ phase advances all lanes one token at a time, round robin, instead of lane after lane --
// the algorithm is built so that the interleaving cannot matter (min / max tables, lane-private parse state), and
// the tests check that both orders give identical bytes.
// The code builder of the BGZF compressor alone (tests: complete, length-limited codes for any counts).
int fqtk_host_huffman_lengths(const uint32_t *counts, int n, int max_bits, uint8_t *len) {
    using namespace fqtk::bgzf;
    if (n < 2 || n > 288 || max_bits < 1 || max_bits > 15) return -1;
    std::vector<uint8_t> mem(sizeof(Shared));
    Shared &S = *reinterpret_cast<Shared *>(mem.data());
    huffman_lengths(huff_scratch_ll(S), counts, n, max_bits, len);
    return 0;
}

int64_t fqtk_host_bgzf_deflate_level(const uint8_t *in, uint32_t n, uint8_t *out, size_t cap, int *stored, int lockstep, int level);
int64_t fqtk_host_bgzf_deflate_emulated(const uint8_t *in, uint32_t n, uint8_t *out, size_t cap, int *stored, int lockstep) {
    return fqtk_host_bgzf_deflate_level(in, n, out, cap, stored, lockstep, 5);
}
int64_t fqtk_host_bgzf_deflate_level(const uint8_t *in, uint32_t n, uint8_t *out, size_t cap, int *stored, int lockstep, int level) {
    using namespace fqtk::bgzf;
    if (n == 0 || n > kMaxIn || cap < kOutStride || level < 1) return -1;
    std::vector<uint8_t> mem(sizeof(Shared));
    Shared &S = *reinterpret_cast<Shared *>(mem.data());
    S.effort = effort_of_level((uint32_t)level);
    std::vector<uint32_t> tok(kTokensPerBlock);
    for (int l = 0; l < kLanes; ++l) phase_load(S, l, in, n);
    for (int l = 0; l < kLanes; ++l) phase_count(S, l, n);
    for (int l = 0; l < kLanes; ++l) phase_literal_costs(S, l, n);
    std::vector<uint64_t> cheap(kLanes);
    for (int l = 0; l < kLanes; ++l) cheap[l] = phase_index(S, l, n);
    if (lockstep) {
        std::vector<LzLane> st(kLanes);
        for (int l = 0; l < kLanes; ++l) lz_begin(S, l, n, st[l], cheap[l]);
        for (bool any = true; any;) {
            any = false;
            for (int l = 0; l < kLanes; ++l) any = lz_step(S, l, n, tok.data(), st[l]) || any;
        }
        for (int l = 0; l < kLanes; ++l) lz_end(S, l, st[l]);
    } else
    for (int l = 0; l < kLanes; ++l) phase_lz(S, l, n, tok.data(), cheap[l]);
    {   // (every lane reads its neighbours' ends before any span is replaced: a barrier on the device)
        std::vector<uint32_t> span(kLanes);
        for (int l = 0; l < kLanes; ++l) phase_reach(S, l, tok.data(), &span[l]);
        for (int l = 0; l < kLanes; ++l) S.span[l] = span[l];
    }
    for (int l = 0; l < kLanes; ++l) phase_clear_out(S, l);
    for (int l = 0; l < kLanes; ++l) phase_code_lengths(S, l);
    for (int l = 0; l < kLanes; ++l) phase_codes(S, l);
    for (int l = 0; l < kLanes; ++l) phase_cl_runs(S, l);
    for (int l = 0; l < kLanes; ++l) phase_cl_emit(S, l);
    phase_cl_code(S);
    for (int l = 0; l < kLanes; ++l) phase_cl_bits(S, l);
    for (int l = 0; l < kLanes; ++l) phase_count_bits(S, l, n, tok.data());
    phase_offsets(S, n);
    for (int l = 0; l < kLanes; ++l) phase_emit(S, l, n, tok.data());
    uint32_t bytes = 0;
    for (int l = 0; l < kLanes; ++l) bytes = phase_store(S, l, in, n, out);
    if (stored) *stored = (int)S.stored;
    return (int64_t)bytes;
}

// CRC-32 of a block the way the BGZF kernel takes it (csrc/bgzf_deflate.hpp: every lane its slice, values folded).
uint32_t fqtk_host_bgzf_crc_emulated(const uint8_t *in, uint32_t n) {
    using namespace fqtk::bgzf;
    if (n == 0 || n > kMaxIn) return 0;
    std::vector<uint8_t> mem(sizeof(Shared));
    Shared &S = *reinterpret_cast<Shared *>(mem.data());
    for (int l = 0; l < kLanes; ++l) crc_tables(S, l);
    for (int l = 0; l < kLanes; ++l) phase_load(S, l, in, n);
    for (int l = 0; l < kLanes; ++l) phase_crc(S, l, n);
    phase_crc_fold(S);
    return S.crc;
}

// Length of the gzip member header at data (bgzf_walk.hpp: what the serial-gzip feeders of `fqtk demux` skip), 0 if there is none.
uint64_t fqtk_host_gzip_header_len(const uint8_t *data, size_t n) { return (uint64_t)fqtk_host::gzip_header_len(data, n); }

// The member walk of the device-inflate feeders (bgzf_walk.hpp): runs of whole members of `path` below max_bytes of file /
// max_text of text each.  Writes up to cap rows of (run, payload offset in the file, payload bytes, ISIZE, CRC-32); returns the
// number of members, -1 with *err when a run is asked for where no BGZF member lies / the file is damaged.
int64_t fqtk_host_bgzf_walk(const char *path, uint64_t max_bytes, uint64_t max_text, uint64_t *rows, size_t cap, char *err, size_t err_cap) {
    fqtk_host::BgzfFile f;
    std::string e;
    auto fail = [&](const std::string &m) { if (err && err_cap) { std::strncpy(err, m.c_str(), err_cap - 1); err[err_cap - 1] = 0; } return (int64_t)-1; };
    if (!f.open(path, &e)) return fail(e);
    std::vector<fqtk_inflate_member> run;
    int64_t n = 0;
    uint64_t r = 0;
    while (!f.at_end()) {
        size_t from = 0, upto = 0;
        if (!f.next_run((size_t)max_bytes, (size_t)max_text, &run, &from, &upto, &e)) return fail(e);
        for (const fqtk_inflate_member &m : run) {
            if ((size_t)n < cap) { rows[5 * n] = r; rows[5 * n + 1] = from + m.payload_off; rows[5 * n + 2] = m.payload_len; rows[5 * n + 3] = m.isize; rows[5 * n + 4] = m.crc; }
            ++n;
        }
        ++r;
    }
    return n;
}

// A raw DEFLATE stream decoded by the BGZF input kernel's own code (csrc/bgzf_inflate.hpp: inflate_member, one wavefront per
// member), run on the CPU as 64 fibers (wave_emu.hpp).  `misalign` (0..3) puts the payload that many bytes behind a 4-byte
// boundary, as a member's payload lies in a file.  Returns the member's status (fqtk::inflate::kOk = 0, kErr*).
int fqtk_host_bgzf_inflate_emulated(const uint8_t *payload, uint32_t payload_len, uint32_t isize, uint8_t *out, uint32_t misalign) {
    using namespace fqtk::inflate;
    misalign &= 3u;
    const uint32_t words = (misalign + payload_len) / 4u;   // whole dwords; the rest are the buffer's tail bytes
    std::vector<uint32_t> buf(words + 2u, 0xA5A5A5A5u);   // what lies behind the payload is not zero (and must not be read)
    std::memcpy(reinterpret_cast<uint8_t *>(buf.data()) + misalign, payload, payload_len);
    std::vector<uint8_t> mem(sizeof(Shared), 0xC3);   // LDS holds whatever the last workgroup left
    Shared &S = *reinterpret_cast<Shared *>(mem.data());
    MemberArgs a;
    a.in_words = buf.data();
    a.first_bit = 8u * misalign;
    a.payload_bits = 8u * payload_len;
    a.readable_words = words;
    a.tail_bytes = (misalign + payload_len) & 3u;
    a.out = out;
    a.isize = isize;
    uint32_t status[64];
    fqtk_host::WaveEmu wave;
    wave.run([&](fqtk_host::WaveEmu &w) { status[w.lane()] = inflate_member(w, S, a); });
    for (int l = 1; l < 64; ++l)
        if (status[l] != status[0]) return -1;   // the status is wave-uniform by construction
    return (int)status[0];
}

// A piece of a serial DEFLATE stream decoded by the device decoder's stream mode (inflate_member<W, true>) on the wave emulator:
// from bit `start_bit` of data (a block boundary) to the first block boundary at or behind stop_bit, into 16-bit symbols
// (< 256: a byte; 256 + j: byte j of the 32 KiB in front of the piece).  Returns the status; res = {symbols, end bit, final block, whole blocks decoded} (the first three as of the last block boundary reached).
int fqtk_host_inflate_stream_emulated(const uint8_t *data, uint32_t len, uint32_t start_bit, uint32_t stop_bit, uint16_t *sym, uint32_t cap, uint32_t *res) {
    using namespace fqtk::inflate;
    const uint32_t words = len / 4u;
    std::vector<uint32_t> buf(words + 2u, 0xA5A5A5A5u);
    std::memcpy(buf.data(), data, len);
    std::vector<uint8_t> mem(sizeof(Shared), 0xC3);
    Shared &S = *reinterpret_cast<Shared *>(mem.data());
    MemberArgs a;
    std::memset(&a, 0, sizeof a);
    a.in_words = buf.data() + (start_bit >> 5);   // (the device wrapper rebases a chunk the same way)
    a.first_bit = start_bit & 31u;
    a.payload_bits = 8u * len - (start_bit & ~31u) - a.first_bit;
    a.readable_words = words - (start_bit >> 5);
    a.tail_bytes = len & 3u;
    a.out = nullptr;
    a.isize = cap;
    a.out_sym = sym;
    a.stop_bit = stop_bit - (start_bit & ~31u);
    uint32_t status[64];
    StreamEnd end = {0, 0, 0, 0, 0};
    fqtk_host::WaveEmu wave;
    wave.run([&](fqtk_host::WaveEmu &w) { status[w.lane()] = inflate_member<fqtk_host::WaveEmu, true>(w, S, a, &end); });
    for (int l = 1; l < 64; ++l)
        if (status[l] != status[0]) return -1;
    res[0] = end.n_sym;
    res[1] = end.end_bit + (start_bit & ~31u);
    res[2] = end.final_block;
    res[3] = end.n_blocks;
    return (int)status[0];
}

// The block-start search of the serial-gzip path (csrc/bgzf_inflate.hpp: find_block_start, a lane per bit position) on the wave
// emulator: first bit in [from_bit, limit_bit) of data where a non-final dynamic-Huffman block can start; ~0 if none.
uint64_t fqtk_host_find_block_start_emulated(const uint8_t *data, uint32_t len, uint32_t from_bit, uint32_t limit_bit, int low_literals_only) {
    using namespace fqtk::inflate;
    const uint32_t words = len / 4u;
    std::vector<uint32_t> buf(words + 2u, 0xA5A5A5A5u);
    std::memcpy(buf.data(), data, len);
    std::vector<uint8_t> mem(sizeof(Shared), 0xC3);
    Shared &S = *reinterpret_cast<Shared *>(mem.data());
    MemberArgs a;
    std::memset(&a, 0, sizeof a);
    a.in_words = buf.data();
    a.first_bit = 0;
    a.payload_bits = 8u * len;
    a.readable_words = words;
    a.tail_bytes = len & 3u;
    uint32_t found[64];
    std::vector<uint8_t> kraft(kKraftLutBytes, 0xEE);   // (the device's: built by the searching wavefront, then five look-ups per candidate)
    fqtk_host::WaveEmu wave;
    wave.run([&](fqtk_host::WaveEmu &w) {
        build_kraft_lut(w, kraft.data());
        found[w.lane()] = find_block_start(w, S, a, from_bit, limit_bit, low_literals_only != 0, kraft.data());
    });
    for (int l = 1; l < 64; ++l)
        if (found[l] != found[0]) return ~0ull - 1u;   // the result is wave-uniform by construction
    return found[0] == 0xFFFFFFFFu ? ~0ull : (uint64_t)found[0];
}
// ... and the host's own (parallel_gunzip.hpp: SpecInflate::find_block_start), which the decoder threads of --host-inflate use.
uint64_t fqtk_host_find_block_start(const uint8_t *data, size_t len, uint64_t from_bit, uint64_t limit_bit) {
    SpecInflate f;
    f.attach(data, len);
    return f.find_block_start(from_bit, limit_bit);
}

// A stretch of a serial DEFLATE stream decoded by the host's sequential decoder from a known block boundary (region_inflate.hpp: what
// `fqtk demux` falls back to where the device cannot cut or decode a stretch).  data[0..n): a raw DEFLATE stream or a whole file;
// window: 32 KiB or NULL (the stream starts at from_bit).  Returns 0, -1 (corrupt: err says how) or -2 (out too small);
// res = {bytes of text, end bit, final block}.
int fqtk_host_region_inflate(const uint8_t *data, size_t n, uint64_t from_bit, const uint8_t *window, uint64_t until_bit, uint64_t max_text, uint8_t *out, size_t cap,
                             uint64_t *res, uint8_t *window_after, char *err, size_t errcap) {
    RegionInflate z;
    z.attach(data, n);
    std::vector<uint8_t> text;
    uint64_t end_bit = 0;
    bool final_block = false;
    std::string e;
    if (!z.run(from_bit, window, until_bit, (size_t)max_text, &text, &end_bit, &final_block, window_after, &e)) { put(e, err, errcap); return -1; }
    res[0] = text.size();
    res[1] = end_bit;
    res[2] = final_block ? 1 : 0;
    if (text.size() > cap) return -2;
    std::memcpy(out, text.data(), text.size());
    return 0;
}

// One output record the way the GPU record pipeline states it (csrc/record_format.hpp: header plan + pieces), built
// from strings: `header` is the first input's header, bsegs / msegs the sample / molecular barcode segments, bases and
// quals the segment the file takes.  Returns the record's length (also what the sizing sink says), -1 - HeaderError for
// a header the reference rejects, -100 when out is too small or the two sinks disagree.
int64_t fqtk_host_format_record(const char *header, uint32_t read_num, const char *const *bsegs, uint32_t nb,
                                const char *const *msegs, uint32_t nm, const char *bases, const char *quals, char *out, size_t cap) {
    using namespace fqtk::fmt;
    // one "text" per input: input 0 = header, then the segments
    std::vector<std::string> texts;
    texts.emplace_back(header);
    std::vector<Span> b, m;
    for (uint32_t i = 0; i < nb; ++i) { texts.emplace_back(bsegs[i]); b.push_back(Span{(uint32_t)texts.size() - 1, 0, (uint32_t)texts.back().size()}); }
    for (uint32_t i = 0; i < nm; ++i) { texts.emplace_back(msegs[i]); m.push_back(Span{(uint32_t)texts.size() - 1, 0, (uint32_t)texts.back().size()}); }
    texts.emplace_back(bases);
    const Span sb{(uint32_t)texts.size() - 1, 0, (uint32_t)texts.back().size()};
    texts.emplace_back(quals);
    const Span sq{(uint32_t)texts.size() - 1, 0, (uint32_t)texts.back().size()};
    const HeaderPlan p = plan_header(reinterpret_cast<const uint8_t *>(texts[0].data()), (uint32_t)texts[0].size(), nm != 0);
    if (p.err) return -1 - (int64_t)p.err;
    LenSink ls;
    emit_record(ls, p, 0, read_num, b.data(), nb, m.data(), nm, sb, sq);
    std::vector<Piece> pcs(kMaxPieces);
    if (max_pieces(nb, nm) > (uint32_t)kMaxPieces) return -100;
    PieceSink ps(pcs.data());
    emit_record(ps, p, 0, read_num, b.data(), nb, m.data(), nm, sb, sq);
    std::string rec;
    for (uint32_t k = 0; k < ps.n; ++k) {
        const Piece &pc = pcs[k];
        if (pc.is_lit) for (uint32_t j = 0; j < pc.len; ++j) rec.push_back((char)(pc.lit >> (8 * j)));
        else rec.append(texts[pc.input], pc.off, pc.len);
    }
    uint32_t bl = 0, ml = 0, digits = 0;
    for (const Span &x : b) bl += x.len;
    for (const Span &x : m) ml += x.len;
    for (uint32_t v = read_num; digits == 0 || v; v /= 10) ++digits;
    if (record_len(p, digits, bl, nb, ml, nm, sb.len) != ls.n) return -100;
    if (rec.size() != ls.n || rec.size() > cap || ps.n > max_pieces(nb, nm)) return -100;
    {   // the slot table k_format's lanes work from must spell the same bytes
        uint32_t numw[4];
        const uint32_t num_len = number_literal(read_num, p.kind, numw);
        std::string rec2;
        for (uint32_t sidx = 0; sidx < record_slots(nb, nm); ++sidx) {
            const Slot z = record_slot(sidx, p, 0, num_len, b.data(), nb, m.data(), nm, sb, sq);
            if (z.kind == kSpan) rec2.append(texts[z.input], z.off, z.len);
            else if (z.kind == kLiteral) for (uint32_t j = 0; j < z.len; ++j) rec2.push_back((char)(z.lit >> (8 * j)));
            else for (uint32_t j = 0; j < z.len; ++j) rec2.push_back((char)(numw[j >> 2] >> (8 * (j & 3))));
        }
        if (rec2 != rec) return -101;
    }
    std::memcpy(out, rec.data(), rec.size());
    return (int64_t)rec.size();
}

// FastqSource::next_raw over a whole file in calls of `batch` records: the concatenated text goes to out (cap bytes),
// the number of records of every call to counts (max_calls).  Returns the number of calls made, -1 on an error.
namespace {
struct HeapRaw : fqtk_host::RawBuffer {
    std::vector<char> v;
    bool grow(size_t want, size_t keep) override { (void)keep; v.resize(want); data = v.data(); cap = v.size(); return true; }
};
}  // namespace
int64_t fqtk_host_read_raw(const char *path, uint64_t batch, char *out, size_t cap, size_t *out_len, uint64_t *counts, size_t max_calls,
                           char *err, size_t errcap) {
    FastqSource src;
    std::string e;
    if (!src.open(path, &e)) { put(e, err, errcap); return -1; }
    HeapRaw buf;
    buf.grow(4096, 0);   // small on purpose: the tests make it grow
    size_t w = 0, calls = 0;
    for (;;) {
        size_t n = 0, bytes = 0;
        if (!src.next_raw((size_t)batch, &buf, &n, &bytes, &e)) { put(e, err, errcap); return -1; }
        if (n == 0) break;
        if (w + bytes > cap || calls >= max_calls) { put("test buffer too small", err, errcap); return -1; }
        std::memcpy(out + w, buf.data, bytes);
        w += bytes;
        counts[calls++] = n;
    }
    *out_len = w;
    return (int64_t)calls;
}

// What `fqtk demux` does with every input before its reader threads start: judge the record size by the first MiB
// (estimate_raw_bytes), then read.  Returns the records of the first next_raw call, or -1 with *err -- and must RETURN when the input
// is damaged: the producer reports an error once, and the estimate used to swallow it (the reader then waited on an empty queue).
int64_t fqtk_host_estimate_then_read(const char *path, uint64_t batch, uint64_t *estimate, char *err, size_t errcap) {
    FastqSource src;
    std::string e;
    if (!src.open(path, &e)) { put(e, err, errcap); return -1; }
    *estimate = src.estimate_raw_bytes(1024);
    HeapRaw buf;
    buf.grow(1 << 16, 0);
    size_t n = 0, bytes = 0;
    if (!src.next_raw((size_t)batch, &buf, &n, &bytes, &e)) { put(e, err, errcap); return -1; }
    return (int64_t)n;
}

// Drives ChunkSchedule the way `fqtk demux` does -- one submitter, chunks collected in order, completions arriving
// whenever `order` says (order[i] != 0: try to collect one chunk before the next submit) -- and checks what the
// pipeline relies on.  Returns 0, or the number of the first rule broken:
//   1 a (device, slot) pair was handed a chunk while its previous one was still outstanding
//   2 a device did not receive its chunks in ascending order
//   3 more than devices * slots chunks were outstanding
//   4 not every chunk was submitted and collected
int fqtk_host_chunk_schedule_check(uint64_t devices, uint64_t slots, uint64_t n_chunks, const uint8_t *order, size_t n_order) {
    ChunkSchedule sc;
    sc.devices = (size_t)devices;
    sc.slots = (size_t)slots;
    std::vector<int64_t> busy(devices * slots, -1), last_on_device(devices, -1);
    uint64_t next = 0, done = 0;
    size_t at = 0;
    auto collect_one = [&]() {
        if (done == next) return;
        const size_t cell = (size_t)sc.device_of(done) * slots + (size_t)sc.slot_of(done);
        busy[cell] = -1;
        ++done;
    };
    auto submit = [&]() -> int {
        const int d = sc.device_of(next), sl = sc.slot_of(next);
        const size_t cell = (size_t)d * slots + (size_t)sl;
        if (busy[cell] >= 0) return 1;
        if (last_on_device[d] >= (int64_t)next) return 2;
        busy[cell] = (int64_t)next;
        last_on_device[d] = (int64_t)next;
        ++next;
        return next - done > devices * slots ? 3 : 0;
    };
    while (done < n_chunks) {
        const bool want_collect = at < n_order ? order[at++] != 0 : true;
        const bool can_submit = next < n_chunks && sc.may_submit(next, done);
        if (can_submit && !want_collect) { if (int rc = submit()) return rc; }
        else if (done < next) collect_one();
        else if (can_submit) { if (int rc = submit()) return rc; }
        else return 4;   // nothing outstanding and nothing may be submitted: stuck
    }
    return (next == n_chunks && done == n_chunks) ? 0 : 4;
}

// The threads of `fqtk demux --devices a,b,..` (chunk_dispatch.hpp) over a FAKE device: submits and collects take random
// times (microseconds drawn from `seed`), every device's chunks are submitted by its own thread, one collector takes them
// back.  Returns 0, or the first rule broken:
//   1 a (device, slot) pair was handed a chunk while its previous one was still outstanding
//   2 a device did not receive its chunks in ascending order, or a chunk went to the wrong device / slot
//   3 chunks were not collected in order 0, 1, 2, ..
//   4 not every chunk was submitted and collected exactly once, or its payload / meta did not come through
//   5 two submits ran on one device at the same time (a device has ONE submit thread)
// *overlap = 1 if submits on two different devices were ever in progress at the same time (the point of the exercise).
int fqtk_host_chunk_dispatch_check(uint64_t devices, uint64_t slots, uint64_t n_chunks, uint64_t seed, int *overlap) {
    struct Job { uint64_t payload; };
    struct Meta { uint64_t payload = 0, k = 0; };
    std::mutex mu;
    std::vector<int64_t> busy(devices * slots, -1), last_on_device(devices, -1);
    std::vector<int> in_submit(devices, 0);
    std::vector<uint8_t> submitted(n_chunks, 0), collected(n_chunks, 0);
    int broken = 0, saw_overlap = 0;
    uint64_t next_collect = 0, next_retire = 0;
    const bool with_retire = (seed & 1u) != 0;
    std::vector<uint8_t> gathered(n_chunks, 0);
    auto rnd = [seed](uint64_t k, uint64_t salt) { uint64_t x = (k + 1) * 0x9E3779B97F4A7C15ull ^ (seed + salt) * 0xBF58476D1CE4E5B9ull; x ^= x >> 29; x *= 0x94D049BB133111EBull; return (x >> 40) % 300; };
    auto flag = [&](int rule) { if (!broken) broken = rule; };
    {
        ChunkDispatcher<Job, Meta> d((size_t)devices, (size_t)slots,
            [&](int dev, int slot, uint64_t k, Job &j) {
                {
                    std::lock_guard<std::mutex> lk(mu);
                    if ((uint64_t)dev != k % devices || (uint64_t)slot != (k / devices) % slots) flag(2);
                    const size_t cell = (size_t)dev * slots + (size_t)slot;
                    if (busy[cell] >= 0) flag(1);
                    busy[cell] = (int64_t)k;
                    if (last_on_device[dev] >= (int64_t)k) flag(2);
                    last_on_device[dev] = (int64_t)k;
                    if (in_submit[dev]++) flag(5);
                    for (uint64_t o = 0; o < devices; ++o) if ((int)o != dev && in_submit[o]) saw_overlap = 1;
                    if (k < n_chunks && submitted[k]++) flag(4);
                }
                std::this_thread::sleep_for(std::chrono::microseconds(rnd(k, 1)));
                {
                    std::lock_guard<std::mutex> lk(mu);
                    --in_submit[dev];
                }
                Meta m;
                m.payload = j.payload;
                m.k = k;
                return m;
            },
            [&](int dev, int slot, uint64_t k, Meta &m) {
                std::this_thread::sleep_for(std::chrono::microseconds(rnd(k, 2)));
                std::lock_guard<std::mutex> lk(mu);
                if (k != next_collect++) flag(3);
                if (m.k != k || m.payload != k * 7 + 1) flag(4);
                const size_t cell = (size_t)dev * slots + (size_t)slot;
                if (busy[cell] != (int64_t)k) flag(1);
                if (with_retire) { if (k < n_chunks && gathered[k]++) flag(6); return; }   // (the slot stays taken until the chunk has retired)
                busy[cell] = -1;
                if (k < n_chunks && collected[k]++) flag(4);
            },
            // odd seeds: a third stage on its own thread (the record pipeline's writers), in order, behind the collection; the slot is free only then
            with_retire ? ChunkDispatcher<Job, Meta>::RetireFn([&](int dev, int slot, uint64_t k, Meta &m) {
                std::this_thread::sleep_for(std::chrono::microseconds(rnd(k, 5)));
                std::lock_guard<std::mutex> lk(mu);
                if (k != next_retire++) flag(6);
                if (k >= n_chunks || gathered[k] != 1 || m.k != k) flag(6);
                const size_t cell = (size_t)dev * slots + (size_t)slot;
                if (busy[cell] != (int64_t)k) flag(1);
                busy[cell] = -1;
                if (k < n_chunks && collected[k]++) flag(4);
            }) : ChunkDispatcher<Job, Meta>::RetireFn(nullptr));
        for (uint64_t k = 0; k < n_chunks; ++k) {
            if (rnd(k, 3) < 30) std::this_thread::sleep_for(std::chrono::microseconds(rnd(k, 4)));   // a reader that stalls now and then
            d.push(Job{k * 7 + 1});
        }
        d.finish();
    }
    for (uint64_t k = 0; k < n_chunks; ++k) if (submitted[k] != 1 || collected[k] != 1) flag(4);
    if (next_collect != n_chunks) flag(4);
    if (overlap) *overlap = saw_overlap;
    return broken;
}

// FastqSource::next_cut over a whole plain (mapped) file: the same contract as fqtk_host_read_raw, the text taken
// from the cuts (+ the newline a cut asks for).  Returns the number of cuts, -1 on an error, -2 if the file is not mapped.
static int64_t read_cuts(const char *path, uint64_t batch, char *out, size_t cap, size_t *out_len, uint64_t *counts, size_t max_calls,
                         char *err, size_t errcap, bool assistant, size_t hold = 0, size_t unmap_step = 0, size_t *unmapped = nullptr);
int64_t fqtk_host_read_cuts(const char *path, uint64_t batch, char *out, size_t cap, size_t *out_len, uint64_t *counts, size_t max_calls,
                            char *err, size_t errcap) {
    return read_cuts(path, batch, out, cap, out_len, counts, max_calls, err, errcap, false);
}
// ... with a second thread counting the later steps of every cut (FastqSource::attach_count_assistant)
int64_t fqtk_host_read_cuts_assisted(const char *path, uint64_t batch, char *out, size_t cap, size_t *out_len, uint64_t *counts, size_t max_calls,
                                     char *err, size_t errcap) {
    return read_cuts(path, batch, out, cap, out_len, counts, max_calls, err, errcap, true);
}
// ... with the consumer `hold` cuts behind the cutter (the copier of `fqtk demux` behind its queue) and the consumed
// input unmapped in steps of `unmap_step` bytes: a cut must stay readable until release_cut() hands it back.
// *unmapped = bytes the source had unmapped when the last cut was taken.
int64_t fqtk_host_read_cuts_held(const char *path, uint64_t batch, uint64_t hold, uint64_t unmap_step, char *out, size_t cap, size_t *out_len,
                                 uint64_t *counts, size_t max_calls, size_t *unmapped, char *err, size_t errcap) {
    return read_cuts(path, batch, out, cap, out_len, counts, max_calls, err, errcap, false, (size_t)hold, (size_t)unmap_step, unmapped);
}
static int64_t read_cuts(const char *path, uint64_t batch, char *out, size_t cap, size_t *out_len, uint64_t *counts, size_t max_calls,
                         char *err, size_t errcap, bool assistant, size_t hold, size_t unmap_step, size_t *unmapped) {
    FastqSource src;
    std::string e;
    if (!src.open(path, &e)) { put(e, err, errcap); return -1; }
    if (!src.mapped()) return -2;
    if (assistant) src.attach_count_assistant();
    if (unmap_step) src.set_cut_unmap_step(unmap_step);
    size_t w = 0, calls = 0;
    std::deque<FastqSource::RawCut> held;
    auto consume = [&]() -> bool {   // copy the oldest cut, then hand it back
        const FastqSource::RawCut c = held.front();
        held.pop_front();
        const size_t bytes = c.bytes + (c.add_newline ? 1 : 0);
        if (w + bytes > cap || calls >= max_calls) { put("test buffer too small", err, errcap); return false; }
        std::memcpy(out + w, c.p, c.bytes);
        if (c.add_newline) out[w + c.bytes] = '\n';
        src.release_cut(c);
        w += bytes;
        counts[calls++] = c.n_records;
        return true;
    };
    for (;;) {
        FastqSource::RawCut c;
        if (!src.next_cut((size_t)batch, &c, &e)) { put(e, err, errcap); return -1; }
        if (c.n_records == 0) break;
        held.push_back(c);
        if (held.size() > hold && !consume()) return -1;
    }
    if (unmapped) *unmapped = src.cut_unmapped_bytes();
    if (unmap_step) usleep(20000);   // let the unmapper thread get to what was queued: a cut unmapped too early must fault HERE
    while (!held.empty()) if (!consume()) return -1;
    *out_len = w;
    return (int64_t)calls;
}
}  // extern "C"
