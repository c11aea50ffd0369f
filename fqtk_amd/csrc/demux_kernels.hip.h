// demux_kernels.hip.h -- gfx950 kernels of the record pipeline (include/fqtk_demux.h): FASTQ text in HBM ->
// line index -> records -> barcode rows -> [matcher] -> placement of every output record by stable per-file prefix
// sums -> records copied into 65 280-byte blocks -> [DEFLATE kernel] -> BGZF members packed per output file.
//
// All of it is byte / integer work bound by HBM or by latency, none of it is a contraction: no MFMA.  One wavefront is
// 64 lanes; where lanes cooperate on one record they do so as a wave (k_format), otherwise one lane owns one template.
// Reference semantics: ReadSetIterator::next /root/reference/src/bin/commands/demux.rs:285-343 (records, too-few-bases),
// ReadSet::sample_barcode_sequence :121-123 (barcode rows), SampleWriters::write :396-415 + write_header_internal
// :171-267 (record text; stated once in record_format.hpp), DemuxMetric counting :970-974 (the count column).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/fqtk_demux.h"
#include "../../include/fqtk_bgzf.h"
#include "record_format.hpp"

namespace fqtk {
namespace demux {

constexpr uint32_t kBlock = FQTK_BGZF_MAX_IN;   // uncompressed bytes of a BGZF block (as the bgzf crate cuts them)
constexpr uint32_t kSlab = 65536;               // bytes a block occupies in HBM
constexpr uint32_t kTile = 1024;                // templates ranked together (one workgroup)
constexpr uint32_t kRankGroup = 2;              // output files ranked per pass over a tile (LDS: 4 KiB each + 4 KiB of keys)
constexpr uint32_t kLineTile = 4096;            // bytes of text per workgroup of the line index (256 lanes x 16 B)
constexpr uint32_t kMaxSegs = 24;               // sample- / molecular-barcode segments over all inputs
constexpr uint32_t kPersist = 3;                // persistent slabs per output file, used in rotation (see FileState)
constexpr unsigned long long kNoError = ~0ull;

struct RecView { uint32_t head_off, head_len, seq_off, seq_len, qual_off; };   // head: after the '@'

// What the kernels need to know of the run (kernel argument: lives in SGPRs / the scalar cache).
struct DevConfig {
    uint32_t n_inputs, n_b, n_m, n_files;     // n_files: output files per sample
    uint32_t n_samples, barcode_len;
    uint32_t fixed_bc_len, variable_bc;       // sample-barcode layout: sum of the fixed B segments; a '+B' exists
    uint32_t skip_short, use_lens;
    uint32_t no_carry;                        // every chunk ends its blocks (several devices): nothing lives in the persistent slabs
    uint32_t min_len[FQTK_DEMUX_MAX_INPUTS];
    fmt::SegPos bseg[kMaxSegs], mseg[kMaxSegs];
    fmt::FileSeg fseg[FQTK_DEMUX_MAX_FILES];
};

struct TextSet {
    const uint8_t *text[FQTK_DEMUX_MAX_INPUTS];
    uint32_t len[FQTK_DEMUX_MAX_INPUTS];
    uint32_t *tile_cnt[FQTK_DEMUX_MAX_INPUTS];   // newlines per 4 KiB tile, then their exclusive prefix sum
    uint32_t *ls[FQTK_DEMUX_MAX_INPUTS];         // line starts: ls[k] = offset of line k; 4N + 1 entries
    RecView *rec[FQTK_DEMUX_MAX_INPUTS];
    // A chunk cut out of text that is already on the device (BGZF inputs inflated there: fqtk_demuxer_submit_fed) is a
    // WINDOW of whole members around the chunk's records: text[i] is 16-byte aligned, its first lead[i] (< 16) bytes belong
    // to something else and read as zeros, the chunk's first line is line first_line[i] of the window and lines behind its
    // last one are ignored.  Text handed over by the host: lead = first_line = 0, and the window holds exactly 4 n lines.
    uint32_t lead[FQTK_DEMUX_MAX_INPUTS];
    uint32_t first_line[FQTK_DEMUX_MAX_INPUTS];
    uint32_t window;   // 1: windows (at least first_line + 4 n lines each)
};

// Device-side status of one chunk (copied to page-locked memory at the end of the chunk).
struct ChunkStatus {
    unsigned long long err_key;       // lowest (template << 24 | class << 20 | input << 9 | stage << 8 | kind), ~0 = none
    unsigned long long matcher_err;   // the matcher's latched length error (read index), ~0 = none
    unsigned long long total_bytes;   // packed BGZF members
    uint32_t n_lines[FQTK_DEMUX_MAX_INPUTS];
    uint32_t end_off[FQTK_DEMUX_MAX_INPUTS];   // offset in the input's text of the byte behind the chunk's last record
    uint32_t n_blocks, n_skipped, max_bc_len, pad;
};
// What fqtk_demuxer_record_text needs of a chunk that failed, saved in the slot's own memory while the chunk's text is still there (fed
// chunks are windows of an input's arena, which may be reused or freed before the host gets to word the error: ADVICE r04): header and
// number of bases, in every input, of the template err_key names and of the one the matcher's length error names.
// (4096 header bytes are kept: as many as the caller's buffer of an error message takes from a chunk of host text -- ADVICE r05: 256 cut a long header short)
constexpr uint32_t kErrTextHead = 4096;
struct ErrorText { uint32_t t, head_len, seq_len, valid; uint8_t head[kErrTextHead]; };   // [2][n_inputs]: candidate 0 = err_key's template, 1 = matcher_err's
__global__ void k_save_error_text(TextSet T, uint32_t n_inputs, uint32_t n, const ChunkStatus *st, ErrorText *out) {
    const uint32_t c = blockIdx.x / n_inputs, i = blockIdx.x % n_inputs;
    const unsigned long long key = c == 0u ? st->err_key : st->matcher_err;
    ErrorText &e = out[c * n_inputs + i];
    if (key == kNoError) { if (threadIdx.x == 0) e.valid = 0; return; }
    const uint32_t t = c == 0u ? (uint32_t)(key >> 24) : (uint32_t)key;
    if (t >= n) { if (threadIdx.x == 0) e.valid = 0; return; }
    const RecView r = T.rec[i][t];
    const uint32_t hl = r.head_len < kErrTextHead ? r.head_len : kErrTextHead;
    // (a chunk whose text does not hold its lines has no record views worth reading: offsets are checked against the text)
    const bool ok = (uint64_t)r.head_off + hl <= (uint64_t)T.len[i];
    for (uint32_t k = threadIdx.x; k < hl; k += blockDim.x) e.head[k] = ok ? T.text[i][r.head_off + k] : (uint8_t)'?';
    if (threadIdx.x == 0) { e.t = t; e.head_len = hl; e.seq_len = r.seq_len; e.valid = 1; }
}

// Order of the errors of ONE template = the order the reference meets them in: its per-input iterators are zipped
// (demux.rs:285-343, 946-951), so input 0's record is parsed AND length-checked before input 1's is looked at -- for the
// record-level stages (0 = malformed record, 1 = too few bases) the input ranks above the stage (class 0); the
// matcher's length error (class 2, kept in matcher_err) and the header errors (class 3) follow, then the internal ones.
__device__ inline unsigned long long error_key(uint32_t t, uint32_t stage, uint32_t input, uint32_t kind) {
    const uint32_t cls = stage <= 1u ? 0u : stage;
    return ((unsigned long long)t << 24) | ((unsigned long long)cls << 20) | ((unsigned long long)input << 9) |
           ((unsigned long long)(stage <= 1u ? stage : 0u) << 8) | kind;
}
__device__ inline void report(ChunkStatus *st, uint32_t t, uint32_t stage, uint32_t input, uint32_t kind) {
    atomicMin(&st->err_key, error_key(t, stage, input, kind));
}

// the line index did not come out as 4 lines per template: k_records wrote nothing, nothing downstream may run
__device__ inline bool no_records(const ChunkStatus *st) {
    const unsigned long long e = st->err_key;
    return e != kNoError && (e >> 20) == 0 && (e & 0xFFu) == FQTK_DEMUX_ERR_LINES;
}

// Per output file (column c = sample * n_files + f), kept across chunks.
// A file's open (partly filled) block lives in one of kPersist slabs of its own; when a chunk closes it, the next open
// block starts in the next slab of the rotation -- so the compressor may still be reading the closed one (and the one
// closed a chunk earlier) while the next chunk is being formatted.
struct FileState {
    uint32_t rem;   // bytes in the file's open block
    uint32_t par;   // which persistent slab holds it
};
__host__ __device__ inline uint32_t next_slab(uint32_t par) { return par + 1u == kPersist ? 0u : par + 1u; }
// Per output file, for one chunk.
struct FileChunk {
    uint32_t rem;        // open bytes when the chunk started
    uint32_t nb;         // blocks this chunk closes (65 280 bytes each)
    uint32_t n_emit;     // blocks it hands to the compressor: nb, + 1 when the open rest is flushed
    uint32_t blk_base;   // index of its first block in the chunk's block list
    uint32_t slab_base;  // first of its nb - 1 chunk slabs
    uint32_t par;        // persistent slab of the block open at the start (the next open one: next_slab(par))
    uint32_t new_rem;    // open bytes after the chunk (before a flush empties them)
    uint32_t chunk_only; // DevConfig::no_carry: all n_emit blocks live in the chunk's own slabs, slab_base + k
};

// Where byte `q` (counted from the start of the block that was open when the chunk began) of file `c` lives.
__device__ inline uint8_t *file_byte(const FileChunk &fc, uint32_t c, uint32_t q, uint8_t *persist, uint8_t *slabs) {
    const uint32_t kth = q / kBlock, r = q - kth * kBlock;
    uint8_t *base;
    if (fc.chunk_only) base = slabs + (size_t)(fc.slab_base + kth) * kSlab;
    else if (kth == 0) base = persist + ((size_t)c * kPersist + fc.par) * kSlab;
    else if (kth < fc.nb) base = slabs + (size_t)(fc.slab_base + kth - 1) * kSlab;
    else base = persist + ((size_t)c * kPersist + next_slab(fc.par)) * kSlab;
    return base + r;
}
__device__ inline uint8_t *block_base(const FileChunk &fc, uint32_t c, uint32_t kth, uint8_t *persist, uint8_t *slabs) {
    if (fc.chunk_only) return slabs + (size_t)(fc.slab_base + kth) * kSlab;
    if (kth == 0) return persist + ((size_t)c * kPersist + fc.par) * kSlab;
    if (kth < fc.nb) return slabs + (size_t)(fc.slab_base + kth - 1) * kSlab;
    return persist + ((size_t)c * kPersist + next_slab(fc.par)) * kSlab;
}

// ---- line index -----------------------------------------------------------------------------------------------------
// 0x80 in every byte of w that is '\n' (exact: no borrow across bytes)
__device__ inline uint32_t newline_mask(uint32_t w) {
    const uint32_t x = w ^ 0x0A0A0A0Au;
    const uint32_t t = (x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu;
    return ~(t | x | 0x7F7F7F7Fu);
}
// the 16 bytes of lane `lane` of tile `tile`, bytes at or past `len` read as zeros
__device__ inline uint4 tile_bytes(const uint8_t *text, uint32_t len, uint32_t tile, uint32_t lane, uint32_t lead = 0) {
    const uint32_t off = tile * kLineTile + lane * 16u;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (off < lead) {   // (lead < 16: the first lane of the first tile) the window's first bytes are not its own
        uint32_t w[4] = {0, 0, 0, 0};
        for (uint32_t k = lead; k < 16u && off + k < len; ++k) w[k >> 2] |= (uint32_t)text[off + k] << (8 * (k & 3u));
        v = make_uint4(w[0], w[1], w[2], w[3]);
    } else if (off + 16u <= len) {
        v = *reinterpret_cast<const uint4 *>(text + off);
    } else if (off < len) {
        uint32_t w[4] = {0, 0, 0, 0};
        for (uint32_t k = 0; off + k < len; ++k) w[k >> 2] |= (uint32_t)text[off + k] << (8 * (k & 3u));
        v = make_uint4(w[0], w[1], w[2], w[3]);
    }
    return v;
}
__device__ inline uint32_t lane_newlines(const uint4 &v, uint32_t *m) {
    m[0] = newline_mask(v.x); m[1] = newline_mask(v.y); m[2] = newline_mask(v.z); m[3] = newline_mask(v.w);
    return (uint32_t)(__popc(m[0]) + __popc(m[1]) + __popc(m[2]) + __popc(m[3]));
}
__device__ inline uint32_t block_sum_256(uint32_t v, uint32_t *sh /* 8 words */) {
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    if ((threadIdx.x & 63u) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

__global__ __launch_bounds__(256) void k_count_lines(TextSet T) {
    __shared__ uint32_t sh[8];
    const uint32_t in = blockIdx.y, tile = blockIdx.x;
    if ((uint64_t)tile * kLineTile >= T.len[in]) return;
    uint32_t m[4];
    const uint32_t c = lane_newlines(tile_bytes(T.text[in], T.len[in], tile, threadIdx.x, T.lead[in]), m);
    const uint32_t tot = block_sum_256(c, sh);
    if (threadIdx.x == 0) T.tile_cnt[in][tile] = tot;
}

// exclusive prefix sum of an input's tile counts, in place; the total = its number of lines
__global__ __launch_bounds__(1024) void k_scan_tiles(TextSet T, ChunkStatus *st) {
    __shared__ uint32_t sh[2][16];
    const uint32_t in = blockIdx.x;
    const uint32_t n_tiles = (T.len[in] + kLineTile - 1) / kLineTile;
    uint32_t *a = T.tile_cnt[in];
    // (a wavefront's prefix sum by shuffles, the sixteen wavefronts' totals through LDS -- two sets in turn, so ONE barrier per 1024 tiles; the
    //  doubling scan over LDS took twenty-two, and this kernel is one workgroup per input: all latency)
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t carry = 0;   // (the same in every lane)
    for (uint32_t base = 0, round = 0; base < n_tiles; base += 1024, round ^= 1u) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = a[i < n_tiles ? i : n_tiles - 1u] * (i < n_tiles ? 1u : 0u);
        uint32_t incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = __shfl_up(incl, d);
            if (lane >= (uint32_t)d) incl += up;
        }
        if (lane == 63u) sh[round][wave] = incl;
        __syncthreads();
        uint32_t before_waves = 0, total = 0;
#pragma unroll
        for (uint32_t w = 0; w < 16u; ++w) {
            const uint32_t tw = sh[round][w];
            before_waves += w < wave ? tw : 0u;
            total += tw;
        }
        if (i < n_tiles) a[i] = carry + before_waves + incl - v;
        carry += total;
    }
    if (threadIdx.x == 0) st->n_lines[in] = carry;
}

// ls[k + 1] = offset of the byte after newline k (k < max_lines)
__global__ __launch_bounds__(256) void k_line_starts(TextSet T, uint32_t max_lines) {
    __shared__ uint32_t sh[4];
    const uint32_t in = blockIdx.y, tile = blockIdx.x;
    if ((uint64_t)tile * kLineTile >= T.len[in]) return;
    uint32_t m[4];
    const uint32_t c = lane_newlines(tile_bytes(T.text[in], T.len[in], tile, threadIdx.x, T.lead[in]), m);
    // inclusive prefix sum of the 256 lanes' counts: by shuffles inside a wavefront, the four wavefronts' totals through LDS (one barrier; the doubling
    // scan over LDS took sixteen)
    uint32_t incl = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = __shfl_up(incl, d);
        if ((threadIdx.x & 63u) >= (uint32_t)d) incl += up;
    }
    if ((threadIdx.x & 63u) == 63u) sh[threadIdx.x >> 6] = incl;
    __syncthreads();
    for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) incl += sh[w];
    // newline number k of the window ends the chunk's line k - first_line (the window's lines before first_line and
    // behind the chunk's last one belong to other chunks)
    const uint32_t first = T.first_line[in];
    uint32_t k = T.tile_cnt[in][tile] + incl - c;
    uint32_t *ls = T.ls[in];
    if (tile == 0 && threadIdx.x == 0 && first == 0u) ls[0] = T.lead[in];
    const uint32_t off = tile * kLineTile + threadIdx.x * 16u;
    for (int w = 0; w < 4; ++w) {
        uint32_t mm = m[w];
        while (mm) {
            const uint32_t bit = (uint32_t)__ffs((int)mm) - 1u;   // 7, 15, 23 or 31
            mm &= mm - 1u;
            if (k + 1u >= first && k + 1u - first <= max_lines) ls[k + 1u - first] = off + (uint32_t)w * 4u + (bit >> 3) + 1u;
            ++k;
        }
    }
}

// ---- records ----------------------------------------------------------------------------------------------------------
// One lane per template: the four lines of its record in every input (seq_io's checks: '@', '+', equal lengths; a
// trailing '\r' is not part of a line), the too-few-bases rule (demux.rs:298-313), the length of its sample barcode.
__global__ __launch_bounds__(256) void k_records(TextSet T, DevConfig C, uint32_t n, uint8_t *skip, uint32_t *bc_len,
                                                 ChunkStatus *st) {
    for (uint32_t i = 0; i < C.n_inputs; ++i)
        if (T.window ? st->n_lines[i] < T.first_line[i] + 4u * n : st->n_lines[i] != 4u * n) {   // the index is not the one the later kernels expect: nothing is touched
            if (blockIdx.x == 0 && threadIdx.x == 0) report(st, 0, 0, i, FQTK_DEMUX_ERR_LINES);
            return;
        }
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= n) return;
    if (t == n - 1u)
        for (uint32_t i = 0; i < C.n_inputs; ++i) st->end_off[i] = T.ls[i][4u * n];
    bool too_short = false;
    for (uint32_t i = 0; i < C.n_inputs; ++i) {
        const uint8_t *x = T.text[i];
        const uint32_t *ls = T.ls[i] + 4u * t;
        const uint32_t s0 = ls[0], s1 = ls[1], s2 = ls[2], s3 = ls[3], s4 = ls[4];
        // [a, b - 1) without the newline, without a trailing '\r'.  The six bytes looked at -- the four lines' last ones, the '@' and the '+' -- are
        // read TOGETHER and unconditionally (an empty line reads its own newline): `if (l && x[...] == '\r')` is a branch around a read with a
        // wait of its own, and six of those per input were six dependent round trips to the L2 per template.
        const uint32_t l0 = s1 - 1u - s0, l1 = s2 - 1u - s1, l2 = s3 - 1u - s2, l3 = s4 - 1u - s3;
        const uint8_t e0 = x[s0 + (l0 ? l0 - 1u : 0u)], e1 = x[s1 + (l1 ? l1 - 1u : 0u)], e2 = x[s2 + (l2 ? l2 - 1u : 0u)], e3 = x[s3 + (l3 ? l3 - 1u : 0u)];
        const uint8_t at = x[s0], plus = x[s2];
        const uint32_t hl = l0 - (l0 && e0 == '\r' ? 1u : 0u), ql = l3 - (l3 && e3 == '\r' ? 1u : 0u), pl = l2 - (l2 && e2 == '\r' ? 1u : 0u);
        uint32_t sl = l1 - (l1 && e1 == '\r' ? 1u : 0u);
        if (hl == 0 || at != '@') report(st, t, 0, i, FQTK_DEMUX_ERR_NO_AT);
        else if (pl == 0 || plus != '+') report(st, t, 0, i, FQTK_DEMUX_ERR_NO_PLUS);
        else if (sl != ql) report(st, t, 0, i, FQTK_DEMUX_ERR_QUAL_LEN);
        if (ql < sl) sl = ql;   // (an error anyway; keeps every later access inside both lines)
        RecView r;
        r.head_off = s0 + 1u;
        r.head_len = hl ? hl - 1u : 0u;
        r.seq_off = s1;
        r.seq_len = sl;
        r.qual_off = s3;
        T.rec[i][t] = r;
        if (sl < C.min_len[i]) {
            too_short = true;
            if (!C.skip_short) report(st, t, 1, i, FQTK_DEMUX_ERR_TOO_SHORT);
        }
    }
    skip[t] = too_short ? 1 : 0;
    if (too_short) atomicAdd(&st->n_skipped, 1u);
    uint32_t bl = 0;
    if (!too_short)
        for (uint32_t b = 0; b < C.n_b; ++b) {
            uint32_t lo, hi;
            fmt::segment_span(C.bseg[b].offset, C.bseg[b].length, T.rec[C.bseg[b].input][t].seq_len, &lo, &hi);
            bl += hi - lo;
        }
    bc_len[t] = bl;
    if (C.variable_bc) atomicMax(&st->max_bc_len, bl);
}

// The sample barcode of every template = its B segments side by side (demux.rs:121-123), one row of `stride` bytes.
// A dropped template gets an empty row (length 0: shorter than any barcode -> no match, no length error).
__global__ __launch_bounds__(256) void k_extract(TextSet T, DevConfig C, uint32_t n, uint32_t stride, const uint8_t *skip,
                                                 const uint32_t *bc_len, uint8_t *obs, uint32_t *lens, const ChunkStatus *st) {
    if (no_records(st)) return;
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= n) return;
    uint8_t *row = obs + (size_t)t * stride;
    uint32_t w = 0;
    if (!skip[t])
        for (uint32_t b = 0; b < C.n_b; ++b) {
            const RecView r = T.rec[C.bseg[b].input][t];
            uint32_t lo, hi;
            fmt::segment_span(C.bseg[b].offset, C.bseg[b].length, r.seq_len, &lo, &hi);
            const uint8_t *src = T.text[C.bseg[b].input] + r.seq_off;
            for (uint32_t k = lo; k < hi && w < stride; ++k) row[w++] = src[k];
        }
    for (uint32_t k = w; k < stride; ++k) row[k] = 0;
    if (lens) lens[t] = skip[t] ? 0u : bc_len[t];
}

// ---- placement ----------------------------------------------------------------------------------------------------------
// Every output file is a byte stream in input order (the reference writes templates one after another).  A template's
// record lands at (bytes of earlier templates of the same sample) in each of that sample's files: a prefix sum keyed by
// sample, stable in the template order.  Per tile of 1024 templates each lane adds up the lengths of the lanes before
// it that share its sample (an O(tile) loop over LDS: ~6 k lane-operations per template, nothing next to the text
// traffic) and the tile's total per (sample, file) goes into a [tiles x columns] matrix whose columns are then summed
// down the tiles (k_column_scan).  Column layout: sample * (n_files + 1) + f, the last one counts templates.
struct TemplatePlan {
    fmt::HeaderPlan h;
    uint32_t base_len;   // bytes of a record of this template with a one-digit read number and an empty segment
};

__device__ inline uint32_t digits_of(uint32_t v) { uint32_t n = 1; while (v >= 10u) { v /= 10u; ++n; } return n; }

__global__ __launch_bounds__(1024) void k_plan_rank(TextSet T, DevConfig C, uint32_t n, const uint32_t *res, const uint8_t *skip,
                                                    TemplatePlan *plans, uint32_t *rec_off /* [n_files][n] */,
                                                    uint32_t *tile_tot /* [tiles][cols] */, ChunkStatus *st) {
    if (no_records(st)) return;
    __shared__ __attribute__((aligned(16))) uint32_t s_key[kTile];
    __shared__ uint32_t s_len[kRankGroup][kTile];
    const uint32_t i = threadIdx.x, tile = blockIdx.x, t = tile * kTile + i;
    const uint32_t cps = C.n_files + 1u, cols = (C.n_samples + 1u) * cps;
    uint32_t key = 0xFFFFFFFFu;
    TemplatePlan tp;
    tp.base_len = 0;
    if (t < n && !skip[t]) {
        const uint32_t idx = res[t] & 0xFFFFu;
        key = idx == FQTK_NO_MATCH ? C.n_samples : idx;
        const RecView r0 = T.rec[0][t];
        tp.h = fmt::plan_header(T.text[0] + r0.head_off, r0.head_len, C.n_m != 0);
        if (tp.h.err) report(st, t, 3, 0, tp.h.err);
        // everything of the record but the read number's extra digits and the segment itself
        uint32_t bl = 0, ml = 0;
        for (uint32_t b = 0; b < C.n_b; ++b) {
            uint32_t lo, hi;
            fmt::segment_span(C.bseg[b].offset, C.bseg[b].length, T.rec[C.bseg[b].input][t].seq_len, &lo, &hi);
            bl += hi - lo;
        }
        for (uint32_t b = 0; b < C.n_m; ++b) {
            uint32_t lo, hi;
            fmt::segment_span(C.mseg[b].offset, C.mseg[b].length, T.rec[C.mseg[b].input][t].seq_len, &lo, &hi);
            ml += hi - lo;
        }
        const uint32_t len = fmt::record_len(tp.h, 1u, bl, C.n_b, ml, C.n_m, 0u);
        tp.base_len = len;
        plans[t] = tp;
    }
    s_key[i] = key;
    uint32_t cnt_before = 0, cnt_tot = 0;
    for (uint32_t f0 = 0; f0 < C.n_files; f0 += kRankGroup) {
        const uint32_t g = C.n_files - f0 < kRankGroup ? C.n_files - f0 : kRankGroup;
        __syncthreads();
        uint32_t mine[kRankGroup] = {0, 0};
        if (key != 0xFFFFFFFFu) {
#pragma unroll
            for (uint32_t f = 0; f < kRankGroup; ++f) {
                if (f >= g) continue;
                const fmt::FileSeg fs = C.fseg[f0 + f];
                uint32_t lo, hi;
                fmt::segment_span(fs.offset, fs.length, T.rec[fs.input][t].seq_len, &lo, &hi);
                mine[f] = tp.base_len + (tp.h.kind != 1 ? digits_of(fs.read_num) - 1u : 0u) + 2u * (hi - lo);
            }
        }
#pragma unroll
        for (uint32_t f = 0; f < kRankGroup; ++f) s_len[f][i] = mine[f];
        __syncthreads();
        uint32_t before[kRankGroup] = {0, 0}, tot[kRankGroup] = {0, 0};
        cnt_before = 0;
        cnt_tot = 0;
        // Eight templates' keys per round (two 16-byte reads, the same address in every lane), then only the lanes that have their own key among
        // them read lengths -- a template shares its sample with two or three of the tile's 1024.  (One key per round was a dependent LDS round trip per
        // template of the tile, 1024 in a row; everything read unconditionally and summed by selection was as slow: ~15 k vector instructions per wavefront.)
        if (key != 0xFFFFFFFFu)
            for (uint32_t j = 0; j < kTile; j += 8u) {
                const uint4 ka = *reinterpret_cast<const uint4 *>(s_key + j), kb = *reinterpret_cast<const uint4 *>(s_key + j + 4u);
                uint32_t hit = (ka.x == key ? 1u : 0u) | (ka.y == key ? 2u : 0u) | (ka.z == key ? 4u : 0u) | (ka.w == key ? 8u : 0u) |
                               (kb.x == key ? 16u : 0u) | (kb.y == key ? 32u : 0u) | (kb.z == key ? 64u : 0u) | (kb.w == key ? 128u : 0u);
                while (hit) {
                    const uint32_t jj = j + (uint32_t)__ffs((int)hit) - 1u;
                    hit &= hit - 1u;
                    const bool b4 = jj < i;
                    cnt_tot += 1u;
                    cnt_before += b4 ? 1u : 0u;
#pragma unroll
                    for (uint32_t f = 0; f < kRankGroup; ++f) {
                        const uint32_t l = s_len[f][jj];
                        tot[f] += l;
                        before[f] += b4 ? l : 0u;
                    }
                }
            }
        if (key != 0xFFFFFFFFu) {
            const bool last = cnt_before + 1u == cnt_tot;
#pragma unroll
            for (uint32_t f = 0; f < kRankGroup; ++f) {
                if (f >= g) continue;
                rec_off[(size_t)(f0 + f) * n + t] = before[f];
                if (last) tile_tot[(size_t)tile * cols + key * cps + f0 + f] = tot[f];
            }
        }
    }
    if (C.n_files == 0) {   // (no output files: the templates are still counted)
        __syncthreads();
        cnt_before = 0;
        cnt_tot = 0;
        if (key != 0xFFFFFFFFu)
            for (uint32_t j = 0; j < kTile; ++j) if (s_key[j] == key) { ++cnt_tot; cnt_before += j < i ? 1u : 0u; }
    }
    if (key != 0xFFFFFFFFu && cnt_before + 1u == cnt_tot) tile_tot[(size_t)tile * cols + key * cps + C.n_files] = cnt_tot;
}

// exclusive prefix sum down the tiles of every column, in place; the column totals of the chunk
__global__ __launch_bounds__(256) void k_column_scan(uint32_t *tile_tot, uint32_t n_tiles, uint32_t cols, uint32_t *chunk_tot) {
    const uint32_t c = blockIdx.x * 256u + threadIdx.x;
    if (c >= cols) return;
    // (eight tiles' values are read before any is written back: read, write, read in place is one memory round trip per tile -- 256 in a row, 61 us
    //  of a chunk for 1155 lanes' worth of additions)
    uint32_t run = 0;
    for (uint32_t tl = 0; tl < n_tiles; tl += 8u) {
        uint32_t v[8];
#pragma unroll
        for (uint32_t u = 0; u < 8u; ++u) v[u] = tile_tot[(size_t)(tl + u < n_tiles ? tl + u : n_tiles - 1u) * cols + c];
#pragma unroll
        for (uint32_t u = 0; u < 8u; ++u)
            if (tl + u < n_tiles) {
                tile_tot[(size_t)(tl + u) * cols + c] = run;
                run += v[u];
            }
    }
    chunk_tot[c] = run;
}

// One workgroup: what every file closes in this chunk, where its blocks go, and the per-sample counts.
__global__ __launch_bounds__(1024) void k_layout(DevConfig C, const uint32_t *chunk_tot, FileState *fs, FileChunk *fc,
                                                 unsigned long long *counts, uint32_t flush, uint32_t max_blocks, ChunkStatus *st) {
    __shared__ uint32_t sh_a[1024], sh_b[1024];
    __shared__ uint32_t carry_a, carry_b;
    const uint32_t cps = C.n_files + 1u, n_cols = (C.n_samples + 1u) * C.n_files;
    if (threadIdx.x == 0) { carry_a = 0; carry_b = 0; }
    __syncthreads();
    for (uint32_t base = 0; base < n_cols; base += 1024) {
        const uint32_t c = base + threadIdx.x;
        FileChunk x;
        x.rem = x.nb = x.n_emit = x.blk_base = x.slab_base = x.par = x.new_rem = 0;
        x.chunk_only = C.no_carry;
        uint32_t n_slabs = 0;
        if (c < n_cols) {
            const uint32_t s = c / C.n_files, f = c - s * C.n_files;
            const uint32_t bytes = chunk_tot ? chunk_tot[s * cps + f] : 0u;
            const FileState cur = fs[c];
            const unsigned long long end = (unsigned long long)cur.rem + bytes;
            x.rem = cur.rem;
            x.par = cur.par;
            x.nb = (uint32_t)(end / kBlock);
            x.new_rem = (uint32_t)(end - (unsigned long long)x.nb * kBlock);
            x.n_emit = x.nb + ((flush && x.new_rem) ? 1u : 0u);
            // (no_carry: a file's persistent slabs would be written by chunk k + 1 while the compressor still reads chunk
            //  k's blocks from them -- two slabs per chunk against a rotation of three, ADVICE r03 -- and nothing has to
            //  persist when every chunk is flushed: the blocks live with the chunk)
            n_slabs = C.no_carry ? x.n_emit : (x.nb ? x.nb - 1u : 0u);
            FileState nx;
            nx.rem = flush ? 0u : x.new_rem;
            // the block open after the chunk sits in the other slab once a block was closed; a flush hands it to the
            // compressor, and the (empty) block after it starts in the slab that is free by then
            nx.par = x.nb ? next_slab(cur.par) : cur.par;
            if (flush && x.new_rem) nx.par = next_slab(nx.par);
            fs[c] = nx;
        }
        sh_a[threadIdx.x] = x.n_emit;
        sh_b[threadIdx.x] = n_slabs;
        __syncthreads();
        for (uint32_t d = 1; d < 1024; d <<= 1) {
            const uint32_t a = threadIdx.x >= d ? sh_a[threadIdx.x - d] : 0, b = threadIdx.x >= d ? sh_b[threadIdx.x - d] : 0;
            __syncthreads();
            sh_a[threadIdx.x] += a;
            sh_b[threadIdx.x] += b;
            __syncthreads();
        }
        const uint32_t ca = carry_a, cb = carry_b;
        if (c < n_cols) {
            x.blk_base = ca + sh_a[threadIdx.x] - x.n_emit;
            x.slab_base = cb + sh_b[threadIdx.x] - n_slabs;
            fc[c] = x;
        }
        __syncthreads();
        if (threadIdx.x == 1023) { carry_a = ca + sh_a[1023]; carry_b = cb + sh_b[1023]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        // (the host sized the slabs and descriptors for max_blocks: more than that cannot be -- its bound covers every byte
        //  a record can hold -- but nothing downstream may run past the buffers if it ever were)
        if (carry_a > max_blocks) { report(st, 0, 4, 0, FQTK_DEMUX_ERR_LINES); carry_a = 0; }
        st->n_blocks = carry_a;
        FileChunk end;   // sentinel: fc[n_cols].blk_base = number of blocks
        end.rem = end.nb = end.n_emit = end.slab_base = end.par = end.new_rem = end.chunk_only = 0;
        end.blk_base = carry_a;
        fc[n_cols] = end;
    }
    if (chunk_tot)
        for (uint32_t s = threadIdx.x; s <= C.n_samples; s += 1024) counts[s] += chunk_tot[s * cps + C.n_files];
}

// Block descriptors for the compressor: block j of the chunk belongs to the file whose [blk_base, blk_base + n_emit)
// holds j.
__global__ __launch_bounds__(256) void k_descs(DevConfig C, const FileChunk *fc, uint8_t *persist, uint8_t *slabs, uint8_t *out_slabs,
                                               fqtk_bgzf_block *desc, uint32_t *blk_file, const ChunkStatus *st) {
    const uint32_t j = blockIdx.x * 256u + threadIdx.x;
    if (j >= st->n_blocks) return;
    const uint32_t n_cols = (C.n_samples + 1u) * C.n_files;
    uint32_t lo = 0, hi = n_cols;   // last column with blk_base <= j (columns without blocks share their successor's base)
    while (hi - lo > 1u) {
        const uint32_t mid = (lo + hi) >> 1;
        if (fc[mid].blk_base <= j) lo = mid; else hi = mid;
    }
    const FileChunk x = fc[lo];
    const uint32_t k = j - x.blk_base;
    fqtk_bgzf_block d;
    d.in = block_base(x, lo, k, persist, slabs);
    d.out = out_slabs + (size_t)j * kSlab;
    d.n_in = k < x.nb ? kBlock : x.new_rem;
    d.reserved = 0;
    desc[j] = d;
    blk_file[j] = lo;
}

// ---- formatting -----------------------------------------------------------------------------------------------------------
// One wavefront per kFormatGroup = 16 consecutive templates; per output file their sixteen records are worked on TOGETHER wherever the
// work is the same for all of them:
//   A. the SLOT TABLES (record_format.hpp: record_slot_with) of all sixteen records at once -- a lane per (record, slot): the slot in
//      closed form from the template's plan, a prefix sum over the lanes of one record places it; tables in LDS (start, source, kind);
//   B. the BODY: a lane per dword of ANY of the sixteen records (their dwords counted through: a search over sixteen prefix sums says
//      whose a lane's is) -- dwords of a file's block that lie inside ONE span slot: a binary search over the record's slot starts, two
//      aligned dwords of the source, one v_alignbyte; two dwords per lane and pass, loads before stores, no pass depends on another.
//      Dwords that are NOT of that kind -- they hold a literal, straddle two slots, or the record begins / ends inside them: about
//      one in six -- are only LISTED (ballot, a queue in LDS);
//   C. the SEAMS of all sixteen records, a lane per listed dword, byte by byte.
// Round 4 did all three per record -- passes in which sixteen to twenty-five lanes had something to do, and a chain of dependent LDS and HBM
// round trips per record that four waves per SIMD did not hide: 549 VALU + 416 SALU wave-instructions per record, 672 us per chunk of
// 262 144 templates (the second-largest device stage of a run); round 3 offered every
// piece to all lanes (740 us); before that one lane listed the pieces (805 us).
// Records of more slots than sixteen (more than two barcode segments) go through in groups of eight or four: the tables hold 256 slots.
constexpr int kFormatWaves = 4;
constexpr uint32_t kFormatGroup = 16;
constexpr uint32_t kFormatSlots = 64;     // record_slots(nb, nm) <= 64: a lane per slot of one record at least (fqtk_demuxer_create)
constexpr uint32_t kFormatTable = 256;    // slots in the tables: 16 records x 16 slots, 8 x 32 or 4 x 64
#ifndef FQTK_FORMAT_QUEUE
#define FQTK_FORMAT_QUEUE 512
#endif
constexpr uint32_t kSeamQueue = FQTK_FORMAT_QUEUE;      // listed dwords (flushed when fewer than 128 places are free)
struct RecParam {                         // where a record of the batch in hand goes (phase C: a lane per listed dword looks its record up)
    unsigned long long base0, base1;      // the block its first byte lies in, and the next one
    uint32_t in0a;                        // offset in that block of the dword-aligned start of its first dword
    uint32_t a, total, flags;             // bytes in front of it in that dword; its length; 1: "<n>:N:0:" (not "<n>:"), 2: two blocks at most
    uint32_t q, c, nb, slab_base, par, pad;   // (records longer than two blocks: file_byte's general arithmetic)
};
struct WaveScratch {
    uint32_t planw[kFormatGroup][4];      // the templates' header plans
    uint32_t start[kFormatTable + 4 * kFormatGroup];   // [record][P + 4] (rows 16-byte aligned): byte offset of every slot in its record; [P] = the record's length
    uint32_t src[kFormatTable];           // span: offset in its input's text; literal: the bytes
    uint32_t info[kFormatTable];          // input | kind << 16
    RecParam rp[kFormatGroup];
    uint32_t queue[kSeamQueue];           // record of the batch << 24 | dword of its block range
};
// dynamic LDS: the inputs' text pointers and the barcode segments' places (one copy per workgroup), then per wave: WaveScratch, RecView rec[n_inputs][kFormatGroup]
constexpr size_t kFormatBlockHead = FQTK_DEMUX_MAX_INPUTS * sizeof(uint64_t) + 2 * kMaxSegs * sizeof(fmt::SegPos);
__host__ __device__ constexpr size_t format_wave_bytes(uint32_t n_inputs) { return (sizeof(WaveScratch) + (size_t)n_inputs * kFormatGroup * sizeof(RecView) + 15u) & ~(size_t)15u; }
__host__ __device__ constexpr size_t format_block_bytes(uint32_t n_inputs) { return ((kFormatBlockHead + 15u) & ~(size_t)15u) + kFormatWaves * format_wave_bytes(n_inputs); }

#ifndef FQTK_FORMAT_OCC   // registers capped so that that many wavefronts fit a SIMD (tools/ab_format.sh: 3 -> 4 took the kernel from 523 to 408 us; 5 and 6 spill)
#define FQTK_FORMAT_OCC 4
#endif
#define FQTK_FORMAT_OCCUPANCY __attribute__((amdgpu_waves_per_eu(FQTK_FORMAT_OCC, 8)))
__global__ __launch_bounds__(64 * kFormatWaves) FQTK_FORMAT_OCCUPANCY void k_format(TextSet T, DevConfig C, uint32_t n, const uint32_t *res, const uint8_t *skip,
                                                               const TemplatePlan *plans, const uint32_t *rec_off, const uint32_t *tile_tot,
                                                               const FileChunk *fc, uint8_t *persist, uint8_t *slabs, const ChunkStatus *st) {
    if (st->err_key != kNoError) return;
    extern __shared__ __attribute__((aligned(16))) uint8_t fmt_lds[];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    // what a lane indexes by a value of its own: the inputs' text pointers, the barcode segments' places
    const uint8_t **text_of = reinterpret_cast<const uint8_t **>(fmt_lds);
    fmt::SegPos *bseg_lds = reinterpret_cast<fmt::SegPos *>(fmt_lds + FQTK_DEMUX_MAX_INPUTS * sizeof(uint64_t)), *mseg_lds = bseg_lds + kMaxSegs;
    if (threadIdx.x < C.n_inputs) text_of[threadIdx.x] = T.text[threadIdx.x];
    if (threadIdx.x >= 64u && threadIdx.x - 64u < C.n_b) bseg_lds[threadIdx.x - 64u] = C.bseg[threadIdx.x - 64u];
    if (threadIdx.x >= 128u && threadIdx.x - 128u < C.n_m) mseg_lds[threadIdx.x - 128u] = C.mseg[threadIdx.x - 128u];
    __syncthreads();
    uint8_t *mine_lds = fmt_lds + ((kFormatBlockHead + 15u) & ~(size_t)15u) + (size_t)wave * format_wave_bytes(C.n_inputs);
    WaveScratch &W = *reinterpret_cast<WaveScratch *>(mine_lds);
    RecView *recs = reinterpret_cast<RecView *>(mine_lds + sizeof(WaveScratch));   // [input][template of the group]
    const uint32_t t0 = (blockIdx.x * kFormatWaves + wave) * kFormatGroup;
    if (t0 >= n) return;
    const uint32_t t = t0 + lane;   // lanes 0 .. kFormatGroup - 1 each hold one template of the group
    const uint32_t cps = C.n_files + 1u, cols = (C.n_samples + 1u) * cps, tile = t0 / kTile;   // (the group divides the tile)
    const bool valid = lane < kFormatGroup && t < n && !skip[t];
    uint32_t s = 0;
    if (lane < kFormatGroup) {
        TemplatePlan tp;
        tp.h.name_len = tp.h.copy_off = tp.h.copy_len = 0;
        tp.h.kind = tp.h.tail = tp.h.msep = tp.h.err = 0;
        if (valid) {
            const uint32_t idx = res[t] & 0xFFFFu;
            s = idx == FQTK_NO_MATCH ? C.n_samples : idx;
            tp = plans[t];
        }
        W.planw[lane][0] = tp.h.name_len;
        W.planw[lane][1] = tp.h.copy_off;
        W.planw[lane][2] = tp.h.copy_len;
        W.planw[lane][3] = (uint32_t)tp.h.kind | ((uint32_t)tp.h.tail << 8) | ((uint32_t)tp.h.msep << 16);
    }
    // the record views of the group: lane = (input, template) pairs, so that the loads go out side by side
    for (uint32_t k = lane; k < C.n_inputs * kFormatGroup; k += 64u) {
        const uint32_t i = k / kFormatGroup, g = k - i * kFormatGroup;
        RecView v;
        v.head_off = v.head_len = v.seq_off = v.seq_len = v.qual_off = 0;
        if (t0 + g < n) v = T.rec[i][t0 + g];
        recs[k] = v;
    }
    const uint64_t live = __ballot(valid);
    if (!live) return;
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    const uint32_t n_slots = fmt::record_slots(C.n_b, C.n_m);
    const uint32_t logP = n_slots <= 16u ? 4u : (n_slots <= 32u ? 5u : 6u), P = 1u << logP, R = kFormatTable >> logP, PS = P + 4u;   // slots per record in the tables; records per batch; a record's row of starts
    const uint8_t *safe = text_of[0];   // (what a lane with nothing to load loads)
    for (uint32_t f = 0; f < C.n_files; ++f) {
        uint32_t my_q = 0, my_nb = 0, my_slab = 0, my_par = 0;   // where this lane's record of file f goes
        if (valid) {
            const FileChunk x = fc[s * C.n_files + f];
            my_q = x.rem + tile_tot[(size_t)tile * cols + s * cps + f] + rec_off[(size_t)f * n + t];
            my_nb = x.nb;
            my_slab = x.slab_base;
            my_par = x.par;
        }
        const fmt::FileSeg fsg = C.fseg[f];
        // "<n>:" and "<n>:N:0:" of this file (wave-uniform; which one a record takes depends on its header)
        uint32_t num0[4], num2[4];
        const uint32_t num0_len = fmt::number_literal(fsg.read_num, 0, num0), num2_len = fmt::number_literal(fsg.read_num, 2, num2);
        auto number_byte = [&](bool number0, uint32_t k) -> uint32_t {
            const uint32_t w = number0 ? (k < 4 ? num0[0] : (k < 8 ? num0[1] : (k < 12 ? num0[2] : num0[3])))
                                       : (k < 4 ? num2[0] : (k < 8 ? num2[1] : (k < 12 ? num2[2] : num2[3])));
            return (w >> (8u * (k & 3u))) & 0xFFu;
        };
        // the last slot of record rr that starts at or before rq (rq < the record's length): empty slots share their successor's start and lose
        auto slot_at = [&](uint32_t rr, uint32_t rq) -> uint32_t {
            const uint32_t *st0 = W.start + rr * PS;
            uint32_t p = 0;
            for (uint32_t step = P >> 1; step; step >>= 1)
                if (st0[p + step] <= rq) p += step;
            return p;
        };
        uint32_t n_queued = 0;   // wave-uniform
        // ---- C. the listed dwords, a lane each: byte by byte -- literals, slot boundaries, the ragged ends of a record
        auto flush_seams = [&]() {
            __builtin_amdgcn_wave_barrier();
            __threadfence_block();
            for (uint32_t g0 = 0; g0 < n_queued; g0 += 64u) {
                const bool mine = g0 + lane < n_queued;
                const uint32_t e = mine ? W.queue[g0 + lane] : 0u;
                const uint32_t rr = e >> 24, sd = e & 0xFFFFFFu;
                const RecParam rp = W.rp[rr];
                const uint32_t a = rp.a, total = rp.total;
                const bool number0 = (rp.flags & 1u) != 0u;
                const uint32_t *st0 = W.start + rr * PS;
                const uint32_t tb = rr << logP;
                const uint8_t *sbyte[4];
                uint32_t slit[4], shave = 0;
                {
                    uint32_t p = slot_at(rr, mine && 4u * sd >= a ? 4u * sd - a : 0u);
#pragma unroll
                    for (uint32_t k = 0; k < 4u; ++k) {
                        const uint32_t pos = 4u * sd + k;
                        const bool in = mine && pos >= a && pos - a < total;
                        const uint32_t rq = in ? pos - a : 0u;
                        while (in && p + 1u < P && st0[p + 1u] <= rq) ++p;
                        const uint32_t inf = W.info[tb + p], kk = rq - st0[p], sv = W.src[tb + p];
                        const bool span = in && (inf >> 16) == fmt::kSpan;
                        sbyte[k] = span ? text_of[inf & 0xFFFFu] + sv + kk : safe;
                        uint32_t lit = (sv >> (8u * (kk & 3u))) & 0xFFu;
                        if ((inf >> 16) == fmt::kNumber) lit = number_byte(number0, kk & 15u);
                        slit[k] = span ? 0x100u : lit;        // 0x100: the byte comes from the text
                        if (in) shave |= 1u << k;
                    }
                }
                uint32_t sb[4];
#pragma unroll
                for (uint32_t k = 0; k < 4u; ++k) sb[k] = *sbyte[k];
                if (shave) {
                    uint32_t word = 0;
#pragma unroll
                    for (uint32_t k = 0; k < 4u; ++k) word |= ((slit[k] & 0x100u) ? sb[k] : slit[k]) << (8u * k);
                    uint8_t *dst;
                    {
                        const uint32_t o = rp.in0a + 4u * sd;   // offset inside the record's first block
                        if (rp.flags & 2u) {
                            dst = o < kBlock ? reinterpret_cast<uint8_t *>(rp.base0) + o : reinterpret_cast<uint8_t *>(rp.base1) + (o - kBlock);
                        } else {
                            FileChunk x;
                            x.rem = x.n_emit = x.blk_base = x.new_rem = 0;
                            x.chunk_only = C.no_carry;
                            x.nb = rp.nb; x.slab_base = rp.slab_base; x.par = rp.par;
                            dst = file_byte(x, rp.c, rp.q - a + 4u * sd, persist, slabs);
                        }
                    }
                    if (shave == 15u) {
                        *reinterpret_cast<uint32_t *>(dst) = word;
                    } else {
#pragma unroll
                        for (uint32_t k = 0; k < 4u; ++k)
                            if (shave & (1u << k)) dst[k] = (uint8_t)(word >> (8u * k));
                    }
                }
            }
            n_queued = 0;
            __builtin_amdgcn_wave_barrier();
        };
        for (uint32_t rb = 0; rb < kFormatGroup; rb += R) {
            const uint64_t batch = (live >> rb) & ((R < 64u ? (1ull << R) : 0ull) - 1ull);
            if (!batch) continue;
            // ---- A. the slot tables of the batch: lane = (record, slot) ----------------------------------------------------
            for (uint32_t pass = 0; pass < kFormatTable / 64u; ++pass) {
                const uint32_t idx = pass * 64u + lane, rr = idx >> logP, sl = idx & (P - 1u), r = rb + rr;
                const bool on = ((batch >> rr) & 1ull) != 0ull && sl < n_slots;
                fmt::Slot z;
                z.len = 0; z.kind = fmt::kLiteral; z.input = 0; z.off = 0; z.lit = 0;
                if (on) {
                    fmt::HeaderPlan h;
                    h.name_len = W.planw[r][0];
                    h.copy_off = W.planw[r][1];
                    h.copy_len = W.planw[r][2];
                    const uint32_t w3 = W.planw[r][3];
                    h.kind = (uint8_t)w3;
                    h.tail = (uint8_t)(w3 >> 8);
                    h.msep = (uint8_t)(w3 >> 16);
                    h.err = 0;
                    auto seg_span = [&](const fmt::SegPos &sp) -> fmt::Span {
                        const RecView v = recs[sp.input * kFormatGroup + r];
                        uint32_t lo, hi;
                        fmt::segment_span(sp.offset, sp.length, v.seq_len, &lo, &hi);
                        return fmt::Span{sp.input, v.seq_off + lo, hi - lo};
                    };
                    const RecView v = recs[fsg.input * kFormatGroup + r];
                    uint32_t lo, hi;
                    fmt::segment_span(fsg.offset, fsg.length, v.seq_len, &lo, &hi);
                    z = fmt::record_slot_with(sl, h, recs[r].head_off, h.kind == 0 ? num0_len : num2_len,
                                              [&](uint32_t i) { return seg_span(bseg_lds[i]); }, C.n_b, [&](uint32_t i) { return seg_span(mseg_lds[i]); }, C.n_m,
                                              fmt::Span{fsg.input, v.seq_off + lo, hi - lo}, fmt::Span{fsg.input, v.qual_off + lo, hi - lo});
                }
                uint32_t incl = z.len;   // inclusive prefix sum over the lanes of one record
                for (uint32_t d = 1; d < P; d <<= 1) {
                    const uint32_t up = __shfl_up(incl, d);
                    if (sl >= d) incl += up;
                }
                W.start[rr * PS + sl] = incl - z.len;
                if (sl == P - 1u) W.start[rr * PS + P] = incl;
                W.src[idx] = z.kind == fmt::kSpan ? z.off : z.lit;
                W.info[idx] = z.input | (z.kind << 16);
            }
            __builtin_amdgcn_wave_barrier();
            __threadfence_block();
            // ---- B. the bodies of the batch: a lane per dword of any of its records -----------------------------------------
            // where the records go (the lanes that hold their templates work it out, each for its own), and how many dwords each touches
            uint32_t my_ndw = 0;
            if (lane >= rb && lane < rb + R && ((live >> lane) & 1ull)) {
                const uint32_t rr = lane - rb;
                const uint32_t c = s * C.n_files + f, q = my_q;
                FileChunk x;
                x.rem = x.n_emit = x.blk_base = x.new_rem = 0;
                x.chunk_only = C.no_carry;
                x.nb = my_nb; x.slab_base = my_slab; x.par = my_par;
                const uint32_t total = W.start[rr * PS + P];
                const uint32_t a = q & 3u;                    // the record starts `a` bytes into a dword of its block
                // the record begins in block q / kBlock of its file's chunk and may run on into the next one (records longer than a
                // block take file_byte's general arithmetic)
                const uint32_t kth0 = q / kBlock, in0 = q - kth0 * kBlock;
                RecParam rp;
                rp.base0 = reinterpret_cast<unsigned long long>(block_base(x, c, kth0, persist, slabs));
                rp.base1 = reinterpret_cast<unsigned long long>(block_base(x, c, kth0 + 1u, persist, slabs));
                rp.in0a = in0 - a;
                rp.a = a;
                rp.total = total;
                rp.flags = ((W.planw[lane][3] & 0xFFu) == 0u ? 1u : 0u) | (in0 + total <= 2u * kBlock ? 2u : 0u);
                rp.q = q; rp.c = c; rp.nb = x.nb; rp.slab_base = x.slab_base; rp.par = x.par; rp.pad = 0;
                W.rp[rr] = rp;
                my_ndw = total ? (a + total + 3u) >> 2 : 0u;  // dwords of the block it touches
            }
            uint32_t dw_incl = my_ndw;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t up = __shfl_up(dw_incl, d);
                if (lane >= (uint32_t)d) dw_incl += up;
            }
            const uint32_t dw_total = (uint32_t)__builtin_amdgcn_readlane((int)dw_incl, 63);
            // first dword of every record of the batch among the batch's dwords: wave-uniform, kept in scalar registers
            uint32_t dwb[kFormatGroup];
#pragma unroll
            for (uint32_t k = 0; k < kFormatGroup; ++k)
                dwb[k] = k < R ? (uint32_t)__builtin_amdgcn_readlane((int)(dw_incl - my_ndw), (int)(rb + (k < R ? k : 0u))) : 0xFFFFFFFFu;
            __builtin_amdgcn_wave_barrier();
            __threadfence_block();
            // a dword that lies inside one span slot is copied here; any other dword of a record is listed
            auto body_plan = [&](uint32_t g, bool &ok, bool &seam, const uint32_t *&aw, uint32_t &mis, uint8_t *&dst, uint32_t &entry) {
                // No look-up here waits for another of its kind (a wave waited 62 % of its cycles in the version that searched LDS step by
                // step): whose dword it is comes from sixteen scalar compares, its slot from ONE read of the record's sixteen starts.
                const bool have = g < dw_total;
                const uint32_t gq = have ? g : 0u;
                uint32_t rr = 0, first = 0;                   // the last record whose first dword is at or before g (empty ones share their successor's and lose)
#pragma unroll
                for (uint32_t k = 1; k < kFormatGroup; ++k) {
                    const bool at = dwb[k] <= gq;
                    rr += at ? 1u : 0u;
                    first = at ? dwb[k] : first;
                }
                const uint32_t d = gq - first;
                const RecParam &rp = W.rp[rr];
                const uint32_t a = rp.a, total = rp.total, flags = rp.flags;
                const uint32_t *st0 = W.start + rr * PS;
                const uint32_t tb = rr << logP;
                const uint32_t rq = 4u * d - a;               // record position of the dword's first byte (wraps for d = 0, a > 0)
                const bool inside = have && 4u * d >= a && rq + 4u <= total;
                const uint32_t rqs = inside ? rq : 0u;
                uint32_t p = 0;
                if (P == 16u) {                               // (wave-uniform) the last slot that starts at or before rq: counted
                    const uint4 s0 = *reinterpret_cast<const uint4 *>(st0), s1 = *reinterpret_cast<const uint4 *>(st0 + 4), s2 = *reinterpret_cast<const uint4 *>(st0 + 8),
                                s3 = *reinterpret_cast<const uint4 *>(st0 + 12);
                    p = (s0.y <= rqs) + (s0.z <= rqs) + (s0.w <= rqs) + (s1.x <= rqs) + (s1.y <= rqs) + (s1.z <= rqs) + (s1.w <= rqs) + (s2.x <= rqs) + (s2.y <= rqs) +
                        (s2.z <= rqs) + (s2.w <= rqs) + (s3.x <= rqs) + (s3.y <= rqs) + (s3.z <= rqs) + (s3.w <= rqs);
                } else {
                    p = slot_at(rr, rqs);
                }
                const uint32_t inf = W.info[tb + p], st_p = st0[p], st_p1 = st0[p + 1u], srcv = W.src[tb + p];
                ok = inside && rq + 4u <= st_p1 && (inf >> 16) == fmt::kSpan;
                seam = have && !ok;
                entry = (rr << 24) | d;
                const uint8_t *sp = ok ? text_of[inf & 0xFFFFu] + srcv + (rq - st_p) : safe;
                mis = (uint32_t)(reinterpret_cast<uintptr_t>(sp) & 3u);
                aw = reinterpret_cast<const uint32_t *>(sp - mis);
                dst = nullptr;
                if (ok) {
                    const uint32_t o = rp.in0a + 4u * d;      // offset inside the record's first block
                    if (flags & 2u) {
                        dst = o < kBlock ? reinterpret_cast<uint8_t *>(rp.base0) + o : reinterpret_cast<uint8_t *>(rp.base1) + (o - kBlock);
                    } else {
                        FileChunk x;
                        x.rem = x.n_emit = x.blk_base = x.new_rem = 0;
                        x.chunk_only = C.no_carry;
                        x.nb = rp.nb; x.slab_base = rp.slab_base; x.par = rp.par;
                        dst = file_byte(x, rp.c, rp.q - a + 4u * d, persist, slabs);
                    }
                }
            };
            auto list = [&](bool seam, uint32_t entry) {
                const uint64_t m = __ballot(seam);
                if (seam) W.queue[n_queued + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = entry;
                n_queued += (uint32_t)__popcll(m);
            };
            for (uint32_t g0 = 0; g0 < dw_total; g0 += 128u) {   // 512 bytes per pass, of whichever records; loads before stores
                bool ok0, ok1, seam0, seam1;
                const uint32_t *aw0, *aw1;
                uint32_t mis0, mis1, e0, e1;
                uint8_t *dst0, *dst1;
                body_plan(g0 + lane, ok0, seam0, aw0, mis0, dst0, e0);
                body_plan(g0 + 64u + lane, ok1, seam1, aw1, mis1, dst1, e1);
                const uint32_t w00 = aw0[0], w01 = aw0[1], w10 = aw1[0], w11 = aw1[1];   // (every text buffer has 64 bytes of slack behind it)
                list(seam0, e0);
                list(seam1, e1);
                if (ok0) *reinterpret_cast<uint32_t *>(dst0) = __builtin_amdgcn_alignbyte(w01, w00, mis0);
                if (ok1) *reinterpret_cast<uint32_t *>(dst1) = __builtin_amdgcn_alignbyte(w11, w10, mis1);
                if (n_queued + 128u > kSeamQueue) flush_seams();
            }
            flush_seams();
        }
    }
}

// ---- packing ------------------------------------------------------------------------------------------------------------
// BGZF member = 18-byte header (BSIZE), DEFLATE payload, CRC-32 and length of the uncompressed block.
// pos[j] = byte offset of member j in the packed result; file_off[c] = offset of file c's first member.
// (256 lanes: a workgroup of 1024 needs a CU with NOTHING else on it when the input decoders' wavefronts -- four per SIMD, the register file's
//  worth -- are about, and waited 0.3 ms per chunk for one in runs from BGZF inputs; four wavefronts fit wherever one decoder wavefront per SIMD
//  has ended.  The prefix sum by shuffles and one barrier per 256 blocks.)
constexpr uint32_t kPackScanLanes = 256;
__global__ __launch_bounds__(kPackScanLanes) void k_pack_scan(DevConfig C, const FileChunk *fc, const uint32_t *out_len, unsigned long long *pos,
                                                              unsigned long long *file_off, ChunkStatus *st) {
    __shared__ uint32_t sh[2][kPackScanLanes / 64];
    const uint32_t nblk = st->n_blocks;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    unsigned long long carry = 0;   // (the same in every lane)
    for (uint32_t base = 0, round = 0; base < nblk; base += kPackScanLanes, round ^= 1u) {
        const uint32_t j = base + threadIdx.x;
        const uint32_t v = j < nblk ? 26u + out_len[j < nblk ? j : 0u] : 0u;   // (a member is < 64 KiB: 256 of them fit 32 bits)
        uint32_t incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = __shfl_up(incl, d);
            if (lane >= (uint32_t)d) incl += up;
        }
        if (lane == 63u) sh[round][wave] = incl;
        __syncthreads();
        uint32_t before_waves = 0, total = 0;
#pragma unroll
        for (uint32_t w = 0; w < kPackScanLanes / 64; ++w) {
            const uint32_t tw = sh[round][w];
            before_waves += w < wave ? tw : 0u;
            total += tw;
        }
        if (j < nblk) pos[j] = carry + before_waves + incl - v;
        carry += total;
    }
    const unsigned long long total = carry;
    if (threadIdx.x == 0) { st->total_bytes = total; pos[nblk] = total; }
    __threadfence();
    __syncthreads();
    const uint32_t n_cols = (C.n_samples + 1u) * C.n_files;
    for (uint32_t c = threadIdx.x; c <= n_cols; c += kPackScanLanes) {
        const uint32_t b = fc[c].blk_base;
        file_off[c] = b < nblk ? pos[b] : total;
    }
}

__global__ __launch_bounds__(256) void k_pack_copy(const fqtk_bgzf_block *desc, const uint32_t *out_len, const uint32_t *crc,
                                                   const unsigned long long *pos, uint8_t *packed, const ChunkStatus *st) {
    for (uint32_t j = blockIdx.x; j < st->n_blocks; j += gridDim.x) {
        const uint32_t len = out_len[j], total = 26u + len, bsize1 = total - 1u;
        const uint8_t *payload = desc[j].out;
        const uint32_t c = crc[j], isz = desc[j].n_in;
        uint8_t *dst = packed + pos[j];
        auto member_byte = [&](uint32_t x) -> uint8_t {
            if (x >= 18u && x < 18u + len) return payload[x - 18u];
            switch (x) {
                case 0: return 0x1f; case 1: return 0x8b; case 2: return 8; case 3: return 4;
                case 9: return 0xff; case 10: return 6; case 12: return 'B'; case 13: return 'C'; case 14: return 2;
                case 16: return (uint8_t)(bsize1 & 0xFFu); case 17: return (uint8_t)(bsize1 >> 8);
                default: break;
            }
            if (x < 18u) return 0;
            const uint32_t y = x - 18u - len;   // trailer
            return y < 4u ? (uint8_t)(c >> (8u * y)) : (uint8_t)(isz >> (8u * (y - 4u)));
        };
        // whole aligned words of the destination, single bytes at its ragged ends
        const uint32_t mis = (uint32_t)(reinterpret_cast<uintptr_t>(dst) & 3u);
        const uint32_t head = mis ? 4u - mis : 0u;
        for (uint32_t x = threadIdx.x; x < head && x < total; x += 256u) dst[x] = member_byte(x);
        const uint32_t words = total > head ? (total - head) >> 2 : 0u;
        uint32_t *dw = reinterpret_cast<uint32_t *>(dst + head);
        for (uint32_t w = threadIdx.x; w < words; w += 256u) {
            const uint32_t x = head + 4u * w;
            uint32_t v;
            if (x >= 18u && x + 4u <= 18u + len) {
                const uint8_t *p = payload + (x - 18u);
                v = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
            } else {
                v = (uint32_t)member_byte(x) | ((uint32_t)member_byte(x + 1) << 8) | ((uint32_t)member_byte(x + 2) << 16) |
                    ((uint32_t)member_byte(x + 3) << 24);
            }
            dw[w] = v;
        }
        for (uint32_t x = head + 4u * words + threadIdx.x; x < total; x += 256u) dst[x] = member_byte(x);
    }
}

}  // namespace demux
}  // namespace fqtk
