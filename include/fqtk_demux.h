/*
 * fqtk_demux.h -- C ABI of the MI355X record pipeline of `fqtk demux` (part of libfqtk_match.so).
 *
 * The rows either side of the barcode matcher, moved onto the device so that the text of a
 * template crosses PCIe once in and its compressed records once out:
 *
 *   reference (host, one template at a time)                           here (device, one chunk of templates at a time)
 *   ------------------------------------------------------------------ -------------------------------------------------
 *   seq_io fastq::Reader + ReadSetIterator::next   demux.rs:285-343    line index + record validation + segment extraction
 *   ReadSet::sample_barcode_sequence               demux.rs:121-123    barcode rows packed for the matcher (fqtk_match.h)
 *   BarcodeMatcher::assign                         demux.rs:968        fqtk_matcher_assign_batch_device
 *   DemuxMetric counting                           demux.rs:970-974    per-sample counts (a column of the placement sums)
 *   ReadSet::write_header_internal                 demux.rs:171-267    header plan + record pieces (csrc/record_format.hpp)
 *   SampleWriters::write                           demux.rs:396-415    records placed by stable per-file prefix sums,
 *                                                                      copied into 65 280-byte blocks in HBM, input order kept
 *   pooled-writer BgzfCompressor (libdeflater)     demux.rs:755-798    fqtk::bgzf::deflate_kernel + CRC-32 on those blocks,
 *                                                                      whole BGZF members packed per output file
 *
 * The caller (csrc/host/demux.cpp) keeps what is I/O: it reads / gunzips each input, cuts it after the same number of
 * records in every input (a record is four lines: counting '\n' is all the host does with the text), hands the raw text
 * over, and appends the returned bytes to the output files.  File f of sample s (s == n_samples: unmatched) is column
 * s * files_per_sample + f; the files of a sample are ordered as demux.rs:674-688 orders them: by output type in the
 * order T, B, M, C (the requested ones), and within a type by position of the segment in the read structures.
 *
 * Plain pointers and sizes; status codes of fqtk_match.h; no CPU fallback (FQTK_ENODEV without a GPU).
 * A handle is not thread-safe: one thread submits, the same or another single thread collects, in slot order.
 */
#ifndef FQTK_DEMUX_H
#define FQTK_DEMUX_H

#include <stddef.h>
#include <stdint.h>

#include "fqtk_inflate.h"
#include "fqtk_match.h"

#ifdef __cplusplus
extern "C" {
#endif

#define FQTK_DEMUX_SLOTS 3        /* chunks in flight */
#define FQTK_DEMUX_MAX_INPUTS 16
#define FQTK_DEMUX_MAX_FILES 32   /* output files per sample */

/* One segment of a read structure (`read-structure` crate: <length|+><T|B|M|C|S>); offset = sum of the lengths before it. */
typedef struct fqtk_demux_segment {
    uint32_t offset;
    int32_t length; /* -1: '+', the rest of the read (last segment only) */
    char kind;      /* 'T' 'B' 'M' 'C' 'S' */
} fqtk_demux_segment;

typedef struct fqtk_demux_config {
    uint32_t n_inputs;
    const uint32_t *n_segments;         /* per input */
    const fqtk_demux_segment *segments; /* all inputs' segments, input after input */
    uint8_t want[4];                    /* which segment types get output files: T, B, M, C (demux.rs:608-612) */
    int skip_too_few_bases;             /* -S too-few-bases: such templates are dropped and counted (demux.rs:300-306),
                                           otherwise the first one is an error (:307-313) */
    uint32_t max_chunk_templates;       /* largest n_templates of a submit (buffers are sized by it) */
    int carry_blocks;                   /* 1: a file's partly filled block waits for the next chunk (one device);
                                           0: every chunk ends all its blocks (chunks of one run spread over devices) */
    int compression_level;              /* --compression-level (demux.rs:641-643): parse effort of the DEFLATE kernel */
} fqtk_demux_config;

typedef struct fqtk_demuxer fqtk_demuxer;

/* Why a chunk failed (status FQTK_EINVAL from collect unless said otherwise); `template_index` is the lowest
 * offending template of the chunk, `input` the input file it was found in. */
#define FQTK_DEMUX_OK 0
#define FQTK_DEMUX_ERR_NO_AT 1        /* record does not start with '@' */
#define FQTK_DEMUX_ERR_NO_PLUS 2      /* third line does not start with '+' */
#define FQTK_DEMUX_ERR_QUAL_LEN 3     /* sequence and quality lengths differ */
#define FQTK_DEMUX_ERR_LINES 4        /* the text does not hold 4 * n_templates lines */
#define FQTK_DEMUX_ERR_TOO_SHORT 5    /* read shorter than its read structure needs (no -S too-few-bases) */
#define FQTK_DEMUX_ERR_BARCODE_LEN 6  /* FQTK_ELEN of the matcher: fqtk_last_error() is the reference's panic text */
#define FQTK_DEMUX_ERR_HEADER 7       /* write_header_internal's errors; `detail` = fqtk::fmt::HeaderError */

typedef struct fqtk_demux_result {
    const uint8_t *bytes;      /* whole BGZF members, the files' runs one after another (valid until the slot is reused) */
    const uint64_t *file_off;  /* n_files + 1 offsets into bytes: file c's members are bytes[file_off[c] .. file_off[c+1]) */
    uint64_t n_files;
    uint64_t n_blocks;         /* BGZF members in this result */
    uint32_t n_templates;      /* templates of the chunk (0 for a flush) */
    uint32_t n_skipped;        /* of them dropped for too few bases */
    int error;                 /* FQTK_DEMUX_ERR_* */
    uint32_t error_input, error_template, error_detail;
    const uint64_t *text_end;  /* chunks of fed text (fqtk_demuxer_submit_fed): per input, the position in the input's whole text
                                  of the byte behind the chunk's last record; NULL otherwise */
} fqtk_demux_result;

/* The matcher decides device and sample count; it must outlive the demuxer and is used by it (do not enqueue on it
 * from elsewhere meanwhile).  FQTK_EINVAL: no sample-barcode... (any configuration the reference accepts is accepted;
 * more than FQTK_DEMUX_MAX_INPUTS inputs, more than FQTK_DEMUX_MAX_FILES files per sample or more than 24 barcode
 * segments are not). */
int fqtk_demuxer_create(fqtk_matcher *m, const fqtk_demux_config *cfg, fqtk_demuxer **out);
void fqtk_demuxer_destroy(fqtk_demuxer *d);

uint32_t fqtk_demuxer_files_per_sample(const fqtk_demuxer *d);

/* Enqueues one chunk on `slot` (0..FQTK_DEMUX_SLOTS-1, used round robin) and returns at once: text[i] is input i's
 * FASTQ text of exactly n_templates records (text_len[i] bytes ending with '\n'; page-locked memory from
 * fqtk_pinned_alloc for an asynchronous copy) and must stay valid until the slot's collect(). */
int fqtk_demuxer_submit(fqtk_demuxer *d, int slot, const uint8_t *const *text, const uint64_t *text_len,
                        uint32_t n_templates);

/* Blocks until the text of the chunk on `slot` has been copied to the device: the caller's buffers may be reused
 * (the chunk itself is still in flight). */
int fqtk_demuxer_text_done(fqtk_demuxer *d, int slot);

/* ---- BGZF inputs inflated on the device (the input side of demux.rs:844-849) --------------------------------------------
 * Instead of text the host may hand over the BGZF members of every input as they lie in the file (fqtk_inflate.h says
 * what a member is): their text is inflated on the device, checked against the members' CRC-32 / ISIZE, and stays there;
 * a chunk is cut out of it by line counts, so the host never sees the text.  A run either feeds or submits text.  (Several
 * devices: every input is fed to ONE demuxer, see fqtk_demuxer_fed_cut below.)
 *
 * fqtk_demuxer_feed: `bytes` (len of them, page-locked for a fast copy) hold n_members members of input `input` in file
 * order (members[j].payload_off is relative to bytes; out_off is ignored).  Blocks until they are inflated -- other
 * inputs' feeds and the chunks in flight go on meanwhile (one thread per input may feed) -- and returns the number of lines
 * (newlines) fed for this input so far.  `last` != 0 with the input's last members (n_members may be 0): a newline is
 * added behind the text, so that a final line without one counts (the blank line this makes otherwise is the caller's to
 * ignore).  FQTK_EINVAL with "corrupt BGZF block k: ..." when a member does not inflate or its CRC-32 / length is wrong. */
int fqtk_demuxer_feed(fqtk_demuxer *d, uint32_t input, const uint8_t *bytes, uint64_t len, const fqtk_inflate_member *members,
                      uint32_t n_members, int last, uint64_t *lines_fed);

/* The next n_templates records (4 n_templates lines) of every input's fed text as one chunk on `slot`: as
 * fqtk_demuxer_submit, but nothing is copied.  Every input must have been fed that many lines beyond what earlier chunks
 * took.  The result's text_end says where each input's text was left. */
int fqtk_demuxer_submit_fed(fqtk_demuxer *d, int slot, uint32_t n_templates);

/* ---- fed text and several devices (`fqtk demux --devices a,b,..`, SURVEY.md 8e: chunk k runs on device k mod G) ------------
 * Every input has ONE home demuxer: the one its compressed bytes are fed to (fqtk_demuxer_feed, fqtk_demuxer_stream_*), whose
 * device inflates them and keeps the text.  Chunks of templates are cut out of that text IN ORDER -- one thread, input by
 * input -- with fqtk_demuxer_fed_cut, and a cut chunk may then be run by ANY demuxer made with the same configuration
 * (fqtk_demuxer_submit_windows, on that demuxer's own submit thread): a window whose home is the running demuxer is used
 * where it lies, a window of another demuxer is copied device to device (xGMI between two devices) into the slot's own text
 * buffer first -- the call returns when those copies are done.  A window stays valid, and holds its home's text in place,
 * from the cut until the submit that takes it has returned; every window cut must be submitted exactly once.
 * fqtk_demuxer_submit_fed(d, slot, n) is the cut of every input of d followed by the submit on d itself.
 * (The reference's readers are per input and feed one loop, demux.rs:928-934,844-849: nothing ties an input's decoding to
 *  the place its templates are matched and written from.) */
typedef struct fqtk_fed_window {
    fqtk_demuxer *home;      /* whose fed text this is */
    uint32_t input;          /* which input's */
    uint32_t lead;           /* bytes in front of the first member's text at base (base is 16-byte aligned) */
    uint32_t first_line;     /* the chunk's first line is this line of the window */
    uint32_t n_templates;    /* the cut's */
    const uint8_t *base;     /* device address on home's device */
    uint64_t len;            /* bytes of the window from base */
    uint64_t pos;            /* position of base in the input's whole text */
    uint32_t arena, reserved; /* (the home's: which of its buffers the window lies in) */
} fqtk_fed_window;
/* The next n_templates records (4 n_templates lines) of `input`'s text fed to `home`.  FQTK_EINVAL when fewer lines have been fed. */
int fqtk_demuxer_fed_cut(fqtk_demuxer *home, uint32_t input, uint32_t n_templates, fqtk_fed_window *out);
/* One chunk on `slot` of d out of windows[0 .. n_inputs) (window i of input i, all cut with the same n_templates), as
 * fqtk_demuxer_submit_fed.  The result's text_end says where each input's text was left. */
int fqtk_demuxer_submit_windows(fqtk_demuxer *d, int slot, const fqtk_fed_window *windows, uint32_t n_templates);

/* The fed text of `input` from position `pos` to its end (*n_bytes; the first min(*n_bytes, cap) of them in buf): what
 * lies behind the last record (end-of-file checks).  pos == ~0: the text's last min(cap, its length) bytes.  Only text no chunk has
 * consumed in full is still there. */
int fqtk_demuxer_fed_tail(fqtk_demuxer *d, uint32_t input, uint64_t pos, uint8_t *buf, size_t cap, uint64_t *n_bytes);

/* ---- serial gzip inputs (`gzip`, bcl2fastq: one member per file) decoded on the device in chunks ---------------------------
 * A gzip member is ONE bit stream; what makes it parallel is host/parallel_gunzip.hpp's plan, moved onto the device: the host
 * finds places where a DEFLATE block starts (a bit offset whose dynamic-Huffman header parses into two complete codes), every
 * chunk between two such places is decoded by a wavefront of its own WITHOUT the 32 KiB of text before it (16-bit symbols:
 * a byte, or a reference into that unknown window), the caller accepts a chunk only if the chunk before it ended on exactly
 * the bit it started at, and the accepted chunks' windows are then handed down the chain and everything is resolved to
 * bytes in the input's fed text (where fqtk_demuxer_submit_fed cuts chunks of templates out of it).
 *
 * fqtk_demuxer_stream_decode: `bytes` (len < 512 MiB, page-locked for a fast copy) are a stretch of the file; chunk k
 * starts at bit chunks[k].start_bit of them -- chunk 0 at a bit the caller KNOWS to be a block boundary (the member's first
 * block, or where the last accepted chunk ended), the others at places it found -- and stops at the first block boundary
 * at or behind chunks[k].stop_bit (~0: at the member's final block).  Blocks until every chunk is decoded and says how
 * each ended (status: FQTK_INFLATE_ERR_*, FQTK_INFLATE_ERR_OUTPUT when the chunk's room for symbols -- eight times its
 * compressed size -- was too small).
 * fqtk_demuxer_stream_commit: the first n_accept chunks of that decode become text (member_start != 0: chunk 0 began a gzip
 * member, nothing lies in front of it; last: as in fqtk_demuxer_feed).  Returns the lines fed so far, and CRC-32 and length
 * of the committed text for the caller to fold into the member's (zlib's crc32_combine) and compare with the trailer. */
typedef struct fqtk_stream_chunk { uint64_t start_bit, stop_bit; } fqtk_stream_chunk;
/* How a chunk ended.  n_blocks whole DEFLATE blocks were decoded; n_bytes / end_bit / final_block describe the LAST block boundary it
 * reached -- also when status != 0: a chunk that ran out of room for symbols (FQTK_INFLATE_ERR_OUTPUT) or of bytes
 * (FQTK_INFLATE_ERR_TRUNCATED) behind a boundary decoded everything up to it, and a caller may accept it as ending there. */
typedef struct fqtk_stream_end { uint32_t status, final_block; uint64_t n_bytes, end_bit, start_bit; uint32_t n_blocks, flags; } fqtk_stream_end;
#define FQTK_STREAM_END_HIGH_LITERALS 1u /* fqtk_stream_end.flags: a block of the chunk gave a code to a literal >= 128 (7-bit text never does) */
#define FQTK_STREAM_SCAN_TEXT 1u         /* fqtk_demuxer_stream_scan flags: headers that give a code to a literal >= 128 are not taken for block
                                            starts -- for a stream whose chunks so far never set FQTK_STREAM_END_HIGH_LITERALS; an accidental
                                            header (about one per 100 MB of FASTQ.gz otherwise, and it costs the rest of its stretch) nearly always does */
int fqtk_demuxer_stream_decode(fqtk_demuxer *d, uint32_t input, const uint8_t *bytes, uint64_t len, const fqtk_stream_chunk *chunks, uint32_t n,
                               fqtk_stream_end *ends);
/* The same with the chunks CUT ON THE DEVICE: slot k >= 1 of n_slots (<= 4096) is the first place in bytes [k, k + 1) x chunk_bytes of the
 * stretch where a block can start (one lane per bit position, csrc/bgzf_inflate.hpp find_block_start), slot 0 is first_bit -- a block
 * boundary the caller knows; chunk j runs from the j-th start found to the (j + 1)-th (the last one found only ends the chunk before it,
 * unless to_end != 0: the stretch reaches the end of the member, and the last chunk runs to its final block).  Room for symbols:
 * sym_per_byte per compressed byte of a chunk + 64 Ki.  ends (n_slots of them) receives *n_chunks entries, start_bit included. */
int fqtk_demuxer_stream_scan(fqtk_demuxer *d, uint32_t input, const uint8_t *bytes, uint64_t len, uint64_t first_bit, uint32_t chunk_bytes, uint32_t n_slots,
                             int to_end, uint32_t sym_per_byte, uint32_t flags, fqtk_stream_end *ends, uint32_t *n_chunks);
int fqtk_demuxer_stream_commit(fqtk_demuxer *d, uint32_t input, uint32_t n_accept, int member_start, int last, uint64_t *lines_fed, uint32_t *crc32,
                               uint64_t *n_text);
/* Optional, before the first stretch of an input: device buffers for stretches of up to max_len bytes in max_slots chunks at
 * sym_per_byte, and two arenas of arena_bytes for the input's text (0: as needed).  Buffers that grow in the middle of a run are freed
 * and allocated again, and a free waits for the whole device (0.05-0.15 s each time, measured). */
int fqtk_demuxer_stream_reserve(fqtk_demuxer *d, uint32_t input, uint64_t max_len, uint32_t max_slots, uint32_t sym_per_byte, uint64_t arena_bytes);
/* A stretch the device could not take (no block start to cut at, a block larger than any room, a decoder in doubt) is decoded by the
 * caller's own sequential decoder: fqtk_demuxer_stream_window hands out the 32 KiB of text in front of the next chunk (zeros where the
 * member has less), fqtk_demuxer_stream_commit_text takes the text back (window_after: the 32 KiB in front of what follows; NULL when a
 * member ends) -- CRC-32 and lines as fqtk_demuxer_stream_commit. */
int fqtk_demuxer_stream_window(fqtk_demuxer *d, uint32_t input, uint8_t *window /* 32768 */);
int fqtk_demuxer_stream_commit_text(fqtk_demuxer *d, uint32_t input, const uint8_t *text, uint64_t n, const uint8_t *window_after, int last,
                                    uint64_t *lines_fed, uint32_t *crc32);

/* Device seconds spent inflating (all inputs). */
int fqtk_demuxer_inflate_seconds(fqtk_demuxer *d, double *seconds);

/* Waits for the chunk on `slot`.  FQTK_OK with res->error == 0: res holds the members to append to the files.
 * res->error != 0: nothing of the chunk was written into the result; the run cannot continue. */
int fqtk_demuxer_collect(fqtk_demuxer *d, int slot, fqtk_demux_result *res);
/* Optional first half of the collect of `slot`: waits for the chunk's kernels and enqueues the copy of its members to the host without
 * waiting for it; fqtk_demuxer_collect then only waits for that copy.  One thread may begin chunk k + 1 while another finishes chunk k
 * (csrc/host/demux.cpp: collector and retire thread), so that a chunk's copy home runs beside the appends of the chunk before. */
int fqtk_demuxer_collect_begin(fqtk_demuxer *d, int slot);

/* For wording an error: header (without '@', at most cap - 1 bytes, NUL-terminated) and number of bases of template
 * `template_index` in input `input` of the chunk last collected on `slot` (its text is still in device memory until
 * the slot's next submit). */
int fqtk_demuxer_record_text(fqtk_demuxer *d, int slot, uint32_t input, uint32_t template_index, char *header, size_t cap,
                             uint32_t *n_bases);

/* After the last chunk has been collected: ends every partly filled block (carry_blocks) and returns those members
 * the same way (the 28-byte BGZF EOF marker is the caller's to write). */
int fqtk_demuxer_flush(fqtk_demuxer *d, fqtk_demux_result *res);

/* Adds the per-sample template counts (n_samples + 1, unmatched last) of all collected chunks into counts. */
int fqtk_demuxer_counts(fqtk_demuxer *d, uint64_t *counts);

/* Seconds of device time per stage, summed over the chunks collected so far (names: fqtk_demuxer_stage_name). */
#define FQTK_DEMUX_STAGES 8
int fqtk_demuxer_stage_seconds(fqtk_demuxer *d, double *seconds /* FQTK_DEMUX_STAGES */);
const char *fqtk_demuxer_stage_name(int stage);

#ifdef __cplusplus
}
#endif
#endif /* FQTK_DEMUX_H */
