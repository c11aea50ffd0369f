/*
 * fqtk_match.h -- C ABI of the MI355X-native sample-barcode matcher (libfqtk_match.so).
 *
 * This is the drop-in boundary for the ONE hot path of `fqtk demux`: the reference's
 * `BarcodeMatcher` (reference: /root/reference/src/lib/barcode_matching.rs).  The reference has no
 * FFI of its own; each entry point below names the Rust item it replaces so a maintainer can bind it
 * from Rust (`extern "C"`), C++ or ctypes.  Plain pointers and sizes only.
 *
 * Semantics every entry point implements, bit for bit:
 *   enc(b):  'N','n','.' -> 0xF; else IUPAC mask of the upper-cased byte (A1 C2 G4 T8 U8 M3 R5 W9 S6
 *            Y10 K12 V7 H11 D13 B14 N15); any other byte -> 0            (src/lib/mod.rs:26-61)
 *   mm[s]  = #{ i < L : enc(read[i]) & ~enc(barcode_s[i]) != 0 }          (src/lib/bitenc.rs:432-459)
 *   best   = min_s mm[s]; idx = LOWEST s attaining it; next = second smallest with multiplicity
 *            (255 when there is one sample)                      (barcode_matching.rs:119-160)
 *   None   when len < L, when best > max_mismatches, or when next - best < min_mismatch_delta
 *                                                                 (barcode_matching.rs:150-186)
 *   error  when len > L and the no-call prefilter (barcode_matching.rs:170-172) does not reject the
 *            read first -- the reference panics there (barcode_matching.rs:95-107) -> FQTK_ELEN.
 *
 * There is NO CPU fallback behind this ABI: every compute entry point needs a gfx950 device and
 * returns FQTK_ENODEV / FQTK_EHIP otherwise.
 *
 * Threading: a handle is not thread-safe (the reference's matcher is `&mut self`); use one handle per
 * (device, host thread).  Batches on one handle complete in submission order.
 */
#ifndef FQTK_MATCH_H
#define FQTK_MATCH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes (the reference aborts with panic!/assert!; nothing here aborts) ------------ */
#define FQTK_OK 0
#define FQTK_EINVAL 1 /* bad argument / table precondition (barcode_matching.rs:61-65, samples.rs:101-133) */
#define FQTK_ELEN 2   /* observed barcode longer than expected: the reference's panic (barcode_matching.rs:95-107) */
#define FQTK_EHIP 3   /* a HIP runtime call failed; see fqtk_last_error() */
#define FQTK_ENOMEM 4
#define FQTK_ENODEV 5 /* no usable gfx950 device */
#define FQTK_ENCCL 6  /* RCCL could not be loaded or a collective failed (fqtk_matchers_allreduce_counts) */

#define FQTK_NO_MATCH 0xFFFFu /* fqtk_match_t.idx of a read the reference assigns None */
#define FQTK_MAX_BARCODE_LEN 128u
#define FQTK_MAX_SAMPLES 65534u

/* Replaces `BarcodeMatch { best_match: usize, best_mismatches: u8, next_best_mismatches: u8 }`
 * (barcode_matching.rs:16-25) / `Option<BarcodeMatch>`.  idx == FQTK_NO_MATCH <=> None; then best and
 * next are both 255 (the reference exposes no values for None). 4 bytes, little-endian. */
typedef struct fqtk_match_t {
    uint16_t idx;
    uint8_t best;
    uint8_t next;
} fqtk_match_t;

/* Opaque; owns the device-resident expected-barcode table, error word and staging buffers.
 * Replaces `struct BarcodeMatcher` (barcode_matching.rs:29-45). */
typedef struct fqtk_matcher fqtk_matcher;

/* Thread-local message of the last non-OK status returned on this thread. */
const char *fqtk_last_error(void);

/* ABI version of this header (bumped on any signature change). */
int fqtk_abi_version(void);

/* Number of visible HIP devices (0 without a GPU; FQTK_OK either way). */
int fqtk_device_count(int *n_devices);

/* Replaces `BarcodeMatcher::new(samples, max_mismatches, min_mismatch_delta, use_cache)`
 * (barcode_matching.rs:55-86; sole call site demux.rs:921-926).  `barcodes[s]` are NUL-terminated
 * ASCII strings of exactly `barcode_len` bytes; they are upper-cased and encoded like the reference
 * does (:71-75) and copied -- the caller keeps ownership.  `use_cache` has its own setter below
 * (default on, like demux.rs:925); the complete memo table is built here, on the device.  Errors: n_samples == 0 ("Must provide at least one
 * sample"), empty barcode ("Sample barcode cannot be empty string"), unequal lengths (the reference
 * would panic on the first assign), barcode_len > FQTK_MAX_BARCODE_LEN, n_samples >
 * FQTK_MAX_SAMPLES -> FQTK_EINVAL; no device -> FQTK_ENODEV. */
int fqtk_matcher_create(const char *const *barcodes, uint32_t n_samples, uint32_t barcode_len,
                        uint8_t max_mismatches, uint8_t min_mismatch_delta, int device,
                        fqtk_matcher **out);

void fqtk_matcher_destroy(fqtk_matcher *m);

/* Optional: the `sample_id` of each sample (Sample.sample_id, samples.rs:19), in table order; copied.
 * Only used to word FQTK_ELEN messages exactly like the reference's panic, which names the sample
 * (barcode_matching.rs:95-107).  Without it samples are called "sample_<index>", as the reference's own
 * test helper names them (barcode_matching.rs:195-211).  NULL clears. */
int fqtk_matcher_set_sample_ids(fqtk_matcher *m, const char *const *sample_ids);

/* Introspection (fields of BarcodeMatcher, barcode_matching.rs:29-45). */
uint32_t fqtk_matcher_n_samples(const fqtk_matcher *m);
uint32_t fqtk_matcher_barcode_len(const fqtk_matcher *m);
uint32_t fqtk_matcher_max_ns_in_barcodes(const fqtk_matcher *m); /* :73-74 */
int fqtk_matcher_device(const fqtk_matcher *m);

/* Replaces the `use_cache` argument/field of BarcodeMatcher (barcode_matching.rs:41-42,59; demux
 * passes true, demux.rs:925).  The reference's cache is a lazily filled AHashMap of Some results
 * keyed on the read bytes (:174-181); here it is a COMPLETE table, precomputed at create time on the
 * device, of every canonical (A/C/G/T/N) read within max_mismatches of a sample -- so a miss proves
 * None.  Default 1.  0 forces the exhaustive per-sample scan for every read.  Results are identical
 * either way (as in the reference).  The table is absent (scan only) when barcode_len > 20 or the
 * candidate set exceeds the build budget; fqtk_matcher_memo_entries() then returns 0. */
int fqtk_matcher_set_use_cache(fqtk_matcher *m, int use_cache);
uint64_t fqtk_matcher_memo_entries(const fqtk_matcher *m);
uint64_t fqtk_matcher_memo_candidates(const fqtk_matcher *m);

/* Where the memo lives (tuning / test knob, no counterpart in the reference; results never depend on it).
 *   FQTK_MEMO_NONE   no memo: every read takes the exhaustive scan
 *   FQTK_MEMO_TABLE  two-choice hash table in HBM/L2 + LDS hot subset (any sample alphabet, any max_mismatches)
 *   FQTK_MEMO_LDS    one-dword entries, whole table resident in each CU's LDS -- built when all sample
 *                    barcodes are plain A/C/G/T, max_mismatches <= 1 and the table fits 160 KiB; for barcodes
 *                    of 17-32 bases whose dwords do not fit (384 samples x 12+12), three-byte entries behind a
 *                    minimal perfect hash
 * fqtk_matcher_memo_kind() reports the form the next batch will use; fqtk_matcher_set_memo_kind(m,
 * FQTK_MEMO_TABLE) pins the HBM/L2 form, FQTK_MEMO_LDS (the default) means "best available". */
#define FQTK_MEMO_NONE 0
#define FQTK_MEMO_TABLE 1
#define FQTK_MEMO_LDS 2
int fqtk_matcher_memo_kind(const fqtk_matcher *m);
int fqtk_matcher_set_memo_kind(fqtk_matcher *m, int kind);
/* FQTK_MEMO_TABLE has a direct-indexed variant for barcodes of <= 10 bases: reads without a no-call index a
 * flat result array by their own 2-bit base codes (no hash, no probe sequence), an LDS cache holds its
 * exact-match entries, and only reads with an N go to the hash table.  Returns the bytes per array entry
 * (2 or 4) when that variant is what FQTK_MEMO_TABLE means for this matcher, else 0. */
int fqtk_matcher_memo_direct_bytes(const fqtk_matcher *m);

/* Replaces one `BarcodeMatcher::assign(&mut self, read_bases: &[u8]) -> Option<BarcodeMatch>` call
 * per template (barcode_matching.rs:165-186; sole call site demux.rs:968) with one call per batch.
 * HOST pointers; synchronous.
 *   obs      n x stride ASCII bytes, read i at obs + i*stride (the SoA form of
 *            ReadSet::sample_barcode_sequence, demux.rs:121-123)
 *   obs_len  per-read length, or NULL when every read has exactly barcode_len bases
 *            (stride must then be >= barcode_len); obs_len[i] <= stride is CHECKED here (FQTK_EINVAL).
 *            With obs_len the memo still serves every read of length barcode_len; shorter reads are
 *            None, longer ones None or FQTK_ELEN (rules above).  stride < barcode_len is legal with
 *            obs_len: every read is then shorter than a barcode and every result None.
 *   out      n results
 *   counts   NULL, or S+1 counters that are ADDED to: counts[idx] for Some, counts[S] for None --
 *            the per-sample `templates` metric of demux.rs:970-974
 * Returns FQTK_ELEN when some read is longer than barcode_len and not rejected by the no-call
 * prefilter; results for the other reads are still written.  fqtk_last_error() is then the
 * reference's panic sentence for the lowest such read (barcode_matching.rs:95-107),
 *   "Read barcode (<decoded read>) length (<n>) differs from expected barcode (<sample 0's barcode>)
 *    length (<L>) for sample <sample 0's id>"   followed by " [read index <i> of its batch]".
 * Uses two pipeline slots of its own: chunks a caller has in flight on slots 0..FQTK_MAX_SLOTS-1 are
 * neither waited on nor disturbed. */
int fqtk_matcher_assign_batch(fqtk_matcher *m, const uint8_t *obs, uint32_t stride,
                              const uint32_t *obs_len, uint64_t n, fqtk_match_t *out,
                              uint64_t *counts);

/* Same, but every pointer is a DEVICE pointer on the matcher's device and the work is enqueued on
 * `hip_stream` (a hipStream_t, NULL = the default stream) without synchronising: the zero-copy form
 * used when reads are already resident in HBM.  d_counts: NULL or S+1 uint64 accumulated with
 * atomics.  Length errors are latched in the handle; collect them with fqtk_matcher_poll_error().
 * d_obs_len[i] <= stride is the caller's contract here (device memory is not inspected on the host);
 * the kernels clamp every access to the read's row regardless.
 * Device memory the handle may add behind this call: once a read with an IUPAC / junk byte has been met, a
 * worklist of up to n bytes per stream in use (n / 4 read indices; grown on demand, freed with the handle). */
int fqtk_matcher_assign_batch_device(fqtk_matcher *m, const void *d_obs, uint32_t stride,
                                     const void *d_obs_len, uint64_t n, void *d_out, void *d_counts,
                                     void *hip_stream);

/* Synchronises `hip_stream`, then reports and clears the latched length error of earlier
 * *_device / enqueue calls: FQTK_OK, or FQTK_ELEN with the lowest offending read index of the
 * batch that raised it in *read_index (may be NULL). */
int fqtk_matcher_poll_error(fqtk_matcher *m, void *hip_stream, uint64_t *read_index);

/* Scalar convenience with the reference's exact call shape (one read in, Option out): returns
 * FQTK_OK and writes *out (idx == FQTK_NO_MATCH for None), or FQTK_ELEN. */
int fqtk_matcher_assign1(fqtk_matcher *m, const uint8_t *read_bases, uint32_t len,
                         fqtk_match_t *out);

/* ---- pinned-buffer pipeline: decompress -> match -> write overlap (north_star) --------------- */
/* Page-locked host memory for the SoA chunk buffers so H2D/D2H run as async DMA. */
int fqtk_pinned_alloc(size_t bytes, void **out);
int fqtk_pinned_free(void *p);

#define FQTK_MAX_SLOTS 8
/* Enqueue one chunk on pipeline slot `slot` (0..FQTK_MAX_SLOTS-1; each slot = its own HIP stream +
 * device staging): async H2D of obs (and obs_len if non-NULL), kernel, async D2H into `out`.
 * Returns immediately.  Buffers must stay valid (and should be pinned) until fqtk_matcher_wait().
 * Per-sample counts are accumulated on the device; read them with fqtk_matcher_counts(). */
int fqtk_matcher_enqueue(fqtk_matcher *m, int slot, const uint8_t *obs, uint32_t stride,
                         const uint32_t *obs_len, uint64_t n, fqtk_match_t *out);
/* ---- packed input: 4 bits per base over PCIe instead of 8 ------------------------------------------------------
 * A packed row holds base k of the barcode in nibble k (byte k / 2, low nibble first); codes A 0, C 1, T 2, G 3 and 7
 * for the no-calls N, n and '.' (mod.rs:85-87) -- lower case like upper case, as the reference's encode() treats it
 * (mod.rs:49-61).  Rows are fqtk_packed_stride(barcode_len) bytes apart (the nibbles rounded up to 4 bytes: 8 + 8
 * bases -> 8 bytes).  A read with any other byte (an IUPAC code, junk) cannot be carried in 4 bits: the packer lists it
 * as an EXCEPTION -- its index and its barcode_len ASCII bytes -- and the enqueue call puts those rows back in place on
 * the device, so results are those of the ASCII entry points for every input.
 *
 * fqtk_pack_barcodes: host-side packer (plain CPU code: one table look-up per base; run it in the threads that produce
 * the barcodes).  obs: n x stride ASCII rows as for fqtk_matcher_enqueue; exc_index / exc_rows: room for exc_cap
 * exceptions (exc_cap * barcode_len bytes of rows), *n_exc = how many there are; FQTK_ENOMEM if they do not fit. */
uint32_t fqtk_packed_stride(uint32_t barcode_len);
int fqtk_pack_barcodes(const uint8_t *obs, uint32_t stride, uint32_t barcode_len, uint64_t n, uint8_t *packed,
                       uint32_t packed_stride, uint32_t *exc_index, uint8_t *exc_rows, uint64_t exc_cap, uint64_t *n_exc);
/* fqtk_matcher_enqueue for packed rows (every read exactly barcode_len bases: no obs_len form): async H2D of the packed
 * rows and the exceptions, unpack to ASCII rows in HBM, the same kernels as every other entry point, async D2H into
 * `out`; fqtk_matcher_wait(slot) completes it, counts accumulate as for fqtk_matcher_enqueue.  Replaces one `assign`
 * per template (demux.rs:968) like fqtk_matcher_enqueue does, at 8 + 4 instead of 16 + 4 bytes per read over the link
 * for 8 + 8-base dual indexes. */
int fqtk_matcher_enqueue_packed(fqtk_matcher *m, int slot, const uint8_t *packed, uint32_t packed_stride, uint64_t n,
                                const uint32_t *exc_index, const uint8_t *exc_rows, uint64_t n_exc, fqtk_match_t *out);

/* Blocks until the chunk on `slot` is complete; FQTK_ELEN as for assign_batch.  Every slot latches its
 * own error: the status belongs to the chunk that was enqueued on THIS slot. */
int fqtk_matcher_wait(fqtk_matcher *m, int slot);
/* Adds the device-side accumulated counts (S+1) of all completed enqueue() chunks into `counts`
 * and resets the device accumulator.  Synchronises all slots. */
int fqtk_matcher_counts(fqtk_matcher *m, uint64_t *counts);

/* ---- several GPUs in one process ------------------------------------------------ */
/* Templates are independent, so chunks shard over devices with no data-path exchange: one matcher per
 * device (the table is replicated by fqtk_matcher_create), chunk k goes to matcher k mod n.  The ONE
 * collective of the path is the end-of-run reduction of the per-sample counts (the `templates` column of
 * demux-metrics.txt, demux.rs:970-974, 994-998): this call all-reduces the enqueue()/wait() accumulators of
 * `n` matchers living on `n` DISTINCT devices with RCCL (ncclAllReduce, ncclUint64, ncclSum over xGMI; the
 * library is loaded with dlopen on first use), ADDS the total into counts[0..S] and resets every
 * accumulator -- what fqtk_matcher_counts does for one matcher.  With n == 1 no collective is needed and
 * none is made unless `force_collective` is non-zero (a one-rank communicator; exercises the RCCL path
 * on a single-GPU box).  All matchers must have the same number of samples. */
int fqtk_matchers_allreduce_counts(fqtk_matcher *const *matchers, int n, int force_collective, uint64_t *counts);

#ifdef __cplusplus
}
#endif
#endif /* FQTK_MATCH_H */
