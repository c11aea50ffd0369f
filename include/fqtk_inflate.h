/*
 * fqtk_inflate.h -- C ABI of the MI355X BGZF member decoder (part of libfqtk_match.so).
 *
 * The reference opens every input FASTQ through a gz-aware reader and inflates it on the host
 * (/root/reference/src/bin/commands/demux.rs:844-849: fgoxide `Io` -> flate2).  A BGZF file (bgzip, htslib, fqtk's own
 * output: demux.rs:755-798) is a sequence of independent gzip members of <= 64 KiB; this entry point inflates a batch of
 * them on the device, one wavefront per member (csrc/bgzf_inflate.hpp), checks every member's CRC-32 and ISIZE against
 * its trailer and counts its newlines -- so that the compressed bytes are what crosses PCIe and the host never touches
 * the text.  The caller walks the member headers (18 bytes: BSIZE) and trailers (CRC-32, ISIZE) and says where every
 * member's text goes.
 *
 * Parity = zlib returns the same bytes for the same members (tests/test_bgzf_inflate.py).
 * Buffers must be device-accessible: device memory, or page-locked host memory from fqtk_pinned_alloc().
 * There is no CPU fallback: FQTK_ENODEV without a GPU.
 */
#ifndef FQTK_INFLATE_H
#define FQTK_INFLATE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FQTK_INFLATE_SLOTS 4
#define FQTK_INFLATE_MAX_ISIZE 65536u /* a BGZF member's text */

/* Per-member status (0 = the member's text is in place and its CRC-32 and length are the trailer's). */
#define FQTK_INFLATE_OK 0
#define FQTK_INFLATE_ERR_BLOCK_TYPE 1   /* reserved block type */
#define FQTK_INFLATE_ERR_STORED_LEN 2   /* stored block: LEN != ~NLEN */
#define FQTK_INFLATE_ERR_CODE_LENGTHS 3 /* bad code-length code, bad repeat, no end-of-block code */
#define FQTK_INFLATE_ERR_HUFFMAN 4      /* over-subscribed or incomplete Huffman code */
#define FQTK_INFLATE_ERR_BAD_CODE 5     /* bits that are no code of the block */
#define FQTK_INFLATE_ERR_DISTANCE 6     /* a match reaches before the start of the member */
#define FQTK_INFLATE_ERR_OUTPUT 7       /* more text than ISIZE */
#define FQTK_INFLATE_ERR_TRUNCATED 8    /* the stream runs past the member's payload */
#define FQTK_INFLATE_ERR_LENGTH 9       /* less text than ISIZE */
#define FQTK_INFLATE_ERR_CRC 10         /* CRC-32 of the text differs from the trailer's */

/* One member: its DEFLATE payload is in[payload_off .. payload_off + payload_len) (of a BGZF member: bytes 18 .. BSIZE - 7
 * of it), `crc` and `isize` are its trailer's; the text is written to out[out_off .. out_off + isize).  24 bytes. */
typedef struct fqtk_inflate_member {
    uint64_t payload_off;
    uint64_t out_off;
    uint32_t payload_len;
    uint32_t isize; /* <= FQTK_INFLATE_MAX_ISIZE */
    uint32_t crc;
    uint32_t reserved;
} fqtk_inflate_member;

typedef struct fqtk_inflate fqtk_inflate;

/* Thread-local message of the last non-OK return of an fqtk_inflate_* call on this thread (status codes: fqtk_match.h). */
const char *fqtk_inflate_last_error(void);

int fqtk_inflate_create(int device, fqtk_inflate **out);
void fqtk_inflate_destroy(fqtk_inflate *z);

/* Enqueues `n` members on pipeline slot `slot` (0..FQTK_INFLATE_SLOTS-1, each its own HIP stream) and returns at once.
 * `in` (4-byte aligned; in_len bytes readable) holds the payloads, `out` receives the text (4 bytes of slack behind the last member's: the check reads whole dwords); status[j] and lines[j]
 * (number of '\n' in member j's text) are written for every member.  All of it must stay valid and device-accessible
 * until fqtk_inflate_wait(). */
int fqtk_inflate_enqueue(fqtk_inflate *z, int slot, const uint8_t *in, uint64_t in_len, const fqtk_inflate_member *members,
                         uint32_t n, uint8_t *out, uint32_t *status, uint32_t *lines);
int fqtk_inflate_wait(fqtk_inflate *z, int slot);

#ifdef __cplusplus
}
#endif
#endif /* FQTK_INFLATE_H */
