/*
 * fqtk_bgzf.h -- C ABI of the MI355X BGZF block compressor (part of libfqtk_match.so).
 *
 * The reference writes every output FASTQ through pooled-writer's BgzfCompressor
 * (/root/reference/src/bin/commands/demux.rs:755-798): independent gzip members of <= 64 KiB carrying the 'BC'
 * extra field.  End to end that compression is what bounds `fqtk demux` on the host while the GPU that does the
 * matching idles; this entry point moves the DEFLATE step of each block onto the device.  One 1024-lane workgroup
 * per block produces ONE dynamic-Huffman DEFLATE block (or a stored block when the data does not compress);
 * the caller wraps it into a BGZF member (18-byte header, payload, CRC32 and ISIZE of the INPUT) and keeps
 * computing the CRC on the host.  Compressed bytes are unpinned by the reference's tests (decompressed content
 * is what they compare, demux.rs:1069-1076): any inflate returns the input bytes.
 *
 * Buffers: `in`, `out` and `out_len` must be device-accessible: page-locked host memory from
 * fqtk_pinned_alloc() (read / written by the kernel directly over PCIe -- no staging copies) or device memory.
 * There is no CPU fallback: FQTK_ENODEV without a GPU.
 */
#ifndef FQTK_BGZF_H
#define FQTK_BGZF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FQTK_BGZF_MAX_IN 65280u      /* uncompressed bytes per block (the bgzf crate's block payload) */
#define FQTK_BGZF_OUT_STRIDE 65536u  /* bytes the caller reserves per block for the DEFLATE payload */
#define FQTK_BGZF_SLOTS 4

typedef struct fqtk_bgzf fqtk_bgzf;

/* One block: `n_in` (1..FQTK_BGZF_MAX_IN) bytes at `in`; the payload is written to `out` (4-byte aligned,
 * FQTK_BGZF_OUT_STRIDE bytes available).  Padded to 24 bytes. */
typedef struct fqtk_bgzf_block {
    const uint8_t *in;
    uint8_t *out;
    uint32_t n_in;
    uint32_t reserved;
} fqtk_bgzf_block;

/* Thread-local message of the last non-OK status of an fqtk_bgzf_* call on this thread (status codes: fqtk_match.h). */
const char *fqtk_bgzf_last_error(void);

int fqtk_bgzf_create(int device, fqtk_bgzf **out);
void fqtk_bgzf_destroy(fqtk_bgzf *z);

/* Enqueues the compression of `n` blocks on pipeline slot `slot` (0..FQTK_BGZF_SLOTS-1, each its own HIP
 * stream) and returns at once.  `blocks` and `out_len` (n entries: payload bytes per block) must stay valid
 * and device-accessible until fqtk_bgzf_wait(). */
int fqtk_bgzf_deflate_enqueue(fqtk_bgzf *z, int slot, const fqtk_bgzf_block *blocks, uint32_t n, uint32_t *out_len);
int fqtk_bgzf_wait(fqtk_bgzf *z, int slot);

#ifdef __cplusplus
}
#endif
#endif /* FQTK_BGZF_H */
