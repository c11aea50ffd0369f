// inflate_internal.hpp -- the BGZF member decoder of fqtk_inflate.hip as the record pipeline (fqtk_demux.hip) launches it.
// Internal to libfqtk_match.so: C++ linkage, not part of include/.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/fqtk_inflate.h"

namespace fqtk {
namespace inflate {
constexpr uint32_t kCrcPowWords = 256 + 265;   // x^(8 * 264 * j), j < 256, then x^(8 * r), r <= 264 (mod the CRC polynomial)
// Fills a host array of kCrcPowWords (copied to the device once per handle).
void crc_pow_table(uint32_t *pow);
// One wavefront per member: text to out + members[j].out_off, status[j]; then CRC-32 / newline count per member
// (status[j] becomes FQTK_INFLATE_ERR_CRC where the text's CRC is not the trailer's), lines[j].
hipError_t inflate_launch(hipStream_t stream, const uint8_t *in, uint64_t in_len, const fqtk_inflate_member *members, uint32_t n,
                          uint8_t *out, uint32_t *status, uint32_t *lines, const uint32_t *crc_pow_dev);
// ---- a serial gzip stream in chunks ----------------------------------------------------------------------------------------
constexpr uint32_t kStreamWindow = 32768;   // DEFLATE's reach
struct StreamChunk { uint64_t start_bit, stop_bit /* ~0: to the final block */, sym_off; uint32_t cap, pad; };   // bits from in[0]; symbols at sym + sym_off
struct StreamChunkEnd { uint32_t status, final_block, n_sym, pad; uint64_t end_bit; };
// One wavefront per chunk: symbols (a byte, or 256 + j = byte j of the 32 KiB in front of the chunk) and where the chunk ended.
hipError_t stream_decode_launch(hipStream_t stream, const uint8_t *in, uint64_t in_len, const StreamChunk *chunks, uint32_t n, uint16_t *sym, StreamChunkEnd *ends);
// windows: (n + 1) x 32 KiB, the first filled by the caller (the text in front of chunk 0), the last = the text behind chunk n - 1;
// then text[out_off[k] ..) = chunk k's bytes.
hipError_t stream_resolve_launch(hipStream_t stream, const StreamChunk *chunks, const StreamChunkEnd *ends, uint32_t n, const uint64_t *out_off, const uint16_t *sym,
                                 uint8_t *windows, uint8_t *text);
// CRC-32 and newline count of pieces of text in place (pieces[j].out_off, .isize <= 65 536); status_zero: n zeros.
hipError_t pieces_check_launch(hipStream_t stream, const fqtk_inflate_member *pieces, uint32_t n, const uint8_t *text, uint32_t *status_zero, uint32_t *lines, uint32_t *crc,
                               const uint32_t *crc_pow_dev);
}  // namespace inflate
}  // namespace fqtk
