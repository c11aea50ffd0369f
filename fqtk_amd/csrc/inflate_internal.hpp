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
// n_blocks whole blocks were decoded; n_sym / end_bit / final_block describe the last block boundary reached (also of a chunk that failed behind it)
struct StreamChunkEnd { uint32_t status, final_block, n_sym, n_blocks; uint64_t end_bit; uint32_t flags /* kStreamHighLiterals */, pad; };
struct StreamPlan { uint32_t n_chunks, n_found; uint64_t sym_wanted; };   // chunks cut by stream_scan_launch; symbols they wished for in all
constexpr uint32_t kMaxStreamSlots = 4096;
// One wavefront per chunk: symbols (a byte, or 256 + j = byte j of the 32 KiB in front of the chunk) and where the chunk ended.
hipError_t stream_decode_launch(hipStream_t stream, const uint8_t *in, uint64_t in_len, const StreamChunk *chunks, uint32_t n, uint16_t *sym, StreamChunkEnd *ends);
// The same with the chunks cut on the device: slot k >= 1 of n_slots is the first place in bytes [k, k + 1) * chunk_bytes of `in` where a block can start
// (a lane per bit position: bgzf_inflate.hpp find_block_start), slot 0 is first_bit (a block boundary the caller knows); chunk j runs from the j-th start
// found to the (j + 1)-th -- the last one found only ends the chunk before it unless to_end -- with room for sym_per_byte symbols per compressed byte + slack.
// low_literals_only: headers that give a code to a literal >= 128 are not taken for starts.  starts: n_slots; chunks / ends: n_slots; plan: how many chunks there are.
hipError_t stream_scan_launch(hipStream_t stream, const uint8_t *in, uint64_t in_len, uint64_t first_bit, uint32_t chunk_bytes, uint32_t n_slots, bool to_end,
                              bool low_literals_only, uint32_t sym_per_byte, uint32_t slack, uint64_t sym_cap, uint64_t *starts, StreamChunk *chunks, StreamPlan *plan, uint16_t *sym,
                              StreamChunkEnd *ends);
// windows: (n + 1) x 32 KiB, the first filled by the caller (the text in front of chunk 0), the last = the text behind chunk n - 1;
// maps: scratch of 2 x n x 32 Ki symbols (the chunks' window maps, composed in log2(n) rounds); then text[out_off[k] ..) = chunk k's bytes.
hipError_t stream_resolve_launch(hipStream_t stream, const StreamChunk *chunks, const StreamChunkEnd *ends, uint32_t n, const uint64_t *out_off, const uint16_t *sym,
                                 uint8_t *windows, uint16_t *maps, uint8_t *text);
// CRC-32 and newline count of pieces of text in place (pieces[j].out_off, .isize <= 65 536); status_zero: n zeros.
hipError_t pieces_check_launch(hipStream_t stream, const fqtk_inflate_member *pieces, uint32_t n, const uint8_t *text, uint32_t *status_zero, uint32_t *lines, uint32_t *crc,
                               const uint32_t *crc_pow_dev);
}  // namespace inflate
}  // namespace fqtk
