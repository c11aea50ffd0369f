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
}  // namespace inflate
}  // namespace fqtk
