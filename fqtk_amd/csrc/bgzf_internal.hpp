// bgzf_internal.hpp -- the DEFLATE kernel of fqtk_bgzf.hip as the record pipeline (fqtk_demux.hip) launches it.
// Internal to libfqtk_match.so: C++ linkage, not part of include/.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/fqtk_bgzf.h"

namespace fqtk {
namespace bgzf {
// Bytes of token scratch one resident workgroup needs; the kernel's LDS.
size_t deflate_token_bytes_per_group();
// Allows the kernel its LDS on the current device (once per device).
hipError_t deflate_prepare();
// One workgroup per CU over blocks[0 .. *n_blocks_dev): DEFLATE payload to blocks[j].out, its size to out_len[j],
// the CRC-32 of the block's bytes to crc[j].  `level`: --compression-level (parse effort).  next_block: a device word the
// workgroups share the blocks out with (zeroed here), or NULL for round robin.
hipError_t deflate_launch(hipStream_t stream, uint32_t groups, const fqtk_bgzf_block *blocks, const uint32_t *n_blocks_dev,
                          uint32_t *out_len, uint32_t *crc, uint32_t *tok, int level, uint32_t *next_block);
}  // namespace bgzf
}  // namespace fqtk
