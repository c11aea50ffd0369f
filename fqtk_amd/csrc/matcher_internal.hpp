// matcher_internal.hpp -- what fqtk_demux.hip (the record pipeline) needs from fqtk_match.hip beyond the C ABI.
// Internal to libfqtk_match.so: C++ linkage, not part of include/.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

struct fqtk_matcher;

namespace fqtk {
namespace internal {
// Sets this thread's fqtk_last_error() text; returns `code`.
int set_error(int code, const std::string &msg);
// Enqueues on `stream`: the latch word of the matcher's *_device entry point -> *d_dst (device memory), latch re-armed.
// Stream-ordered, so the word belongs to the batches enqueued on `stream` since the last call: no synchronisation.
int take_device_error(fqtk_matcher *m, hipStream_t stream, unsigned long long *d_dst);
// fqtk_last_error() := the reference's panic sentence (barcode_matching.rs:95-107) for read `index` of a device batch.
// Returns FQTK_ELEN.
int word_device_length_error(fqtk_matcher *m, const uint8_t *d_obs, const uint32_t *d_lens, uint32_t stride, uint64_t n, uint64_t index);
}  // namespace internal
}  // namespace fqtk
