// chunk_schedule.hpp -- which device and pipeline slot a chunk of templates goes to, and when it may go.
//
// templates are independent, so the chunks of one run shard over the devices with no data-path
// exchange.  Chunk k goes to device k mod G, on that device's slot (k / G) mod S; chunks are COLLECTED in order of k
// (every output file is a stream in input order, as the reference's sequential loop writes it, demux.rs:945-977) and a
// slot is free again once its chunk has been collected and written -- so chunk k may be submitted once fewer than
// G * S chunks are outstanding.  Plain arithmetic, kept apart so that the CPU test-suite can drive it without a GPU.
#pragma once
#include <cstddef>
#include <cstdint>

namespace fqtk_host {

struct ChunkSchedule {
    size_t devices = 1, slots = 1;
    int device_of(uint64_t k) const { return (int)(k % devices); }
    int slot_of(uint64_t k) const { return (int)((k / devices) % slots); }
    // `done` = chunks collected and written so far (they complete in order)
    bool may_submit(uint64_t k, uint64_t done) const { return k - done < devices * slots; }
};

}  // namespace fqtk_host
