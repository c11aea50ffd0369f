// gpu_bgzf_stage.hpp -- host plumbing of `fqtk demux --gpu-bgzf`: BGZF blocks are DEFLATE-compressed on the GPU
// (include/fqtk_bgzf.h) instead of by the libdeflate thread pool.
//
// The reference compresses on the host (pooled-writer -> bgzf -> libdeflater, demux.rs:755-798), and so does this
// binary by default; measured end to end, that compression takes two thirds of all CPU time while the GPU that
// matched the barcodes idles.  With --gpu-bgzf:
//   routers   cut a full block as before, but copy it into a 64 KiB slab of ONE page-locked arena, take its CRC32
//             (the BGZF trailer carries the CRC of the uncompressed bytes) and queue {file, sequence number, slab};
//   GPU stage (one thread) gathers whatever is queued -- up to 256 blocks -- into a descriptor array, launches the
//             kernel, which reads the slabs and writes the payload slabs straight over PCIe (no staging copies),
//             and keeps up to three launches in flight;
//   writers   wrap each payload into a BGZF member (header, payload, CRC32, ISIZE) and write the members of a file
//             in sequence order.
// `--compression-level` does not apply to this path (one strategy: greedy LZ77 + dynamic Huffman per block); on
// FASTQ text the files come out between zlib's level 1 and level 5 (tools/bgzf_ratio.py), on the scope-E files the
// size of libdeflate's level 5.
#pragma once
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../../include/fqtk_bgzf.h"
#include "../../../include/fqtk_match.h"

namespace fqtk_host {

// Blocking stack of free slab indices.
class SlabPool {
  public:
    void fill(uint32_t n) { for (uint32_t i = n; i-- > 0;) free_.push_back(i); }
    uint32_t get() {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return !free_.empty(); });
        const uint32_t s = free_.back();
        free_.pop_back();
        return s;
    }
    void put(uint32_t s) {
        { std::lock_guard<std::mutex> lk(mu_); free_.push_back(s); }
        cv_.notify_one();
    }
  private:
    std::mutex mu_;
    std::condition_variable cv_;
    std::vector<uint32_t> free_;
};

struct GpuBlock {         // one BGZF block on its way through the GPU stage
    void *file = nullptr; // OutFile*; nullptr = end of input
    uint64_t seq = 0;
    uint32_t in_slab = 0, out_slab = 0, n = 0, crc = 0, out_len = 0;
};

template <typename T>
class BlockingQueue {
  public:
    explicit BlockingQueue(size_t cap) : cap_(cap) {}
    void push(const T &v) {
        std::unique_lock<std::mutex> lk(mu_);
        not_full_.wait(lk, [&] { return q_.size() < cap_; });
        q_.push_back(v);
        not_empty_.notify_one();
    }
    T pop() {
        std::unique_lock<std::mutex> lk(mu_);
        not_empty_.wait(lk, [&] { return !q_.empty(); });
        T v = q_.front();
        q_.pop_front();
        not_full_.notify_one();
        return v;
    }
    bool try_pop(T *v) {
        std::unique_lock<std::mutex> lk(mu_);
        if (q_.empty()) return false;
        *v = q_.front();
        q_.pop_front();
        not_full_.notify_one();
        return true;
    }
  private:
    size_t cap_;
    std::mutex mu_;
    std::condition_variable not_full_, not_empty_;
    std::deque<T> q_;
};

class GpuBgzfStage {
  public:
    static constexpr uint32_t kSlab = 65536, kBatch = 256, kInFlight = 3;
    BlockingQueue<GpuBlock> to_gpu{8192}, to_writers{8192};

    // Allocates the arenas and the device-side compressor.  false + *err on failure.
    bool init(int device, uint32_t n_slabs, std::string *err) {
        n_slabs_ = n_slabs;
        void *p = nullptr;
        if (fqtk_pinned_alloc((size_t)n_slabs * kSlab, &p) != FQTK_OK) { *err = fqtk_last_error(); return false; }
        in_ = static_cast<uint8_t *>(p);
        if (fqtk_pinned_alloc((size_t)n_slabs * kSlab, &p) != FQTK_OK) { *err = fqtk_last_error(); return false; }
        out_ = static_cast<uint8_t *>(p);
        if (fqtk_pinned_alloc((size_t)FQTK_BGZF_SLOTS * kBatch * sizeof(fqtk_bgzf_block), &p) != FQTK_OK) { *err = fqtk_last_error(); return false; }
        desc_ = static_cast<fqtk_bgzf_block *>(p);
        if (fqtk_pinned_alloc((size_t)FQTK_BGZF_SLOTS * kBatch * sizeof(uint32_t), &p) != FQTK_OK) { *err = fqtk_last_error(); return false; }
        lens_ = static_cast<uint32_t *>(p);
        if (fqtk_bgzf_create(device, &z_) != FQTK_OK) { *err = fqtk_bgzf_last_error(); return false; }
        in_pool.fill(n_slabs);
        out_pool.fill(n_slabs);
        return true;
    }
    uint8_t *in_slab(uint32_t s) const { return in_ + (size_t)s * kSlab; }
    const uint8_t *out_slab(uint32_t s) const { return out_ + (size_t)s * kSlab; }
    SlabPool in_pool, out_pool;

    // The GPU stage thread.  Ends after `n_producers` end markers; then sends `n_writers` end markers downstream.
    // Returns false + *err on a device error.
    bool run(size_t n_producers, size_t n_writers, std::string *err) {
        struct Flight { int slot; std::vector<GpuBlock> blocks; };
        std::deque<Flight> flying;
        size_t ended = 0;
        int next_slot = 0;
        auto land = [&]() -> bool {   // oldest launch done: its blocks go to the writers
            Flight f = std::move(flying.front());
            flying.pop_front();
            if (fqtk_bgzf_wait(z_, f.slot) != FQTK_OK) { *err = fqtk_bgzf_last_error(); return false; }
            const uint32_t *lens = lens_ + (size_t)f.slot * kBatch;
            for (size_t i = 0; i < f.blocks.size(); ++i) {
                f.blocks[i].out_len = lens[i];
                to_writers.push(f.blocks[i]);
            }
            return true;
        };
        while (ended < n_producers) {
            GpuBlock b;
            if (!flying.empty()) {   // something is in flight: do not sleep on the queue while it could land
                if (!to_gpu.try_pop(&b)) { if (!land()) return false; continue; }
            } else {
                b = to_gpu.pop();
            }
            if (!b.file) { ++ended; continue; }
            std::vector<GpuBlock> batch{b};
            while (batch.size() < kBatch && to_gpu.try_pop(&b)) {
                if (!b.file) { ++ended; continue; }
                batch.push_back(b);
            }
            if (flying.size() == kInFlight && !land()) return false;
            const int slot = next_slot;
            next_slot = (next_slot + 1) % FQTK_BGZF_SLOTS;
            fqtk_bgzf_block *d = desc_ + (size_t)slot * kBatch;
            for (size_t i = 0; i < batch.size(); ++i) {
                batch[i].out_slab = out_pool.get();
                d[i].in = in_slab(batch[i].in_slab);
                d[i].out = out_ + (size_t)batch[i].out_slab * kSlab;
                d[i].n_in = batch[i].n;
                d[i].reserved = 0;
            }
            if (fqtk_bgzf_deflate_enqueue(z_, slot, d, (uint32_t)batch.size(), lens_ + (size_t)slot * kBatch) != FQTK_OK) {
                *err = fqtk_bgzf_last_error();
                return false;
            }
            blocks_ += batch.size();
            ++launches_;
            flying.push_back(Flight{slot, std::move(batch)});
        }
        while (!flying.empty())
            if (!land()) return false;
        for (size_t w = 0; w < n_writers; ++w) to_writers.push(GpuBlock{});
        return true;
    }
    uint64_t blocks() const { return blocks_; }
    uint64_t launches() const { return launches_; }

  private:
    uint32_t n_slabs_ = 0;
    uint8_t *in_ = nullptr, *out_ = nullptr;
    fqtk_bgzf_block *desc_ = nullptr;
    uint32_t *lens_ = nullptr;
    fqtk_bgzf *z_ = nullptr;
    uint64_t blocks_ = 0, launches_ = 0;
};

}  // namespace fqtk_host
