// chunk_dispatch.hpp -- the threads between the readers and the devices of `fqtk demux --devices a,b,..`.
//
// chunk k goes to device k mod G (chunk_schedule.hpp).  One thread (the caller of push) cuts the
// stream of chunks in order; EVERY DEVICE HAS ITS OWN SUBMIT THREAD -- a submit is page-table work, a dozen kernel launches
// and the enqueue of the chunk's host-to-device copies, and with one thread for all devices the second device's copy
// could not start before the first one's submit had returned -- and one collector thread takes the
// chunks back IN ORDER of k (every output file is a stream in input order: demux.rs:945-977), whatever order the
// devices' threads got their submits out in.  A (device, slot) pair is handed its next chunk only after the previous one
// has been collected.  The device operations are the caller's callables, so the CPU test-suite drives the same threads
// with a fake device (host_capi.cpp: fqtk_host_chunk_dispatch_check, also under TSan).
// Round 6: an optional THIRD stage, `retire` (its own thread, chunks in order of k): what is done with a collected chunk that need
// not hold up the collection of the next one -- the record pipeline's collector waited for the device, brought the chunk's members
// home AND waited for the writer threads to append them, one after the other: 2.6 + 0.7 + 1.7 ms a chunk, the whole run's clock.
// With the appends on the retire thread the next chunk's wait and copy run beside them.  A slot is free once its chunk has RETIRED.
#pragma once
#include <condition_variable>
#include <cstdint>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "chunk_schedule.hpp"

namespace fqtk_host {

// Job: what a chunk is made of (moved to the device's thread).  Meta: what the collector needs to know of a submitted chunk.
template <class Job, class Meta>
class ChunkDispatcher {
  public:
    using SubmitFn = std::function<Meta(int dev, int slot, uint64_t k, Job &job)>;     // on device dev's thread
    using CollectFn = std::function<void(int dev, int slot, uint64_t k, Meta &meta)>;  // on the collector thread, k ascending
    using RetireFn = std::function<void(int dev, int slot, uint64_t k, Meta &meta)>;   // on the retire thread, k ascending, after collect(k)
    ChunkDispatcher(size_t devices, size_t slots, SubmitFn submit, CollectFn collect, RetireFn retire = nullptr)
        : submit_(std::move(submit)), collect_(std::move(collect)), retire_(std::move(retire)), ring_(devices * slots + 1), queues_(devices) {
        sched_.devices = devices;
        sched_.slots = slots;
        for (size_t g = 0; g < devices; ++g) submitters_.emplace_back([this, g] { run_device(g); });
        collector_ = std::thread([this] { run_collector(); });
        if (retire_) retirer_ = std::thread([this] { run_retirer(); });
    }
    ~ChunkDispatcher() { finish(); }
    // The next chunk (k = 0, 1, 2, .. in call order).  Blocks while devices * slots chunks are outstanding.
    // prepare: called on the pushing thread, in chunk order, once the chunk's (device, slot) is free and before the device's thread gets the
    // chunk -- what must happen in order but should not be held across the wait for a slot (fed text: the cut of the chunk's windows, which
    // pins the inputs' text in place; cut before the wait, the pins of chunk k + 1 overlapped those of chunk k and an input's feeder never
    // found a moment to change arena: 65-75 ms per change, round 6).
    void push(Job job, const std::function<void(Job &)> &prepare = nullptr) {
        const uint64_t k = next_++;
        {
            std::unique_lock<std::mutex> lk(mu_);
            cv_done_.wait(lk, [&] { return sched_.may_submit(k, done_); });
        }
        if (prepare) prepare(job);
        {
            std::lock_guard<std::mutex> lk(mu_);
            queues_[(size_t)sched_.device_of(k)].push_back(Item{k, std::move(job)});
        }
        cv_jobs_.notify_all();
    }
    // No more chunks: waits until every chunk has been collected.
    void finish() {
        if (finished_) return;
        finished_ = true;
        {
            std::lock_guard<std::mutex> lk(mu_);
            total_ = next_;
        }
        cv_jobs_.notify_all();
        cv_flights_.notify_all();
        cv_retire_.notify_all();
        for (auto &t : submitters_) t.join();
        collector_.join();
        if (retirer_.joinable()) retirer_.join();
    }
    uint64_t pushed() const { return next_; }

  private:
    struct Item { uint64_t k; Job job; };
    struct Flight { uint64_t k = 0; int dev = 0, slot = 0; bool submitted = false; Meta meta{}; };
    void run_device(size_t g) {
        for (;;) {
            Item it;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_jobs_.wait(lk, [&] { return !queues_[g].empty() || total_ != kOpen; });
                if (queues_[g].empty()) return;
                it = std::move(queues_[g].front());
                queues_[g].pop_front();
            }
            const int slot = sched_.slot_of(it.k);
            Meta m = submit_((int)g, slot, it.k, it.job);
            {
                std::lock_guard<std::mutex> lk(mu_);
                Flight &f = ring_[it.k % ring_.size()];
                f.k = it.k; f.dev = (int)g; f.slot = slot; f.meta = std::move(m); f.submitted = true;
            }
            cv_flights_.notify_all();
        }
    }
    void run_collector() {
        for (uint64_t want = 0;; ++want) {
            Flight f;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_flights_.wait(lk, [&] { return want == total_ || (ring_[want % ring_.size()].submitted && ring_[want % ring_.size()].k == want); });
                if (want == total_) return;
                f = std::move(ring_[want % ring_.size()]);
                ring_[want % ring_.size()].submitted = false;
            }
            collect_(f.dev, f.slot, f.k, f.meta);
            if (retire_) {   // the chunk's slot stays taken until the retire thread is through with it
                {
                    std::lock_guard<std::mutex> lk(mu_);
                    to_retire_.push_back(std::move(f));
                }
                cv_retire_.notify_all();
                continue;
            }
            {
                std::lock_guard<std::mutex> lk(mu_);
                ++done_;
            }
            cv_done_.notify_all();
        }
    }
    void run_retirer() {
        for (uint64_t want = 0;; ++want) {
            Flight f;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_retire_.wait(lk, [&] { return want == total_ || !to_retire_.empty(); });
                if (to_retire_.empty()) return;   // (want == total_: every chunk has been through)
                f = std::move(to_retire_.front());
                to_retire_.pop_front();
            }
            retire_(f.dev, f.slot, f.k, f.meta);
            {
                std::lock_guard<std::mutex> lk(mu_);
                ++done_;
            }
            cv_done_.notify_all();
        }
    }
    static constexpr uint64_t kOpen = ~0ull;
    SubmitFn submit_;
    CollectFn collect_;
    RetireFn retire_;
    ChunkSchedule sched_;
    std::mutex mu_;
    std::condition_variable cv_jobs_, cv_flights_, cv_done_, cv_retire_;
    std::deque<Flight> to_retire_;
    std::vector<Flight> ring_;
    std::vector<std::deque<Item>> queues_;
    uint64_t next_ = 0, done_ = 0, total_ = kOpen;
    bool finished_ = false;
    std::vector<std::thread> submitters_;
    std::thread collector_, retirer_;
};

}  // namespace fqtk_host
