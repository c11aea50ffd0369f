// wave_emu.hpp -- a 64-lane wavefront on the CPU, for the test-suite only.
//
// Wave-level device code (csrc/bgzf_inflate.hpp) is written per lane, with cross-lane operations in the middle of its
// control flow (ballot, readlane, prefix sum, barrier): it cannot be run "lane after lane" the way the phase functions of the
// compressor are.  Here the 64 lanes are 64 fibers (ucontext) on one thread; a cross-lane operation stores the lane's
// contribution and yields to the next lane; when the round comes back every lane has contributed and each reads the
// result.  Lanes run from one cross-lane operation to the next one at a time, so memory shared between lanes must be
// ordered by barrier() exactly where the device needs __syncthreads() / a fence -- an omission shows up here as a wrong result,
// not as a race.  All lanes must reach the same sequence of cross-lane operations (the device code's wave-uniform control flow).
#pragma once
#include <ucontext.h>

#include <cstdint>
#include <cstdlib>
#include <functional>
#include <vector>

namespace fqtk_host {

class WaveEmu {
  public:
    static constexpr int kLanesPerWave = 64;
    // runs body(lane) for the 64 lanes
    void run(const std::function<void(WaveEmu &)> &body) {
        body_ = &body;
        if (stacks_.empty()) stacks_.resize((size_t)kLanesPerWave * kStack);
        done_ = 0;
        for (int l = 0; l < kLanesPerWave; ++l) {
            getcontext(&ctx_[l]);
            ctx_[l].uc_stack.ss_sp = stacks_.data() + (size_t)l * kStack;
            ctx_[l].uc_stack.ss_size = kStack;
            ctx_[l].uc_link = &main_;
            finished_[l] = false;
            makecontext(&ctx_[l], (void (*)())trampoline, 2, (unsigned)((uintptr_t)this >> 32), (unsigned)((uintptr_t)this & 0xFFFFFFFFu));
        }
        while (done_ < kLanesPerWave)
            for (int l = 0; l < kLanesPerWave; ++l) {
                if (finished_[l]) continue;
                cur_ = l;
                swapcontext(&main_, &ctx_[l]);
                if (!yielded_) { finished_[l] = true; ++done_; }
                yielded_ = false;
            }
    }

    // ---- what device code sees (the W of bgzf_inflate.hpp)
    uint32_t lane() const { return (uint32_t)cur_; }
    uint64_t ballot(bool p) {
        const uint64_t *all = exchange(p ? 1u : 0u);
        uint64_t m = 0;
        for (int l = 0; l < kLanesPerWave; ++l) m |= (uint64_t)(all[l] & 1u) << l;
        return m;
    }
    uint32_t readlane(uint32_t v, uint32_t l) { return (uint32_t)exchange(v)[l & 63u]; }
    uint32_t uniform(uint32_t v) { return v; }
    void set_bit64(uint64_t &m, uint32_t bit) { m |= 1ull << (bit & 63u); }   // (the device: one s_bitset1_b64 on a scalar register pair)
    uint32_t scan_incl(uint32_t v) {
        const uint64_t *all = exchange(v);
        uint32_t s = 0;
        for (int l = 0; l <= cur_; ++l) s += (uint32_t)all[l];
        return s;
    }
    uint32_t scan_max_incl(uint32_t v) {
        const uint64_t *all = exchange(v);
        uint32_t s = 0;
        for (int l = 0; l <= cur_; ++l) s = (uint32_t)all[l] > s ? (uint32_t)all[l] : s;
        return s;
    }
    uint32_t shuffle(uint32_t v, uint32_t l) { return (uint32_t)exchange(v)[l & 63u]; }
    float rcp(float x) const { return 1.0f / x; }
    void barrier() { exchange(0); }
    void fence_global() { exchange(0); }

  private:
    static constexpr size_t kStack = 128 * 1024;
    static void trampoline(unsigned hi, unsigned lo) {
        WaveEmu *self = reinterpret_cast<WaveEmu *>(((uintptr_t)hi << 32) | (uintptr_t)lo);
        (*self->body_)(*self);
        // returning ends the fiber: uc_link resumes run()
    }
    // every lane's value of this round (valid until the lane's next cross-lane operation)
    const uint64_t *exchange(uint64_t v) {
        const int lane_now = cur_;
        const unsigned round = op_count_[lane_now]++;
        slots_[round & 1u][lane_now] = v;
        yielded_ = true;
        swapcontext(&ctx_[lane_now], &main_);
        cur_ = lane_now;
        return slots_[round & 1u];
    }
    const std::function<void(WaveEmu &)> *body_ = nullptr;
    ucontext_t main_, ctx_[kLanesPerWave];
    std::vector<char> stacks_;
    bool finished_[kLanesPerWave];
    unsigned op_count_[kLanesPerWave] = {0};
    uint64_t slots_[2][kLanesPerWave];
    int cur_ = 0, done_ = 0;
    bool yielded_ = false;
};

}  // namespace fqtk_host
