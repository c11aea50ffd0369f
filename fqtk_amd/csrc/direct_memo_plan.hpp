// direct_memo_plan.hpp -- host-side construction of the direct-indexed form of the complete memo that
// memo_kernel<..., DIRECT> probes (barcodes of <= 10 bases; layout in memo_hash.hpp).  Plain C++17 (no
// HIP): the matcher calls it at create time, and the CPU test-suite calls it through libfqtk_host.so and
// replays the kernel's lookup.
//
// Input: the distinct Some entries of the memo that carry NO no-call -- unfolded key words (lo = bases 0-7,
// hi = bases 8-9 as memo_key_of(fold = false) builds them) and the result word.  Output: the flat result
// array indexed by memo_direct_index(), and the LDS cache of its exact-match entries.
#pragma once
#include <stdint.h>

#include <algorithm>
#include <vector>

#include "memo_hash.hpp"

namespace fqtk {

struct DirectEntry { uint32_t lo, hi, val; };

struct DirectMemoPlan {
    int entry_bytes = 0;             // 2: packed 16-bit entries (table16), 4: result words (table32)
    uint32_t ib = 0, bb = 0;         // 16-bit layout [idx : ib | best : bb | next : rest]
    std::vector<uint16_t> table16;
    std::vector<uint32_t> table32;
    std::vector<uint32_t> hot2;      // (2 << hot2_bits) slots, empty when no cache was built
    uint32_t hot2_bits = 0;
    uint64_t hot2_wanted = 0, hot2_placed = 0;
};

// The kernel's lookup of a no-call-free read, for the builder's self-check and for tests.
// *from_cache says whether the LDS cache answered.
inline uint32_t direct_memo_lookup(const DirectMemoPlan &p, uint32_t lo_unf, uint32_t c2, bool *from_cache = nullptr) {
    const uint32_t didx = memo_direct_index(lo_unf, c2);
    if (from_cache) *from_cache = false;
    if (!p.hot2.empty()) {
        const uint32_t mask = (1u << p.hot2_bits) - 1u;
        const uint32_t key[2] = {didx, memo_hot2_rot(didx)};
        for (uint32_t w = 0; w < 2; ++w)
            for (uint32_t s = 0; s < 2; ++s) {
                const uint32_t e = p.hot2[(size_t)(key[w] & mask) * 2 + s];
                if ((e >> 16) == memo_hot2_want(key[w], p.hot2_bits, w)) {
                    if (from_cache) *from_cache = true;
                    return memo_direct_unpack16(e & 0xFFFFu, p.ib, p.bb);
                }
            }
    }
    return p.entry_bytes == 2 ? memo_direct_unpack16(p.table16[didx], p.ib, p.bb) : p.table32[didx];
}

inline DirectMemoPlan plan_direct_memo(uint32_t S, uint32_t L, const std::vector<DirectEntry> &ents) {
    DirectMemoPlan plan;
    if (L == 0 || L > kDirectMaxLen) return plan;
    const uint32_t n_dir = memo_direct_entries(L);
    uint32_t max_best = 0, max_next = 0;
    for (const DirectEntry &e : ents) {
        max_best = std::max(max_best, (e.val >> 16) & 0xFFu);
        max_next = std::max(max_next, e.val >> 24);
    }
    uint32_t ib = 1, bb = 0, nb = 0;
    while ((1u << ib) - 1u < S) ++ib;             // idx < S <= 2^ib - 1: the all-ones pattern stays free for None
    while ((1u << bb) <= max_best) ++bb;
    while ((1u << nb) <= max_next) ++nb;
    const bool packed = ib + bb + nb <= 16;
    plan.ib = ib;
    plan.bb = bb;
    plan.entry_bytes = packed ? 2 : 4;
    if (packed) {
        plan.table16.assign(n_dir, 0xFFFFu);
        for (const DirectEntry &e : ents) plan.table16[memo_direct_index(e.lo, e.hi)] = (uint16_t)memo_direct_pack16(e.val, ib, bb);
    } else {
        plan.table32.assign(n_dir, kMemoEmpty);
        for (const DirectEntry &e : ents) plan.table32[memo_direct_index(e.lo, e.hi)] = e.val;
        return plan;                              // the LDS cache holds 16-bit values
    }
    // ---- LDS cache of the exact-match entries: two-choice cuckoo over two-slot buckets.  It is only a
    //      cache: a key that cannot be placed is served by the array.  Low sample index is placed first.
    struct Hot { uint32_t didx, val; };
    std::vector<Hot> hot;
    for (const DirectEntry &e : ents)
        if (((e.val >> 16) & 0xFFu) == 0) hot.push_back({memo_direct_index(e.lo, e.hi), e.val});
    std::stable_sort(hot.begin(), hot.end(), [](const Hot &a, const Hot &b) { return (a.val & 0xFFFFu) < (b.val & 0xFFFFu); });
    plan.hot2_wanted = hot.size();
    if (hot.empty()) return plan;
    uint32_t B = 6;                               // >= 6: the tag (20 - B bits) must stay below bit 14 of a slot's upper half
    while (B < kHot2MaxBucketBits && (double)(2ull << B) * 0.8 < (double)hot.size()) ++B;
    const uint32_t mask = (1u << B) - 1u;
    std::vector<int64_t> owner((size_t)2 << B, -1);
    std::vector<uint8_t> which((size_t)2 << B, 0);
    auto bucket_of = [&](size_t i, uint32_t w) { return (w ? memo_hot2_rot(hot[i].didx) : hot[i].didx) & mask; };
    uint64_t rng = 0x9E3779B97F4A7C15ull;
    for (size_t i = 0; i < hot.size(); ++i) {
        int64_t cur = (int64_t)i;
        for (int kick = 0; kick < 500 && cur >= 0; ++kick) {
            for (uint32_t w = 0; w < 2 && cur >= 0; ++w)
                for (uint32_t s = 0; s < 2 && cur >= 0; ++s) {
                    const size_t slot = (size_t)bucket_of((size_t)cur, w) * 2 + s;
                    if (owner[slot] < 0) { owner[slot] = cur; which[slot] = (uint8_t)w; cur = -1; }
                }
            if (cur < 0) break;
            rng = rng * 6364136223846793005ull + 1442695040888963407ull;
            const uint32_t w = (uint32_t)(rng >> 33) & 1u, s = (uint32_t)(rng >> 34) & 1u;
            const size_t slot = (size_t)bucket_of((size_t)cur, w) * 2 + s;
            std::swap(cur, owner[slot]);          // evict the occupant, re-home it
            which[slot] = (uint8_t)w;
        }
    }
    plan.hot2.assign((size_t)2 << B, 0xFFFFFFFFu);
    plan.hot2_bits = B;
    for (size_t slot = 0; slot < owner.size(); ++slot) {
        if (owner[slot] < 0) continue;
        const Hot &h = hot[(size_t)owner[slot]];
        const uint32_t key20 = which[slot] ? memo_hot2_rot(h.didx) : h.didx;
        plan.hot2[slot] = memo_direct_pack16(h.val, ib, bb) | (memo_hot2_want(key20, B, which[slot]) << 16);
        ++plan.hot2_placed;
    }
    return plan;
}

}  // namespace fqtk
