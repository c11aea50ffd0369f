// direct_memo_plan.hpp -- host-side construction of the direct-indexed form of the complete memo that
// memo_kernel<..., DIRECT> probes (barcodes of <= 10 bases; layout in memo_hash.hpp).  Plain C++17 (no
// HIP): the matcher calls it at create time, and the CPU test-suite calls it through libfqtk_host.so and
// replays the kernel's lookup.
//
// Input: the distinct Some entries of the memo that carry NO no-call -- unfolded key words (lo = bases 0-7,
// hi = bases 8-9 as memo_key_of(fold = false) builds them) and the result word -- plus the strings that ARE a
// spelling of some sample barcode but resolve to None (two samples admit them: min_mismatch_delta), which the
// array already answers (absent = None) and the LDS cache should too: they are as popular as any exact match.
// Output: the flat result array indexed by memo_direct_index(), the LDS cache of the exact-match strings, and
// (plan_nbuckets) the bucketed cuckoo table of the entries WITH a no-call.
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
    uint32_t hot2_bits = 0, nbits = 0;   // log2(buckets); bits of the index (memo_direct_index_bits)
    uint64_t hot2_wanted = 0, hot2_placed = 0;
};

// The kernel's lookup of a no-call-free read, for the builder's self-check and for tests.
// *from_cache says whether the LDS cache answered.
inline uint32_t direct_memo_lookup(const DirectMemoPlan &p, uint32_t lo_unf, uint32_t c2, bool *from_cache = nullptr) {
    const uint32_t didx = memo_direct_index(lo_unf, c2);
    if (from_cache) *from_cache = false;
    if (!p.hot2.empty()) {
        for (uint32_t w = 0; w < 2; ++w)
            for (uint32_t s = 0; s < 2; ++s) {
                const uint32_t e = p.hot2[(size_t)memo_hot2_bucket(didx, p.hot2_bits, p.nbits, w) * 2 + s];
                if ((e >> 16) == memo_hot2_want(didx, p.hot2_bits, p.nbits, w)) {
                    if (from_cache) *from_cache = true;
                    return memo_direct_unpack16(e & 0xFFFFu, p.ib, p.bb);
                }
            }
    }
    return p.entry_bytes == 2 ? memo_direct_unpack16(p.table16[didx], p.ib, p.bb) : p.table32[didx];
}

inline DirectMemoPlan plan_direct_memo(uint32_t S, uint32_t L, const std::vector<DirectEntry> &ents,
                                       const std::vector<DirectEntry> &exact_none = {}) {
    DirectMemoPlan plan;
    if (L == 0 || L > kDirectMaxLen) return plan;
    const uint32_t n_dir = memo_direct_entries(L);
    plan.nbits = memo_direct_index_bits(L);
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
    // ... then the exact spellings that are None (val = kMemoEmpty; packed: 0xFFFF): placed after every Some entry
    for (const DirectEntry &e : exact_none) hot.push_back({memo_direct_index(e.lo, e.hi), kMemoEmpty});
    plan.hot2_wanted = hot.size();
    if (hot.empty()) return plan;
    uint32_t B = 6;                               // >= 6: the tag (nbits - B bits) must stay below bit 14 of a slot's upper half
    while (B < kHot2MaxBucketBits && B < plan.nbits && (double)(2ull << B) * 0.8 < (double)hot.size()) ++B;
    std::vector<int64_t> owner((size_t)2 << B, -1);
    std::vector<uint8_t> which((size_t)2 << B, 0);
    auto bucket_of = [&](size_t i, uint32_t w) { return memo_hot2_bucket(hot[i].didx, B, plan.nbits, w); };
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
        plan.hot2[slot] = (h.val == kMemoEmpty ? 0xFFFFu : memo_direct_pack16(h.val, ib, bb)) | (memo_hot2_want(h.didx, B, plan.nbits, which[slot]) << 16);
        ++plan.hot2_placed;
    }
    return plan;
}

// ---- the entries WITH a no-call: two-choice cuckoo over BUCKETS of two slots ------------------------------------
// A bucket is one aligned 16-byte line {key0 | spill << 31, val0, key1, val1} (one-word folded keys, memo_key_of):
// the kernel fetches a read's FIRST bucket (memo_hash2's s1) with its other gathers, finds the key in either slot,
// and only where the bucket's SPILL bit says that one of its would-be owners lives in its second bucket (s2) does a
// wave go round once more.  Empty slots: key 0x7FFFFFFF (no folded key has bit 23 set), val None.
struct NBucketPlan {
    std::vector<uint32_t> words;   // 4 per bucket
    uint32_t mask = 0;             // buckets - 1
    uint64_t second = 0;           // keys living in their second bucket
    bool ok = false;
};
struct NKey { uint32_t lo, val; };   // folded 4-bit key, result

inline uint32_t nbucket_lookup(const NBucketPlan &p, uint32_t lo) {
    const uint32_t sh = memo_nbucket_shift(p.mask), s1 = memo_nbucket1(lo, sh), s2 = memo_nbucket2(lo, sh);
    const uint32_t *b = &p.words[(size_t)s1 * 4];
    if ((b[0] & 0x7FFFFFFFu) == lo) return b[1];
    if ((b[2] & 0x7FFFFFFFu) == lo) return b[3];
    if (!(b[0] >> 31)) return kMemoEmpty;
    b = &p.words[(size_t)s2 * 4];
    if ((b[0] & 0x7FFFFFFFu) == lo) return b[1];
    if ((b[2] & 0x7FFFFFFFu) == lo) return b[3];
    return kMemoEmpty;
}

inline NBucketPlan plan_nbuckets(const std::vector<NKey> &keys, uint32_t min_buckets = 64) {
    NBucketPlan plan;
    uint64_t nb = min_buckets;
    while (nb * 2 * 0.6 < (double)keys.size()) nb <<= 1;   // load <= 0.6 of the slots
    for (int attempt = 0; attempt < 6 && nb <= (1u << 24); ++attempt, nb <<= 1) {
        const uint32_t mask = (uint32_t)(nb - 1);
        std::vector<int64_t> owner(nb * 2, -1);
        auto buckets_of = [&](int64_t k, uint32_t &a1, uint32_t &a2) { a1 = memo_nbucket1(keys[(size_t)k].lo, memo_nbucket_shift(mask)); a2 = memo_nbucket2(keys[(size_t)k].lo, memo_nbucket_shift(mask)); };
        auto free_slot = [&](uint32_t b) -> int64_t { return owner[(size_t)b * 2] < 0 ? (int64_t)b * 2 : (owner[(size_t)b * 2 + 1] < 0 ? (int64_t)b * 2 + 1 : -1); };
        bool ok = true;
        uint64_t rng = 0x9E3779B97F4A7C15ull;
        for (size_t i = 0; i < keys.size() && ok; ++i) {
            int64_t cur = (int64_t)i;
            for (int kick = 0;; ++kick) {
                uint32_t a1, a2;
                buckets_of(cur, a1, a2);
                int64_t at = free_slot(a1);
                if (at < 0) at = free_slot(a2);
                if (at >= 0) { owner[(size_t)at] = cur; break; }
                if (kick == 1000) { ok = false; break; }
                rng = rng * 6364136223846793005ull + 1442695040888963407ull;
                const size_t victim = (size_t)(((rng >> 33) & 1 ? a1 : a2)) * 2 + ((rng >> 34) & 1);
                std::swap(cur, owner[victim]);
            }
        }
        if (!ok) continue;
        for (bool moved = true; moved;) {   // back into the first bucket wherever it has room by now
            moved = false;
            for (size_t p = 0; p < owner.size(); ++p) {
                if (owner[p] < 0) continue;
                uint32_t a1, a2;
                buckets_of(owner[p], a1, a2);
                if (a1 == (uint32_t)(p / 2)) continue;
                const int64_t at = free_slot(a1);
                if (at >= 0) { owner[(size_t)at] = owner[p]; owner[p] = -1; moved = true; }
            }
        }
        plan.words.assign(nb * 4, 0u);
        for (size_t b = 0; b < nb; ++b) { plan.words[b * 4] = plan.words[b * 4 + 2] = 0x7FFFFFFFu; plan.words[b * 4 + 1] = plan.words[b * 4 + 3] = kMemoEmpty; }
        plan.second = 0;
        for (size_t p = 0; p < owner.size(); ++p) {
            if (owner[p] < 0) continue;
            const NKey &k = keys[(size_t)owner[p]];
            uint32_t a1, a2;
            buckets_of(owner[p], a1, a2);
            uint32_t *w = &plan.words[(p / 2) * 4 + (p & 1) * 2];
            w[0] = (w[0] & 0x80000000u) | k.lo;
            w[1] = k.val;
            if (a1 != (uint32_t)(p / 2)) { plan.words[(size_t)a1 * 4] |= 0x80000000u; ++plan.second; }
        }
        plan.mask = mask;
        plan.ok = true;
        return plan;
    }
    return plan;
}

}  // namespace fqtk
