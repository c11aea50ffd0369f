// lds_memo_plan.hpp -- host-side construction of the LDS-resident compact memo that
// lds_memo_kernels.hip.h probes.  Plain C++17 (no HIP): the matcher calls it at create time, and the
// CPU test-suite calls it through libfqtk_host.so and replays the kernel's lookup in numpy.
//
// Input: the distinct Some entries of the complete memo (unfolded 4-bit keys + result word) and the
// encoded sample barcodes.  Output: the LDS image [entry table | sample keys], or ok = false when the
// memo is not of the required shape (a sample with an IUPAC code / N, an entry more than one base
// away from its sample, S = 1, too many samples for the index field) or does not fit one CU's LDS.
#pragma once
#include <stdint.h>

#include <algorithm>
#include <vector>

#include "memo_hash.hpp"

namespace fqtk {

struct LdsEntry { uint32_t k[4]; uint32_t val; };

struct LdsMemoPlan {
    bool ok = false;
    std::vector<uint32_t> image;   // n_slots entry dwords, then (S + 1) sample keys (key_stride dwords each)
    uint32_t n_slots = 0;
    bool pow2 = true;              // slot = hash & mask; false: slot = hash16 * n_slots >> 16 (lds_slot_any)
    uint32_t slot_mask_b = 0;      // (n_slots - 1) << 2
    uint32_t idx_bits = 0;
    uint32_t skey_off_b = 0;
    uint32_t salt = 0;
    int kw = 0;                    // key words: 1 (L <= 8), 2 (L <= 16), 3 (L <= 24), 4 (L <= 32) -- never folded
    int key_stride = 0;            // dwords per sample key (4 for kw = 3 and 4)
    uint64_t multi_score = 0;      // (exact-match keys with > 1 fingerprint match) << 32 | all such keys
    // the minimal-perfect-hash form (memo_hash.hpp, plan_lds_memo_mph below): image = [n_slots x u16 | n_slots x u8 | buckets x u16 | keys]
    bool mph = false;
    uint32_t t8_off_b = 0;         // byte offset of the entries' third bytes
    uint32_t aux_off_b = 0;        // byte offset of the buckets' displacements
    uint32_t bucket_mask = 0;      // buckets - 1 (a power of two)
    uint32_t max_displacement = 0; // the largest displacement any bucket needed (reported; tests)
};

// The kernel's lookup, for the builder's self-check and for tests: returns the result word or kMemoEmpty.
// A candidate whose fingerprint agrees is VERIFIED against its sample's key; candidates are tried in
// probe order until one verifies (the kernel does the same, the later rounds behind wave-uniform branches).
inline uint32_t lds_memo_lookup(const LdsMemoPlan &p, const uint32_t key[4]) {
    if (p.mph) {   // one candidate: the slot the perfect hash sends the key to
        const uint8_t *bytes = reinterpret_cast<const uint8_t *>(p.image.data());
        auto half = [&](uint32_t off_b) { return (uint32_t)bytes[off_b] | ((uint32_t)bytes[off_b + 1] << 8); };
        uint32_t a, b;
        mph_hashes(key[0], key[1], key[2], key[3], p.salt, a, b);
        const uint32_t d = half(p.aux_off_b + ((a & p.bucket_mask) << 1));
        const uint32_t slot = mph_slot(a, b, d, p.n_slots);
        const uint32_t e = half(slot << 1) | ((uint32_t)bytes[p.t8_off_b + slot] << 16);
        const uint32_t idx = e & ((1u << kMphIdxBits) - 1u), xnib = (e >> 15) & 7u, pos = (e >> 18) & 31u;
        const uint32_t *sk = &p.image[(p.skey_off_b >> 2) + (size_t)idx * p.key_stride];
        uint32_t diff = 0;
        for (int w = 0; w < p.kw; ++w)
            diff |= key[w] ^ sk[w] ^ ((pos >> 3) == (uint32_t)w ? xnib << ((pos & 7u) * 4u) : 0u);
        return diff == 0 ? mph_entry_result(e) : kMemoEmpty;
    }
    uint32_t h[3], fps;
    memo_hash3(key[0], key[1], key[2], key[3], p.salt, h[0], h[1], h[2], fps);
    const uint32_t fp_mask = lds_fp_mask(p.idx_bits, p.kw);
    uint32_t a[3];
    lds_slots(p.pow2, h[0], h[1], p.slot_mask_b, p.n_slots, a[0], a[1], a[2]);
    for (int c = 0; c < 3; ++c) {
        const uint32_t e = p.image[a[c] >> 2];
        if ((e ^ fps) & fp_mask) continue;
        const uint32_t idx = e & ((1u << p.idx_bits) - 1u), pos = lds_entry_pos(e, p.kw), xnib = (e >> 17) & 7u;
        const uint32_t *sk = &p.image[(p.skey_off_b >> 2) + (size_t)idx * p.key_stride];
        uint32_t diff = 0;
        for (int w = 0; w < p.kw; ++w)
            diff |= key[w] ^ sk[w] ^ ((pos >> 3) == (uint32_t)w ? xnib << ((pos & 7u) * 4u) : 0u);
        if (diff == 0) return e & lds_res_mask(p.idx_bits);
    }
    return kMemoEmpty;
}

// What both planners start from: the samples' keys (plain A/C/G/T samples only) and every entry described relative to its sample
// (idx, best <= 1, next, the differing nibble and its position).  false: the memo is not of that shape.
struct LdsRelEntry { uint32_t idx, best, next, xnib, pos; };
inline bool lds_relative_entries(uint32_t S, uint32_t L, const std::vector<LdsEntry> &ents, const std::vector<std::vector<uint8_t>> &enc,
                                 int kw, int ks, std::vector<uint32_t> &skeys, std::vector<LdsRelEntry> &rel) {
    skeys.assign((size_t)(S + 1) * ks, 0xFFFFFFFFu);   // row S = "no sample": equals no key
    for (uint32_t s = 0; s < S; ++s) {
        uint32_t k[4] = {0, 0, 0, 0};
        for (uint32_t i = 0; i < L; ++i) {
            const uint8_t nib = enc[s][i];
            if (nib != 1 && nib != 2 && nib != 4 && nib != 8) return false;   // an IUPAC / N sample matches several strings
            const uint32_t code = nib == 1 ? 0u : nib == 2 ? 1u : nib == 8 ? 2u : 3u;   // A C T G, as memo_code_of
            k[i >> 3] |= code << memo_nibble_shift(i);
        }
        for (int w = 0; w < ks; ++w) skeys[(size_t)s * ks + w] = w < kw ? k[w] : 0u;
    }
    // every entry must be "its sample's barcode with `best` (<= 1) bases replaced"
    rel.resize(ents.size());
    for (size_t i = 0; i < ents.size(); ++i) {
        const uint32_t idx = ents[i].val & 0xFFFFu, best = (ents[i].val >> 16) & 0xFFu, next = ents[i].val >> 24;
        if (idx >= S || next > 31 || best > 1) return false;
        uint32_t pos = 0, xnib = 0, ndiff = 0;
        for (uint32_t b = 0; b < 32; ++b) {
            const uint32_t w = b >> 3;
            const uint32_t sk = w < (uint32_t)kw ? skeys[(size_t)idx * ks + w] : 0u;
            const uint32_t x = ((ents[i].k[w] ^ sk) >> (4 * (b & 7))) & 0xFu;
            if (x) { ++ndiff; pos = b; xnib = x; }
        }
        if (ndiff != best || xnib > 7) return false;
        rel[i] = LdsRelEntry{idx, best, next, xnib, pos};
    }
    return true;
}

inline LdsMemoPlan plan_lds_memo(uint32_t S, uint32_t L, const std::vector<LdsEntry> &ents,
                                 const std::vector<std::vector<uint8_t>> &enc, uint32_t salt_offset = 0,
                                 int salt_trials = 8) {
    LdsMemoPlan plan;
    if (S < 2 || ents.empty() || L > kMemoMaxLen) return plan;
    uint32_t idx_bits = 1;
    while ((1u << idx_bits) < S + 1) ++idx_bits;
    if (idx_bits > kLdsMaxIdxBits) return plan;
    const int kw = L <= 8 ? 1 : (L <= 16 ? 2 : (L <= 24 ? 3 : 4));
    const int ks = kw >= 3 ? 4 : kw;
    std::vector<uint32_t> skeys;
    std::vector<LdsRelEntry> rel;
    if (!lds_relative_entries(S, L, ents, enc, kw, ks, skeys, rel)) return plan;
    std::vector<uint32_t> fields(ents.size());
    for (size_t i = 0; i < ents.size(); ++i) fields[i] = lds_entry_fields(rel[i].idx, rel[i].best, rel[i].next, rel[i].xnib, rel[i].pos);
    const uint32_t fp_mask = lds_fp_mask(idx_bits, kw);
    const uint32_t empty = S;   // idx = S (the sentinel key row), everything else 0
    uint64_t nslots = 256;
    while ((double)nslots * 0.86 < (double)ents.size()) nslots <<= 1;
    const size_t fixed = skeys.size() * 4 + 1024 + (size_t)(S + 1) * 4;   // keys + LUT + histogram
    bool pow2 = true;
    if (nslots * 4 + fixed > kLdsMemoMaxBytes) {
        // the power of two does not fit one CU's LDS: take every slot that does (multiply-shift slot
        // mapping in the kernel, a few more VALU ops per read) if that leaves the cuckoo table <= 0.88 full
        const uint64_t room = (kLdsMemoMaxBytes - fixed) / 4;
        nslots = room < 65535 ? room : 65535;
        nslots &= ~3ull;                                   // keep the sample keys 16-byte aligned
        pow2 = false;
        if ((double)nslots * 0.88 < (double)ents.size()) return plan;
    }
    std::vector<int64_t> owner;
    std::vector<uint32_t> h(ents.size() * 3), fps(ents.size());
    // The salt matters for speed, not only for feasibility.  Most reads ARE a sample barcode, so nearly
    // every wave looks up exact-match keys; if such a key finds a second entry with its fingerprint
    // among its three slots, every wave carrying that sample pays the extra verification round
    // (measured on MI355X: -20 % at S = 24 for ONE such key).  So several salts are built and the one
    // with the fewest multi-match exact keys (then the fewest multi-match keys overall) is kept.
    const int kSaltTrials = salt_trials < 1 ? 1 : salt_trials;
    LdsMemoPlan best;
    uint64_t best_score = ~0ull;
    int successes = 0, failures = 0;
    for (int attempt = 0; attempt < 12 + kSaltTrials && successes < kSaltTrials; ++attempt) {
        if (nslots * 4 + fixed > kLdsMemoMaxBytes || (pow2 && nslots > 32768)) break;   // does not fit one CU's LDS
        const uint32_t mask_b = pow2 ? (uint32_t)(nslots - 1) << 2 : 0u;
        const uint32_t salt = 0x51ED27u * (uint32_t)(attempt + 1 + salt_offset);
        owner.assign(nslots, -1);
        for (size_t i = 0; i < ents.size(); ++i)
            memo_hash3(ents[i].k[0], ents[i].k[1], ents[i].k[2], ents[i].k[3], salt, h[3 * i], h[3 * i + 1], h[3 * i + 2], fps[i]);
        auto slot_of = [&](size_t i, int c) {
            uint32_t a[3];
            lds_slots(pow2, h[3 * i], h[3 * i + 1], mask_b, (uint32_t)nslots, a[0], a[1], a[2]);
            return a[c] >> 2;
        };
        auto fp_of = [&](size_t i) { return fps[i] & fp_mask; };
        uint64_t rng = 0x9E3779B97F4A7C15ull ^ salt;
        auto next_rand = [&]() { rng = rng * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(rng >> 33); };
        // cuckoo insert of entry `cur` (random-walk eviction); false = gave up
        auto insert = [&](int64_t cur) {
            int64_t avoid = -1;
            for (int kick = 0; kick < 5000; ++kick) {
                for (int c = 0; c < 3; ++c) {
                    const uint32_t p = slot_of((size_t)cur, c);
                    if (owner[p] < 0) { owner[p] = cur; return true; }
                }
                uint32_t p = 0;
                int tries = 0;
                do { p = slot_of((size_t)cur, (int)(next_rand() % 3u)); } while ((int64_t)p == avoid && ++tries < 8);
                std::swap(cur, owner[p]);
                avoid = p;
            }
            return false;
        };
        bool ok = true;
        for (size_t i = 0; i < ents.size() && ok; ++i) ok = insert((int64_t)i);
        const bool clean = ok;
        if (clean) {
            plan.image.assign(nslots, empty);
            for (uint64_t p = 0; p < nslots; ++p)
                if (owner[p] >= 0) plan.image[p] = fields[(size_t)owner[p]] | fp_of((size_t)owner[p]);
            plan.image.insert(plan.image.end(), skeys.begin(), skeys.end());
            plan.n_slots = (uint32_t)nslots;
            plan.pow2 = pow2;
            plan.slot_mask_b = mask_b;
            plan.idx_bits = idx_bits;
            plan.skey_off_b = (uint32_t)(nslots * 4);
            plan.salt = salt;
            plan.kw = kw;
            plan.key_stride = ks;
            // self-check: replay the kernel's lookup for every stored key
            bool good = true;
            for (size_t i = 0; i < ents.size() && good; ++i) good = lds_memo_lookup(plan, ents[i].k) == ents[i].val;
            if (good) {
                uint64_t hot_multi = 0, all_multi = 0;
                for (size_t i = 0; i < ents.size(); ++i) {
                    int matches = 0;
                    for (int c = 0; c < 3; ++c) {
                        bool seen = false;   // the same slot reached through two hashes is one entry
                        for (int d = 0; d < c; ++d) seen = seen || slot_of(i, d) == slot_of(i, c);
                        if (!seen && ((plan.image[slot_of(i, c)] ^ fps[i]) & fp_mask) == 0) ++matches;
                    }
                    if (matches > 1) { ++all_multi; if (((ents[i].val >> 16) & 0xFFu) == 0) ++hot_multi; }
                }
                const uint64_t score = (hot_multi << 32) | all_multi;
                ++successes;
                if (score < best_score) { best_score = score; best = plan; best.ok = true; best.multi_score = score; }
                if (score == 0) break;
                continue;
            }
        }
        if (successes == 0 && ++failures % 3 == 0) { if (!pow2) break; nslots <<= 1; }
    }
    return best;
}

// The minimal-perfect-hash form (memo_hash.hpp): for key widths of three and four words whose entries do not fit as cuckoo slots.
// Hash-and-displace: the keys are dealt into buckets by one hash; the buckets, largest first, each take the first displacement d
// (a 16-bit word kept in LDS) under which mph_slot sends all their keys to free slots.  ok = false: not of the shape, or no room.
inline LdsMemoPlan plan_lds_memo_mph(uint32_t S, uint32_t L, const std::vector<LdsEntry> &ents,
                                     const std::vector<std::vector<uint8_t>> &enc, uint32_t salt_offset = 0) {
    LdsMemoPlan plan;
    if (S < 2 || ents.empty() || L <= 16 || L > kMemoMaxLen || S + 1 > (1u << kMphIdxBits)) return plan;
    const int kw = L <= 24 ? 3 : 4, ks = 4;
    std::vector<uint32_t> skeys;
    std::vector<LdsRelEntry> rel;
    if (!lds_relative_entries(S, L, ents, enc, kw, ks, skeys, rel)) return plan;
    const size_t n = ents.size();
    const size_t fixed = skeys.size() * 4 + 1024 + (size_t)(S + 1) * 4;   // keys + LUT + histogram, as plan_lds_memo
    // buckets: a power of two with at most ~5 keys each (half as many when the displacements would take the entries' room: the search
    // for a bucket's displacement gets longer with its size, not hopeless); slots: every one that fits, up to 1/0.90 of the keys (the
    // fuller the table, the longer the search for the last buckets' displacements -- still well under a second at 0.98)
    uint32_t buckets = 64;
    while ((uint64_t)buckets * 5 < n) buckets <<= 1;
    if (buckets > 32768) return plan;
    auto pad16 = [](size_t v) { return (v + 15) & ~(size_t)15; };
    size_t slots = 0;
    for (int shrink = 0; shrink < 2; ++shrink, buckets >>= 1) {
        if (fixed + 2 * (size_t)buckets + 48 >= kLdsMemoMaxBytes) continue;
        const size_t room_slots = (kLdsMemoMaxBytes - fixed - 2 * (size_t)buckets - 48) / 3;
        slots = std::min<size_t>(room_slots, (n * 100 + 89) / 90) & ~(size_t)15;
        if ((double)slots * 0.985 >= (double)n) break;
        slots = 0;
    }
    if (!slots) return plan;
    std::vector<uint32_t> ha(n), hb(n), order(buckets), start(buckets + 1), members(n), disp(buckets);
    std::vector<int64_t> owner;
    std::vector<uint32_t> trial;
    for (int attempt = 0; attempt < 6; ++attempt) {
        const uint32_t salt = 0x51ED27u * (uint32_t)(attempt + 1 + salt_offset);
        for (size_t i = 0; i < n; ++i) mph_hashes(ents[i].k[0], ents[i].k[1], ents[i].k[2], ents[i].k[3], salt, ha[i], hb[i]);
        std::fill(start.begin(), start.end(), 0u);
        for (size_t i = 0; i < n; ++i) ++start[(ha[i] & (buckets - 1)) + 1];
        for (uint32_t b = 0; b < buckets; ++b) start[b + 1] += start[b];
        {
            std::vector<uint32_t> fill(start.begin(), start.end() - 1);
            for (size_t i = 0; i < n; ++i) members[fill[ha[i] & (buckets - 1)]++] = (uint32_t)i;
        }
        for (uint32_t b = 0; b < buckets; ++b) order[b] = b;
        std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return start[x + 1] - start[x] > start[y + 1] - start[y]; });
        owner.assign(slots, -1);
        bool ok = true;
        uint32_t max_d = 0;
        for (uint32_t oi = 0; oi < buckets && ok; ++oi) {
            const uint32_t b = order[oi], lo = start[b], hi = start[b + 1];
            if (lo == hi) { disp[b] = 0; continue; }
            bool placed = false;
            for (uint32_t d = 0; d <= kMphMaxDisplacement && !placed; ++d) {
                trial.clear();
                bool fits = true;
                for (uint32_t j = lo; j < hi && fits; ++j) {
                    const uint32_t sl = mph_slot(ha[members[j]], hb[members[j]], d, (uint32_t)slots);
                    fits = owner[sl] < 0 && std::find(trial.begin(), trial.end(), sl) == trial.end();
                    trial.push_back(sl);
                }
                if (!fits) continue;
                for (uint32_t j = lo; j < hi; ++j) owner[trial[j - lo]] = members[j];
                disp[b] = d;
                max_d = std::max(max_d, d);
                placed = true;
            }
            ok = placed;   // (two keys of a bucket with the same pair of hashes can never be parted: another salt)
        }
        if (!ok) continue;
        const size_t t8_off = pad16(2 * slots), aux_off = pad16(t8_off + slots), skey_off = pad16(aux_off + 2 * (size_t)buckets);
        plan.image.assign(skey_off / 4, 0u);
        uint8_t *bytes = reinterpret_cast<uint8_t *>(plan.image.data());
        for (size_t sl = 0; sl < slots; ++sl) {
            uint32_t e = mph_entry_fields(S, 0, 0, 0, 0);   // empty: the sentinel key row, which equals no key
            if (owner[sl] >= 0) {
                const LdsRelEntry &r = rel[(size_t)owner[sl]];
                e = mph_entry_fields(r.idx, r.best, r.next, r.xnib, r.pos);
            }
            bytes[2 * sl] = (uint8_t)e;
            bytes[2 * sl + 1] = (uint8_t)(e >> 8);
            bytes[t8_off + sl] = (uint8_t)(e >> 16);
        }
        for (uint32_t b = 0; b < buckets; ++b) {
            bytes[aux_off + 2 * (size_t)b] = (uint8_t)disp[b];
            bytes[aux_off + 2 * (size_t)b + 1] = (uint8_t)(disp[b] >> 8);
        }
        plan.image.insert(plan.image.end(), skeys.begin(), skeys.end());
        plan.mph = true;
        plan.pow2 = false;
        plan.n_slots = (uint32_t)slots;
        plan.slot_mask_b = 0;
        plan.idx_bits = kMphIdxBits;
        plan.t8_off_b = (uint32_t)t8_off;
        plan.aux_off_b = (uint32_t)aux_off;
        plan.bucket_mask = buckets - 1;
        plan.skey_off_b = (uint32_t)skey_off;
        plan.salt = salt;
        plan.kw = kw;
        plan.key_stride = ks;
        plan.max_displacement = max_d;
        bool good = true;   // self-check: replay the kernel's lookup for every stored key
        for (size_t i = 0; i < n && good; ++i) good = lds_memo_lookup(plan, ents[i].k) == ents[i].val;
        if (good) { plan.ok = true; return plan; }
        plan = LdsMemoPlan{};
    }
    return plan;
}

}  // namespace fqtk
