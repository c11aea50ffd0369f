// lds_memo_kernels.hip.h -- the LDS-resident form of the complete memo (gfx950).
//
// memo_kernels.hip.h keeps the memo in HBM/L2 and pays one dependent L2 gather per read that is not an
// exact sample barcode; measured on MI355X that gather phase is ~30 % of the kernel and is what makes
// it occupancy-hungry (rocprofv3 + tools/ablate.sh).  When every sample barcode
// is plain A/C/G/T and max_mismatches <= 1 -- fqtk's defaults and the common case -- a memo entry is
// fully described RELATIVE to the sample it resolves to:
//
//     the read is sample idx's barcode, possibly with ONE base replaced at position pos
//
// so an entry shrinks from 16 bytes (key + value) to ONE dword
//
//     [ fp | next : 5 | pos : 4-5 | xnib : 3 | best : 1 | fp | idx : IB ]    xnib = code(read base) ^ code(sample base)
//
// (idx, best, next sit where fqtk_match_t wants them: entry & mask IS the result, memo_hash.hpp)
//
// and the whole table (cfg 3: 24 960 entries -> 128 KiB) fits the CU's 160 KiB LDS next to the S
// sample keys.  A lookup is three independent ds_read_b32 (3-choice cuckoo), a fingerprint select,
// one ds_read_b64 of the candidate sample's key and an exact check
//
//     key ^ sample_key[idx] == xnib << 4*pos
//
// -- no vector-memory gather at all: the only HBM/L2 traffic left is the barcode stream in and the
// result stream out.  Exactness does not rest on the fingerprint: it only picks which candidate to
// VERIFY against the sample key first; if that fails and another slot carries the same fingerprint
// (rare) it is verified too.  No candidate verifies = "not in the memo" = None, as in memo_kernels.hip.h.
//
// One workgroup of 1024 lanes per CU (two when the table is small): the table is loaded once per
// workgroup; without gathers 16 waves/CU stream as fast as 32 (measured).
//
// FORM of the entry table (a template parameter): kLdsFormAny / kLdsFormPow2 = the three-choice cuckoo table above, its slot count
// any number / a power of two; kLdsFormMph (round 6, three and four key words) = a minimal perfect hash over three-byte entries
// (memo_hash.hpp, lds_memo_plan.hpp plan_lds_memo_mph): ds_read_u16 of the key's bucket displacement, ds_read_u16 + ds_read_u8 of
// the ONE slot the key can live in, the same check against the sample's key -- for the tables whose four-byte slots overflow LDS.
#pragma once
#include <type_traits>

#include "memo_kernels.hip.h"

namespace fqtk {

#ifndef FQTK_LDS_BLOCK
#define FQTK_LDS_BLOCK 1024
#endif
constexpr int kLdsBlock = FQTK_LDS_BLOCK;
constexpr int kLdsFormAny = 0, kLdsFormPow2 = 1, kLdsFormMph = 2;
// Developer ablations of the look-up (tools/ab_ldsm_ablate.sh), compile-time: 1 = no table look-up at all (the
// result is a fold of the key), 2 = no histogram, 4 = the first candidate is taken unverified.  0 in the product.
#ifndef FQTK_LDSM_ABL
#define FQTK_LDSM_ABL 0
#endif
struct LdsMemoParams {
    MatchParams m;
    const uint32_t *image;    // [n_slots] entries, then (S + 1) sample keys of key_stride words each
    uint32_t slot_mask_b;     // (n_slots - 1) << 2: byte-address mask of the entry table (power-of-two tables); kLdsFormMph: buckets - 1
    uint32_t n_slots;         // slot count (any-size tables: slot = hash16 * n_slots >> 16)
    uint32_t idx_bits;        // IB
    uint32_t image_words;     // dwords to stage into LDS
    uint32_t skey_off_b;      // byte offset of the sample keys inside LDS
    uint32_t salt;            // hash salt the builder settled on
    uint32_t hist_shift;      // the LDS histogram holds 1 << hist_shift copies of every bin (lane & mask picks one):
                              // with few samples the lanes of a wave pile up on a few counters, and same-address
                              // LDS atomics of one instruction run one after the other
    uint32_t t8_off_b;        // kLdsFormMph: byte offset of the entries' third bytes (the first two, 16-bit words, sit at byte 0) ...
    uint32_t aux_off_b;       // ... and of the buckets' 16-bit displacements
};

// Raw LDS accesses by BYTE ADDRESS.  The kernel has no static LDS, so its dynamic LDS starts at address
// 0 (checked once per workgroup) and a masked hash can be used as the ds_read address as is -- no base
// add per probe.
typedef __attribute__((address_space(3))) const uint32_t lds_u32;
typedef __attribute__((address_space(3))) const u32x2v lds_u2;
typedef __attribute__((address_space(3))) const u32x4v lds_u4;
__device__ __forceinline__ uint32_t lds_word(uint32_t byte_addr) {
    return *reinterpret_cast<lds_u32 *>((uintptr_t)byte_addr);
}
__device__ __forceinline__ uint32_t lds_half(uint32_t byte_addr) {
    return *reinterpret_cast<__attribute__((address_space(3))) const uint16_t *>((uintptr_t)byte_addr);
}
__device__ __forceinline__ uint32_t lds_byte(uint32_t byte_addr) {
    return *reinterpret_cast<__attribute__((address_space(3))) const uint8_t *>((uintptr_t)byte_addr);
}
__device__ __forceinline__ void lds_atomic_inc(uint32_t byte_addr) {
    __hip_atomic_fetch_add(reinterpret_cast<__attribute__((address_space(3))) uint32_t *>((uintptr_t)byte_addr), 1u,
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// mask ? a : b with the condition given as a 64-lane mask in SGPRs (one v_cndmask_b32)
__device__ __forceinline__ uint32_t lane_select(uint64_t mask, uint32_t a, uint32_t b) {
    uint32_t d;
    asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(d) : "v"(b), "v"(a), "s"(mask));
    return d;
}

// PF: software pipeline of the full-tile loop -- the loads of tile t + grid are issued BEFORE tile t is
// looked up, so a wave always has a load in flight while it computes.  With R = 1 this is the stream shape
// the memory system likes best (tools/hbm_stream.hip: one 1-KiB request per wave at a time, 256 tiles = a
// 4-MiB window sweeping the buffer) without the compute latency serialised behind every load.
// INDEXED: the second pass.  The first pass (any instantiation without it) lists the reads that carry a byte of no IUPAC
// meaning (ambiguity codes and '.' get N's code where they stand: recode_flagged_bytes; until round 6 they were listed too)
// and gives them a placeholder result; this one takes the listed reads -- each wave the segments of the worklist it owns, 64
// reads at a time -- and scans them in place, a wave per read.
template <int VEC, int KW, int R, int FORM, bool LENS = false, bool PF = false, bool INDEXED = false>   // LENS: see memo_kernel
__global__ __launch_bounds__(kLdsBlock) __attribute__((amdgpu_waves_per_eu(4, 8)))
void lds_memo_kernel(const LdsMemoParams Q) {
    static_assert(!INDEXED || (VEC <= 0 && !LENS && !PF), "the second pass gathers single rows");
    static_assert(FORM != kLdsFormMph || KW >= 3, "the perfect-hash form is planned for three and four key words");
    const MatchParams &P = Q.m;
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    // LDS: [entry table | sample keys | spread LUT (fallback scan) | histogram]; the entry table sits
    // at byte 0 so a masked hash is used as the ds_read address as is
    constexpr int KS = KW >= 3 ? 4 : KW;                      // key stride in dwords (b128 reads for KW 3 and 4)
    const uint32_t tid = threadIdx.x;
    if constexpr (INDEXED) {   // nothing listed for this workgroup's waves (the usual case): leave before staging anything
        uint32_t any = 0;
        for (uint32_t seg = blockIdx.x * (kLdsBlock / 64u) + (tid >> 6); seg < P.work_segs; seg += gridDim.x * (kLdsBlock / 64u))
            any |= P.work_n[seg];
        // (no __syncthreads_or: it brings static LDS with it, and the table may need all of the dynamic 160 KiB)
        if (tid == 0) smem[0] = 0;
        __syncthreads();
        if (any) smem[0] = 1;
        __syncthreads();
        const bool listed = smem[0] != 0;
        __syncthreads();
        if (!listed) return;
    }
    for (uint32_t w = tid; w < Q.image_words; w += kLdsBlock) smem[w] = Q.image[w];
    uint32_t *lds_lut = smem + Q.image_words;
    uint32_t *lds_hist = lds_lut + 256;
    if (tid < 256) lds_lut[tid] = P.lut[tid];
    const uint32_t bins = P.S + 1;
    const uint32_t hist_words = bins << Q.hist_shift;
    // [S][1][4] planes for the wave scan of non-canonical reads, 16-byte aligned behind the histogram
    uint32_t *lds_tab = P.scan_tab_lds ? smem + ((Q.image_words + 256u + ((P.counts && P.lds_hist) ? hist_words : 0u) + 3u) & ~3u) : nullptr;
    if (P.counts && P.lds_hist)
        for (uint32_t b = tid; b < hist_words; b += kLdsBlock) lds_hist[b] = 0;
    if (lds_tab)
        for (uint32_t w = tid; w < P.S * 4u; w += kLdsBlock) lds_tab[w] = P.table[w];
    __syncthreads();
    if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)smem != 0u) __builtin_trap();

    const uint32_t L = P.L;
    const uint32_t nwords = (L + 3u) >> 2;
    constexpr int NWD = VEC >= 1 ? VEC : 2 * KW;
    static_assert(NWD <= 2 * KW, "key too narrow for the load width");
    uint32_t kc[NWD], kv[NWD];
#pragma unroll
    for (int w = 0; w < NWD; ++w) {
        const int rem = (int)L - 4 * w;
        const uint32_t keep = rem >= 4 ? 0xFFFFFFFFu : (rem <= 0 ? 0u : ((1u << (8 * rem)) - 1u));
        kc[w] = keep & 0x07070707u;
        kv[w] = keep & 0xDFDFDFDFu;
    }
    const uint32_t fp_mask = lds_fp_mask(Q.idx_bits, KW);     // ((entry ^ g) & fp_mask) == 0 <=> fingerprints agree
    const uint32_t idx_mask = (1u << Q.idx_bits) - 1u;
    const uint32_t res_mask = lds_res_mask(Q.idx_bits);       // entry & res_mask = the fqtk_match_t word
    const uint64_t tile = (uint64_t)kLdsBlock * R;
    const uint64_t ntiles = (P.n + tile - 1) / tile;
    // per-lane byte offsets inside a tile: loop-invariant, so a full tile's loads and stores are
    // "uniform 64-bit base (SGPRs) + 32-bit lane offset" with no per-read address arithmetic
    // Lane -> read inside a tile is WAVE-CONTIGUOUS: a wave's R loads cover R x 64 consecutive reads (R KiB
    // in a row at 16 B/read), not R slices a whole workgroup apart.  tools/hbm_stream.hip: the same
    // 16 B-in / 4 B-out stream with nothing else runs at 5.4 TB/s in this order vs 4.9 TB/s block-strided
    // (8 B-in: 5.3 vs 3.9 TB/s) -- DRAM locality across the chip's 256 concurrent tiles.
    uint32_t local[R], in_off[R], out_off[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        local[r] = (tid >> 6) * (64u * R) + (uint32_t)r * 64u + (tid & 63u);
        in_off[r] = local[r] * P.stride;
        out_off[r] = local[r] * 4u;
    }
    const uint32_t hist_on = (P.counts && P.lds_hist) ? 1u : 0u;
    // this lane's copy of bin b: byte address hist_base_b + (b << (hist_shift + 2))
    const uint32_t hist_base_b = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)lds_hist +
                                 ((tid & ((1u << Q.hist_shift) - 1u)) << 2);
    const uint32_t hist_bin_shift = Q.hist_shift + 2u;

    // The packed vector loads of one full tile (every read exists, the rows are VEC dwords).
    auto load_full = [&](uint64_t t, uint32_t (&words)[R][kRowWords]) {
        const uint8_t *tile_in = P.obs + t * tile * (uint64_t)P.stride;   // wave-uniform
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint8_t *src = tile_in + in_off[r];
            if constexpr (VEC == 4) {
                const u32x4v v = FQTK_STREAM_LOAD(reinterpret_cast<const u32x4v *>(src));
                words[r][0] = v.x; words[r][1] = v.y; words[r][2] = v.z; words[r][3] = v.w;
            } else if constexpr (VEC == 3 || VEC == 5 || VEC == 7) {   // 12- / 20- / 28-byte rows are only 4-byte aligned: dword pieces,
                const uint32_t *s32 = reinterpret_cast<const uint32_t *>(src);   // non-temporal like the rest of the stream
#pragma unroll
                for (int w = 0; w < VEC; ++w) words[r][w] = FQTK_STREAM_LOAD(s32 + w);
            } else if constexpr (VEC == 2) {
                const u32x2v v = FQTK_STREAM_LOAD(reinterpret_cast<const u32x2v *>(src));
                words[r][0] = v.x; words[r][1] = v.y;
            } else if constexpr (VEC == 6) {   // 24-byte rows (12 + 12 dual index): three 8-byte pieces
#pragma unroll
                for (int w = 0; w < 3; ++w) {
                    const u32x2v v = FQTK_STREAM_LOAD(reinterpret_cast<const u32x2v *>(src) + w);
                    words[r][2 * w] = v.x; words[r][2 * w + 1] = v.y;
                }
            } else if constexpr (VEC == 8) {   // 32-byte rows: two 16-byte pieces
#pragma unroll
                for (int w = 0; w < 2; ++w) {
                    const u32x4v v = FQTK_STREAM_LOAD(reinterpret_cast<const u32x4v *>(src) + w);
                    words[r][4 * w] = v.x; words[r][4 * w + 1] = v.y; words[r][4 * w + 2] = v.z; words[r][4 * w + 3] = v.w;
                }
            } else {
                words[r][0] = FQTK_STREAM_LOAD(reinterpret_cast<const uint32_t *>(src));
            }
            // a variable-length batch: the read's length travels with its row (the buffer's last word), in the same
            // group of loads -- so these batches take the pipelined loop too
            if constexpr (LENS) words[r][kLenWord] = FQTK_STREAM_LOAD(P.lens + t * tile + local[r]);
        }
    };
    // Any tile through the generic path (ragged last tile, unaligned strides): bounds-checked loads.
    auto load_any = [&](uint64_t t, uint32_t (&words)[R][kRowWords], bool (&live)[R]) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint64_t i = t * tile + local[r];
            live[r] = i < P.n;
#pragma unroll
            for (int w = 0; w < 8; ++w) words[r][w] = 0x41414141u;   // dead lanes look like "AAAA"
            if (live[r]) load_words<1, VEC, kRowWords>(P, i, nwords, words[r]);
            if constexpr (LENS) words[r][kLenWord] = live[r] ? P.lens[i] : L;
        }
    };

    // this wave's segment of the second pass's worklist, and how much of it is used
    const uint32_t work_seg = blockIdx.x * (kLdsBlock / 64u) + (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));   // wave-uniform: SGPRs
    uint32_t work_fill = 0;
    // Looks one tile up: res[r] = the result word of the tile's r-th read; also feeds the LDS histogram.
    auto compute = [&](uint64_t t, uint32_t (&words)[R][kRowWords], const bool (&live)[R], uint32_t (&res)[R]) {
        uint32_t bflag[R];
        // One read at a time: hash, three entry reads, select, verify; the rare extra verification
        // rounds sit behind wave-uniform branches.
        uint32_t key[R][4];
        // candidate -> exact check against the sample's own key: key ^ sample_key == xnib << 4*pos
        auto verify = [&](int r, uint32_t e) -> uint32_t {
            const uint32_t ka = Q.skey_off_b + (e & idx_mask) * (KS * 4u);
            const uint32_t xnib = __builtin_amdgcn_ubfe(e, 17, 3);
            const uint32_t tsh = xnib << ((e >> 18) & 28u);        // the differing nibble, in its word
            uint32_t diff;
            if constexpr (KW == 1) {
                diff = key[r][0] ^ lds_word(ka) ^ tsh;
            } else if constexpr (KW == 2) {
                const u32x2v sk = *reinterpret_cast<lds_u2 *>((uintptr_t)ka);
                const bool w1 = (e & (1u << 23)) != 0;
                diff = (key[r][0] ^ sk.x ^ (w1 ? 0u : tsh)) | (key[r][1] ^ sk.y ^ (w1 ? tsh : 0u));
            } else {   // the differing nibble's word: bit 23 = pos bit 3, bit 29 = pos bit 4
                const u32x4v sk = *reinterpret_cast<lds_u4 *>((uintptr_t)ka);
                const bool w1 = (e & (1u << 23)) != 0, w2 = (e & (1u << 29)) != 0;
                diff = (key[r][0] ^ sk.x ^ ((w1 || w2) ? 0u : tsh)) | (key[r][1] ^ sk.y ^ ((w1 && !w2) ? tsh : 0u)) |
                       (key[r][2] ^ sk.z ^ ((w2 && !w1) ? tsh : 0u));
                if constexpr (KW == 4) diff |= key[r][3] ^ sk.w ^ ((w1 && w2) ? tsh : 0u);
            }
            return diff == 0 ? (e & res_mask) : kMemoEmpty;
        };
        // the same check for a three-byte entry of the perfect-hash form: idx 0-8 | best 9 | next 10-14 | xnib 15-17 | pos 18-22
        auto verify_mph = [&](int r, uint32_t e) -> uint32_t {
            const uint32_t ka = Q.skey_off_b + (e & ((1u << kMphIdxBits) - 1u)) * (KS * 4u);
            const uint32_t tsh = __builtin_amdgcn_ubfe(e, 15, 3) << ((e >> 16) & 28u);
            const u32x4v sk = *reinterpret_cast<lds_u4 *>((uintptr_t)ka);
            const bool w1 = (e & (1u << 21)) != 0, w2 = (e & (1u << 22)) != 0;
            uint32_t diff = (key[r][0] ^ sk.x ^ ((w1 || w2) ? 0u : tsh)) | (key[r][1] ^ sk.y ^ ((w1 && !w2) ? tsh : 0u)) |
                            (key[r][2] ^ sk.z ^ ((w2 && !w1) ? tsh : 0u));
            if constexpr (KW == 4) diff |= key[r][3] ^ sk.w ^ ((w1 && w2) ? tsh : 0u);
            return diff == 0 ? mph_entry_result(e) : kMemoEmpty;
        };
        // ---- ASCII -> codes; the rare bytes that are not A C G T N get their codes where they stand (recode_flagged_bytes: this
        //      form is for plain A/C/G/T samples only), behind wave-uniform branches: one test per tile, one per word of a tile that
        //      holds such a byte, ~30 instructions for that word.  bflag is left set only by bytes of no IUPAC meaning.
        uint32_t cw[R][NWD];
        uint32_t any_odd = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            encode_codes<NWD, (VEC >= 1)>(words[r], kc, kv, cw[r], bflag[r]);
            any_odd |= bflag[r];
        }
        if (__builtin_amdgcn_uicmp(any_odd, 0u, 33 /* ne */)) {   // wave-uniform
#pragma unroll
            for (int r = 0; r < R; ++r) {
                bflag[r] = 0;
#pragma unroll
                for (int w = 0; w < NWD; ++w) {
                    const bool full = (VEC >= 1) && w < NWD - 1;
                    uint32_t x = flagged_bytes_of_word(words[r][w], cw[r][w], full ? 0xDFDFDFDFu : kv[w]);
                    if (__builtin_amdgcn_uicmp(x, 0u, 33)) x = recode_flagged_bytes(words[r][w], x, cw[r][w], full ? 0x07070707u : kc[w]);
                    bflag[r] |= x;
                }
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            assemble_key<NWD>(cw[r], key[r]);
            if constexpr ((FQTK_LDSM_ABL & 1) != 0) { res[r] = (key[r][0] ^ (KW >= 2 ? key[r][1] : 0u)) | 0xFFFFu; continue; }
            if constexpr (FORM == kLdsFormMph) {
                uint32_t ha, hb;
                mph_hashes(key[r][0], key[r][1], key[r][2], KW >= 4 ? key[r][3] : 0u, Q.salt, ha, hb);
                const uint32_t d = lds_half(Q.aux_off_b + ((ha & Q.slot_mask_b) << 1));
                const uint32_t slot = mph_slot(ha, hb, d, Q.n_slots);
                res[r] = verify_mph(r, lds_half(slot << 1) | (lds_byte(Q.t8_off_b + slot) << 16));
                continue;
            }
            uint32_t h1, h2, h3, fps;
            memo_hash3(key[r][0], KW >= 2 ? key[r][1] : 0u, KW >= 3 ? key[r][2] : 0u, KW >= 4 ? key[r][3] : 0u, Q.salt, h1, h2, h3, fps);
            (void)h3;
            uint32_t a1, a2, a3;
            lds_slots(FORM == kLdsFormPow2, h1, h2, Q.slot_mask_b, Q.n_slots, a1, a2, a3);
            const uint32_t e1 = lds_word(a1), e2 = lds_word(a2), e3 = lds_word(a3);
            // fingerprint matches as LANE MASKS (SGPR pairs): the "more than one match" test below is
            // then scalar ALU + one scalar branch instead of per-lane selects
            const uint64_t M1 = __builtin_amdgcn_uicmp((e1 ^ fps) & fp_mask, 0u, 32 /* eq */);
            const uint64_t M2 = __builtin_amdgcn_uicmp((e2 ^ fps) & fp_mask, 0u, 32);
            const uint64_t M3 = __builtin_amdgcn_uicmp((e3 ^ fps) & fp_mask, 0u, 32);
            // the first fingerprint match in probe order (e3 if none: it then cannot verify either) ...
            uint32_t v = (FQTK_LDSM_ABL & 4) ? (lane_select(M1, e1, lane_select(M2, e2, e3)) & res_mask)
                                             : verify(r, lane_select(M1, e1, lane_select(M2, e2, e3)));
            // ... and, rarely (two entries among the three slots share the fingerprint: ~0.1 % of lanes),
            // the later matches
            const uint64_t multi = (M1 & (M2 | M3)) | (M2 & M3);
            if (multi) {   // wave-uniform, scalar
                const bool m1 = ((e1 ^ fps) & fp_mask) == 0, m2 = ((e2 ^ fps) & fp_mask) == 0, m3 = ((e3 ^ fps) & fp_mask) == 0;
                const bool need2 = v == kMemoEmpty && ((m1 && (m2 || m3)) || (m2 && m3));
                const uint32_t v2 = verify(r, (m1 && m2) ? e2 : e3);
                if (need2) v = v2;
                const bool need3 = need2 && v2 == kMemoEmpty && m1 && m2 && m3;
                const uint32_t v3 = verify(r, e3);
                if (need3) v = v3;
            }
            res[r] = v;
        }
        // ---- variable-length batches: the memo served the reads of length L; shorter -> None
        //      (barcode_matching.rs:167-169), longer -> None or the reference's panic (:170-172, :95-107)
        if constexpr (LENS) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (!live[r]) continue;
                const uint32_t len = words[r][kLenWord];
                if (len != L) {
                    res[r] = kMemoEmpty;
                    bflag[r] = 0;
                    if (len > L) overlong_read(P, t * tile + local[r], len);
                }
            }
        }
        // ---- rare: non-canonical reads -> wave-cooperative exhaustive scan ---------------------
        uint32_t any_bad = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) any_bad |= live[r] ? bflag[r] : 0u;
        if (__builtin_amdgcn_uicmp(any_bad, 0u, 33 /* ne */)) {   // wave-uniform; one test per tile
#pragma unroll
            for (int r = 0; r < R; ++r) {
                // no-calls and ambiguity codes were looked up under N's key already: only bytes of no IUPAC meaning need the scan
                const uint32_t really = (live[r] && bflag[r]) ? 1u : 0u;
                uint64_t todo = __builtin_amdgcn_uicmp(really, 0u, 33);
                if constexpr (!INDEXED)
                    todo = defer_to_second_pass(P, work_seg, work_fill, todo, really != 0u, t * tile + local[r], res[r], words[r]);   // normally all of them
                if (todo) {
                    Planes<1> mine;
                    encode_planes<1>(words[r], nwords, L, lds_lut, mine);
                    while (todo) {
                        const int src = __ffsll((unsigned long long)todo) - 1;
                        todo &= todo - 1;
                        uint32_t b, s;
                        wave_scan<1>(mine, src, P, b, s, lds_tab);
                        if ((int)__lane_id() == src) res[r] = decide(b, s, P.max_mm, P.delta);
                    }
                }
            }
        }
        // ---- per-sample counts ---------------------------------------------------------------------
        if (P.counts && !(FQTK_LDSM_ABL & 2)) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (!live[r] || res[r] == kMemoDeferred) continue;
                const uint32_t bin = min(res[r] & 0xFFFFu, P.S);   // None (0xFFFF) -> bin S
                if (hist_on) lds_atomic_inc(hist_base_b + (bin << hist_bin_shift));
                else atomicAdd(&P.counts[bin], 1ull);
            }
        }
    };
    // The result stream of one tile.
    auto store_full = [&](uint64_t t, const uint32_t (&res)[R]) {
        uint8_t *tile_out = reinterpret_cast<uint8_t *>(P.out + t * tile);   // wave-uniform
#pragma unroll
        for (int r = 0; r < R; ++r) FQTK_STREAM_STORE(res[r], reinterpret_cast<uint32_t *>(tile_out + out_off[r]));
    };
    auto store_any = [&](uint64_t t, const uint32_t (&res)[R], const bool (&live)[R]) {
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (live[r]) FQTK_STREAM_STORE(res[r], &P.out[t * tile + local[r]]);
    };

    if constexpr (INDEXED) {
        const uint32_t lane = tid & 63u;
        const uint32_t waves = gridDim.x * (kLdsBlock / 64u);
        for (uint32_t seg = work_seg; seg < P.work_segs; seg += waves) {   // wave-private: no barrier in here
            const uint32_t filled = P.work_n[seg];
            const uint32_t cnt = filled < P.work_cap ? filled : P.work_cap;
            if (cnt == 0) continue;
            const uint32_t ew = 1u + P.work_rw;   // an entry: the read's index, then (work_rw dwords of) its row
            const uint32_t *list = P.work + (uint64_t)seg * P.work_cap * ew;
            // R reads per lane: a wave's iteration is a chain of dependent memory round trips (list entry [, row], result
            // store) and the R of them overlap
            for (uint32_t base = 0; base < cnt; base += 64u * R) {
                uint32_t words[R][kRowWords], res[R];
                bool live[R];
                uint64_t row[R];
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const uint32_t j = base + (uint32_t)r * 64u + lane;
                    live[r] = j < cnt;
                    row[r] = live[r] ? list[(uint64_t)j * ew] : 0;
#pragma unroll
                    for (int w = 0; w < 8; ++w) words[r][w] = 0x41414141u;
                    if (P.work_rw && live[r]) {   // wave-uniform: the rows stream in with the indices
                        const uint32_t *e = list + (uint64_t)j * ew + 1u;
                        if (P.work_rw == 4) {
                            const u32x4v v = *reinterpret_cast<const u32x4v *>(e);
                            words[r][0] = v.x; words[r][1] = v.y; words[r][2] = v.z; words[r][3] = v.w;
                        } else {
#pragma unroll
                            for (int w = 0; w < 8; ++w)
                                if ((uint32_t)w < P.work_rw && (uint32_t)w < nwords) words[r][w] = e[w];
                        }
                    }
                }
                if (!P.work_rw) {
#pragma unroll
                    for (int r = 0; r < R; ++r)
                        if (live[r]) load_words<1, VEC, kRowWords>(P, row[r], nwords, words[r]);
                }
                compute(0, words, live, res);
#pragma unroll
                for (int r = 0; r < R; ++r)
                    if (live[r]) FQTK_STREAM_STORE(res[r], &P.out[row[r]]);
            }
            if (lane == 0) P.work_n[seg] = 0;   // all zero again for the next launch
        }
    }
    const uint64_t full_tiles = (VEC >= 1 && !INDEXED) ? P.n / tile : 0;
    bool all_live[R];
#pragma unroll
    for (int r = 0; r < R; ++r) all_live[r] = true;
    if constexpr (INDEXED) {
    } else if constexpr (PF && VEC >= 1) {
        // Software pipeline, one tile deep on both streams.  gfx950 counts loads AND stores in vmcnt and
        // they complete out of order with respect to each other, so "wait for my loads" also waits for every
        // store issued since: the plain loop (load, look up, store) pays a store acknowledgement plus a load
        // round trip, back to back, per tile.  Here a wave waits ONCE per tile, at the top, for operations it
        // issued a whole look-up phase earlier (the loads of this tile, the stores of the tile before), and only
        // then issues the next tile's loads and the previous tile's stores, which fly during the look-up.
        // Two word buffers used alternately (no register rotation); the prefetch is unconditional -- past the
        // end it re-reads the last full tile and the words are never used -- so that it cannot sit in a branch.
        uint32_t wa[R][kRowWords], wb[R][kRowWords], held[R];
        auto landed = [&](uint32_t (&w)[R][kRowWords]) {   // the words are in registers; nothing below moves above this point
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int k = 0; k < NWD; ++k) asm volatile("" : "+v"(w[r][k]) : : "memory");
            if constexpr (LENS) {
#pragma unroll
                for (int r = 0; r < R; ++r) asm volatile("" : "+v"(w[r][kLenWord]) : : "memory");
            }
        };
        auto computed = [&](uint32_t (&v)[R]) {     // the results exist now (their gathers / LDS reads were waited for HERE)
#pragma unroll
            for (int r = 0; r < R; ++r) asm volatile("" : "+v"(v[r]) : : "memory");
        };
        uint64_t t = blockIdx.x;
        if (t < full_tiles) {
            const uint64_t last = full_tiles - 1;
            uint32_t res[R];
            load_full(t, wa);
            landed(wa);
            load_full(min(t + gridDim.x, last), wb);
            compute(t, wa, all_live, held);
            computed(held);
            uint64_t t_held = t;
            t += gridDim.x;
            while (t < full_tiles) {
                landed(wb);
                load_full(min(t + gridDim.x, last), wa);
                store_full(t_held, held);
                compute(t, wb, all_live, res);
                computed(res);
#pragma unroll
                for (int r = 0; r < R; ++r) held[r] = res[r];
                t_held = t;
                t += gridDim.x;
                if (t >= full_tiles) break;
                landed(wa);
                load_full(min(t + gridDim.x, last), wb);
                store_full(t_held, held);
                compute(t, wa, all_live, res);
                computed(res);
#pragma unroll
                for (int r = 0; r < R; ++r) held[r] = res[r];
                t_held = t;
                t += gridDim.x;
            }
            store_full(t_held, held);
        }
    } else {
        for (uint64_t t = blockIdx.x; t < full_tiles; t += gridDim.x) {
            uint32_t words[R][kRowWords], res[R];
            load_full(t, words);
            compute(t, words, all_live, res);
            store_full(t, res);
        }
    }
    // whatever is left (the ragged last tile; every tile on the generic load paths)
    for (uint64_t t = full_tiles + (blockIdx.x + gridDim.x - full_tiles % gridDim.x) % gridDim.x; t < (INDEXED ? 0 : ntiles); t += gridDim.x) {
        uint32_t words[R][kRowWords], res[R];
        bool live[R];
        load_any(t, words, live);
        compute(t, words, live, res);
        store_any(t, res, live);
    }
    if constexpr (!INDEXED) publish_worklist_fill(P, work_seg, work_fill);

    if (P.counts && P.lds_hist) {
        __syncthreads();
        for (uint32_t b = tid; b < bins; b += kLdsBlock) {
            uint32_t c = 0;
            for (uint32_t k = 0; k < (1u << Q.hist_shift); ++k) c += lds_hist[(b << Q.hist_shift) + k];
            if (c) atomicAdd(&P.counts[b], (unsigned long long)c);
        }
    }
}

}  // namespace fqtk
