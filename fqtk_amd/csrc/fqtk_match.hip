// fqtk_match.hip -- C-ABI implementation (include/fqtk_match.h) over the gfx950 kernels.
//
// Host side of the drop-in boundary for the reference's BarcodeMatcher
// (/root/reference/src/lib/barcode_matching.rs:29-186).  No CPU compute path exists here: table
// preparation is host work (once per run, as in BarcodeMatcher::new :55-86); every assign goes to
// the device.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

#include <algorithm>
#include <array>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <deque>
#include <vector>

#include "../../include/fqtk_match.h"
#include "match_kernels.hip.h"
#ifdef FQTK_DEV_TIMING
namespace fqtk { __device__ unsigned long long g_dev_phase[16]; }
// developer builds: cycles per phase summed over the waves of every launch since the last call ([15] = waves), then zeroed
extern "C" int fqtk_dev_phase_cycles(unsigned long long *out16) {
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(fqtk::g_dev_phase), 16 * sizeof(unsigned long long)) != hipSuccess) return 1;
    unsigned long long zero[16] = {};
    return hipMemcpyToSymbol(HIP_SYMBOL(fqtk::g_dev_phase), zero, sizeof zero) == hipSuccess ? 0 : 1;
}
#endif
#include "lds_memo_kernels.hip.h"
#include "lds_memo_plan.hpp"
#include "direct_memo_plan.hpp"

namespace {

thread_local std::string g_last_error;

// switches of tests and A/B runs: set and neither empty nor "0"
bool env_flag(const char *name) {
    const char *v = std::getenv(name);
    return v && *v && !(v[0] == '0' && v[1] == '\0');
}

int fail(int code, const std::string &msg) {
    g_last_error = msg;
    return code;
}

#define HIP_TRY(expr)                                                                       \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess) {                                                             \
            return fail(FQTK_EHIP, std::string(#expr) + ": " + hipGetErrorString(_e));      \
        }                                                                                   \
    } while (0)

// enc(): reference src/lib/mod.rs:26-61.  'N','n','.' -> 15 (checked before upper-casing), else the
// IUPAC mask of the upper-cased byte, else 0.
uint8_t enc_byte(uint8_t b) {
    if (b == 'N' || b == 'n' || b == '.') return 15;
    if (b >= 'a' && b <= 'z') b = (uint8_t)(b - 32);
    switch (b) {
        case 'A': return 1;
        case 'C': return 2;
        case 'G': return 4;
        case 'T': return 8;
        case 'U': return 8;
        case 'M': return 3;
        case 'R': return 5;
        case 'W': return 9;
        case 'S': return 6;
        case 'Y': return 10;
        case 'K': return 12;
        case 'V': return 7;
        case 'H': return 11;
        case 'D': return 13;
        case 'B': return 14;
        default: return 0;
    }
}

// What a latched length error refers to: the batch whose kernel raised it (so that the reference's
// panic sentence -- which quotes the read -- can be rebuilt on the host, barcode_matching.rs:95-107).
struct ErrCtx {
    const uint8_t *obs = nullptr;    // host pointers (enqueue / assign_batch) or device pointers (*_device)
    const uint32_t *lens = nullptr;
    uint32_t stride = 0;
    uint64_t n = 0;
    bool device = false;
};

// Worklist of the memo kernels' non-canonical reads (MatchParams::work): one per stream batches run on.
struct Worklist {
    uint32_t *d_list = nullptr;   // [segments][cap] entries of `ew` dwords: read index + row
    uint32_t *d_fill = nullptr;   // [segments] entries used; zero between launches
    uint32_t cap = 0, ew = 1;
    void release() {
        if (d_list) (void)hipFree(d_list);
        if (d_fill) (void)hipFree(d_fill);
        d_list = d_fill = nullptr;
        cap = 0;
    }
};

struct Slot {
    hipStream_t stream = nullptr;
    uint8_t *d_obs = nullptr;
    size_t obs_cap = 0;
    uint32_t *d_len = nullptr;
    size_t len_cap = 0;  // elements
    uint32_t *d_out = nullptr;
    size_t out_cap = 0;  // elements
    uint8_t *d_packed = nullptr;     // fqtk_matcher_enqueue_packed: the 4-bit rows as they arrive, the exceptions' indices and rows
    size_t packed_cap = 0;
    uint32_t *d_exc_index = nullptr;
    size_t exc_index_cap = 0;
    uint8_t *d_exc_rows = nullptr;
    size_t exc_rows_cap = 0;
    bool busy = false;
    ErrCtx ctx;          // the chunk in flight on this slot
    Worklist work;
};

// Pipeline slots 0..FQTK_MAX_SLOTS-1 belong to the caller (enqueue/wait); two more are private to the
// synchronous fqtk_matcher_assign_batch(), so a caller's chunk in flight is never waited on or
// re-used behind its back.  Every slot has its OWN latched error word (device + pinned mirror), so an
// error is reported by the wait() of the chunk that raised it; one more word serves the *_device entry.
constexpr int kSyncSlot0 = FQTK_MAX_SLOTS;
constexpr int kNumSlots = FQTK_MAX_SLOTS + 2;
constexpr int kDeviceErrWord = kNumSlots;

}  // namespace

struct fqtk_matcher {
    int device = 0;
    uint32_t S = 0, L = 0, NW = 0;
    uint32_t max_mm = 0, delta = 0, max_ns = 0;
    bool plain_samples = false;                // every base of every sample is one of A C G T
    int num_cus = 256;
    uint32_t *d_table = nullptr;
    uint32_t *d_lut = nullptr;
    unsigned long long *d_err = nullptr;     // [kNumSlots + 1] min offending index per slot (+ *_device entry), ~0 = none
    unsigned long long *d_counts = nullptr;       // S+1, accumulator of the enqueue()/wait() pipeline
    unsigned long long *d_counts_sync = nullptr;  // S+1, private to the synchronous assign_batch()
    unsigned long long *h_err = nullptr;     // pinned mirror, same layout
    ErrCtx device_ctx;                       // last *_device batch (for the error sentence)
    std::vector<std::string> barcodes_upper; // as the reference keeps them (:71)
    std::vector<std::string> sample_ids;     // optional (fqtk_matcher_set_sample_ids); default "sample_<i>"
    // complete memo (memo_kernels.hip.h); absent when the candidate set is over budget or L > 20
    void *d_memo = nullptr;
    uint32_t *d_hot = nullptr;               // hot subset (0-mismatch entries) for the LDS table
    uint32_t hot_mask = 0;
    uint32_t *d_filter = nullptr;            // presence filter over all keys of the hash table (memo_hash.hpp), kept in LDS with the hot table
    uint32_t filter_bits = 0;                // log2 of its bits; 0 = none (then the hot table is single-choice)
    uint32_t memo_mask = 0;
    int memo_kw = 1;   // key words per entry (fqtk::memo_key_words)
    // direct-indexed form (L <= 10; memo_hash.hpp): flat result array + LDS cache of its exact-match entries;
    // d_memo then holds only the entries with a no-call
    void *d_direct = nullptr;
    int direct_bytes = 0;                      // 0 = not built, 2 = packed 16-bit entries, 4 = result words
    uint32_t *d_hot2 = nullptr;
    uint32_t hot2_bits = 0, direct_nbits = 0;
    uint32_t direct_ib = 0, direct_bb = 0;     // 16-bit entry layout
    uint64_t hot2_placed = 0, hot2_wanted = 0;
    uint64_t memo_entries = 0;
    uint64_t memo_candidates = 0;
    uint64_t memo_second_slot = 0;             // entries living in their second-choice slot
    // LDS-resident compact memo (lds_memo_kernels.hip.h); present only when every entry is "sample idx
    // with at most one base replaced" and the table fits one CU's LDS
    uint32_t *d_ldsm = nullptr;
    fqtk::LdsMemoParams ldsm{};                // image / masks / salt (m is filled per launch)
    int ldsm_kw = 0;
    int ldsm_form = fqtk::kLdsFormPow2;        // kLdsFormAny / kLdsFormPow2 / kLdsFormMph
    mutable std::vector<const void *> ldsm_big_lds_ok;   // kernels already allowed > 64 KiB LDS on this device
    int memo_kind_wanted = 0;                  // 0 = best available, 1 = force the HBM/L2 table (tests, A/B)
    size_t ldsm_lds_bytes = 0;                 // image + LUT (histogram added at launch)
    std::deque<std::pair<hipStream_t, Worklist>> stream_work;   // worklists of the *_device entry point, by caller stream
    volatile uint32_t *h_seen = nullptr;     // page-locked: set by a memo kernel that met an IUPAC / junk byte in a read
    uint32_t *d_seen = nullptr;              // the same word, device address
    int use_cache = 1;                       // BarcodeMatcher.use_cache (barcode_matching.rs:41-42)
    Slot slots[kNumSlots];
};

namespace {

template <typename T>
int ensure_cap(T *&ptr, size_t &cap, size_t want) {
    if (want <= cap) return FQTK_OK;
    if (ptr) {
        HIP_TRY(hipFree(ptr));
        ptr = nullptr;
        cap = 0;
    }
    size_t ncap = std::max(want, cap + cap / 2);
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&ptr), ncap * sizeof(T)));
    cap = ncap;
    return FQTK_OK;
}

int ensure_slot(fqtk_matcher *m, int slot) {
    if (slot < 0 || slot >= kNumSlots) return fail(FQTK_EINVAL, "slot out of range");
    Slot &s = m->slots[slot];
    if (!s.stream) HIP_TRY(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
    return FQTK_OK;
}

template <int NW, int R, int VEC>
int launch_t(const fqtk::MatchParams &P, int num_cus, hipStream_t stream) {
    const uint64_t tile = (uint64_t)fqtk::kBlock * R;
    const uint64_t ntiles = (P.n + tile - 1) / tile;
    if (ntiles == 0) return FQTK_OK;
    // persistent-ish grid: enough workgroups to fill every CU at full occupancy, grid-stride beyond
    const uint64_t max_blocks = (uint64_t)num_cus * 8;
    const uint32_t grid = (uint32_t)std::min<uint64_t>(ntiles, max_blocks);
    size_t shmem = 256 * sizeof(uint32_t);
    if (P.counts && P.lds_hist) shmem += (size_t)(P.S + 1) * sizeof(uint32_t);
    hipLaunchKernelGGL((fqtk::match_kernel<NW, R, VEC>), dim3(grid), dim3(fqtk::kBlock), shmem, stream, P);
    HIP_TRY(hipGetLastError());
    return FQTK_OK;
}

template <int NW, int R>
int launch_vec(const fqtk::MatchParams &P, int num_cus, hipStream_t stream) {
    const uintptr_t base = reinterpret_cast<uintptr_t>(P.obs);
    const uint32_t nwords = (P.L + 3) / 4;
    if (P.stride % 4 == 0 && base % 4 == 0 && P.stride >= nwords * 4) {
        const uint32_t sw = P.stride / 4;
        if (sw == nwords) {  // the read fills its slot: one vector load per read
            if (sw == 4 && base % 16 == 0) return launch_t<NW, R, 4>(P, num_cus, stream);
            if (sw == 2 && base % 8 == 0) return launch_t<NW, R, 2>(P, num_cus, stream);
            if (sw == 1) return launch_t<NW, R, 1>(P, num_cus, stream);
            if (sw == 3) return launch_t<NW, R, 3>(P, num_cus, stream);
        }
        return launch_t<NW, R, -1>(P, num_cus, stream);
    }
    return launch_t<NW, R, 0>(P, num_cus, stream);
}

template <int KW>
int launch_memo_vec(const fqtk_matcher *m, fqtk::MemoParams Q, hipStream_t stream) {
    const fqtk::MatchParams &P = Q.m;
    const uintptr_t base = reinterpret_cast<uintptr_t>(P.obs);
    const uint32_t nwords = (P.L + 3) / 4;
    int vec = 0;
    if (P.stride % 4 == 0 && base % 4 == 0 && P.stride >= nwords * 4) {
        const uint32_t sw = P.stride / 4;
        vec = -1;
        if (sw == nwords) {
            if (sw == 4 && base % 16 == 0) vec = 4;
            else if (sw == 2 && base % 8 == 0) vec = 2;
            else if (sw == 1) vec = 1;
            else if (sw == 3) vec = 3;
            else if (sw == 5) vec = 5;
            else if (sw == 6 && base % 8 == 0) vec = 6;
            else if (sw == 7) vec = 7;
            else if (sw == 8 && base % 16 == 0) vec = 8;
        }
    }
    // Two reads per lane on the packed vector paths, one on the generic ones and for variable-length batches.
    // Round 4, the direct form with its gathers batched (cfg 5, 12-byte rows; tools/_g4.sh-style A/B, one box, G reads/s):
    // R=2 plain 217-226 (the product shape) / R=2 pipelined 201 (36 bytes of scratch) / one 1024-lane workgroup per CU at
    // 128 VGPRs, pipelined, R=2 207-212, R=4 212-216: the kernel is bound by VALU issue, not by loads in flight.
    // 4- and 8-byte rows run their full tiles software-pipelined one tile deep; wider rows would spill 28-60
    // bytes per lane out of the 64 VGPRs that keep 8 waves per SIMD (hipcc -Rpass-analysis=kernel-resource-usage)
    // and take the plain loop.  Measured on MI355X (tools/ab_table.sh, G reads/s, R=1 pipelined / R=1 plain /
    // R=2 pipelined / R=2 plain / R=4 plain):  cfg 5 (12-byte rows) 181.6 / 173.1 / 165.8 (spills) / 188.1 / 136.7;
    // cfg 3 table pinned (16-byte rows) 148.0 / 141.6 / 139.4 (spills) / 178.0 / 91.2;
    // cfg 2 table pinned (8-byte rows) 236.4 / 203.0 / 248.2 / 234.1 / 182.9.
    const int direct = (KW == 1 && Q.direct) ? m->direct_bytes : 0;
    int R = (vec > 0 && vec < 5) ? 2 : 1;   // rows of 20 bytes and more: two reads per lane would spill
    int abl = 0;
    bool pf = vec == 1 || vec == 2;
#ifdef FQTK_DEV_ABLATE
    if (!P.lens && vec > 0)
        if (const char *rr = std::getenv("FQTK_MEMO_R")) R = std::atoi(rr);
    if (const char *ab = std::getenv("FQTK_MEMO_ABLATE")) abl = std::atoi(ab);
    if (env_flag("FQTK_MEMO_PF")) pf = vec > 0;
    pf = pf && R <= 2 && !P.lens && !env_flag("FQTK_MEMO_NOPF");
    size_t lds_pad = 0;   // occupancy experiments: pad the workgroup's LDS so fewer fit on a CU
    if (const char *lp = std::getenv("FQTK_MEMO_LDS_PAD")) lds_pad = (size_t)std::atol(lp);
#endif
    size_t shmem = 256 * sizeof(uint32_t);
    if (direct) shmem += Q.hot2 ? ((size_t)8 << Q.hot2_bits) : 0;
    else if (Q.hot_mask) shmem += (size_t)(Q.hot_mask + 1) * (KW == 4 ? 32 : (KW >= 2 ? 16 : 8));
    if (!direct && Q.filter_bits) shmem += (size_t)1 << (Q.filter_bits - 3u);
    if (P.counts && P.lds_hist) shmem += (size_t)(P.S + 1) * sizeof(uint32_t);
    {   // the expected-barcode planes for the wave scan of non-canonical reads, while two workgroups still fit a CU
        const size_t with_tab = ((shmem + 15) & ~(size_t)15) + (size_t)P.S * 16;
        if (P.L <= 32 && 2 * (with_tab + 1024) <= fqtk::kLdsMemoMaxBytes) {
            Q.m.scan_tab_lds = 1;
            shmem = with_tab;
        }
    }
#ifdef FQTK_DEV_ABLATE
    shmem += lds_pad;
#endif
    const uint64_t tile = (uint64_t)fqtk::kMemoBlock * R;
    const uint64_t ntiles = (P.n + tile - 1) / tile;
    if (ntiles == 0) return FQTK_OK;
    uint32_t per_cu = 2048 / fqtk::kMemoBlock;
#ifdef FQTK_DEV_ABLATE
    if (const char *pc = std::getenv("FQTK_MEMO_PER_CU")) per_cu = (uint32_t)std::atoi(pc);   // occupancy experiments
#endif
    const uint32_t grid = (uint32_t)std::min<uint64_t>(ntiles, (uint64_t)m->num_cus * per_cu);
    // One launch; the `if constexpr` drops the (load width, key width, form) combinations that cannot occur:
    // the packed vector paths imply the key width (stride 16 B -> 2 key words, 12 B -> 1 or 2, 8/4 B -> 1, 20 / 24 B -> 3,
    // 28 / 32 B -> 4) and the direct form exists for one-word keys only.
#define FQTK_MEMO_LAUNCH_P(V, RR, A, LENS, D, PFV)                                                         \
    do {                                                                                                   \
        if constexpr (((V) <= 0 || ((V) == 3 && KW <= 2) ||                                                \
                       KW == ((V) >= 7 ? 4 : ((V) >= 5 ? 3 : ((V) == 4 ? 2 : 1)))) &&                      \
                      ((D) == 0 || (KW == 1 && (V) <= 3))) {                                               \
            auto kern = fqtk::memo_kernel<V, KW, RR, A, LENS, D, PFV>;                                     \
            const void *fn = reinterpret_cast<const void *>(kern);                                         \
            if (shmem > 64 * 1024 &&                                                                       \
                std::find(m->ldsm_big_lds_ok.begin(), m->ldsm_big_lds_ok.end(), fn) == m->ldsm_big_lds_ok.end()) { \
                HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize,                \
                                            (int)fqtk::kLdsMemoMaxBytes));                                 \
                m->ldsm_big_lds_ok.push_back(fn);                                                          \
            }                                                                                              \
            hipLaunchKernelGGL(kern, dim3(grid), dim3(fqtk::kMemoBlock), shmem, stream, Q);                \
        } else                                                                                             \
            return fail(FQTK_EINVAL, "memo: load width, key width and table form disagree");               \
    } while (0)
    // every load path of one (reads per lane, ablation, lens, form, pipelined) combination
#define FQTK_MEMO_ALL_VEC(RR, A, LENS, D, PFV)                          \
    switch (vec) {                                                      \
        case 8: FQTK_MEMO_LAUNCH_P(8, RR, A, LENS, D, PFV); break;      \
        case 7: FQTK_MEMO_LAUNCH_P(7, RR, A, LENS, D, PFV); break;      \
        case 6: FQTK_MEMO_LAUNCH_P(6, RR, A, LENS, D, PFV); break;      \
        case 5: FQTK_MEMO_LAUNCH_P(5, RR, A, LENS, D, PFV); break;      \
        case 4: FQTK_MEMO_LAUNCH_P(4, RR, A, LENS, D, PFV); break;      \
        case 3: FQTK_MEMO_LAUNCH_P(3, RR, A, LENS, D, PFV); break;      \
        case 2: FQTK_MEMO_LAUNCH_P(2, RR, A, LENS, D, PFV); break;      \
        case 1: FQTK_MEMO_LAUNCH_P(1, RR, A, LENS, D, PFV); break;      \
        case -1: FQTK_MEMO_LAUNCH_P(-1, RR, A, LENS, D, PFV); break;    \
        default: FQTK_MEMO_LAUNCH_P(0, RR, A, LENS, D, PFV); break;     \
    }
    // packed paths only (vec > 0)
#define FQTK_MEMO_PACKED(RR, A, D, PFV)                                 \
    switch (vec) {                                                      \
        case 8: FQTK_MEMO_LAUNCH_P(8, RR, A, false, D, PFV); break;     \
        case 7: FQTK_MEMO_LAUNCH_P(7, RR, A, false, D, PFV); break;     \
        case 6: FQTK_MEMO_LAUNCH_P(6, RR, A, false, D, PFV); break;     \
        case 5: FQTK_MEMO_LAUNCH_P(5, RR, A, false, D, PFV); break;     \
        case 4: FQTK_MEMO_LAUNCH_P(4, RR, A, false, D, PFV); break;     \
        case 3: FQTK_MEMO_LAUNCH_P(3, RR, A, false, D, PFV); break;     \
        case 2: FQTK_MEMO_LAUNCH_P(2, RR, A, false, D, PFV); break;     \
        default: FQTK_MEMO_LAUNCH_P(1, RR, A, false, D, PFV); break;    \
    }
#define FQTK_MEMO_BY_FORM(WHAT, ...)                                    \
    do {                                                                \
        if (direct == 2) { WHAT(__VA_ARGS__, 2); }                      \
        else if (direct == 4) { WHAT(__VA_ARGS__, 4); }                 \
        else { WHAT(__VA_ARGS__, 0); }                                  \
    } while (0)
#ifdef FQTK_DEV_ABLATE
    if (abl > 0 && !P.lens && (vec == 4 || (vec == 3 && direct == 2)) && R == 2) {   // ablations of the product shape
#define FQTK_AB(A)                                                                                   \
        case A:                                                                                      \
            if (vec == 4) FQTK_MEMO_LAUNCH_P(4, 2, A, false, 0, false);                              \
            else FQTK_MEMO_LAUNCH_P(3, 2, A, false, 2, false);                                       \
            break;
        switch (abl) {
            FQTK_AB(1) FQTK_AB(2) FQTK_AB(4) FQTK_AB(8) FQTK_AB(16) FQTK_AB(17) FQTK_AB(32) FQTK_AB(64) FQTK_AB(128) FQTK_AB(256) FQTK_AB(272)
            default: return fail(FQTK_EINVAL, "ablation variant not built");
        }
#undef FQTK_AB
        HIP_TRY(hipGetLastError());
        return FQTK_OK;
    }
    if (!P.lens && vec > 0 && (R != (vec >= 5 ? 1 : 2) || pf != (vec <= 2))) {   // A/B of reads per lane and of the pipeline
#define FQTK_X(RR, D) FQTK_MEMO_PACKED(RR, 0, D, false)
#define FQTK_Y(RR, D) FQTK_MEMO_PACKED(RR, 0, D, true)
        if (R == 4) FQTK_MEMO_BY_FORM(FQTK_X, 4);
        else if (R == 2 && pf) FQTK_MEMO_BY_FORM(FQTK_Y, 2);
        else if (R == 2) FQTK_MEMO_BY_FORM(FQTK_X, 2);
        else if (pf) FQTK_MEMO_BY_FORM(FQTK_Y, 1);
        else FQTK_MEMO_BY_FORM(FQTK_X, 1);
#undef FQTK_X
#undef FQTK_Y
        HIP_TRY(hipGetLastError());
        return FQTK_OK;
    }
#endif
    (void)abl;
    (void)pf;
    if (P.lens) {          // variable-length batch (the LENS instantiations): the shapes of the fixed-length batches --
                           // the length word is loaded with the row -- and one read per lane on the generic load paths
#define FQTK_X(D)                                                       \
        switch (vec) {                                                  \
            case 8: FQTK_MEMO_LAUNCH_P(8, 1, 0, true, D, false); break; \
            case 7: FQTK_MEMO_LAUNCH_P(7, 1, 0, true, D, false); break; \
            case 6: FQTK_MEMO_LAUNCH_P(6, 1, 0, true, D, false); break; \
            case 5: FQTK_MEMO_LAUNCH_P(5, 1, 0, true, D, false); break; \
            case 4: FQTK_MEMO_LAUNCH_P(4, 2, 0, true, D, false); break; \
            case 3: FQTK_MEMO_LAUNCH_P(3, 2, 0, true, D, false); break; \
            case 2: FQTK_MEMO_LAUNCH_P(2, 2, 0, true, D, true); break;  \
            case 1: FQTK_MEMO_LAUNCH_P(1, 2, 0, true, D, true); break;  \
            case -1: FQTK_MEMO_LAUNCH_P(-1, 1, 0, true, D, false); break; \
            default: FQTK_MEMO_LAUNCH_P(0, 1, 0, true, D, false); break;  \
        }
        if (direct == 2) { FQTK_X(2) } else if (direct == 4) { FQTK_X(4) } else { FQTK_X(0) }
#undef FQTK_X
    } else if (vec == 1 || vec == 2) {  // 4- / 8-byte rows: two reads per lane, pipelined
#define FQTK_X(D)                                                       \
        if (vec == 2) FQTK_MEMO_LAUNCH_P(2, 2, 0, false, D, true);      \
        else FQTK_MEMO_LAUNCH_P(1, 2, 0, false, D, true);
        if (direct == 2) { FQTK_X(2) } else if (direct == 4) { FQTK_X(4) } else { FQTK_X(0) }
#undef FQTK_X
    } else if (vec > 0) {               // 12- / 16-byte rows: two reads per lane (wider rows: one), plain loop
#define FQTK_X(D)                                                       \
        if (vec == 3) FQTK_MEMO_LAUNCH_P(3, 2, 0, false, D, false);     \
        else if (vec == 4) FQTK_MEMO_LAUNCH_P(4, 2, 0, false, D, false);\
        else if (vec == 5) FQTK_MEMO_LAUNCH_P(5, 1, 0, false, D, false);\
        else if (vec == 6) FQTK_MEMO_LAUNCH_P(6, 1, 0, false, D, false);   /* (pipelined: 128 vs 133 G reads/s on 384 x 24) */ \
        else if (vec == 7) FQTK_MEMO_LAUNCH_P(7, 1, 0, false, D, false);\
        else FQTK_MEMO_LAUNCH_P(8, 1, 0, false, D, false);
        if (direct == 2) { FQTK_X(2) } else if (direct == 4) { FQTK_X(4) } else { FQTK_X(0) }
#undef FQTK_X
    } else {               // generic load paths: one read per lane
#define FQTK_X(D)                                                       \
        if (vec == -1) FQTK_MEMO_LAUNCH_P(-1, 1, 0, false, D, false);   \
        else FQTK_MEMO_LAUNCH_P(0, 1, 0, false, D, false);
        if (direct == 2) { FQTK_X(2) } else if (direct == 4) { FQTK_X(4) } else { FQTK_X(0) }
#undef FQTK_X
    }
#undef FQTK_MEMO_BY_FORM
#undef FQTK_MEMO_PACKED
#undef FQTK_MEMO_ALL_VEC
#undef FQTK_MEMO_LAUNCH_P
    HIP_TRY(hipGetLastError());
    return FQTK_OK;
}

template <int KW, int FORM>
int launch_lds_memo(const fqtk_matcher *m, fqtk::LdsMemoParams Q, hipStream_t stream, bool second_pass = false) {
    const fqtk::MatchParams &P = Q.m;
    const uintptr_t base = reinterpret_cast<uintptr_t>(P.obs);
    const uint32_t nwords = (P.L + 3) / 4;
    int vec = 0;
    if (P.stride % 4 == 0 && base % 4 == 0 && P.stride >= nwords * 4) {
        const uint32_t sw = P.stride / 4;
        vec = -1;
        if (sw == nwords) {
            if (sw == 4 && base % 16 == 0) vec = 4;
            else if (sw == 2 && base % 8 == 0) vec = 2;
            else if (sw == 1) vec = 1;
            else if (sw == 3) vec = 3;
            else if (sw == 5) vec = 5;
            else if (sw == 6 && base % 8 == 0) vec = 6;
            else if (sw == 7) vec = 7;
            else if (sw == 8 && base % 16 == 0) vec = 8;
        }
    }
    size_t shmem = m->ldsm_lds_bytes;
    Q.hist_shift = 0;
    if (P.counts && P.lds_hist) {
        // copies of the histogram: as many as fit 4 KiB (8 at most) without costing a workgroup per CU
        const size_t one = (size_t)(P.S + 1) * sizeof(uint32_t);
        const bool two_now = 2 * (shmem + one + 1024) <= fqtk::kLdsMemoMaxBytes;
        while (Q.hist_shift < 3 && (one << (Q.hist_shift + 1)) <= 4096 &&
               shmem + (one << (Q.hist_shift + 1)) <= fqtk::kLdsMemoMaxBytes &&
               (!two_now || 2 * (shmem + (one << (Q.hist_shift + 1)) + 1024) <= fqtk::kLdsMemoMaxBytes))
            ++Q.hist_shift;
#ifdef FQTK_DEV_ABLATE
        if (const char *hs = std::getenv("FQTK_LDSM_HIST_SHIFT")) Q.hist_shift = (uint32_t)std::atoi(hs);
#endif
        shmem += one << Q.hist_shift;
    }
    if (shmem > fqtk::kLdsMemoMaxBytes) return fail(FQTK_EINVAL, "lds memo: table does not fit LDS");
    {   // the expected-barcode planes next to it, for the wave scan of non-canonical reads -- when there is room
        // and it does not cost a workgroup per CU
        const size_t with_tab = ((shmem + 15) & ~(size_t)15) + (size_t)P.S * 16;
        const bool two_before = 2 * (shmem + 1024) <= fqtk::kLdsMemoMaxBytes, two_after = 2 * (with_tab + 1024) <= fqtk::kLdsMemoMaxBytes;
        if (P.L <= 32 && with_tab <= fqtk::kLdsMemoMaxBytes && (two_after || !two_before)) {
            Q.m.scan_tab_lds = 1;
            shmem = with_tab;
        }
    }
    // reads per lane: 16 bytes of barcode per lane on the packed paths, one read on the generic ones
    int R = vec >= 3 ? 1 : (vec == 2 ? 2 : (vec == 1 ? 4 : 1));
#ifdef FQTK_DEV_ABLATE
    if (const char *rr = std::getenv("FQTK_MEMO_R")) R = std::atoi(rr);
#endif
    if ((P.lens && vec <= 0) || second_pass) R = 1;
    const uint64_t tile = (uint64_t)fqtk::kLdsBlock * R;
    const uint64_t ntiles = (P.n + tile - 1) / tile;
    if (ntiles == 0) return FQTK_OK;
    // workgroups per CU: LDS-limited, and never more than 2 x 1024 lanes
    const uint32_t per_cu = (uint32_t)std::max<size_t>(1, std::min<size_t>(2048 / fqtk::kLdsBlock, fqtk::kLdsMemoMaxBytes / (shmem + 1024)));
    // (the second pass: every wave the chip holds takes the worklist segments it owns, most of them empty)
    const uint32_t grid = second_pass ? (uint32_t)m->num_cus * per_cu : (uint32_t)std::min<uint64_t>(ntiles, (uint64_t)m->num_cus * per_cu);
#define FQTK_LDSM_LAUNCH(V, RR) FQTK_LDSM_LAUNCH_L(V, RR, false)
#define FQTK_LDSM_LAUNCH_L(V, RR, LENS) FQTK_LDSM_LAUNCH_P(V, RR, LENS, false)
#define FQTK_LDSM_LAUNCH_P(V, RR, LENS, PF) FQTK_LDSM_LAUNCH_I(V, RR, LENS, PF, false)
#define FQTK_LDSM_LAUNCH_I(V, RR, LENS, PF, IDX)                                                           \
    do {                                                                                                   \
        if constexpr ((V) <= 0 || KW == ((V) >= 7 ? 4 : ((V) >= 5 ? 3 : ((V) >= 3 ? 2 : 1)))) {           \
            auto kern = fqtk::lds_memo_kernel<V, KW, RR, FORM, LENS, PF, IDX>;                             \
            /* > 64 KiB of dynamic LDS must be allowed per kernel AND per device: remembered per matcher */ \
            const void *fn = reinterpret_cast<const void *>(kern);                                         \
            if (shmem > 64 * 1024 &&                                                                       \
                std::find(m->ldsm_big_lds_ok.begin(), m->ldsm_big_lds_ok.end(), fn) == m->ldsm_big_lds_ok.end()) { \
                HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize,                \
                                            (int)fqtk::kLdsMemoMaxBytes));                                 \
                m->ldsm_big_lds_ok.push_back(fn);                                                          \
            }                                                                                              \
            hipLaunchKernelGGL(kern, dim3(grid), dim3(fqtk::kLdsBlock), shmem, stream, Q);                \
        } else {                                                                                           \
            return fail(FQTK_EINVAL, "lds memo: load width and key width disagree");                       \
        }                                                                                                  \
    } while (0)
#ifdef FQTK_DEV_ABLATE
    if (R == 8 && (vec == 4 || vec == 2)) {
        if (vec == 4) FQTK_LDSM_LAUNCH(4, 8); else FQTK_LDSM_LAUNCH(2, 8);
        HIP_TRY(hipGetLastError());
        return FQTK_OK;
    }
#endif
    if (second_pass) {   // rows gathered through the worklist: the generic load paths
        if (vec != 0) FQTK_LDSM_LAUNCH_I(-1, 4, false, false, true);
        else FQTK_LDSM_LAUNCH_I(0, 4, false, false, true);
    } else if (P.lens) {   // variable-length batch (the LENS instantiations): packed rows take the pipelined loop at the
                           // fixed-length shapes (the length word is loaded with the row), the generic paths one read per lane
        switch (vec) {
            case 8: FQTK_LDSM_LAUNCH_P(8, 1, true, true); break;
            case 7: FQTK_LDSM_LAUNCH_P(7, 1, true, true); break;
            case 6: FQTK_LDSM_LAUNCH_P(6, 1, true, true); break;
            case 5: FQTK_LDSM_LAUNCH_P(5, 1, true, true); break;
            case 4: FQTK_LDSM_LAUNCH_P(4, 1, true, true); break;
            case 3: FQTK_LDSM_LAUNCH_P(3, 1, true, true); break;
            case 2: FQTK_LDSM_LAUNCH_P(2, 2, true, true); break;
            case 1: FQTK_LDSM_LAUNCH_P(1, 4, true, true); break;
            case -1: FQTK_LDSM_LAUNCH_L(-1, 1, true); break;
            default: FQTK_LDSM_LAUNCH_L(0, 1, true); break;
        }
    } else if (vec > 0) {
        // packed rows: full tiles software-pipelined one tile deep on both streams, 16 bytes of barcode per lane
        // and tile -- one read of 12-20 bytes, two of 8, four of 4.  Measured on MI355X (tools/ab_pf.sh, G reads/s):
        //   cfg 3 (16-byte rows)  R=1 plain 243.5  R=1 pipelined 282.8  R=2 plain 281.3  R=2 pipelined 269.7  R=4 plain 263.8
        //   cfg 2 ( 8-byte rows)  R=1 plain 283.1  R=1 pipelined 377.7  R=2 plain 378.8  R=2 pipelined 472.0  R=4 plain 452.4
#ifdef FQTK_DEV_ABLATE
        const int product_r = vec >= 3 ? 1 : (vec == 2 ? 2 : 4);
        if (env_flag("FQTK_LDSM_NOPF")) {   // A/B: the plain loop at the requested reads per lane
            switch (R * 10 + vec) {
                case 15: FQTK_LDSM_LAUNCH(5, 1); break; case 14: FQTK_LDSM_LAUNCH(4, 1); break; case 13: FQTK_LDSM_LAUNCH(3, 1); break;
                case 12: FQTK_LDSM_LAUNCH(2, 1); break; case 11: FQTK_LDSM_LAUNCH(1, 1); break;
                case 25: FQTK_LDSM_LAUNCH(5, 2); break; case 24: FQTK_LDSM_LAUNCH(4, 2); break; case 23: FQTK_LDSM_LAUNCH(3, 2); break;
                case 22: FQTK_LDSM_LAUNCH(2, 2); break; case 21: FQTK_LDSM_LAUNCH(1, 2); break;
                case 45: FQTK_LDSM_LAUNCH(5, 4); break; case 44: FQTK_LDSM_LAUNCH(4, 4); break; case 43: FQTK_LDSM_LAUNCH(3, 4); break;
                case 42: FQTK_LDSM_LAUNCH(2, 4); break; default: FQTK_LDSM_LAUNCH(1, 4); break;
            }
        } else if (R != product_r && (vec == 4 || vec == 2)) {   // A/B: pipelined at another reads-per-lane
            switch (R * 10 + vec) {
                case 14: FQTK_LDSM_LAUNCH_P(4, 1, false, true); break; case 24: FQTK_LDSM_LAUNCH_P(4, 2, false, true); break;
                case 44: FQTK_LDSM_LAUNCH_P(4, 4, false, true); break; case 12: FQTK_LDSM_LAUNCH_P(2, 1, false, true); break;
                case 22: FQTK_LDSM_LAUNCH_P(2, 2, false, true); break; default: FQTK_LDSM_LAUNCH_P(2, 4, false, true); break;
            }
        } else
#endif
        switch (vec) {
            case 8: FQTK_LDSM_LAUNCH_P(8, 1, false, true); break;
            case 7: FQTK_LDSM_LAUNCH_P(7, 1, false, true); break;
            case 6: FQTK_LDSM_LAUNCH_P(6, 1, false, true); break;
            case 5: FQTK_LDSM_LAUNCH_P(5, 1, false, true); break;
            case 4: FQTK_LDSM_LAUNCH_P(4, 1, false, true); break;
            case 3: FQTK_LDSM_LAUNCH_P(3, 1, false, true); break;
            case 2: FQTK_LDSM_LAUNCH_P(2, 2, false, true); break;
            default: FQTK_LDSM_LAUNCH_P(1, 4, false, true); break;
        }
    } else {
        switch (vec) {
            case -1: FQTK_LDSM_LAUNCH(-1, 1); break;
            default: FQTK_LDSM_LAUNCH(0, 1); break;
        }
    }
#undef FQTK_LDSM_LAUNCH
#undef FQTK_LDSM_LAUNCH_L
#undef FQTK_LDSM_LAUNCH_P
#undef FQTK_LDSM_LAUNCH_I
    HIP_TRY(hipGetLastError());
    return FQTK_OK;
}

// The LDS form's instantiation for this matcher's key width and table form.
int dispatch_lds_memo(const fqtk_matcher *m, const fqtk::LdsMemoParams &Q, hipStream_t stream, bool second_pass) {
    using namespace fqtk;
    switch (m->ldsm_kw * 4 + m->ldsm_form) {
        case 1 * 4 + kLdsFormPow2: return launch_lds_memo<1, kLdsFormPow2>(m, Q, stream, second_pass);
        case 1 * 4 + kLdsFormAny: return launch_lds_memo<1, kLdsFormAny>(m, Q, stream, second_pass);
        case 2 * 4 + kLdsFormPow2: return launch_lds_memo<2, kLdsFormPow2>(m, Q, stream, second_pass);
        case 2 * 4 + kLdsFormAny: return launch_lds_memo<2, kLdsFormAny>(m, Q, stream, second_pass);
        case 3 * 4 + kLdsFormPow2: return launch_lds_memo<3, kLdsFormPow2>(m, Q, stream, second_pass);
        case 3 * 4 + kLdsFormAny: return launch_lds_memo<3, kLdsFormAny>(m, Q, stream, second_pass);
        case 3 * 4 + kLdsFormMph: return launch_lds_memo<3, kLdsFormMph>(m, Q, stream, second_pass);
        case 4 * 4 + kLdsFormPow2: return launch_lds_memo<4, kLdsFormPow2>(m, Q, stream, second_pass);
        case 4 * 4 + kLdsFormAny: return launch_lds_memo<4, kLdsFormAny>(m, Q, stream, second_pass);
        case 4 * 4 + kLdsFormMph: return launch_lds_memo<4, kLdsFormMph>(m, Q, stream, second_pass);
        default: return fail(FQTK_EINVAL, "lds memo: no kernel for this key width and table form");
    }
}

// Second pass of a memo launch, over the reads the memo kernel listed (a byte other than A C G T N . in them).
// LDS form (plain A/C/G/T samples by construction): the memo again, with the reads' ambiguity codes spelled as N.
// Table form: the scan kernel, one lane per listed read.
int launch_second_pass(const fqtk_matcher *m, const fqtk::MatchParams &P0, hipStream_t stream) {
    if (m->d_ldsm && m->memo_kind_wanted != 1) {
        fqtk::LdsMemoParams Q = m->ldsm;
        Q.m = P0;
        Q.m.lens = nullptr;   // only reads of length L are ever listed
        return dispatch_lds_memo(m, Q, stream, true);
    }
    const fqtk::MatchParams &P = P0;
    const uint32_t grid = (uint32_t)m->num_cus * 8;   // a workgroup takes whole segments; empty ones cost it one scalar load
    const size_t shmem = 256 * sizeof(uint32_t) + ((P.counts && P.lds_hist) ? (size_t)(P.S + 1) * sizeof(uint32_t) : 0);
    const uintptr_t base = reinterpret_cast<uintptr_t>(P.obs);
    if (P.stride % 4 == 0 && base % 4 == 0 && P.stride >= ((P.L + 3) / 4) * 4)
        hipLaunchKernelGGL((fqtk::match_kernel<1, 1, -1, true>), dim3(grid), dim3(fqtk::kBlock), shmem, stream, P);
    else
        hipLaunchKernelGGL((fqtk::match_kernel<1, 1, 0, true>), dim3(grid), dim3(fqtk::kBlock), shmem, stream, P);
    HIP_TRY(hipGetLastError());
    return FQTK_OK;
}

int launch_memo(const fqtk_matcher *m, const fqtk::MatchParams &P, hipStream_t stream);

// One batch: the memo kernel (or the scan, or the all-None fill) and, behind a memo kernel, the second pass.
// `wl` = the worklist of the stream the batch runs on (batches on different streams may overlap).
int launch(const fqtk_matcher *m, const fqtk::MatchParams &P0, hipStream_t stream, Worklist &wl) {
    const bool memo = m->use_cache && P0.stride >= P0.L && ((m->d_ldsm && m->memo_kind_wanted != 1) || m->d_memo);
    if (!memo || P0.n > 0xFFFFFFFFull) return launch_memo(m, P0, stream);
    // The list and the second launch only once a read with an IUPAC / junk byte has been met (MatchParams::seen);
    // FQTK_SECOND_PASS=always / never pins it (tests, A/B runs).
    bool listed = m->h_seen && m->h_seen[0] != 0;
    if (const char *sp = std::getenv("FQTK_SECOND_PASS")) {
        if (!std::strcmp(sp, "always")) listed = true;
        else if (!std::strcmp(sp, "never")) listed = false;
    }
    if (!listed) return launch_memo(m, P0, stream);
    // One segment per wave the chip can hold (the memo grids never exceed that; the LDS form may run half as many
    // waves); room for one read in four overall, i.e. at least one in eight of any wave's reads -- what does not
    // fit is scanned in place by its wave.  Grown on demand.
    const uint32_t segs = (uint32_t)m->num_cus * 32u;
    uint64_t want = std::max<uint64_t>(64, (P0.n / 4 + segs - 1) / segs);
    if (const char *cap = std::getenv("FQTK_WORKLIST_CAP"))   // test knob: entries per segment (0: no list, no second pass)
        if (*cap) want = (uint64_t)std::max(0l, std::atol(cap));
    // rows of at most eight dwords travel with their index (MatchParams::work_rw)
    const uint32_t rw = (P0.stride % 4 == 0 && P0.stride <= 32 && reinterpret_cast<uintptr_t>(P0.obs) % 4 == 0) ? P0.stride / 4 : 0;
    if (want > wl.cap || !wl.d_fill || 1u + rw > wl.ew) {
        if (wl.d_list) { HIP_TRY(hipStreamSynchronize(stream)); HIP_TRY(hipFree(wl.d_list)); wl.d_list = nullptr; }
        if (!wl.d_fill) {
            HIP_TRY(hipMalloc(reinterpret_cast<void **>(&wl.d_fill), segs * sizeof(uint32_t)));
            HIP_TRY(hipMemsetAsync(wl.d_fill, 0, segs * sizeof(uint32_t), stream));   // the second pass keeps it zero from here on
        }
        wl.ew = std::max(wl.ew, 1u + rw);
        want = std::max<uint64_t>(want, wl.cap);
        if (want) HIP_TRY(hipMalloc(reinterpret_cast<void **>(&wl.d_list), (size_t)segs * want * wl.ew * sizeof(uint32_t)));
        wl.cap = (uint32_t)want;
    }
    fqtk::MatchParams P = P0;
    if (want && wl.cap) {
        P.work = wl.d_list;
        P.work_n = wl.d_fill;
        P.work_cap = wl.cap;
        P.work_segs = segs;
        P.work_rw = rw;
    }
    const int rc = launch_memo(m, P, stream);
    if (rc != FQTK_OK || !P.work_segs) return rc;
    return launch_second_pass(m, P, stream);
}

// The worklist of a caller's stream (the *_device entry point).  A handful of streams at most in practice;
// past 16 the table is dropped (after the device has drained) and starts over.
int worklist_of_stream(fqtk_matcher *m, hipStream_t stream, Worklist **out) {
    for (auto &e : m->stream_work)
        if (e.first == stream) { *out = &e.second; return FQTK_OK; }
    if (m->stream_work.size() >= 16) {
        HIP_TRY(hipDeviceSynchronize());
        for (auto &e : m->stream_work) e.second.release();
        m->stream_work.clear();
    }
    m->stream_work.emplace_back(stream, Worklist{});
    *out = &m->stream_work.back().second;
    return FQTK_OK;
}

int launch_memo(const fqtk_matcher *m, const fqtk::MatchParams &P, hipStream_t stream) {
    // Rows shorter than a barcode (only legal with obs_len, check_batch_args): every read is shorter
    // than L, so every result is None (barcode_matching.rs:167-169) -- and no kernel below may read L
    // bytes from rows that do not hold them.
    if (P.stride < P.L) {
        const uint32_t grid = (uint32_t)std::min<uint64_t>((P.n + 255) / 256, (uint64_t)m->num_cus * 8);
        hipLaunchKernelGGL(fqtk::none_kernel, dim3(grid), dim3(256), 0, stream, P);
        HIP_TRY(hipGetLastError());
        return FQTK_OK;
    }
    // memo path (table present, caller did not opt out).  With obs_len the memo serves the reads whose
    // length is exactly L; the others follow the length rules inside the same kernel.
    if (m->use_cache && m->d_ldsm && m->memo_kind_wanted != 1) {
        fqtk::LdsMemoParams Q = m->ldsm;
        Q.m = P;
        return dispatch_lds_memo(m, Q, stream, false);
    }
    if (m->use_cache && m->d_memo) {
        fqtk::MemoParams Q;
        Q.m = P;
        Q.slots = m->d_memo;
        Q.mask = m->memo_mask;
        Q.hot = m->d_hot;
        Q.hot_mask = m->hot_mask;
        Q.filter = m->d_filter;
        Q.filter_bits = m->d_filter ? m->filter_bits : 0u;
        Q.direct = m->d_direct;
        Q.hot2 = m->d_hot2;
        Q.hot2_bits = m->hot2_bits;
        Q.d_nbits = m->direct_nbits;
        Q.d_ib = m->direct_ib;
        Q.d_bb = m->direct_bb;
        switch (m->memo_kw) {
            case 1: return launch_memo_vec<1>(m, Q, stream);
            case 2: return launch_memo_vec<2>(m, Q, stream);
            case 3: return launch_memo_vec<3>(m, Q, stream);
            default: return launch_memo_vec<4>(m, Q, stream);
        }
    }
    switch (m->NW) {
        case 1: return launch_vec<1, 4>(P, m->num_cus, stream);
        case 2: return launch_vec<2, 2>(P, m->num_cus, stream);
        case 3: return launch_vec<3, 1>(P, m->num_cus, stream);
        case 4: return launch_vec<4, 1>(P, m->num_cus, stream);
        default: return fail(FQTK_EINVAL, "unsupported barcode length");
    }
}

fqtk::MatchParams make_params(const fqtk_matcher *m, const void *d_obs, uint32_t stride,
                              const void *d_len, uint64_t n, void *d_out, void *d_counts, int err_word) {
    fqtk::MatchParams P;
    P.obs = static_cast<const uint8_t *>(d_obs);
    P.lens = static_cast<const uint32_t *>(d_len);
    P.out = static_cast<uint32_t *>(d_out);
    P.counts = static_cast<unsigned long long *>(d_counts);
    P.table = m->d_table;
    P.lut = m->d_lut;
    P.err = m->d_err + err_word;
    P.n = n;
    P.stride = stride;
    P.S = m->S;
    P.L = m->L;
    P.max_mm = m->max_mm;
    P.delta = m->delta;
    P.nocall_limit = m->max_mm + m->max_ns;
    P.lds_hist = (m->S + 1 <= fqtk::kMaxLdsHist) ? 1u : 0u;
    P.scan_tab_lds = 0;   // the memo launchers turn it on when the table fits their LDS budget
    P.plain_samples = m->plain_samples ? 1u : 0u;
    P.work = nullptr;     // launch() attaches the worklist for the memo kernels
    P.work_n = nullptr;
    P.work_cap = 0;
    P.work_segs = 0;
    P.work_rw = 0;
    P.seen = m->d_seen;
    return P;
}

int check_batch_args(const fqtk_matcher *m, const void *obs, uint32_t stride, const void *lens,
                     uint64_t n, const void *out) {
    if (!m) return fail(FQTK_EINVAL, "matcher is NULL");
    if (n == 0) return FQTK_OK;
    if (!obs || !out) return fail(FQTK_EINVAL, "obs/out is NULL");
    if (!lens && stride < m->L)
        return fail(FQTK_EINVAL, "stride < barcode_len and no obs_len given");
    if (stride == 0) return fail(FQTK_EINVAL, "stride is 0");
    return FQTK_OK;
}

// decode(encode(read)) as the reference's panic message prints it (mod.rs:62-80): every base becomes
// the FIRST letter of "ACGTMRWSYKVHDBN" with its mask; a byte with no mask makes decode() itself panic.
bool decode_like_reference(const uint8_t *read, uint32_t len, std::string *out) {
    static const char kIupacBases[] = "ACGTMRWSYKVHDBN";
    out->clear();
    for (uint32_t i = 0; i < len; ++i) {
        const uint8_t e = enc_byte(read[i]);
        char c = 0;
        for (const char *b = kIupacBases; *b; ++b)
            if (enc_byte((uint8_t)*b) == e) { c = *b; break; }
        if (!c) return false;
        out->push_back(c);
    }
    return true;
}

// The reference's panic sentence for read `e` of a batch (barcode_matching.rs:95-107: assign_internal compares with
// sample 0 first, so that is the sample it names), followed by the read's index in its batch.  Returns FQTK_ELEN.
int word_length_error(fqtk_matcher *m, const ErrCtx &ctx, unsigned long long e) {
    std::string msg;
    std::vector<uint8_t> read;
    uint32_t len = 0;
    bool have = false;
    if (ctx.obs && ctx.lens && e < ctx.n) {
        read.resize(ctx.stride);
        if (ctx.device) {
            have = hipMemcpy(&len, ctx.lens + e, sizeof len, hipMemcpyDeviceToHost) == hipSuccess &&
                   hipMemcpy(read.data(), ctx.obs + e * (uint64_t)ctx.stride, ctx.stride, hipMemcpyDeviceToHost) == hipSuccess;
            if (!have) (void)hipGetLastError();
        } else {
            len = ctx.lens[e];
            std::memcpy(read.data(), ctx.obs + e * (uint64_t)ctx.stride, ctx.stride);
            have = true;
        }
        len = std::min(len, ctx.stride);
    }
    const std::string id = m->sample_ids.empty() ? std::string("sample_0") : m->sample_ids[0];
    std::string decoded;
    if (have && !decode_like_reference(read.data(), len, &decoded)) {
        msg = "Invalid bit mask for base: 0";   // decode() panics first on a byte with no IUPAC mask (mod.rs:80)
    } else if (have) {
        msg = "Read barcode (" + decoded + ") length (" + std::to_string(len) + ") differs from expected barcode (" +
              m->barcodes_upper[0] + ") length (" + std::to_string(m->L) + ") for sample " + id;
    } else {
        msg = "Read barcode length differs from expected barcode (" + m->barcodes_upper[0] + ") length (" +
              std::to_string(m->L) + ") for sample " + id;
    }
    msg += " [read index " + std::to_string(e) + " of its batch]";
    return fail(FQTK_ELEN, msg);
}

// Reads + clears one latched error word.  Stream must be idle for the value to be final.  On an error
// the message is the reference's own sentence (barcode_matching.rs:95-107: assign_internal compares
// with sample 0 first, so that is the sample it names), followed by the read's index in its batch.
int collect_error(fqtk_matcher *m, hipStream_t stream, int err_word, const ErrCtx &ctx, uint64_t *read_index) {
    unsigned long long *d_err = m->d_err + err_word, *h_err = m->h_err + err_word;
    HIP_TRY(hipMemcpyAsync(h_err, d_err, sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    const unsigned long long e = *h_err;
    if (e == ~0ull) return FQTK_OK;
    HIP_TRY(hipMemsetAsync(d_err, 0xFF, sizeof(unsigned long long), stream));
    HIP_TRY(hipStreamSynchronize(stream));
    if (read_index) *read_index = (uint64_t)e;
    return word_length_error(m, ctx, e);
}

// ---- complete-memo construction (see memo_kernels.hip.h) -----------------------------------------
// Strings enumerated and scanned at create time, at most.  They are streamed through the scan kernel in chunks, so the
// bound is on the TABLE they leave behind (16-byte slots at load <= 0.5; a quarter of the strings are Some entries in
// the worst case seen) and on create time (~1 s of host work per 10 M strings), not on host memory.
constexpr uint64_t kMemoCandidateBudget = 24000000;
constexpr uint64_t kMemoScanChunk = 4000000;
const uint8_t kCanonNib[5] = {1, 2, 4, 8, 15};
const char kCanonChr[5] = {'A', 'C', 'G', 'T', 'N'};

// #canonical strings within <= max_mm mismatches of one expected barcode (saturating).
uint64_t count_candidates(const uint8_t *e, uint32_t L, uint32_t max_mm, uint64_t cap) {
    std::vector<double> ways(max_mm + 1, 0.0);   // ways[k] = #strings with exactly k mismatches so far
    ways[0] = 1.0;
    for (uint32_t i = 0; i < L; ++i) {
        uint32_t a = 0;
        for (int c = 0; c < 5; ++c) a += ((kCanonNib[c] & ~e[i] & 0xF) == 0) ? 1u : 0u;
        const uint32_t x = 5 - a;
        for (int k = (int)max_mm; k >= 0; --k)
            ways[k] = ways[k] * a + (k > 0 ? ways[k - 1] * x : 0.0);
    }
    double tot = 0;
    for (double w : ways) tot += w;
    return tot > (double)cap ? cap + 1 : (uint64_t)tot;
}

// `exact` gets one byte per string: 1 = it is a spelling of this very barcode (no mismatch used)
void enumerate_candidates(const uint8_t *e, uint32_t L, uint32_t budget, uint32_t pos, char *cur,
                          std::vector<char> &out, std::vector<uint8_t> &exact, uint32_t used = 0) {
    if (pos == L) {
        out.insert(out.end(), cur, cur + L);
        exact.push_back(used == 0 ? 1 : 0);
        return;
    }
    for (int c = 0; c < 5; ++c) {
        const bool mis = (kCanonNib[c] & ~e[pos] & 0xF) != 0;
        if (mis && budget == 0) continue;
        cur[pos] = kCanonChr[c];
        enumerate_candidates(e, L, budget - (mis ? 1u : 0u), pos + 1, cur, out, exact, used + (mis ? 1u : 0u));
    }
}

// 4 bits per base at memo_nibble_shift(k) of word k >> 3: the key the kernels' encode_nibbles builds, never folded
void memo_key_of(const char *q, uint32_t L, uint32_t (&k)[4]) {
    k[0] = k[1] = k[2] = k[3] = 0;
    for (uint32_t i = 0; i < L; ++i) k[i >> 3] |= fqtk::memo_code_of(q[i]) << fqtk::memo_nibble_shift(i);
}
// the one-word key of a barcode of <= 10 bases: bases 8-9 ride in lo's spare bits (kFoldMul)
uint32_t memo_folded(const uint32_t (&k)[4]) {
    const uint32_t x = (k[1] & 7u) | (((k[1] >> 8) & 7u) << 8);   // c[2] as the kernel sees it: codes in bytes 0, 1
    return k[0] | ((x * fqtk::kFoldMul) & fqtk::kFoldMask);
}

// ---- LDS-resident compact memo (lds_memo_kernels.hip.h; planned on the host by lds_memo_plan.hpp) ----
int build_lds_memo(fqtk_matcher *m, const std::vector<fqtk::LdsEntry> &ents,
                   const std::vector<std::vector<uint8_t>> &enc) {
    uint32_t salt_offset = 0;
#ifdef FQTK_DEV_ABLATE
    if (const char *so = std::getenv("FQTK_LDSM_SALT")) salt_offset = (uint32_t)std::atoi(so);
    int trials = 8;
    if (const char *st = std::getenv("FQTK_LDSM_TRIALS")) trials = std::atoi(st);
    fqtk::LdsMemoPlan plan = fqtk::plan_lds_memo(m->S, m->L, ents, enc, salt_offset, trials);
    if (std::getenv("FQTK_LDSM_TRIALS")) std::fprintf(stderr, "ldsm salt_offset %u score %llu slots %u\n", salt_offset, (unsigned long long)plan.multi_score, plan.n_slots);
#else
    fqtk::LdsMemoPlan plan = fqtk::plan_lds_memo(m->S, m->L, ents, enc, salt_offset);
#endif
    // four-byte cuckoo slots have no room for it (12+12 dual indexes of 384 samples): three-byte entries behind a perfect hash.
    // FQTK_LDS_MPH=0: never (the HBM/L2 table serves such shapes, as before round 6); =2: wherever it can be planned (A/B)
    {
        const char *e = std::getenv("FQTK_LDS_MPH");
        const int mode = e ? std::atoi(e) : 1;
        if ((mode == 1 && !plan.ok) || mode == 2) {
            fqtk::LdsMemoPlan mph = fqtk::plan_lds_memo_mph(m->S, m->L, ents, enc, salt_offset);
            if (mph.ok) plan = std::move(mph);
        }
    }
    if (!plan.ok) return FQTK_OK;   // not of that shape / does not fit: the HBM/L2 table serves
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&m->d_ldsm), plan.image.size() * 4));
    HIP_TRY(hipMemcpy(m->d_ldsm, plan.image.data(), plan.image.size() * 4, hipMemcpyHostToDevice));
    m->ldsm.image = m->d_ldsm;
    m->ldsm.slot_mask_b = plan.mph ? plan.bucket_mask : plan.slot_mask_b;
    m->ldsm.t8_off_b = plan.t8_off_b;
    m->ldsm.aux_off_b = plan.aux_off_b;
    m->ldsm.n_slots = plan.n_slots;
    m->ldsm_form = plan.mph ? fqtk::kLdsFormMph : (plan.pow2 ? fqtk::kLdsFormPow2 : fqtk::kLdsFormAny);
    m->ldsm.idx_bits = plan.idx_bits;
    m->ldsm.image_words = (uint32_t)plan.image.size();
    m->ldsm.skey_off_b = plan.skey_off_b;
    m->ldsm.salt = plan.salt;
    m->ldsm_kw = plan.kw;
    m->ldsm_lds_bytes = plan.image.size() * 4 + 1024;
    return FQTK_OK;
}

// One Some entry of the memo: the unfolded key of its string, the result word, whether the string carries a no-call.
struct Entry { uint32_t k[4]; uint32_t val; bool has_n; };
inline bool key_less(const uint32_t (&a)[4], const uint32_t (&b)[4]) {
    for (int w = 3; w >= 0; --w) if (a[w] != b[w]) return a[w] < b[w];
    return false;
}
inline bool key_equal(const uint32_t (&a)[4], const uint32_t (&b)[4]) { return a[0] == b[0] && a[1] == b[1] && a[2] == b[2] && a[3] == b[3]; }

// Direct-indexed form of the memo for barcodes of <= 10 bases: planned on the host (direct_memo_plan.hpp),
// uploaded here.  exact_none: spellings of a sample barcode that resolve to None (another sample admits them too) --
// as popular as any exact match, and the LDS cache may as well say so (the array's answer for them is the absent entry).
int build_direct(fqtk_matcher *m, const std::vector<Entry> &ents, const std::vector<fqtk::DirectEntry> &exact_none) {
#ifdef FQTK_DEV_ABLATE
    if (env_flag("FQTK_NO_DIRECT")) return FQTK_OK;
#endif
    std::vector<fqtk::DirectEntry> dir;
    for (const Entry &e : ents)
        if (!e.has_n) dir.push_back(fqtk::DirectEntry{e.k[0], e.k[1], e.val});
    const fqtk::DirectMemoPlan plan = fqtk::plan_direct_memo(m->S, m->L, dir, exact_none);
    if (!plan.entry_bytes) return FQTK_OK;
    for (const fqtk::DirectEntry &d : dir)   // self-check: the kernel's lookup returns every stored entry
        if (fqtk::direct_memo_lookup(plan, d.lo, d.hi) != d.val) return fail(FQTK_EINVAL, "direct memo: self-check failed");
    for (const fqtk::DirectEntry &d : exact_none)
        if (fqtk::direct_memo_lookup(plan, d.lo, d.hi) != fqtk::kMemoEmpty) return fail(FQTK_EINVAL, "direct memo: self-check failed (None spelling)");
    const void *src = plan.entry_bytes == 2 ? (const void *)plan.table16.data() : (const void *)plan.table32.data();
    const size_t bytes = plan.entry_bytes == 2 ? plan.table16.size() * 2 : plan.table32.size() * 4;
    HIP_TRY(hipMalloc(&m->d_direct, bytes));
    HIP_TRY(hipMemcpy(m->d_direct, src, bytes, hipMemcpyHostToDevice));
    if (!plan.hot2.empty()) {
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&m->d_hot2), plan.hot2.size() * 4));
        HIP_TRY(hipMemcpy(m->d_hot2, plan.hot2.data(), plan.hot2.size() * 4, hipMemcpyHostToDevice));
    }
    m->direct_bytes = plan.entry_bytes;
    m->direct_ib = plan.ib;
    m->direct_bb = plan.bb;
    m->hot2_bits = plan.hot2_bits;
    m->direct_nbits = plan.nbits;
    m->hot2_placed = plan.hot2_placed;
    m->hot2_wanted = plan.hot2_wanted;
    return FQTK_OK;
}

// Slot image of the hash-table form by key width (MemoParams::slots): words per slot, and one slot written.
inline size_t slot_words(int kw) { return kw == 1 ? 2 : (kw == 4 ? 8 : 4); }
inline void write_slot(uint32_t *w, int kw, const uint32_t *key /* folded for kw 1; NULL = empty */, uint32_t val, bool spill) {
    const uint32_t sp = spill ? 1u : 0u;
    switch (kw) {
        case 1: w[0] = (key ? key[0] : 0x7FFFFFFFu) | (sp << 31); w[1] = key ? val : fqtk::kMemoEmpty; break;
        case 2: w[0] = key ? key[0] : ~0u; w[1] = key ? key[1] : ~0u; w[2] = key ? val : fqtk::kMemoEmpty; w[3] = sp; break;
        case 3: w[0] = key ? key[0] : ~0u; w[1] = key ? key[1] : ~0u; w[2] = key ? key[2] : ~0u;
                w[3] = (key ? val : 0x7FFFFFFFu) | (sp << 31); break;   // (a result word never has bit 31: next <= 32)
        default: for (int j = 0; j < 4; ++j) w[j] = key ? key[j] : ~0u;
                 w[4] = key ? val : fqtk::kMemoEmpty; w[5] = sp; w[6] = w[7] = 0; break;
    }
}

int build_memo(fqtk_matcher *m, const std::vector<std::vector<uint8_t>> &enc) {
    if (m->L > fqtk::kMemoMaxLen) return FQTK_OK;
    uint64_t total = 0;
    for (uint32_t s = 0; s < m->S && total <= kMemoCandidateBudget; ++s)
        total += count_candidates(enc[s].data(), m->L, m->max_mm, kMemoCandidateBudget);
    if (total > kMemoCandidateBudget) return FQTK_OK;   // over budget: exhaustive scan only
    m->memo_kw = fqtk::memo_key_words(m->L);
    const int kw = m->memo_kw;
    const bool direct_len = m->L <= fqtk::kDirectMaxLen;
    // ---- enumerate every canonical string within max_mismatches of a sample, scan them on the device in chunks
    //      (d_memo is still NULL: the scan kernel), keep the Some results (and, for the direct form, the exact
    //      spellings that are None)
    std::vector<Entry> ents;
    std::vector<fqtk::DirectEntry> exact_none;
    {
        std::vector<char> cand, cur(m->L);
        std::vector<uint8_t> exact;
        std::vector<fqtk_match_t> res;
        uint64_t nc_total = 0;
        auto flush = [&]() -> int {
            const uint64_t nc = cand.size() / m->L;
            if (!nc) return FQTK_OK;
            res.resize(nc);
            const int rc = fqtk_matcher_assign_batch(m, reinterpret_cast<const uint8_t *>(cand.data()), m->L, nullptr, nc, res.data(), nullptr);
            if (rc != FQTK_OK) return rc;
            for (uint64_t i = 0; i < nc; ++i) {
                const char *q = cand.data() + i * m->L;
                if (res[i].idx != FQTK_NO_MATCH) {
                    Entry e;
                    memo_key_of(q, m->L, e.k);
                    std::memcpy(&e.val, &res[i], 4);
                    e.has_n = std::memchr(q, 'N', m->L) != nullptr;
                    ents.push_back(e);
                } else if (direct_len && exact[i] && !std::memchr(q, 'N', m->L)) {
                    uint32_t k[4];
                    memo_key_of(q, m->L, k);
                    exact_none.push_back(fqtk::DirectEntry{k[0], k[1], fqtk::kMemoEmpty});
                }
            }
            nc_total += nc;
            cand.clear();
            exact.clear();
            return FQTK_OK;
        };
        cand.reserve((size_t)std::min<uint64_t>(total, kMemoScanChunk + 65536) * m->L);
        for (uint32_t s = 0; s < m->S; ++s) {
            enumerate_candidates(enc[s].data(), m->L, m->max_mm, 0, cur.data(), cand, exact);
            if (cand.size() / m->L >= kMemoScanChunk) { const int rc = flush(); if (rc != FQTK_OK) return rc; }
        }
        { const int rc = flush(); if (rc != FQTK_OK) return rc; }
        m->memo_candidates = nc_total;
    }
    // distinct Some entries (a string can neighbour several samples)
    std::sort(ents.begin(), ents.end(), [](const Entry &a, const Entry &b) { return key_less(a.k, b.k); });
    ents.erase(std::unique(ents.begin(), ents.end(), [](const Entry &a, const Entry &b) { return key_equal(a.k, b.k); }), ents.end());
    std::sort(exact_none.begin(), exact_none.end(), [](const fqtk::DirectEntry &a, const fqtk::DirectEntry &b) { return a.hi != b.hi ? a.hi < b.hi : a.lo < b.lo; });
    exact_none.erase(std::unique(exact_none.begin(), exact_none.end(), [](const fqtk::DirectEntry &a, const fqtk::DirectEntry &b) { return a.lo == b.lo && a.hi == b.hi; }), exact_none.end());
    const uint64_t entries = ents.size();
    // ---- LDS-resident form (plain A/C/G/T samples, <= 1 mismatch, fits one CU's LDS): sees every entry
    {
        std::vector<fqtk::LdsEntry> le(ents.size());
        for (size_t i = 0; i < ents.size(); ++i) { std::memcpy(le[i].k, ents[i].k, sizeof le[i].k); le[i].val = ents[i].val; }
        const int rc = build_lds_memo(m, le, enc);
        if (rc != FQTK_OK) return rc;
    }
    // ---- direct-indexed form (L <= 10): entries without a no-call go to a flat array indexed by the read
    //      itself; the entries WITH one go to buckets of two slots (direct_memo_plan.hpp) ---------------
    // Measured on MI355X (tools/ab_direct.sh, G reads/s, direct / hash table only): it pays where the hash
    // table is big or most non-exact reads are unmatched -- cfg 5 (1536 IUPAC x 10) 177.5 / 149.3, 1536 x 10 plain
    // 215.9 / 207.7, 1536 x 8 227.2 / 203.2 -- and costs a little where the old LDS hot table already held every
    // exact match of a small table -- cfg 2 pinned (96 x 8) 243.3 / 276.2, 96 x 8 with two mismatches 257.5 / 276.3.
    bool want_direct = direct_len && (m->L >= 9 || m->S >= 512);
#ifdef FQTK_DEV_ABLATE
    if (env_flag("FQTK_FORCE_DIRECT")) want_direct = direct_len;
#endif
    if (want_direct) {
        int rc = build_direct(m, ents, exact_none);
        if (rc != FQTK_OK) return rc;
        if (m->d_direct) {
            std::vector<fqtk::NKey> with_n;
            for (const Entry &e : ents)
                if (e.has_n) with_n.push_back(fqtk::NKey{memo_folded(e.k), e.val});
            const fqtk::NBucketPlan nb = fqtk::plan_nbuckets(with_n);
            bool good = nb.ok;
            for (size_t i = 0; i < with_n.size() && good; ++i) good = fqtk::nbucket_lookup(nb, with_n[i].lo) == with_n[i].val;
            if (good) {
                void *d = nullptr;
                HIP_TRY(hipMalloc(&d, nb.words.size() * sizeof(uint32_t)));
                HIP_TRY(hipMemcpy(d, nb.words.data(), nb.words.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
                HIP_TRY(hipDeviceSynchronize());
                m->memo_second_slot = nb.second;
                m->memo_mask = nb.mask;
                m->memo_entries = entries;
                m->d_memo = d;   // last: enables the memo path
                return FQTK_OK;
            }
            // (placement kept failing: drop the direct form, the hash table below serves every read)
            (void)hipFree(m->d_direct); m->d_direct = nullptr;
            if (m->d_hot2) { (void)hipFree(m->d_hot2); m->d_hot2 = nullptr; }
            m->direct_bytes = 0;
        }
    }
    // ---- hash-table form: two-choice (cuckoo) placement with random-walk eviction; grow on the (unlikely) failure.
    //      Keys as the kernel hashes them: folded to one word for L <= 10, else the unfolded words.
    const size_t wps = slot_words(kw);
    std::vector<std::array<uint32_t, 4>> keys(ents.size());
    for (size_t i = 0; i < ents.size(); ++i) {
        if (kw == 1) keys[i] = {memo_folded(ents[i].k), 0u, 0u, 0u};
        else keys[i] = {ents[i].k[0], ents[i].k[1], ents[i].k[2], ents[i].k[3]};
    }
    auto slots_of = [&](size_t i, uint32_t mask, uint32_t &a1, uint32_t &a2) { fqtk::memo_hash2(keys[i][0], keys[i][1], keys[i][2], keys[i][3], mask, a1, a2); };
    uint64_t nslots = 1024;
    uint64_t slot_factor = ents.size() > (1u << 20) ? 2 : 4;   // small tables sparse (fewer second probes), big ones at load <= 0.5
#ifdef FQTK_DEV_ABLATE
    if (const char *sf = std::getenv("FQTK_MEMO_SLOT_FACTOR")) slot_factor = (uint64_t)std::atoi(sf);
#endif
    while (nslots < ents.size() * slot_factor) nslots <<= 1;
    std::vector<uint32_t> slots;
    uint32_t mask = 0;
    for (int attempt = 0;; ++attempt) {
        if (attempt == 6 || nslots * wps * 4 >= (1ull << 32)) return FQTK_OK;   // placement keeps failing / past the kernel's
                                                                              // 32-bit slot offsets: no memo, the scan kernel
        mask = (uint32_t)(nslots - 1);
        std::vector<int64_t> owner(nslots, -1);
        bool ok = true;
        uint64_t rng = 0x9E3779B97F4A7C15ull;
        for (size_t i = 0; i < ents.size() && ok; ++i) {
            int64_t cur = (int64_t)i;
            uint32_t a1, a2;
            slots_of((size_t)cur, mask, a1, a2);
            uint32_t pos = owner[a1] < 0 ? a1 : a2;
            for (int kick = 0;; ++kick) {
                if (owner[pos] < 0) { owner[pos] = cur; break; }
                if (kick == 1000) { ok = false; break; }
                std::swap(cur, owner[pos]);   // evict the occupant, re-home it
                slots_of((size_t)cur, mask, a1, a2);
                rng = rng * 6364136223846793005ull + 1442695040888963407ull;
                pos = (a1 == pos) ? a2 : ((a2 == pos) ? a1 : ((rng >> 33) & 1 ? a1 : a2));
                if (a1 == a2 && owner[pos] >= 0 && kick > 8) { ok = false; break; }
            }
        }
        if (!ok) { nslots <<= 1; continue; }
        // pull keys back into their FIRST slot where it has become free, then mark the slots whose
        // would-be first-slot owner still lives in its second slot (the kernel's SPILL bit)
        for (bool moved = true; moved;) {
            moved = false;
            for (uint64_t p = 0; p < nslots; ++p) {
                if (owner[p] < 0) continue;
                uint32_t a1, a2;
                slots_of((size_t)owner[p], mask, a1, a2);
                if (a1 != p && owner[a1] < 0) { owner[a1] = owner[p]; owner[p] = -1; moved = true; }
            }
        }
        std::vector<uint8_t> spill(nslots, 0);
        uint64_t n_second = 0;
        for (uint64_t p = 0; p < nslots; ++p) {
            if (owner[p] < 0) continue;
            uint32_t a1, a2;
            slots_of((size_t)owner[p], mask, a1, a2);
            if (a1 != p) { spill[a1] = 1; ++n_second; }
        }
        m->memo_second_slot = n_second;
        slots.assign(nslots * wps, 0u);
        for (uint64_t p = 0; p < nslots; ++p)
            write_slot(&slots[p * wps], kw, owner[p] < 0 ? nullptr : keys[(size_t)owner[p]].data(),
                       owner[p] < 0 ? 0u : ents[(size_t)owner[p]].val, spill[p] != 0);
        break;
    }
    // hot table for LDS: 0-mismatch entries (it is only a cache: an entry that finds no place is served by the global table).
    // Round 5: where the exact-match entries are few (any plain table: S of them), the hot table is TWO-choice -- slot h1 or h2 of the key's
    // two full hashes, filled to 40 % at most, so that every entry finds a place in a table a quarter of the single-choice one's size -- and
    // the LDS it leaves goes to a PRESENCE FILTER over all keys of the memo (memo_hash.hpp): a read that is in no slot of the table never
    // gathers.  Tables whose exact-match entries fill the LDS budget anyway (IUPAC expansions) keep the single-choice table and no filter.
    {
        const uint32_t slot_bytes = (uint32_t)wps * 4;
        uint64_t n_hot = 0;
        for (const Entry &e : ents) n_hot += ((e.val >> 16) & 0xFFu) == 0;
        std::vector<size_t> order;
        for (size_t i = 0; i < ents.size(); ++i) if (((ents[i].val >> 16) & 0xFFu) == 0) order.push_back(i);
        std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) {
            return (ents[a].val & 0xFFFFu) < (ents[b].val & 0xFFFFu);   // low sample index first
        });
        auto full_hashes = [&](size_t i, uint32_t &h1, uint32_t &h2) {
            h1 = fqtk::memo_hash1_full(keys[i][0], keys[i][1], keys[i][2], keys[i][3]);
            h2 = fqtk::memo_hash2_full(keys[i][0], keys[i][1], keys[i][2], keys[i][3]);
        };
        // two-choice + filter?
        uint32_t two_slots = 64;
        while ((uint64_t)two_slots * 2 < n_hot * 5) two_slots <<= 1;            // load <= 0.4
        uint32_t fbits = 0;
        // (keys of one word -- barcodes of up to ten bases -- do without: their tables are a few thousand 8-byte slots that the L1s hold, a
        //  gather there costs less than the second hash and the three LDS reads; measured on one box, tools/ab_filter.sh: cfg 4 pinned 324 -> 290
        //  G reads/s, cfg 2 281 -> 286; with two key words and more: cfg 3 pinned 196 -> 219, 12+12 129 -> 144, 16+16 92 -> 107)
        if (n_hot && kw >= 2 && (uint64_t)two_slots * slot_bytes <= fqtk::kHotBytes / 2 && !env_flag("FQTK_MEMO_NO_FILTER")) {
            uint64_t fbytes = 4096;
            while (fbytes * 2 + (uint64_t)two_slots * slot_bytes <= fqtk::kHotBytes) fbytes <<= 1;
            if (fbytes * 8 >= ents.size() * 4)                                             // four bits per key at least, or it filters nothing
                for (uint64_t b = fbytes * 8; b > 1; b >>= 1) ++fbits;
        }
        bool two_choice = false;
        if (fbits) {
            const uint32_t hmask = two_slots - 1;
            std::vector<int64_t> own(two_slots, -1);
            bool ok = true;
            uint64_t rng = 0x2545F4914F6CDD1Dull;
            for (size_t i : order) {
                int64_t cur = (int64_t)i;
                uint32_t h1, h2;
                full_hashes((size_t)cur, h1, h2);
                uint32_t pos = own[h1 & hmask] < 0 ? (h1 & hmask) : (h2 & hmask);
                for (int kick = 0;; ++kick) {
                    if (own[pos] < 0) { own[pos] = cur; break; }
                    if (kick == 500) { ok = false; break; }
                    std::swap(cur, own[pos]);
                    full_hashes((size_t)cur, h1, h2);
                    rng = rng * 6364136223846793005ull + 1442695040888963407ull;
                    const uint32_t a1 = h1 & hmask, a2 = h2 & hmask;
                    pos = (a1 == pos) ? a2 : ((a2 == pos) ? a1 : ((rng >> 33) & 1 ? a1 : a2));
                    if (a1 == a2 && kick > 8) { ok = false; break; }
                }
                if (!ok) break;
            }
            if (ok) {
                std::vector<uint32_t> hot((size_t)two_slots * wps, 0u);
                for (uint32_t p = 0; p < two_slots; ++p)
                    write_slot(&hot[(size_t)p * wps], kw, own[p] < 0 ? nullptr : keys[(size_t)own[p]].data(), own[p] < 0 ? 0u : ents[(size_t)own[p]].val, false);
                std::vector<uint32_t> filt((size_t)1 << (fbits - 5), 0u);
                for (size_t i = 0; i < ents.size(); ++i) {
                    uint32_t h1, h2;
                    full_hashes(i, h1, h2);
                    const uint32_t p1 = fqtk::memo_filter_pos1(h1, fbits), p2 = fqtk::memo_filter_pos2(h2, fbits);
                    filt[p1 >> 5] |= 1u << (p1 & 31u);
                    filt[p2 >> 5] |= 1u << (p2 & 31u);
                }
                HIP_TRY(hipMalloc(reinterpret_cast<void **>(&m->d_hot), hot.size() * sizeof(uint32_t)));
                HIP_TRY(hipMemcpy(m->d_hot, hot.data(), hot.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
                HIP_TRY(hipMalloc(reinterpret_cast<void **>(&m->d_filter), filt.size() * sizeof(uint32_t)));
                HIP_TRY(hipMemcpy(m->d_filter, filt.data(), filt.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
                m->hot_mask = hmask;
                m->filter_bits = fbits;
                two_choice = true;
            }
        }
        if (n_hot && !two_choice) {   // single choice, in the low bits of the entry's FIRST slot's number
            uint32_t hot_slots = fqtk::kHotBytes / slot_bytes;
            while (hot_slots > 64 && hot_slots / 2 >= n_hot * 8) hot_slots >>= 1;   // an eighth full at most while LDS allows
            const uint32_t hmask = hot_slots - 1;
            std::vector<uint32_t> hot((size_t)hot_slots * wps, 0u);
            std::vector<uint8_t> used(hot_slots, 0);
            for (uint32_t p = 0; p < hot_slots; ++p) write_slot(&hot[(size_t)p * wps], kw, nullptr, 0u, false);
            for (size_t i : order) {
                uint32_t a1, a2;
                slots_of(i, mask, a1, a2);
                const uint32_t a = a1 & hmask;
                if (used[a]) continue;
                used[a] = 1;
                write_slot(&hot[(size_t)a * wps], kw, keys[i].data(), ents[i].val, false);
            }
            HIP_TRY(hipMalloc(reinterpret_cast<void **>(&m->d_hot), hot.size() * sizeof(uint32_t)));
            HIP_TRY(hipMemcpy(m->d_hot, hot.data(), hot.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
            m->hot_mask = hmask;
        }
    }
    void *d = nullptr;
    HIP_TRY(hipMalloc(&d, slots.size() * sizeof(uint32_t)));
    HIP_TRY(hipMemcpy(d, slots.data(), slots.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIP_TRY(hipDeviceSynchronize());   // pageable H2D copies may still be in flight; slot streams are non-blocking
    m->memo_mask = mask;
    m->memo_entries = entries;
    m->d_memo = d;   // last: enables the memo path
    return FQTK_OK;
}

}  // namespace

extern "C" {

const char *fqtk_last_error(void) { return g_last_error.c_str(); }

int fqtk_abi_version(void) { return 5; }

int fqtk_device_count(int *n_devices) {
    if (!n_devices) return fail(FQTK_EINVAL, "n_devices is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        n = 0;
    }
    *n_devices = n;
    return FQTK_OK;
}

int fqtk_matcher_create(const char *const *barcodes, uint32_t n_samples, uint32_t barcode_len,
                        uint8_t max_mismatches, uint8_t min_mismatch_delta, int device,
                        fqtk_matcher **out) {
    if (!out) return fail(FQTK_EINVAL, "out is NULL");
    *out = nullptr;
    // barcode_matching.rs:61-65
    if (n_samples == 0 || !barcodes) return fail(FQTK_EINVAL, "Must provide at least one sample");
    if (n_samples > FQTK_MAX_SAMPLES) return fail(FQTK_EINVAL, "too many samples (max 65534)");
    for (uint32_t s = 0; s < n_samples; ++s)
        if (!barcodes[s] || barcodes[s][0] == '\0')
            return fail(FQTK_EINVAL, "Sample barcode cannot be empty string");
    if (barcode_len == 0) return fail(FQTK_EINVAL, "Sample barcode cannot be empty string");
    if (barcode_len > FQTK_MAX_BARCODE_LEN)
        return fail(FQTK_EINVAL, "barcode_len > 128 is not supported");
    for (uint32_t s = 0; s < n_samples; ++s)
        if (std::strlen(barcodes[s]) != barcode_len)
            return fail(FQTK_EINVAL, "All barcodes must have the same length");  // samples.rs:117-122

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        (void)hipGetLastError();
        return fail(FQTK_ENODEV, "no HIP device available (this library has no CPU fallback)");
    }
    if (device < 0 || device >= ndev) return fail(FQTK_ENODEV, "device index out of range");
    HIP_TRY(hipSetDevice(device));

    fqtk_matcher *m = new (std::nothrow) fqtk_matcher();
    if (!m) return fail(FQTK_ENOMEM, "out of host memory");
    m->device = device;
    m->S = n_samples;
    m->L = barcode_len;
    m->NW = (barcode_len + 31) / 32;
    m->max_mm = max_mismatches;
    m->delta = min_mismatch_delta;

    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
        m->num_cus = prop.multiProcessorCount;

    // Table: upper-case (:71), count no-calls (:73-74), encode (:75); stored as PRE-INVERTED
    // bit-planes [S][NW][4] so the kernel's inner op is (o & ~e) with no NOT.
    std::vector<uint32_t> table((size_t)n_samples * m->NW * 4, 0u);
    std::vector<std::vector<uint8_t>> enc(n_samples, std::vector<uint8_t>(barcode_len));
    uint32_t max_ns = 0;
    bool plain = true;
    m->barcodes_upper.resize(n_samples);
    for (uint32_t s = 0; s < n_samples; ++s) {
        uint32_t ns = 0;
        for (uint32_t i = 0; i < barcode_len; ++i) {
            uint8_t b = (uint8_t)barcodes[s][i];
            if (b >= 'a' && b <= 'z') b = (uint8_t)(b - 32);
            m->barcodes_upper[s].push_back((char)b);
            if (b == 'N' || b == 'n' || b == '.') ns++;
            const uint8_t e = enc_byte(b);
            enc[s][i] = e;
            plain = plain && (e == 1 || e == 2 || e == 4 || e == 8);
            const uint32_t w = i / 32, bit = i % 32;
            for (uint32_t j = 0; j < 4; ++j)
                if (!((e >> j) & 1u)) table[((size_t)s * m->NW + w) * 4 + j] |= (1u << bit);
        }
        max_ns = std::max(max_ns, ns);
    }
    m->max_ns = max_ns;
    m->plain_samples = plain;

    std::vector<uint32_t> lut(256);
    for (int b = 0; b < 256; ++b) {
        const uint32_t e = enc_byte((uint8_t)b);
        lut[b] = (e & 1u) | (((e >> 1) & 1u) << 8) | (((e >> 2) & 1u) << 16) | (((e >> 3) & 1u) << 24);
    }

    auto cleanup = [&](int code) {
        fqtk_matcher_destroy(m);
        return code;
    };
#define HIP_TRY_C(expr)                                                                      \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) {                                                              \
            return cleanup(fail(FQTK_EHIP, std::string(#expr) + ": " + hipGetErrorString(_e))); \
        }                                                                                    \
    } while (0)
    HIP_TRY_C(hipMalloc(reinterpret_cast<void **>(&m->d_table), table.size() * sizeof(uint32_t)));
    HIP_TRY_C(hipMemcpy(m->d_table, table.data(), table.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIP_TRY_C(hipMalloc(reinterpret_cast<void **>(&m->d_lut), 256 * sizeof(uint32_t)));
    HIP_TRY_C(hipMemcpy(m->d_lut, lut.data(), 256 * sizeof(uint32_t), hipMemcpyHostToDevice));
    constexpr size_t kErrBytes = (size_t)(kNumSlots + 1) * sizeof(unsigned long long);
    HIP_TRY_C(hipMalloc(reinterpret_cast<void **>(&m->d_err), kErrBytes));
    HIP_TRY_C(hipMemset(m->d_err, 0xFF, kErrBytes));
    HIP_TRY_C(hipMalloc(reinterpret_cast<void **>(&m->d_counts), (size_t)(n_samples + 1) * sizeof(unsigned long long)));
    HIP_TRY_C(hipMemset(m->d_counts, 0, (size_t)(n_samples + 1) * sizeof(unsigned long long)));
    HIP_TRY_C(hipMalloc(reinterpret_cast<void **>(&m->d_counts_sync), (size_t)(n_samples + 1) * sizeof(unsigned long long)));
    HIP_TRY_C(hipMemset(m->d_counts_sync, 0, (size_t)(n_samples + 1) * sizeof(unsigned long long)));
    HIP_TRY_C(hipHostMalloc(reinterpret_cast<void **>(&m->h_err), kErrBytes, hipHostMallocDefault));
    std::memset(m->h_err, 0xFF, kErrBytes);
    {
        void *seen = nullptr;
        HIP_TRY_C(hipHostMalloc(&seen, sizeof(uint32_t), hipHostMallocMapped));
        m->h_seen = static_cast<volatile uint32_t *>(seen);
        m->h_seen[0] = 0;
        HIP_TRY_C(hipHostGetDevicePointer(reinterpret_cast<void **>(&m->d_seen), seen, 0));
    }
    // The pipeline slots use NON-BLOCKING streams, which do not order against the legacy NULL stream the
    // initialisation above ran on: make it all visible before any slot stream touches these buffers.
    HIP_TRY_C(hipDeviceSynchronize());
#undef HIP_TRY_C
    {
        const int rc = build_memo(m, enc);
        if (rc != FQTK_OK) return cleanup(rc);
    }
    *out = m;
    return FQTK_OK;
}

void fqtk_matcher_destroy(fqtk_matcher *m) {
    if (!m) return;
    (void)hipSetDevice(m->device);
    for (Slot &s : m->slots) {
        if (s.stream) {
            (void)hipStreamSynchronize(s.stream);
            (void)hipStreamDestroy(s.stream);
        }
        if (s.d_obs) (void)hipFree(s.d_obs);
        if (s.d_len) (void)hipFree(s.d_len);
        if (s.d_out) (void)hipFree(s.d_out);
        if (s.d_packed) (void)hipFree(s.d_packed);
        if (s.d_exc_index) (void)hipFree(s.d_exc_index);
        if (s.d_exc_rows) (void)hipFree(s.d_exc_rows);
        s.work.release();
    }
    if (m->d_memo) (void)hipFree(m->d_memo);
    if (m->d_hot) (void)hipFree(m->d_hot);
    if (m->d_filter) (void)hipFree(m->d_filter);
    for (auto &e : m->stream_work) e.second.release();
    if (m->d_direct) (void)hipFree(m->d_direct);
    if (m->d_hot2) (void)hipFree(m->d_hot2);
    if (m->d_ldsm) (void)hipFree(m->d_ldsm);
    if (m->d_table) (void)hipFree(m->d_table);
    if (m->d_lut) (void)hipFree(m->d_lut);
    if (m->d_err) (void)hipFree(m->d_err);
    if (m->d_counts) (void)hipFree(m->d_counts);
    if (m->d_counts_sync) (void)hipFree(m->d_counts_sync);
    if (m->h_err) (void)hipHostFree(m->h_err);
    if (m->h_seen) (void)hipHostFree(const_cast<uint32_t *>(m->h_seen));
    delete m;
}

uint32_t fqtk_matcher_n_samples(const fqtk_matcher *m) { return m ? m->S : 0; }
uint32_t fqtk_matcher_barcode_len(const fqtk_matcher *m) { return m ? m->L : 0; }
uint32_t fqtk_matcher_max_ns_in_barcodes(const fqtk_matcher *m) { return m ? m->max_ns : 0; }
int fqtk_matcher_device(const fqtk_matcher *m) { return m ? m->device : -1; }

int fqtk_matcher_set_sample_ids(fqtk_matcher *m, const char *const *sample_ids) {
    if (!m) return fail(FQTK_EINVAL, "matcher is NULL");
    m->sample_ids.clear();
    if (!sample_ids) return FQTK_OK;
    for (uint32_t s = 0; s < m->S; ++s) {
        if (!sample_ids[s]) { m->sample_ids.clear(); return fail(FQTK_EINVAL, "sample id is NULL"); }
        m->sample_ids.emplace_back(sample_ids[s]);
    }
    return FQTK_OK;
}

int fqtk_matcher_set_use_cache(fqtk_matcher *m, int use_cache) {
    if (!m) return fail(FQTK_EINVAL, "matcher is NULL");
    m->use_cache = use_cache ? 1 : 0;
    return FQTK_OK;
}
uint64_t fqtk_matcher_memo_entries(const fqtk_matcher *m) { return (m && m->d_memo) ? m->memo_entries : 0; }
uint64_t fqtk_matcher_memo_candidates(const fqtk_matcher *m) { return (m && m->d_memo) ? m->memo_candidates : 0; }
int fqtk_matcher_memo_direct_bytes(const fqtk_matcher *m) { return (m && m->d_memo && m->d_direct) ? m->direct_bytes : 0; }

int fqtk_matcher_memo_kind(const fqtk_matcher *m) {
    if (!m || !m->use_cache) return FQTK_MEMO_NONE;
    if (m->d_ldsm && m->memo_kind_wanted != FQTK_MEMO_TABLE) return FQTK_MEMO_LDS;
    return m->d_memo ? FQTK_MEMO_TABLE : FQTK_MEMO_NONE;
}

int fqtk_matcher_set_memo_kind(fqtk_matcher *m, int kind) {
    if (!m) return fail(FQTK_EINVAL, "matcher is NULL");
    if (kind != FQTK_MEMO_TABLE && kind != FQTK_MEMO_LDS) return fail(FQTK_EINVAL, "memo kind must be FQTK_MEMO_TABLE or FQTK_MEMO_LDS");
    m->memo_kind_wanted = kind == FQTK_MEMO_TABLE ? FQTK_MEMO_TABLE : 0;
    return FQTK_OK;
}

int fqtk_matcher_assign_batch_device(fqtk_matcher *m, const void *d_obs, uint32_t stride,
                                     const void *d_obs_len, uint64_t n, void *d_out, void *d_counts,
                                     void *hip_stream) {
    int rc = check_batch_args(m, d_obs, stride, d_obs_len, n, d_out);
    if (rc != FQTK_OK || n == 0) return rc;
    HIP_TRY(hipSetDevice(m->device));
    const fqtk::MatchParams P = make_params(m, d_obs, stride, d_obs_len, n, d_out, d_counts, kDeviceErrWord);
    m->device_ctx = ErrCtx{static_cast<const uint8_t *>(d_obs), static_cast<const uint32_t *>(d_obs_len), stride, n, true};
    Worklist *wl = nullptr;
    if ((rc = worklist_of_stream(m, static_cast<hipStream_t>(hip_stream), &wl)) != FQTK_OK) return rc;
    return launch(m, P, static_cast<hipStream_t>(hip_stream), *wl);
}

int fqtk_matcher_poll_error(fqtk_matcher *m, void *hip_stream, uint64_t *read_index) {
    if (!m) return fail(FQTK_EINVAL, "matcher is NULL");
    HIP_TRY(hipSetDevice(m->device));
    return collect_error(m, static_cast<hipStream_t>(hip_stream), kDeviceErrWord, m->device_ctx, read_index);
}

}  // extern "C"

namespace {
// One chunk on one pipeline slot; counts go to `d_counts_target` (device, S+1) or nowhere (NULL).
int enqueue_impl(fqtk_matcher *m, int slot, const uint8_t *obs, uint32_t stride, const uint32_t *obs_len,
                 uint64_t n, fqtk_match_t *out, unsigned long long *d_counts_target) {
    int rc = check_batch_args(m, obs, stride, obs_len, n, out);
    if (rc != FQTK_OK) return rc;
    HIP_TRY(hipSetDevice(m->device));
    rc = ensure_slot(m, slot);
    if (rc != FQTK_OK) return rc;
    Slot &s = m->slots[slot];
    if (s.busy) return fail(FQTK_EINVAL, "slot is busy: call fqtk_matcher_wait() first");
    if (n == 0) return FQTK_OK;
    if (obs_len)   // a length beyond its row would make the kernels (and the error report) read past it
        for (uint64_t i = 0; i < n; ++i)
            if (obs_len[i] > stride)
                return fail(FQTK_EINVAL, "obs_len[" + std::to_string(i) + "] = " + std::to_string(obs_len[i]) +
                                             " exceeds stride " + std::to_string(stride));
    const size_t obs_bytes = (size_t)n * stride;
    if ((rc = ensure_cap(s.d_obs, s.obs_cap, obs_bytes + 16)) != FQTK_OK) return rc;
    if ((rc = ensure_cap(s.d_out, s.out_cap, (size_t)n)) != FQTK_OK) return rc;
    if (obs_len && (rc = ensure_cap(s.d_len, s.len_cap, (size_t)n)) != FQTK_OK) return rc;
    HIP_TRY(hipMemcpyAsync(s.d_obs, obs, obs_bytes, hipMemcpyHostToDevice, s.stream));
    if (obs_len)
        HIP_TRY(hipMemcpyAsync(s.d_len, obs_len, (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice, s.stream));
    const fqtk::MatchParams P =
        make_params(m, s.d_obs, stride, obs_len ? s.d_len : nullptr, n, s.d_out, d_counts_target, slot);
    s.ctx = ErrCtx{obs, obs_len, stride, n, false};
    rc = launch(m, P, s.stream, s.work);
    if (rc != FQTK_OK) return rc;
    HIP_TRY(hipMemcpyAsync(out, s.d_out, (size_t)n * sizeof(fqtk_match_t), hipMemcpyDeviceToHost, s.stream));
    s.busy = true;
    return FQTK_OK;
}
}  // namespace

namespace {
int wait_impl(fqtk_matcher *m, int slot) {
    Slot &s = m->slots[slot];
    if (!s.busy) return FQTK_OK;
    HIP_TRY(hipSetDevice(m->device));
    HIP_TRY(hipStreamSynchronize(s.stream));
    s.busy = false;
    return collect_error(m, s.stream, slot, s.ctx, nullptr);   // this slot's own error word
}
}  // namespace

// ---- packed input (include/fqtk_match.h: fqtk_pack_barcodes / fqtk_matcher_enqueue_packed) -----------------------------
// Over PCIe a barcode costs its ASCII bytes; 4 bits per base halve that (cfg 3: 8 + 4 instead of 16 + 4 bytes per read).
// The packed rows are turned back into ASCII rows in HBM -- 100x the link's bandwidth -- and the matcher's kernels run
// on those unchanged, so every path (LDS / table / direct memo, second pass, scan, counts) is the one the ASCII entry
// takes and results are identical by construction.
namespace fqtk {
__global__ __launch_bounds__(256) void unpack_kernel(const uint8_t *packed, uint32_t ps, uint32_t L, uint64_t n, uint8_t *obs, uint32_t stride) {
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint32_t *src = reinterpret_cast<const uint32_t *>(packed + i * ps);
    uint32_t *dst = reinterpret_cast<uint32_t *>(obs + i * stride);
    const uint32_t out_words = stride >> 2;
    for (uint32_t w = 0; w < out_words; ++w) {
        const uint32_t sw = w >> 1;
        const uint32_t x = sw < (ps >> 2) ? (src[sw] >> (16u * (w & 1u))) & 0xFFFFu : 0u;   // the four nibbles of bases 4w .. 4w + 3
        const uint32_t c = (x & 0xFu) | ((x & 0xF0u) << 4) | ((x & 0xF00u) << 8) | ((x & 0xF000u) << 12);
        uint32_t chars = __builtin_amdgcn_perm(kCodePoolHi, kCodePoolLo, c & 0x07070707u);
        const int rem = (int)L - 4 * (int)w;                                                           // pad bytes are zero
        if (rem < 4) chars &= rem <= 0 ? 0u : ((1u << (8 * rem)) - 1u);
        dst[w] = chars;
    }
}
__global__ __launch_bounds__(256) void patch_rows_kernel(const uint32_t *index, const uint8_t *rows, uint32_t rows_stride, uint64_t n_exc,
                                                         uint64_t n, uint32_t L, uint8_t *obs, uint32_t stride) {
    const uint64_t j = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (j >= n_exc) return;
    const uint64_t i = index[j];
    if (i >= n) return;
    for (uint32_t k = 0; k < L; ++k) obs[i * stride + k] = rows[j * rows_stride + k];
}
}  // namespace fqtk

namespace {
// code of a base in a packed row (A 0, C 1, T 2, G 3, no-call 7), 0xFF for a byte the packed form cannot carry
struct PackLut {
    uint8_t code[256];
    PackLut() {
        std::memset(code, 0xFF, sizeof code);
        const char *b = "ACTG";
        for (int k = 0; k < 4; ++k) { code[(uint8_t)b[k]] = (uint8_t)k; code[(uint8_t)(b[k] | 0x20)] = (uint8_t)k; }
        code[(uint8_t)'N'] = code[(uint8_t)'n'] = code[(uint8_t)'.'] = 7;   // the no-calls (mod.rs:85-87)
    }
};
const PackLut kPackLut;

// One row: the bases from `k` on, a table look-up per base (the tail of a row behind the SIMD blocks; every base elsewhere).
inline void pack_row_tail(const uint8_t *r, uint32_t k, uint32_t barcode_len, uint8_t *o, uint32_t packed_stride, uint32_t *bad) {
    const uint8_t *lut = kPackLut.code;
    for (; k + 1 < barcode_len; k += 2) {
        const uint32_t a = lut[r[k]], b = lut[r[k + 1]];
        *bad |= a | b;
        o[k >> 1] = (uint8_t)((a & 7u) | ((b & 7u) << 4));
    }
    if (k < barcode_len) { const uint32_t a = lut[r[k]]; *bad |= a; o[k >> 1] = (uint8_t)(a & 7u); k += 2; }
    for (uint32_t j = k >> 1; j < packed_stride; ++j) o[j] = 0;
}
// Returns false when there are more exception rows than exc_cap.
inline bool pack_exception(const uint8_t *r, uint64_t i, uint32_t barcode_len, uint32_t *exc_index, uint8_t *exc_rows, uint64_t exc_cap, uint64_t *ne) {
    if (*ne >= exc_cap || !exc_index || !exc_rows) return false;
    exc_index[*ne] = (uint32_t)i;
    std::memcpy(exc_rows + *ne * barcode_len, r, barcode_len);
    ++*ne;
    return true;
}
bool pack_rows_scalar(const uint8_t *obs, uint32_t stride, uint32_t barcode_len, uint64_t n, uint8_t *packed, uint32_t packed_stride,
                      uint32_t *exc_index, uint8_t *exc_rows, uint64_t exc_cap, uint64_t *ne) {
    for (uint64_t i = 0; i < n; ++i) {
        const uint8_t *r = obs + i * stride;
        uint32_t bad = 0;
        pack_row_tail(r, 0, barcode_len, packed + i * packed_stride, packed_stride, &bad);
        if ((bad & 0x80u) && !pack_exception(r, i, barcode_len, exc_index, exc_rows, exc_cap, ne)) return false;
    }
    return true;
}
#if defined(__x86_64__)
// The same on 16 (then 8) bases at a time: code = bits 1..3 of the byte, one pshufb maps a code back to the letter it
// stands for ('A' 'C' 'T' 'G' . . . 'N': fqtk::kCodePool) -- the byte is canonical iff it equals that letter in either case,
// or is '.', the legacy no-call, which has N's code -- and one pmaddubsw puts two codes into a byte.  The whole row loop
// lives in this function so that the constants stay in registers (a per-row call re-made them: 150 M reads/s).
// One table look-up per base did 143 M reads/s per host thread (round 3).
__attribute__((target("ssse3"))) bool pack_rows_ssse3(const uint8_t *obs, uint32_t stride, uint32_t barcode_len, uint64_t n, uint8_t *packed,
                                                      uint32_t packed_stride, uint32_t *exc_index, uint8_t *exc_rows, uint64_t exc_cap, uint64_t *ne) {
    const __m128i pool = _mm_setr_epi8(0x41, 0x43, 0x54, 0x47, (char)0xFF, (char)0xFF, (char)0xFF, 0x4E, 0x41, 0x43, 0x54, 0x47, (char)0xFF, (char)0xFF, (char)0xFF, 0x4E);
    const __m128i seven = _mm_set1_epi8(7), upper = _mm_set1_epi8((char)0xDF), dotc = _mm_set1_epi8(0x2E), pair = _mm_set1_epi16(0x1001);
    const __m128i zero = _mm_setzero_si128();
    for (uint64_t i = 0; i < n; ++i) {
        const uint8_t *r = obs + i * stride;
        uint8_t *o = packed + i * packed_stride;
        __m128i flagged = zero;
        uint32_t k = 0;
        for (; k + 16 <= barcode_len; k += 16) {
            const __m128i v = _mm_loadu_si128(reinterpret_cast<const __m128i *>(r + k));
            const __m128i c = _mm_and_si128(_mm_srli_epi16(v, 1), seven);
            const __m128i e = _mm_shuffle_epi8(pool, c);
            flagged = _mm_or_si128(flagged, _mm_andnot_si128(_mm_cmpeq_epi8(v, dotc), _mm_and_si128(_mm_xor_si128(v, e), upper)));
            const __m128i two = _mm_maddubs_epi16(c, pair);             // code[2j] + 16 * code[2j + 1] in every 16-bit lane
            _mm_storel_epi64(reinterpret_cast<__m128i *>(o + (k >> 1)), _mm_packus_epi16(two, two));
        }
        if (k + 8 <= barcode_len) {
            const __m128i v = _mm_loadl_epi64(reinterpret_cast<const __m128i *>(r + k));
            const __m128i c = _mm_and_si128(_mm_srli_epi16(v, 1), seven);
            const __m128i e = _mm_shuffle_epi8(pool, c);
            const __m128i off = _mm_andnot_si128(_mm_cmpeq_epi8(v, dotc), _mm_and_si128(_mm_xor_si128(v, e), upper));
            flagged = _mm_or_si128(flagged, _mm_move_epi64(off));       // (the upper half holds zeros' "mismatch" with 'A')
            const __m128i two = _mm_maddubs_epi16(c, pair);
            const uint32_t w = (uint32_t)_mm_cvtsi128_si32(_mm_packus_epi16(two, two));
            std::memcpy(o + (k >> 1), &w, 4);
            k += 8;
        }
        uint32_t bad = _mm_movemask_epi8(_mm_cmpeq_epi8(flagged, zero)) != 0xFFFF ? 0x80u : 0u;
        pack_row_tail(r, k, barcode_len, o, packed_stride, &bad);
        if ((bad & 0x80u) && !pack_exception(r, i, barcode_len, exc_index, exc_rows, exc_cap, ne)) return false;
    }
    return true;
}
// The common case -- rows of exactly 16 bases, back to back (8 + 8 dual indices) -- two rows per step in one 256-bit
// register: the buffer is one stream, 32 bytes in, 16 out.
__attribute__((target("avx2"))) bool pack_rows16_avx2(const uint8_t *obs, uint64_t n, uint8_t *packed, uint32_t *exc_index, uint8_t *exc_rows,
                                                      uint64_t exc_cap, uint64_t *ne) {
    const __m256i pool = _mm256_setr_epi8(0x41, 0x43, 0x54, 0x47, (char)0xFF, (char)0xFF, (char)0xFF, 0x4E, 0x41, 0x43, 0x54, 0x47, (char)0xFF, (char)0xFF, (char)0xFF, 0x4E,
                                          0x41, 0x43, 0x54, 0x47, (char)0xFF, (char)0xFF, (char)0xFF, 0x4E, 0x41, 0x43, 0x54, 0x47, (char)0xFF, (char)0xFF, (char)0xFF, 0x4E);
    const __m256i seven = _mm256_set1_epi8(7), upper = _mm256_set1_epi8((char)0xDF), dotc = _mm256_set1_epi8(0x2E), pair = _mm256_set1_epi16(0x1001);
    const __m256i zero = _mm256_setzero_si256();
    uint64_t i = 0;
    for (; i + 2 <= n; i += 2) {
        const __m256i v = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(obs + i * 16));
        const __m256i c = _mm256_and_si256(_mm256_srli_epi16(v, 1), seven);
        const __m256i e = _mm256_shuffle_epi8(pool, c);
        const __m256i off = _mm256_andnot_si256(_mm256_cmpeq_epi8(v, dotc), _mm256_and_si256(_mm256_xor_si256(v, e), upper));
        const __m256i two = _mm256_maddubs_epi16(c, pair);
        const __m256i pk = _mm256_permute4x64_epi64(_mm256_packus_epi16(two, two), 0x08);   // quadwords 0 and 2: the two rows
        _mm_storeu_si128(reinterpret_cast<__m128i *>(packed + i * 8), _mm256_castsi256_si128(pk));
        const uint32_t clean = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(off, zero));
        if (clean != 0xFFFFFFFFu) {
            if ((clean & 0xFFFFu) != 0xFFFFu && !pack_exception(obs + i * 16, i, 16, exc_index, exc_rows, exc_cap, ne)) return false;
            if ((clean >> 16) != 0xFFFFu && !pack_exception(obs + (i + 1) * 16, i + 1, 16, exc_index, exc_rows, exc_cap, ne)) return false;
        }
    }
    if (i < n) {
        uint64_t ne1 = 0;
        uint32_t idx1 = 0;
        uint8_t row1[16];
        if (!pack_rows_ssse3(obs + i * 16, 16, 16, 1, packed + i * 8, 8, &idx1, row1, 1, &ne1)) return false;
        if (ne1 && !pack_exception(obs + i * 16, i, 16, exc_index, exc_rows, exc_cap, ne)) return false;
    }
    return true;
}
#endif
}  // namespace

extern "C" {

uint32_t fqtk_packed_stride(uint32_t barcode_len) { return (((barcode_len + 1u) / 2u) + 3u) & ~3u; }

int fqtk_pack_barcodes(const uint8_t *obs, uint32_t stride, uint32_t barcode_len, uint64_t n, uint8_t *packed, uint32_t packed_stride,
                       uint32_t *exc_index, uint8_t *exc_rows, uint64_t exc_cap, uint64_t *n_exc) {
    if (!obs || !packed || !n_exc) return fail(FQTK_EINVAL, "NULL argument");
    if (barcode_len == 0 || barcode_len > FQTK_MAX_BARCODE_LEN || stride < barcode_len) return fail(FQTK_EINVAL, "stride < barcode_len, or barcode_len out of range");
    if (packed_stride < fqtk_packed_stride(barcode_len)) return fail(FQTK_EINVAL, "packed_stride smaller than fqtk_packed_stride(barcode_len)");
    if (n > 0xFFFFFFFFull) return fail(FQTK_EINVAL, "at most 2^32 - 1 reads per call");
    uint64_t ne = 0;
    bool ok;
#if defined(__x86_64__)
    if (barcode_len == 16 && stride == 16 && packed_stride == 8 && __builtin_cpu_supports("avx2")) ok = pack_rows16_avx2(obs, n, packed, exc_index, exc_rows, exc_cap, &ne);
    else if (__builtin_cpu_supports("ssse3")) ok = pack_rows_ssse3(obs, stride, barcode_len, n, packed, packed_stride, exc_index, exc_rows, exc_cap, &ne);
    else
#endif
        ok = pack_rows_scalar(obs, stride, barcode_len, n, packed, packed_stride, exc_index, exc_rows, exc_cap, &ne);
    if (!ok) return fail(FQTK_ENOMEM, "more reads with bytes outside A C G T N . than exc_cap");
    *n_exc = ne;
    return FQTK_OK;
}

int fqtk_matcher_enqueue_packed(fqtk_matcher *m, int slot, const uint8_t *packed, uint32_t packed_stride, uint64_t n,
                                const uint32_t *exc_index, const uint8_t *exc_rows, uint64_t n_exc, fqtk_match_t *out) {
    if (!m) return fail(FQTK_EINVAL, "matcher is NULL");
    if (slot < 0 || slot >= FQTK_MAX_SLOTS) return fail(FQTK_EINVAL, "slot out of range");
    if (n == 0) return FQTK_OK;
    if (!packed || !out || (n_exc && (!exc_index || !exc_rows))) return fail(FQTK_EINVAL, "NULL argument");
    if (packed_stride < fqtk_packed_stride(m->L) || packed_stride % 4) return fail(FQTK_EINVAL, "packed_stride must be a multiple of 4 and at least fqtk_packed_stride(barcode_len)");
    if (n > 0xFFFFFFFFull) return fail(FQTK_EINVAL, "at most 2^32 - 1 reads per chunk");
    HIP_TRY(hipSetDevice(m->device));
    int rc = ensure_slot(m, slot);
    if (rc != FQTK_OK) return rc;
    Slot &s = m->slots[slot];
    if (s.busy) return fail(FQTK_EINVAL, "slot is busy: call fqtk_matcher_wait() first");
    const uint32_t stride = (m->L + 3u) & ~3u;
    if ((rc = ensure_cap(s.d_packed, s.packed_cap, (size_t)n * packed_stride + 16)) != FQTK_OK) return rc;
    if ((rc = ensure_cap(s.d_obs, s.obs_cap, (size_t)n * stride + 16)) != FQTK_OK) return rc;
    if ((rc = ensure_cap(s.d_out, s.out_cap, (size_t)n)) != FQTK_OK) return rc;
    HIP_TRY(hipMemcpyAsync(s.d_packed, packed, (size_t)n * packed_stride, hipMemcpyHostToDevice, s.stream));
    hipLaunchKernelGGL(fqtk::unpack_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s.stream, s.d_packed, packed_stride, m->L, n, s.d_obs, stride);
    HIP_TRY(hipGetLastError());
    if (n_exc) {
        if ((rc = ensure_cap(s.d_exc_index, s.exc_index_cap, (size_t)n_exc)) != FQTK_OK) return rc;
        if ((rc = ensure_cap(s.d_exc_rows, s.exc_rows_cap, (size_t)n_exc * m->L)) != FQTK_OK) return rc;
        HIP_TRY(hipMemcpyAsync(s.d_exc_index, exc_index, (size_t)n_exc * sizeof(uint32_t), hipMemcpyHostToDevice, s.stream));
        HIP_TRY(hipMemcpyAsync(s.d_exc_rows, exc_rows, (size_t)n_exc * m->L, hipMemcpyHostToDevice, s.stream));
        hipLaunchKernelGGL(fqtk::patch_rows_kernel, dim3((uint32_t)((n_exc + 255) / 256)), dim3(256), 0, s.stream, s.d_exc_index, s.d_exc_rows,
                           m->L, n_exc, n, m->L, s.d_obs, stride);
        HIP_TRY(hipGetLastError());
    }
    const fqtk::MatchParams P = make_params(m, s.d_obs, stride, nullptr, n, s.d_out, m->d_counts, slot);
    s.ctx = ErrCtx{nullptr, nullptr, stride, n, false};   // (every row is exactly one barcode long: no length error can arise)
    rc = launch(m, P, s.stream, s.work);
    if (rc != FQTK_OK) return rc;
    HIP_TRY(hipMemcpyAsync(out, s.d_out, (size_t)n * sizeof(fqtk_match_t), hipMemcpyDeviceToHost, s.stream));
    s.busy = true;
    return FQTK_OK;
}

}  // extern "C"

extern "C" {

int fqtk_matcher_enqueue(fqtk_matcher *m, int slot, const uint8_t *obs, uint32_t stride,
                         const uint32_t *obs_len, uint64_t n, fqtk_match_t *out) {
    if (!m) return fail(FQTK_EINVAL, "matcher is NULL");
    if (slot < 0 || slot >= FQTK_MAX_SLOTS) return fail(FQTK_EINVAL, "slot out of range");
    return enqueue_impl(m, slot, obs, stride, obs_len, n, out, m->d_counts);
}

int fqtk_matcher_wait(fqtk_matcher *m, int slot) {
    if (!m) return fail(FQTK_EINVAL, "matcher is NULL");
    if (slot < 0 || slot >= FQTK_MAX_SLOTS) return fail(FQTK_EINVAL, "slot out of range");
    return wait_impl(m, slot);
}

int fqtk_matcher_counts(fqtk_matcher *m, uint64_t *counts) {
    if (!m || !counts) return fail(FQTK_EINVAL, "NULL argument");
    HIP_TRY(hipSetDevice(m->device));
    for (Slot &s : m->slots)
        if (s.stream) HIP_TRY(hipStreamSynchronize(s.stream));
    const size_t bins = (size_t)m->S + 1;
    std::vector<unsigned long long> tmp(bins);
    HIP_TRY(hipMemcpy(tmp.data(), m->d_counts, bins * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemset(m->d_counts, 0, bins * sizeof(unsigned long long)));
    HIP_TRY(hipStreamSynchronize(nullptr));   // NULL-stream memset vs. the next chunk on a non-blocking slot stream
    for (size_t b = 0; b < bins; ++b) counts[b] += (uint64_t)tmp[b];
    return FQTK_OK;
}

int fqtk_matcher_assign_batch(fqtk_matcher *m, const uint8_t *obs, uint32_t stride,
                              const uint32_t *obs_len, uint64_t n, fqtk_match_t *out,
                              uint64_t *counts) {
    int rc = check_batch_args(m, obs, stride, obs_len, n, out);
    if (rc != FQTK_OK || n == 0) return rc;
    HIP_TRY(hipSetDevice(m->device));
    const size_t bins = (size_t)m->S + 1;
    // counts of THIS call only: a private accumulator, so a concurrent enqueue()/counts() session on
    // the same handle keeps its own totals
    if (counts) {
        HIP_TRY(hipMemset(m->d_counts_sync, 0, bins * sizeof(unsigned long long)));
        HIP_TRY(hipStreamSynchronize(nullptr));   // must land before the kernels on the non-blocking slot streams
    }
    // Chunk so staging stays bounded, ping-pong over two slots so copy and compute overlap.
    const uint64_t chunk = std::max<uint64_t>(1, std::min<uint64_t>(n, (256ull << 20) / stride));
    int first_err = FQTK_OK;
    std::string first_msg;
    uint64_t done = 0;
    int k = 0;
    uint64_t pending_base[2] = {0, 0};
    auto drain = [&](int slot) {   // slot = kSyncSlot0 + {0, 1}: private to this call
        int w = wait_impl(m, slot);
        if (w != FQTK_OK && first_err == FQTK_OK) {
            first_err = w;
            first_msg = g_last_error;
            if (w == FQTK_ELEN) first_msg += " (chunk base " + std::to_string(pending_base[slot - kSyncSlot0]) + ")";
        }
    };
    while (done < n) {
        const int slot = kSyncSlot0 + (k & 1);
        drain(slot);
        const uint64_t cur = std::min(chunk, n - done);
        pending_base[slot - kSyncSlot0] = done;
        rc = enqueue_impl(m, slot, obs + done * stride, stride, obs_len ? obs_len + done : nullptr, cur,
                          out + done, counts ? m->d_counts_sync : nullptr);
        if (rc != FQTK_OK) {
            const std::string msg = g_last_error;
            drain(kSyncSlot0);
            drain(kSyncSlot0 + 1);
            return fail(rc, msg);
        }
        done += cur;
        ++k;
    }
    drain(kSyncSlot0);
    drain(kSyncSlot0 + 1);
    if (counts) {
        std::vector<unsigned long long> tmp(bins);
        HIP_TRY(hipMemcpy(tmp.data(), m->d_counts_sync, bins * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        for (size_t b = 0; b < bins; ++b) counts[b] += (uint64_t)tmp[b];
    }
    if (first_err != FQTK_OK) return fail(first_err, first_msg);
    return FQTK_OK;
}

int fqtk_matcher_assign1(fqtk_matcher *m, const uint8_t *read_bases, uint32_t len, fqtk_match_t *out) {
    if (!m || !out) return fail(FQTK_EINVAL, "NULL argument");
    if (len == 0 || !read_bases) {  // barcode_matching.rs:167-169: shorter than expected -> None
        out->idx = FQTK_NO_MATCH;
        out->best = 255;
        out->next = 255;
        return FQTK_OK;
    }
    const uint32_t l = len;
    return fqtk_matcher_assign_batch(m, read_bases, len, &l, 1, out, nullptr);
}

int fqtk_pinned_alloc(size_t bytes, void **out) {
    if (!out) return fail(FQTK_EINVAL, "out is NULL");
    *out = nullptr;
    HIP_TRY(hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault));
    return FQTK_OK;
}

int fqtk_pinned_free(void *p) {
    if (!p) return FQTK_OK;
    HIP_TRY(hipHostFree(p));
    return FQTK_OK;
}


// ---- the one collective: per-sample counts over RCCL (single process, one communicator rank per device) ----
namespace {
struct Rccl {   // the six entry points used, bound at first use (libfqtk_match.so does not link RCCL)
    bool ok = false;
    std::string why;
    int (*CommInitAll)(void **, int, const int *) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    static const Rccl &get() {
        static const Rccl r = [] {
            Rccl x;
            void *h = nullptr;
            for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
                if ((h = dlopen(name, RTLD_NOW | RTLD_LOCAL))) break;
            if (!h) { x.why = std::string("cannot load librccl: ") + dlerror(); return x; }
            x.CommInitAll = reinterpret_cast<int (*)(void **, int, const int *)>(dlsym(h, "ncclCommInitAll"));
            x.CommDestroy = reinterpret_cast<int (*)(void *)>(dlsym(h, "ncclCommDestroy"));
            x.AllReduce = reinterpret_cast<int (*)(const void *, void *, size_t, int, int, void *, hipStream_t)>(dlsym(h, "ncclAllReduce"));
            x.GroupStart = reinterpret_cast<int (*)()>(dlsym(h, "ncclGroupStart"));
            x.GroupEnd = reinterpret_cast<int (*)()>(dlsym(h, "ncclGroupEnd"));
            x.GetErrorString = reinterpret_cast<const char *(*)(int)>(dlsym(h, "ncclGetErrorString"));
            x.ok = x.CommInitAll && x.CommDestroy && x.AllReduce && x.GroupStart && x.GroupEnd && x.GetErrorString;
            if (!x.ok) x.why = "librccl lacks an expected symbol";
            return x;
        }();
        return r;
    }
};
constexpr int kNcclUint64 = 5, kNcclSum = 0;   // rccl.h: ncclDataType_t / ncclRedOp_t
}  // namespace

int fqtk_matchers_allreduce_counts(fqtk_matcher *const *ms, int n, int force_collective, uint64_t *counts) {
    if (!ms || n <= 0 || !counts) return fail(FQTK_EINVAL, "NULL / empty argument");
    for (int i = 0; i < n; ++i) {
        if (!ms[i]) return fail(FQTK_EINVAL, "matcher is NULL");
        if (ms[i]->S != ms[0]->S) return fail(FQTK_EINVAL, "matchers with different sample tables");
        for (int j = 0; j < i; ++j)
            if (ms[j]->device == ms[i]->device)
                return fail(FQTK_EINVAL, "fqtk_matchers_allreduce_counts needs one matcher per DISTINCT device");
    }
    if (n == 1 && !force_collective) return fqtk_matcher_counts(ms[0], counts);
    const Rccl &R = Rccl::get();
    if (!R.ok) return fail(FQTK_ENCCL, R.why);
    const size_t bins = (size_t)ms[0]->S + 1;
    std::vector<int> devs(n);
    std::vector<void *> comms(n, nullptr);
    std::vector<hipStream_t> streams(n, nullptr);
    std::vector<unsigned long long *> recv(n, nullptr);
    for (int i = 0; i < n; ++i) devs[i] = ms[i]->device;
    int rc = FQTK_OK;
    auto nccl_fail = [&](int e, const char *what) { rc = fail(FQTK_ENCCL, std::string(what) + ": " + R.GetErrorString(e)); };
    auto hip_fail = [&](hipError_t e, const char *what) { rc = fail(FQTK_EHIP, std::string(what) + ": " + hipGetErrorString(e)); };
    int e = R.CommInitAll(comms.data(), n, devs.data());
    if (e != 0) { nccl_fail(e, "ncclCommInitAll"); return rc; }
    for (int i = 0; i < n && rc == FQTK_OK; ++i) {
        hipError_t he = hipSetDevice(devs[i]);
        if (he == hipSuccess) he = hipDeviceSynchronize();   // every chunk of every slot has been counted
        if (he == hipSuccess) he = hipStreamCreateWithFlags(&streams[i], hipStreamNonBlocking);
        if (he == hipSuccess) he = hipMalloc(reinterpret_cast<void **>(&recv[i]), bins * sizeof(unsigned long long));
        if (he != hipSuccess) hip_fail(he, "preparing the count reduction");
    }
    if (rc == FQTK_OK) {
        if ((e = R.GroupStart()) != 0) nccl_fail(e, "ncclGroupStart");
        for (int i = 0; i < n && rc == FQTK_OK; ++i)
            if ((e = R.AllReduce(ms[i]->d_counts, recv[i], bins, kNcclUint64, kNcclSum, comms[i], streams[i])) != 0)
                nccl_fail(e, "ncclAllReduce");
        if ((e = R.GroupEnd()) != 0 && rc == FQTK_OK) nccl_fail(e, "ncclGroupEnd");
    }
    std::vector<unsigned long long> total(bins, 0);
    for (int i = 0; i < n && rc == FQTK_OK; ++i) {
        hipError_t he = hipSetDevice(devs[i]);
        if (he == hipSuccess) he = hipStreamSynchronize(streams[i]);
        if (he == hipSuccess && i == 0)   // every rank holds the same sum: read rank 0's
            he = hipMemcpy(total.data(), recv[0], bins * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        if (he == hipSuccess) he = hipMemset(ms[i]->d_counts, 0, bins * sizeof(unsigned long long));
        if (he == hipSuccess) he = hipDeviceSynchronize();
        if (he != hipSuccess) hip_fail(he, "finishing the count reduction");
    }
    for (int i = 0; i < n; ++i) {
        (void)hipSetDevice(devs[i]);
        if (recv[i]) (void)hipFree(recv[i]);
        if (streams[i]) (void)hipStreamDestroy(streams[i]);
        if (comms[i]) (void)R.CommDestroy(comms[i]);
    }
    if (rc != FQTK_OK) return rc;
    for (size_t b = 0; b < bins; ++b) counts[b] += (uint64_t)total[b];
    return FQTK_OK;
}

}  // extern "C"

// ---- internal to libfqtk_match.so (matcher_internal.hpp): the record pipeline of fqtk_demux.hip ----------------------
namespace fqtk {
namespace internal {
int set_error(int code, const std::string &msg) { return fail(code, msg); }

int take_device_error(fqtk_matcher *m, hipStream_t stream, unsigned long long *d_dst) {
    unsigned long long *w = m->d_err + kDeviceErrWord;
    HIP_TRY(hipMemcpyAsync(d_dst, w, sizeof(unsigned long long), hipMemcpyDeviceToDevice, stream));
    HIP_TRY(hipMemsetAsync(w, 0xFF, sizeof(unsigned long long), stream));
    return FQTK_OK;
}

int word_device_length_error(fqtk_matcher *m, const uint8_t *d_obs, const uint32_t *d_lens, uint32_t stride, uint64_t n, uint64_t index) {
    (void)hipSetDevice(m->device);
    return word_length_error(m, ErrCtx{d_obs, d_lens, stride, n, true}, index);
}
}  // namespace internal
}  // namespace fqtk
