// fqtk_inflate.hip -- device side and C ABI (include/fqtk_inflate.h) of the BGZF member decoder.
// The algorithm lives in bgzf_inflate.hpp (one function, shared with the CPU test-suite's wave emulator); this file gives
// it a real wavefront, adds the per-member check (CRC-32 of the text against the trailer, newline count) and the handle.
#include <hip/hip_runtime.h>

#include <new>
#include <string>

#include "../../include/fqtk_inflate.h"
#include "../../include/fqtk_match.h"
#include "bgzf_deflate.hpp"   // crc_gf_mul / crc_x_pow
#include "bgzf_inflate.hpp"
#include "inflate_internal.hpp"

namespace fqtk {
namespace inflate {

// The W of bgzf_inflate.hpp on gfx950: a workgroup is ONE wavefront, so a barrier is an LDS wait and wave-uniform values
// live in scalar registers (readfirstlane tells the compiler so).
struct DeviceWave {
    __device__ inline uint32_t lane() const { return threadIdx.x; }
    __device__ inline uint64_t ballot(bool p) const { return __ballot(p); }
    __device__ inline uint32_t readlane(uint32_t v, uint32_t l) const {
        return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)__builtin_amdgcn_readfirstlane((int)l));
    }
    __device__ inline uint32_t uniform(uint32_t v) const { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
    // m |= 1 << bit for a wave-uniform m and bit: one scalar instruction (the compiler writes s_lshl_b64 + s_or_b64)
    __device__ inline void set_bit64(uint64_t &m, uint32_t bit) const { asm("s_bitset1_b64 %0, %1" : "+s"(m) : "s"(bit)); }
    // inclusive prefix sum over the 64 lanes in registers (DPP): within rows of 16 by shifts, then the rows' totals
    // (row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3)
    __device__ inline uint32_t scan_incl(uint32_t v) const {
        v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false);   // row_shr:1
        v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false);   // row_shr:2
        v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false);   // row_shr:4
        v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false);   // row_shr:8
        v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);   // row_bcast:15
        v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);   // row_bcast:31
        return v;
    }
    __device__ inline uint32_t scan_max_incl(uint32_t v) const {
        auto mx = [](uint32_t a, uint32_t b) { return a > b ? a : b; };
        v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false));
        v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false));
        v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false));
        v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false));
        v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false));
        v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false));
        return v;
    }
    __device__ inline uint32_t shuffle(uint32_t v, uint32_t l) const { return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(l << 2), (int)v); }
    __device__ inline float rcp(float x) const { return __builtin_amdgcn_rcpf(x); }
    __device__ inline void barrier() const { __syncthreads(); }
    __device__ inline void fence_global() const { __threadfence_block(); }
};

// Wavefronts per SIMD the decoders are compiled for (registers capped at 512 / that; tools/ab_inflate.sh "-DFQTK_INFLATE_WAVES=5").  A window of
// the decoder is one chain of dependent LDS look-ups, a scalar walk and, where a match reaches back, a round trip to memory: what keeps a
// SIMD busy is other wavefronts.  Six fit since round 6 (5.8 KB of LDS a wavefront, 80 registers); seven spill nothing but measure 3 % lower.
#ifndef FQTK_INFLATE_WAVES
#define FQTK_INFLATE_WAVES 6
#endif
#define FQTK_INFLATE_OCCUPANCY __attribute__((amdgpu_waves_per_eu(FQTK_INFLATE_WAVES, FQTK_INFLATE_WAVES)))
// CRC-32 (RFC 1952 8.) and newline count of the text a wavefront has just written, by that wavefront: lane k takes the k-th of 64 slices
// (whole 16-byte pieces of the dword-aligned stream; the bytes come back out of the L2 they were written to a moment ago), byte by byte through
// the table in LDS, and the slices' values -- each multiplied by x^(8 * bytes behind it) -- XOR to the member's (bgzf_deflate.hpp: crc_gf_mul; the
// powers are member_check_kernel's table: x^(8 * 264 * j), x^(8 * r)).  ~6 k wave-instructions per 64 KiB member beside the ~65 k that decode it;
// the separate check kernel (256 lanes per member, text staged in LDS) cost 1.0 ms per launch beside the decoder's 4.5 (r04_pipeline_bgzf_inputs_kernel_stats.csv).
constexpr uint32_t kSlice = 264, kSliceWords = kSlice / 4, kSliceStride = kSliceWords + 1;   // 256 x 264 >= 3 + 65 536
__device__ inline void wave_crc_and_lines(const uint8_t *text, uint32_t isize, const uint32_t *tab /* LDS, 256 */, const uint32_t *crc_pow, uint32_t *crc_out, uint32_t *lines_out) {
    const uint32_t lane = threadIdx.x;
    const uint32_t a = (uint32_t)(reinterpret_cast<uintptr_t>(text) & 3u), end = a + isize;   // the text is bytes [a, end) of the aligned stream
    const uint8_t *src = text - a;
    const uint32_t S = (((end + 63u) >> 6) + 15u) & ~15u;                                     // bytes per slice: 64 of them cover the stream
    const uint32_t lo0 = lane * S, lo = lo0 < a ? a : lo0, hi = lo0 + S < end ? lo0 + S : end;
    uint32_t c = 0xFFFFFFFFu, nl = 0;
    auto step = [&](uint32_t b) { nl += b == 0x0Au ? 1u : 0u; c = tab[(c ^ b) & 0xFFu] ^ (c >> 8); };
    uint32_t q = lo;
    for (; q < hi && (q & 15u); ++q) step(src[q]);                                            // up to the first whole piece
    for (uint32_t k = 0; k < S; k += 16u) {                                                   // (the same trip count in every lane)
        if (q + 16u <= hi) {
            const uint4 v = *reinterpret_cast<const uint4 *>(src + q);
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int b = 0; b < 4; ++b) step((w[i] >> (8 * b)) & 0xFFu);
            }
            q += 16u;
        }
    }
    for (; q < hi; ++q) step(src[q]);                                                          // what is left of the last piece
    uint32_t mine = 0;
    if (lo < hi) {
        c = ~c;
        const uint32_t behind = end - hi;                                                      // <= 65 536 + 3: 264 * j + r
        const uint32_t j = behind / kSlice, r = behind - j * kSlice;
        if (behind) {
            if (j) c = bgzf::crc_gf_mul(c, crc_pow[j]);
            if (r) c = bgzf::crc_gf_mul(c, crc_pow[256u + r]);
        }
        mine = c;
    }
    for (int d = 32; d >= 1; d >>= 1) { mine ^= __shfl_xor(mine, d); nl += __shfl_xor(nl, d); }
    *crc_out = mine;
    *lines_out = nl;
}

__global__ __launch_bounds__(64) FQTK_INFLATE_OCCUPANCY void inflate_kernel(const uint8_t *in, uint64_t in_len, const fqtk_inflate_member *members, uint32_t n,
                                                      uint8_t *out, uint32_t *status, uint32_t *lines, const uint32_t *crc_pow) {
    __shared__ Shared S;
    const uint32_t j = blockIdx.x;
    if (j >= n) return;
    const fqtk_inflate_member m = members[j];
    uint32_t st, nl = 0;
    if (m.isize > FQTK_INFLATE_MAX_ISIZE || m.payload_off > in_len || (uint64_t)m.payload_len > in_len - m.payload_off) {
        st = kErrTruncated;
    } else {
        MemberArgs a;
        const uint64_t base = m.payload_off & ~(uint64_t)3;
        a.in_words = reinterpret_cast<const uint32_t *>(in + base);
        a.first_bit = 8u * (uint32_t)(m.payload_off & 3u);
        a.payload_bits = 8u * m.payload_len;
        const uint64_t words_left = (in_len - base) / 4u;   // whole dwords of the caller's buffer from `base` on
        a.readable_words = words_left > 0x7FFFFFFFull ? 0x7FFFFFFFu : (uint32_t)words_left;
        a.tail_bytes = words_left > 0x7FFFFFFFull ? 0u : (uint32_t)((in_len - base) & 3u);
        a.out = out + m.out_off;
        a.isize = m.isize;
        DeviceWave w;
        st = inflate_member(w, S, a);
        if (st == kOk && crc_pow != nullptr) {   // the member's check, by the wavefront that wrote the text
            uint32_t crc = 0;
            if (m.isize) {
                __threadfence_block();           // (the stores of the last rounds: visible to this wave's loads)
                __syncthreads();
                uint32_t *const crc_tab = S.lit; // (the decoder is done with its tables: the CRC's 256 entries take their room)
                for (uint32_t k = threadIdx.x; k < 256u; k += 64u) {
                    uint32_t c = k;
#pragma unroll
                    for (int b = 0; b < 8; ++b) c = (c & 1u) ? (c >> 1) ^ 0xEDB88320u : (c >> 1);
                    crc_tab[k] = c;
                }
                __syncthreads();
                wave_crc_and_lines(out + m.out_off, m.isize, crc_tab, crc_pow, &crc, &nl);
            }
            if (crc != m.crc) st = kErrCrc;
        }
    }
    if (threadIdx.x == 0) { status[j] = st; if (lines) lines[j] = nl; }
}

// CRC-32 of every member's text (RFC 1952 8.) and its number of newlines, one workgroup of 256 lanes per member.  The
// text is brought into LDS in whole, coalesced dwords (from the 4-byte boundary at or before its first byte), cut into 256
// slices of kSlice bytes laid kSlice / 4 + 1 dwords apart (lane k's dword j sits in bank 3 k + j: no conflicts); lane k
// takes slice k byte by byte, and the slices' values -- each multiplied by x^(8 * bytes behind it) -- XOR to the member's CRC
// (bgzf_deflate.hpp: crc_gf_mul; the powers come from a table made once per handle).
__global__ __launch_bounds__(256) void member_check_kernel(const fqtk_inflate_member *members, uint32_t n, const uint8_t *out,
                                                           uint32_t *status, uint32_t *lines, const uint32_t *crc_pow, uint32_t *crc_out) {
    extern __shared__ uint32_t lds[];
    uint32_t *tab = lds, *part = lds + 256, *cnt = lds + 260, *text = lds + 264;
    const uint32_t j = blockIdx.x, lane = threadIdx.x;
    if (j >= n) return;
    {
        uint32_t c = lane;
#pragma unroll
        for (int k = 0; k < 8; ++k) c = (c & 1u) ? (c >> 1) ^ 0xEDB88320u : (c >> 1);
        tab[lane] = c;
    }
    const fqtk_inflate_member m = members[j];
    const uint32_t st = status[j];
    uint32_t mine = 0, nl = 0;
    const bool live = st == kOk && m.isize != 0u;
    const uint32_t a = (uint32_t)(m.out_off & 3u), end = a + m.isize;   // the text is bytes [a, end) of the aligned stream
    if (live) {
        const uint32_t *src = reinterpret_cast<const uint32_t *>(out + (m.out_off & ~(uint64_t)3));
        const uint32_t words = (end + 3u) / 4u;
        for (uint32_t i = lane; i < words; i += 256u) text[(i / kSliceWords) * kSliceStride + i % kSliceWords] = src[i];
    }
    __syncthreads();
    if (live) {
        const uint32_t lo0 = lane * kSlice, lo = lo0 < a ? a : lo0;
        if (lo < end && lo0 + kSlice > a) {
            const uint32_t hi = lo0 + kSlice < end ? lo0 + kSlice : end;
            const uint32_t *row = text + lane * kSliceStride;
            uint32_t c = 0xFFFFFFFFu;
            for (uint32_t q = lo; q < hi; ++q) {
                const uint32_t b = (row[(q - lo0) >> 2] >> (8u * (q & 3u))) & 0xFFu;
                nl += b == 0x0Au ? 1u : 0u;
                c = tab[(c ^ b) & 0xFFu] ^ (c >> 8);
            }
            c = ~c;
            if (hi < end) {   // whole slices between this one and the last, then the last one's bytes
                const uint32_t last = (end - 1u) / kSlice;
                c = bgzf::crc_gf_mul(c, crc_pow[last - 1u - lane]);
                c = bgzf::crc_gf_mul(c, crc_pow[256u + (end - last * kSlice)]);
            }
            mine = c;
        }
    }
    for (int d = 32; d >= 1; d >>= 1) { mine ^= __shfl_xor(mine, d); nl += __shfl_xor(nl, d); }
    if ((lane & 63u) == 0u) { part[lane >> 6] = mine; cnt[lane >> 6] = nl; }
    __syncthreads();
    if (lane == 0) {
        const uint32_t crc = part[0] ^ part[1] ^ part[2] ^ part[3];
        lines[j] = cnt[0] + cnt[1] + cnt[2] + cnt[3];
        if (crc_out) crc_out[j] = crc;   // (pieces of a stream: the caller folds them into the member's CRC)
        else if (st == kOk && crc != m.crc) status[j] = kErrCrc;
    }
}
constexpr size_t kCheckLds = (264 + 256 * kSliceStride) * sizeof(uint32_t);

// ---- a serial gzip stream in chunks (stream mode of inflate_member; host/parallel_gunzip.hpp is the CPU form of the same plan) -----
// One wavefront per chunk: from the chunk's block boundary to the first block boundary at or behind the next chunk's, 16-bit symbols out.
// (plan != nullptr: the chunks were cut on the device, stream_plan_kernel says how many there are)
__global__ __launch_bounds__(64) FQTK_INFLATE_OCCUPANCY void stream_kernel(const uint8_t *in, uint64_t in_len, const StreamChunk *chunks, uint32_t n, const StreamPlan *plan, uint16_t *sym,
                                                    StreamChunkEnd *ends) {
    __shared__ Shared S;
    __shared__ StreamEnd end;
    const uint32_t j = blockIdx.x;
    if (j >= (plan ? plan->n_chunks : n)) return;
    const StreamChunk c = chunks[j];
    if (threadIdx.x == 0) { end.n_sym = 0; end.end_bit = 0; end.final_block = 0; end.n_blocks = 0; end.flags = 0; }
    __syncthreads();
    const uint64_t base_word = c.start_bit >> 5;
    uint32_t st = kErrTruncated;
    if (base_word * 4u < in_len) {
        MemberArgs a;
        a.in_words = reinterpret_cast<const uint32_t *>(in) + base_word;
        a.first_bit = (uint32_t)(c.start_bit & 31u);
        const uint64_t bytes_left = in_len - base_word * 4u;
        const uint64_t bits_left = bytes_left * 8u;
        a.payload_bits = (uint32_t)(bits_left > 0xFFFFFF00ull ? 0xFFFFFF00ull : bits_left) - a.first_bit;
        a.readable_words = (uint32_t)(bytes_left / 4u > 0x7FFFFFFFull ? 0x7FFFFFFFull : bytes_left / 4u);
        a.tail_bytes = (uint32_t)(bytes_left & 3u);
        a.out = nullptr;
        a.isize = c.cap;
        a.out_sym = sym + c.sym_off;
        const uint64_t stop_rel = c.stop_bit - base_word * 32u;
        a.stop_bit = c.stop_bit == ~0ull || stop_rel > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)stop_rel;
        DeviceWave w;
        st = inflate_member<DeviceWave, true>(w, S, a, &end);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        StreamChunkEnd e;
        e.status = st;
        e.final_block = end.final_block;
        e.n_sym = end.n_sym;
        e.n_blocks = end.n_blocks;
        e.flags = end.flags;
        e.pad = 0;
        e.end_bit = base_word * 32u + end.end_bit;
        ends[j] = e;
    }
}

// Where the chunks of a stretch can start: slot k >= 1 is the first block start in bits [k * chunk_bits, (k + 1) * chunk_bits) of the stretch
// (~0: none), found by a wavefront of its own (find_block_start: a lane per bit position); slot 0 is the bit the caller knows.
__global__ __launch_bounds__(64) void stream_search_kernel(const uint8_t *in, uint64_t in_len, uint64_t first_bit, uint32_t chunk_bits, uint32_t n_slots, uint32_t low_literals_only, uint64_t *starts) {
    __shared__ Shared S;
    __shared__ uint8_t kraft_lut[kKraftLutBytes];
    const uint32_t k = blockIdx.x;
    if (k >= n_slots) return;
    if (k == 0u) { if (threadIdx.x == 0) starts[0] = first_bit; return; }
    const uint64_t from = (uint64_t)k * chunk_bits, limit = from + chunk_bits;
    uint64_t found = ~0ull;
    const uint64_t total_bits = in_len * 8u;
    if (from > first_bit && from < total_bits && total_bits < 0xFFFFFF00ull) {
        MemberArgs a;
        a.in_words = reinterpret_cast<const uint32_t *>(in);
        a.first_bit = 0;
        a.payload_bits = (uint32_t)total_bits;
        a.readable_words = (uint32_t)(in_len / 4u);
        a.tail_bytes = (uint32_t)(in_len & 3u);
        a.out = nullptr; a.isize = 0; a.out_sym = nullptr; a.stop_bit = 0;
        DeviceWave w;
        build_kraft_lut(w, kraft_lut);
        const uint32_t r = find_block_start(w, S, a, (uint32_t)from, (uint32_t)(limit < total_bits ? limit : total_bits), low_literals_only != 0u, kraft_lut);
        if (r != 0xFFFFFFFFu) found = r;
    }
    if (threadIdx.x == 0) starts[k] = found;
}

// The chunks of the stretch from the slots' finds (one wavefront, 64 slots per lane): chunk j runs from the j-th start found to the
// (j + 1)-th; the last start found only ends the chunk before it (it opens the next stretch) unless the stretch reaches the end of the
// stream (to_end), when the last chunk runs to the final block.  Nothing found: one chunk that runs as far as the bytes go.  Every
// chunk gets room for sym_per_byte symbols per compressed byte + slack, as far as the symbol buffer reaches (a chunk left with less
// reports FQTK_INFLATE_ERR_OUTPUT at the block where it runs out).
__global__ __launch_bounds__(64) void stream_plan_kernel(const uint64_t *starts, uint32_t n_slots, uint64_t in_len, uint32_t to_end, uint32_t sym_per_byte, uint32_t slack,
                                                         uint64_t sym_cap, StreamChunk *chunks, StreamPlan *plan) {
    __shared__ uint64_t found[kMaxStreamSlots];
    const uint32_t lane = threadIdx.x, per = (n_slots + 63u) / 64u;
    const uint32_t lo = lane * per, hi = lo + per < n_slots ? lo + per : n_slots;
    uint32_t cnt = 0;
    for (uint32_t k = lo; k < hi; ++k) cnt += starts[k] != ~0ull ? 1u : 0u;
    uint32_t incl = cnt;
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(incl, d); if ((int)lane >= d) incl += o; }
    const uint32_t n_found = __shfl(incl, 63);
    uint32_t at = incl - cnt;
    for (uint32_t k = lo; k < hi; ++k) { const uint64_t v = starts[k]; if (v != ~0ull) found[at++] = v; }
    __syncthreads();
    const uint32_t n_chunks = to_end || n_found == 1u ? n_found : n_found - 1u;
    // room: a prefix sum of the chunks' wishes, cut off at the buffer's end
    const uint32_t cper = (n_chunks + 63u) / 64u;
    const uint32_t clo = lane * cper, chi = clo + cper < n_chunks ? clo + cper : n_chunks;
    auto wish = [&](uint32_t j) -> uint64_t {
        const uint64_t stop = j + 1u < n_found ? found[j + 1u] : in_len * 8u;
        const uint64_t want = ((stop - found[j]) / 8u + 1u) * sym_per_byte + slack;
        return ((want < 0xFFFFFF00ull ? want : 0xFFFFFF00ull) + 7u) & ~7ull;
    };
    uint64_t mine = 0;
    for (uint32_t j = clo; j < chi; ++j) mine += wish(j);
    uint64_t pre = mine;
    for (int d = 1; d < 64; d <<= 1) { const uint64_t o = __shfl_up(pre, d); if ((int)lane >= d) pre += o; }
    uint64_t off = pre - mine;
    for (uint32_t j = clo; j < chi; ++j) {
        const uint64_t want = wish(j);
        const uint64_t room = off < sym_cap ? sym_cap - off : 0ull;
        StreamChunk c;
        c.start_bit = found[j];
        c.stop_bit = j + 1u < n_found ? found[j + 1u] : ~0ull;
        c.sym_off = off < sym_cap ? off : sym_cap;
        c.cap = (uint32_t)(want < room ? want : room);
        c.pad = 0;
        chunks[j] = c;
        off += want;
    }
    if (lane == 63u) { plan->n_chunks = n_chunks; plan->n_found = n_found; plan->sym_wanted = pre; }
}

__device__ inline uint32_t resolve_symbol(uint32_t s, const uint8_t *window) { return s < 256u ? s : (uint32_t)window[(s - 256u) & (kWindow - 1u)]; }

// The windows down the chain: windows[k] = the 32 KiB of text in front of chunk k (windows[0] is given; windows[n] = behind the last chunk).
// What chunk k does to its window is a MAP of 32 Ki entries -- entry j of the next window is a byte, or entry j' of this one (the chunk's last
// 32 Ki symbols; where the chunk is shorter, the tail of its own window shifted down) -- and maps compose: (g after f)[j] = g[j] if that is a
// byte, else f[g[j] - 256].  So the chain is a prefix "sum" over the chunks: log2(n) rounds in which EVERY chunk composes its map with the
// map of the chunk 2^r before it (Hillis-Steele, two buffers in turn), all 32 Ki entries of all chunks side by side; afterwards map k holds
// chunk 0 .. k in one and is applied to windows[0].  (The first version walked the chunks one after another in one workgroup: 25 us a
// chunk, 27 ms of a 1023-chunk stretch's 55.)
__global__ __launch_bounds__(256) void window_map_init_kernel(const StreamChunk *chunks, const StreamChunkEnd *ends, const uint16_t *sym, uint16_t *maps) {
    const uint32_t k = blockIdx.y;
    const uint16_t *s = sym + chunks[k].sym_off;
    const uint32_t ns = ends[k].n_sym;
    uint16_t *m = maps + (size_t)k * kWindow;
    for (uint32_t j = blockIdx.x * 256u + threadIdx.x; j < kWindow; j += gridDim.x * 256u) {
        const int64_t p = (int64_t)ns - (int64_t)kWindow + (int64_t)j;   // position in chunk k's text of byte j of the next window
        m[j] = p >= 0 ? s[p] : (uint16_t)(256 + (int64_t)kWindow + p);
    }
}
__global__ __launch_bounds__(256) void window_map_round_kernel(const uint16_t *src, uint16_t *dst, uint32_t d) {
    const uint32_t k = blockIdx.y;
    const uint16_t *g = src + (size_t)k * kWindow;
    uint16_t *o = dst + (size_t)k * kWindow;
    if (k < d) {
        for (uint32_t j = blockIdx.x * 256u + threadIdx.x; j < kWindow; j += gridDim.x * 256u) o[j] = g[j];
        return;
    }
    const uint16_t *f = src + (size_t)(k - d) * kWindow;
    for (uint32_t j = blockIdx.x * 256u + threadIdx.x; j < kWindow; j += gridDim.x * 256u) {
        const uint32_t v = g[j];
        o[j] = v < 256u ? (uint16_t)v : f[v - 256u];
    }
}
__global__ __launch_bounds__(256) void window_map_apply_kernel(const uint16_t *maps, uint8_t *windows) {
    const uint32_t k = blockIdx.y;
    const uint16_t *m = maps + (size_t)k * kWindow;
    const uint8_t *w0 = windows;
    uint8_t *w1 = windows + (size_t)(k + 1u) * kWindow;
    for (uint32_t j = blockIdx.x * 256u + threadIdx.x; j < kWindow; j += gridDim.x * 256u) w1[j] = (uint8_t)resolve_symbol(m[j], w0);
}

// Every chunk's symbols to bytes, side by side: text[out_off[k] + i].
__global__ __launch_bounds__(256) void stream_resolve_kernel(const StreamChunk *chunks, const StreamChunkEnd *ends, const uint64_t *out_off, const uint16_t *sym,
                                                             const uint8_t *windows, uint8_t *text) {
    const uint32_t k = blockIdx.y;
    const uint16_t *s = sym + chunks[k].sym_off;
    const uint8_t *w = windows + (size_t)k * kWindow;
    uint8_t *t = text + out_off[k];
    const uint32_t ns = ends[k].n_sym;
    // four symbols per lane and step where the text is aligned (it is, up to the chunk's first bytes)
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < ns; i += gridDim.x * 256u) t[i] = (uint8_t)resolve_symbol(s[i], w);
}

hipError_t stream_decode_launch(hipStream_t stream, const uint8_t *in, uint64_t in_len, const StreamChunk *chunks, uint32_t n, uint16_t *sym, StreamChunkEnd *ends) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(stream_kernel, dim3(n), dim3(64), 0, stream, in, in_len, chunks, n, (const StreamPlan *)nullptr, sym, ends);
    return hipGetLastError();
}
hipError_t stream_scan_launch(hipStream_t stream, const uint8_t *in, uint64_t in_len, uint64_t first_bit, uint32_t chunk_bytes, uint32_t n_slots, bool to_end,
                              bool low_literals_only, uint32_t sym_per_byte, uint32_t slack, uint64_t sym_cap, uint64_t *starts, StreamChunk *chunks, StreamPlan *plan, uint16_t *sym,
                              StreamChunkEnd *ends) {
    if (n_slots == 0 || n_slots > kMaxStreamSlots) return hipErrorInvalidValue;
    hipLaunchKernelGGL(stream_search_kernel, dim3(n_slots), dim3(64), 0, stream, in, in_len, first_bit, chunk_bytes * 8u, n_slots, low_literals_only ? 1u : 0u, starts);
    hipLaunchKernelGGL(stream_plan_kernel, dim3(1), dim3(64), 0, stream, (const uint64_t *)starts, n_slots, in_len, to_end ? 1u : 0u, sym_per_byte, slack, sym_cap, chunks, plan);
    hipLaunchKernelGGL(stream_kernel, dim3(n_slots), dim3(64), 0, stream, in, in_len, (const StreamChunk *)chunks, n_slots, (const StreamPlan *)plan, sym, ends);
    return hipGetLastError();
}
hipError_t stream_resolve_launch(hipStream_t stream, const StreamChunk *chunks, const StreamChunkEnd *ends, uint32_t n, const uint64_t *out_off, const uint16_t *sym,
                                 uint8_t *windows, uint16_t *maps, uint8_t *text) {
    if (n == 0) return hipSuccess;
    uint16_t *buf[2] = {maps, maps + (size_t)n * kWindow};
    hipLaunchKernelGGL(window_map_init_kernel, dim3(16, n), dim3(256), 0, stream, chunks, ends, sym, buf[0]);
    int cur = 0;
    for (uint32_t d = 1; d < n; d <<= 1) {
        hipLaunchKernelGGL(window_map_round_kernel, dim3(16, n), dim3(256), 0, stream, (const uint16_t *)buf[cur], buf[cur ^ 1], d);
        cur ^= 1;
    }
    hipLaunchKernelGGL(window_map_apply_kernel, dim3(16, n), dim3(256), 0, stream, (const uint16_t *)buf[cur], windows);
    hipLaunchKernelGGL(stream_resolve_kernel, dim3(64, n), dim3(256), 0, stream, chunks, ends, out_off, sym, (const uint8_t *)windows, text);
    return hipGetLastError();
}
// CRC-32 and newlines of pieces of text that is already in place (members[j].out_off / isize; nothing else of a member is used)
hipError_t pieces_check_launch(hipStream_t stream, const fqtk_inflate_member *pieces, uint32_t n, const uint8_t *text, uint32_t *status_zero, uint32_t *lines, uint32_t *crc,
                               const uint32_t *crc_pow_dev) {
    if (n == 0) return hipSuccess;
    static const hipError_t prepared = hipFuncSetAttribute(reinterpret_cast<const void *>(member_check_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kCheckLds);
    if (prepared != hipSuccess) return prepared;
    hipLaunchKernelGGL(member_check_kernel, dim3(n), dim3(256), kCheckLds, stream, pieces, n, text, status_zero, lines, crc_pow_dev, crc);
    return hipGetLastError();
}

void crc_pow_table(uint32_t *pow) {
    for (uint32_t k = 0; k < 256u; ++k) pow[k] = bgzf::crc_x_pow(8u * kSlice * k);
    for (uint32_t r = 0; r <= kSlice; ++r) pow[256u + r] = bgzf::crc_x_pow(8u * r);
}

hipError_t inflate_launch(hipStream_t stream, const uint8_t *in, uint64_t in_len, const fqtk_inflate_member *members, uint32_t n,
                          uint8_t *out, uint32_t *status, uint32_t *lines, const uint32_t *crc_pow_dev) {
    if (n == 0) return hipSuccess;
    static const hipError_t prepared = hipFuncSetAttribute(reinterpret_cast<const void *>(member_check_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kCheckLds);
    if (prepared != hipSuccess) return prepared;
    // (CRC-32 and newline count by the wavefront that wrote the text: no second kernel, no second pass over the text from HBM)
    hipLaunchKernelGGL(inflate_kernel, dim3(n), dim3(64), 0, stream, in, in_len, members, n, out, status, lines, crc_pow_dev);
    return hipGetLastError();
}

}  // namespace inflate
}  // namespace fqtk

namespace {
thread_local std::string g_inflate_error;
int ifail(int code, const std::string &msg) { g_inflate_error = msg; return code; }
#define INFLATE_TRY(expr)                                                                    \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) return ifail(FQTK_EHIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)
}  // namespace

struct fqtk_inflate {
    int device = 0;
    hipStream_t streams[FQTK_INFLATE_SLOTS] = {};
    uint32_t *d_pow = nullptr;
    bool busy[FQTK_INFLATE_SLOTS] = {};
};

extern "C" {

const char *fqtk_inflate_last_error(void) { return g_inflate_error.c_str(); }

int fqtk_inflate_create(int device, fqtk_inflate **out) {
    if (!out) return ifail(FQTK_EINVAL, "out is NULL");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        (void)hipGetLastError();
        return ifail(FQTK_ENODEV, "no HIP device available (the BGZF decoder has no CPU fallback)");
    }
    if (device < 0 || device >= ndev) return ifail(FQTK_ENODEV, "device index out of range");
    INFLATE_TRY(hipSetDevice(device));
    fqtk_inflate *z = new (std::nothrow) fqtk_inflate();
    if (!z) return ifail(FQTK_ENOMEM, "out of host memory");
    z->device = device;
    uint32_t pow[fqtk::inflate::kCrcPowWords];
    fqtk::inflate::crc_pow_table(pow);
    if (hipMalloc(reinterpret_cast<void **>(&z->d_pow), sizeof pow) != hipSuccess ||
        hipMemcpy(z->d_pow, pow, sizeof pow, hipMemcpyHostToDevice) != hipSuccess) {
        fqtk_inflate_destroy(z);
        return ifail(FQTK_EHIP, "cannot allocate the BGZF decoder's tables");
    }
    for (int s = 0; s < FQTK_INFLATE_SLOTS; ++s)
        if (hipStreamCreateWithFlags(&z->streams[s], hipStreamNonBlocking) != hipSuccess) {
            fqtk_inflate_destroy(z);
            return ifail(FQTK_EHIP, "cannot create the BGZF decoder's streams");
        }
    *out = z;
    return FQTK_OK;
}

void fqtk_inflate_destroy(fqtk_inflate *z) {
    if (!z) return;
    (void)hipSetDevice(z->device);
    for (int s = 0; s < FQTK_INFLATE_SLOTS; ++s)
        if (z->streams[s]) { (void)hipStreamSynchronize(z->streams[s]); (void)hipStreamDestroy(z->streams[s]); }
    if (z->d_pow) (void)hipFree(z->d_pow);
    delete z;
}

int fqtk_inflate_enqueue(fqtk_inflate *z, int slot, const uint8_t *in, uint64_t in_len, const fqtk_inflate_member *members,
                         uint32_t n, uint8_t *out, uint32_t *status, uint32_t *lines) {
    if (!z) return ifail(FQTK_EINVAL, "decoder is NULL");
    if (slot < 0 || slot >= FQTK_INFLATE_SLOTS) return ifail(FQTK_EINVAL, "slot out of range");
    if (z->busy[slot]) return ifail(FQTK_EINVAL, "slot is busy: call fqtk_inflate_wait() first");
    if (n == 0) return FQTK_OK;
    if (!in || !members || !out || !status || !lines) return ifail(FQTK_EINVAL, "in / members / out / status / lines is NULL");
    if ((uintptr_t)in & 3u) return ifail(FQTK_EINVAL, "in must be 4-byte aligned");
    INFLATE_TRY(hipSetDevice(z->device));
    INFLATE_TRY(fqtk::inflate::inflate_launch(z->streams[slot], in, in_len, members, n, out, status, lines, z->d_pow));
    z->busy[slot] = true;
    return FQTK_OK;
}

int fqtk_inflate_wait(fqtk_inflate *z, int slot) {
    if (!z) return ifail(FQTK_EINVAL, "decoder is NULL");
    if (slot < 0 || slot >= FQTK_INFLATE_SLOTS) return ifail(FQTK_EINVAL, "slot out of range");
    if (!z->busy[slot]) return FQTK_OK;
    INFLATE_TRY(hipSetDevice(z->device));
    INFLATE_TRY(hipStreamSynchronize(z->streams[slot]));
    z->busy[slot] = false;
    return FQTK_OK;
}

}  // extern "C"
