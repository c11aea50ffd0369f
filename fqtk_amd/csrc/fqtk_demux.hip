// fqtk_demux.hip -- C ABI (include/fqtk_demux.h) of the record pipeline: host side of demux_kernels.hip.h.
//
// One chunk = the raw FASTQ text of n templates from every input.  Two HIP streams carry it:
//   stream A   H2D wait -> line index -> records -> barcode rows -> matcher -> placement -> formatting
//   stream B   DEFLATE + CRC of the chunk's blocks -> packing into BGZF members -> status to the host
// B(k) waits for A(k); A(k) waits for B(k - 2) (a file's persistent slabs rotate over three, demux_kernels.hip.h; with
// carry_blocks = 0 every block lives in its chunk's own slabs and A(k) waits for nothing), so
// the compressor -- the longest kernel by far -- works on chunk k while chunk k + 1 is indexed, matched and formatted.
// A copy stream moves the text in, another the packed members out.  Three chunks may be in flight (slots).
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/fqtk_demux.h"
#include "bgzf_deflate.hpp"
#include "bgzf_internal.hpp"
#include "demux_kernels.hip.h"
#include "inflate_internal.hpp"
#include "matcher_internal.hpp"

namespace {
using namespace fqtk::demux;
using fqtk::internal::set_error;

#define DX_TRY(expr)                                                                                   \
    do {                                                                                               \
        hipError_t _e = (expr);                                                                        \
        if (_e != hipSuccess) return set_error(FQTK_EHIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

template <typename T>
struct DevBuf {   // grow-only device buffer
    T *p = nullptr;
    size_t cap = 0;   // elements
    int ensure(size_t want) {
        if (want <= cap) return FQTK_OK;
        if (p) { DX_TRY(hipFree(p)); p = nullptr; cap = 0; }
        const size_t n = want + want / 8 + 64;
        DX_TRY(hipMalloc(reinterpret_cast<void **>(&p), n * sizeof(T)));
        cap = n;
        return FQTK_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};
template <typename T>
struct PinBuf {   // grow-only page-locked host buffer
    T *p = nullptr;
    size_t cap = 0;
    int ensure(size_t want) {
        if (want <= cap) return FQTK_OK;
        if (p) { DX_TRY(hipHostFree(p)); p = nullptr; cap = 0; }
        const size_t n = want + want / 4 + 64;
        static const bool timing = std::getenv("FQTK_TIMING") != nullptr;
        const auto t0 = std::chrono::steady_clock::now();
        DX_TRY(hipHostMalloc(reinterpret_cast<void **>(&p), n * sizeof(T), hipHostMallocDefault));
        if (timing && n * sizeof(T) >= (8u << 20))
            std::fprintf(stderr, "(timing) page-locked buffer of %zu MB: %.1f ms, done at epoch %.3f\n", (n * sizeof(T)) >> 20, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(),
                         std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count());
        cap = n;
        return FQTK_OK;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
};

constexpr int kStageEvents = 8;   // a0 a1 a2 a3 a4 (stream A), b0 b1 b2 (stream B)

struct Slot {
    DevBuf<uint8_t> text[FQTK_DEMUX_MAX_INPUTS];
    DevBuf<uint32_t> tile_cnt[FQTK_DEMUX_MAX_INPUTS], ls[FQTK_DEMUX_MAX_INPUTS];
    DevBuf<RecView> rec[FQTK_DEMUX_MAX_INPUTS];
    DevBuf<uint8_t> skip, obs, slabs, out_slabs, packed;
    DevBuf<uint32_t> bc_len, lens, res, rec_off, tile_tot, chunk_tot, blk_file, out_len, crc;
    DevBuf<TemplatePlan> plans;
    DevBuf<FileChunk> fc;
    DevBuf<fqtk_bgzf_block> desc;
    DevBuf<unsigned long long> pos, file_off;
    ChunkStatus *d_status = nullptr;
    uint32_t *d_next_block = nullptr;      // the compressor's work counter
    ErrorText *d_err_text = nullptr;       // fed chunks: the offending records' headers, saved while the window's arena holds them
    ChunkStatus *h_status = nullptr;       // page-locked
    PinBuf<unsigned long long> h_file_off;
    PinBuf<uint8_t> h_packed;
    hipEvent_t ev_h2d0 = nullptr, ev_h2d1 = nullptr, ev_fmt = nullptr, ev_status = nullptr, ev_d2h0 = nullptr, ev_d2h1 = nullptr;
    hipEvent_t ev[kStageEvents] = {};
    bool busy = false, flush_only = false;
    bool d2h_issued = false;               // fqtk_demuxer_collect_begin has enqueued the copy home of the chunk's members
    uint32_t n = 0, stride = 0;
    uint64_t win_pos[FQTK_DEMUX_MAX_INPUTS] = {};           // fed text: position of text_base in the input's whole text
    uint64_t text_end[FQTK_DEMUX_MAX_INPUTS] = {};          // ... and of the byte behind the chunk's last record
    bool fed = false;
    const uint8_t *text_base[FQTK_DEMUX_MAX_INPUTS] = {};   // where the chunk's record views point (the slot's own copy, or a window of fed text)
    size_t max_blocks = 0;
};


// ---- BGZF inputs inflated on the device (fqtk_demuxer_feed / fqtk_demuxer_submit_fed) ------------------------------------
// The text of an input lives in one of two arenas; members are appended at `tail`.  When the arena is full the part no
// chunk has consumed yet moves to the front of the other arena (after every chunk submitted so far has been formatted:
// their record views point into the arenas) and the feed goes on there.
struct FedMember { uint64_t off; uint32_t isize, lines; uint64_t lines_before, pos; };   // off: in the current arena; pos: in the input's text
struct FedInput {
    std::mutex mu;                      // one feeder thread per input + the submit thread
    DevBuf<uint8_t> arena[2], comp;
    DevBuf<fqtk_inflate_member> d_members;
    DevBuf<uint32_t> d_status, d_lines;
    PinBuf<uint32_t> h_status, h_lines;
    PinBuf<fqtk_inflate_member> h_members;
    hipStream_t stream = nullptr;
    hipEvent_t ev_moved = nullptr;      // behind the copy of the last change of arena (chunks cut afterwards wait for it)
    hipEvent_t ev_reader[2] = {};       // per arena: behind the formatting of the latest chunk that runs on a window of it where it lies (the home's own chunks)
    bool has_reader[2] = {false, false};
    hipEvent_t ev_t0 = nullptr, ev_t1 = nullptr;   // around a feed's kernels
    bool moved = false;
    int cur = 0;
    uint64_t tail = 0;                  // bytes of arena[cur] in use
    std::deque<FedMember> members;      // not yet consumed in full, in file order
    uint64_t lines_total = 0;           // newlines fed so far
    uint64_t lines_consumed = 0;        // lines taken by chunks so far
    uint64_t members_fed = 0;
    uint64_t text_total = 0;            // bytes of text fed so far
    bool ended = false;
    uint32_t pins = 0;                  // windows cut (fqtk_demuxer_fed_cut) and not yet taken by a submit: the text stays where it is meanwhile
    std::condition_variable cv_pins;
    // a serial gzip stream in chunks (fqtk_demuxer_stream_decode / _commit)
    DevBuf<uint16_t> sym;
    DevBuf<uint8_t> windows;
    DevBuf<uint16_t> maps;             // the chunks' window maps (stream_resolve_launch)
    DevBuf<fqtk::inflate::StreamChunk> d_chunks;
    DevBuf<fqtk::inflate::StreamChunkEnd> d_ends;
    DevBuf<unsigned long long> d_out_off;
    DevBuf<uint32_t> d_crc;
    PinBuf<fqtk::inflate::StreamChunk> h_chunks;
    PinBuf<fqtk::inflate::StreamChunkEnd> h_ends;
    PinBuf<unsigned long long> h_out_off;
    PinBuf<uint32_t> h_crc;
    uint8_t *d_last_window = nullptr;   // the 32 KiB of text behind the last committed chunk
    uint32_t stream_chunks = 0;         // chunks of the decode in hand
    // ... with the chunks cut on the device (fqtk_demuxer_stream_scan)
    DevBuf<unsigned long long> d_starts;
    fqtk::inflate::StreamPlan *d_plan = nullptr;
    fqtk::inflate::StreamPlan *h_plan = nullptr;   // page-locked
    uint32_t scans = 0;
};
constexpr uint64_t kFedSlack = 256;     // bytes kept free behind the text (the check kernel and the line index read whole dwords / 16 bytes)

__global__ void k_put_newline(uint8_t *p) { *p = 0x0A; }

// Text from one arena to the other, 16 bytes a lane and step (both addresses 16-byte aligned).  A kernel, not hipMemcpyAsync: the runtime gives a
// device-to-device copy inside ONE device to a DMA engine, which moved the 2 GB a serial gzip input has live at 25-30 GB/s -- 70-80 ms in which
// the input's stream did nothing else, every fifth stretch (FQTK_TIMING, round 6); the CUs move it at the memory's rate.
__global__ __launch_bounds__(256) void k_copy16(uint4 *dst, const uint4 *src, uint64_t n16) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * 256u) dst[i] = src[i];
}

}  // namespace

struct fqtk_demuxer {
    fqtk_matcher *m = nullptr;
    int device = 0, num_cus = 256;
    DevConfig C{};
    uint32_t n_cols = 0, cols = 0;   // output files; columns of the placement matrix
    uint32_t max_chunk = 0;
    bool carry = true;
    int level = 5;
    hipStream_t s_in = nullptr, s_a = nullptr, s_b = nullptr, s_out = nullptr;
    uint8_t *d_persist = nullptr;
    FileState *d_fs = nullptr;
    unsigned long long *d_counts = nullptr;
    uint32_t *d_tok = nullptr;
    Slot slots[FQTK_DEMUX_SLOTS];
    uint64_t chunk_no = 0;              // chunks submitted
    int slot_of_chunk[2] = {-1, -1};    // slots of the last two chunks submitted (A(k) waits for B(k - 2))
    double stage_s[FQTK_DEMUX_STAGES] = {};
    // device-side inflate of BGZF inputs
    FedInput *fed = nullptr;            // n_inputs of them, made by the first feed
    std::mutex fed_mu;                  // one submit_fed at a time (taken BEFORE an input's mutex, never after)
    std::mutex init_mu, stat_mu;
    std::atomic<uint64_t> chunks_submitted{0};
    uint32_t *d_crc_pow = nullptr;
    double inflate_s = 0;
    bool dynamic_blocks = false;        // FQTK_DYNAMIC_BLOCKS=1 (A/B runs): the compressor's workgroups share the blocks out with a counter instead of
                                        // round robin (measured with decoder wavefronts on the CUs: 14.3 vs 16.9 M templates/s steady from gzip inputs, 45.0 vs 47.2 from text)
    bool ranked = false;                // stream priorities in use
    int feed_priority = 0;
};

namespace {

int alloc_slot_fixed(fqtk_demuxer *d, Slot &s) {
    DX_TRY(hipMalloc(reinterpret_cast<void **>(&s.d_status), sizeof(ChunkStatus)));
    DX_TRY(hipMalloc(reinterpret_cast<void **>(&s.d_next_block), sizeof(uint32_t)));
    DX_TRY(hipMalloc(reinterpret_cast<void **>(&s.d_err_text), 2 * FQTK_DEMUX_MAX_INPUTS * sizeof(ErrorText)));
    DX_TRY(hipMemset(s.d_err_text, 0, 2 * FQTK_DEMUX_MAX_INPUTS * sizeof(ErrorText)));
    DX_TRY(hipHostMalloc(reinterpret_cast<void **>(&s.h_status), sizeof(ChunkStatus), hipHostMallocDefault));
    for (hipEvent_t *e : {&s.ev_h2d0, &s.ev_h2d1, &s.ev_fmt, &s.ev_status, &s.ev_d2h0, &s.ev_d2h1}) DX_TRY(hipEventCreate(e));
    for (int k = 0; k < kStageEvents; ++k) DX_TRY(hipEventCreate(&s.ev[k]));
    int rc;
    if ((rc = s.fc.ensure((size_t)d->n_cols + 1)) != FQTK_OK) return rc;
    if ((rc = s.file_off.ensure((size_t)d->n_cols + 1)) != FQTK_OK) return rc;
    if ((rc = s.h_file_off.ensure((size_t)d->n_cols + 1)) != FQTK_OK) return rc;
    // room for a chunk's members to come back into, made NOW, while the device is idle: grown at the slot's first collect -- beside the next chunks' kernels and
    // copies -- a 33 MB page-locking call took 0.1-0.3 s on a box with tired memory and held every other HIP call of the process up meanwhile (it still grows
    // there when a chunk's members are more than this: FQTK_DEMUX_PACKED_MB)
    {
        const char *e = std::getenv("FQTK_DEMUX_PACKED_MB");
        const long mb = e && *e ? std::atol(e) : 32;
        if (mb > 0 && (rc = s.h_packed.ensure((size_t)mb << 20)) != FQTK_OK) return rc;
    }
    if ((rc = s.chunk_tot.ensure(d->cols)) != FQTK_OK) return rc;
    return FQTK_OK;
}

// buffers whose size follows the number of blocks a chunk can produce
int ensure_blocks(Slot &s, size_t max_blocks) {
    int rc;
    if ((rc = s.slabs.ensure(max_blocks * kSlab)) != FQTK_OK) return rc;
    if ((rc = s.out_slabs.ensure(max_blocks * kSlab)) != FQTK_OK) return rc;
    if ((rc = s.packed.ensure(max_blocks * kSlab)) != FQTK_OK) return rc;
    if ((rc = s.desc.ensure(max_blocks)) != FQTK_OK) return rc;
    if ((rc = s.blk_file.ensure(max_blocks)) != FQTK_OK) return rc;
    if ((rc = s.out_len.ensure(max_blocks)) != FQTK_OK) return rc;
    if ((rc = s.crc.ensure(max_blocks)) != FQTK_OK) return rc;
    if ((rc = s.pos.ensure(max_blocks + 1)) != FQTK_OK) return rc;
    s.max_blocks = max_blocks;
    return FQTK_OK;
}

// stream B: compress the chunk's blocks, pack them, bring the status home
int enqueue_compress(fqtk_demuxer *d, Slot &s) {
    DX_TRY(hipEventRecord(s.ev[5], d->s_b));
    DX_TRY(fqtk::bgzf::deflate_launch(d->s_b, (uint32_t)d->num_cus, s.desc.p, &s.d_status->n_blocks, s.out_len.p, s.crc.p, d->d_tok, d->level,
                                      d->dynamic_blocks ? s.d_next_block : nullptr));
    DX_TRY(hipEventRecord(s.ev[6], d->s_b));
    hipLaunchKernelGGL(k_pack_scan, dim3(1), dim3(kPackScanLanes), 0, d->s_b, d->C, s.fc.p, s.out_len.p, s.pos.p, s.file_off.p, s.d_status);
    DX_TRY(hipGetLastError());
    hipLaunchKernelGGL(k_pack_copy, dim3((uint32_t)d->num_cus * 4), dim3(256), 0, d->s_b, s.desc.p, s.out_len.p, s.crc.p, s.pos.p, s.packed.p, s.d_status);
    DX_TRY(hipGetLastError());
    DX_TRY(hipMemcpyAsync(s.h_status, s.d_status, sizeof(ChunkStatus), hipMemcpyDeviceToHost, d->s_b));
    DX_TRY(hipMemcpyAsync(s.h_file_off.p, s.file_off.p, ((size_t)d->n_cols + 1) * sizeof(unsigned long long), hipMemcpyDeviceToHost, d->s_b));
    DX_TRY(hipEventRecord(s.ev[7], d->s_b));
    DX_TRY(hipEventRecord(s.ev_status, d->s_b));
    return FQTK_OK;
}

void add_elapsed(double *acc, hipEvent_t a, hipEvent_t b) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, a, b) == hipSuccess) *acc += ms * 1e-3;
    else (void)hipGetLastError();
}

// The chunk's packed members on their way home (its status is in: total_bytes says how many).
int issue_d2h(fqtk_demuxer *d, Slot &s) {
    const ChunkStatus &st = *s.h_status;
    int rc;
    if ((rc = s.h_packed.ensure((size_t)st.total_bytes + 16)) != FQTK_OK) return rc;
    DX_TRY(hipEventRecord(s.ev_d2h0, d->s_out));
    if (st.total_bytes) DX_TRY(hipMemcpyAsync(s.h_packed.p, s.packed.p, (size_t)st.total_bytes, hipMemcpyDeviceToHost, d->s_out));
    DX_TRY(hipEventRecord(s.ev_d2h1, d->s_out));
    s.d2h_issued = true;
    return FQTK_OK;
}

int fill_result(fqtk_demuxer *d, Slot &s, fqtk_demux_result *res) {
    std::memset(res, 0, sizeof *res);
    DX_TRY(hipEventSynchronize(s.ev_status));
    const ChunkStatus &st = *s.h_status;
    res->n_files = d->n_cols;
    res->n_templates = s.n;
    res->n_skipped = st.n_skipped;
    if (!s.flush_only) {
        add_elapsed(&d->stage_s[0], s.ev_h2d0, s.ev_h2d1);
        for (int k = 0; k < 4; ++k) add_elapsed(&d->stage_s[1 + k], s.ev[k], s.ev[k + 1]);
    }
    add_elapsed(&d->stage_s[5], s.ev[5], s.ev[6]);
    add_elapsed(&d->stage_s[6], s.ev[6], s.ev[7]);
    if (st.matcher_err != kNoError && (st.err_key == kNoError || (st.err_key >> 24) > st.matcher_err ||
                                       ((st.err_key >> 24) == st.matcher_err && ((st.err_key >> 20) & 15u) > 2u))) {
        // the matcher's length error comes first in template order (stage 2: after the records, before the headers)
        res->error = FQTK_DEMUX_ERR_BARCODE_LEN;
        res->error_template = (uint32_t)st.matcher_err;
        fqtk::internal::word_device_length_error(d->m, s.obs.p, s.lens.p, s.stride, s.n, st.matcher_err);
        return FQTK_OK;
    }
    if (st.err_key != kNoError) {
        res->error = (int)(st.err_key & 0xFFu);
        res->error_input = (uint32_t)((st.err_key >> 9) & 0x7FFu);
        res->error_template = (uint32_t)(st.err_key >> 24);
        if (((st.err_key >> 20) & 15u) == 3u) { res->error_detail = (uint32_t)res->error; res->error = FQTK_DEMUX_ERR_HEADER; }
        return FQTK_OK;
    }
    if (st.err_key == kNoError && st.n_blocks > s.max_blocks) return set_error(FQTK_EINVAL, "internal error: more blocks than the chunk's bound");
    int rc;
    if (!s.d2h_issued && (rc = issue_d2h(d, s)) != FQTK_OK) return rc;
    s.d2h_issued = false;
    DX_TRY(hipEventSynchronize(s.ev_d2h1));
    add_elapsed(&d->stage_s[7], s.ev_d2h0, s.ev_d2h1);
    if (s.fed && !s.flush_only) {
        for (uint32_t i = 0; i < d->C.n_inputs; ++i) s.text_end[i] = s.win_pos[i] + st.end_off[i];
        res->text_end = s.text_end;
    }
    res->bytes = s.h_packed.p;
    res->file_off = reinterpret_cast<const uint64_t *>(s.h_file_off.p);
    res->n_blocks = st.n_blocks;
    return FQTK_OK;
}

}  // namespace

extern "C" {

int fqtk_demuxer_create(fqtk_matcher *m, const fqtk_demux_config *cfg, fqtk_demuxer **out) {
    if (!out) return set_error(FQTK_EINVAL, "out is NULL");
    *out = nullptr;
    if (!m || !cfg || !cfg->n_segments || !cfg->segments) return set_error(FQTK_EINVAL, "matcher / configuration is NULL");
    if (cfg->n_inputs == 0 || cfg->n_inputs > FQTK_DEMUX_MAX_INPUTS)
        return set_error(FQTK_EINVAL, "between 1 and " + std::to_string(FQTK_DEMUX_MAX_INPUTS) + " inputs are supported by the GPU record pipeline");
    if (cfg->max_chunk_templates == 0 || cfg->max_chunk_templates > (1u << 22)) return set_error(FQTK_EINVAL, "max_chunk_templates must be 1 .. 4194304");
    fqtk_demuxer *d = new (std::nothrow) fqtk_demuxer();
    if (!d) return set_error(FQTK_ENOMEM, "out of host memory");
    d->m = m;
    d->device = fqtk_matcher_device(m);
    d->carry = cfg->carry_blocks != 0;
    d->level = cfg->compression_level;
    d->max_chunk = cfg->max_chunk_templates;
    DevConfig &C = d->C;
    std::memset(&C, 0, sizeof C);
    C.n_inputs = cfg->n_inputs;
    C.n_samples = fqtk_matcher_n_samples(m);
    C.barcode_len = fqtk_matcher_barcode_len(m);
    C.skip_short = cfg->skip_too_few_bases ? 1u : 0u;
    C.no_carry = d->carry ? 0u : 1u;
    // segments by type in (input, position) order: demux.rs:100-118 filter the combined read set's segments in place
    static const char kKinds[4] = {'T', 'B', 'M', 'C'};
    size_t at = 0;
    uint32_t per_type[4] = {0, 0, 0, 0};
    std::vector<fqtk::fmt::FileSeg> by_type[4];
    for (uint32_t i = 0; i < cfg->n_inputs; ++i) {
        uint32_t min_len = 0;
        for (uint32_t k = 0; k < cfg->n_segments[i]; ++k, ++at) {
            const fqtk_demux_segment &sg = cfg->segments[at];
            min_len += sg.length >= 0 ? (uint32_t)sg.length : 1u;   // demux.rs:298
            if (sg.length < 0 && k + 1 != cfg->n_segments[i]) { delete d; return set_error(FQTK_EINVAL, "only the last segment of a read structure may be '+'"); }
            int ty = -1;
            for (int q = 0; q < 4; ++q) if (sg.kind == kKinds[q]) ty = q;
            if (ty < 0) { if (sg.kind == 'S') continue; delete d; return set_error(FQTK_EINVAL, "unknown segment kind"); }
            by_type[ty].push_back(fqtk::fmt::FileSeg{i, sg.offset, sg.length, ++per_type[ty]});
        }
        C.min_len[i] = min_len;
    }
    if (by_type[1].size() > kMaxSegs || by_type[2].size() > kMaxSegs) { delete d; return set_error(FQTK_EINVAL, "more than 24 sample- or molecular-barcode segments"); }
    for (const auto &b : by_type[1]) {
        C.bseg[C.n_b++] = fqtk::fmt::SegPos{b.input, b.offset, b.length};
        if (b.length >= 0) C.fixed_bc_len += (uint32_t)b.length; else C.variable_bc = 1;
    }
    for (const auto &b : by_type[2]) C.mseg[C.n_m++] = fqtk::fmt::SegPos{b.input, b.offset, b.length};
    if (fqtk::fmt::max_pieces(C.n_b, C.n_m) > (uint32_t)fqtk::fmt::kMaxPieces || fqtk::fmt::record_slots(C.n_b, C.n_m) > kFormatSlots) {
        delete d;
        return set_error(FQTK_EINVAL, "too many barcode segments for one record (sample + molecular: at most 23)");
    }
    for (int ty = 0; ty < 4; ++ty) {
        if (!cfg->want[ty]) continue;
        for (const auto &f : by_type[ty]) {
            if (C.n_files == FQTK_DEMUX_MAX_FILES) { delete d; return set_error(FQTK_EINVAL, "more than " + std::to_string(FQTK_DEMUX_MAX_FILES) + " output files per sample"); }
            C.fseg[C.n_files++] = f;
        }
    }
    // the lengths travel with the rows whenever a row may not be exactly one barcode long (fqtk_match.h: obs_len)
    C.use_lens = (C.variable_bc || C.fixed_bc_len != C.barcode_len || C.skip_short) ? 1u : 0u;
    d->n_cols = (C.n_samples + 1u) * C.n_files;
    d->cols = (C.n_samples + 1u) * (C.n_files + 1u);

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { (void)hipGetLastError(); delete d; return set_error(FQTK_ENODEV, "no HIP device available (the record pipeline has no CPU fallback)"); }
    auto bail = [&](int rc) { fqtk_demuxer_destroy(d); return rc; };
#define DX_OR_BAIL(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) return bail(set_error(FQTK_EHIP, std::string(#expr) + ": " + hipGetErrorString(_e))); } while (0)
    DX_OR_BAIL(hipSetDevice(d->device));
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, d->device) == hipSuccess && prop.multiProcessorCount > 0) d->num_cus = prop.multiProcessorCount;
    DX_OR_BAIL(fqtk::bgzf::deflate_prepare());
    static_assert(format_block_bytes(FQTK_DEMUX_MAX_INPUTS) <= 48 * 1024, "k_format: slot tables and record views of a group per wave fit the default LDS");
    {
        // The chunk's streams rank above the feed streams of the device-side inflate (FQTK_STREAM_PRIORITY=0: all streams equal, as until round 6).
        // Round 4 measured nothing either way (39.4 / 39.1 against 39.9 / 38.8 M templates/s from BGZF inputs: the decoder's wavefronts took all of
        // a CU's LDS then); with the decoder at 5.8 KB a wavefront and the feeders well ahead of the chunks it pays: same box, alternating, 64 M
        // templates: serial gzip 35.5 / 36.0 -> 38.3 / 39.3 (a second box: 35.2-36.1 -> 39.6-40.0), BGZF 59.4 / 61.1 -> 62.0 / 63.0; plain text has no feed
        // streams (profiles/r06_stream_priority_ab.txt).
        int lo = 0, hi = 0;
        const char *e = std::getenv("FQTK_STREAM_PRIORITY");
        const bool ranked = !(e && e[0] == '0') && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && hi < lo;
        { const char *sb = std::getenv("FQTK_DYNAMIC_BLOCKS"); d->dynamic_blocks = sb && sb[0] == '1'; }
        d->feed_priority = ranked ? lo : 0;
        d->ranked = ranked;
        for (hipStream_t *st : {&d->s_in, &d->s_a, &d->s_b, &d->s_out}) {
            if (ranked) DX_OR_BAIL(hipStreamCreateWithPriority(st, hipStreamNonBlocking, hi));
            else DX_OR_BAIL(hipStreamCreateWithFlags(st, hipStreamNonBlocking));
        }
    }
    const size_t persist_bytes = (size_t)std::max<uint32_t>(d->n_cols, 1) * kPersist * kSlab;
    DX_OR_BAIL(hipMalloc(reinterpret_cast<void **>(&d->d_persist), persist_bytes));
    DX_OR_BAIL(hipMalloc(reinterpret_cast<void **>(&d->d_fs), (size_t)std::max<uint32_t>(d->n_cols, 1) * sizeof(FileState)));
    DX_OR_BAIL(hipMemset(d->d_fs, 0, (size_t)std::max<uint32_t>(d->n_cols, 1) * sizeof(FileState)));
    DX_OR_BAIL(hipMalloc(reinterpret_cast<void **>(&d->d_counts), ((size_t)C.n_samples + 1) * sizeof(unsigned long long)));
    DX_OR_BAIL(hipMemset(d->d_counts, 0, ((size_t)C.n_samples + 1) * sizeof(unsigned long long)));
    DX_OR_BAIL(hipMalloc(reinterpret_cast<void **>(&d->d_tok), (size_t)d->num_cus * fqtk::bgzf::deflate_token_bytes_per_group()));
#undef DX_OR_BAIL
    for (Slot &s : d->slots) {
        const int rc = alloc_slot_fixed(d, s);
        if (rc != FQTK_OK) return bail(rc);
    }
    *out = d;
    return FQTK_OK;
}

void fqtk_demuxer_destroy(fqtk_demuxer *d) {
    if (!d) return;
    (void)hipSetDevice(d->device);
    for (hipStream_t st : {d->s_in, d->s_a, d->s_b, d->s_out}) if (st) (void)hipStreamSynchronize(st);
    for (Slot &s : d->slots) {
        for (int i = 0; i < FQTK_DEMUX_MAX_INPUTS; ++i) { s.text[i].release(); s.tile_cnt[i].release(); s.ls[i].release(); s.rec[i].release(); }
        s.skip.release(); s.obs.release(); s.slabs.release(); s.out_slabs.release(); s.packed.release();
        s.bc_len.release(); s.lens.release(); s.res.release(); s.rec_off.release(); s.tile_tot.release(); s.chunk_tot.release();
        s.blk_file.release(); s.out_len.release(); s.crc.release(); s.plans.release(); s.fc.release(); s.desc.release();
        s.pos.release(); s.file_off.release(); s.h_file_off.release(); s.h_packed.release();
        if (s.d_status) (void)hipFree(s.d_status);
        if (s.d_next_block) (void)hipFree(s.d_next_block);
        if (s.d_err_text) (void)hipFree(s.d_err_text);
        if (s.h_status) (void)hipHostFree(s.h_status);
        for (hipEvent_t e : {s.ev_h2d0, s.ev_h2d1, s.ev_fmt, s.ev_status, s.ev_d2h0, s.ev_d2h1}) if (e) (void)hipEventDestroy(e);
        for (hipEvent_t e : s.ev) if (e) (void)hipEventDestroy(e);
    }
    if (d->fed) {
        for (uint32_t i = 0; i < d->C.n_inputs; ++i) {
            FedInput &F = d->fed[i];
            if (F.stream) { (void)hipStreamSynchronize(F.stream); (void)hipStreamDestroy(F.stream); }
            for (hipEvent_t e : {F.ev_moved, F.ev_t0, F.ev_t1, F.ev_reader[0], F.ev_reader[1]}) if (e) (void)hipEventDestroy(e);
            F.arena[0].release(); F.arena[1].release(); F.comp.release(); F.d_members.release(); F.d_status.release(); F.d_lines.release();
            F.h_status.release(); F.h_lines.release(); F.h_members.release();
            F.sym.release(); F.windows.release(); F.maps.release(); F.d_chunks.release(); F.d_ends.release(); F.d_out_off.release(); F.d_crc.release();
            F.h_chunks.release(); F.h_ends.release(); F.h_out_off.release(); F.h_crc.release();
            if (F.d_last_window) (void)hipFree(F.d_last_window);
            F.d_starts.release();
            if (F.d_plan) (void)hipFree(F.d_plan);
            if (F.h_plan) (void)hipHostFree(F.h_plan);
        }
        delete[] d->fed;
    }
    if (d->d_crc_pow) (void)hipFree(d->d_crc_pow);
    if (d->d_persist) (void)hipFree(d->d_persist);
    if (d->d_fs) (void)hipFree(d->d_fs);
    if (d->d_counts) (void)hipFree(d->d_counts);
    if (d->d_tok) (void)hipFree(d->d_tok);
    for (hipStream_t st : {d->s_in, d->s_a, d->s_b, d->s_out}) if (st) (void)hipStreamDestroy(st);
    delete d;
}

uint32_t fqtk_demuxer_files_per_sample(const fqtk_demuxer *d) { return d ? d->C.n_files : 0; }

namespace {
// (a window's length travels as text_len; copy_from: the window lies in another demuxer's arena -- on device copy_dev -- and is copied into the slot's own
//  text buffer first, base being where it lies there)
struct Window { const uint8_t *base; uint32_t lead, first_line; const uint8_t *copy_from; int copy_dev; };
}
// One chunk: `text` (host buffers, copied in) or `win` (text already on the device).
static int submit_common(fqtk_demuxer *d, int slot, const uint8_t *const *text, const uint64_t *text_len, const Window *win, uint32_t n) {
    if (slot < 0 || slot >= FQTK_DEMUX_SLOTS) return set_error(FQTK_EINVAL, "slot out of range");
    Slot &s = d->slots[slot];
    if (s.busy) return set_error(FQTK_EINVAL, "slot is busy: call fqtk_demuxer_collect() first");
    if (n == 0 || n > d->max_chunk) return set_error(FQTK_EINVAL, "n_templates must be 1 .. max_chunk_templates");
    const DevConfig &C = d->C;
    uint64_t sum_text = 0, seg_text = 0;
    for (uint32_t i = 0; i < C.n_inputs; ++i) {
        if ((!win && !text[i]) || text_len[i] == 0 || text_len[i] >= (1ull << 31)) return set_error(FQTK_EINVAL, "an input's text is empty or 2 GiB or more: use smaller chunks");
        sum_text += text_len[i];
    }
    for (uint32_t f = 0; f < C.n_files; ++f) seg_text += text_len[C.fseg[f].input];
    // placement is 32-bit arithmetic per output file (rec_off, tile_tot, chunk_tot, a record's place q): a file's bytes of
    // one chunk -- at most every input's text once more its own segment's, plus the per-record bytes below -- must fit
    {
        uint64_t max_text = 0;
        for (uint32_t i = 0; i < C.n_inputs; ++i) max_text = std::max<uint64_t>(max_text, text_len[i]);
        if (sum_text + max_text + (32ull + C.n_b + C.n_m) * n + kBlock >= (1ull << 32))
            return set_error(FQTK_EINVAL, "the chunk's text adds up to 4 GiB or more per output file: use smaller chunks");
    }
    DX_TRY(hipSetDevice(d->device));
    int rc;
    // bound of the chunk's output: every file gets the first input's header, all barcode segments and its own
    // segment's bases and qualities (all of them parts of the text), plus a few bytes per record
    // (per record: '@', ' ', up to 10 digits, ':', "N:0:" or a tail byte, the separators of the barcode segments, 5 line bytes)
    const uint64_t out_bound = (uint64_t)C.n_files * (sum_text + (32ull + C.n_b + C.n_m) * n) + seg_text;
    const size_t max_blocks = (size_t)(out_bound / kBlock) + (size_t)d->n_cols + 2;
    if ((rc = ensure_blocks(s, max_blocks)) != FQTK_OK) return rc;
    TextSet T;
    std::memset(&T, 0, sizeof T);
    uint32_t max_tiles = 0;
    for (uint32_t i = 0; i < C.n_inputs; ++i) {
        const uint32_t tiles = (uint32_t)((text_len[i] + kLineTile - 1) / kLineTile);
        max_tiles = std::max(max_tiles, tiles);
        const bool own_copy = !win || win[i].copy_from;
        if (own_copy && (rc = s.text[i].ensure((size_t)text_len[i] + 64)) != FQTK_OK) return rc;
        if ((rc = s.tile_cnt[i].ensure(tiles)) != FQTK_OK) return rc;
        if ((rc = s.ls[i].ensure(4 * (size_t)n + 2)) != FQTK_OK) return rc;
        if ((rc = s.rec[i].ensure(n)) != FQTK_OK) return rc;
        T.text[i] = own_copy ? s.text[i].p : win[i].base;
        T.lead[i] = win ? win[i].lead : 0u;
        T.first_line[i] = win ? win[i].first_line : 0u;
        s.text_base[i] = T.text[i];
        T.len[i] = (uint32_t)text_len[i];
        T.tile_cnt[i] = s.tile_cnt[i].p;
        T.ls[i] = s.ls[i].p;
        T.rec[i] = s.rec[i].p;
    }
    T.window = win ? 1u : 0u;
    const uint32_t n_tiles = (n + kTile - 1) / kTile;
    if ((rc = s.skip.ensure(n)) != FQTK_OK) return rc;
    if ((rc = s.bc_len.ensure(n)) != FQTK_OK) return rc;
    if ((rc = s.lens.ensure(n)) != FQTK_OK) return rc;
    if ((rc = s.res.ensure(n)) != FQTK_OK) return rc;
    if ((rc = s.plans.ensure(n)) != FQTK_OK) return rc;
    if ((rc = s.rec_off.ensure((size_t)std::max<uint32_t>(C.n_files, 1) * n)) != FQTK_OK) return rc;
    if ((rc = s.tile_tot.ensure((size_t)n_tiles * d->cols)) != FQTK_OK) return rc;

    // text in (copy stream)
    DX_TRY(hipEventRecord(s.ev_h2d0, d->s_in));
    for (uint32_t i = 0; i < C.n_inputs; ++i) {
        if (!win) DX_TRY(hipMemcpyAsync(s.text[i].p, text[i], (size_t)text_len[i], hipMemcpyHostToDevice, d->s_in));
        else if (win[i].copy_from && win[i].copy_dev == d->device) {   // (two pipelines on one device: the CUs copy -- k_copy16 says why; both ends are 16-byte aligned)
            const uint64_t n16 = (text_len[i] + 15u) / 16u;
            hipLaunchKernelGGL(k_copy16, dim3((uint32_t)std::min<uint64_t>((n16 + 255u) / 256u, (uint64_t)d->num_cus * 8u)), dim3(256), 0, d->s_in,
                               reinterpret_cast<uint4 *>(s.text[i].p), reinterpret_cast<const uint4 *>(win[i].copy_from), n16);
            DX_TRY(hipGetLastError());
        }
        else if (win[i].copy_from) DX_TRY(hipMemcpyPeerAsync(s.text[i].p, d->device, win[i].copy_from, win[i].copy_dev, (size_t)text_len[i], d->s_in));   // over xGMI
    }
    DX_TRY(hipEventRecord(s.ev_h2d1, d->s_in));

    // stream A
    hipStream_t A = d->s_a;
    DX_TRY(hipStreamWaitEvent(A, s.ev_h2d1, 0));
    if (d->carry && d->slot_of_chunk[0] >= 0) DX_TRY(hipStreamWaitEvent(A, d->slots[d->slot_of_chunk[0]].ev_status, 0));   // B(k - 2): the persistent slabs it read
    ChunkStatus init;
    std::memset(&init, 0, sizeof init);
    init.err_key = kNoError;
    init.matcher_err = kNoError;
    *s.h_status = init;
    DX_TRY(hipMemcpyAsync(s.d_status, s.h_status, sizeof init, hipMemcpyHostToDevice, A));
    DX_TRY(hipEventRecord(s.ev[0], A));
    hipLaunchKernelGGL(k_count_lines, dim3(max_tiles, C.n_inputs), dim3(256), 0, A, T);
    DX_TRY(hipGetLastError());
    hipLaunchKernelGGL(k_scan_tiles, dim3(C.n_inputs), dim3(1024), 0, A, T, s.d_status);
    DX_TRY(hipGetLastError());
    hipLaunchKernelGGL(k_line_starts, dim3(max_tiles, C.n_inputs), dim3(256), 0, A, T, 4u * n);
    DX_TRY(hipGetLastError());
    const uint32_t g256 = (n + 255u) / 256u;
    hipLaunchKernelGGL(k_records, dim3(g256), dim3(256), 0, A, T, C, n, s.skip.p, s.bc_len.p, s.d_status);
    DX_TRY(hipGetLastError());
    uint32_t stride = (C.fixed_bc_len + 3u) / 4u * 4u;
    if (C.variable_bc) {   // a '+B' segment: the longest barcode of the chunk decides the row width (one host round trip)
        uint32_t mx = 0;
        DX_TRY(hipMemcpyAsync(&mx, &s.d_status->max_bc_len, sizeof mx, hipMemcpyDeviceToHost, A));
        DX_TRY(hipStreamSynchronize(A));
        stride = (std::max(mx, C.fixed_bc_len) + 3u) / 4u * 4u;
    }
    if (stride == 0) stride = 4;
    s.stride = stride;
    if ((rc = s.obs.ensure((size_t)n * stride + 64)) != FQTK_OK) return rc;
    hipLaunchKernelGGL(k_extract, dim3(g256), dim3(256), 0, A, T, C, n, stride, s.skip.p, s.bc_len.p, s.obs.p, C.use_lens ? s.lens.p : (uint32_t *)nullptr, s.d_status);
    DX_TRY(hipGetLastError());
    DX_TRY(hipEventRecord(s.ev[1], A));
    if ((rc = fqtk_matcher_assign_batch_device(d->m, s.obs.p, stride, C.use_lens ? s.lens.p : nullptr, n, s.res.p, nullptr, A)) != FQTK_OK) return rc;
    if ((rc = fqtk::internal::take_device_error(d->m, A, &s.d_status->matcher_err)) != FQTK_OK) return rc;
    DX_TRY(hipEventRecord(s.ev[2], A));
    DX_TRY(hipMemsetAsync(s.tile_tot.p, 0, (size_t)n_tiles * d->cols * sizeof(uint32_t), A));
    hipLaunchKernelGGL(k_plan_rank, dim3(n_tiles), dim3(kTile), 0, A, T, C, n, s.res.p, s.skip.p, s.plans.p, s.rec_off.p, s.tile_tot.p, s.d_status);
    DX_TRY(hipGetLastError());
    hipLaunchKernelGGL(k_column_scan, dim3((d->cols + 255u) / 256u), dim3(256), 0, A, s.tile_tot.p, n_tiles, d->cols, s.chunk_tot.p);
    DX_TRY(hipGetLastError());
    hipLaunchKernelGGL(k_layout, dim3(1), dim3(1024), 0, A, C, s.chunk_tot.p, d->d_fs, s.fc.p, d->d_counts, d->carry ? 0u : 1u, (uint32_t)max_blocks, s.d_status);
    DX_TRY(hipGetLastError());
    hipLaunchKernelGGL(k_descs, dim3((uint32_t)((max_blocks + 255) / 256)), dim3(256), 0, A, C, s.fc.p, d->d_persist, s.slabs.p, s.out_slabs.p, s.desc.p, s.blk_file.p, s.d_status);
    DX_TRY(hipGetLastError());
    DX_TRY(hipEventRecord(s.ev[3], A));
    hipLaunchKernelGGL(k_format, dim3((n + kFormatGroup * kFormatWaves - 1) / (kFormatGroup * kFormatWaves)), dim3(64 * kFormatWaves), format_block_bytes(C.n_inputs), A, T, C, n, s.res.p, s.skip.p, s.plans.p,
                       s.rec_off.p, s.tile_tot.p, s.fc.p, d->d_persist, s.slabs.p, s.d_status);
    DX_TRY(hipGetLastError());
    DX_TRY(hipEventRecord(s.ev[4], A));
    if (T.window) {   // (before ev_fmt: the arena the window lies in may be reused once that event is reached)
        hipLaunchKernelGGL(k_save_error_text, dim3(2 * C.n_inputs), dim3(64), 0, A, T, C.n_inputs, n, (const ChunkStatus *)s.d_status, s.d_err_text);
        DX_TRY(hipGetLastError());
    }
    DX_TRY(hipEventRecord(s.ev_fmt, A));
    // stream B
    DX_TRY(hipStreamWaitEvent(d->s_b, s.ev_fmt, 0));
    if ((rc = enqueue_compress(d, s)) != FQTK_OK) return rc;
    s.busy = true;
    s.flush_only = false;
    s.fed = false;
    s.n = n;
    d->slot_of_chunk[0] = d->slot_of_chunk[1];
    d->slot_of_chunk[1] = slot;
    ++d->chunk_no;
    d->chunks_submitted.fetch_add(1);
    return FQTK_OK;
}

int fqtk_demuxer_submit(fqtk_demuxer *d, int slot, const uint8_t *const *text, const uint64_t *text_len, uint32_t n) {
    if (!d || !text || !text_len) return set_error(FQTK_EINVAL, "NULL argument");
    return submit_common(d, slot, text, text_len, nullptr, n);
}

// A decoder's stream.  cu_share = k in 1..7: it may use only the CUs whose number is below k modulo 8.  A decoder wavefront lives for
// milliseconds (a BGZF member) to a tenth of a second (a chunk of a serial gzip stream) and is dispatched to whatever CU has
// room, while a DEFLATE workgroup needs a CU's whole LDS: without a fence every CU holds some decoder wave most of the time and the
// compressor waits for one to drain.  FQTK_FEED_CUS=k, an A/B switch: at 64 M templates, one box, gzip inputs went 16.9 -> 20.0-20.3 M
// templates/s steady with 3 or 5 of every 8 CUs and BGZF inputs 39.6 -> 38.5-38.7; as a default for gzip inputs it took the bench
// line's 16 M-template run from 11.3 to 6.5 M templates/s wall on another box: not a default.
static int make_feed_stream(fqtk_demuxer *d, hipStream_t *out, long cu_share) {
    if (cu_share > 0 && cu_share < 8) {
        uint32_t mask[16];
        const uint32_t words = (uint32_t)std::min<int>(16, (d->num_cus + 31) / 32);
        for (uint32_t w = 0; w < words; ++w) {
            mask[w] = 0;
            for (uint32_t b = 0; b < 32; ++b)
                if ((int)(w * 32 + b) < d->num_cus && (long)((w * 32 + b) % 8u) < cu_share) mask[w] |= 1u << b;
        }
        DX_TRY(hipExtStreamCreateWithCUMask(out, words, mask));
    } else if (d->ranked) {
        DX_TRY(hipStreamCreateWithPriority(out, hipStreamNonBlocking, d->feed_priority));
    } else {
        DX_TRY(hipStreamCreateWithFlags(out, hipStreamNonBlocking));
    }
    return FQTK_OK;
}

// The per-input state of fed text: made by the first feed of the run.
static int fed_init(fqtk_demuxer *d) {
    {   // first feed of the run: the per-input state
        std::lock_guard<std::mutex> lk(d->init_mu);
        if (!d->fed) {
            uint32_t pow[fqtk::inflate::kCrcPowWords];
            fqtk::inflate::crc_pow_table(pow);
            DX_TRY(hipMalloc(reinterpret_cast<void **>(&d->d_crc_pow), sizeof pow));
            DX_TRY(hipMemcpy(d->d_crc_pow, pow, sizeof pow, hipMemcpyHostToDevice));
            FedInput *f = new (std::nothrow) FedInput[d->C.n_inputs];
            if (!f) return set_error(FQTK_ENOMEM, "out of host memory");
            for (uint32_t i = 0; i < d->C.n_inputs; ++i) {
                static const long cu_share = [] { const char *e = std::getenv("FQTK_FEED_CUS"); return e && *e ? std::atol(e) : 0; }();
                {
                    const int rc1 = make_feed_stream(d, &f[i].stream, cu_share);
                    if (rc1 != FQTK_OK) return rc1;
                }
                DX_TRY(hipEventCreateWithFlags(&f[i].ev_moved, hipEventDisableTiming));
                for (int k = 0; k < 2; ++k) DX_TRY(hipEventCreateWithFlags(&f[i].ev_reader[k], hipEventDisableTiming));
                DX_TRY(hipEventCreate(&f[i].ev_t0));
                DX_TRY(hipEventCreate(&f[i].ev_t1));
            }
            d->fed = f;
        }
    }
    return FQTK_OK;
}

// Room for text_bytes more text of an input (its mutex held through `lk`): when the arena is full, what chunks have not consumed yet
// moves to the front of the other arena.  A change of arena excludes a submit in progress (a chunk whose window was cut out
// of the old arena but whose kernels were not enqueued yet was not covered by the event waited for): the input's lock is given up, submit_fed's
// taken, the input's again (that order everywhere), and with both held every window ever cut has its ev_fmt behind its arena's ev_reader.
static int fed_make_room(fqtk_demuxer *d, FedInput &F, std::unique_lock<std::mutex> &lk, uint64_t text_bytes, bool tight = false) {
    int rc;
    if (F.tail + text_bytes + kFedSlack <= F.arena[F.cur].cap) return FQTK_OK;
    // (... and windows that were cut -- fqtk_demuxer_fed_cut -- but not yet taken by a submit: they name places in these arenas.  With both locks
    //  held and no window pinned, every window ever cut has either been copied out or has its chunk's ev_fmt behind its arena's ev_reader.)
    static const bool timing = std::getenv("FQTK_TIMING") != nullptr;
    const auto t_begin = std::chrono::steady_clock::now();
    std::unique_lock<std::mutex> glk(d->fed_mu, std::defer_lock);
    for (;;) {
        lk.unlock();
        glk.lock();
        lk.lock();
        if (F.pins == 0) break;
        glk.unlock();
        F.cv_pins.wait(lk, [&] { return F.pins == 0; });
    }
    const auto t_locked = std::chrono::steady_clock::now();
    struct Report { bool on; std::chrono::steady_clock::time_point t0, t1; ~Report() { if (on) { const auto t2 = std::chrono::steady_clock::now();
        if (std::chrono::duration<double, std::milli>(t2 - t0).count() > 10.0) std::fprintf(stderr, "(timing) change of arena: %.1f ms for the locks and the windows in flight, %.1f ms for the rest\n",
            std::chrono::duration<double, std::milli>(t1 - t0).count(), std::chrono::duration<double, std::milli>(t2 - t1).count()); } } } report{timing, t_begin, t_locked};
    // (only this input's feeder moves its tail; chunks may have consumed members meanwhile, which only shrinks what is live)
    const uint64_t live_from = F.members.empty() ? F.tail : F.members.front().off;
    {
        const uint64_t live = F.tail - live_from;
        const int other = 1 - F.cur;
        // (FQTK_FED_ARENA_MIN: tests make the arenas small so that a short run changes arena many times)
        static const uint64_t arena_min = [] { const char *e = std::getenv("FQTK_FED_ARENA_MIN"); return e && *e ? (uint64_t)std::strtoull(e, nullptr, 10) : (1ull << 30); }();
        // (tight: gigabytes of text at a time take an arena that holds one of them and what is live, and change arena every time)
        const uint64_t want = std::max<uint64_t>(tight ? (live + text_bytes) + (live + text_bytes) / 4 + kFedSlack : (live + text_bytes + kFedSlack) * 2, arena_min);
        {
            // Chunks whose record views point into the OTHER arena were cut before this input last changed arenas: its old text may go, and
            // this stream's copy may start, when the LAST of them has been formatted -- that arena's own event (until round 6 the event of the
            // latest chunk submitted anywhere: every change of arena then waited for the three chunks in flight, 30-60 ms in which the feeder
            // decoded nothing -- a quarter of a serial gzip input's time; the chunks that read the other arena finished a whole arena ago).
            if (F.has_reader[other]) DX_TRY(hipStreamWaitEvent(F.stream, F.ev_reader[other], 0));
            // The other arena is taken as it is when it HOLDS what must go in (what is live, the new text, the slack); only one that does not
            // is freed and made again, with room to spare (`want`).  Until round 6 every arena below `want` -- twice the need -- was made again: a
            // serial gzip input whose arenas had been reserved for the need (stream_reserve) freed and allocated one every few stretches, 35-110 ms
            // each time with the whole device waiting.
            const uint64_t need = 16u + live + text_bytes + kFedSlack;
            if (F.arena[other].cap < need) {
                if (F.has_reader[other]) DX_TRY(hipEventSynchronize(F.ev_reader[other]));   // (freeing memory a kernel may still read)
                if ((rc = F.arena[other].ensure((size_t)std::max(want, need))) != FQTK_OK) return rc;
            }
        }
        const uint64_t shift = live_from & 15u;   // members keep their alignment modulo 16 (nothing depends on it; windows are aligned down anyway)
        if (live) {   // (from the 16-byte boundary at or before the first live byte: the bytes in front of it are the arena's own, and nobody's over there)
            const uint64_t n16 = (shift + live + 15u) / 16u;
            hipLaunchKernelGGL(k_copy16, dim3((uint32_t)std::min<uint64_t>((n16 + 255u) / 256u, (uint64_t)d->num_cus * 8u)), dim3(256), 0, F.stream,
                               reinterpret_cast<uint4 *>(F.arena[other].p), reinterpret_cast<const uint4 *>(F.arena[F.cur].p + (live_from - shift)), n16);
            DX_TRY(hipGetLastError());
        }
        DX_TRY(hipEventRecord(F.ev_moved, F.stream));
        F.moved = true;
        for (FedMember &m : F.members) m.off = m.off - live_from + shift;
        F.cur = other;
        F.tail = shift + live;
    }
    return FQTK_OK;
}

// ---- BGZF inputs inflated on the device ---------------------------------------------------------------------------------
int fqtk_demuxer_feed(fqtk_demuxer *d, uint32_t input, const uint8_t *bytes, uint64_t len, const fqtk_inflate_member *members,
                      uint32_t n_members, int last, uint64_t *lines_available) {
    if (!d || (n_members && (!bytes || !members))) return set_error(FQTK_EINVAL, "NULL argument");
    if (input >= d->C.n_inputs) return set_error(FQTK_EINVAL, "input out of range");
    DX_TRY(hipSetDevice(d->device));
    {
        const int rc0 = fed_init(d);
        if (rc0 != FQTK_OK) return rc0;
    }
    FedInput &F = d->fed[input];
    std::unique_lock<std::mutex> lk(F.mu);
    if (F.ended) return set_error(FQTK_EINVAL, "the input's last members have been fed already");
    uint64_t text_bytes = last ? 1 : 0;
    for (uint32_t j = 0; j < n_members; ++j) {
        if (members[j].isize > FQTK_INFLATE_MAX_ISIZE) return set_error(FQTK_EINVAL, "a BGZF member of more than 64 KiB of text");
        text_bytes += members[j].isize;
    }
    int rc;
    if ((rc = fed_make_room(d, F, lk, text_bytes)) != FQTK_OK) return rc;
    // the members' places, the copy in, the kernels, the counts back
    if ((rc = F.h_members.ensure(n_members + 1)) != FQTK_OK) return rc;
    if ((rc = F.d_members.ensure(n_members + 1)) != FQTK_OK) return rc;
    if ((rc = F.d_status.ensure(n_members + 1)) != FQTK_OK) return rc;
    if ((rc = F.d_lines.ensure(n_members + 1)) != FQTK_OK) return rc;
    if ((rc = F.h_status.ensure(n_members + 1)) != FQTK_OK) return rc;
    if ((rc = F.h_lines.ensure(n_members + 1)) != FQTK_OK) return rc;
    if ((rc = F.comp.ensure((size_t)len + 16)) != FQTK_OK) return rc;
    uint64_t at = F.tail;
    for (uint32_t j = 0; j < n_members; ++j) {
        F.h_members.p[j] = members[j];
        F.h_members.p[j].out_off = at;
        at += members[j].isize;
    }
    uint8_t *const arena = F.arena[F.cur].p;
    F.tail = at + (last ? 1 : 0);   // the place is taken; the members count once they are inflated (below)
    // The copy, the kernels and the wait run without the input's lock: chunks go on being cut out of what was fed before.
    lk.unlock();
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (n_members) {
        e0 = F.ev_t0;
        e1 = F.ev_t1;
        DX_TRY(hipMemcpyAsync(F.comp.p, bytes, (size_t)len, hipMemcpyHostToDevice, F.stream));
        DX_TRY(hipMemcpyAsync(F.d_members.p, F.h_members.p, (size_t)n_members * sizeof(fqtk_inflate_member), hipMemcpyHostToDevice, F.stream));
        DX_TRY(hipEventRecord(e0, F.stream));
        DX_TRY(fqtk::inflate::inflate_launch(F.stream, F.comp.p, len, F.d_members.p, n_members, arena, F.d_status.p, F.d_lines.p, d->d_crc_pow));
        DX_TRY(hipEventRecord(e1, F.stream));
        DX_TRY(hipMemcpyAsync(F.h_status.p, F.d_status.p, (size_t)n_members * sizeof(uint32_t), hipMemcpyDeviceToHost, F.stream));
        DX_TRY(hipMemcpyAsync(F.h_lines.p, F.d_lines.p, (size_t)n_members * sizeof(uint32_t), hipMemcpyDeviceToHost, F.stream));
    }
    if (last) {   // the text may end without a newline: one is added (a blank line more, if it does not)
        hipLaunchKernelGGL(k_put_newline, dim3(1), dim3(1), 0, F.stream, arena + at);
        DX_TRY(hipGetLastError());
    }
    DX_TRY(hipStreamSynchronize(F.stream));
    if (e0) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, e0, e1) == hipSuccess) { std::lock_guard<std::mutex> glk(d->stat_mu); d->inflate_s += ms * 1e-3; }
    }
    for (uint32_t j = 0; j < n_members; ++j)
        if (F.h_status.p[j] != 0) {
            static const char *const kWhat[12] = {"", "reserved block type", "stored block length check", "bad code lengths", "over-subscribed or incomplete Huffman code",
                                                  "invalid code", "distance too far back", "more text than the member's ISIZE", "stream runs past the member",
                                                  "less text than the member's ISIZE", "CRC mismatch", "bad member header"};
            return set_error(FQTK_EINVAL, std::string("corrupt BGZF block ") + std::to_string(F.members_fed + j) + ": " + kWhat[std::min<uint32_t>(F.h_status.p[j], 11)]);
        }
    lk.lock();
    for (uint32_t j = 0; j < n_members; ++j) {
        F.members.push_back(FedMember{F.h_members.p[j].out_off, members[j].isize, F.h_lines.p[j], F.lines_total, F.text_total});
        F.lines_total += F.h_lines.p[j];
        F.text_total += members[j].isize;
    }
    F.members_fed += n_members;
    if (last) {
        if (F.members.empty()) F.members.push_back(FedMember{at, 0, 0, F.lines_total, F.text_total});
        F.text_total += 1;
        F.members.back().isize += 1;
        F.members.back().lines += 1;
        F.lines_total += 1;
        F.ended = true;
    }
    if (lines_available) *lines_available = F.lines_total;
    return FQTK_OK;
}

// The next n records of an input's fed text: where they lie.  The window is pinned (F.pins) until a submit has taken it.
int fqtk_demuxer_fed_cut(fqtk_demuxer *home, uint32_t input, uint32_t n, fqtk_fed_window *out) {
    if (!home || !out) return set_error(FQTK_EINVAL, "NULL argument");
    if (!home->fed) return set_error(FQTK_EINVAL, "nothing has been fed");
    if (input >= home->C.n_inputs) return set_error(FQTK_EINVAL, "input out of range");
    if (n == 0 || n > home->max_chunk) return set_error(FQTK_EINVAL, "n_templates must be 1 .. max_chunk_templates");
    FedInput &F = home->fed[input];
    std::lock_guard<std::mutex> lk(F.mu);
    const uint64_t l0 = F.lines_consumed, l1 = l0 + 4ull * n;   // the chunk's lines: [l0, l1); newline l1 - 1 ends the last one
    if (l1 > F.lines_total) return set_error(FQTK_EINVAL, "fewer lines have been fed than the chunk takes");
    // members that end before line l0 begins are done with (newline l0 - 1 is not theirs, nor any later one)
    while (F.members.size() > 1 && l0 > 0 && F.members.front().lines_before + F.members.front().lines <= l0 - 1) F.members.pop_front();
    const FedMember &first = F.members.front();
    size_t k = 0;
    while (F.members[k].lines_before + F.members[k].lines < l1) ++k;   // the member that holds newline l1 - 1
    const FedMember &lastm = F.members[k];
    const uint8_t *p = F.arena[F.cur].p + first.off;
    const uint32_t lead = (uint32_t)(reinterpret_cast<uintptr_t>(p) & 15u);
    out->home = home;
    out->input = input;
    out->lead = lead;
    out->first_line = (uint32_t)(l0 - first.lines_before);
    out->n_templates = n;
    out->base = p - lead;
    out->len = lead + (lastm.off + lastm.isize - first.off);
    out->pos = first.pos - lead;   // (mod 2^64 for the first member: added back with the chunk's end_off)
    out->arena = (uint32_t)F.cur;
    out->reserved = 0;
    F.lines_consumed = l1;
    ++F.pins;
    return FQTK_OK;
}

namespace {
// Device `dev` reads `peer`'s memory directly over xGMI where the platform lets it (otherwise the runtime stages a peer copy through the host).
void allow_peer(int dev, int peer) {
    static std::mutex mu;
    static std::vector<std::pair<int, int>> done;
    std::lock_guard<std::mutex> lk(mu);
    for (const auto &p : done) if (p.first == dev && p.second == peer) return;
    done.emplace_back(dev, peer);
    int can = 0;
    if (hipDeviceCanAccessPeer(&can, dev, peer) == hipSuccess && can) (void)hipDeviceEnablePeerAccess(peer, 0);
    (void)hipGetLastError();   // (already enabled: fine)
}
void unpin(const fqtk_fed_window &w) {
    FedInput &F = w.home->fed[w.input];
    {
        std::lock_guard<std::mutex> lk(F.mu);
        if (F.pins) --F.pins;
    }
    F.cv_pins.notify_all();
}
}  // namespace

int fqtk_demuxer_submit_windows(fqtk_demuxer *d, int slot, const fqtk_fed_window *w, uint32_t n) {
    if (!d || !w) return set_error(FQTK_EINVAL, "NULL argument");
    if (slot < 0 || slot >= FQTK_DEMUX_SLOTS) return set_error(FQTK_EINVAL, "slot out of range");
    const uint32_t n_inputs = d->C.n_inputs;
    for (uint32_t i = 0; i < n_inputs; ++i) {
        if (!w[i].home || !w[i].home->fed || w[i].input != i || w[i].n_templates != n || w[i].home->C.n_inputs != n_inputs)
            return set_error(FQTK_EINVAL, "window i must be a cut of input i, of n_templates records, out of a demuxer of the same configuration");
    }
    DX_TRY(hipSetDevice(d->device));
    if (d->slots[slot].busy) {
        for (uint32_t i = 0; i < n_inputs; ++i) unpin(w[i]);
        return set_error(FQTK_EINVAL, "slot is busy: call fqtk_demuxer_collect() first");
    }
    for (uint32_t i = 0; i < n_inputs; ++i) if (w[i].home->device != d->device) allow_peer(d->device, w[i].home->device);
    Window win[FQTK_DEMUX_MAX_INPUTS];
    uint64_t text_len[FQTK_DEMUX_MAX_INPUTS];
    int rc;
    {
        // (a chunk run where some of its text lives: no arena of d's may change between here and the chunk's ev_fmt being recorded)
        std::lock_guard<std::mutex> glk(d->fed_mu);
        for (uint32_t i = 0; i < n_inputs; ++i) {
            FedInput &F = w[i].home->fed[i];
            const bool local = w[i].home == d;
            {
                std::lock_guard<std::mutex> lk(F.mu);
                // the text may just have changed arena: behind that copy (another device's stream: its event is waited for here)
                if (F.moved && local) DX_TRY(hipStreamWaitEvent(d->s_a, F.ev_moved, 0));
                else if (F.moved) DX_TRY(hipEventSynchronize(F.ev_moved));
            }
            win[i].lead = w[i].lead;
            win[i].first_line = w[i].first_line;
            win[i].base = w[i].base;
            win[i].copy_from = local ? nullptr : w[i].base;
            win[i].copy_dev = w[i].home->device;
            text_len[i] = w[i].len;
            d->slots[slot].win_pos[i] = w[i].pos;
        }
        rc = submit_common(d, slot, nullptr, text_len, win, n);
        if (rc == FQTK_OK) {
            d->slots[slot].fed = true;
            for (uint32_t i = 0; i < n_inputs; ++i) {   // behind the chunk's formatting on stream A: the arena's latest reader
                if (w[i].home != d) continue;
                FedInput &F = d->fed[i];
                const uint32_t ar = w[i].arena & 1u;
                DX_TRY(hipEventRecord(F.ev_reader[ar], d->s_a));
                F.has_reader[ar] = true;
            }
        }
    }
    // windows of d's own text are covered by their arena's ev_reader from here on; the others once their copies are done
    bool remote = false;
    for (uint32_t i = 0; i < n_inputs; ++i) remote = remote || w[i].home != d;
    hipError_t e = hipSuccess;
    if (rc == FQTK_OK && remote) e = hipEventSynchronize(d->slots[slot].ev_h2d1);
    for (uint32_t i = 0; i < n_inputs; ++i) unpin(w[i]);
    if (rc != FQTK_OK) return rc;
    if (e != hipSuccess) return set_error(FQTK_EHIP, std::string("copy of a chunk's text between devices: ") + hipGetErrorString(e));
    return FQTK_OK;
}

int fqtk_demuxer_submit_fed(fqtk_demuxer *d, int slot, uint32_t n) {
    if (!d) return set_error(FQTK_EINVAL, "NULL argument");
    if (!d->fed) return set_error(FQTK_EINVAL, "nothing has been fed");
    if (slot < 0 || slot >= FQTK_DEMUX_SLOTS) return set_error(FQTK_EINVAL, "slot out of range");
    if (d->slots[slot].busy) return set_error(FQTK_EINVAL, "slot is busy: call fqtk_demuxer_collect() first");
    if (n == 0 || n > d->max_chunk) return set_error(FQTK_EINVAL, "n_templates must be 1 .. max_chunk_templates");
    for (uint32_t i = 0; i < d->C.n_inputs; ++i) {   // (nothing is cut unless every input has the lines)
        std::lock_guard<std::mutex> lk(d->fed[i].mu);
        if (d->fed[i].lines_consumed + 4ull * n > d->fed[i].lines_total) return set_error(FQTK_EINVAL, "fewer lines have been fed than the chunk takes");
    }
    fqtk_fed_window w[FQTK_DEMUX_MAX_INPUTS];
    for (uint32_t i = 0; i < d->C.n_inputs; ++i) {
        const int rc = fqtk_demuxer_fed_cut(d, i, n, &w[i]);
        if (rc != FQTK_OK) { for (uint32_t q = 0; q < i; ++q) unpin(w[q]); return rc; }
    }
    return fqtk_demuxer_submit_windows(d, slot, w, n);
}

// ---- serial gzip inputs (one member per file: gzip, bcl2fastq) decoded on the device in chunks ------------------------------
int fqtk_demuxer_stream_decode(fqtk_demuxer *d, uint32_t input, const uint8_t *bytes, uint64_t len, const fqtk_stream_chunk *chunks, uint32_t n,
                               fqtk_stream_end *ends) {
    if (!d || !bytes || !chunks || !ends || n == 0) return set_error(FQTK_EINVAL, "NULL argument / no chunks");
    if (input >= d->C.n_inputs) return set_error(FQTK_EINVAL, "input out of range");
    if (len >= (1ull << 29)) return set_error(FQTK_EINVAL, "a stretch of 512 MiB or more: use shorter stretches");
    DX_TRY(hipSetDevice(d->device));
    int rc;
    if ((rc = fed_init(d)) != FQTK_OK) return rc;
    FedInput &F = d->fed[input];
    if ((rc = F.comp.ensure((size_t)len + 16)) != FQTK_OK) return rc;
    if ((rc = F.h_chunks.ensure(n)) != FQTK_OK) return rc;
    if ((rc = F.d_chunks.ensure(n)) != FQTK_OK) return rc;
    if ((rc = F.h_ends.ensure(n)) != FQTK_OK) return rc;
    if ((rc = F.d_ends.ensure(n)) != FQTK_OK) return rc;
    // room for every chunk's symbols: FASTQ text deflates 3-7 : 1; eight times the chunk's compressed bytes (+ a block's worth) is
    // given, and a chunk that needs more reports FQTK_INFLATE_ERR_OUTPUT (the caller then cuts the stretch shorter)
    uint64_t sym_total = 0;
    for (uint32_t k = 0; k < n; ++k) {
        const uint64_t stop = chunks[k].stop_bit == ~0ull ? len * 8u : chunks[k].stop_bit;
        if (chunks[k].start_bit > len * 8u || stop < chunks[k].start_bit) return set_error(FQTK_EINVAL, "a chunk outside the stretch");
        const uint64_t cbytes = (stop - chunks[k].start_bit) / 8u + 65536u;
        const uint64_t cap = std::min<uint64_t>(cbytes * 8u + 262144u, 0xFFFFFF00ull);
        F.h_chunks.p[k] = fqtk::inflate::StreamChunk{chunks[k].start_bit, chunks[k].stop_bit, sym_total, (uint32_t)cap, 0u};
        sym_total += (cap + 7u) & ~7ull;
    }
    if ((rc = F.sym.ensure((size_t)sym_total + 64)) != FQTK_OK) return rc;
    DX_TRY(hipMemcpyAsync(F.comp.p, bytes, (size_t)len, hipMemcpyHostToDevice, F.stream));
    DX_TRY(hipMemcpyAsync(F.d_chunks.p, F.h_chunks.p, (size_t)n * sizeof(fqtk::inflate::StreamChunk), hipMemcpyHostToDevice, F.stream));
    DX_TRY(hipEventRecord(F.ev_t0, F.stream));
    DX_TRY(fqtk::inflate::stream_decode_launch(F.stream, F.comp.p, len, F.d_chunks.p, n, F.sym.p, F.d_ends.p));
    DX_TRY(hipEventRecord(F.ev_t1, F.stream));
    DX_TRY(hipMemcpyAsync(F.h_ends.p, F.d_ends.p, (size_t)n * sizeof(fqtk::inflate::StreamChunkEnd), hipMemcpyDeviceToHost, F.stream));
    DX_TRY(hipStreamSynchronize(F.stream));
    {
        float ms = 0;
        if (hipEventElapsedTime(&ms, F.ev_t0, F.ev_t1) == hipSuccess) { std::lock_guard<std::mutex> glk(d->stat_mu); d->inflate_s += ms * 1e-3; }
    }
    for (uint32_t k = 0; k < n; ++k) {
        ends[k].status = F.h_ends.p[k].status;
        ends[k].final_block = F.h_ends.p[k].final_block;
        ends[k].n_bytes = F.h_ends.p[k].n_sym;
        ends[k].end_bit = F.h_ends.p[k].end_bit;
        ends[k].start_bit = chunks[k].start_bit;
        ends[k].n_blocks = F.h_ends.p[k].n_blocks;
        ends[k].flags = F.h_ends.p[k].flags;
    }
    F.stream_chunks = n;
    return FQTK_OK;
}

// Room for the stretches to come, made once: growing a device buffer later means hipFree, which waits for the whole device (measured in a
// run whose stretches ramped 16 -> 32 -> 64 MB: 130-150 ms per growth of the symbol buffer, 40-70 ms per growth of an arena).
static uint64_t stream_sym_cap(uint64_t len, uint32_t n_slots, uint32_t sym_per_byte) {
    return (len + 1u) * sym_per_byte + (uint64_t)n_slots * (65536u + 8u * (uint64_t)sym_per_byte + 8u);
}
int fqtk_demuxer_stream_reserve(fqtk_demuxer *d, uint32_t input, uint64_t max_len, uint32_t max_slots, uint32_t sym_per_byte, uint64_t arena_bytes) {
    using namespace fqtk::inflate;
    if (!d || input >= d->C.n_inputs) return set_error(FQTK_EINVAL, "NULL argument / input out of range");
    if (max_len >= (1ull << 29) || max_slots == 0 || max_slots > kMaxStreamSlots || sym_per_byte == 0 || sym_per_byte > 2048u) return set_error(FQTK_EINVAL, "stretch geometry out of range");
    DX_TRY(hipSetDevice(d->device));
    int rc;
    if ((rc = fed_init(d)) != FQTK_OK) return rc;
    FedInput &F = d->fed[input];
    static const bool timing = std::getenv("FQTK_TIMING") != nullptr;
    const auto t_begin = std::chrono::steady_clock::now();
    struct Report { bool on; uint32_t input; std::chrono::steady_clock::time_point t0; uint64_t arena;
                    ~Report() { if (on) std::fprintf(stderr, "(timing) stream reserve of input %u (arenas 2 x %llu MB): %.1f ms\n", input, (unsigned long long)(arena >> 20),
                                                     std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count()); } } report{timing, input, t_begin, arena_bytes};
    std::lock_guard<std::mutex> lk(F.mu);
    if ((rc = F.comp.ensure((size_t)max_len + 16)) != FQTK_OK) return rc;
    if ((rc = F.sym.ensure((size_t)stream_sym_cap(max_len, max_slots, sym_per_byte) + 64)) != FQTK_OK) return rc;
    if ((rc = F.d_chunks.ensure(max_slots)) != FQTK_OK) return rc;
    if ((rc = F.h_chunks.ensure(max_slots)) != FQTK_OK) return rc;
    if ((rc = F.h_ends.ensure(max_slots)) != FQTK_OK) return rc;
    if ((rc = F.d_ends.ensure(max_slots)) != FQTK_OK) return rc;
    if ((rc = F.d_starts.ensure(max_slots)) != FQTK_OK) return rc;
    if ((rc = F.h_out_off.ensure(max_slots + 1)) != FQTK_OK) return rc;
    if ((rc = F.d_out_off.ensure(max_slots + 1)) != FQTK_OK) return rc;
    if ((rc = F.windows.ensure((size_t)(max_slots + 1) * kStreamWindow)) != FQTK_OK) return rc;
    if ((rc = F.maps.ensure((size_t)max_slots * 2u * kStreamWindow)) != FQTK_OK) return rc;
    if (arena_bytes) {
        if (F.tail != 0 || !F.members.empty()) return set_error(FQTK_EINVAL, "arenas are reserved before anything is fed");
        for (int k = 0; k < 2; ++k)
            if ((rc = F.arena[k].ensure((size_t)arena_bytes)) != FQTK_OK) return rc;
    }
    return FQTK_OK;
}

// The same with the chunks cut on the device (fqtk_inflate.hip: stream_search_kernel, stream_plan_kernel).
int fqtk_demuxer_stream_scan(fqtk_demuxer *d, uint32_t input, const uint8_t *bytes, uint64_t len, uint64_t first_bit, uint32_t chunk_bytes, uint32_t n_slots, int to_end,
                             uint32_t sym_per_byte, uint32_t flags, fqtk_stream_end *ends, uint32_t *n_chunks) {
    using namespace fqtk::inflate;
    if (!d || !bytes || !ends || !n_chunks) return set_error(FQTK_EINVAL, "NULL argument");
    if (input >= d->C.n_inputs) return set_error(FQTK_EINVAL, "input out of range");
    if (len >= (1ull << 29)) return set_error(FQTK_EINVAL, "a stretch of 512 MiB or more: use shorter stretches");
    if (n_slots == 0 || n_slots > kMaxStreamSlots) return set_error(FQTK_EINVAL, "n_slots must be 1 .. 4096");
    if (chunk_bytes < 4096u || (chunk_bytes & 3u) || first_bit >= len * 8u) return set_error(FQTK_EINVAL, "chunk_bytes must be a multiple of 4 of 4096 at least; first_bit inside the stretch");
    if (sym_per_byte == 0 || sym_per_byte > 2048u) return set_error(FQTK_EINVAL, "sym_per_byte must be 1 .. 2048");
    DX_TRY(hipSetDevice(d->device));
    int rc;
    if ((rc = fed_init(d)) != FQTK_OK) return rc;
    FedInput &F = d->fed[input];
    static const bool timing = std::getenv("FQTK_TIMING") != nullptr;
    const auto t_begin = std::chrono::steady_clock::now();
    if ((rc = F.comp.ensure((size_t)len + 16)) != FQTK_OK) return rc;
    if ((rc = F.d_chunks.ensure(n_slots)) != FQTK_OK) return rc;
    if ((rc = F.h_chunks.ensure(n_slots)) != FQTK_OK) return rc;
    if ((rc = F.h_ends.ensure(n_slots)) != FQTK_OK) return rc;
    if ((rc = F.d_ends.ensure(n_slots)) != FQTK_OK) return rc;
    if ((rc = F.d_starts.ensure(n_slots)) != FQTK_OK) return rc;
    if (!F.d_plan) {
        DX_TRY(hipMalloc(reinterpret_cast<void **>(&F.d_plan), sizeof(StreamPlan)));
        DX_TRY(hipHostMalloc(reinterpret_cast<void **>(&F.h_plan), sizeof(StreamPlan), hipHostMallocDefault));
    }
    // room for the symbols: sym_per_byte per compressed byte of the stretch and a block's worth per chunk (the plan kernel shares it out
    // and never hands out more than there is)
    const uint32_t slack = 65536u;
    // (the 128 KiB the caller adds behind the last slot's bytes only let the last chunk run to its block's end: they get the base room, not sym_per_byte times it --
    //  at 2048 symbols per byte that tail alone asked for half a gigabyte)
    const uint64_t body = std::min<uint64_t>(len, (uint64_t)n_slots * chunk_bytes);
    const uint64_t sym_cap = stream_sym_cap(body, n_slots, sym_per_byte) + (len - body) * std::min<uint32_t>(sym_per_byte, 8u);
    if ((rc = F.sym.ensure((size_t)sym_cap + 64)) != FQTK_OK) return rc;
    const auto t_alloc = std::chrono::steady_clock::now();
    DX_TRY(hipMemcpyAsync(F.comp.p, bytes, (size_t)len, hipMemcpyHostToDevice, F.stream));
    DX_TRY(hipEventRecord(F.ev_t0, F.stream));
    DX_TRY(stream_scan_launch(F.stream, F.comp.p, len, first_bit, chunk_bytes, n_slots, to_end != 0, (flags & FQTK_STREAM_SCAN_TEXT) != 0, sym_per_byte, slack, sym_cap, reinterpret_cast<uint64_t *>(F.d_starts.p),
                              F.d_chunks.p, F.d_plan, F.sym.p, F.d_ends.p));
    DX_TRY(hipEventRecord(F.ev_t1, F.stream));
    DX_TRY(hipMemcpyAsync(F.h_plan, F.d_plan, sizeof(StreamPlan), hipMemcpyDeviceToHost, F.stream));
    DX_TRY(hipMemcpyAsync(F.h_chunks.p, F.d_chunks.p, (size_t)n_slots * sizeof(StreamChunk), hipMemcpyDeviceToHost, F.stream));
    DX_TRY(hipMemcpyAsync(F.h_ends.p, F.d_ends.p, (size_t)n_slots * sizeof(StreamChunkEnd), hipMemcpyDeviceToHost, F.stream));
    DX_TRY(hipStreamSynchronize(F.stream));
    {
        float ms = 0;
        if (hipEventElapsedTime(&ms, F.ev_t0, F.ev_t1) == hipSuccess) { std::lock_guard<std::mutex> glk(d->stat_mu); d->inflate_s += ms * 1e-3; }
        if (timing && F.scans++ < 3) {
            const auto t_end = std::chrono::steady_clock::now();
            std::fprintf(stderr, "(timing) stream scan %u of input %u: buffers %.1f ms, copy in + kernels + results %.1f ms (the kernels: %.1f ms)\n", F.scans, input,
                         std::chrono::duration<double, std::milli>(t_alloc - t_begin).count(), std::chrono::duration<double, std::milli>(t_end - t_alloc).count(), ms);
        }
    }
    const uint32_t n = F.h_plan->n_chunks;
    if (n == 0 || n > n_slots) return set_error(FQTK_EHIP, "the stretch's plan came back empty");
    for (uint32_t k = 0; k < n; ++k) {
        ends[k].status = F.h_ends.p[k].status;
        ends[k].final_block = F.h_ends.p[k].final_block;
        ends[k].n_bytes = F.h_ends.p[k].n_sym;
        ends[k].end_bit = F.h_ends.p[k].end_bit;
        ends[k].start_bit = F.h_chunks.p[k].start_bit;
        ends[k].n_blocks = F.h_ends.p[k].n_blocks;
        ends[k].flags = F.h_ends.p[k].flags;
    }
    *n_chunks = n;
    F.stream_chunks = n;
    return FQTK_OK;
}

int fqtk_demuxer_stream_commit(fqtk_demuxer *d, uint32_t input, uint32_t n_accept, int member_start, int last, uint64_t *lines_fed, uint32_t *crc32,
                               uint64_t *n_text) {
    if (!d || !d->fed || input >= d->C.n_inputs) return set_error(FQTK_EINVAL, "nothing decoded / input out of range");
    DX_TRY(hipSetDevice(d->device));
    FedInput &F = d->fed[input];
    if (n_accept > F.stream_chunks) return set_error(FQTK_EINVAL, "more chunks accepted than decoded");
    std::unique_lock<std::mutex> lk(F.mu);
    if (F.ended) return set_error(FQTK_EINVAL, "the input has ended already");
    int rc;
    uint64_t total = 0;
    if ((rc = F.h_out_off.ensure(n_accept + 1)) != FQTK_OK) return rc;
    if ((rc = F.d_out_off.ensure(n_accept + 1)) != FQTK_OK) return rc;
    for (uint32_t k = 0; k < n_accept; ++k) total += F.h_ends.p[k].n_sym;
    const uint64_t text_bytes = total + (last ? 1 : 0);
    if ((rc = fed_make_room(d, F, lk, text_bytes, text_bytes > (1ull << 30))) != FQTK_OK) return rc;
    const uint64_t at = F.tail;
    {
        uint64_t o = at;
        for (uint32_t k = 0; k < n_accept; ++k) { F.h_out_off.p[k] = o; o += F.h_ends.p[k].n_sym; }
    }
    const uint32_t n_pieces = (uint32_t)((total + 65535u) / 65536u);
    if ((rc = F.h_members.ensure(n_pieces + 1)) != FQTK_OK) return rc;
    if ((rc = F.d_members.ensure(n_pieces + 1)) != FQTK_OK) return rc;
    if ((rc = F.d_status.ensure(n_pieces + 1)) != FQTK_OK) return rc;
    if ((rc = F.d_lines.ensure(n_pieces + 1)) != FQTK_OK) return rc;
    if ((rc = F.h_lines.ensure(n_pieces + 1)) != FQTK_OK) return rc;
    if ((rc = F.d_crc.ensure(n_pieces + 1)) != FQTK_OK) return rc;
    if ((rc = F.h_crc.ensure(n_pieces + 1)) != FQTK_OK) return rc;
    if ((rc = F.windows.ensure((size_t)(n_accept + 1) * fqtk::inflate::kStreamWindow)) != FQTK_OK) return rc;
    if ((rc = F.maps.ensure((size_t)n_accept * 2u * fqtk::inflate::kStreamWindow)) != FQTK_OK) return rc;
    if (!F.d_last_window) {
        DX_TRY(hipMalloc(reinterpret_cast<void **>(&F.d_last_window), fqtk::inflate::kStreamWindow));
        DX_TRY(hipMemsetAsync(F.d_last_window, 0, fqtk::inflate::kStreamWindow, F.stream));
    }
    for (uint32_t p = 0; p < n_pieces; ++p) {
        fqtk_inflate_member m;
        std::memset(&m, 0, sizeof m);
        m.out_off = at + (uint64_t)p * 65536u;
        m.isize = (uint32_t)std::min<uint64_t>(65536u, total - (uint64_t)p * 65536u);
        F.h_members.p[p] = m;
    }
    uint8_t *const arena = F.arena[F.cur].p;
    F.tail = at + text_bytes;
    lk.unlock();
    if (n_accept) {
        // the window in front of the first chunk: none at the start of a member, else what the last commit left
        if (member_start) DX_TRY(hipMemsetAsync(F.windows.p, 0, fqtk::inflate::kStreamWindow, F.stream));
        else DX_TRY(hipMemcpyAsync(F.windows.p, F.d_last_window, fqtk::inflate::kStreamWindow, hipMemcpyDeviceToDevice, F.stream));
        DX_TRY(hipMemcpyAsync(F.d_out_off.p, F.h_out_off.p, (size_t)n_accept * sizeof(unsigned long long), hipMemcpyHostToDevice, F.stream));
        // (window n_accept -- the text behind the last accepted chunk -- comes out of the same chain: the next commit starts from it)
        DX_TRY(fqtk::inflate::stream_resolve_launch(F.stream, F.d_chunks.p, F.d_ends.p, n_accept, reinterpret_cast<const uint64_t *>(F.d_out_off.p), F.sym.p, F.windows.p, F.maps.p, arena));
        DX_TRY(hipMemcpyAsync(F.d_last_window, F.windows.p + (size_t)n_accept * fqtk::inflate::kStreamWindow, fqtk::inflate::kStreamWindow, hipMemcpyDeviceToDevice, F.stream));
    }
    if (n_pieces) {
        DX_TRY(hipMemcpyAsync(F.d_members.p, F.h_members.p, (size_t)n_pieces * sizeof(fqtk_inflate_member), hipMemcpyHostToDevice, F.stream));
        DX_TRY(hipMemsetAsync(F.d_status.p, 0, (size_t)n_pieces * sizeof(uint32_t), F.stream));
        DX_TRY(fqtk::inflate::pieces_check_launch(F.stream, F.d_members.p, n_pieces, arena, F.d_status.p, F.d_lines.p, F.d_crc.p, d->d_crc_pow));
        DX_TRY(hipMemcpyAsync(F.h_lines.p, F.d_lines.p, (size_t)n_pieces * sizeof(uint32_t), hipMemcpyDeviceToHost, F.stream));
        DX_TRY(hipMemcpyAsync(F.h_crc.p, F.d_crc.p, (size_t)n_pieces * sizeof(uint32_t), hipMemcpyDeviceToHost, F.stream));
    }
    if (last) {
        hipLaunchKernelGGL(k_put_newline, dim3(1), dim3(1), 0, F.stream, arena + at + total);
        DX_TRY(hipGetLastError());
    }
    DX_TRY(hipStreamSynchronize(F.stream));
    // CRC-32 of the committed text: the pieces' values folded (crc(A || B) = crc(A) * x^(8 |B|) + crc(B), bgzf_deflate.hpp)
    uint32_t crc = 0;
    {
        static const uint32_t pow_full = fqtk::bgzf::crc_x_pow(8u * 65536u);
        for (uint32_t p = 0; p < n_pieces; ++p) {
            const uint32_t len_p = F.h_members.p[p].isize;
            crc = fqtk::bgzf::crc_gf_mul(crc, len_p == 65536u ? pow_full : fqtk::bgzf::crc_x_pow(8u * len_p)) ^ F.h_crc.p[p];
        }
    }
    lk.lock();
    for (uint32_t p = 0; p < n_pieces; ++p) {
        F.members.push_back(FedMember{F.h_members.p[p].out_off, F.h_members.p[p].isize, F.h_lines.p[p], F.lines_total, F.text_total});
        F.lines_total += F.h_lines.p[p];
        F.text_total += F.h_members.p[p].isize;
    }
    F.members_fed += n_pieces;
    if (last) {
        if (F.members.empty()) F.members.push_back(FedMember{at + total, 0, 0, F.lines_total, F.text_total});
        F.text_total += 1;
        F.members.back().isize += 1;
        F.members.back().lines += 1;
        F.lines_total += 1;
        F.ended = true;
    }
    F.stream_chunks = 0;
    if (lines_fed) *lines_fed = F.lines_total;
    if (crc32) *crc32 = crc;
    if (n_text) *n_text = total;
    return FQTK_OK;
}

// The 32 KiB of text in front of the next chunk of a stream (zeros where the member has less): what a decoder that takes over needs.
int fqtk_demuxer_stream_window(fqtk_demuxer *d, uint32_t input, uint8_t *window) {
    if (!d || !window || input >= d->C.n_inputs) return set_error(FQTK_EINVAL, "NULL argument / input out of range");
    DX_TRY(hipSetDevice(d->device));
    int rc;
    if ((rc = fed_init(d)) != FQTK_OK) return rc;
    FedInput &F = d->fed[input];
    if (!F.d_last_window) { std::memset(window, 0, fqtk::inflate::kStreamWindow); return FQTK_OK; }
    DX_TRY(hipMemcpyAsync(window, F.d_last_window, fqtk::inflate::kStreamWindow, hipMemcpyDeviceToHost, F.stream));
    DX_TRY(hipStreamSynchronize(F.stream));
    return FQTK_OK;
}

// Text of a stream that was decoded elsewhere (the host's sequential decoder, where the chunks could not be cut or did not decode) joins the
// input's fed text; `window_after` (32 KiB, NULL with `last`) is the text in front of whatever follows.
int fqtk_demuxer_stream_commit_text(fqtk_demuxer *d, uint32_t input, const uint8_t *text, uint64_t n, const uint8_t *window_after, int last, uint64_t *lines_fed,
                                    uint32_t *crc32) {
    if (!d || input >= d->C.n_inputs || (n && !text)) return set_error(FQTK_EINVAL, "NULL argument / input out of range");
    if (n >= (1ull << 32)) return set_error(FQTK_EINVAL, "4 GiB of text or more in one call");
    DX_TRY(hipSetDevice(d->device));
    int rc;
    if ((rc = fed_init(d)) != FQTK_OK) return rc;
    FedInput &F = d->fed[input];
    std::unique_lock<std::mutex> lk(F.mu);
    if (F.ended) return set_error(FQTK_EINVAL, "the input has ended already");
    const uint64_t text_bytes = n + (last ? 1 : 0);
    if ((rc = fed_make_room(d, F, lk, text_bytes)) != FQTK_OK) return rc;
    const uint64_t at = F.tail;
    const uint32_t n_pieces = (uint32_t)((n + 65535u) / 65536u);
    if ((rc = F.h_members.ensure(n_pieces + 1)) != FQTK_OK) return rc;
    if ((rc = F.d_members.ensure(n_pieces + 1)) != FQTK_OK) return rc;
    if ((rc = F.d_status.ensure(n_pieces + 1)) != FQTK_OK) return rc;
    if ((rc = F.d_lines.ensure(n_pieces + 1)) != FQTK_OK) return rc;
    if ((rc = F.h_lines.ensure(n_pieces + 1)) != FQTK_OK) return rc;
    if ((rc = F.d_crc.ensure(n_pieces + 1)) != FQTK_OK) return rc;
    if ((rc = F.h_crc.ensure(n_pieces + 1)) != FQTK_OK) return rc;
    if (!F.d_last_window) {
        DX_TRY(hipMalloc(reinterpret_cast<void **>(&F.d_last_window), fqtk::inflate::kStreamWindow));
        DX_TRY(hipMemsetAsync(F.d_last_window, 0, fqtk::inflate::kStreamWindow, F.stream));
    }
    for (uint32_t p = 0; p < n_pieces; ++p) {
        fqtk_inflate_member m;
        std::memset(&m, 0, sizeof m);
        m.out_off = at + (uint64_t)p * 65536u;
        m.isize = (uint32_t)std::min<uint64_t>(65536u, n - (uint64_t)p * 65536u);
        F.h_members.p[p] = m;
    }
    uint8_t *const arena = F.arena[F.cur].p;
    F.tail = at + text_bytes;
    lk.unlock();
    if (n) DX_TRY(hipMemcpyAsync(arena + at, text, (size_t)n, hipMemcpyHostToDevice, F.stream));
    if (window_after) DX_TRY(hipMemcpyAsync(F.d_last_window, window_after, fqtk::inflate::kStreamWindow, hipMemcpyHostToDevice, F.stream));
    if (n_pieces) {
        DX_TRY(hipMemcpyAsync(F.d_members.p, F.h_members.p, (size_t)n_pieces * sizeof(fqtk_inflate_member), hipMemcpyHostToDevice, F.stream));
        DX_TRY(hipMemsetAsync(F.d_status.p, 0, (size_t)n_pieces * sizeof(uint32_t), F.stream));
        DX_TRY(fqtk::inflate::pieces_check_launch(F.stream, F.d_members.p, n_pieces, arena, F.d_status.p, F.d_lines.p, F.d_crc.p, d->d_crc_pow));
        DX_TRY(hipMemcpyAsync(F.h_lines.p, F.d_lines.p, (size_t)n_pieces * sizeof(uint32_t), hipMemcpyDeviceToHost, F.stream));
        DX_TRY(hipMemcpyAsync(F.h_crc.p, F.d_crc.p, (size_t)n_pieces * sizeof(uint32_t), hipMemcpyDeviceToHost, F.stream));
    }
    if (last) {
        hipLaunchKernelGGL(k_put_newline, dim3(1), dim3(1), 0, F.stream, arena + at + n);
        DX_TRY(hipGetLastError());
    }
    DX_TRY(hipStreamSynchronize(F.stream));
    uint32_t crc = 0;
    {
        static const uint32_t pow_full = fqtk::bgzf::crc_x_pow(8u * 65536u);
        for (uint32_t p = 0; p < n_pieces; ++p) {
            const uint32_t len_p = F.h_members.p[p].isize;
            crc = fqtk::bgzf::crc_gf_mul(crc, len_p == 65536u ? pow_full : fqtk::bgzf::crc_x_pow(8u * len_p)) ^ F.h_crc.p[p];
        }
    }
    lk.lock();
    for (uint32_t p = 0; p < n_pieces; ++p) {
        F.members.push_back(FedMember{F.h_members.p[p].out_off, F.h_members.p[p].isize, F.h_lines.p[p], F.lines_total, F.text_total});
        F.lines_total += F.h_lines.p[p];
        F.text_total += F.h_members.p[p].isize;
    }
    F.members_fed += n_pieces;
    if (last) {
        if (F.members.empty()) F.members.push_back(FedMember{at + n, 0, 0, F.lines_total, F.text_total});
        F.text_total += 1;
        F.members.back().isize += 1;
        F.members.back().lines += 1;
        F.lines_total += 1;
        F.ended = true;
    }
    if (lines_fed) *lines_fed = F.lines_total;
    if (crc32) *crc32 = crc;
    return FQTK_OK;
}

int fqtk_demuxer_fed_tail(fqtk_demuxer *d, uint32_t input, uint64_t pos, uint8_t *buf, size_t cap, uint64_t *n_bytes) {
    if (!d || !d->fed || input >= d->C.n_inputs || !buf || !n_bytes) return set_error(FQTK_EINVAL, "nothing has been fed / bad argument");
    DX_TRY(hipSetDevice(d->device));
    FedInput &F = d->fed[input];
    std::lock_guard<std::mutex> lk(F.mu);
    if (pos == ~0ull) pos = F.text_total > cap ? F.text_total - cap : 0;   // (the last `cap` bytes of the text)
    *n_bytes = pos < F.text_total ? F.text_total - pos : 0;
    if (*n_bytes == 0) return FQTK_OK;
    if (F.members.empty() || pos < F.members.front().pos) return set_error(FQTK_EINVAL, "that text has been consumed");
    const uint64_t off = F.members.front().off + (pos - F.members.front().pos);   // live members are contiguous in the arena
    DX_TRY(hipMemcpy(buf, F.arena[F.cur].p + off, (size_t)std::min<uint64_t>(*n_bytes, cap), hipMemcpyDeviceToHost));
    return FQTK_OK;
}

int fqtk_demuxer_inflate_seconds(fqtk_demuxer *d, double *seconds) {
    if (!d || !seconds) return set_error(FQTK_EINVAL, "NULL argument");
    std::lock_guard<std::mutex> glk(d->stat_mu);
    *seconds = d->inflate_s;
    return FQTK_OK;
}

// First half of a collect, for callers that overlap it with the second half of the chunk before: waits for the chunk's kernels and status and
// enqueues the copy home of its members, without waiting for the copy.
int fqtk_demuxer_collect_begin(fqtk_demuxer *d, int slot) {
    if (!d) return set_error(FQTK_EINVAL, "NULL argument");
    if (slot < 0 || slot >= FQTK_DEMUX_SLOTS) return set_error(FQTK_EINVAL, "slot out of range");
    Slot &s = d->slots[slot];
    if (!s.busy) return set_error(FQTK_EINVAL, "nothing was submitted on this slot");
    if (s.d2h_issued) return FQTK_OK;
    DX_TRY(hipSetDevice(d->device));
    DX_TRY(hipEventSynchronize(s.ev_status));
    const ChunkStatus &st = *s.h_status;
    // (a chunk that failed, or one whose status makes no sense, brings nothing home: fqtk_demuxer_collect says what happened)
    if (st.err_key != kNoError || st.matcher_err != kNoError || st.n_blocks > s.max_blocks) return FQTK_OK;
    return issue_d2h(d, s);
}

int fqtk_demuxer_collect(fqtk_demuxer *d, int slot, fqtk_demux_result *res) {
    if (!d || !res) return set_error(FQTK_EINVAL, "NULL argument");
    if (slot < 0 || slot >= FQTK_DEMUX_SLOTS) return set_error(FQTK_EINVAL, "slot out of range");
    Slot &s = d->slots[slot];
    if (!s.busy) return set_error(FQTK_EINVAL, "nothing was submitted on this slot");
    DX_TRY(hipSetDevice(d->device));
    const int rc = fill_result(d, s, res);
    s.busy = false;
    return rc;
}

int fqtk_demuxer_text_done(fqtk_demuxer *d, int slot) {
    if (!d) return set_error(FQTK_EINVAL, "NULL argument");
    if (slot < 0 || slot >= FQTK_DEMUX_SLOTS) return set_error(FQTK_EINVAL, "slot out of range");
    DX_TRY(hipSetDevice(d->device));
    DX_TRY(hipEventSynchronize(d->slots[slot].ev_h2d1));
    return FQTK_OK;
}

int fqtk_demuxer_record_text(fqtk_demuxer *d, int slot, uint32_t input, uint32_t t, char *header, size_t cap, uint32_t *n_bases) {
    if (!d || !header || cap == 0) return set_error(FQTK_EINVAL, "NULL argument");
    if (slot < 0 || slot >= FQTK_DEMUX_SLOTS || input >= d->C.n_inputs) return set_error(FQTK_EINVAL, "slot / input out of range");
    Slot &s = d->slots[slot];
    if (t >= s.n) return set_error(FQTK_EINVAL, "template index out of range");
    DX_TRY(hipSetDevice(d->device));
    if (s.fed) {   // the window's arena may be gone: what the chunk saved of its offending records
        ErrorText e[2];
        for (int c = 0; c < 2; ++c) DX_TRY(hipMemcpy(&e[c], s.d_err_text + (size_t)c * d->C.n_inputs + input, sizeof(ErrorText), hipMemcpyDeviceToHost));
        for (int c = 0; c < 2; ++c)
            if (e[c].valid && e[c].t == t) {
                const size_t n = std::min<size_t>(e[c].head_len, cap - 1);
                std::memcpy(header, e[c].head, n);
                header[n] = 0;
                if (n_bases) *n_bases = e[c].seq_len;
                return FQTK_OK;
            }
        return set_error(FQTK_EINVAL, "a chunk of fed text keeps the text of the templates its errors name only");
    }
    RecView r;
    DX_TRY(hipMemcpy(&r, s.rec[input].p + t, sizeof r, hipMemcpyDeviceToHost));
    const size_t n = std::min<size_t>(r.head_len, cap - 1);
    if (n) DX_TRY(hipMemcpy(header, s.text_base[input] + r.head_off, n, hipMemcpyDeviceToHost));
    header[n] = 0;
    if (n_bases) *n_bases = r.seq_len;
    return FQTK_OK;
}

int fqtk_demuxer_flush(fqtk_demuxer *d, fqtk_demux_result *res) {
    if (!d || !res) return set_error(FQTK_EINVAL, "NULL argument");
    for (Slot &s : d->slots) if (s.busy) return set_error(FQTK_EINVAL, "collect every chunk before the flush");
    DX_TRY(hipSetDevice(d->device));
    for (hipStream_t st : {d->s_a, d->s_b}) DX_TRY(hipStreamSynchronize(st));
    Slot &s = d->slots[0];
    int rc;
    const size_t max_blocks = (size_t)d->n_cols + 2;
    if (s.max_blocks < max_blocks && (rc = ensure_blocks(s, max_blocks)) != FQTK_OK) return rc;
    hipStream_t A = d->s_a;
    ChunkStatus init;
    std::memset(&init, 0, sizeof init);
    init.err_key = kNoError;
    init.matcher_err = kNoError;
    *s.h_status = init;
    DX_TRY(hipMemcpyAsync(s.d_status, s.h_status, sizeof init, hipMemcpyHostToDevice, A));
    hipLaunchKernelGGL(k_layout, dim3(1), dim3(1024), 0, A, d->C, (const uint32_t *)nullptr, d->d_fs, s.fc.p, d->d_counts, 1u, (uint32_t)s.max_blocks, s.d_status);
    DX_TRY(hipGetLastError());
    hipLaunchKernelGGL(k_descs, dim3((uint32_t)((max_blocks + 255) / 256)), dim3(256), 0, A, d->C, s.fc.p, d->d_persist, s.slabs.p, s.out_slabs.p, s.desc.p, s.blk_file.p, s.d_status);
    DX_TRY(hipGetLastError());
    DX_TRY(hipEventRecord(s.ev_fmt, A));
    DX_TRY(hipStreamWaitEvent(d->s_b, s.ev_fmt, 0));
    if ((rc = enqueue_compress(d, s)) != FQTK_OK) return rc;
    s.flush_only = true;
    s.n = 0;
    rc = fill_result(d, s, res);
    s.flush_only = false;
    return rc;
}

int fqtk_demuxer_counts(fqtk_demuxer *d, uint64_t *counts) {
    if (!d || !counts) return set_error(FQTK_EINVAL, "NULL argument");
    DX_TRY(hipSetDevice(d->device));
    DX_TRY(hipStreamSynchronize(d->s_a));
    std::vector<unsigned long long> h((size_t)d->C.n_samples + 1);
    DX_TRY(hipMemcpy(h.data(), d->d_counts, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < h.size(); ++i) counts[i] += (uint64_t)h[i];
    return FQTK_OK;
}

int fqtk_demuxer_stage_seconds(fqtk_demuxer *d, double *seconds) {
    if (!d || !seconds) return set_error(FQTK_EINVAL, "NULL argument");
    for (int k = 0; k < FQTK_DEMUX_STAGES; ++k) seconds[k] = d->stage_s[k];
    return FQTK_OK;
}

const char *fqtk_demuxer_stage_name(int stage) {
    static const char *kNames[FQTK_DEMUX_STAGES] = {"h2d", "index", "match", "place", "format", "deflate", "pack", "d2h"};
    return stage >= 0 && stage < FQTK_DEMUX_STAGES ? kNames[stage] : "";
}

}  // extern "C"
