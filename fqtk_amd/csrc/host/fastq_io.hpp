// fastq_io.hpp -- gz-aware FASTQ reading in record batches.
//
// SURVEY.md section 8(f) row 3 (input side).  The reference reads through fgoxide's Io::new_reader
// (gz-aware, 1 MiB buffer; /root/reference/src/bin/commands/demux.rs:844-849) and seq_io's
// fastq::Reader (demux.rs:16-17,289-294,891): four-line records, `head` = line 1 without '@'.
// Three byte sources behind one interface, chosen by looking at the file (not at its name):
//   plain text      a regular file is memory-mapped and parsed where it lies (records are offsets into the
//                   mapping: no copy at all); anything else (a pipe) is read(2) into piece buffers
//   gzip            zlib inflate (gzread: single- and multi-member streams); one stream cannot be split
//   BGZF            independent <= 64 KiB members ('BC' extra field): a group of blocks is inflated IN PARALLEL
//                   by a few helper threads (libdeflate through dlopen, zlib's inflate if it is absent)
// In every case a producer thread runs ahead of the parser (bounded queue of 4 MiB pieces), so decompression
// and record parsing overlap -- the reference gets the same from fgoxide's read-ahead (demux.rs:928-934).
// Unpinned by the reference's tests (choices here): a trailing '\r' is stripped from every line;
// multi-line FASTQ is not supported (seq_io's fastq reader does not support it either).
#pragma once
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <cerrno>
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <string_view>
#include <thread>
#include <vector>

#include "env.hpp"

#include "fast_inflate.hpp"
#include "parallel_gunzip.hpp"

namespace fqtk_host {

struct FastqRec {
    uint32_t head_off, head_len;   // header without the leading '@'
    uint32_t seq_off, seq_len;
    uint32_t qual_off;             // qual_len == seq_len
    uint32_t buf = 0;              // which of the batch's buffers the offsets are in (decoded inputs)
};

// The bytes of a batch are NOT copied together.  A mapped plain file: `mapped` points into the mapping.  Anything that
// is decoded or read in pieces: the batch holds (shares) the pieces its records lie in -- a record never straddles
// two of them, the reader moves the few hundred bytes of a straddling record in front of the next piece.
struct RecBatch {
    std::vector<std::shared_ptr<ByteVec>> bufs;
    const char *mapped = nullptr;
    std::vector<FastqRec> recs;
    // filled by the demux reader threads (not by the parser): reads shorter than the read structure needs,
    // and this input's fixed-length sample-barcode segments, packed side by side (one row per record)
    std::vector<uint8_t> too_short, bc;
    size_t n_short = 0;
    const char *rec_base(size_t i) const { return mapped ? mapped : bufs[recs[i].buf]->data(); }
    const char *head(size_t i) const { return rec_base(i) + recs[i].head_off; }
    const char *seq(size_t i) const { return rec_base(i) + recs[i].seq_off; }
    const char *qual(size_t i) const { return rec_base(i) + recs[i].qual_off; }
};

// libdeflate's decompressor, bound at run time (the image ships libdeflate.so.0 without its header).
struct LibInflate {
    bool ok = false;
    void *(*alloc)() = nullptr;
    int (*run)(void *, const void *, size_t, void *, size_t, size_t *) = nullptr;
    void (*free_)(void *) = nullptr;
    static const LibInflate &get() {
        static const LibInflate l = [] {
            LibInflate x;
            if (env_on("FQTK_NO_LIBDEFLATE")) return x;
            void *h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
            if (!h) h = dlopen("libdeflate.so", RTLD_NOW | RTLD_LOCAL);
            if (!h) return x;
            x.alloc = reinterpret_cast<void *(*)()>(dlsym(h, "libdeflate_alloc_decompressor"));
            x.run = reinterpret_cast<int (*)(void *, const void *, size_t, void *, size_t, size_t *)>(dlsym(h, "libdeflate_deflate_decompress"));
            x.free_ = reinterpret_cast<void (*)(void *)>(dlsym(h, "libdeflate_free_decompressor"));
            x.ok = x.alloc && x.run && x.free_;
            return x;
        }();
        return l;
    }
};

// Raw DEFLATE payload of one BGZF member -> out (exactly out_len bytes).  Per-thread state.
class BlockInflater {
  public:
    BlockInflater() { if (LibInflate::get().ok) d_ = LibInflate::get().alloc(); }
    ~BlockInflater() { if (d_) LibInflate::get().free_(d_); }
    BlockInflater(const BlockInflater &) = delete;
    BlockInflater &operator=(const BlockInflater &) = delete;
    bool inflate_block(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len) {
        if (d_) {
            size_t got = 0;
            return LibInflate::get().run(d_, in, in_len, out, out_len, &got) == 0 && got == out_len;
        }
        z_stream zs;
        std::memset(&zs, 0, sizeof zs);
        if (inflateInit2(&zs, -15) != Z_OK) return false;
        zs.next_in = const_cast<Bytef *>(in);
        zs.avail_in = (uInt)in_len;
        zs.next_out = out;
        zs.avail_out = (uInt)out_len;
        const int rc = ::inflate(&zs, Z_FINISH);
        const bool ok = rc == Z_STREAM_END && zs.total_out == out_len;
        inflateEnd(&zs);
        return ok;
    }
  private:
    void *d_ = nullptr;
};

// Number of '\n' in [p, p + n): 32 bytes per step where the CPU has AVX2 (the record pipeline cuts its inputs by this
// count alone: a FASTQ record is four lines).
inline size_t count_newlines_scalar(const char *p, size_t n) {
    size_t c = 0;
    for (size_t i = 0; i < n; ++i) c += p[i] == '\n';
    return c;
}
#if defined(__x86_64__) && defined(__GNUC__)
__attribute__((target("avx2,popcnt"))) inline size_t count_newlines_avx2(const char *p, size_t n) {
    typedef char v32 __attribute__((vector_size(32), aligned(1)));
    const v32 nl = {'\n', '\n', '\n', '\n', '\n', '\n', '\n', '\n', '\n', '\n', '\n', '\n', '\n', '\n', '\n', '\n',
                    '\n', '\n', '\n', '\n', '\n', '\n', '\n', '\n', '\n', '\n', '\n', '\n', '\n', '\n', '\n', '\n'};
    size_t c = 0, i = 0;
    for (; i + 128 <= n; i += 128) {
        const v32 a = *reinterpret_cast<const v32 *>(p + i) == nl, b = *reinterpret_cast<const v32 *>(p + i + 32) == nl;
        const v32 d = *reinterpret_cast<const v32 *>(p + i + 64) == nl, e = *reinterpret_cast<const v32 *>(p + i + 96) == nl;
        c += (size_t)__builtin_popcount((unsigned)__builtin_ia32_pmovmskb256(a)) + (size_t)__builtin_popcount((unsigned)__builtin_ia32_pmovmskb256(b)) +
             (size_t)__builtin_popcount((unsigned)__builtin_ia32_pmovmskb256(d)) + (size_t)__builtin_popcount((unsigned)__builtin_ia32_pmovmskb256(e));
    }
    for (; i < n; ++i) c += p[i] == '\n';
    return c;
}
#endif
inline size_t count_newlines(const char *p, size_t n) {
#if defined(__x86_64__) && defined(__GNUC__)
    static const bool avx2 = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("popcnt");
    if (avx2) return count_newlines_avx2(p, n);
#endif
    return count_newlines_scalar(p, n);
}

// Destination of FastqSource::next_raw: a buffer the caller can enlarge (page-locked memory in `fqtk demux`).
struct RawBuffer {
    char *data = nullptr;
    size_t cap = 0;
    virtual bool grow(size_t want, size_t keep) = 0;   // cap >= want afterwards, the first `keep` bytes preserved
    virtual ~RawBuffer() = default;
};

class FastqSource {
  public:
    enum class Kind { Plain, Gzip, Bgzf };
    ~FastqSource() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        cv_space_.notify_all();
        cv_work_.notify_all();
        if (producer_.joinable()) producer_.join();
        for (auto &t : helpers_) if (t.joinable()) t.join();
        if (gz_) gzclose(gz_);
        pgz_.reset();   // their decoder threads read the mapping: gone before it is
        fast_.reset();
        if (gz_map_) munmap(const_cast<uint8_t *>(gz_map_), gz_map_size_);   // compressed bytes: nothing points into them
        if (fd_ >= 0) ::close(fd_);
        // a mapping stays for the life of the process: record batches point into it
    }
    // inflate_helpers: extra threads that inflate BGZF blocks next to the producer (ignored for other kinds)
    // gz_threads: decoders for a single-stream gzip input (0 = decide here; 1 = the sequential decoder)
    bool open(const std::string &path, std::string *err, unsigned inflate_helpers = 2, unsigned gz_threads = 0) {
        path_ = path;
        fd_ = ::open(path.c_str(), O_RDONLY);
        if (fd_ < 0) { *err = "Error opening input files for reading: " + path; return false; }
        uint8_t hdr[18];
        const ssize_t n = ::pread(fd_, hdr, sizeof hdr, 0);
        kind_ = Kind::Plain;
        if (n >= 2 && hdr[0] == 0x1f && hdr[1] == 0x8b) {
            kind_ = Kind::Gzip;
            if (n == 18 && hdr[2] == 8 && (hdr[3] & 4) && hdr[10] == 6 && hdr[11] == 0 && hdr[12] == 'B' && hdr[13] == 'C' &&
                hdr[14] == 2 && hdr[15] == 0)
                kind_ = Kind::Bgzf;
        }
        if (kind_ == Kind::Gzip && !env_on("FQTK_ZLIB_INFLATE")) {
            // a regular file: map it and decode with the streaming decoder of fast_inflate.hpp (2x zlib's inflate)
            struct stat st;
            if (fstat(fd_, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0) {
                void *m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd_, 0);
                if (m != MAP_FAILED) {
                    gz_map_ = static_cast<const uint8_t *>(m);
                    gz_map_size_ = (size_t)st.st_size;
                    madvise(m, gz_map_size_, MADV_SEQUENTIAL);
                    // Large files: several decoders side by side (parallel_gunzip.hpp); the caller shares the CPUs out
                    // among its inputs (demux.cpp), FQTK_GZ_THREADS overrides (1: the sequential decoder).
                    unsigned t = gz_threads ? gz_threads : std::min(8u, std::max(2u, usable_cpus() * 3 / 8));
                    if (const char *g = std::getenv("FQTK_GZ_THREADS")) if (*g) t = (unsigned)std::atoi(g);
                    size_t chunk = ParallelGunzip::kChunk;   // (FQTK_GZ_CHUNK: tests cross many chunk boundaries in small files)
                    if (const char *g = std::getenv("FQTK_GZ_CHUNK")) if (*g) chunk = (size_t)std::atol(g);
                    if (t >= 2 && gz_map_size_ >= 32 * chunk) {   // 64 MiB by default: below that the sequential decoder is done in 0.2 s
                        pgz_.reset(new ParallelGunzip());
                        pgz_->open(gz_map_, gz_map_size_, &FastqSource::crc32_fn, t, chunk, kHead);
                    } else {
                        fast_.reset(new FastInflate());
                        fast_->open(gz_map_, gz_map_size_, &FastqSource::crc32_fn);
                    }
                }
            }
        }
        if (kind_ == Kind::Gzip && !fast_ && !pgz_) {   // pipes and the like: zlib's gzread
            gz_ = gzdopen(fd_, "rb");
            if (!gz_) { *err = "Error opening input files for reading: " + path; return false; }
            fd_ = -1;               // owned by gz_ now
            gzbuffer(gz_, 1 << 20);
        }
        n_helpers_ = kind_ == Kind::Bgzf ? inflate_helpers : 0;
        if (kind_ == Kind::Plain && !env_on("FQTK_NO_MMAP")) {
            struct stat st;
            if (fstat(fd_, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0) {
                void *m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd_, 0);
                if (m != MAP_FAILED) {
                    map_ = static_cast<const char *>(m);
                    map_size_ = (size_t)st.st_size;
                    madvise(m, map_size_, MADV_SEQUENTIAL);
                }
            }
        }
        return true;
    }
    Kind kind() const { return kind_; }

    // Reads up to max_records records.  Returns false on a malformed file (*err set).  An empty batch = EOF.
    //
    // Parsed IN PLACE: pieces are appended to the batch's own data vector, lines are located with memchr and a
    // record is four (offset, length) pairs into that vector -- no per-record copy.  Bytes read past the
    // last record of this batch (at most one piece) are carried into the next batch.
    bool next_batch(size_t max_records, RecBatch *out, std::string *err) {
        if (map_) return next_batch_mapped(max_records, out, err);
        if (!producer_.joinable()) start();
        out->recs.clear();
        out->recs.reserve(std::min<size_t>(max_records, 1u << 20));
        out->bufs.clear();
        out->mapped = nullptr;
        int cur_idx = -1;   // index of cur_ in out->bufs (-1: not referenced by this batch yet)
        while (out->recs.size() < max_records) {
            // four complete lines in what is left of the current piece?
            size_t lo[4], len[4], p = cur_pos_;
            int got = 0;
            const char *base = cur_ ? cur_->data() : nullptr;
            while (got < 4 && p < cur_end_) {
                const char *nl = (const char *)memchr(base + p, '\n', cur_end_ - p);
                if (!nl) break;
                lo[got] = p;
                len[got] = (size_t)(nl - (base + p));
                p += len[got] + 1;
                ++got;
            }
            if (got < 4) {
                if (!eof_) {   // the record continues in the next piece: its beginning moves in front of that piece
                    if (!next_piece(err)) return false;
                    cur_idx = -1;
                    continue;
                }
                if (got == 3 && p < cur_end_) {   // final line without '\n'
                    lo[3] = p;
                    len[3] = cur_end_ - p;
                    p = cur_end_;
                    got = 4;
                } else {
                    // EOF: anything left must be blank
                    for (size_t q = cur_pos_; q < cur_end_; ++q)
                        if (base[q] != '\n' && base[q] != '\r') {
                            *err = "Unexpected error parsing FASTQs: truncated record at end of " + path_;
                            return false;
                        }
                    cur_pos_ = cur_end_;
                    break;
                }
            }
            for (int k = 0; k < 4; ++k)
                if (len[k] > 0 && base[lo[k] + len[k] - 1] == '\r') --len[k];
            if (len[0] == 0 || base[lo[0]] != '@') {
                *err = "Unexpected error parsing FASTQs: expected '@' at record " + std::to_string(nrec_) + " of " + path_;
                return false;
            }
            if (len[2] == 0 || base[lo[2]] != '+') {
                *err = "Unexpected error parsing FASTQs: expected '+' at record " + std::to_string(nrec_) + " of " + path_;
                return false;
            }
            if (len[1] != len[3]) {
                *err = "Unexpected error parsing FASTQs: sequence and quality lengths differ at record " +
                       std::to_string(nrec_) + " of " + path_;
                return false;
            }
            if (p > 0xFFFFFFFFull) { *err = "Unexpected error parsing FASTQs: piece larger than 4 GiB in " + path_; return false; }
            if (cur_idx < 0) {
                cur_idx = (int)out->bufs.size();
                out->bufs.push_back(cur_);
            }
            FastqRec r;
            r.head_off = (uint32_t)lo[0] + 1;
            r.head_len = (uint32_t)len[0] - 1;
            r.seq_off = (uint32_t)lo[1];
            r.seq_len = (uint32_t)len[1];
            r.qual_off = (uint32_t)lo[3];
            r.buf = (uint32_t)cur_idx;
            out->recs.push_back(r);
            cur_pos_ = p;
            ++nrec_;
        }
        return true;
    }

    // The text of up to max_records records, uninterpreted, copied to dst: *n_records records in *n_bytes bytes, the
    // last byte a newline.  Four lines are a record (seq_io's reader knows no multi-line FASTQ either): the cut is found
    // by counting newlines; what the lines hold is checked where the text is used (the GPU record pipeline).  At the
    // end of the input fewer records come back, then none; up to three trailing blank lines are dropped, as the
    // parsing reader drops them.  Do not mix with next_batch() on one source.
    bool next_raw(size_t max_records, RawBuffer *dst, size_t *n_records, size_t *n_bytes, std::string *err) {
        if (!map_ && !producer_.joinable()) start();
        size_t w = 0, lines = 0;
        const size_t target = 4 * max_records;
        auto room = [&](size_t want) { return want <= dst->cap || dst->grow(std::max(want + 65536, dst->cap + dst->cap / 2), w); };
        bool eof = false;
        // A plain regular file is copied out of its mapping (the part already consumed is unmapped behind the reader, by
        // a helper thread: tearing down the page tables of tens of GB at exit took half a second).  FQTK_RAW_PREAD=1
        // reads it with pread() straight into the destination instead (measured on tmpfs: 6 GB/s per thread against
        // 10 through the mapping).  Everything else comes in windows of decoded / piped pieces.
        const bool by_pread = map_ && fd_ >= 0 && env_on("FQTK_RAW_PREAD");
        while (lines < target) {
            if (by_pread) {
                if (map_pos_ >= map_size_) { eof = true; break; }
                const size_t want = std::min<size_t>(map_size_ - map_pos_, 256u << 10);
                if (!room(w + want + 1)) { *err = "out of page-locked memory"; return false; }
                const ssize_t got = ::pread(fd_, dst->data + w, want, (off_t)map_pos_);
                if (got < 0) { if (errno == EINTR) continue; *err = "Unexpected error parsing FASTQs: read failed in " + path_; return false; }
                if (got == 0) { eof = true; break; }
                size_t take = (size_t)got, c = count_newlines(dst->data + w, take);
                if (lines + c >= target) {
                    const char *q = dst->data + w;
                    for (size_t need = target - lines; need; --need) q = static_cast<const char *>(std::memchr(q, '\n', (size_t)(dst->data + w + take - q))) + 1;
                    take = (size_t)(q - (dst->data + w));
                    c = target - lines;
                }
                w += take;
                lines += c;
                map_pos_ += take;
                continue;
            }
            const char *p = nullptr;
            size_t avail = 0;
            if (!raw_window(&p, &avail, &eof, err)) return false;
            if (eof) break;
            size_t take = std::min<size_t>(avail, 256u << 10);
            size_t c = count_newlines(p, take);
            if (lines + c >= target) {   // the cut lies in this block: right behind its (target - lines)-th newline
                const char *q = p;
                for (size_t need = target - lines; need; --need) q = static_cast<const char *>(std::memchr(q, '\n', (size_t)(p + take - q))) + 1;
                take = (size_t)(q - p);
                c = target - lines;
            }
            if (!room(w + take + 1)) { *err = "out of page-locked memory"; return false; }
            std::memcpy(dst->data + w, p, take);
            w += take;
            lines += c;
            raw_consume(take);
        }
        if (lines < target) {   // end of the input
            if (w && dst->data[w - 1] != '\n') { dst->data[w++] = '\n'; ++lines; }   // final line without '\n' (room was kept)
            size_t extra = lines % 4;
            while (extra) {   // lines beyond the last whole record: blank, or the file is cut short
                size_t e = w - 1;                      // the newline that ends the last line
                size_t b = e;
                while (b > 0 && dst->data[b - 1] != '\n') --b;
                for (size_t q = b; q < e; ++q)
                    if (dst->data[q] != '\r') { *err = "Unexpected error parsing FASTQs: truncated record at end of " + path_; return false; }
                w = b;
                --lines;
                --extra;
            }
        }
        *n_records = lines / 4;
        *n_bytes = w;
        nrec_ += lines / 4;
        return true;
    }
    uint64_t records_read() const { return nrec_; }
    // A plain mapped file can be cut WITHOUT being copied: next_cut only counts newlines (twice the speed of counting and
    // copying) and hands out the range; whoever copies it (several threads, `fqtk demux`) does so while the next cut is
    // being counted.  The range stays mapped until its consumer hands it back with release_cut(): however far the cutter
    // runs ahead of the copiers (queues between them, cuts of any size), nothing below the last released cut's end is
    // unmapped -- a distance in bytes cannot promise that (four cuts of more than 256 MiB outran a 1 GiB lag).
    struct RawCut { const char *p = nullptr; size_t bytes = 0, n_records = 0; bool add_newline = false; };
    // Cuts are released in the order they were handed out (one copier per source).
    void release_cut(const RawCut &c) { cut_released_.store((size_t)(c.p - map_) + c.bytes, std::memory_order_release); }
    bool mapped() const { return map_ != nullptr; }
    size_t mapped_size() const { return map_size_; }
    // A second thread may count the later steps of a cut while this one counts the first ones (the counts of whole 256 KiB
    // steps do not depend on where the cut will fall): one thread counts ~15 GB/s out of the page cache, and the device
    // takes the two 150-base files of a run faster than that.
    void set_cut_unmap_step(size_t bytes) { cut_unmap_step_ = std::max<size_t>(4096, bytes / 4096 * 4096); }
    size_t cut_unmapped_bytes() const { return map_unmapped_; }
    void attach_count_assistant() { if (!assistant_) assistant_ = std::make_unique<CountAssistant>(); }
    bool next_cut(size_t max_records, RawCut *c, std::string *err) {
        const char *base = map_ + map_pos_;
        const size_t avail = map_size_ - map_pos_, target = 4 * max_records;
        constexpr size_t kCutStep = 256u << 10;
        size_t lines = 0, w = 0;
        // the assistant takes the second half of what the last cut was long (whole steps that lie inside the file)
        size_t a_first = 0, a_steps = 0;
        if (assistant_ && last_cut_bytes_ >= 16 * kCutStep) {
            const size_t guess = std::min(last_cut_bytes_, avail) / kCutStep;   // whole steps
            a_first = guess / 2;
            a_steps = guess - a_first;
            if (a_steps) assistant_->start(base, a_first, a_steps, kCutStep);
        }
        while (lines < target && w < avail) {
            size_t take = std::min<size_t>(avail - w, kCutStep);
            const size_t step = w / kCutStep;
            size_t cnt = (a_steps && step >= a_first && step < a_first + a_steps && take == kCutStep) ? assistant_->wait(step - a_first) : count_newlines(base + w, take);
            if (lines + cnt >= target) {
                const char *q = base + w;
                for (size_t need = target - lines; need; --need) q = static_cast<const char *>(std::memchr(q, '\n', (size_t)(base + w + take - q))) + 1;
                take = (size_t)(q - (base + w));
                cnt = target - lines;
            }
            w += take;
            lines += cnt;
        }
        size_t end = w;
        bool add_nl = false;
        if (lines < target) {   // end of the input: a final line without '\n', up to three trailing blank lines (as next_raw)
            if (w && base[w - 1] != '\n') { add_nl = true; ++lines; }
            for (size_t extra = lines % 4; extra; --extra) {
                const size_t e = add_nl ? end : end - 1;
                size_t b = e;
                while (b > 0 && base[b - 1] != '\n') --b;
                for (size_t q = b; q < e; ++q)
                    if (base[q] != '\r') { *err = "Unexpected error parsing FASTQs: truncated record at end of " + path_; return false; }
                end = b;
                add_nl = false;
                --lines;
            }
        }
        if (a_steps) assistant_->finish();   // (it may still be counting steps behind the cut: they are not needed)
        last_cut_bytes_ = w;
        c->p = base;
        c->bytes = end;
        c->n_records = lines / 4;
        c->add_newline = add_nl;
        nrec_ += lines / 4;
        map_pos_ += w;
        // unmap what the copiers have handed back, in steps of kCutUnmapStep, whole pages only
        const size_t released = cut_released_.load(std::memory_order_acquire);
        if (released >= map_unmapped_ + 2 * cut_unmap_step_) {
            const size_t upto = released / cut_unmap_step_ * cut_unmap_step_;
            Unmapper::get().push(const_cast<char *>(map_) + map_unmapped_, upto - map_unmapped_);
            map_unmapped_ = upto;
        }
        return true;
    }
    // Bytes next_raw will want for max_records records, judged by the lines of the first MiB at hand (0: nothing to go by).
    size_t estimate_raw_bytes(size_t max_records) {
        if (!map_ && !producer_.joinable()) start();
        const char *p = nullptr;
        size_t avail = 0;
        bool eof = false;
        std::string err;
        if (!raw_window(&p, &avail, &eof, &err) || eof || avail == 0) return 0;
        const size_t look = std::min<size_t>(avail, 1u << 20);
        const size_t lines = count_newlines(p, look);
        if (lines < 4) return 0;
        const double per_record = 4.0 * (double)look / (double)lines;
        size_t want = (size_t)(per_record * 1.08 * (double)max_records) + 65536;
        if (map_) want = std::min(want, map_size_ - map_pos_ + 65536);
        return want;
    }

  private:
    // next_raw's view of the input: the unread rest of the mapping, or of the current piece of a decoded / piped input.
    bool raw_window(const char **p, size_t *avail, bool *eof, std::string *err) {
        if (map_) {
            *p = map_ + map_pos_;
            *avail = map_size_ - map_pos_;
            *eof = *avail == 0;
            return true;
        }
        // (the producer reports damage ONCE, as its last piece, and ends: whoever asks again -- `fqtk demux` judges the record size by the
        //  first MiB before its reader threads start -- gets the same answer, not an empty queue to wait on for ever)
        if (!failed_.empty()) { *err = failed_; return false; }
        while (cur_pos_ == cur_end_ && !eof_) {
            Piece pc;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_data_.wait(lk, [&] { return !q_.empty(); });
                pc = std::move(q_.front());
                q_.pop_front();
            }
            cv_space_.notify_one();
            if (!pc.error.empty()) { failed_ = pc.error; *err = pc.error; return false; }
            if (pc.eof) { eof_ = true; break; }
            cur_ = std::move(pc.buf);
            cur_pos_ = pc.head;
            cur_end_ = pc.head + pc.len;
        }
        *eof = cur_pos_ == cur_end_;
        *p = cur_ ? cur_->data() + cur_pos_ : nullptr;
        *avail = cur_end_ - cur_pos_;
        return true;
    }
    void raw_consume(size_t n) {
        if (!map_) { cur_pos_ += n; return; }
        map_pos_ += n;
        constexpr size_t kStep = 64u << 20;
        if (map_pos_ - map_unmapped_ >= 2 * kStep) {   // whole steps, at least one step behind the reader
            const size_t upto = (map_pos_ - kStep) / kStep * kStep;
            Unmapper::get().push(const_cast<char *>(map_) + map_unmapped_, upto - map_unmapped_);
            map_unmapped_ = upto;
        }
    }
    std::string failed_;                    // the producer's error, once it has been taken off the queue
    size_t map_unmapped_ = 0;
    std::atomic<size_t> cut_released_{0};   // end offset of the last cut its consumer is done with
    size_t cut_unmap_step_ = 64u << 20;     // (tests shrink it: set_cut_unmap_step)
    size_t last_cut_bytes_ = 0;
    class CountAssistant {
      public:
        CountAssistant() : th_([this] { run(); }) {}
        ~CountAssistant() { { std::lock_guard<std::mutex> lk(mu_); stop_ = true; } cv_.notify_all(); th_.join(); }
        void start(const char *base, size_t first, size_t steps, size_t step_bytes) {
            {
                std::lock_guard<std::mutex> lk(mu_);
                base_ = base; first_ = first; steps_ = steps; step_ = step_bytes;
                if (counts_.size() < steps) counts_ = std::vector<std::atomic<int64_t>>(steps);
                for (size_t j = 0; j < steps; ++j) counts_[j].store(-1, std::memory_order_relaxed);
                cancel_.store(false, std::memory_order_relaxed);
                busy_ = true;
            }
            cv_.notify_all();
        }
        size_t wait(size_t j) {   // the count of step first + j
            int64_t v;
            while ((v = counts_[j].load(std::memory_order_acquire)) < 0) std::this_thread::yield();
            return (size_t)v;
        }
        void finish() {   // the cut is made: stop counting, and do not return before the thread has let go of the mapping
            cancel_.store(true, std::memory_order_relaxed);
            std::unique_lock<std::mutex> lk(mu_);
            cv_.wait(lk, [&] { return !busy_; });
        }
      private:
        void run() {
            for (;;) {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return stop_ || busy_; });
                if (stop_) return;
                lk.unlock();
                for (size_t j = 0; j < steps_ && !cancel_.load(std::memory_order_relaxed); ++j)
                    counts_[j].store((int64_t)count_newlines(base_ + (first_ + j) * step_, step_), std::memory_order_release);
                lk.lock();
                busy_ = false;
                cv_.notify_all();
            }
        }
        std::mutex mu_;
        std::condition_variable cv_;
        const char *base_ = nullptr;
        size_t first_ = 0, steps_ = 0, step_ = 0;
        std::vector<std::atomic<int64_t>> counts_;
        std::atomic<bool> cancel_{false};
        bool busy_ = false, stop_ = false;
        std::thread th_;
    };
    std::unique_ptr<CountAssistant> assistant_;
    // munmap() of consumed input, off the readers' critical path
    class Unmapper {
      public:
        static Unmapper &get() { static Unmapper *u = new Unmapper(); return *u; }   // (lives until the process ends)
        void push(char *p, size_t n) {
            { std::lock_guard<std::mutex> lk(mu_); q_.emplace_back(p, n); }
            cv_.notify_one();
        }
      private:
        Unmapper() {
            std::thread([this] {
                for (;;) {
                    std::pair<char *, size_t> job;
                    {
                        std::unique_lock<std::mutex> lk(mu_);
                        cv_.wait(lk, [&] { return !q_.empty(); });
                        job = q_.front();
                        q_.pop_front();
                    }
                    munmap(job.first, job.second);
                }
            }).detach();
        }
        std::mutex mu_;
        std::condition_variable cv_;
        std::deque<std::pair<char *, size_t>> q_;
    };

    // Plain regular file: the records of a batch are located in the mapping itself.
    bool next_batch_mapped(size_t max_records, RecBatch *out, std::string *err) {
        out->recs.clear();
        out->recs.reserve(std::min<size_t>(max_records, 1u << 20));
        out->bufs.clear();
        const char *base = map_ + map_pos_;
        const size_t avail = map_size_ - map_pos_;
        out->mapped = base;
        size_t pos = 0;
        while (out->recs.size() < max_records && pos < avail) {
            size_t lo[4], len[4], p = pos;
            int got = 0;
            while (got < 4) {
                const char *nl = p < avail ? (const char *)memchr(base + p, '\n', avail - p) : nullptr;
                if (!nl) break;
                lo[got] = p;
                len[got] = (size_t)(nl - (base + p));
                p += len[got] + 1;
                ++got;
            }
            if (got < 4) {
                if (got == 3 && p < avail) {   // final line without '\n'
                    lo[3] = p;
                    len[3] = avail - p;
                    p = avail;
                } else {
                    for (size_t q = pos; q < avail; ++q)
                        if (base[q] != '\n' && base[q] != '\r') {
                            *err = "Unexpected error parsing FASTQs: truncated record at end of " + path_;
                            return false;
                        }
                    pos = avail;
                    break;
                }
            }
            for (int k = 0; k < 4; ++k)
                if (len[k] > 0 && base[lo[k] + len[k] - 1] == '\r') --len[k];
            if (len[0] == 0 || base[lo[0]] != '@') {
                *err = "Unexpected error parsing FASTQs: expected '@' at record " + std::to_string(nrec_) + " of " + path_;
                return false;
            }
            if (len[2] == 0 || base[lo[2]] != '+') {
                *err = "Unexpected error parsing FASTQs: expected '+' at record " + std::to_string(nrec_) + " of " + path_;
                return false;
            }
            if (len[1] != len[3]) {
                *err = "Unexpected error parsing FASTQs: sequence and quality lengths differ at record " +
                       std::to_string(nrec_) + " of " + path_;
                return false;
            }
            if (p > 0xFFFFFFFFull) { *err = "Unexpected error parsing FASTQs: batch larger than 4 GiB in " + path_; return false; }
            FastqRec r;
            r.head_off = (uint32_t)lo[0] + 1;
            r.head_len = (uint32_t)len[0] - 1;
            r.seq_off = (uint32_t)lo[1];
            r.seq_len = (uint32_t)len[1];
            r.qual_off = (uint32_t)lo[3];
            out->recs.push_back(r);
            pos = p;
            ++nrec_;
        }
        map_pos_ += pos;
        return true;
    }

    static constexpr size_t kPiece = 4u << 20;
    static constexpr size_t kHead = 65536;   // free bytes in front of every piece (a straddling record's beginning goes there)
    struct Piece { std::shared_ptr<ByteVec> buf; size_t head = 0, len = 0; std::string error; bool eof = false; };

    // consumer side: next piece from the producer's queue
    // Makes the next piece current.  The unparsed rest of the current one (the beginning of a record) is moved in
    // front of it -- into the head room every piece comes with, or, if a record is longer than that, into a copy.
    bool next_piece(std::string *err) {
        if (!failed_.empty()) { *err = failed_; return false; }
        Piece pc;
        {
            std::unique_lock<std::mutex> lk(mu_);
            cv_data_.wait(lk, [&] { return !q_.empty(); });
            pc = std::move(q_.front());
            q_.pop_front();
        }
        cv_space_.notify_one();
        if (!pc.error.empty()) { failed_ = pc.error; *err = pc.error; return false; }
        if (pc.eof) { eof_ = true; return true; }
        if (pc.len == 0) return true;   // (a BGZF round that only read more input)
        const size_t rest = cur_ ? cur_end_ - cur_pos_ : 0;
        if (rest <= pc.head) {
            if (rest) std::memcpy(pc.buf->data() + pc.head - rest, cur_->data() + cur_pos_, rest);
            cur_pos_ = pc.head - rest;
            cur_end_ = pc.head + pc.len;
            cur_ = std::move(pc.buf);
        } else {
            auto joined = std::make_shared<ByteVec>();
            joined->resize(rest + pc.len);
            std::memcpy(joined->data(), cur_->data() + cur_pos_, rest);
            std::memcpy(joined->data() + rest, pc.buf->data() + pc.head, pc.len);
            cur_pos_ = 0;
            cur_end_ = rest + pc.len;
            cur_ = std::move(joined);
        }
        return true;
    }
    void push(Piece &&pc) {
        std::unique_lock<std::mutex> lk(mu_);
        // (a parallel gzip decoder delivers a whole stretch -- a dozen pieces -- at once and then computes the next one:
        //  the queue must hold a stretch, or parser and decoder take turns instead of overlapping)
        const size_t depth = pgz_ ? 32 : 4;
        cv_space_.wait(lk, [&] { return q_.size() < depth || stop_; });
        if (stop_) return;
        q_.push_back(std::move(pc));
        cv_data_.notify_one();
    }
    void start() {
        for (unsigned h = 0; h < n_helpers_; ++h) helpers_.emplace_back([this] { helper_loop(); });
        producer_ = std::thread([this] {
            for (;;) {
                Piece pc;
                bool more = kind_ == Kind::Plain ? produce_plain(pc) : (kind_ == Kind::Gzip || streaming_ ? produce_gzip(pc) : produce_bgzf(pc));
                const bool last = !more || !pc.error.empty() || pc.eof;
                push(std::move(pc));
                {
                    std::lock_guard<std::mutex> lk(mu_);
                    if (stop_) return;
                }
                if (last) return;
            }
        });
    }
    static void fresh(Piece &pc, size_t room) {
        pc.buf = std::make_shared<ByteVec>();
        pc.buf->resize(kHead + room);
        pc.head = kHead;
        pc.len = 0;
    }
    bool produce_plain(Piece &pc) {
        fresh(pc, kPiece);
        const ssize_t n = ::read(fd_, pc.buf->data() + kHead, kPiece);
        if (n < 0) { pc.error = "Unexpected error parsing FASTQs: read failed in " + path_; return false; }
        if (n == 0) { pc.eof = true; return false; }
        pc.len = (size_t)n;
        return true;
    }
    bool produce_gzip(Piece &pc) {
        if (pgz_) {   // a whole chunk of a decoded stretch, by swap
            auto v = std::make_shared<ByteVec>();
            if (pgz_->take_chunk(*v)) {
                pc.head = kHead;
                pc.len = v->size() - kHead;
                pc.buf = std::move(v);
                return true;
            }
        }
        if (fast_ || pgz_) {
            const uint8_t *p = nullptr;
            size_t n = 0;
            std::string e;
            if (!(pgz_ ? pgz_->next(&p, &n, &e) : fast_->next(&p, &n, &e))) {
                pc.error = "Unexpected error parsing FASTQs: " + e + " in " + path_;
                return false;
            }
            if (n == 0) { pc.eof = true; return false; }
            fresh(pc, n);
            std::memcpy(pc.buf->data() + kHead, p, n);
            pc.len = n;
            return true;
        }
        fresh(pc, kPiece);
        const int n = gzread(gz_, pc.buf->data() + kHead, (unsigned)kPiece);
        if (n < 0) {
            int e = 0;
            pc.error = std::string("Unexpected error parsing FASTQs: ") + gzerror(gz_, &e) + " in " + path_;
            return false;
        }
        if (n == 0) { pc.eof = true; return false; }
        pc.len = (size_t)n;
        return true;
    }
    // ---- BGZF: read a group of whole members, inflate them in parallel into one piece --------------------
    struct Block { size_t in_off, in_len, out_off, out_len; uint32_t crc; };
    bool produce_bgzf(Piece &pc) {
        // top the compressed buffer up, then cut it at member boundaries (header: 18 bytes, BSIZE at 16)
        if (!craw_eof_ && craw_.size() - cpos_ < kPiece) {
            craw_.erase(craw_.begin(), craw_.begin() + (std::ptrdiff_t)cpos_);
            craw_file_off_ += cpos_;
            cpos_ = 0;
            const size_t old = craw_.size();
            craw_.resize(old + kPiece);
            const ssize_t n = ::read(fd_, craw_.data() + old, kPiece);
            if (n < 0) { pc.error = "Unexpected error parsing FASTQs: read failed in " + path_; return false; }
            craw_.resize(old + (size_t)std::max<ssize_t>(n, 0));
            if (n == 0) craw_eof_ = true;
        }
        blocks_.clear();
        size_t p = cpos_, out_total = 0;
        while (p + 18 <= craw_.size() && out_total < kPiece) {
            const uint8_t *h = craw_.data() + p;
            if (h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4) || h[10] != 6 || h[11] != 0 || h[12] != 'B' || h[13] != 'C') {
                if (h[0] == 0x1f && h[1] == 0x8b) {
                    // an ordinary gzip member behind the BGZF ones (gzread and the reference's reader take the file as
                    // one multi-member stream): what was cut so far goes out, the rest through the streaming decoder
                    if (blocks_.empty()) return switch_to_stream(p, pc);
                    break;
                }
                pc.error = "Unexpected error parsing FASTQs: not a BGZF member at compressed offset in " + path_;
                return false;
            }
            const size_t bsize = (size_t)h[16] + ((size_t)h[17] << 8) + 1;
            if (bsize < 26) { pc.error = "Unexpected error parsing FASTQs: bad BGZF block size in " + path_; return false; }
            if (p + bsize > craw_.size()) break;
            const uint8_t *t = h + bsize - 4;
            const size_t isize = (size_t)t[0] | ((size_t)t[1] << 8) | ((size_t)t[2] << 16) | ((size_t)t[3] << 24);
            if (isize > 65536) { pc.error = "Unexpected error parsing FASTQs: BGZF member larger than 64 KiB in " + path_; return false; }
            const uint32_t crc = (uint32_t)t[-4] | ((uint32_t)t[-3] << 8) | ((uint32_t)t[-2] << 16) | ((uint32_t)t[-1] << 24);
            blocks_.push_back(Block{p + 18, bsize - 26, out_total, isize, crc});
            out_total += isize;
            p += bsize;
        }
        if (blocks_.empty()) {
            if (craw_eof_) {
                if (cpos_ != craw_.size()) { pc.error = "Unexpected error parsing FASTQs: truncated BGZF member at end of " + path_; return false; }
                pc.eof = true;
                return false;
            }
            return true;   // a member larger than what is buffered: read more next round (empty piece)
        }
        cpos_ = p;
        fresh(pc, out_total);
        pc.len = out_total;
        // fan the blocks out: helpers and this thread claim them one by one
        {
            std::lock_guard<std::mutex> lk(mu_);
            job_out_ = reinterpret_cast<uint8_t *>(pc.buf->data() + kHead);
            job_next_.store(0);
            job_done_ = 0;
            job_failed_ = false;
            job_size_.store(blocks_.size());   // opens the job: blocks_, craw_ and job_out_ are frozen until it closes
            ++job_epoch_;
        }
        cv_work_.notify_all();
        work(own_inflater_);
        {
            std::unique_lock<std::mutex> lk(mu_);
            cv_done_.wait(lk, [&] { return job_done_ == blocks_.size(); });
            job_size_.store(0);                // closed: a helper that wakes late claims nothing
            if (job_failed_) { pc.error = "Unexpected error parsing FASTQs: corrupt BGZF block in " + path_; return false; }
        }
        return true;
    }
    // BGZF members followed by an ordinary gzip member at craw_[p]: map the file and decode from there on with the
    // streaming decoder (produce_gzip's path).
    bool switch_to_stream(size_t p, Piece &pc) {
        const size_t abs = craw_file_off_ + p;
        struct stat st;
        void *m = MAP_FAILED;
        if (fstat(fd_, &st) == 0 && S_ISREG(st.st_mode) && (size_t)st.st_size > abs)
            m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd_, 0);
        if (m == MAP_FAILED) {
            pc.error = "Unexpected error parsing FASTQs: not a BGZF member at compressed offset in " + path_;
            return false;
        }
        gz_map_ = static_cast<const uint8_t *>(m);
        gz_map_size_ = (size_t)st.st_size;
        fast_.reset(new FastInflate());
        fast_->open(gz_map_ + abs, gz_map_size_ - abs, &FastqSource::crc32_fn);
        streaming_ = true;
        return produce_gzip(pc);
    }
    void work(BlockInflater &inf) {
        size_t done = 0;
        bool failed = false;
        for (;;) {
            const size_t i = job_next_.fetch_add(1);
            if (i >= job_size_.load()) break;
            const Block &b = blocks_[i];
            if (!inf.inflate_block(craw_.data() + b.in_off, b.in_len, job_out_ + b.out_off, b.out_len) ||
                crc32_fn(0, job_out_ + b.out_off, b.out_len) != b.crc)
                failed = true;
            ++done;
        }
        if (done || failed) {
            std::lock_guard<std::mutex> lk(mu_);
            job_done_ += done;
            job_failed_ = job_failed_ || failed;
            cv_done_.notify_all();
        }
    }
    void helper_loop() {
        BlockInflater inf;
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_work_.wait(lk, [&] { return stop_ || job_epoch_ != seen; });
                if (stop_) return;
                seen = job_epoch_;
            }
            work(inf);
        }
    }

    Kind kind_ = Kind::Plain;
    const char *map_ = nullptr;
    size_t map_size_ = 0, map_pos_ = 0;
    gzFile gz_ = nullptr;
    const uint8_t *gz_map_ = nullptr;       // single-stream gzip, regular file: decoded by fast_
    size_t gz_map_size_ = 0;
    std::unique_ptr<FastInflate> fast_;
    std::unique_ptr<ParallelGunzip> pgz_;
    // CRC-32 for the gzip trailers: libdeflate's (PCLMUL) when the library is there, zlib's otherwise
    static uint32_t crc32_fn(uint32_t seed, const void *p, size_t n) {
        using Fn = uint32_t (*)(uint32_t, const void *, size_t);
        static const Fn fast = [] {
            if (env_on("FQTK_NO_LIBDEFLATE")) return (Fn) nullptr;
            void *h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
            if (!h) h = dlopen("libdeflate.so", RTLD_NOW | RTLD_LOCAL);
            return h ? reinterpret_cast<Fn>(dlsym(h, "libdeflate_crc32")) : (Fn) nullptr;
        }();
        if (fast) return fast(seed, p, n);
        uLong c = seed;
        const Bytef *b = static_cast<const Bytef *>(p);
        while (n) {   // zlib takes 32-bit lengths
            const uInt k = n > (1u << 30) ? (1u << 30) : (uInt)n;
            c = ::crc32(c, b, k);
            b += k;
            n -= k;
        }
        return (uint32_t)c;
    }
    int fd_ = -1;
    std::string path_;
    std::shared_ptr<ByteVec> cur_;          // the piece being parsed (decoded inputs), its unparsed range
    size_t cur_pos_ = 0, cur_end_ = 0;
    bool eof_ = false;
    uint64_t nrec_ = 0;
    // producer / consumer
    std::thread producer_;
    std::mutex mu_;
    std::condition_variable cv_data_, cv_space_, cv_work_, cv_done_;
    std::deque<Piece> q_;
    bool stop_ = false;
    // BGZF group state
    unsigned n_helpers_ = 0;
    std::vector<std::thread> helpers_;
    std::vector<uint8_t> craw_;
    size_t cpos_ = 0, craw_file_off_ = 0;   // craw_[0] is byte craw_file_off_ of the file
    bool craw_eof_ = false;
    bool streaming_ = false;                // an ordinary gzip member followed the BGZF ones: fast_ decodes the rest
    std::vector<Block> blocks_;
    BlockInflater own_inflater_;
    uint8_t *job_out_ = nullptr;
    std::atomic<size_t> job_next_{0}, job_size_{0};
    size_t job_done_ = 0;
    bool job_failed_ = false;
    uint64_t job_epoch_ = 0;
};

}  // namespace fqtk_host
