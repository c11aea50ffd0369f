// bgzf.hpp -- BGZF block compression for the per-sample output FASTQs.
//
// SURVEY.md section 8(f) row 3 (output side).  The reference writes through pooled-writer's
// BgzfCompressor (/root/reference/src/bin/commands/demux.rs:755-798): independent <= 64 KiB gzip
// members carrying the 'BC' extra field, terminated by the 28-byte EOF block.  The reference's tests
// only compare DECOMPRESSED content (demux.rs:1069-1076), so parity here = a valid BGZF stream whose
// decompressed bytes are identical; the compressed bytes differ (zlib here, libdeflate there).
#pragma once
#include <dlfcn.h>
#include <zlib.h>

#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "env.hpp"

namespace fqtk_host {

constexpr size_t kBgzfBlockSize = 65280;   // uncompressed payload per block (as the bgzf crate)

static const uint8_t kBgzfEof[28] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0, 0x42, 0x43,
                                     0x02, 0, 0x1b, 0, 0x03, 0, 0, 0, 0, 0, 0, 0, 0, 0};

// Per-thread block compressor.  The reference compresses with libdeflate (pooled-writer -> bgzf ->
// libdeflater); the image ships libdeflate's runtime (libdeflate.so.0) but not its header, so its five
// public entry points are declared here and bound with dlopen; when the library is absent the
// compressor falls back to a persistent zlib stream (deflateReset per block).  Either way the output
// is a valid BGZF block; the compressed bytes are unpinned by the reference's tests.
class BlockCompressor {
  public:
    explicit BlockCompressor(int level) : level_(level) {
        static const LibDeflate ld = LibDeflate::load();
        ld_ = &ld;
        if (ld_->ok) {
            // libdeflate levels run 0..12; the reference passes its --compression-level straight through
            comp_ = ld_->alloc(level);
        }
        if (!comp_) {
            memset(&zs_, 0, sizeof zs_);
            zinit_ = deflateInit2(&zs_, level > 9 ? 9 : level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) == Z_OK;
        }
    }
    ~BlockCompressor() {
        if (comp_) ld_->free_(comp_);
        if (zinit_) deflateEnd(&zs_);
    }
    BlockCompressor(const BlockCompressor &) = delete;
    BlockCompressor &operator=(const BlockCompressor &) = delete;
    bool using_libdeflate() const { return comp_ != nullptr; }

    // Appends one BGZF block holding in[0..n) (n <= kBgzfBlockSize) to `out`.
    bool compress(const uint8_t *in, size_t n, std::vector<uint8_t> &out, std::string *err) {
        const size_t start = out.size();
        size_t cap = comp_ ? ld_->bound(comp_, n) : (size_t)deflateBound(&zs_, (uLong)n);
        cap += 64;
        out.resize(start + 18 + cap + 8);
        size_t clen = 0;
        uint32_t crc = 0;
        if (comp_) {
            clen = ld_->compress(comp_, in, n, out.data() + start + 18, cap);
            if (clen == 0) { *err = "libdeflate_deflate_compress failed"; return false; }
            crc = ld_->crc32(0, in, n);
        } else {
            if (!zinit_) { *err = "deflateInit2 failed"; return false; }
            deflateReset(&zs_);
            zs_.next_in = const_cast<Bytef *>(in);
            zs_.avail_in = (uInt)n;
            zs_.next_out = out.data() + start + 18;
            zs_.avail_out = (uInt)cap;
            if (deflate(&zs_, Z_FINISH) != Z_STREAM_END) { *err = "deflate failed"; return false; }
            clen = zs_.total_out;
            crc = (uint32_t)::crc32(::crc32(0L, Z_NULL, 0), in, (uInt)n);
        }
        const size_t bsize = 18 + clen + 8;
        if (bsize > 65536) { *err = "BGZF block overflow"; return false; }
        uint8_t *h = out.data() + start;
        const uint8_t hdr[18] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0, 0x42, 0x43, 0x02, 0,
                                 (uint8_t)((bsize - 1) & 0xff), (uint8_t)((bsize - 1) >> 8)};
        memcpy(h, hdr, 18);
        uint8_t *t = h + 18 + clen;
        t[0] = crc & 0xff; t[1] = (crc >> 8) & 0xff; t[2] = (crc >> 16) & 0xff; t[3] = (crc >> 24) & 0xff;
        t[4] = n & 0xff; t[5] = (n >> 8) & 0xff; t[6] = (n >> 16) & 0xff; t[7] = (n >> 24) & 0xff;
        out.resize(start + bsize);
        return true;
    }

  private:
    struct LibDeflate {
        bool ok = false;
        void *(*alloc)(int) = nullptr;
        size_t (*compress)(void *, const void *, size_t, void *, size_t) = nullptr;
        size_t (*bound)(void *, size_t) = nullptr;
        void (*free_)(void *) = nullptr;
        uint32_t (*crc32)(uint32_t, const void *, size_t) = nullptr;
        static LibDeflate load() {
            LibDeflate l;
            if (env_on("FQTK_NO_LIBDEFLATE")) return l;
            void *h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
            if (!h) h = dlopen("libdeflate.so", RTLD_NOW | RTLD_LOCAL);
            if (!h) return l;
            l.alloc = reinterpret_cast<void *(*)(int)>(dlsym(h, "libdeflate_alloc_compressor"));
            l.compress = reinterpret_cast<size_t (*)(void *, const void *, size_t, void *, size_t)>(dlsym(h, "libdeflate_deflate_compress"));
            l.bound = reinterpret_cast<size_t (*)(void *, size_t)>(dlsym(h, "libdeflate_deflate_compress_bound"));
            l.free_ = reinterpret_cast<void (*)(void *)>(dlsym(h, "libdeflate_free_compressor"));
            l.crc32 = reinterpret_cast<uint32_t (*)(uint32_t, const void *, size_t)>(dlsym(h, "libdeflate_crc32"));
            l.ok = l.alloc && l.compress && l.bound && l.free_ && l.crc32;
            return l;
        }
    };
    const LibDeflate *ld_ = nullptr;
    void *comp_ = nullptr;
    z_stream zs_;
    bool zinit_ = false;
    int level_;
};

// CRC32 of a block's uncompressed bytes (the BGZF trailer): libdeflate's (PCLMUL, several GB/s) when the library is
// there, zlib's otherwise.
class BgzfCrc {
  public:
    BgzfCrc() {
        if (env_on("FQTK_NO_LIBDEFLATE")) return;
        void *h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
        if (!h) h = dlopen("libdeflate.so", RTLD_NOW | RTLD_LOCAL);
        if (h) fn_ = reinterpret_cast<uint32_t (*)(uint32_t, const void *, size_t)>(dlsym(h, "libdeflate_crc32"));
    }
    uint32_t operator()(const void *p, size_t n) const {
        if (fn_) return fn_(0, p, n);
        return (uint32_t)::crc32(::crc32(0L, Z_NULL, 0), static_cast<const Bytef *>(p), (uInt)n);
    }
  private:
    uint32_t (*fn_)(uint32_t, const void *, size_t) = nullptr;
};

// One-shot convenience (tests, small buffers).
inline bool bgzf_compress_block(const uint8_t *in, size_t n, int level, std::vector<uint8_t> &out, std::string *err) {
    BlockCompressor c(level);
    return c.compress(in, n, out, err);
}

}  // namespace fqtk_host
