// bgzf.hpp -- BGZF block compression for the per-sample output FASTQs.
//
// SURVEY.md section 8(f) row 3 (output side).  The reference writes through pooled-writer's
// BgzfCompressor (/root/reference/src/bin/commands/demux.rs:755-798): independent <= 64 KiB gzip
// members carrying the 'BC' extra field, terminated by the 28-byte EOF block.  The reference's tests
// only compare DECOMPRESSED content (demux.rs:1069-1076), so parity here = a valid BGZF stream whose
// decompressed bytes are identical; the compressed bytes differ (zlib here, libdeflate there).
#pragma once
#include <zlib.h>

#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace fqtk_host {

constexpr size_t kBgzfBlockSize = 65280;   // uncompressed payload per block (as the bgzf crate)

static const uint8_t kBgzfEof[28] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0, 0x42, 0x43,
                                     0x02, 0, 0x1b, 0, 0x03, 0, 0, 0, 0, 0, 0, 0, 0, 0};

// Appends one BGZF block holding in[0..n) (n <= kBgzfBlockSize) to `out`.
inline bool bgzf_compress_block(const uint8_t *in, size_t n, int level, std::vector<uint8_t> &out, std::string *err) {
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) {
        *err = "deflateInit2 failed";
        return false;
    }
    const size_t start = out.size();
    const size_t cap = deflateBound(&zs, (uLong)n) + 64;
    out.resize(start + 18 + cap + 8);
    zs.next_in = const_cast<Bytef *>(in);
    zs.avail_in = (uInt)n;
    zs.next_out = out.data() + start + 18;
    zs.avail_out = (uInt)cap;
    int rc = deflate(&zs, Z_FINISH);
    const size_t clen = zs.total_out;
    deflateEnd(&zs);
    if (rc != Z_STREAM_END) { *err = "deflate failed"; return false; }
    const size_t bsize = 18 + clen + 8;
    if (bsize > 65536) { *err = "BGZF block overflow"; return false; }
    uint8_t *h = out.data() + start;
    const uint8_t hdr[18] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0, 0x42, 0x43, 0x02, 0,
                             (uint8_t)((bsize - 1) & 0xff), (uint8_t)((bsize - 1) >> 8)};
    memcpy(h, hdr, 18);
    const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), in, (uInt)n);
    uint8_t *t = h + 18 + clen;
    t[0] = crc & 0xff; t[1] = (crc >> 8) & 0xff; t[2] = (crc >> 16) & 0xff; t[3] = (crc >> 24) & 0xff;
    t[4] = n & 0xff; t[5] = (n >> 8) & 0xff; t[6] = (n >> 16) & 0xff; t[7] = (n >> 24) & 0xff;
    out.resize(start + bsize);
    return true;
}

}  // namespace fqtk_host
