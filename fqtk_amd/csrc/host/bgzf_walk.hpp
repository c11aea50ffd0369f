// bgzf_walk.hpp -- the members of a BGZF file as they lie in it (the feeder threads of `fqtk demux`, demux.cpp; CPU-tested
// through host_capi.cpp).
#pragma once
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../../../include/fqtk_inflate.h"

namespace fqtk_host {

// A BGZF input whose members go to the device as they are (fqtk_demuxer_feed): the file is mapped, this walks the member
// headers (18 bytes: gzip header with the 'BC' extra field, BSIZE) and trailers (CRC-32, ISIZE) and hands out runs of
// whole members.  Nothing is inflated here.
struct BgzfFile {
    std::string path;
    int fd = -1;
    const uint8_t *map = nullptr;
    size_t size = 0, pos = 0;
    BgzfFile() = default;
    BgzfFile(const BgzfFile &) = delete;
    BgzfFile &operator=(const BgzfFile &) = delete;
    ~BgzfFile() { close_file(); }
    void close_file() {
        if (map) munmap(const_cast<uint8_t *>(map), size);
        if (fd >= 0) ::close(fd);
        map = nullptr;
        fd = -1;
        size = pos = 0;
    }
    bool open(const std::string &p, std::string *err) {
        close_file();
        path = p;
        fd = ::open(p.c_str(), O_RDONLY | O_CLOEXEC);
        struct stat st;
        if (fd < 0 || fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || st.st_size <= 0) { *err = "Error opening input files for reading: " + p; close_file(); return false; }
        void *m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) { *err = "Error opening input files for reading: " + p; close_file(); return false; }
        map = static_cast<const uint8_t *>(m);
        size = (size_t)st.st_size;
        madvise(m, size, MADV_SEQUENTIAL);
        return true;
    }
    // a standard BGZF member's header?  (every member is looked at as it is walked: a member that is none ends the run before it)
    static bool looks_like_bgzf(const uint8_t *h, size_t n) {
        return n >= 18 && h[0] == 0x1f && h[1] == 0x8b && h[2] == 8 && h[3] == 4 && h[10] == 6 && h[11] == 0 && h[12] == 'B' && h[13] == 'C' && h[14] == 2 && h[15] == 0;
    }
    // Members from pos on while they are BGZF members and the run stays below the limits (at least one).  [*from, *upto): their bytes in the map.
    bool next_run(size_t max_bytes, size_t max_text, std::vector<fqtk_inflate_member> *out, size_t *from, size_t *upto, std::string *err) {
        out->clear();
        *from = pos;
        size_t text = 0;
        while (pos < size) {
            const uint8_t *h = map + pos;
            if (!looks_like_bgzf(h, size - pos)) {
                // a gzip member without the BC field (`cat a.bgz b.gz`), or whatever else lies behind the BGZF members: the run ends here and the
                // caller looks at it (a serial member is decoded in chunks; what is no gzip member is ignored, as gzread does)
                if (!out->empty()) break;
                *err = "Unexpected error parsing FASTQs: " + path + " holds no BGZF member at byte " + std::to_string(pos);
                return false;
            }
            const size_t bsize = (size_t)h[16] + ((size_t)h[17] << 8) + 1;
            if (bsize < 26 || pos + bsize > size) { *err = "Unexpected error parsing FASTQs: bad BGZF block size in " + path; return false; }
            uint32_t crc, isize;
            std::memcpy(&crc, h + bsize - 8, 4);
            std::memcpy(&isize, h + bsize - 4, 4);
            if (isize > 65536) { *err = "Unexpected error parsing FASTQs: bad BGZF block (more than 64 KiB of text) in " + path; return false; }
            if (!out->empty() && (pos + bsize - *from > max_bytes || text + isize > max_text)) break;
            fqtk_inflate_member m;
            std::memset(&m, 0, sizeof m);
            m.payload_off = pos + 18 - *from;
            m.payload_len = (uint32_t)(bsize - 26);
            m.isize = isize;
            m.crc = crc;
            out->push_back(m);
            text += isize;
            pos += bsize;
        }
        *upto = pos;
        return true;
    }
    bool at_end() const { return pos >= size; }
};

// Length of the gzip member header at p (RFC 1952 2.3; fast_inflate.hpp parses the same fields), 0 if there is none.
inline size_t gzip_header_len(const uint8_t *p, size_t n) {
    if (n < 18 || p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || (p[3] & 0xE0)) return 0;
    const uint8_t flg = p[3];
    size_t at = 10;
    if (flg & 4) {
        if (at + 2 > n) return 0;
        at += 2 + ((size_t)p[at] | ((size_t)p[at + 1] << 8));
    }
    for (int f = 8; f <= 16; f <<= 1) {
        if (!(flg & f)) continue;
        const void *z = at < n ? std::memchr(p + at, 0, n - at) : nullptr;
        if (!z) return 0;
        at = (size_t)(static_cast<const uint8_t *>(z) - p) + 1;
    }
    if (flg & 2) at += 2;
    return at < n ? at : 0;
}


}  // namespace fqtk_host
