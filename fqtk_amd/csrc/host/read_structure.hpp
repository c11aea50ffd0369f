// read_structure.hpp -- read structures ("8B92T", "+T", "10M8B7C100T") and segment extraction.
//
// SURVEY.md section 8(f) row 1.  The reference takes these from the `read-structure` crate 0.2.0
// (source not under /root/reference); behaviour is pinned by the reference's call sites and tests:
//   parse + Display round trip      demux.rs:607,1114-1119,1985 ("+T" in the panic text)
//   segment kinds T B M C S         demux.rs:523-530
//   only the LAST segment may be +  demux.rs:531-533
//   min length = sum(fixed) + 1/+   demux.rs:298
//   extraction                      demux.rs:316-336 (ReadSegment::extract_bases_and_quals)
// Unpinned by the reference (documented choices): lower-case operators are accepted and upper-cased,
// embedded whitespace is ignored (fgbio's ReadStructure does both); bases beyond a fully fixed
// structure are ignored.
#pragma once
#include <cctype>
#include <cstdint>
#include <string>
#include <vector>

namespace fqtk_host {

enum class SegType : char { Template = 'T', SampleBarcode = 'B', MolecularBarcode = 'M', CellularBarcode = 'C', Skip = 'S' };

inline bool seg_type_from_char(char c, SegType *out) {
    switch (c) {
        case 'T': *out = SegType::Template; return true;
        case 'B': *out = SegType::SampleBarcode; return true;
        case 'M': *out = SegType::MolecularBarcode; return true;
        case 'C': *out = SegType::CellularBarcode; return true;
        case 'S': *out = SegType::Skip; return true;
        default: return false;
    }
}

struct ReadSegment {
    size_t offset = 0;      // start within the read
    int64_t length = -1;    // -1 = '+' (all remaining bases)
    SegType kind = SegType::Template;
    bool has_length() const { return length >= 0; }
    std::string to_string() const {
        return (has_length() ? std::to_string(length) : std::string("+")) + std::string(1, (char)kind);
    }
};

struct ReadStructure {
    std::vector<ReadSegment> segments;

    // Returns false and sets *err on malformed input.
    static bool parse(const std::string &text, ReadStructure *out, std::string *err) {
        std::string s;
        for (char c : text)
            if (!std::isspace((unsigned char)c)) s.push_back((char)std::toupper((unsigned char)c));
        out->segments.clear();
        if (s.empty()) { *err = "Read structure contained zero elements"; return false; }
        size_t i = 0, offset = 0;
        while (i < s.size()) {
            ReadSegment seg;
            seg.offset = offset;
            if (s[i] == '+') {
                seg.length = -1;
                ++i;
            } else if (std::isdigit((unsigned char)s[i])) {
                uint64_t v = 0;
                while (i < s.size() && std::isdigit((unsigned char)s[i])) {
                    v = v * 10 + (uint64_t)(s[i] - '0');
                    if (v > (1ull << 40)) { *err = "Read structure segment length too large: " + text; return false; }
                    ++i;
                }
                if (v == 0) { *err = "Read structure contained a zero-length segment: " + text; return false; }
                seg.length = (int64_t)v;
            } else {
                *err = "Read structure missing length information: " + text + " at position " + std::to_string(i);
                return false;
            }
            if (i >= s.size()) { *err = "Read structure missing operator: " + text; return false; }
            if (!seg_type_from_char(s[i], &seg.kind)) {
                *err = std::string("Read structure had unknown type: ") + s[i] + " in " + text;
                return false;
            }
            ++i;
            if (!seg.has_length() && i < s.size()) {
                *err = "Read structure had a non-terminal indefinite length (+) segment: " + text;
                return false;
            }
            if (seg.has_length()) offset += (size_t)seg.length;
            out->segments.push_back(seg);
        }
        return true;
    }

    std::string to_string() const {
        std::string r;
        for (const ReadSegment &s : segments) r += s.to_string();
        return r;
    }

    // demux.rs:298: sum of fixed lengths, one base for a variable-length segment
    size_t min_length() const {
        size_t n = 0;
        for (const ReadSegment &s : segments) n += s.has_length() ? (size_t)s.length : 1;
        return n;
    }

    size_t count(SegType t) const {
        size_t n = 0;
        for (const ReadSegment &s : segments) n += s.kind == t;
        return n;
    }
};

// [begin, end) of a segment within a read of `read_len` bases (read_len >= structure.min_length()).
inline void segment_span(const ReadSegment &seg, size_t read_len, size_t *begin, size_t *end) {
    *begin = seg.offset;
    *end = seg.has_length() ? seg.offset + (size_t)seg.length : read_len;
    if (*end > read_len) *end = read_len;
    if (*begin > *end) *begin = *end;
}

}  // namespace fqtk_host
