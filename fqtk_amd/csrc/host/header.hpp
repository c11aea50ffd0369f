// header.hpp -- FASTQ header rewriting of the demultiplexed records.
//
// SURVEY.md section 8(f) row 2.  Follows ReadSet::write_header_internal
// (/root/reference/src/bin/commands/demux.rs:171-267), pinned by its tests (:2084-2196):
//   @name[:UMI(+UMI..)] <read_num>:<rest of comment | N:0:><B1+B2+..>
#pragma once
#include <string>
#include <string_view>
#include <vector>

namespace fqtk_host {

// Appends the rewritten header (starting with '@', no newline) to `out`.
// Returns false and sets *err where the reference returns an error / panics.
inline bool write_header(std::string &out, size_t read_num, std::string_view header,
                         const std::vector<std::string_view> &sample_barcode_segments,
                         const std::vector<std::string_view> &molecular_barcode_segments, std::string *err) {
    // name / optional comment split at the first space (demux.rs:179-182)
    std::string_view name = header, comment;
    bool has_comment = false;
    const size_t sp = header.find(' ');
    if (sp != std::string_view::npos) {
        name = header.substr(0, sp);
        comment = header.substr(sp + 1);
        has_comment = true;
    }
    out.push_back('@');
    if (!molecular_barcode_segments.empty()) {   // demux.rs:188-211
        size_t sep_count = 0;
        for (char c : name) sep_count += c == ':';
        if (sep_count > 7) {
            *err = "Can't handle read name with more than 8 segments: " + std::string(header);
            return false;
        }
        out.append(name);
        out.push_back(sep_count == 7 ? '+' : ':');
        for (size_t i = 0; i < molecular_barcode_segments.size(); ++i) {
            if (i) out.push_back('+');
            out.append(molecular_barcode_segments[i]);
        }
    } else {
        out.append(name);
    }
    out.push_back(' ');
    if (!has_comment) {   // demux.rs:218-222
        out.append(std::to_string(read_num));
        out.append(":N:0:");
    } else {
        size_t sep_count = 0;
        for (char c : comment) sep_count += c == ':';
        if (comment.empty()) {   // the reference unwraps chars.last() here and panics
            *err = "Empty comment in FASTQ header: " + std::string(header);
            return false;
        }
        if (sep_count < 3) {   // demux.rs:227-232
            out.append(comment);
            if (comment.back() != ':') out.push_back(':');
        } else {
            if (sep_count != 3) {
                *err = "Comment in did not have 4 segments: " + std::string(header);
                return false;
            }
            const size_t first_colon = comment.find(':');
            // Illumina can place a "0" in the index position of unmatched FASTQs (demux.rs:241-246)
            const bool last_digit = comment.back() >= '0' && comment.back() <= '9';
            std::string_view remainder =
                comment.substr(first_colon + 1, comment.size() - (first_colon + 1) - (last_digit ? 1 : 0));
            out.append(std::to_string(read_num));
            out.push_back(':');
            out.append(remainder);
            if (remainder.empty()) {   // reference: remainder.last().unwrap() would panic
                *err = "Malformed comment in FASTQ header: " + std::string(header);
                return false;
            }
            if (remainder.back() != ':') out.push_back('+');
        }
    }
    for (size_t i = 0; i < sample_barcode_segments.size(); ++i) {   // demux.rs:258-264
        if (i) out.push_back('+');
        out.append(sample_barcode_segments[i]);
    }
    return true;
}

}  // namespace fqtk_host
