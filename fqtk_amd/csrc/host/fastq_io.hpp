// fastq_io.hpp -- gz-aware FASTQ reading in record batches.
//
// SURVEY.md section 8(f) row 3 (input side).  The reference reads through fgoxide's Io::new_reader
// (gz-aware, 1 MiB buffer; /root/reference/src/bin/commands/demux.rs:844-849) and seq_io's
// fastq::Reader (demux.rs:16-17,289-294,891): four-line records, `head` = line 1 without '@'.
// zlib's gzread transparently handles plain files, gzip, multi-member gzip and BGZF.
// Unpinned by the reference's tests (choices here): a trailing '\r' is stripped from every line;
// multi-line FASTQ is not supported (seq_io's fastq reader does not support it either).
#pragma once
#include <zlib.h>

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <memory>
#include <string>
#include <string_view>
#include <vector>

namespace fqtk_host {

struct FastqRec {
    uint32_t head_off, head_len;   // header without the leading '@'
    uint32_t seq_off, seq_len;
    uint32_t qual_off;             // qual_len == seq_len
};

struct RecBatch {
    std::vector<char> data;
    std::vector<FastqRec> recs;
    const char *head(size_t i) const { return data.data() + recs[i].head_off; }
    const char *seq(size_t i) const { return data.data() + recs[i].seq_off; }
    const char *qual(size_t i) const { return data.data() + recs[i].qual_off; }
};

class FastqSource {
  public:
    ~FastqSource() { if (gz_) gzclose(gz_); }
    bool open(const std::string &path, std::string *err) {
        gz_ = gzopen(path.c_str(), "rb");
        if (!gz_) { *err = "Error opening input files for reading: " + path; return false; }
        gzbuffer(gz_, 1 << 20);
        path_ = path;
        buf_.resize(4 << 20);
        return true;
    }
    // Reads up to max_records records.  Returns false on a malformed file (*err set).  An empty batch = EOF.
    bool next_batch(size_t max_records, RecBatch *out, std::string *err) {
        out->data.clear();
        out->recs.clear();
        out->recs.reserve(max_records);
        while (out->recs.size() < max_records) {
            std::string_view line[4];
            size_t consumed = 0;
            int got = 0;
            // need four complete lines starting at pos_
            for (;;) {
                got = 0;
                size_t p = pos_;
                while (got < 4) {
                    const char *nl = (const char *)memchr(buf_.data() + p, '\n', end_ - p);
                    if (!nl) break;
                    size_t len = (size_t)(nl - (buf_.data() + p));
                    line[got] = std::string_view(buf_.data() + p, len);
                    p += len + 1;
                    ++got;
                }
                if (got == 4) { consumed = p - pos_; break; }
                if (eof_) {
                    // final line without '\n'
                    if (got == 3 && p < end_) {
                        line[3] = std::string_view(buf_.data() + p, end_ - p);
                        got = 4;
                        consumed = end_ - pos_;
                    }
                    break;
                }
                if (!fill(err)) return false;
            }
            if (got < 4) {
                // EOF: anything left must be blank
                for (size_t q = pos_; q < end_; ++q)
                    if (buf_[q] != '\n' && buf_[q] != '\r') {
                        *err = "Unexpected error parsing FASTQs: truncated record at end of " + path_;
                        return false;
                    }
                pos_ = end_;
                break;
            }
            for (auto &l : line)
                if (!l.empty() && l.back() == '\r') l.remove_suffix(1);
            if (line[0].empty() || line[0][0] != '@') {
                *err = "Unexpected error parsing FASTQs: expected '@' at record " + std::to_string(nrec_) + " of " + path_;
                return false;
            }
            if (line[2].empty() || line[2][0] != '+') {
                *err = "Unexpected error parsing FASTQs: expected '+' at record " + std::to_string(nrec_) + " of " + path_;
                return false;
            }
            if (line[1].size() != line[3].size()) {
                *err = "Unexpected error parsing FASTQs: sequence and quality lengths differ at record " +
                       std::to_string(nrec_) + " of " + path_;
                return false;
            }
            // one copy per record: the four lines are contiguous in the read buffer, so the whole span from
            // the header (after '@') to the end of the quality line goes over as is; offsets skip the rest
            FastqRec r;
            const char *rec0 = line[0].data() + 1;
            const size_t span = (size_t)(line[3].data() + line[3].size() - rec0);
            const size_t base = out->data.size();
            if (base == 0) out->data.reserve(std::min<size_t>(max_records, 1u << 20) * (span + 16));
            out->data.insert(out->data.end(), rec0, rec0 + span);
            r.head_off = (uint32_t)base;
            r.head_len = (uint32_t)line[0].size() - 1;
            r.seq_off = (uint32_t)(base + (size_t)(line[1].data() - rec0));
            r.seq_len = (uint32_t)line[1].size();
            r.qual_off = (uint32_t)(base + (size_t)(line[3].data() - rec0));
            out->recs.push_back(r);
            pos_ += consumed;
            ++nrec_;
        }
        return true;
    }

  private:
    bool fill(std::string *err) {
        if (pos_ > 0) {   // compact
            memmove(buf_.data(), buf_.data() + pos_, end_ - pos_);
            end_ -= pos_;
            pos_ = 0;
        }
        if (end_ == buf_.size()) buf_.resize(buf_.size() * 2);   // one very long line
        int n = gzread(gz_, buf_.data() + end_, (unsigned)std::min<size_t>(buf_.size() - end_, 1u << 30));
        if (n < 0) {
            int e = 0;
            *err = std::string("Unexpected error parsing FASTQs: ") + gzerror(gz_, &e) + " in " + path_;
            return false;
        }
        if (n == 0) eof_ = true;
        end_ += (size_t)n;
        return true;
    }
    gzFile gz_ = nullptr;
    std::string path_;
    std::vector<char> buf_;
    size_t pos_ = 0, end_ = 0;
    bool eof_ = false;
    uint64_t nrec_ = 0;
};

}  // namespace fqtk_host
