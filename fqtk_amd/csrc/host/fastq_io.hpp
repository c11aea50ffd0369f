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
        return true;
    }
    // Reads up to max_records records.  Returns false on a malformed file (*err set).  An empty batch = EOF.
    //
    // Parsed IN PLACE: gzread fills the batch's own data vector, lines are located with memchr and a
    // record is four (offset, length) pairs into that vector -- no per-record copy.  Bytes read past the
    // last record of this batch (at most one read piece) are carried into the next batch.
    bool next_batch(size_t max_records, RecBatch *out, std::string *err) {
        out->recs.clear();
        out->recs.reserve(std::min<size_t>(max_records, 1u << 20));
        std::vector<char> &d = out->data;
        d.clear();
        d.reserve(cap_hint_);           // batches of one input are alike: no regrowth copies after the first
        d.insert(d.end(), carry_.begin(), carry_.end());   // bytes already read that belong to this batch
        size_t pos = 0;                 // first unparsed byte
        carry_.clear();
        while (out->recs.size() < max_records) {
            // four complete lines starting at pos?
            size_t lo[4], len[4], p = pos;
            int got = 0;
            while (got < 4) {
                const char *nl = p < d.size() ? (const char *)memchr(d.data() + p, '\n', d.size() - p) : nullptr;
                if (!nl) break;
                lo[got] = p;
                len[got] = (size_t)(nl - (d.data() + p));
                p += len[got] + 1;
                ++got;
            }
            if (got < 4) {
                if (!eof_) {            // need more bytes: append one piece to the batch
                    if (!fill(d, err)) return false;
                    continue;
                }
                if (got == 3 && p < d.size()) {   // final line without '\n'
                    lo[3] = p;
                    len[3] = d.size() - p;
                    p = d.size();
                    got = 4;
                } else {
                    // EOF: anything left must be blank
                    for (size_t q = pos; q < d.size(); ++q)
                        if (d[q] != '\n' && d[q] != '\r') {
                            *err = "Unexpected error parsing FASTQs: truncated record at end of " + path_;
                            return false;
                        }
                    pos = d.size();
                    break;
                }
            }
            for (int k = 0; k < 4; ++k)
                if (len[k] > 0 && d[lo[k] + len[k] - 1] == '\r') --len[k];
            if (len[0] == 0 || d[lo[0]] != '@') {
                *err = "Unexpected error parsing FASTQs: expected '@' at record " + std::to_string(nrec_) + " of " + path_;
                return false;
            }
            if (len[2] == 0 || d[lo[2]] != '+') {
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
        // what was read beyond this batch's last record starts the next batch
        carry_.assign(d.begin() + (std::ptrdiff_t)pos, d.end());
        cap_hint_ = std::max(cap_hint_, d.size() + (8u << 20));
        d.resize(pos);
        return true;
    }

  private:
    bool fill(std::vector<char> &d, std::string *err) {
        const size_t piece = 4u << 20;
        const size_t old = d.size();
        d.resize(old + piece);
        int n = gzread(gz_, d.data() + old, (unsigned)piece);
        if (n < 0) {
            int e = 0;
            *err = std::string("Unexpected error parsing FASTQs: ") + gzerror(gz_, &e) + " in " + path_;
            return false;
        }
        if (n == 0) eof_ = true;
        d.resize(old + (size_t)n);
        return true;
    }
    gzFile gz_ = nullptr;
    std::string path_;
    std::vector<char> carry_;
    size_t cap_hint_ = 8u << 20;
    bool eof_ = false;
    uint64_t nrec_ = 0;
};

}  // namespace fqtk_host
