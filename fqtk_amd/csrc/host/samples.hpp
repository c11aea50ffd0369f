// samples.hpp -- sample metadata TSV + table preconditions (host side of SURVEY.md 8a row A11).
// Follows /root/reference/src/lib/samples.rs: Sample::new (:49-57), SampleGroup::from_samples
// (:101-133), from_file (:144-147: headered TSV `sample_id<TAB>barcode`), with the same messages.
#pragma once
#include <fstream>
#include <set>
#include <sstream>
#include <string>
#include <vector>

namespace fqtk_host {

struct Sample {
    std::string sample_id, barcode;
};

inline bool is_valid_iupac(unsigned char b) {   // src/lib/mod.rs:90-92
    switch (b) {
        case 'A': case 'C': case 'G': case 'T': case 'U': case 'M': case 'R': case 'W': case 'S': case 'Y':
        case 'K': case 'V': case 'H': case 'D': case 'B': case 'N': case 'n': case '.': return true;
        default: return false;
    }
}

inline bool validate_samples(const std::vector<Sample> &samples, std::string *err) {
    if (samples.empty()) { *err = "Must provide one or more sample"; return false; }
    std::set<std::string> ids, bcs;
    for (const Sample &s : samples) ids.insert(s.sample_id);
    if (ids.size() != samples.size()) { *err = "Each sample name must be unique, duplicate identified"; return false; }
    for (const Sample &s : samples) bcs.insert(s.barcode);
    if (bcs.size() != samples.size()) { *err = "Each sample barcode must be unique, duplicate identified"; return false; }
    for (const Sample &s : samples)
        if (s.barcode.size() != samples[0].barcode.size()) { *err = "All barcodes must have the same length"; return false; }
    for (const Sample &s : samples) {
        if (s.sample_id.empty()) { *err = "Sample name cannot be empty"; return false; }
        if (s.barcode.empty()) { *err = "Sample barcode cannot be empty"; return false; }
        for (unsigned char c : s.barcode)
            if (!is_valid_iupac(c)) {
                *err = "All sample barcode bases must be one of A, C, G, T, U, R, Y, S, W, K, M, D, V, H, B, N";
                return false;
            }
    }
    return true;
}

inline bool load_samples(const std::string &path, std::vector<Sample> *out, std::string *err) {
    std::ifstream in(path);
    if (!in) { *err = "Error reading sample metadata " + path + ": No such file or directory (os error 2)"; return false; }
    std::vector<std::string> lines;
    std::string line;
    while (std::getline(in, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        lines.push_back(line);
    }
    while (!lines.empty() && lines.back().empty()) lines.pop_back();   // samples.rs tests :181-201
    if (lines.empty()) { *err = "Must provide one or more sample"; return false; }   // samples.rs test_reading_empty_file
    if (lines[0] != "sample_id\tbarcode") {
        *err = "Error reading sample metadata: header mismatch: expected `sample_id\tbarcode`, found `" + lines[0] + "`";
        return false;
    }
    out->clear();
    for (size_t i = 1; i < lines.size(); ++i) {
        const size_t tab = lines[i].find('\t');
        if (tab == std::string::npos || lines[i].find('\t', tab + 1) != std::string::npos) {
            *err = "Error reading sample metadata: line " + std::to_string(i + 1) + " does not have 2 fields";
            return false;
        }
        out->push_back({lines[i].substr(0, tab), lines[i].substr(tab + 1)});
    }
    return validate_samples(*out, err);
}

}  // namespace fqtk_host
