// metrics.hpp -- demux-metrics.txt.
// Follows DemuxMetric (/root/reference/src/bin/commands/demux.rs:452-497) and its TSV output
// (:994-998): columns sample_id, barcode, templates, frac_templates, ratio_to_mean, ratio_to_best;
// unmatched row last with barcode "."; mean/best exclude the unmatched pseudo-sample.
// Float TEXT formatting is not pinned by the reference's tests (only `templates` is read back,
// demux.rs:2059-2064); shortest round-trip decimals in ryu style ("1.0", "0.25", "NaN", "inf") are
// emitted, which is what the csv crate produces.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

namespace fqtk_host {

inline std::string format_f64(double v) {
    if (std::isnan(v)) return "NaN";
    if (std::isinf(v)) return v < 0 ? "-inf" : "inf";
    char buf[64];
    for (int prec = 1; prec <= 17; ++prec) {
        std::snprintf(buf, sizeof buf, "%.*g", prec, v);
        if (std::strtod(buf, nullptr) == v) break;
    }
    std::string s(buf);
    const size_t e = s.find('e');
    if (e != std::string::npos) {   // C prints e+16 / e-05; ryu prints e16 / e-5
        std::string mant = s.substr(0, e), ex = s.substr(e + 1);
        bool neg = !ex.empty() && ex[0] == '-';
        if (!ex.empty() && (ex[0] == '+' || ex[0] == '-')) ex.erase(0, 1);
        while (ex.size() > 1 && ex[0] == '0') ex.erase(0, 1);
        return mant + "e" + (neg ? "-" : "") + ex;
    }
    if (s.find('.') == std::string::npos) s += ".0";
    return s;
}

struct DemuxMetric {
    std::string sample_id, barcode;
    uint64_t templates = 0;
    double frac_templates = 0, ratio_to_mean = 0, ratio_to_best = 0;
};

// demux.rs:481-496
inline void update_metrics(std::vector<DemuxMetric> &samples, DemuxMetric &unmatched) {
    double sample_total = 0, best = 0;
    for (const DemuxMetric &m : samples) {
        sample_total += (double)m.templates;
        if ((double)m.templates > best) best = (double)m.templates;
    }
    const double total = sample_total + (double)unmatched.templates;
    const double mean = sample_total / (double)samples.size();
    auto upd = [&](DemuxMetric &m) {
        m.frac_templates = (double)m.templates / total;
        m.ratio_to_mean = (double)m.templates / mean;
        m.ratio_to_best = (double)m.templates / best;
    };
    for (DemuxMetric &m : samples) upd(m);
    upd(unmatched);
}

inline bool write_metrics_tsv(const std::string &path, const std::vector<DemuxMetric> &rows, std::string *err) {
    FILE *f = std::fopen(path.c_str(), "w");
    if (!f) { *err = "cannot write " + path; return false; }
    std::fputs("sample_id\tbarcode\ttemplates\tfrac_templates\tratio_to_mean\tratio_to_best\n", f);
    for (const DemuxMetric &m : rows)
        std::fprintf(f, "%s\t%s\t%llu\t%s\t%s\t%s\n", m.sample_id.c_str(), m.barcode.c_str(),
                     (unsigned long long)m.templates, format_f64(m.frac_templates).c_str(),
                     format_f64(m.ratio_to_mean).c_str(), format_f64(m.ratio_to_best).c_str());
    std::fclose(f);
    return true;
}

}  // namespace fqtk_host
