// fasta_stats.hpp -- `--stats` columns (GC, CpG, masked fraction per window).
//
// The reference appends "\t%.3g\t%.3g\t%.3g" of faidx.Stats(chrom, s, e)
// (/root/reference/depth/depth.go:191-200).  faidx is an external module
// (github.com/brentp/faidx @c39eb85, go.mod:12) that is not in /root/reference,
// and no reference test asserts these values: PARITY UNPINNED.  Semantics
// restated from the module's documentation: GC = fraction of G/C (either
// case), masked = fraction of lower-case bases, CpG = 2 * (#C followed by G,
// looking one base past the window) / window length.
#pragma once

#include <cstdint>
#include <cstdio>
#include <map>
#include <string>

namespace gdh {

class FastaStats {
public:
    ~FastaStats() { if (fp_) fclose(fp_); }

    bool open(const std::string& fasta, std::string* err)
    {
        FILE* fi = fopen((fasta + ".fai").c_str(), "r");
        if (!fi) { if (err) *err = "cannot open " + fasta + ".fai"; return false; }
        char name[4096];
        long long len, off, lb, lw;
        char line[8192];
        while (fgets(line, sizeof line, fi)) {
            if (sscanf(line, "%4095[^\t]\t%lld\t%lld\t%lld\t%lld", name, &len, &off, &lb, &lw) == 5)
                idx_[name] = Entry{len, off, lb, lw};
        }
        fclose(fi);
        fp_ = fopen(fasta.c_str(), "rb");
        if (!fp_) { if (err) *err = "cannot open " + fasta; return false; }
        return true;
    }

    // "\tGC\tCpG\tMasked" formatted like the reference (%.3g each)
    std::string stats_columns(const std::string& chrom, int64_t start, int64_t end)
    {
        double gc = 0, cpg = 0, masked = 0;
        auto it = idx_.find(chrom);
        if (it != idx_.end() && end > start) {
            const Entry& e = it->second;
            int64_t n_gc = 0, n_cpg = 0, n_mask = 0;
            const int64_t stop = end < e.len ? end + 1 : e.len;    // one look-ahead base
            std::string seq;
            fetch(e, start < e.len ? start : e.len, stop, &seq);
            const int64_t n = (int64_t)seq.size() < end - start ? (int64_t)seq.size() : end - start;
            for (int64_t i = 0; i < n; ++i) {
                const char c = seq[(size_t)i];
                if (c == 'G' || c == 'C' || c == 'g' || c == 'c') ++n_gc;
                if (c >= 'a' && c <= 'z') ++n_mask;
                if ((c == 'C' || c == 'c') && (size_t)(i + 1) < seq.size() &&
                    (seq[(size_t)i + 1] == 'G' || seq[(size_t)i + 1] == 'g')) ++n_cpg;
            }
            const double tot = (double)(end - start);
            gc = n_gc / tot; cpg = 2.0 * n_cpg / tot; masked = n_mask / tot;
        }
        char buf[96];
        snprintf(buf, sizeof buf, "\t%.3g\t%.3g\t%.3g", gc, cpg, masked);
        return buf;
    }

private:
    struct Entry { int64_t len, off, lb, lw; };

    void fetch(const Entry& e, int64_t s, int64_t t, std::string* out)
    {
        out->clear();
        if (t <= s || e.lb <= 0) return;
        const int64_t b0 = e.off + (s / e.lb) * e.lw + s % e.lb;
        const int64_t b1 = e.off + ((t - 1) / e.lb) * e.lw + (t - 1) % e.lb + 1;
        std::string raw((size_t)(b1 - b0), '\0');
        if (fseeko(fp_, (off_t)b0, SEEK_SET) != 0) return;
        const size_t got = fread(&raw[0], 1, raw.size(), fp_);
        raw.resize(got);
        for (char c : raw)
            if (c != '\n' && c != '\r') out->push_back(c);
    }

    FILE* fp_ = nullptr;
    std::map<std::string, Entry> idx_;
};

}  // namespace gdh
