// fasta_stats.hpp -- FASTA access for the `--stats` columns (GC, CpG, masked
// fraction per window; /root/reference/depth/depth.go:191-200, :244-252).
//
// The reference mmaps the FASTA through faidx (external module, go.mod:12) and
// scans each window on the CPU.  Here the host only reads the .fai and hands a
// contig's bases to the device (gd_seq_load); the per-window base counting is
// gd_seq_stats (csrc/gd_seqstats.hpp) and the three "%.3g" columns are
// formatted in depth_host.cpp.  Semantics: see gd_seqstats.hpp (PARITY UNPINNED).
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

    // All bases of one contig, line breaks removed; false when the contig is not in the .fai.
    // line_bases (optional): the .fai LINEBASES column of the contig.
    bool contig_bases(const std::string& chrom, std::string* out, int64_t* line_bases = nullptr)
    {
        out->clear();
        auto it = idx_.find(chrom);
        if (it == idx_.end()) return false;
        fetch(it->second, 0, it->second.len, out);
        if (line_bases) *line_bases = it->second.lb;
        return true;
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
