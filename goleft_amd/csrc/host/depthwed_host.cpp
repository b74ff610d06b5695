// depthwed_host.cpp -- host twin of `goleft depthwed` (text in, text out), the
// consumer of `*.depth.bed` files (/root/reference/depthwed/depthwed.go).
//
//   goleft-depth depthwed -s SIZE a.depth.bed b.depth.bed ...
//
// One record is read from every file in lockstep (depthwed.go:128-153); records
// are accumulated until the span of the first file's group reaches SIZE or its
// next record is on another chromosome (:126); each constituent record adds
// int(0.5 + mean) (:103); the printed cell is the SUM of those integers (:68,:151).
// The device-side twin (gd_depthwed, include/goleft_depth.h) builds the same
// matrix from integer window sums without any text; gd_round4g.hpp is the shared
// rounding step and is unit-tested through gdh_depthwed_cells below.
#include <zlib.h>

#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../../include/goleft_depth_host.h"
#include "../gd_round4g.hpp"

namespace {

struct BedReader {
    gzFile f = nullptr;
    std::string path;
    std::string pending;      // next line (without '\n'), valid when has_pending
    bool has_pending = false;
    bool at_eof = false;

    bool fill()
    {
        if (has_pending || at_eof) return has_pending;
        pending.clear();
        char buf[4096];
        for (;;) {
            if (!gzgets(f, buf, sizeof buf)) {
                at_eof = true;
                if (pending.empty()) return false;
                break;                               // last line without newline
            }
            const size_t n = strlen(buf);
            if (n && buf[n - 1] == '\n') { pending.append(buf, n - 1); break; }
            pending.append(buf, n);
        }
        has_pending = true;
        return true;
    }
    // depthwed.go:108-115 getNextChrom: text before the first tab of the next line, "" at EOF
    std::string next_chrom()
    {
        if (!fill()) return std::string();
        const size_t t = pending.find('\t');
        return t == std::string::npos ? pending : pending.substr(0, t);
    }
    bool read_line(std::string* out)
    {
        if (!fill()) return false;
        out->swap(pending);
        has_pending = false;
        return true;
    }
};

struct Depth {
    std::string chrom;
    long long start = 0, end = 0, depth = 0;
};

// depthwed.go:37-46
std::string name_from_file(const std::string& f)
{
    std::string n = f.substr(f.find_last_of('/') == std::string::npos ? 0 : f.find_last_of('/') + 1);
    for (const char* suff : {".gz", ".bed", ".depth"}) {
        const size_t k = strlen(suff);
        if (n.size() >= k && n.compare(n.size() - k, k, suff) == 0) n.resize(n.size() - k);
    }
    return n;
}

// depthwed.go:93-106 sFromLine
bool parse_line(const std::string& l, Depth* d)
{
    size_t a = l.find('\t');
    if (a == std::string::npos) return false;
    size_t b = l.find('\t', a + 1);
    if (b == std::string::npos) return false;
    size_t c = l.find('\t', b + 1);
    if (c == std::string::npos) return false;
    size_t e = l.find('\t', c + 1);
    d->chrom = l.substr(0, a);
    char* endp = nullptr;
    errno = 0;
    d->start = strtoll(l.c_str() + a + 1, &endp, 10);
    if (endp != l.c_str() + b) return false;
    d->end = strtoll(l.c_str() + b + 1, &endp, 10);
    if (endp != l.c_str() + c) return false;
    const std::string tok = l.substr(c + 1, e == std::string::npos ? std::string::npos : e - c - 1);
    const double dep = strtod(tok.c_str(), &endp);
    if (endp == tok.c_str() || *endp != '\0') return false;
    d->depth = (long long)(0.5 + dep);               // depthwed.go:103
    return true;
}

int run(long long size, const std::vector<std::string>& paths, FILE* out)
{
    std::vector<BedReader> beds(paths.size());
    for (size_t i = 0; i < paths.size(); ++i) {
        beds[i].path = paths[i];
        beds[i].f = gzopen(paths[i].c_str(), "rb");
        if (!beds[i].f) {
            fprintf(stderr, "goleft depthwed: open %s: %s\n", paths[i].c_str(), strerror(errno));
            for (auto& b : beds) if (b.f) gzclose(b.f);
            return 1;
        }
    }
    fputs("#chrom\tstart\tend", out);
    for (const auto& p : paths) fprintf(out, "\t%s", name_from_file(p).c_str());
    fputc('\n', out);

    int rc = 0;
    bool warned_chrom = false;
    std::string warned_for;
    for (;;) {
        // next(): depthwed.go:117-157
        std::vector<Depth> depths(beds.size());
        bool eof = false;
        int k = 0;
        const std::string chrom = beds[0].next_chrom();
        if (chrom != warned_for) { warned_chrom = false; warned_for = chrom; }
        while (!eof && depths[0].end - depths[0].start < size && chrom == beds[0].next_chrom()) {
            for (size_t i = 0; i < beds.size(); ++i) {
                std::string line;
                if (!beds[i].read_line(&line)) {
                    if (i > 0 && !eof) {
                        fprintf(stderr, "goleft depthwed: not all files have same number of records\n");
                        rc = 2;             // the reference panics here
                        goto done;
                    }
                    eof = true;
                    continue;
                }
                Depth tmp;
                if (!parse_line(line, &tmp)) {
                    fprintf(stderr, "goleft depthwed: bad record in %s: %s\n", beds[i].path.c_str(), line.c_str());
                    rc = 1;
                    goto done;
                }
                if (k == 0) {
                    depths[i] = tmp;
                    if (tmp.chrom != chrom) {
                        fprintf(stderr, "goleft depthwed: got unexpected chromosome from %s: %s\n",
                                beds[i].path.c_str(), tmp.chrom.c_str());
                        rc = 1;
                        goto done;
                    }
                    if (tmp.end > tmp.start && size % (tmp.end - tmp.start) != 0 && !warned_chrom) {
                        warned_chrom = true;
                        fprintf(stderr, "size %lld indivisible by interval in line: %s likely chromosome change.\n",
                                size, line.c_str());
                    }
                } else {
                    depths[i].end = tmp.end;
                    depths[i].depth += tmp.depth;
                }
            }
            ++k;
        }
        if (eof) break;
        fprintf(out, "%s\t%lld\t%lld", depths[0].chrom.c_str(), depths[0].start, depths[0].end);
        for (const auto& d : depths) fprintf(out, "\t%lld", d.depth);
        fputc('\n', out);
    }
done:
    for (auto& b : beds) if (b.f) gzclose(b.f);
    if (fflush(out) != 0) rc = rc ? rc : 1;
    return rc;
}

}  // namespace

extern "C" {

int gdh_depthwed_run(int64_t size, const char* const* paths, int n_paths, const char* out_path)
{
    if (size < 1 || !paths || n_paths < 1) return 255;
    std::vector<std::string> p(paths, paths + n_paths);
    FILE* out = out_path ? fopen(out_path, "w") : stdout;
    if (!out) return 1;
    const int rc = run(size, p, out);
    if (out_path && fclose(out) != 0) return rc ? rc : 1;
    return rc;
}

int gdh_depthwed_main(int argc, const char* const* argv)
{
    // go-arg: -s/--size required int, positional beds required (depthwed.go:18-21)
    long long size = -1;
    std::vector<const char*> beds;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        if (a == "-h" || a == "--help") {
            printf("usage: goleft depthwed --size SIZE BEDS [BEDS ...]\n\n"
                   "positional arguments:\n  beds                   depth.bed files from goleft depth\n\n"
                   "options:\n  --size SIZE, -s SIZE   sizes of windows to aggregate to must be >= window in input files.\n");
            return 0;
        }
        if (a == "-s" || a == "--size" || a.rfind("--size=", 0) == 0) {
            const char* v = nullptr;
            if (a.rfind("--size=", 0) == 0) v = argv[i] + 7;
            else if (i + 1 < argc) v = argv[++i];
            char* e = nullptr;
            if (v) size = strtoll(v, &e, 10);
            if (!v || e == v || *e) { fprintf(stderr, "error: error processing --size\n"); return 255; }
        } else if (a.size() > 1 && a[0] == '-') {
            fprintf(stderr, "error: unknown argument %s\n", a.c_str());
            return 255;
        } else {
            beds.push_back(argv[i]);
        }
    }
    if (size < 0) { fprintf(stderr, "error: --size is required\n"); return 255; }
    if (beds.empty()) { fprintf(stderr, "error: beds is required\n"); return 255; }
    if (size < 1) { fprintf(stderr, "error: --size must be >= 1\n"); return 255; }
    return gdh_depthwed_run(size, beds.data(), (int)beds.size(), nullptr);
}

void gdh_depthwed_cells(const int64_t* sums, const int64_t* lens, size_t n, int64_t* out)
{
    for (size_t i = 0; i < n; ++i) out[i] = gd_depthwed_cell(sums[i], lens[i]);
}

}  // extern "C"
