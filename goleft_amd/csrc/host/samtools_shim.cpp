// samtools_shim.cpp -- `samtools depth -Q q -d D -r 'chr:s-e' in.bam` served by the MI355X engine.
//
// SURVEY.md section 8(b) option A: the ONE drop-in that needs no change to goleft at all.  The reference runs
//     echo '<region>'; samtools depth -Q %d -d %d -r '<region>' '<bam>'
// per 10 Mb tile (/root/reference/depth/depth.go:45, :392-394) and parses `chrom \t pos(1-based) \t depth` lines
// (getPosDepth, :202-221).  An executable NAMED samtools (goleft_amd/shim/samtools: put goleft_amd/shim first on PATH)
// that answers exactly that invocation lets an unmodified goleft binary run on the engine: BASELINE.json config 1,
// plumbing -- every tile pays a process start, a BAM read and a text print, so this is the compatibility path, not
// the fast one (the fast one is the C ABI: INTEGRATION.md).
//
// Semantics are those the oracle restates for samtools >= 1.13 (oracle/depth_oracle.c): reads with flag & 0x704 or
// MAPQ < Q are dropped, M/=/X count, D/N do not, no base-quality test, -d ignored (no cap); positions of depth 0
// are not printed (-a: they are).  Anything else of samtools is refused with exit code 1.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../../include/goleft_depth.h"
#include "../../../include/goleft_depth_host.h"
#include "bam_reader.hpp"

extern "C" int gdh_samtools_main(int argc, const char* const* argv)
{
    auto die = [](const char* msg) { fprintf(stderr, "samtools (goleft_amd shim): %s\n", msg); return 1; };
    if (argc < 2 || strcmp(argv[1], "depth") != 0)
        return die("only `samtools depth [-a] [-Q mapq] [-d max] [-r region] in.bam` is served by this shim");
    int Q = 0;
    bool all = false;
    std::string region, bam;
    for (int i = 2; i < argc; ++i) {
        const std::string a = argv[i];
        auto value = [&](const char** out) { if (i + 1 >= argc) return false; *out = argv[++i]; return true; };
        const char* v = nullptr;
        if (a == "-Q") { if (!value(&v)) return die("-Q needs a value"); Q = atoi(v); }
        else if (a == "-d" || a == "-m") { if (!value(&v)) return die("-d needs a value"); }      // ignored, as samtools >= 1.13 ignores it
        else if (a == "-r") { if (!value(&v)) return die("-r needs a value"); region = v; }
        else if (a == "-a") all = true;
        else if (a == "-q") { if (!value(&v)) return die("-q needs a value"); if (atoi(v) > 0) return die("-q (base quality) is not served"); }
        else if (!a.empty() && a[0] == '-') return die(("unsupported option " + a).c_str());
        else if (bam.empty()) bam = a;
        else return die("exactly one BAM is served");
    }
    if (bam.empty()) return die("no BAM given");
    gdh::BamReader rd;
    std::string err;
    if (!rd.open(bam, 0, &err)) return die(err.empty() ? "cannot open the BAM" : err.c_str());
    const auto& ctgs = rd.contigs();
    // region: chr | chr:beg-end (1-based, inclusive), contig names may hold colons (HLA-A*01:01:01:01:1-16571)
    std::vector<int32_t> tids;
    int64_t s = 0, e = 0;
    if (region.empty()) {
        for (size_t t = 0; t < ctgs.size(); ++t) tids.push_back((int32_t)t);
    } else {
        std::string chrom = region;
        int32_t tid = -1;
        for (size_t t = 0; t < ctgs.size(); ++t) if (ctgs[t].name == chrom) tid = (int32_t)t;
        if (tid >= 0) { s = 0; e = ctgs[(size_t)tid].length; }
        else {
            char cbuf[4096];
            const std::string line = region + "\n";
            if (gdh_chrom_start_end(line.c_str(), line.size(), cbuf, sizeof cbuf, &s, &e) != 0) return die("cannot parse the region");
            chrom = cbuf;
            for (size_t t = 0; t < ctgs.size(); ++t) if (ctgs[t].name == chrom) tid = (int32_t)t;
            if (tid < 0) return die(("region names an unknown reference: " + chrom).c_str());
            if (e > ctgs[(size_t)tid].length) e = ctgs[(size_t)tid].length;
        }
        tids.push_back(tid);
    }
    gd_ctx* ctx = nullptr;
    const char* dv = getenv("GOLEFT_DEVICE");
    if (gd_create(dv ? atoi(dv) : 0, &ctx) != GD_OK) return die("no usable MI355X device (this build has no CPU path)");
    auto fail = [&](const char* what) { fprintf(stderr, "samtools (goleft_amd shim): %s: %s\n", what, gd_last_error(ctx)); gd_destroy(ctx); return 1; };
    gd_params P;
    gd_default_params(&P);
    P.min_mapq = Q;
    P.window_size = 1000;
    if (gd_set_params(ctx, &P) != GD_OK) return fail("gd_set_params");
    std::vector<int64_t> lens;
    for (const auto& c : ctgs) lens.push_back(c.length);
    if (gd_set_contigs(ctx, (int)lens.size(), lens.data()) != GD_OK) return fail("gd_set_contigs");
    if (gd_select_contigs(ctx, (int)tids.size(), tids.data()) != GD_OK) return fail("gd_select_contigs");
    // records: the reference's run of the file up to the region's end (reads that start before the region may reach in)
    // ... found through the .bai: a reference the index knows to be empty (decoy and alt contigs: most of an assembly's
    // names) is answered at once instead of by inflating the whole file; only WITHOUT a usable index is the file scanned;
    // an index that points where the file cannot be read is an error, not zero coverage
    bool scan = true;
    if (!region.empty()) {
        const int sk = rd.seek_contig_ex(tids[0], &err);
        if (sk == -2) { fprintf(stderr, "samtools (goleft_amd shim): %s\n", err.c_str()); gd_destroy(ctx); return 1; }
        if (sk == 0) scan = false;
    }
    gdh::RecordBlock blk;
    while (scan) {
        const int rc = rd.next_block(blk, 1u << 20, &err);
        if (rc < 0) { fprintf(stderr, "samtools (goleft_amd shim): %s\n", err.c_str()); gd_destroy(ctx); return 1; }
        if (rc == 0) break;
        if (!region.empty()) {
            if (blk.tid < tids[0]) continue;
            if (blk.tid > tids[0]) break;
        }
        size_t n = blk.pos.size();
        if (!region.empty()) {                              // sorted: nothing at or past the region's end is needed
            size_t k = 0;
            while (k < n && blk.pos[k] < e) ++k;
            n = k;
        }
        if (n && gd_push(ctx, blk.tid, blk.pos.data(), blk.flag.data(), blk.mapq.data(), blk.cigar_off.data(), blk.cigar.data(),
                         n, blk.cigar_off[n]) != GD_OK)
            return fail("gd_push");
        if (!region.empty() && n < blk.pos.size()) break;
    }
    if (gd_compute(ctx) != GD_OK) return fail("gd_compute");
    std::vector<int32_t> d;
    static char obuf[1 << 20];
    setvbuf(stdout, obuf, _IOFBF, sizeof obuf);
    for (int32_t tid : tids) {
        const int64_t a = region.empty() ? 0 : s, b = region.empty() ? ctgs[(size_t)tid].length : e;
        if (b <= a) continue;
        d.resize((size_t)(b - a));
        if (gd_perbase(ctx, tid, a, b, d.data()) != GD_OK) return fail("gd_perbase");
        const char* name = ctgs[(size_t)tid].name.c_str();
        for (int64_t p = a; p < b; ++p)
            if (all || d[(size_t)(p - a)] != 0) printf("%s\t%lld\t%d\n", name, (long long)(p + 1), d[(size_t)(p - a)]);
    }
    fflush(stdout);
    gd_destroy(ctx);
    return 0;
}
