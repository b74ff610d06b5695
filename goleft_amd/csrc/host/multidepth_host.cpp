// multidepth_host.cpp -- host twin of `multidepth` (/root/reference/multidepth/multidepth.go),
// the multi-BAM sibling of `goleft depth`.
//
//   goleft-depth multidepth -c CHROM [-Q 10] [--mincov 7] [--maxcov 1000] [-k 10] [-m 15]
//                           [-w 10000000] [-p P] [--minsamples 0.5] a.bam b.bam ...
//
// The reference spawns, per 5 Mb chunk, `samtools depth -q 0 -Q Q -d MaxCov -r chrom:start
// bams...` (:203-207), parses one text line per covered position (:148-161) and cuts blocks
// out of the stream with a small state machine (:217-258).  Here every BAM is decoded once
// into one contig of a single engine (S samples = S contigs), gd_compute produces the S
// per-base vectors, gd_md_flags reduces them to two bitmaps (printed positions, sufficient
// positions), the state machine below walks the bitmaps -- the same decisions in the same
// order, with word-level shortcuts -- and gd_md_sums returns the running sums the reference
// would hold for every block, so the "%.2f" columns are formatted from identical doubles.
// Chunks are processed in genome order (what `-p 1` prints; with more workers the reference
// prints the chunks' blocks in completion order).
#include <cerrno>
#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../../include/goleft_depth.h"
#include "../../../include/goleft_depth_host.h"
#include "bam_reader.hpp"
#include "gpu_ingest.hpp"

namespace {

// ---- multidepth.go:18-30 dargs, defaults :52 ---------------------------------
struct MArgs {
    int q = 10;                 // -Q / --q
    std::string chrom;          // -c / --chrom (required)
    int min_cov = 7;            // --mincov
    int max_cov = 1000;         // --maxcov (samtools -d: ignored by samtools >= 1.13, the contract here)
    int max_skip = 10;          // -k / --maxskip
    int min_size = 15;          // -m / --minsize
    int window = 10000000;      // -w / --window
    int processes = 0;          // -p / --processes (BGZF inflate threads here)
    double min_samples = 0.5;   // --minsamples
    std::vector<std::string> bams;
};

void usage(FILE* f)
{
    fputs("usage: multidepth [--q Q] --chrom CHROM [--mincov MINCOV] [--maxcov MAXCOV] [--maxskip MAXSKIP] "
          "[--minsize MINSIZE] [--window WINDOW] [--processes PROCESSES] [--minsamples MINSAMPLES] BAMS [BAMS ...]\n", f);
}

int parse_args(int argc, const char* const* argv, MArgs* a)
{
    struct Opt { const char* lng; const char* sht; int kind; };   // 1 int, 2 string, 3 float
    static const Opt opts[] = {
        {"--q", "-Q", 1}, {"--chrom", "-c", 2}, {"--mincov", nullptr, 1}, {"--maxcov", nullptr, 1},
        {"--maxskip", "-k", 1}, {"--minsize", "-m", 1}, {"--window", "-w", 1}, {"--processes", "-p", 1},
        {"--minsamples", nullptr, 3}};
    for (int i = 1; i < argc; ++i) {
        std::string arg = argv[i];
        if (arg == "--help" || arg == "-h") { usage(stdout); return 1; }
        if (arg.size() < 2 || arg[0] != '-') { a->bams.push_back(arg); continue; }
        std::string val;
        bool has_val = false;
        const size_t eq = arg.find('=');
        if (eq != std::string::npos) { val = arg.substr(eq + 1); arg = arg.substr(0, eq); has_val = true; }
        const Opt* o = nullptr;
        for (const Opt& c : opts)
            if (arg == c.lng || (c.sht && arg == c.sht)) o = &c;
        if (!o) { fprintf(stderr, "error: unknown argument %s\n", arg.c_str()); usage(stderr); return -1; }
        if (!has_val) {
            if (i + 1 >= argc) { fprintf(stderr, "error: missing value for %s\n", arg.c_str()); usage(stderr); return -1; }
            val = argv[++i];
        }
        char* end = nullptr;
        errno = 0;
        long iv = 0;
        double dv = 0;
        if (o->kind == 1) { iv = strtol(val.c_str(), &end, 10); }
        if (o->kind == 3) { dv = strtod(val.c_str(), &end); }
        if (o->kind != 2 && (errno || end == val.c_str() || *end)) {
            fprintf(stderr, "error: error processing %s: invalid value %s\n", arg.c_str(), val.c_str());
            usage(stderr);
            return -1;
        }
        const std::string n = o->lng;
        if (n == "--q") a->q = (int)iv;
        else if (n == "--chrom") a->chrom = val;
        else if (n == "--mincov") a->min_cov = (int)iv;
        else if (n == "--maxcov") a->max_cov = (int)iv;
        else if (n == "--maxskip") a->max_skip = (int)iv;
        else if (n == "--minsize") a->min_size = (int)iv;
        else if (n == "--window") a->window = (int)iv;
        else if (n == "--processes") a->processes = (int)iv;
        else if (n == "--minsamples") a->min_samples = dv;
    }
    if (a->chrom.empty()) { fprintf(stderr, "error: --chrom is required\n"); usage(stderr); return -1; }
    if (a->bams.empty()) { fprintf(stderr, "error: bams is required\n"); usage(stderr); return -1; }   // :53-55
    return 0;
}

// indexcov.GetShortName(b, false) (indexcov/indexcov.go:213-246): the single @RG SM value,
// else derived from the file name.  More than one distinct SM is an error.
bool short_name(const std::string& path, const std::string& header, std::string* out)
{
    std::vector<std::string> sms;
    size_t p = 0;
    while (p < header.size()) {
        size_t e = header.find('\n', p);
        if (e == std::string::npos) e = header.size();
        if (header.compare(p, 3, "@RG") == 0) {
            std::string sm;                                   // a read group without SM counts as ""
            size_t t = header.find('\t', p);
            while (t != std::string::npos && t < e) {
                size_t n = header.find('\t', t + 1);
                if (n == std::string::npos || n > e) n = e;
                if (header.compare(t + 1, 3, "SM:") == 0) sm = header.substr(t + 4, n - t - 4);
                t = n < e ? n : std::string::npos;
            }
            bool dup = false;
            for (const auto& s : sms) dup = dup || s == sm;
            if (!dup) sms.push_back(sm);
        }
        p = e + 1;
    }
    if (sms.size() > 1) return false;
    if (sms.size() == 1) { *out = sms[0]; return true; }
    std::string v = path.substr(path.find_last_of('/') == std::string::npos ? 0 : path.find_last_of('/') + 1);
    std::vector<std::string> parts;
    size_t s = 0;
    for (;;) {
        const size_t d = v.find('.', s);
        parts.push_back(v.substr(s, d == std::string::npos ? std::string::npos : d - s));
        if (d == std::string::npos) break;
        s = d + 1;
    }
    if (parts.size() <= 2) { *out = parts[0]; return true; }
    out->clear();
    for (size_t i = 0; i + 1 < parts.size(); ++i) { if (i) out->push_back('-'); out->append(parts[i]); }
    return true;
}

struct Block { int64_t start, end; };        // 0-based start, 1-based end (:175-180)

#define MDCHK(call)                                                                     \
    do {                                                                                \
        int rc_ = (call);                                                               \
        if (rc_ != GD_OK) {                                                             \
            fprintf(stderr, "multidepth: %s failed: %s (%s)\n", #call, gd_strerror(rc_), \
                    ctx ? gd_last_error(ctx) : "");                                     \
            if (ctx) gd_destroy(ctx);                                                   \
            return 2;                                                                   \
        }                                                                               \
    } while (0)

int run(const MArgs& a, FILE* out)
{
    const int S = (int)a.bams.size();
    gd_ctx* ctx = nullptr;
    std::string err;
    // header names (:59-73) and the chromosome length from the first BAM (:32-49)
    std::vector<std::string> names((size_t)S);
    std::vector<int32_t> chrom_tid((size_t)S, -1);
    int64_t L = -1;
    for (int s = 0; s < S; ++s) {
        gdh::BamReader br;
        if (!br.open(a.bams[(size_t)s], 1, &err)) {
            fprintf(stderr, "multidepth: %s: %s\n", a.bams[(size_t)s].c_str(), err.c_str());
            return 2;                                       // the reference panics
        }
        if (!short_name(a.bams[(size_t)s], br.header_text(), &names[(size_t)s])) {
            fprintf(stderr, "multidepth: bam reagroup: more than one RG for %s\n", a.bams[(size_t)s].c_str());
            return 2;
        }
        for (size_t t = 0; t < br.contigs().size(); ++t)
            if (br.contigs()[t].name == a.chrom) {
                chrom_tid[(size_t)s] = (int32_t)t;
                if (s == 0) L = br.contigs()[t].length;
            }
        if (s == 0 && L < 0) {
            fprintf(stderr, "multidepth: chromosome %s not found in %s\n", a.chrom.c_str(), a.bams[0].c_str());
            return 2;
        }
    }
    fputs("#chrom\tstart\tend", out);
    for (const auto& n : names) fprintf(out, "\t%s", n.c_str());
    fputc('\n', out);
    if (L <= 0) return 0;

    int device = 0;
    if (const char* e = getenv("GOLEFT_DEVICE")) device = atoi(e);
    {
        const int rc = gd_create(device, &ctx);
        if (rc != GD_OK) {
            fprintf(stderr, "multidepth: no usable MI355X device (%s); this build has no CPU path\n", gd_strerror(rc));
            return 2;
        }
    }
    gd_params P;
    gd_default_params(&P);
    P.min_mapq = a.q;                                       // samtools depth -Q (:205); -q 0 = no base-quality test
    P.window_size = 1000;                                   // windows are not used by multidepth
    MDCHK(gd_set_params(ctx, &P));
    std::vector<int64_t> lens((size_t)S, L);
    MDCHK(gd_set_contigs(ctx, S, lens.data()));

    // ---- every BAM's records of the chromosome -> contig s of the engine -----------
    const char* gd_env = getenv("GOLEFT_GPU_DECODE");
    const bool want_gpu = !(gd_env && gd_env[0] == '0');
    for (int s = 0; s < S; ++s) {
        if (chrom_tid[(size_t)s] < 0) continue;             // samtools prints 0 for a file without the contig
        const int32_t want = chrom_tid[(size_t)s];
        // with a .bai: the file's bytes of this reference are inflated and decoded on the device
        std::vector<std::vector<uint64_t>> lin;
        if (want_gpu && gdh::BamReader::linear_index(a.bams[(size_t)s], &lin, &err) && (size_t)want < lin.size()) {
            gdh::FileMap fm;
            if (fm.open(a.bams[(size_t)s])) {
                uint64_t n = 0;
                bool io_ok = true;
                MDCHK(gdh::ingest_reference_on_device(ctx, fm, lin, want, s, &n, &io_ok));
                if (io_ok) continue;
            }
        }
        gdh::BamReader br;
        if (!br.open(a.bams[(size_t)s], a.processes, &err)) { fprintf(stderr, "multidepth: %s\n", err.c_str()); gd_destroy(ctx); return 2; }
        br.seek_contig(want, &err);                         // .bai shortcut when there is one
        gdh::RecordBlock blk;
        for (;;) {
            const int rc = br.next_block(blk, 1u << 21, &err);
            if (rc < 0) { fprintf(stderr, "multidepth: %s\n", err.c_str()); gd_destroy(ctx); return 2; }
            if (rc == 0 || blk.tid > want) break;           // coordinate sorted: done
            if (blk.tid != want) continue;
            gd_batch b;
            MDCHK(gd_acquire(ctx, blk.size(), blk.cigar.size(), &b));
            memcpy(b.pos, blk.pos.data(), blk.size() * sizeof(int32_t));
            memcpy(b.flag, blk.flag.data(), blk.size() * sizeof(uint16_t));
            memcpy(b.mapq, blk.mapq.data(), blk.size() * sizeof(uint8_t));
            memcpy(b.cigar_off, blk.cigar_off.data(), (blk.size() + 1) * sizeof(uint32_t));
            if (!blk.cigar.empty()) memcpy(b.cigar, blk.cigar.data(), blk.cigar.size() * sizeof(uint32_t));
            MDCHK(gd_commit(ctx, &b, s, blk.size(), blk.cigar.size()));
        }
    }
    // ---- bitmaps, blocks, means ---------------------------------------------------
    // The records of all S samples stay in HBM (15 bytes per read); their per-base vectors (4 bytes per
    // position and sample) exist for ONE GROUP of samples at a time: gd_select_contigs + gd_compute per
    // group, twice -- once to accumulate the per-position counts behind the two bitmaps, once (the blocks
    // known) for the block means.  The reference bounds memory with 5 Mb position chunks (:114,126); a
    // group of samples over the whole chromosome does the same job and keeps every launch large.
    int group = 16;                                                                      // 16 x 4 x L bytes resident
    if (const char* e = getenv("GOLEFT_MD_GROUP")) group = std::max(1, atoi(e));
    const int need = (int)(0.5 + a.min_samples * (double)S);                              // :66
    MDCHK(gd_md_begin(ctx, L));
    std::vector<int32_t> tids;
    for (int g0 = 0; g0 < S; g0 += group) {
        tids.clear();
        for (int s = g0; s < std::min(S, g0 + group); ++s) tids.push_back(s);
        MDCHK(gd_select_contigs(ctx, (int)tids.size(), tids.data()));
        MDCHK(gd_compute(ctx));
        MDCHK(gd_md_accumulate(ctx, (int)tids.size(), tids.data(), a.min_cov));
    }
    MDCHK(gd_md_finish(ctx, need, nullptr, nullptr, 0));
    int64_t chunk = 5000000;                                                             // :119
    if (S > 50) chunk /= 5;                                                              // :62-64
    // aggregate + splitBlocks of every chunk (:188-268, genRegions :130-141), on the device
    size_t nb = 0;
    std::vector<int64_t> bs, be;
    {
        int rc = gd_md_blocks(ctx, chunk, a.max_skip, a.min_size, a.window, nullptr, nullptr, 0, &nb);
        if (rc != GD_OK && rc != GD_E_CAPACITY) MDCHK(rc);
        bs.resize(nb); be.resize(nb);
        if (nb) MDCHK(gd_md_blocks(ctx, chunk, a.max_skip, a.min_size, a.window, bs.data(), be.data(), nb, &nb));
    }
    std::vector<Block> blocks(nb);
    for (size_t k = 0; k < nb; ++k) blocks[k] = Block{bs[k], be[k]};
    std::vector<double> sums(blocks.size() * (size_t)S);
    std::vector<double> gsum;
    for (int g0 = 0; g0 < S && nb; g0 += group) {
        tids.clear();
        for (int s = g0; s < std::min(S, g0 + group); ++s) tids.push_back(s);
        if (S > group) {                                                                 // one group: its vectors are still there
            MDCHK(gd_select_contigs(ctx, (int)tids.size(), tids.data()));
            MDCHK(gd_compute(ctx));
        }
        gsum.assign(nb * tids.size(), 0.0);
        MDCHK(gd_md_sums_group(ctx, (int)tids.size(), tids.data(), nb, bs.data(), be.data(), gsum.data()));
        for (size_t k = 0; k < nb; ++k)
            for (size_t t = 0; t < tids.size(); ++t) sums[k * (size_t)S + (size_t)tids[t]] = gsum[k * tids.size() + t];
    }
    if (!gdh_get_fast_exit()) gd_destroy(ctx);             // (goleft-depth exits right after: main.cpp)
    ctx = nullptr;
    for (size_t k = 0; k < blocks.size(); ++k) {
        fprintf(out, "%s\t%" PRId64 "\t%" PRId64, a.chrom.c_str(), blocks[k].start, blocks[k].end);   // :182-184
        const double l = (double)(blocks[k].end - blocks[k].start);                      // last.pos0 - first.pos0 + 1
        for (int s = 0; s < S; ++s) fprintf(out, "\t%.2f", sums[k * (size_t)S + (size_t)s] / l * 1000);   // :279-281
        fputc('\n', out);
    }
    return ferror(out) ? 1 : 0;
}

}  // namespace

extern "C" {

int gdh_multidepth_run(int argc, const char* const* argv, const char* out_path)
{
    MArgs a;
    const int rc = parse_args(argc, argv, &a);
    if (rc > 0) return 0;
    if (rc < 0) return 255;           // go-arg MustParse -> os.Exit(-1)
    FILE* out = stdout;
    if (out_path) {
        out = fopen(out_path, "w");
        if (!out) { fprintf(stderr, "multidepth: cannot create %s\n", out_path); return 1; }
    }
    int r = run(a, out);
    if (out_path) { if (fclose(out) != 0 && r == 0) r = 1; }
    else fflush(stdout);
    return r;
}

int gdh_multidepth_main(int argc, const char* const* argv) { return gdh_multidepth_run(argc, argv, nullptr); }

// The block state machine alone, over caller-provided bitmaps: they are uploaded (gd_md_load_flags) and the
// device finds the blocks (gd_md_blocks) -- there is no host state machine any more.  Returns the number of
// blocks (fills up to cap {start, end} pairs), -1 on bad arguments, -2 without a usable device.
int64_t gdh_multidepth_blocks(const uint32_t* any_bits, const uint32_t* suf_bits, int64_t len, int64_t chunk,
                              int32_t max_skip, int32_t min_size, int32_t window, int64_t* starts, int64_t* ends,
                              int64_t cap)
{
    if (len < 0 || chunk < 1 || max_skip < 0 || (len && (!any_bits || !suf_bits))) return -1;
    int device = 0;
    if (const char* e = getenv("GOLEFT_DEVICE")) device = atoi(e);
    gd_ctx* ctx = nullptr;
    if (gd_create(device, &ctx) != GD_OK) return -2;
    size_t nb = 0;
    int rc = gd_md_load_flags(ctx, any_bits, suf_bits, len);
    std::vector<int64_t> bs, be;
    if (rc == GD_OK) {
        rc = gd_md_blocks(ctx, chunk, max_skip, min_size, window, nullptr, nullptr, 0, &nb);
        if (rc == GD_E_CAPACITY) rc = GD_OK;
        if (rc == GD_OK && nb) {
            bs.resize(nb); be.resize(nb);
            rc = gd_md_blocks(ctx, chunk, max_skip, min_size, window, bs.data(), be.data(), nb, &nb);
        }
    }
    if (rc != GD_OK) fprintf(stderr, "multidepth: %s (%s)\n", gd_strerror(rc), gd_last_error(ctx));
    gd_destroy(ctx);
    if (rc != GD_OK) return -2;
    for (size_t k = 0; k < nb && (int64_t)k < cap; ++k) {
        if (starts) starts[k] = bs[k];
        if (ends) ends[k] = be[k];
    }
    return (int64_t)nb;
}

}  // extern "C"
