// depth_host.cpp -- host side of `goleft depth` above the device C ABI.
//
// C++ twin of the reference's Go front end (/root/reference/depth/depth.go;
// the Go toolchain is absent from the build image).  It keeps the reference's
// flags (:27-41), defaults (:164-167), .fai tiling (:122-159), --bed regions
// (:103-120), output naming (:382-388) and the exact BED rows of the callback
// (:238-364, including quirks Q1/Q2 of SURVEY.md section 3.3), but where the
// reference spawns `samtools depth` per tile and parses its text, this host
// streams decoded BAM records into the HIP engine (goleft_depth.h) and formats
// rows from the integer results (window sums, coverage-class runs).
#include "../../../include/goleft_depth_host.h"

#include <algorithm>
#include <cerrno>
#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <chrono>
#include <functional>
#include <string>
#include <atomic>
#include <thread>
#include <vector>

#include "bam_reader.hpp"
#include "fasta_stats.hpp"
#include "gpu_ingest.hpp"

namespace {

const char* const kClassName[4] = {"NO_COVERAGE", "LOW_COVERAGE", "CALLABLE", "EXCESSIVE_COVERAGE"};

// ---- depth/depth.go:27-41 dargs ------------------------------------------
struct DArgs {
    int window_size = 250;      // --windowsize / -w
    int max_mean_depth = 0;     // --maxmeandepth / -m
    bool ordered = false;       // --ordered / -o   (output is always ordered here)
    int q = 1;                  // --q / -Q
    std::string chrom;          // --chrom / -c
    int min_cov = 4;            // --mincov
    bool stats = false;         // --stats / -s
    std::string reference;      // --reference / -r
    int processes = 0;          // --processes / -p  (BGZF inflate threads here)
    std::string bed;            // --bed / -b
    std::string prefix;         // --prefix (required)
    std::string bam;            // positional (required)
};

void usage(FILE* f)
{
    fputs("usage: goleft depth [--windowsize WINDOWSIZE] [--maxmeandepth MAXMEANDEPTH] [--ordered] "
          "[--q Q] [--chrom CHROM] [--mincov MINCOV] [--stats] [--reference REFERENCE] "
          "[--processes PROCESSES] [--bed BED] --prefix PREFIX BAM\n", f);
}

bool parse_int(const char* s, int* out)
{
    char* end = nullptr;
    errno = 0;
    long v = strtol(s, &end, 10);
    if (errno || end == s || *end || v < INT32_MIN || v > INT32_MAX) return false;
    *out = (int)v;
    return true;
}

// returns 0 ok, 1 help shown, -1 error (message printed)
int parse_args(int argc, const char* const* argv, DArgs* a)
{
    struct Opt { const char* lng; const char* sht; int kind; };  // kind 0 flag, 1 int, 2 string
    static const Opt opts[] = {
        {"--windowsize", "-w", 1}, {"--maxmeandepth", "-m", 1}, {"--ordered", "-o", 0},
        {"--q", "-Q", 1}, {"--chrom", "-c", 2}, {"--mincov", nullptr, 1}, {"--stats", "-s", 0},
        {"--reference", "-r", 2}, {"--processes", "-p", 1}, {"--bed", "-b", 2}, {"--prefix", nullptr, 2}};
    std::vector<std::string> positional;
    for (int i = 1; i < argc; ++i) {
        std::string arg = argv[i];
        if (arg == "--help" || arg == "-h") { usage(stdout); return 1; }
        if (arg.size() < 2 || arg[0] != '-') { positional.push_back(arg); continue; }
        std::string val;
        bool has_val = false;
        size_t eq = arg.find('=');
        if (eq != std::string::npos) { val = arg.substr(eq + 1); arg = arg.substr(0, eq); has_val = true; }
        const Opt* o = nullptr;
        for (const Opt& c : opts)
            if (arg == c.lng || (c.sht && arg == c.sht)) o = &c;
        if (!o) { fprintf(stderr, "error: unknown argument %s\n", arg.c_str()); usage(stderr); return -1; }
        if (o->kind != 0 && !has_val) {
            if (i + 1 >= argc) { fprintf(stderr, "error: missing value for %s\n", arg.c_str()); usage(stderr); return -1; }
            val = argv[++i];
        }
        int iv = 0;
        if (o->kind == 1 && !parse_int(val.c_str(), &iv)) {
            fprintf(stderr, "error: error processing %s: invalid integer %s\n", arg.c_str(), val.c_str());
            usage(stderr);
            return -1;
        }
        const std::string n = o->lng;
        if (n == "--windowsize") a->window_size = iv;
        else if (n == "--maxmeandepth") a->max_mean_depth = iv;
        else if (n == "--ordered") a->ordered = !has_val || val == "true";
        else if (n == "--q") a->q = iv;
        else if (n == "--chrom") a->chrom = val;
        else if (n == "--mincov") a->min_cov = iv;
        else if (n == "--stats") a->stats = !has_val || val == "true";
        else if (n == "--reference") a->reference = val;
        else if (n == "--processes") a->processes = iv;
        else if (n == "--bed") a->bed = val;
        else if (n == "--prefix") a->prefix = val;
    }
    if (positional.size() > 1) { fprintf(stderr, "error: too many positional arguments at '%s'\n", positional[1].c_str()); usage(stderr); return -1; }
    if (positional.size() == 1) a->bam = positional[0];
    if (a->prefix.empty()) { fprintf(stderr, "error: --prefix is required\n"); usage(stderr); return -1; }
    if (a->bam.empty()) { fprintf(stderr, "error: bam is required\n"); usage(stderr); return -1; }
    if (a->window_size < 1) { fprintf(stderr, "error: --windowsize must be >= 1\n"); return -1; }
    return 0;
}

// ---- depth/depth.go:73-94 -------------------------------------------------
// regexp `(.+?)[:\t](\d+)([\-\t])(\d+).*?`, leftmost match with a lazy chrom:
// the first separator (':' or TAB, not at column 0) that is followed by
// digits, '-' or TAB, and at least one more digit.
bool chrom_start_end(const char* line, size_t len, std::string* chrom, int64_t* start, int64_t* end)
{
    auto digits = [&](size_t from) {
        size_t j = from;
        while (j < len && line[j] >= '0' && line[j] <= '9') ++j;
        return j;
    };
    for (size_t sep = 1; sep < len; ++sep) {
        const char c = line[sep];
        if (c == '\n') break;                    // '.' does not cross a newline
        if (c != ':' && c != '\t') continue;
        const size_t d1 = digits(sep + 1);
        if (d1 == sep + 1 || d1 >= len) continue;
        const char mid = line[d1];
        if (mid != '-' && mid != '\t') continue;
        const size_t d2 = digits(d1 + 1);
        if (d2 == d1 + 1) continue;
        chrom->assign(line, sep);
        int64_t s = strtoll(std::string(line + sep + 1, d1 - sep - 1).c_str(), nullptr, 10);
        const int64_t e = strtoll(std::string(line + d1 + 1, d2 - d1 - 1).c_str(), nullptr, 10);
        if (mid == '-') --s;                     // chr:s-e is 1-based (:86-88)
        *start = s < 0 ? 0 : s;                  // :93
        *end = e;
        return true;
    }
    return false;
}

int64_t step_for(int window_size)
{
    // depth/depth.go:48,:132
    int64_t n = 10000000 / window_size;
    if (n < 1) n = 1;
    return n * window_size;
}

// ---- rows of one region (depth/depth.go:238-364 restated over integer results)
struct RowWriter {
    std::string hd, ca;     // buffered rows
};

void fmt_depth_row(std::string* out, const char* chrom, int64_t s, int64_t e, int64_t sum,
                   const std::string& stats)
{
    char buf[96];
    const int64_t l = e - s;
    const double mean = (sum == 0 || l == 0) ? 0.0 : (double)sum / (double)l;   // :181-189
    int n = snprintf(buf, sizeof buf, "\t%" PRId64 "\t%" PRId64 "\t%.4g", s, e, mean);
    out->append(chrom);
    out->append(buf, (size_t)n);
    out->append(stats);
    out->push_back('\n');
}

void fmt_callable_row(std::string* out, const char* chrom, int64_t s, int64_t e, int cls)
{
    char buf[64];
    int n = snprintf(buf, sizeof buf, "\t%" PRId64 "\t%" PRId64 "\t", s, e);
    out->append(chrom);
    out->append(buf, (size_t)n);
    out->append(kClassName[cls & 3]);
    out->push_back('\n');
}

// --stats columns of the depth.bed rows of ONE format_region call (depth/depth.go:191-200).
// The windows a region emits are data dependent (quirk Q2), so format_region runs twice:
// collecting == true records [s, e) of every row in emission order; the device counts the
// bases of all of them in one gd_seq_stats call; the second run pops the formatted columns.
struct StatsPlan {
    bool collecting = true;
    std::vector<int64_t> s, e;
    std::vector<std::string> cols;
    size_t next = 0;
    std::string take(int64_t ws, int64_t we)
    {
        if (collecting) { s.push_back(ws); e.push_back(we); return std::string(); }
        return next < cols.size() ? cols[next++] : std::string();
    }
};

// sums[k] belongs to window first_win + k.  sp may be null (no --stats).
void format_region(RowWriter* w, const char* chrom, int64_t rs, int64_t re, int W,
                   const int64_t* sums, size_t n_sums, const gd_run* runs, size_t n_runs,
                   StatsPlan* sp)
{
    if (re <= rs) return;
    const int64_t first_win = rs / W;
    auto sum_of = [&](int64_t iw) -> int64_t {
        const int64_t k = iw - first_win;
        return (k >= 0 && (size_t)k < n_sums) ? sums[k] : 0;
    };
    auto stats_of = [&](int64_t s, int64_t e) { return sp ? sp->take(s, e) : std::string(); };
    // callable.bed: the run-length encoding is exactly what :307-328 and :343-350 emit
    int64_t lastcov = -1;
    for (size_t i = 0; i < n_runs; ++i) {
        fmt_callable_row(&w->ca, chrom, runs[i].start, runs[i].end, runs[i].cls);
        if (runs[i].cls != GD_NO_COVERAGE) lastcov = (int64_t)runs[i].end - 1;
    }
    // depth.bed
    int64_t pos = 0;
    if (lastcov >= 0) {
        const int64_t this_window = lastcov / W;
        for (int64_t iw = first_win; iw < this_window; ++iw) {            // :296-303
            const int64_t s = std::max(rs, iw * W), e = std::min(re, (iw + 1) * W);
            fmt_depth_row(&w->hd, chrom, s, e, sum_of(iw), stats_of(s, e));
        }
        const int64_t s = std::max(this_window * (int64_t)W, rs);         // :330-333 (quirk Q2)
        const int64_t e = std::min(re, s + W);
        fmt_depth_row(&w->hd, chrom, s, e, sum_of(this_window), stats_of(s, e));
        pos = e;                                                          // :338
    }
    if (lastcov + 1 < re) {                                               // :343
        for (int64_t ds = std::max(rs, pos) / W * W; ds < re && pos < re; ds += W) {   // :351
            const int64_t de = std::min(re, ds + W), s = std::max(ds, rs);
            fmt_depth_row(&w->hd, chrom, s, de, 0, stats_of(s, de));
        }
    }
}

bool flush_rows(RowWriter* w, FILE* fhd, FILE* fca)
{
    bool ok = true;
    if (!w->hd.empty()) ok = fwrite(w->hd.data(), 1, w->hd.size(), fhd) == w->hd.size() && ok;
    if (!w->ca.empty()) ok = fwrite(w->ca.data(), 1, w->ca.size(), fca) == w->ca.size() && ok;
    w->hd.clear();
    w->ca.clear();
    return ok;
}

struct Region {
    std::string chrom;
    int64_t start, end;
    int tid;
};

struct FaiEntry {
    std::string name;
    int64_t length;
};

bool read_fai(const std::string& path, std::vector<FaiEntry>* out)
{
    FILE* f = fopen(path.c_str(), "r");
    if (!f) return false;
    char* line = nullptr;
    size_t cap = 0;
    ssize_t n;
    while ((n = getline(&line, &cap, f)) > 0) {
        if (line[n - 1] != '\n') break;          // quirk Q6 (depth/depth.go:137-140): an unterminated last .fai line is dropped
        std::string s(line, (size_t)n);
        const size_t t1 = s.find('\t');
        if (t1 == std::string::npos) continue;
        FaiEntry e;
        e.name = s.substr(0, t1);
        e.length = strtoll(s.c_str() + t1 + 1, nullptr, 10);
        out->push_back(e);
    }
    free(line);
    fclose(f);
    return true;
}

// One engine context per device ("shard"): the reference parallelises over tiles inside one process
// (`-p`, depth/depth.go:392-394) and merges in :394-421; here the unit is a contig, assigned to a device.
struct Shard {
    gd_ctx* ctx = nullptr;
    int device = 0;
    std::vector<int32_t> wanted;        // the references this shard computes (ascending)
    uint64_t n_gpu_records = 0;
    int rc = GD_OK;                     // result of the shard's worker thread
    bool io_ok = true;
    std::string what;                   // the failing call
};

// The HIP runtime and an engine context take 0.2 - 0.3 s to come up: they are created on threads of their own the
// moment the device list is known, while the main thread reads the BAM header, the .fai and the .bai.
struct EarlyContexts {
    std::vector<int> devices;
    std::vector<gd_ctx*> ctx;
    std::vector<int> rc;
    std::vector<std::thread> th;
    void start(const std::vector<int>& devs)
    {
        devices = devs;
        ctx.assign(devs.size(), nullptr);
        rc.assign(devs.size(), GD_OK);
        for (size_t k = 0; k < devs.size(); ++k)
            th.emplace_back([this, k]() { rc[k] = gd_create(devices[k], &ctx[k]); });
    }
    void join() { for (auto& t : th) if (t.joinable()) t.join(); }
    // the context of devices[k]: the caller owns it from now on
    int take(size_t k, gd_ctx** out) { join(); *out = ctx[k]; ctx[k] = nullptr; return rc[k]; }
    ~EarlyContexts()
    {
        join();
        if (gdh_get_fast_exit()) return;
        for (gd_ctx* c : ctx) if (c) gd_destroy(c);
    }
};

struct Shards {
    std::vector<Shard> v;
    ~Shards()
    {
        if (gdh_get_fast_exit()) return;                     // the process is about to exit (main.cpp)
        for (Shard& s : v) if (s.ctx) gd_destroy(s.ctx);
    }
};

// Longest-processing-time-first assignment of contigs (by length) to n shards; deterministic.
std::vector<std::vector<int32_t>> lpt_assign(const std::vector<int32_t>& tids, const std::vector<int64_t>& lens, size_t n)
{
    std::vector<int32_t> order(tids);
    std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return lens[(size_t)a] > lens[(size_t)b]; });
    std::vector<std::vector<int32_t>> out(n);
    std::vector<int64_t> load(n, 0);
    for (int32_t t : order) {
        size_t k = 0;
        for (size_t i = 1; i < n; ++i) if (load[i] < load[k]) k = i;
        out[k].push_back(t);
        load[k] += lens[(size_t)t];
    }
    for (auto& o : out) std::sort(o.begin(), o.end());
    return out;
}

// GOLEFT_DEVICES: "all", or a comma separated list of HIP device ids, one engine context each
// (an id may repeat: several contexts on one device -- "virtual shards", how the multi-device path is
// tested on a one-GPU box).  Unset: GOLEFT_DEVICE (default 0), one context.
bool parse_devices(std::vector<int>* out, std::string* err)
{
    out->clear();
    const char* e = getenv("GOLEFT_DEVICES");
    if (!e || !*e) {
        int d = 0;
        if (const char* one = getenv("GOLEFT_DEVICE")) d = atoi(one);
        out->push_back(d);
        return true;
    }
    if (strcmp(e, "all") == 0) {
        int n = 0;
        if (gd_device_count(&n) != GD_OK || n < 1) { *err = "GOLEFT_DEVICES=all: no device"; return false; }
        for (int i = 0; i < n; ++i) out->push_back(i);
        return true;
    }
    const char* p = e;
    while (*p) {
        char* end = nullptr;
        const long v = strtol(p, &end, 10);
        if (end == p || v < 0 || v > 4095) { *err = std::string("GOLEFT_DEVICES: cannot parse '") + e + "'"; return false; }
        out->push_back((int)v);
        p = end;
        if (*p == ',') ++p;
        else if (*p) { *err = std::string("GOLEFT_DEVICES: cannot parse '") + e + "'"; return false; }
    }
    if (out->empty() || out->size() > 64) { *err = "GOLEFT_DEVICES: between 1 and 64 contexts"; return false; }
    return true;
}

#define GDCHK_ON(cx_, call)                                                            \
    do {                                                                               \
        int rc_ = (call);                                                              \
        if (rc_ != GD_OK) {                                                            \
            fprintf(stderr, "goleft depth: %s failed: %s (%s)\n", #call, gd_strerror(rc_), \
                    (cx_) ? gd_last_error(cx_) : "");                                  \
            return 1;                                                                  \
        }                                                                              \
    } while (0)

// measurement switches of the CLI (README): an integer from the environment, or the default
int env_int(const char* name, int dflt)
{
    const char* e = getenv(name);
    return e && *e ? atoi(e) : dflt;
}

// The regions of a run, in output order: the rows of --bed (depth/depth.go:103-120; quirk Q6: an unterminated last line
// is dropped as ReadBytes + io.EOF drop it) or the W-aligned 10 Mb tiles of every .fai contig (:122-159; --chrom keeps
// one).  tid = the BAM reference of the name, -1 when the BAM has none.  0, or the process's exit code.
static int collect_regions(const DArgs& args, const std::map<std::string, int>& tid_of, int64_t step, std::vector<Region>* out)
{
    std::vector<Region>& regions = *out;
    if (!args.bed.empty()) {
        FILE* f = fopen(args.bed.c_str(), "r");
        if (!f) { fprintf(stderr, "goleft depth: open %s: %s\n", args.bed.c_str(), strerror(errno)); return 1; }
        char* line = nullptr;
        size_t cap = 0;
        ssize_t n;
        while ((n = getline(&line, &cap, f)) > 0) {
            if (line[n - 1] != '\n') break;      // quirk Q6: ReadBytes returns the unterminated last line WITH io.EOF
                                                 // and :108-110 breaks before using it -- the row is dropped
            Region r;
            if (!chrom_start_end(line, (size_t)n, &r.chrom, &r.start, &r.end)) {
                fprintf(stderr, "couldn't get region from line%s", line);      // :78 log.Fatal
                free(line);
                fclose(f);
                return 1;
            }
            auto it = tid_of.find(r.chrom);
            r.tid = it == tid_of.end() ? -1 : it->second;
            regions.push_back(r);
        }
        free(line);
        fclose(f);
    } else {
        std::vector<FaiEntry> fai;
        if (!read_fai(args.reference + ".fai", &fai)) {
            fprintf(stderr, "goleft depth: open %s.fai: %s\n", args.reference.c_str(), strerror(errno));
            return 1;                                                           // pcheck -> log.Fatal
        }
        for (const FaiEntry& e : fai) {
            if (!args.chrom.empty() && e.name != args.chrom) continue;          // :145
            auto it = tid_of.find(e.name);
            const int tid = it == tid_of.end() ? -1 : it->second;
            for (int64_t i = 0; i < e.length; i += step)                        // :150-154
                regions.push_back(Region{e.name, i, std::min(i + step, e.length), tid});
        }
    }
    return 0;
}

// fn on every shard that has contigs, each on its own OS thread (every gd_* entry point re-issues hipSetDevice); false
// (after a message) when one failed.
static bool run_on_every_shard(Shards& S, const char* what, const std::function<int(Shard&)>& fn)
{
    std::vector<std::thread> th;
    for (size_t k = 1; k < S.v.size(); ++k)
        th.emplace_back([&, k]() { S.v[k].rc = S.v[k].wanted.empty() ? GD_OK : fn(S.v[k]); });
    S.v[0].rc = S.v[0].wanted.empty() ? GD_OK : fn(S.v[0]);
    for (auto& t : th) t.join();
    for (Shard& sh : S.v)
        if (sh.rc != GD_OK) {
            fprintf(stderr, "goleft depth: %s failed on device %d: %s (%s)\n", what, sh.device, gd_strerror(sh.rc),
                    gd_last_error(sh.ctx));
            return false;
        }
    return true;
}

struct ReadTimes { std::chrono::steady_clock::time_point indexed, mapped, read; };

// ---- records into HBM (replaces the samtools children) ---------------------------
// With a .bai next to the BAM (goleft depth needs one anyway: `samtools depth -r`), the whole
// read happens on the device: the contig's byte range goes to gd_ingest_bgzf, which inflates
// the BGZF members and decodes the records there (GOLEFT_GPU_DECODE=0 keeps the host decoder).
// Every shard reads its own contigs' byte ranges from the shared mapping.
// -> 0, or the process's exit code; *n_gpu_records: what the device decoder delivered (0: the host decoder ran).
static int read_records(const DArgs& args, gdh::BamReader& bam, const std::vector<int32_t>& wanted, Shards& S,
                        const std::vector<int>& shard_of, uint64_t* n_gpu_records_out, ReadTimes* T)
{
    const auto& contigs = bam.contigs();
    std::string err;
    uint64_t n_gpu_records = 0;
    auto now = []() { return std::chrono::steady_clock::now(); };
    auto ctx_of = [&](int tid) -> gd_ctx* { return S.v[(size_t)shard_of[(size_t)tid]].ctx; };
    auto on_every_shard = [&](const char* what, const std::function<int(Shard&)>& fn) { return run_on_every_shard(S, what, fn); };
    auto& t_indexed = T->indexed; auto& t_mapped = T->mapped; auto& t_read = T->read;
    {
        std::vector<std::vector<uint64_t>> lin;
        std::vector<char> has_chunks;
        std::vector<uint64_t> chunk_end;
        const char* gd_env = getenv("GOLEFT_GPU_DECODE");
        bool gpu_decode = !(gd_env && gd_env[0] == '0') && gdh::BamReader::linear_index(args.bam, &lin, &err, &has_chunks, &chunk_end) &&
                          lin.size() == contigs.size();
        // a reference whose bins hold chunks but whose linear index is empty (a non-htslib indexer):
        // the anchors cannot be trusted to mean "no records" -> host decoder
        for (size_t r = 0; gpu_decode && r < lin.size(); ++r)
            if (lin[r].empty() && r < has_chunks.size() && has_chunks[r]) gpu_decode = false;
        t_indexed = now();
        if (gpu_decode) {
            gdh::FileMap fm;
            if (!fm.open(args.bam)) gpu_decode = false;
            fm.keep = gdh_get_fast_exit() != 0;
            t_mapped = now();
            if (gpu_decode) {
                if (!on_every_shard("the device BAM read", [&](Shard& sh) {
                        return gdh::ingest_references_on_device(sh.ctx, fm, lin, sh.wanted, sh.wanted, &sh.n_gpu_records, &sh.io_ok,
                                                                (uint64_t)env_int("GOLEFT_INGEST_GROUP_MB", 512) << 20, &chunk_end,
                                                                // a reference larger than this is read in parts cut at .bai anchors
                                                                // (0: never; _KB: tests cut small files)
                                                                getenv("GOLEFT_INGEST_PART_KB") ? (uint64_t)env_int("GOLEFT_INGEST_PART_KB", 0) << 10
                                                                                                : (uint64_t)env_int("GOLEFT_INGEST_PART_MB", 2048) << 20);
                    }))
                    return 1;
                for (Shard& sh : S.v) {
                    if (!sh.io_ok) gpu_decode = false;
                    n_gpu_records += sh.n_gpu_records;
                }
                if (!gpu_decode) n_gpu_records = 0;
                t_read = now();
            }
            if (!gpu_decode)
                for (Shard& sh : S.v) GDCHK_ON(sh.ctx, gd_reset(sh.ctx));   // fall back to the host decoder below
        }
        if (!gpu_decode) {
        if (wanted.size() == 1) bam.seek_contig(wanted[0], &err);   // .bai shortcut for --chrom
        std::vector<char> want(contigs.size(), 0);
        for (int32_t t : wanted) want[(size_t)t] = 1;
        const int32_t last_wanted = wanted.back();
        gdh::RecordBlock blk;
        for (;;) {
            const int rc = bam.next_block(blk, 1u << 21, &err);
            if (rc < 0) {
                fprintf(stderr, "goleft depth: %s\n", err.c_str());
                return 1;
            }
            if (rc == 0) break;
            if (blk.tid > last_wanted) break;                       // coordinate sorted: done
            if (blk.tid >= (int32_t)contigs.size() || !want[(size_t)blk.tid]) continue;
            gd_ctx* const ctx = ctx_of(blk.tid);
            gd_batch b;
            GDCHK_ON(ctx, gd_acquire(ctx, blk.size(), blk.cigar.size(), &b));
            memcpy(b.pos, blk.pos.data(), blk.size() * sizeof(int32_t));
            memcpy(b.flag, blk.flag.data(), blk.size() * sizeof(uint16_t));
            memcpy(b.mapq, blk.mapq.data(), blk.size() * sizeof(uint8_t));
            memcpy(b.cigar_off, blk.cigar_off.data(), (blk.size() + 1) * sizeof(uint32_t));
            if (!blk.cigar.empty()) memcpy(b.cigar, blk.cigar.data(), blk.cigar.size() * sizeof(uint32_t));
            GDCHK_ON(ctx, gd_commit(ctx, &b, blk.tid, blk.size(), blk.cigar.size()));
        }
        }
    }
    *n_gpu_records_out = n_gpu_records;
    return 0;
}

// One engine context per device (created since the start of run(), EarlyContexts), configured for this job: parameters,
// how the BAM's bytes reach the device, the contig table, the outputs the rows need, and the contigs LPT gave the shard.
// `S` is the Shards object of the caller (named S so that the body reads like the rest of run()).
static int configure_shards(const std::vector<int>& devices, EarlyContexts* early, const std::vector<std::vector<int32_t>>& assignment,
                            const std::vector<int64_t>& lens, const gd_params& P, bool need_perbase, Shards* shards,
                            std::vector<int>* shard_of)
{
    Shards& S = *shards;
    const size_t n_shards = assignment.size();
    S.v.resize(n_shards);
    for (size_t k = 0; k < n_shards; ++k) {
        Shard& sh = S.v[k];
        sh.device = devices[k];
        sh.wanted = assignment[k];
        for (int32_t t : sh.wanted) (*shard_of)[(size_t)t] = (int)k;
        const int rc = early->take(k, &sh.ctx);               // (created since the start of run(), on its own thread)
        if (rc != GD_OK) {
            fprintf(stderr, "goleft depth: no usable MI355X device %d (%s); this build has no CPU path\n", sh.device, gd_strerror(rc));
            return 1;
        }
        GDCHK_ON(sh.ctx, gd_set_params(sh.ctx, &P));
        GDCHK_ON(sh.ctx, gd_set_option(sh.ctx, GD_OPT_COPY_THREADS, env_int("GOLEFT_COPY_THREADS", 4)));   // staging copies of the device BAM read
        // How the BAM's bytes reach the device: a copy KERNEL on CUs of its own (every 8th), the inflate launches on the
        // others (CU-masked streams).  One copy engine moves 21-22 GB/s next to the inflate kernels (and its reads of host
        // memory slow the pread into the staging buffers down: 0.8-2.1 s per genome by box against 0.65 s); the kernel on 32
        // dedicated CUs moves 35 GB/s.  GOLEFT_INGEST_DMA=1 brings the copy engine back, GOLEFT_INGEST_CU_SPLIT=0 the
        // unmasked streams (profiles/r11d_, r11e_scope3_genome.json).
        GDCHK_ON(sh.ctx, gd_set_option(sh.ctx, GD_OPT_INGEST_DMA, env_int("GOLEFT_INGEST_DMA", 0)));
        if (!getenv("GOLEFT_INGEST_CU_SPLIT") && env_int("GOLEFT_INGEST_DMA", 0) == 0)
            GDCHK_ON(sh.ctx, gd_set_option(sh.ctx, GD_OPT_INGEST_CU_SPLIT, 8));
        if (getenv("GOLEFT_INGEST_PIECE_STREAMS")) GDCHK_ON(sh.ctx, gd_set_option(sh.ctx, GD_OPT_INGEST_PIECE_STREAMS, env_int("GOLEFT_INGEST_PIECE_STREAMS", 1)));
        if (getenv("GOLEFT_INGEST_CU_SPLIT")) GDCHK_ON(sh.ctx, gd_set_option(sh.ctx, GD_OPT_INGEST_CU_SPLIT, env_int("GOLEFT_INGEST_CU_SPLIT", 0)));
        if (getenv("GOLEFT_INGEST_BATCHES")) GDCHK_ON(sh.ctx, gd_set_option(sh.ctx, GD_OPT_INGEST_BATCHES, env_int("GOLEFT_INGEST_BATCHES", 8)));
        if (getenv("GOLEFT_INGEST_WALK_CUS")) GDCHK_ON(sh.ctx, gd_set_option(sh.ctx, GD_OPT_INGEST_WALK_CUS, env_int("GOLEFT_INGEST_WALK_CUS", 0)));
        if (getenv("GOLEFT_INGEST_HYBRID")) GDCHK_ON(sh.ctx, gd_set_option(sh.ctx, GD_OPT_INGEST_HYBRID, env_int("GOLEFT_INGEST_HYBRID", 0)));
        if (getenv("GOLEFT_INGEST_COPY_GRID")) GDCHK_ON(sh.ctx, gd_set_option(sh.ctx, GD_OPT_INGEST_COPY_GRID, env_int("GOLEFT_INGEST_COPY_GRID", 16)));   // workgroups of the copy kernel
        if (getenv("GOLEFT_INFLATE_KERNEL")) GDCHK_ON(sh.ctx, gd_set_option(sh.ctx, GD_OPT_INFLATE_KERNEL, env_int("GOLEFT_INFLATE_KERNEL", 0)));
        if (getenv("GOLEFT_INFLATE_LDS_PAD")) GDCHK_ON(sh.ctx, gd_set_option(sh.ctx, GD_OPT_INFLATE_LDS_PAD, env_int("GOLEFT_INFLATE_LDS_PAD", 0)));
        if (env_int("GOLEFT_TRUST_BGZF", 0)) GDCHK_ON(sh.ctx, gd_set_option(sh.ctx, GD_OPT_INGEST_CRC, 0));
        // the context's worker threads (they pread the file's pieces into the staging buffers): 16 unless the container grants
        // few CPUs -- under a 16-CPU quota sixteen of them and the member lister's threads together ran into the quota (the
        // kernel then stops EVERY thread until the period ends: 6 - 13 throttled periods per genome, the file's pieces read at
        // half the rate); ten of them still read faster than the link takes the bytes
        {
            const int cpus = gdh::usable_cpus();
            const int pt = env_int("GOLEFT_PUSH_THREADS", cpus <= 16 ? std::max(4, cpus * 5 / 8) : 0);
            if (pt) GDCHK_ON(sh.ctx, gd_set_option(sh.ctx, GD_OPT_PUSH_THREADS, pt));
        }
        GDCHK_ON(sh.ctx, gd_set_contigs(sh.ctx, (int)lens.size(), lens.data()));
        GDCHK_ON(sh.ctx, gd_set_option(sh.ctx, GD_OPT_BAM_REFS, (int64_t)lens.size()));   // engine contigs = the BAM's references
        if (!need_perbase) GDCHK_ON(sh.ctx, gd_set_outputs(sh.ctx, 0));   // windows + class runs are all the rows need
        if (!lens.empty() && !sh.wanted.empty())
            GDCHK_ON(sh.ctx, gd_select_contigs(sh.ctx, (int)sh.wanted.size(), sh.wanted.data()));
    }
    return 0;
}

// The common case -- a whole-genome run: every region a fused tile of a known contig, no --stats -- formats its rows on
// all cores (a `%.4g` per window: 0.15 us each, half a second for a genome on one core): the contigs' results come off the
// device one after another, the tiles are cut into slices, every slice is formatted into its own buffer, the buffers are
// written in input order (what --ordered gives: depth/depth.go:394-421).
static int emit_fused_genome(const std::vector<Region>& regions, const std::vector<int64_t>& lens,
                             const std::function<gd_ctx*(int)>& ctx_of, int W, FILE* fhd, FILE* fca, bool* io_ok)
{
    gd_ctx* ctx = nullptr;                          // the context GDCHK reports on
#define GDCHK(call) GDCHK_ON(ctx, call)
    struct Group { size_t r0, r1; std::vector<int64_t> sums; std::vector<gd_run> runs; };
    std::vector<Group> groups;
    for (size_t i = 0; i < regions.size();) {
        size_t j = i;
        while (j < regions.size() && regions[j].tid == regions[i].tid) ++j;
        groups.push_back(Group{i, j, {}, {}});
        i = j;
    }
    for (Group& g : groups) {
        const int tid = regions[g.r0].tid;
        ctx = ctx_of(tid);
        size_t n = 0;
        g.sums.resize((size_t)((lens[(size_t)tid] + W - 1) / W));
        GDCHK(gd_windows(ctx, tid, g.sums.data(), nullptr, g.sums.size(), &n));
        const int rc = gd_callable(ctx, tid, nullptr, 0, &n);
        if (rc != GD_OK && rc != GD_E_CAPACITY) GDCHK(rc);
        g.runs.resize(n);
        if (n) GDCHK(gd_callable(ctx, tid, g.runs.data(), g.runs.size(), &n));
    }
    struct Slice { const Group* g; size_t r0, r1; RowWriter w; };
    std::vector<Slice> slices;
    constexpr size_t kSlice = 8;                        // tiles (10 Mb each at the default window) per slice
    for (const Group& g : groups)
        for (size_t i = g.r0; i < g.r1; i += kSlice) slices.push_back(Slice{&g, i, std::min(i + kSlice, g.r1), {}});
    std::atomic<size_t> next{0};
    auto work = [&]() {
        for (size_t k; (k = next.fetch_add(1)) < slices.size();) {
            Slice& sl = slices[k];
            const std::vector<gd_run>& ru = sl.g->runs;
            // runs are split at multiples of step, so each belongs to exactly one tile
            size_t cur = (size_t)(std::lower_bound(ru.begin(), ru.end(), regions[sl.r0].start,
                                                   [](const gd_run& a, int64_t x) { return a.start < x; }) - ru.begin());
            for (size_t i = sl.r0; i < sl.r1; ++i) {
                const Region& r = regions[i];
                while (cur < ru.size() && ru[cur].start < r.start) ++cur;
                size_t e = cur;
                while (e < ru.size() && ru[e].start < r.end) ++e;
                const size_t w0 = (size_t)(r.start / W), w1 = (size_t)((r.end + W - 1) / W);
                format_region(&sl.w, r.chrom.c_str(), r.start, r.end, W, sl.g->sums.data() + w0, w1 - w0, ru.data() + cur,
                              e - cur, nullptr);
                cur = e;
            }
        }
    };
    const unsigned nt = (unsigned)std::min<size_t>(std::max(1u, std::min((unsigned)gdh::usable_cpus(), 32u)), slices.size());
    std::vector<std::thread> th;
    for (unsigned t = 1; t < nt; ++t) th.emplace_back(work);
    work();
    for (auto& t : th) t.join();
    for (Slice& sl : slices) *io_ok = flush_rows(&sl.w, fhd, fca) && *io_ok;
#undef GDCHK
    return 0;
}

// Rows in input order for everything emit_fused_genome does not cover: fused tiles one contig at a time (its window sums and
// class runs fetched once), --bed rows and tiles cut by a disagreeing .fai through the region reductions (collected and
// reduced in batches: one gd_regions call for up to kBatch of them instead of several launches, allocations and
// synchronisations per row), regions of references the BAM header does not know (all-zero rows and a non-zero exit code,
// depth/depth.go:395-399), and the --stats columns (the bases of the current contig live in HBM -- gd_seq_load --, per
// region one gd_seq_stats call counts GC / CpG / lower-case bases of every emitted window; depth/depth.go:244-252, :199).
// add / finish: 0, or 1 after the failed call has been reported.
class RegionRows {
public:
    bool io_ok = true;
    int exit_code = 0;

    RegionRows(const std::vector<int64_t>& lens, Shards& S, const std::vector<int>& shard_of, gdh::FastaStats* fa, int W, FILE* fhd,
               FILE* fca)
        : lens_(lens), S_(S), shard_of_(shard_of), fa_(fa), W_(W), fhd_(fhd), fca_(fca), stats_contract_(gdh_get_stats_contract()),
          seq_ctx_(S.v[0].ctx)                                 // --stats: the FASTA windows are counted on the first device
    {
    }

    int add(const Region& r, bool fused)
    {
        gd_ctx* ctx = seq_ctx_;                                // the context GDCHK reports on
#define GDCHK(call) GDCHK_ON(ctx, call)
        if (stats_rc_ != GD_OK) GDCHK(stats_rc_);
        if (r.tid < 0) {
            GDCHK(flush_batch());
            // samtools would fail on an unknown reference name: the callback then sees an
            // empty stream (all-zero rows) and the exit code becomes non-zero (:395-399)
            fprintf(stderr, "ERROR with command: region %s:%" PRId64 "-%" PRId64 " not in the BAM header\n",
                    r.chrom.c_str(), r.start + 1, r.end);
            exit_code = std::max(exit_code, 1);
            gd_run nr{(int32_t)r.start, (int32_t)r.end, GD_NO_COVERAGE};
            emit_region(r.chrom.c_str(), r.start, r.end, nullptr, 0, &nr, r.end > r.start ? 1 : 0);
        } else if (fused) {
            ctx = ctx_of(r.tid);
            GDCHK(flush_batch());
            if (cached_tid_ != r.tid) {
                size_t n = 0;
                csums_.resize((size_t)((lens_[(size_t)r.tid] + W_ - 1) / W_));
                GDCHK(gd_windows(ctx, r.tid, csums_.data(), nullptr, csums_.size(), &n));
                int rc = gd_callable(ctx, r.tid, nullptr, 0, &n);
                if (rc != GD_OK && rc != GD_E_CAPACITY) GDCHK(rc);
                cruns_.resize(n);
                if (n) GDCHK(gd_callable(ctx, r.tid, cruns_.data(), cruns_.size(), &n));
                cached_tid_ = r.tid;
                run_cursor_ = 0;
            }
            // runs are split at multiples of step, so each belongs to exactly one tile
            while (run_cursor_ < cruns_.size() && cruns_[run_cursor_].start < r.start) ++run_cursor_;
            size_t e = run_cursor_;
            while (e < cruns_.size() && cruns_[e].start < r.end) ++e;
            const size_t w0 = (size_t)(r.start / W_), w1 = (size_t)((r.end + W_ - 1) / W_);
            emit_region(r.chrom.c_str(), r.start, r.end, csums_.data() + w0, w1 - w0, cruns_.data() + run_cursor_, e - run_cursor_);
            run_cursor_ = e;
        } else if (r.end > r.start) {
            ctx = ctx_of(r.tid);
            if (!batch_.empty() && shard_of_[(size_t)batch_[0]->tid] != shard_of_[(size_t)r.tid]) {
                ctx = ctx_of(batch_[0]->tid);
                GDCHK(flush_batch());
            }
            batch_.push_back(&r);
            if (batch_.size() >= kBatch) GDCHK(flush_batch());
        }
        if (rows_.hd.size() + rows_.ca.size() > (8u << 20)) io_ok = flush_rows(&rows_, fhd_, fca_) && io_ok;
        return 0;
    }

    int finish()
    {
        gd_ctx* ctx = batch_.empty() ? seq_ctx_ : ctx_of(batch_[0]->tid);
        GDCHK(flush_batch());
        ctx = seq_ctx_;
        if (stats_rc_ != GD_OK) GDCHK(stats_rc_);
#undef GDCHK
        io_ok = flush_rows(&rows_, fhd_, fca_) && io_ok;
        return 0;
    }

private:
    static constexpr size_t kBatch = 4096;
    const std::vector<int64_t>& lens_;
    Shards& S_;
    const std::vector<int>& shard_of_;
    gdh::FastaStats* const fa_;
    const int W_;
    FILE *const fhd_, *const fca_;
    const int stats_contract_;
    gd_ctx* const seq_ctx_;
    RowWriter rows_;
    std::vector<int64_t> sums_, csums_;
    std::vector<gd_run> runs_, cruns_;
    int cached_tid_ = -1;                                      // whole-contig results of the current contig
    size_t run_cursor_ = 0;
    std::vector<const Region*> batch_;                         // regions of ONE shard (flushed when the shard changes)
    std::string seq_chrom_, seq_bases_;
    bool seq_known_ = false;
    int64_t seq_line_bases_ = 0;
    int stats_rc_ = GD_OK;

    gd_ctx* ctx_of(int tid) const { return S_.v[(size_t)shard_of_[(size_t)tid]].ctx; }

    void emit_region(const char* chrom, int64_t rs, int64_t re, const int64_t* su, size_t n_su, const gd_run* ru, size_t n_ru)
    {
        if (!fa_) { format_region(&rows_, chrom, rs, re, W_, su, n_su, ru, n_ru, nullptr); return; }
        StatsPlan sp;
        RowWriter probe;
        format_region(&probe, chrom, rs, re, W_, su, n_su, ru, n_ru, &sp);
        if (seq_chrom_ != chrom || seq_chrom_.empty()) {
            seq_chrom_ = chrom;
            seq_known_ = fa_->contig_bases(chrom, &seq_bases_, &seq_line_bases_);
            if (seq_known_ && stats_rc_ == GD_OK)
                stats_rc_ = gd_seq_load(seq_ctx_, reinterpret_cast<const uint8_t*>(seq_bases_.data()), (int64_t)seq_bases_.size());
            seq_bases_.clear();
            seq_bases_.shrink_to_fit();
        }
        const size_t n = sp.s.size();
        std::vector<uint32_t> gc(n, 0), cpg(n, 0), low(n, 0), acgt(n, 0), lacgt(n, 0);
        if (seq_known_ && n && stats_rc_ == GD_OK) {
            const bool raw = (stats_contract_ & GDH_STATS_CPG_RAW_LINES) && seq_line_bases_ > 0 && seq_line_bases_ < 0x7fffffff;
            stats_rc_ = gd_seq_stats_ex(seq_ctx_, n, sp.s.data(), sp.e.data(), raw ? (int32_t)seq_line_bases_ : 0,
                                        gc.data(), cpg.data(), low.data(), acgt.data(), lacgt.data());
        }
        sp.cols.resize(n);
        for (size_t k = 0; k < n; ++k) {
            char buf[96];
            gdh_format_stats(stats_contract_, seq_known_ ? 1 : 0, sp.s[k], sp.e[k], gc[k], cpg[k], low[k], acgt[k], lacgt[k],
                             buf, sizeof buf);                               // :199
            sp.cols[k] = buf;
        }
        sp.collecting = false;
        format_region(&rows_, chrom, rs, re, W_, su, n_su, ru, n_ru, &sp);
    }

    // the collected regions through one gd_regions call; GD_* status
    int flush_batch()
    {
        if (batch_.empty()) return GD_OK;
        gd_ctx* const ctx = ctx_of(batch_[0]->tid);
        const size_t nb = batch_.size();
        std::vector<int32_t> b_tid(nb);
        std::vector<int64_t> b_start(nb), b_end(nb);
        size_t nw = 0;
        for (size_t k = 0; k < nb; ++k) {
            b_tid[k] = batch_[k]->tid; b_start[k] = batch_[k]->start; b_end[k] = batch_[k]->end;
            nw += (size_t)((batch_[k]->end - 1) / W_ - batch_[k]->start / W_ + 1);
        }
        std::vector<size_t> woff(nb + 1), roff(nb + 1);
        sums_.resize(nw);
        runs_.resize(std::max<size_t>(runs_.size(), 4 * nb + 1024));
        int rc = gd_regions(ctx, nb, b_tid.data(), b_start.data(), b_end.data(), sums_.data(), nullptr, nw, woff.data(),
                            runs_.data(), runs_.size(), roff.data());
        if (rc == GD_E_CAPACITY && woff[nb] <= nw && roff[nb] > runs_.size()) {
            runs_.resize(roff[nb]);
            rc = gd_regions(ctx, nb, b_tid.data(), b_start.data(), b_end.data(), sums_.data(), nullptr, nw, woff.data(),
                            runs_.data(), runs_.size(), roff.data());
        }
        if (rc != GD_OK) return rc;
        for (size_t k = 0; k < nb; ++k) {
            emit_region(batch_[k]->chrom.c_str(), batch_[k]->start, batch_[k]->end, sums_.data() + woff[k], woff[k + 1] - woff[k],
                        runs_.data() + roff[k], roff[k + 1] - roff[k]);
            if (rows_.hd.size() + rows_.ca.size() > (8u << 20)) io_ok = flush_rows(&rows_, fhd_, fca_) && io_ok;
        }
        batch_.clear();
        return GD_OK;
    }
};

int run(const DArgs& args)
{
    const auto t_run = std::chrono::steady_clock::now();
    int exit_code = 0;
    std::string err;
    EarlyContexts early;                                     // (declared first: joined and emptied last)
    std::vector<int> devices;
    if (!parse_devices(&devices, &err)) { fprintf(stderr, "goleft depth: %s\n", err.c_str()); return 1; }
    early.start(devices);
    gdh::BamReader bam;
    if (!bam.open(args.bam, args.processes, &err)) {
        fprintf(stderr, "goleft depth: %s\n", err.c_str());
        return 1;
    }
    const auto& contigs = bam.contigs();
    std::map<std::string, int> tid_of;
    for (size_t i = 0; i < contigs.size(); ++i) tid_of.emplace(contigs[i].name, (int)i);

    // ---- regions: --bed rows (:103-120) or .fai tiles (:122-159) ------------
    const int W = args.window_size;
    const int64_t step = step_for(W);
    std::vector<Region> regions;
    if (int rc = collect_regions(args, tid_of, step, &regions)) return rc;
    std::vector<int32_t> wanted;
    for (const Region& r : regions)
        if (r.tid >= 0) wanted.push_back(r.tid);
    std::sort(wanted.begin(), wanted.end());
    wanted.erase(std::unique(wanted.begin(), wanted.end()), wanted.end());

    // ---- outputs (:382-388) ---------------------------------------------------
    const std::string suffix = args.chrom.empty() ? "" : "." + args.chrom;
    const std::string ca_path = args.prefix + suffix + ".callable.bed";
    const std::string hd_path = args.prefix + suffix + ".depth.bed";
    FILE* fca = fopen(ca_path.c_str(), "w");
    FILE* fhd = fca ? fopen(hd_path.c_str(), "w") : nullptr;
    if (!fca || !fhd) {
        fprintf(stderr, "goleft depth: cannot create %s\n", fca ? hd_path.c_str() : ca_path.c_str());
        if (fca) fclose(fca);
        return 1;
    }
    struct Closer { FILE *a, *b; bool done = false; ~Closer() { if (!done) { fclose(a); fclose(b); } } } closer{fca, fhd};

    gdh::FastaStats fasta;
    gdh::FastaStats* fa = nullptr;
    if (args.stats) {                                                           // :244-252
        if (!fasta.open(args.reference, &err)) {
            fprintf(stderr, "goleft depth: %s\n", err.c_str());
            return 1;
        }
        fa = &fasta;
    }

    // ---- device engines: one context per device, contigs assigned by LPT -------------------
    Shards S;                                                // destroys every context on any return
    std::vector<int64_t> lens(contigs.size());
    for (size_t i = 0; i < contigs.size(); ++i) lens[i] = contigs[i].length;
    const size_t n_shards = std::max<size_t>(1, std::min(devices.size(), std::max<size_t>(wanted.size(), 1)));
    const std::vector<std::vector<int32_t>> assignment = lpt_assign(wanted, lens, n_shards);
    std::vector<int> shard_of(contigs.size(), 0);
    // fused device results are W/step aligned over the BAM contig length; a tile cut short by a
    // disagreeing .fai length, and every --bed row, goes through the region reductions instead,
    // which need the per-base vector; a whole-genome run with an agreeing .fai does not
    auto is_fused = [&](const Region& r) {
        if (r.tid < 0) return false;
        const int64_t clen = contigs[(size_t)r.tid].length;
        return args.bed.empty() && r.start % step == 0 && (r.end == clen || (r.end < clen && r.end % step == 0));
    };
    bool need_perbase = false;
    for (const Region& r : regions)
        if (r.tid >= 0 && r.end > r.start && !is_fused(r)) { need_perbase = true; break; }
    gd_params P;
    gd_default_params(&P);
    P.window_size = W;
    P.min_mapq = args.q;
    P.min_cov = args.min_cov;
    P.max_mean_depth = args.max_mean_depth;
    P.step = args.bed.empty() ? step : 0;
    if (int rc = configure_shards(devices, &early, assignment, lens, P, need_perbase, &S, &shard_of)) return rc;
    auto ctx_of = [&](int tid) -> gd_ctx* { return S.v[(size_t)shard_of[(size_t)tid]].ctx; };
    auto on_every_shard = [&](const char* what, const std::function<int(Shard&)>& fn) { return run_on_every_shard(S, what, fn); };

    // GOLEFT_DEPTH_TIMING=1: wall-clock phases on stderr (measurement only, SURVEY.md 8d scope iii)
    const bool timing = getenv("GOLEFT_DEPTH_TIMING") != nullptr;
    auto now = []() { return std::chrono::steady_clock::now(); };
    auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
        return std::chrono::duration<double>(b - a).count();
    };
    const auto t_begin = now();
    uint64_t n_gpu_records = 0;
    auto t_ingested = t_begin, t_computed = t_begin, t_indexed = t_begin, t_mapped = t_begin, t_read = t_begin;
    // ---- records into HBM: the device decoder when a .bai exists, else the host decoder through the pinned ring ----
    if (!wanted.empty()) {
        ReadTimes T{t_begin, t_begin, t_begin};
        if (int rc = read_records(args, bam, wanted, S, shard_of, &n_gpu_records, &T)) return rc;
        t_indexed = T.indexed; t_mapped = T.mapped; t_read = T.read;
        t_ingested = now();
        if (!on_every_shard("gd_compute", [&](Shard& sh) { return gd_compute(sh.ctx); })) return 1;
        t_computed = now();
    }

    // ---- rows, in input order (what --ordered gives; Q4) -------------------------
    bool io_ok = true;
    // the common case (a whole-genome run without --stats): emit_fused_genome formats the rows on all cores
    bool sliced_rows = !fa && args.bed.empty() && !regions.empty();
    for (size_t i = 0; sliced_rows && i < regions.size(); ++i)
        if (regions[i].tid < 0 || !is_fused(regions[i])) sliced_rows = false;
    if (sliced_rows) {
        if (int rc = emit_fused_genome(regions, lens, ctx_of, W, fhd, fca, &io_ok)) return rc;
    } else {
        RegionRows rr(lens, S, shard_of, fa, W, fhd, fca);
        for (const Region& r : regions)
            if (int rc = rr.add(r, r.tid >= 0 && is_fused(r))) return rc;
        if (int rc = rr.finish()) return rc;
        io_ok = rr.io_ok;
        exit_code = std::max(exit_code, rr.exit_code);
    }
    closer.done = true;
    if (fclose(fca) != 0) io_ok = false;
    if (fclose(fhd) != 0) io_ok = false;
    if (!io_ok) { fprintf(stderr, "goleft depth: write error\n"); return 1; }
    if (timing)
        fprintf(stderr, "{\"setup_s\": %.4f, \"index_s\": %.4f, \"map_s\": %.4f, \"read_s\": %.4f, \"decode_and_ingest_s\": %.4f, \"compute_s\": %.4f, \"rows_s\": %.4f, \"records\": %llu, \"decoder\": \"%s\"}\n",
                secs(t_run, t_begin), secs(t_begin, t_indexed), secs(t_indexed, t_mapped), secs(t_mapped, t_read), secs(t_begin, t_ingested), secs(t_ingested, t_computed), secs(t_computed, now()),
                (unsigned long long)(n_gpu_records ? n_gpu_records : bam.n_records()), n_gpu_records ? "device" : "host");
    return exit_code;
}

}  // namespace

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------
extern "C" {

int gdh_depth_main(int argc, const char* const* argv)
{
    DArgs args;                       // defaults: depth/depth.go:164-167
    const int rc = parse_args(argc, argv, &args);
    if (rc > 0) return 0;
    if (rc < 0) return 255;           // go-arg p.Fail -> os.Exit(-1)
    return run(args);
}

int gdh_chrom_start_end(const char* line, size_t len, char* chrom, size_t cap, int64_t* start, int64_t* end)
{
    std::string c;
    int64_t s = 0, e = 0;
    if (!line || !chrom || !start || !end) return -1;
    if (!chrom_start_end(line, len, &c, &s, &e)) return -1;
    if (c.size() + 1 > cap) return -1;
    memcpy(chrom, c.c_str(), c.size() + 1);
    *start = s;
    *end = e;
    return 0;
}

int64_t gdh_step(int32_t window_size) { return window_size > 0 ? step_for(window_size) : 0; }

int gdh_lpt_assign(const int32_t* tids, size_t n_tids, const int64_t* lengths, size_t n_contigs, size_t n_shards,
                   int32_t* shard_of_tid)
{
    if ((n_tids && (!tids || !shard_of_tid)) || !lengths || n_shards == 0) return -1;
    for (size_t i = 0; i < n_tids; ++i)
        if (tids[i] < 0 || (size_t)tids[i] >= n_contigs) return -1;
    const std::vector<int32_t> t(tids, tids + n_tids);
    const std::vector<int64_t> l(lengths, lengths + n_contigs);
    const auto a = lpt_assign(t, l, n_shards);
    for (size_t k = 0; k < a.size(); ++k)
        for (int32_t x : a[k])
            for (size_t i = 0; i < n_tids; ++i)
                if (tids[i] == x) shard_of_tid[i] = (int32_t)k;
    return 0;
}

int gdh_format_region(const char* chrom, int64_t rs, int64_t re, int32_t W, const int64_t* sums,
                      size_t n_sums, const gd_run* runs, size_t n_runs, const char* depth_path,
                      const char* callable_path)
{
    if (!chrom || W < 1 || !depth_path || !callable_path) return -1;
    RowWriter w;
    format_region(&w, chrom, rs, re, W, sums, n_sums, runs, n_runs, nullptr);
    FILE* fhd = fopen(depth_path, "a");
    if (!fhd) return -1;
    FILE* fca = fopen(callable_path, "a");
    if (!fca) { fclose(fhd); return -1; }
    const bool ok = flush_rows(&w, fhd, fca);
    const int c1 = fclose(fhd), c2 = fclose(fca);
    return ok && c1 == 0 && c2 == 0 ? 0 : -1;
}

}  // extern "C"
