// host_api.cpp -- C ABI wrappers: BAM decode and depth/intervals.go.
#include "../../../include/goleft_depth_host.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "bam_reader.hpp"
#include "gpu_ingest.hpp"

struct gdh_bam {
    gdh::BamReader rd;
    gdh::RecordBlock blk;
    std::string err;
};

// depth/intervals.go: per-chromosome interval sets.  The reference keeps a
// biogo interval tree per chromosome (intervals.go:42-79) and only ever asks
// "does anything overlap [s,e)?" (Overlaps, :25-39), so a start-sorted array
// with a running maximum of the ends answers the same query in O(log n).
struct gdh_intervals {
    struct Set {
        std::vector<std::pair<int64_t, int64_t>> iv;   // (start, end), sorted by start
        std::vector<int64_t> max_end;                  // prefix maximum of end
    };
    std::map<std::string, Set> by_chrom;
};

namespace {
// the `--stats` contract of this process (include/goleft_depth_host.h GDH_STATS_*): GOLEFT_STATS_CONTRACT
// at first use, gdh_set_stats_contract afterwards
int g_stats_contract = -1;
int stats_contract_from_env()
{
    const char* e = getenv("GOLEFT_STATS_CONTRACT");
    if (!e || !*e || !strcmp(e, "faidx")) return GDH_STATS_FAIDX;
    if (!strcmp(e, "window")) return GDH_STATS_WINDOW;
    char* end = nullptr;
    const long v = strtol(e, &end, 0);
    if (end == e || *end || v < 0 || (v & ~(long)GDH_STATS_FAIDX)) {
        fprintf(stderr, "goleft-depth: GOLEFT_STATS_CONTRACT=%s is not faidx, window or a mask of 0..%d; using faidx\n",
                e, GDH_STATS_FAIDX);
        return GDH_STATS_FAIDX;
    }
    return (int)v;
}
}  // namespace

extern "C" {

int gdh_produce_in_place(void* vctx, int32_t tid, const int32_t* pos, const uint16_t* flag, const uint8_t* mapq,
                         const uint32_t* cigar_off, const uint32_t* cigar, size_t n_reads, size_t n_ops,
                         int threads, size_t chunk)
{
    gd_ctx* ctx = static_cast<gd_ctx*>(vctx);
    if (!ctx || (n_reads && (!pos || !flag || !mapq || !cigar_off))) return GD_E_INVALID;
    if (threads < 1) threads = 1;
    if (chunk < 4096) chunk = 4096;
    if (int r = gd_reserve(ctx, tid, n_reads, n_ops)) return r;   // (a decoder has the count from the .bai metadata bin)
    if (!n_reads) return GD_OK;
    // the records are checked by the pass that indexes them on the device (GD_OPT_COMMIT_CHECK = 1) -- on the host that
    // is a second pass over every block by CPU threads, and CPU is what a decoder is short of; the verdict is collected
    // before this call returns, so its answers are the ones gd_commit would have given
    struct CheckOnDevice {
        gd_ctx* ctx; int64_t was = 0; bool set = false;
        explicit CheckOnDevice(gd_ctx* c) : ctx(c) { set = gd_get_option(c, GD_OPT_COMMIT_CHECK, &was) == GD_OK && gd_set_option(c, GD_OPT_COMMIT_CHECK, 1) == GD_OK; }
        ~CheckOnDevice() { if (set) (void)gd_set_option(ctx, GD_OPT_COMMIT_CHECK, was); }
    } scoped(ctx);
    // `threads` producers for the whole call (like decoder goroutines), each writing its share of every block, and
    // this thread, the only one that talks to the context: it holds kDepth blocks (gd_acquire hands out the next one
    // before the last is committed), so the producers write block k+1 while block k is validated and sent.  Two, not
    // three: the ring has four, and with three held only one copy can be in flight -- the link then idles while the
    // next block is validated (measured: 0.54 ms per 15 MB block instead of the link's 0.34)
    constexpr int kDepth = 2;
    struct Block {
        gd_batch b{};
        size_t i = 0, n = 0, o0 = 0, o1 = 0;
        std::atomic<int> filled{0};
    } blocks[kDepth];
    const size_t n_blocks = (n_reads + chunk - 1) / chunk;
    std::atomic<size_t> acquired{0};          // blocks [0, acquired) are described in `blocks` and may be written
    std::atomic<int> status{GD_OK};
    std::atomic<bool> stop{false};
    // waiting: a short spin (the other side is usually a fraction of a millisecond away), then asleep -- threads that
    // spin through sched_yield for a whole genome use up a container's CPU quota, and the kernel then stops ALL of the
    // process's threads for the rest of the accounting period (measured: 40 ms stalls, a third of the run)
    std::mutex mu;
    std::condition_variable cv;
    auto wait_for = [&](auto&& ready) {
        for (int spin = 0; spin < 4000; ++spin) {
            if (ready()) return;
            __builtin_ia32_pause();
        }
        std::unique_lock<std::mutex> lk(mu);
        while (!ready()) cv.wait_for(lk, std::chrono::microseconds(200));
    };
    auto wake = [&] { cv.notify_all(); };
    auto acquire = [&](size_t j) -> int {
        Block& k = blocks[j % kDepth];
        k.i = j * chunk;
        k.n = std::min(chunk, n_reads - k.i);
        k.o0 = cigar_off[k.i]; k.o1 = cigar_off[k.i + k.n];
        if (k.o1 < k.o0 || k.o1 > n_ops) return GD_E_INVALID;   // (before anything is sized or copied by these offsets)
        if (int r = gd_acquire(ctx, k.n, k.o1 - k.o0, &k.b)) return r;
        k.filled.store(0, std::memory_order_relaxed);
        acquired.store(j + 1, std::memory_order_release);
        wake();
        return GD_OK;
    };
    auto part = [&](const Block& k, int w) {
        const size_t a = k.n * (size_t)w / (size_t)threads, e = k.n * (size_t)(w + 1) / (size_t)threads;
        if (e <= a) return;
        const gd_batch& b = k.b;
        memcpy(b.pos + a, pos + k.i + a, (e - a) * sizeof(int32_t));
        memcpy(b.flag + a, flag + k.i + a, (e - a) * sizeof(uint16_t));
        memcpy(b.mapq + a, mapq + k.i + a, (e - a) * sizeof(uint8_t));
        const uint32_t* so = cigar_off + k.i;
        for (size_t r = a; r < e; ++r) b.cigar_off[r] = so[r] - (uint32_t)k.o0;
        if (e == k.n) b.cigar_off[k.n] = so[k.n] - (uint32_t)k.o0;
        const size_t ca = so[a], ce = so[e];
        // a share's ops must lie inside the block's [o0, o1): offsets that leave it would write outside the pinned block
        // before gd_commit's validation ever sees them
        if (ca < k.o0 || ce > k.o1 || ce < ca) { status.store(GD_E_INVALID); stop.store(true); wake(); return; }
        if (ce > ca) memcpy(b.cigar + (ca - k.o0), cigar + ca, (ce - ca) * sizeof(uint32_t));
    };
    auto producer = [&](int w) {
        for (size_t j = 0; j < n_blocks; ++j) {
            wait_for([&] { return acquired.load(std::memory_order_acquire) > j || stop.load(std::memory_order_relaxed); });
            if (stop.load(std::memory_order_relaxed)) return;
            Block& k = blocks[j % kDepth];
            part(k, w);
            if (k.filled.fetch_add(1, std::memory_order_release) + 1 == threads) wake();
        }
    };
    size_t next = 0, held_from = 0;           // blocks [held_from, next) are held
    int rc = GD_OK;
    for (; next < n_blocks && next < (size_t)kDepth && rc == GD_OK; ) {
        rc = acquire(next);
        if (rc == GD_OK) ++next;
    }
    std::vector<std::thread> th;
    if (rc == GD_OK)
        for (int w = 0; w < threads; ++w) th.emplace_back(producer, w);
    for (size_t j = 0; j < n_blocks && rc == GD_OK; ++j) {
        Block& k = blocks[j % kDepth];
        wait_for([&] { return k.filled.load(std::memory_order_acquire) >= threads || stop.load(std::memory_order_relaxed); });
        if ((rc = status.load()) != GD_OK) break;
        rc = gd_commit(ctx, &k.b, tid, k.n, k.o1 - k.o0);
        held_from = j + 1;
        if (rc == GD_OK && next < n_blocks) {
            rc = acquire(next);
            if (rc == GD_OK) ++next;
        }
    }
    if (rc != GD_OK) { stop.store(true); wake(); }
    for (auto& t : th) t.join();
    for (size_t j = held_from; j < next; ++j)             // (an error: the blocks still out go back unused)
        (void)gd_commit(ctx, &blocks[j % kDepth].b, tid, 0, 0);
    if (rc == GD_OK) rc = status.load();
    if (rc == GD_OK) rc = gd_check_commits(ctx);
    return rc;
}

static int g_fast_exit = 0;
int gdh_set_fast_exit(int on) { g_fast_exit = on != 0; return 0; }
int gdh_get_fast_exit(void) { return g_fast_exit; }

int gdh_set_stats_contract(int contract)
{
    if (contract < 0 || (contract & ~GDH_STATS_FAIDX)) return -1;
    g_stats_contract = contract;
    return 0;
}

int gdh_get_stats_contract(void)
{
    if (g_stats_contract < 0) g_stats_contract = stats_contract_from_env();
    return g_stats_contract;
}

int gdh_format_stats(int contract, int known, int64_t start, int64_t end, uint32_t n_gc, uint32_t n_cpg,
                     uint32_t n_masked, uint32_t n_acgt, uint32_t n_masked_acgt, char* out, size_t cap)
{
    // depth/depth.go:191-200; the forks are the named switches of goleft_depth_host.h
    const double tot = (contract & GDH_STATS_DENOM_ACGT) ? (double)n_acgt : (double)(end - start);
    double gc = 0, cpg = 0, masked = 0;
    if (known && end > start && tot > 0) {
        gc = n_gc / tot;
        cpg = 2.0 * n_cpg / tot;
        if ((contract & GDH_STATS_CPG_CLAMP) && cpg > 1.0) cpg = 1.0;
        masked = ((contract & GDH_STATS_MASKED_ACGT) ? n_masked_acgt : n_masked) / tot;
    }
    return snprintf(out, cap, "\t%.3g\t%.3g\t%.3g", gc, cpg, masked);
}

int64_t gdh_list_members(const uint8_t* data, size_t n_bytes, uint64_t beg, const uint64_t* member_starts, size_t n_starts,
                         unsigned threads, size_t min_bytes, size_t cap, uint64_t* off, uint32_t* size, uint16_t* hdr,
                         uint32_t* isize, uint32_t* crc)
{
    if (!data || (n_starts && !member_starts)) return -1;
    gdh::MemberTable t;
    const std::vector<uint64_t> st(member_starts, member_starts + n_starts);
    if (!gdh::list_members(data, n_bytes, beg, st, &t, threads, min_bytes)) return -1;
    for (size_t k = 0; k < t.n && k < cap; ++k) {
        if (off) off[k] = t.off[k];
        if (size) size[k] = t.size[k];
        if (hdr) hdr[k] = t.hdr[k];
        if (isize) isize[k] = t.isize[k];
        if (crc) crc[k] = t.crc[k];
    }
    return (int64_t)t.n;
}

int64_t gdh_list_members_fd(int fd, uint64_t beg, size_t n_bytes, const uint64_t* member_starts, size_t n_starts,
                            unsigned threads, size_t min_bytes, size_t cap, uint64_t* off, uint32_t* size, uint16_t* hdr,
                            uint32_t* isize, uint32_t* crc)
{
    if (fd < 0 || (n_starts && !member_starts)) return -1;
    gdh::MemberTable t;
    const std::vector<uint64_t> st(member_starts, member_starts + n_starts);
    if (!gdh::list_members(nullptr, n_bytes, beg, st, &t, threads, min_bytes, fd)) return -1;
    for (size_t k = 0; k < t.n && k < cap; ++k) {
        if (off) off[k] = t.off[k];
        if (size) size[k] = t.size[k];
        if (hdr) hdr[k] = t.hdr[k];
        if (isize) isize[k] = t.isize[k];
        if (crc) crc[k] = t.crc[k];
    }
    return (int64_t)t.n;
}

size_t gdh_plan_ingest_passes(const uint64_t* start, const uint8_t* has, size_t n_refs, const int32_t* wanted,
                              size_t n_wanted, uint64_t file_size, uint64_t group_bytes, size_t cap,
                              uint64_t* first, uint64_t* last, uint64_t* beg, uint64_t* end)
{
    if ((n_refs && (!start || !has)) || (n_wanted && !wanted)) return 0;
    for (size_t k = 0; k < n_wanted; ++k)
        if (wanted[k] < 0 || (size_t)wanted[k] >= n_refs) return 0;
    const std::vector<uint64_t> st(start, start + n_refs);
    const std::vector<char> hs(has, has + n_refs);
    const std::vector<int32_t> w(wanted, wanted + n_wanted);
    const std::vector<gdh::IngestPass> ps = gdh::plan_ingest_passes(st, hs, w, file_size, group_bytes);
    for (size_t k = 0; k < ps.size() && k < cap; ++k) {
        if (first) first[k] = ps[k].first;
        if (last) last[k] = ps[k].last;
        if (beg) beg[k] = ps[k].beg;
        if (end) end[k] = ps[k].end;
    }
    return ps.size();
}

// Test hook: how ONE reference with the linear-index anchors `anchors` (virtual offsets of record starts, strictly
// ascending), whose records end by file offset `end`, is cut into parts of about part_bytes (gdh::plan_ingest_passes).
size_t gdh_plan_ingest_parts(const uint64_t* anchors, size_t n_anchors, uint64_t end, uint64_t part_bytes, size_t cap,
                             uint64_t* a_lo, uint64_t* a_hi, uint64_t* beg, uint64_t* pend, double* scale)
{
    if (!anchors || n_anchors == 0) return 0;
    std::vector<std::vector<uint64_t>> lin(1, std::vector<uint64_t>(anchors, anchors + n_anchors));
    const std::vector<uint64_t> st{anchors[0] >> 16};
    const std::vector<char> hs{1};
    const std::vector<int32_t> w{0};
    const std::vector<gdh::IngestPass> ps = gdh::plan_ingest_passes(st, hs, w, end, ~0ull, nullptr, &lin, part_bytes);
    for (size_t k = 0; k < ps.size() && k < cap; ++k) {
        if (a_lo) a_lo[k] = ps[k].part ? ps[k].a_lo : 0;
        if (a_hi) a_hi[k] = ps[k].part ? ps[k].a_hi : n_anchors;
        if (beg) beg[k] = ps[k].beg;
        if (pend) pend[k] = ps[k].end;
        if (scale) scale[k] = ps[k].scale;
    }
    return ps.size();
}

int gdh_bam_open(const char* path, int threads, gdh_bam** out)
{
    if (!path || !out) return -1;
    gdh_bam* b = new (std::nothrow) gdh_bam();
    if (!b) return -1;
    *out = b;                                   // returned even on failure so the error can be read
    return b->rd.open(path, threads, &b->err) ? 0 : -1;
}

void gdh_bam_close(gdh_bam* b) { delete b; }
const char* gdh_bam_error(const gdh_bam* b) { return b ? b->err.c_str() : "null"; }
int gdh_bam_n_contigs(const gdh_bam* b) { return b ? (int)b->rd.contigs().size() : 0; }

const char* gdh_bam_contig_name(const gdh_bam* b, int tid)
{
    if (!b || tid < 0 || (size_t)tid >= b->rd.contigs().size()) return nullptr;
    return b->rd.contigs()[(size_t)tid].name.c_str();
}

int64_t gdh_bam_contig_length(const gdh_bam* b, int tid)
{
    if (!b || tid < 0 || (size_t)tid >= b->rd.contigs().size()) return -1;
    return b->rd.contigs()[(size_t)tid].length;
}

int gdh_bam_seek_contig(gdh_bam* b, int tid)
{
    if (!b) return -1;
    return b->rd.seek_contig(tid, &b->err) ? 1 : 0;
}

int gdh_bam_next(gdh_bam* b, size_t max_reads, int32_t* tid, size_t* n_reads, size_t* n_ops,
                 const int32_t** pos, const uint16_t** flag, const uint8_t** mapq,
                 const uint32_t** cigar_off, const uint32_t** cigar)
{
    if (!b || !tid || !n_reads || !n_ops) return -1;
    const int rc = b->rd.next_block(b->blk, max_reads ? max_reads : 1, &b->err);
    if (rc <= 0) { *n_reads = 0; *n_ops = 0; return rc; }
    *tid = b->blk.tid;
    *n_reads = b->blk.size();
    *n_ops = b->blk.cigar.size();
    if (pos) *pos = b->blk.pos.data();
    if (flag) *flag = b->blk.flag.data();
    if (mapq) *mapq = b->blk.mapq.data();
    if (cigar_off) *cigar_off = b->blk.cigar_off.data();
    if (cigar) *cigar = b->blk.cigar.data();
    return 1;
}

uint64_t gdh_bam_n_records(const gdh_bam* b) { return b ? b->rd.n_records() : 0; }

int gdh_intervals_read(const char* const* paths, int n_paths, gdh_intervals** out)
{
    if (!out || n_paths < 0) return -1;
    gdh_intervals* t = new (std::nothrow) gdh_intervals();
    if (!t) return -1;
    for (int i = 0; i < n_paths; ++i) {
        if (!paths[i] || !paths[i][0]) continue;                 // intervals.go:46-48
        FILE* f = fopen(paths[i], "r");
        if (!f) { delete t; return -1; }                         // the reference panics
        char* line = nullptr;
        size_t cap = 0;
        ssize_t n;
        while ((n = getline(&line, &cap, f)) > 0) {
            char chrom[4096];
            int64_t s, e;
            if (gdh_chrom_start_end(line, (size_t)n, chrom, sizeof chrom, &s, &e) != 0) {
                free(line); fclose(f); delete t;
                return -1;                                       // log.Fatal in the reference
            }
            if (s >= e) continue;                                // intervals.go:66
            t->by_chrom[chrom].iv.emplace_back(s, e);
        }
        free(line);
        fclose(f);
    }
    for (auto& kv : t->by_chrom) {
        auto& set = kv.second;
        std::sort(set.iv.begin(), set.iv.end());
        set.max_end.resize(set.iv.size());
        int64_t m = INT64_MIN;
        for (size_t i = 0; i < set.iv.size(); ++i) { m = std::max(m, set.iv[i].second); set.max_end[i] = m; }
    }
    *out = t;
    return 0;
}

void gdh_intervals_free(gdh_intervals* t) { delete t; }

int gdh_intervals_overlaps(const gdh_intervals* t, const char* chrom, int64_t start, int64_t end)
{
    if (!t || !chrom) return 0;                                  // nil tree -> false (intervals.go:26-28)
    auto it = t->by_chrom.find(chrom);
    if (it == t->by_chrom.end()) return 0;
    const auto& set = it->second;
    // half-open: i.End > q.Start && i.Start < q.End (intervals.go:16-19)
    const size_t k = (size_t)(std::lower_bound(set.iv.begin(), set.iv.end(), std::make_pair(end, INT64_MIN)) - set.iv.begin());
    if (k == 0) return 0;
    return set.max_end[k - 1] > start ? 1 : 0;
}

size_t gdh_intervals_count(const gdh_intervals* t, const char* chrom)
{
    if (!t || !chrom) return 0;
    auto it = t->by_chrom.find(chrom);
    return it == t->by_chrom.end() ? 0 : it->second.iv.size();
}

}  // extern "C"
