// gd_api.hip -- C ABI of the per-base depth engine (include/goleft_depth.h).
//
// Host-side runtime around the CDNA4 kernels of gd_kernels.hpp: contexts,
// pinned staging ring + copy stream (records host -> HBM), HBM-resident
// per-contig record streams, launch sequencing, result read-back.
// Replaces gargs' process.Runner + the samtools child + the callback's parse
// loop (/root/reference/depth/depth.go:392-394, :45, :282-325).
#include "../../include/goleft_depth.h"
#include "gd_kernels.hpp"

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include <unistd.h>

#include "gd_api_state.hpp"

extern "C" {

const char* gd_strerror(int s)
{
    switch (s) {
    case GD_OK: return "ok";
    case GD_E_INVALID: return "invalid argument";
    case GD_E_NOMEM: return "out of memory";
    case GD_E_HIP: return "HIP runtime error";
    case GD_E_STATE: return "call out of order";
    case GD_E_RANGE: return "tid or coordinate out of range";
    case GD_E_NODEVICE: return "no usable HIP device";
    case GD_E_UNSORTED: return "records not coordinate sorted";
    case GD_E_CAPACITY: return "output buffer too small";
    }
    return "unknown status";
}

int gd_abi_version(void) { return GD_ABI_VERSION; }

int gd_device_count(int* n)
{
    if (!n) return GD_E_INVALID;
    int k = 0;
    hipError_t e = hipGetDeviceCount(&k);
    if (e != hipSuccess) { *n = 0; return GD_E_NODEVICE; }
    *n = k;
    return GD_OK;
}

int gd_default_params(gd_params* p)
{
    if (!p) return GD_E_INVALID;
    // depth/depth.go:164-167
    p->window_size = 250;
    p->min_mapq = 1;
    p->min_cov = 4;
    p->max_mean_depth = 0;
    p->flag_mask = GD_DEFAULT_FLAG_MASK;
    p->max_span_hint = 0;
    p->step = 0;
    return GD_OK;
}

int gd_create(int device_id, gd_ctx** out)
{
    if (!out) return GD_E_INVALID;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return GD_E_NODEVICE;
    if (device_id < 0 || device_id >= n) return GD_E_RANGE;
    gd_ctx* c = new (std::nothrow) gd_ctx();
    if (!c) return GD_E_NOMEM;
    c->device = device_id;
    gd_default_params(&c->params);
    auto bail = [&](hipError_t e) {
        (void)e;
        gd_destroy(c);
        return GD_E_HIP;
    };
    hipError_t e;
    if ((e = hipSetDevice(device_id)) != hipSuccess) return bail(e);
    if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) return bail(e);
    if ((e = hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking)) != hipSuccess) return bail(e);
    if ((e = hipEventCreateWithFlags(&c->copy_done, hipEventDisableTiming)) != hipSuccess) return bail(e);
    for (auto& s : c->ring)
        if ((e = hipEventCreateWithFlags(&s.done, hipEventDisableTiming)) != hipSuccess) return bail(e);
    for (auto& ev : c->ev)
        if ((e = hipEventCreate(&ev)) != hipSuccess) return bail(e);
    if ((e = hipMalloc(reinterpret_cast<void**>(&c->d_counters), sizeof(gd::Counters))) != hipSuccess) return bail(e);
    if ((e = hipMalloc(reinterpret_cast<void**>(&c->d_region_cursor), sizeof(uint32_t))) != hipSuccess) return bail(e);
    if ((e = hipHostMalloc(reinterpret_cast<void**>(&c->h_counters), sizeof(gd::Counters), hipHostMallocDefault)) != hipSuccess) return bail(e);
    if ((e = hipHostMalloc(reinterpret_cast<void**>(&c->h_bounds), kSpecBounds * sizeof(int2), hipHostMallocDefault)) != hipSuccess) return bail(e);
    if ((e = hipMemset(c->d_counters, 0, sizeof(gd::Counters))) != hipSuccess) return bail(e);
    if ((e = hipMalloc(reinterpret_cast<void**>(&c->d_ingest), 4 * sizeof(uint32_t))) != hipSuccess) return bail(e);
    if ((e = hipMemset(c->d_ingest, 0, 4 * sizeof(uint32_t))) != hipSuccess) return bail(e);
    if ((e = hipHostMalloc(reinterpret_cast<void**>(&c->h_ingest), 4 * sizeof(uint32_t), hipHostMallocDefault)) != hipSuccess) return bail(e);
    // Warm-up: one compute of an EMPTY two-tile contig runs every kernel of the tile path once (prep, slow list, straight-
    // line tile kernel, run ordering, read-back) and makes the small fixed-size allocations -- the first launch of a kernel in
    // a process costs tens of microseconds that a job of chr20's size (0.14 ms a step) saw as 1.3 x in its first, usually
    // only, compute.  A context comes up on a thread of its own while the host reads the BAM header (0.2-0.3 s): the
    // millisecond belongs there.  Nothing of it stays: no contigs, no records, default look-back.
    // A warm-up that fails is a device that fails: the context is not handed out in a state the caller did not ask for (ADVICE
    // r5: a phantom contig, a cleared error).  GOLEFT_NO_WARMUP=1 skips it (tests that create contexts by the hundred).
    if (const char* nw = getenv("GOLEFT_NO_WARMUP"); !(nw && *nw == '1')) {
        const int64_t len = 2 * 4096;
        int rc = gd_set_contigs(c, 1, &len);
        if (rc == GD_OK) rc = gd_compute(c);
        const int rc2 = gd_set_contigs(c, 0, nullptr);
        if (rc != GD_OK || rc2 != GD_OK) {
            fprintf(stderr, "gd_create: the warm-up compute failed (%d / %d): %s\n", rc, rc2, c->err.c_str());
            gd_destroy(c);
            return rc != GD_OK ? rc : rc2;
        }
        c->stats = gd_stats{};
        for (double& t : c->timing) t = 0;
    }
    *out = c;
    return GD_OK;
}

namespace { static void drop_pool(gd_ctx* c); }

void gd_destroy(gd_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);
    (void)gd_ingest_abort(c);
    (void)gd_comm_destroy(c);
    for (hipEvent_t e : c->comm_ev) if (e) (void)hipEventDestroy(e);
    for (int k = 0; k < 8; ++k) {
        if (c->ing_stage[k]) (void)hipHostFree(c->ing_stage[k]);
        if (c->ing_staged[k]) (void)hipEventDestroy(c->ing_staged[k]);
    }
    for (hipStream_t st : c->ing_stream) if (st) (void)hipStreamDestroy(st);
    for (hipStream_t st : c->ing_dma) if (st) (void)hipStreamDestroy(st);
    if (c->ing_hp) (void)hipStreamDestroy(c->ing_hp);
    if (c->ing_walk) (void)hipStreamDestroy(c->ing_walk);
    if (c->ing_walk_ev) (void)hipEventDestroy(c->ing_walk_ev);
    for (auto& evs : c->ing_dma_ev) for (hipEvent_t e : evs) if (e) (void)hipEventDestroy(e);
    for (auto& b : c->ing_bufs) b.drop_all();
    if (c->h_walk) (void)hipHostFree(c->h_walk);
    if (c->d_rectab) (void)hipFree(c->d_rectab);
    if (c->h_ingest) (void)hipHostFree(c->h_ingest);
    for (auto& h : c->contigs) free_contig(h);
    for (auto& s : c->ring) {
        if (s.b.pos) (void)hipHostFree(s.b.pos);
        if (s.b.flag) (void)hipHostFree(s.b.flag);
        if (s.b.mapq) (void)hipHostFree(s.b.mapq);
        if (s.b.cigar_off) (void)hipHostFree(s.b.cigar_off);
        if (s.b.cigar) (void)hipHostFree(s.b.cigar);
        if (s.done) (void)hipEventDestroy(s.done);
    }
    for (auto& ev : c->ev) if (ev) (void)hipEventDestroy(ev);
    if (c->copy_done) (void)hipEventDestroy(c->copy_done);
    void* frees[] = {c->d_ctgs, c->d_tiles, c->d_ftiles, c->d_perbase, c->d_wsum, c->d_wmin, c->d_chunks,
                     c->d_ordered, c->d_tile_cnt, c->d_tile_off, c->d_super_cnt, c->d_counters,
                     c->d_region_cursor, c->d_status, c->d_seq, c->d_md_bits, c->d_md_cnt, c->d_wed, c->d_scan_tmp, c->d_ingest};
    for (void* p : frees) if (p) (void)hipFree(p);
    if (c->h_counters) (void)hipHostFree(c->h_counters);
    if (c->h_bounds) (void)hipHostFree(c->h_bounds);
    if (c->h_batch) (void)hipHostFree(c->h_batch);
    drop_pool(c);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    delete c;
}

const char* gd_last_error(const gd_ctx* c) { return c ? c->err.c_str() : "null context"; }

int gd_set_stream(gd_ctx* c, void* s)
{
    if (!c) return GD_E_INVALID;
    if (int r = set_device(c)) return r;
    if (c->own_stream && c->stream) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, hipStreamDestroy(c->stream));
    }
    if (s) {
        c->stream = static_cast<hipStream_t>(s);
        c->own_stream = false;
    } else {
        HIPCHK(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        c->own_stream = true;
    }
    return GD_OK;
}

int gd_set_params(gd_ctx* c, const gd_params* p)
{
    if (c && c->cs.pending) return fail(c, GD_E_STATE, "a compute is in flight: gd_compute_finish first");
    if (!c || !p) return GD_E_INVALID;
    if (p->window_size < 1) return fail(c, GD_E_INVALID, "window_size must be >= 1");
    if (p->step < 0) return fail(c, GD_E_INVALID, "step must be >= 0");
    if (p->step > 0 && p->step % p->window_size != 0)
        return fail(c, GD_E_INVALID, "step must be a multiple of window_size");
    c->params = *p;
    if (p->max_span_hint > 0) { c->lookback = p->max_span_hint; c->lookback_pinned = true; }
    else c->lookback_pinned = false;
    c->computed = false;
    return GD_OK;
}

int gd_set_path(gd_ctx* c, int path)
{
    if (c && c->cs.pending) return fail(c, GD_E_STATE, "a compute is in flight: gd_compute_finish first");
    if (!c) return GD_E_INVALID;
    if (path != GD_PATH_AUTO && path != GD_PATH_TILE && path != GD_PATH_SCATTER && path != GD_PATH_CHUNK)
        return fail(c, GD_E_INVALID, "unknown path %d", path);
    c->path = path;
    c->computed = false;
    return GD_OK;
}

int gd_set_outputs(gd_ctx* c, unsigned flags)
{
    if (c && c->cs.pending) return fail(c, GD_E_STATE, "a compute is in flight: gd_compute_finish first");
    if (!c) return GD_E_INVALID;
    if (flags & ~(unsigned)(GD_OUT_PERBASE | GD_OUT_SUMS_ONLY)) return fail(c, GD_E_INVALID, "unknown output flags 0x%x", flags);
    if ((flags & GD_OUT_PERBASE) && (flags & GD_OUT_SUMS_ONLY))
        return fail(c, GD_E_INVALID, "GD_OUT_SUMS_ONLY excludes GD_OUT_PERBASE");
    c->keep_perbase = (flags & GD_OUT_PERBASE) != 0;
    c->sums_only = (flags & GD_OUT_SUMS_ONLY) != 0;
    c->computed = false;
    return GD_OK;
}

int gd_set_contigs(gd_ctx* c, int n, const int64_t* lengths)
{
    if (c && c->cs.pending) return fail(c, GD_E_STATE, "a compute is in flight: gd_compute_finish first");
    if (!c || n < 0 || (n > 0 && !lengths)) return GD_E_INVALID;
    if (int r = set_device(c)) return r;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipStreamSynchronize(c->copy_stream));
    for (auto& h : c->contigs) free_contig(h);
    c->contigs.assign(n, ContigHost());
    for (int i = 0; i < n; ++i) {
        if (lengths[i] < 0 || lengths[i] > GD_MAX_CONTIG_LENGTH)
            return fail(c, GD_E_RANGE, "contig %d length %lld out of range", i, (long long)lengths[i]);
        c->contigs[i].length = lengths[i];
    }
    c->selected.clear();
    c->computed = false;
    // a new data set: forget the look-back learnt from the previous one
    c->lookback = c->params.max_span_hint > 0 ? c->params.max_span_hint : kDefaultLookback;
    c->span_forces_long = false;
    HIPCHK(c, hipMemsetAsync(c->d_ingest, 0, 4 * sizeof(uint32_t), c->stream));   // spans of records that are gone
    HIPCHK(c, hipStreamSynchronize(c->stream));            // (the next block's index pass runs on the copy stream: not before this)
    c->ingest_span = 0;
    c->ingest_span_dirty = false;
    for (auto& s : c->ring) s.held = false;
    c->commit_checks_pending = false;
    return GD_OK;
}

int gd_select_contigs(gd_ctx* c, int n, const int32_t* tids)
{
    if (c && c->cs.pending) return fail(c, GD_E_STATE, "a compute is in flight: gd_compute_finish first");
    if (!c || n < 0 || (n > 0 && !tids)) return GD_E_INVALID;
    std::vector<int32_t> s(tids, tids + n);
    for (int32_t t : s)
        if (t < 0 || (size_t)t >= c->contigs.size()) return fail(c, GD_E_RANGE, "tid %d out of range", t);
    std::sort(s.begin(), s.end());
    s.erase(std::unique(s.begin(), s.end()), s.end());
    c->selected.swap(s);
    c->computed = false;
    return GD_OK;
}

// Worker threads of a context (created by the first large gd_push / gd_commit, kept until gd_destroy): the staging blocks are filled by several threads (one core moves ~11 GB/s into pinned
// memory, a Gen5 x16 link takes five times that), block k + 1 while the copies of block k are on the link.
namespace {
struct FillPool {
    // kind 0: plain copy; 1: int32 positions, copied and checked (non-decreasing from `prev`, not negative);
    // 2: 32-bit CSR offsets, rebased by -sub and checked (non-decreasing from `prev`); 3 / 4: the checks of 1 / 2 alone (gd_commit:
    // the caller filled the block itself); 5: `bytes` from offset `src` of the file descriptor `sub` (gd_ingest_feed_fd)
    struct Item { void* dst; const void* src; size_t bytes; uint32_t sub; int kind; int32_t prev; };
    std::atomic<uint32_t> bad{0};
    std::vector<std::thread> th;
    std::vector<Item> items;
    std::atomic<size_t> next{0}, done{0};
    std::atomic<uint64_t> gen{0};
    std::atomic<bool> quit{false};
    std::mutex mu;
    std::condition_variable cv;
    void run_item(const Item& it)
    {
        if (it.kind == 0) { memcpy(it.dst, it.src, it.bytes); return; }
        if (it.kind == 5) {
            size_t got = 0;
            while (got < it.bytes) {
                const ssize_t r = pread((int)it.sub, static_cast<char*>(it.dst) + got, it.bytes - got,
                                        (off_t)(reinterpret_cast<uintptr_t>(it.src) + got));
                if (r <= 0) { bad.store(1); return; }
                got += (size_t)r;
            }
            return;
        }
        const size_t n = it.bytes / 4;
        uint32_t wrong = 0;
        if (it.kind == 3) {
            const int32_t* __restrict__ s = static_cast<const int32_t*>(it.src);
            if (n) wrong = (uint32_t)(s[0] < it.prev) | (uint32_t)(s[0] < 0);
            for (size_t k = 1; k < n; ++k) wrong |= (uint32_t)(s[k] < s[k - 1]);
        } else if (it.kind == 4) {
            const uint32_t* __restrict__ s = static_cast<const uint32_t*>(it.src);
            for (size_t k = 1; k < n; ++k) wrong |= (uint32_t)(s[k] < s[k - 1]);
        } else if (it.kind == 1) {
            int32_t* __restrict__ d = static_cast<int32_t*>(it.dst);
            const int32_t* __restrict__ s = static_cast<const int32_t*>(it.src);
            if (n) { wrong = (uint32_t)(s[0] < it.prev) | (uint32_t)(s[0] < 0); d[0] = s[0]; }
            for (size_t k = 1; k < n; ++k) { wrong |= (uint32_t)(s[k] < s[k - 1]); d[k] = s[k]; }
        } else {
            uint32_t* __restrict__ d = static_cast<uint32_t*>(it.dst);
            const uint32_t* __restrict__ s = static_cast<const uint32_t*>(it.src);
            // (`prev` = the offset in front of this item, as bits: items are cut every 1 MB without overlap, and a dip exactly
            // at a cut -- possibly below `sub`, so that the rebased offset wraps -- must not pass)
            if (n) { wrong = (uint32_t)(s[0] < (uint32_t)it.prev); d[0] = s[0] - it.sub; }
            for (size_t k = 1; k < n; ++k) { wrong |= (uint32_t)(s[k] < s[k - 1]); d[k] = s[k] - it.sub; }
        }
        if (wrong) bad.store(1);
    }
    void drain()
    {
        for (;;) {
            const size_t i = next.fetch_add(1);
            if (i >= items.size()) break;
            run_item(items[i]);
            done.fetch_add(1);
        }
    }
    std::atomic<int> active{0};
    // how long a worker spins for the next batch before it sleeps: gd_push's blocks follow each other within a fraction of a
    // millisecond (20 000); the pieces of a device BAM read are 2 ms apart, and fifteen workers spinning through that use up
    // CPU time a container's quota then takes from the threads that read the file (gd_ingest_feed_fd sets 500)
    std::atomic<int> spin_limit{20000};
    void start(int n)
    {
        for (int k = 0; k < n; ++k)
            th.emplace_back([this] {
                uint64_t seen = 0;
                for (;;) {
                    // the next block of a push follows within a fraction of a millisecond: spin that long before sleeping
                    // (a condition-variable wake-up costs tens of microseconds per worker and block)
                    for (int spin = 0; spin < spin_limit.load(std::memory_order_relaxed) && gen.load(std::memory_order_relaxed) == seen && !quit.load(std::memory_order_relaxed); ++spin)
                        __builtin_ia32_pause();
                    {
                        std::unique_lock<std::mutex> lk(mu);
                        cv.wait(lk, [&] { return quit.load() || gen.load() != seen; });
                        if (quit.load()) return;
                        seen = gen.load();
                        active.fetch_add(1);               // (under the lock: run() never swaps the items under a worker)
                    }
                    drain();
                    active.fetch_sub(1);
                }
            });
    }
    // runs the items on the workers and the calling thread; returns when all are done
    void run(std::vector<Item>&& work)
    {
        for (;;) {
            std::unique_lock<std::mutex> lk(mu);
            if (active.load() != 0) { lk.unlock(); std::this_thread::yield(); continue; }
            items = std::move(work);
            next.store(0); done.store(0);
            gen.fetch_add(1);
            break;
        }
        cv.notify_all();
        drain();
        while (done.load() < items.size()) std::this_thread::yield();
    }
    ~FillPool()
    {
        { std::lock_guard<std::mutex> lk(mu); quit.store(true); }
        cv.notify_all();
        for (auto& t : th) t.join();
    }
};

static void drop_pool(gd_ctx* c) { delete c->pool; c->pool = nullptr; c->pool_workers = 0; }

// the context's pool, with push_threads - 1 workers (the calling thread works too)
static FillPool* ctx_pool(gd_ctx* c)
{
    const int want = c->push_threads - 1;
    if (c->pool && c->pool_workers != want) { delete c->pool; c->pool = nullptr; }
    if (!c->pool && want > 0) {
        c->pool = new (std::nothrow) FillPool();
        if (c->pool) { c->pool->start(want); c->pool_workers = want; }
    }
    return c->pool;
}
}  // namespace


// gd_index_records_kernel over the reads [r0, r1) of a contig that are resident (or will be, in stream order) on `st`:
// position index (allocated on first use), spans, and -- check != 0 -- the record checks.
static int index_records(gd_ctx* c, ContigHost& h, size_t r0, size_t r1, int32_t prev_pos, bool check, hipStream_t st, bool committed = false)
{
    if (r1 <= r0) return GD_OK;
    const size_t n_idx = (size_t)(h.length >> 6) + 2;
    const bool idx = c->ingest_index && h.ridx_reads == r0;      // (an index with a hole is no index)
    if (idx && !h.ridx) HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&h.ridx), n_idx * sizeof(uint32_t)));
    if (!idx && !check) return GD_OK;
    gd::norm::IndexJob j{};
    j.pos = h.pos; j.off = h.off; j.cigar = h.cigar;
    j.ridx = idx ? h.ridx : nullptr;
    j.n_idx = (uint32_t)n_idx;
    j.out = c->d_ingest;
    j.bad_out = c->d_ingest + (committed ? 3 : 0);
    j.r0 = (uint32_t)r0; j.r1 = (uint32_t)r1;
    j.n_reads_total = (uint32_t)r1;
    j.n_ops_total = (uint32_t)std::min<size_t>(h.n_ops, 0xffffffffu);
    j.prev_pos = r0 ? prev_pos : -1;
    j.check = check ? 1u : 0u;
    // spans are measured for short-read shaped data only (a lane walks its read's ops one by one)
    j.walk_ops = (c->ingest_index && h.n_ops <= 6 * r1) ? 1u : 0u;          // (r1 = the contig's records once this block is in)
    hipLaunchKernelGGL(gd::norm::gd_index_records_kernel, dim3((unsigned)((r1 - r0 + 255) / 256)), dim3(256), 0, st, j);
    HIPCHK(c, hipGetLastError());
    if (idx) h.ridx_reads = r1;
    if (j.walk_ops) c->ingest_span_dirty = true;
    return GD_OK;
}

// d_ingest's words on the host: a one-wave kernel stores them into page-locked memory and the stream is waited for -- no
// copy command (a device-to-host copy queues on the copy engine behind whatever a read in progress has put there).
static int read_ingest_words(gd_ctx* c, hipStream_t st, uint32_t (&w)[3])
{
    hipLaunchKernelGGL(gd::gd_copy_words_kernel, dim3(1), dim3(64), 0, st, c->d_ingest, c->h_ingest, 4u);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(st));
    w[0] = c->h_ingest[0]; w[1] = c->h_ingest[1]; w[2] = c->h_ingest[2];
    return GD_OK;
}

// GD_OPT_COMMIT_CHECK = 1: what the index pass found in the blocks committed since the last look (h_ingest[3], just read
// on a stream that is behind the copy stream).  A failure stays on the device word: the records are part of the contig,
// gd_compute keeps refusing until gd_reset.
static int commit_verdict(gd_ctx* c)
{
    c->commit_checks_pending = false;
    const uint32_t bad = c->h_ingest[3];
    if (!bad) return GD_OK;
    c->commit_checks_pending = true;
    const int lo = c->commit_tid_lo, hi = c->commit_tid_hi;
    if (bad & 4u) return fail(c, GD_E_RANGE, "contigs %d..%d: a committed record has a negative position (a placed BAM record has POS >= 0); gd_reset", lo, hi);
    if (bad & 1u) return fail(c, GD_E_UNSORTED, "contigs %d..%d: committed records are not coordinate sorted; gd_reset", lo, hi);
    return fail(c, GD_E_INVALID, "contigs %d..%d: cigar_off of committed records not monotone; gd_reset", lo, hi);
}

// The spans the index kernel has measured so far become the look-back of the next gd_compute (verified there as ever).
static void take_ingest_span(gd_ctx* c, int32_t span)
{
    c->ingest_span_dirty = false;
    if (span <= 0 || span == c->ingest_span) return;
    c->ingest_span = span;
    if (!c->lookback_pinned && span <= kAutoLongSpan) c->lookback = std::max(64, (span + 63) & ~63);
}

int gd_acquire(gd_ctx* c, size_t reads_cap, size_t ops_cap, gd_batch* out)
{
    if (!c || !out) return GD_E_INVALID;
    if (int r = set_device(c)) return r;
    // the cursor moves here, not at the commit: a producer may hold several blocks (its threads fill block k+1 while
    // block k is validated and committed); a slot that comes round while still held means every slot is out
    RingSlot& s = c->ring[c->ring_next];
    if (s.held)
        return fail(c, GD_E_STATE, "all %d staging blocks are held: gd_commit one (n_reads 0 gives it back unused)", kRingSlots);
    if (s.busy) {
        HIPCHK(c, hipEventSynchronize(s.done));
        s.busy = false;
    }
    if (reads_cap < 1) reads_cap = 1;
    if (ops_cap < 1) ops_cap = 1;
    if (s.b.reads_cap < reads_cap) {
        if (s.b.pos) { (void)hipHostFree(s.b.pos); (void)hipHostFree(s.b.flag);
                       (void)hipHostFree(s.b.mapq); (void)hipHostFree(s.b.cigar_off); }
        s.b.pos = nullptr; s.b.flag = nullptr; s.b.mapq = nullptr; s.b.cigar_off = nullptr;
        s.b.reads_cap = 0;
        HIPCHK(c, hipHostMalloc(reinterpret_cast<void**>(&s.b.pos), reads_cap * sizeof(int32_t), hipHostMallocDefault));
        HIPCHK(c, hipHostMalloc(reinterpret_cast<void**>(&s.b.flag), reads_cap * sizeof(uint16_t), hipHostMallocDefault));
        HIPCHK(c, hipHostMalloc(reinterpret_cast<void**>(&s.b.mapq), reads_cap * sizeof(uint8_t), hipHostMallocDefault));
        HIPCHK(c, hipHostMalloc(reinterpret_cast<void**>(&s.b.cigar_off), (reads_cap + 1) * sizeof(uint32_t), hipHostMallocDefault));
        s.b.reads_cap = reads_cap;
    }
    if (s.b.ops_cap < ops_cap) {
        if (s.b.cigar) (void)hipHostFree(s.b.cigar);
        s.b.cigar = nullptr;
        s.b.ops_cap = 0;
        HIPCHK(c, hipHostMalloc(reinterpret_cast<void**>(&s.b.cigar), ops_cap * sizeof(uint32_t), hipHostMallocDefault));
        s.b.ops_cap = ops_cap;
    }
    s.b.slot = c->ring_next;
    s.held = true;
    c->ring_next = (c->ring_next + 1) % kRingSlots;
    *out = s.b;
    return GD_OK;
}

static int commit_block(gd_ctx* c, const gd_batch* b, int32_t tid, size_t n_reads, size_t n_ops, bool validated);

int gd_commit(gd_ctx* c, const gd_batch* b, int32_t tid, size_t n_reads, size_t n_ops)
{
    return commit_block(c, b, tid, n_reads, n_ops, false);
}

// Makes room for n_reads / n_ops more records of a contig in ONE step (a producer that knows its totals: gd_push;
// growing geometrically block by block drains the copy pipeline at every step).
static int reserve_records(gd_ctx* c, ContigHost& h, size_t n_reads, size_t n_ops)
{
    const size_t need_r = h.n_reads + n_reads, need_o = h.n_ops + n_ops;
    if (need_r > h.cap_reads) {
        size_t c1 = h.cap_reads, c2 = h.cap_reads, c3 = h.cap_reads, c4 = h.cap_reads ? h.cap_reads + 1 : 0;
        if (int r = ensure_dev(c, &h.pos, &c1, need_r, true, h.n_reads)) return r;
        if (int r = ensure_dev(c, &h.flag, &c2, need_r, true, h.n_reads)) return r;
        if (int r = ensure_dev(c, &h.mapq, &c3, need_r, true, h.n_reads)) return r;
        if (int r = ensure_dev(c, &h.off, &c4, need_r + 1, true, h.n_reads ? h.n_reads + 1 : 0)) return r;
        h.cap_reads = need_r;
    }
    if (need_o > h.cap_ops) {
        size_t co = h.cap_ops;
        if (int r = ensure_dev(c, &h.cigar, &co, need_o, true, h.n_ops)) return r;
        h.cap_ops = need_o;
    }
    return GD_OK;
}

static int commit_block(gd_ctx* c, const gd_batch* b, int32_t tid, size_t n_reads, size_t n_ops, bool validated)
{
    if (c && c->cs.pending) return fail(c, GD_E_STATE, "a compute is in flight: gd_compute_finish first");
    if (!c || !b) return GD_E_INVALID;
    if (int r = set_device(c)) return r;
    if (b->slot < 0 || b->slot >= kRingSlots || c->ring[b->slot].b.pos != b->pos)
        return fail(c, GD_E_INVALID, "batch was not obtained from gd_acquire");
    RingSlot& s = c->ring[b->slot];
    if (!s.held) return fail(c, GD_E_STATE, "batch was committed already");
    s.held = false;                                      // (whatever happens below, the block goes back to the ring)
    if (tid < 0 || (size_t)tid >= c->contigs.size()) return fail(c, GD_E_RANGE, "tid %d out of range", tid);
    if (n_reads > b->reads_cap || n_ops > b->ops_cap) return fail(c, GD_E_INVALID, "batch overflow");
    if (n_reads == 0) return GD_OK;
    ContigHost& h = c->contigs[tid];
    if (h.adopted) return fail(c, GD_E_STATE, "contig %d holds adopted device records", tid);
    if (b->cigar_off[0] != 0 || b->cigar_off[n_reads] != n_ops)
        return fail(c, GD_E_INVALID, "cigar_off must start at 0 and end at n_ops");
    if ((uint64_t)h.n_ops + n_ops > 0xffffffffull)
        return fail(c, GD_E_RANGE, "more than 2^32 CIGAR ops on contig %d", tid);
    if ((uint64_t)h.n_reads + n_reads >= kMaxReadsPerContig)
        return fail(c, GD_E_RANGE, "more than 2^30 records on contig %d", tid);
    // coordinate order (BAM SO:coordinate) is what makes the tile search valid.  The common case -- nothing
    // wrong -- is three branch-free passes the compiler vectorises (a 12.6 M-record chromosome: ~2 ms instead of
    // ~13 ms for the record-by-record loop, which was a quarter of the whole host-to-results path); only a block
    // that fails them is walked again to say where.
    int32_t last = h.last_pos;
    const bool on_device = !validated && c->commit_check_device && c->h2d_kernel && n_reads >= 4096;
    if (validated) last = b->pos[n_reads - 1];           // (gd_push: its filler threads checked the block while copying)
    else if (on_device) {
        // the seam with what is there already is looked at here; the rest by the index pass once the block has landed
        if (b->pos[0] < 0) return fail(c, GD_E_RANGE, "contig %d record %zu: negative position %d (a placed BAM record has POS >= 0)", tid, (size_t)h.n_reads, b->pos[0]);
        if (b->pos[0] < last) return fail(c, GD_E_UNSORTED, "contig %d record %zu: pos %d < %d", tid, (size_t)h.n_reads, b->pos[0], last);
        last = b->pos[n_reads - 1];
    } else {
        const int32_t* __restrict__ const p = b->pos;
        const uint32_t* __restrict__ const o = b->cigar_off;
        // records [a, e): positions non-decreasing (from the record before), offsets non-decreasing
        auto check = [p, o](size_t a, size_t e, int32_t before) -> uint32_t {
            uint32_t bad = (uint32_t)(p[a] < before);
            for (size_t i = a + 1; i < e; ++i) bad |= (uint32_t)(p[i] < p[i - 1]);
            for (size_t i = a; i < e; ++i) bad |= (uint32_t)(o[i + 1] < o[i]);
            return bad;
        };
        uint32_t bad = (uint32_t)(p[0] < 0);                 // sorted: p[0] is the smallest
        FillPool* const pool = n_reads >= (1u << 18) ? ctx_pool(c) : nullptr;
        if (pool) {                                          // a large block: the context's worker threads, 64 k records each
            std::vector<FillPool::Item> work;
            const size_t piece = 1u << 16;
            for (size_t a = 0; a < n_reads; a += piece) {
                const size_t e = std::min(n_reads, a + piece);
                work.push_back({nullptr, p + a, (e - a) * 4, 0u, 3, a ? p[a - 1] : last});
                work.push_back({nullptr, o + a, (e - a + 1) * 4, 0u, 4, 0});
            }
            pool->bad.store(0);
            pool->run(std::move(work));
            bad |= pool->bad.load();
        } else {
            bad |= check(0, n_reads, last);
        }
        if (bad) {
            for (size_t i = 0; i < n_reads; ++i) {
                if (p[i] < 0) return fail(c, GD_E_RANGE, "contig %d record %zu: negative position %d (a placed BAM record has POS >= 0)", tid, h.n_reads + i, p[i]);
                if (p[i] < last) return fail(c, GD_E_UNSORTED, "contig %d record %zu: pos %d < %d", tid, h.n_reads + i, p[i], last);
                if (o[i + 1] < o[i]) return fail(c, GD_E_INVALID, "cigar_off not monotone");
                last = p[i];
            }
        }
        last = p[n_reads - 1];
    }
    // the CSR offsets are rebased to the contig stream: on the way by the copy kernel, else here
    const uint32_t base = (uint32_t)h.n_ops;
    const bool blit = c->h2d_kernel && n_reads >= 4096;
    if (base && !blit) {
        uint32_t* __restrict__ const o = b->cigar_off;
        for (size_t i = 0; i <= n_reads; ++i) o[i] += base;
    }
    size_t cr = h.cap_reads, cr1 = h.cap_reads ? h.cap_reads + 1 : 0, co = h.cap_ops;
    size_t need_r = h.n_reads + n_reads;
    if (need_r > h.cap_reads) {
        size_t ncap = std::max(need_r, h.cap_reads * 2);
        size_t c1 = cr, c2 = cr, c3 = cr;
        if (int r = ensure_dev(c, &h.pos, &c1, ncap, true, h.n_reads)) return r;
        if (int r = ensure_dev(c, &h.flag, &c2, ncap, true, h.n_reads)) return r;
        if (int r = ensure_dev(c, &h.mapq, &c3, ncap, true, h.n_reads)) return r;
        if (int r = ensure_dev(c, &h.off, &cr1, ncap + 1, true, h.n_reads ? h.n_reads + 1 : 0)) return r;
        h.cap_reads = ncap;
    }
    if (h.n_ops + n_ops > h.cap_ops) {
        size_t ncap = std::max(h.n_ops + n_ops, h.cap_ops * 2);
        if (int r = ensure_dev(c, &h.cigar, &co, ncap, true, h.n_ops)) return r;
        h.cap_ops = ncap;
    }
    hipStream_t cs = c->copy_stream;
    if (blit) {
        // one launch: workgroups read the page-locked block over the link (gd_stage.hpp)
        gd::H2DJob j{};
        j.off_add = base;
        j.seg[0] = {h.pos + h.n_reads, b->pos, n_reads * sizeof(int32_t)};
        j.seg[1] = {h.off + h.n_reads, b->cigar_off, (n_reads + 1) * sizeof(uint32_t)};
        j.seg[2] = {h.cigar + h.n_ops, b->cigar, n_ops * sizeof(uint32_t)};
        j.seg[3] = {h.flag + h.n_reads, b->flag, n_reads * sizeof(uint16_t)};
        j.seg[4] = {h.mapq + h.n_reads, b->mapq, n_reads * sizeof(uint8_t)};
        hipLaunchKernelGGL(gd::gd_h2d_kernel, dim3(c->h2d_grid), dim3(256), 0, cs, j);
        HIPCHK(c, hipGetLastError());
    } else {
        HIPCHK(c, hipMemcpyAsync(h.pos + h.n_reads, b->pos, n_reads * sizeof(int32_t), hipMemcpyHostToDevice, cs));
        HIPCHK(c, hipMemcpyAsync(h.flag + h.n_reads, b->flag, n_reads * sizeof(uint16_t), hipMemcpyHostToDevice, cs));
        HIPCHK(c, hipMemcpyAsync(h.mapq + h.n_reads, b->mapq, n_reads * sizeof(uint8_t), hipMemcpyHostToDevice, cs));
        HIPCHK(c, hipMemcpyAsync(h.off + h.n_reads, b->cigar_off, (n_reads + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, cs));
        if (n_ops)
            HIPCHK(c, hipMemcpyAsync(h.cigar + h.n_ops, b->cigar, n_ops * sizeof(uint32_t), hipMemcpyHostToDevice, cs));
    }
    {
        // the block is part of the contig's stream now (in copy-stream order): index it, measure its spans
        const size_t r0 = h.n_reads;
        const int32_t before = h.last_pos;
        h.n_ops += n_ops;                                  // (index_records reads the contig's totals)
        const int ri = index_records(c, h, r0, r0 + n_reads, before, on_device, cs, true);
        h.n_ops -= n_ops;
        if (ri) return ri;
    }
    HIPCHK(c, hipEventRecord(s.done, cs));
    s.busy = true;
    if (h.ck_ok) {                          // the long-read structures no longer cover the stream
        HIPCHK(c, hipStreamSynchronize(c->stream));
        drop_norm(h);
    }
    h.n_reads += n_reads;
    h.n_ops += n_ops;
    h.last_pos = last;
    c->computed = false;
    if (on_device) {
        if (!c->commit_checks_pending) { c->commit_tid_lo = c->commit_tid_hi = tid; c->commit_checks_pending = true; }
        c->commit_tid_lo = std::min(c->commit_tid_lo, tid);
        c->commit_tid_hi = std::max(c->commit_tid_hi, tid);
    }
    return GD_OK;
}

int gd_check_commits(gd_ctx* c)
{
    if (!c) return GD_E_INVALID;
    if (c->cs.pending) return fail(c, GD_E_STATE, "a compute is in flight: gd_compute_finish first");
    if (!c->commit_checks_pending) return GD_OK;
    if (int r = set_device(c)) return r;
    uint32_t w[3];
    if (int r = read_ingest_words(c, c->copy_stream, w)) return r;
    return commit_verdict(c);
}

int gd_push(gd_ctx* c, int32_t tid, const int32_t* pos, const uint16_t* flag, const uint8_t* mapq,
            const uint32_t* cigar_off, const uint32_t* cigar, size_t n_reads, size_t n_ops)
{
    if (!c) return GD_E_INVALID;
    if (n_reads == 0) return GD_OK;
    if (!pos || !flag || !mapq || !cigar_off || (n_ops && !cigar)) return GD_E_INVALID;
    if (tid < 0 || (size_t)tid >= c->contigs.size()) return fail(c, GD_E_RANGE, "tid %d out of range", tid);
    if (c->cs.pending) return fail(c, GD_E_STATE, "a compute is in flight: gd_compute_finish first");
    if (int r = set_device(c)) return r;
    {
        ContigHost& h = c->contigs[tid];
        if (h.adopted) return fail(c, GD_E_STATE, "contig %d holds adopted device records", tid);
        if (cigar_off[n_reads] > n_ops) return fail(c, GD_E_INVALID, "cigar_off out of range");
        if (int r = reserve_records(c, h, n_reads, cigar_off[n_reads] - cigar_off[0])) return r;   // one allocation, not one per doubling
    }
    const size_t chunk = c->push_chunk;   // records per staging block (kRingSlots blocks: one being filled, the others on the link)
    FillPool local;                                     // (no workers: small pushes run on the calling thread)
    FillPool* const shared = n_reads >= (1u << 18) ? ctx_pool(c) : nullptr;
    FillPool& pool = shared ? *shared : local;
    const int workers = shared ? 1 : 0;
    const size_t piece = 1u << 20;   // bytes per work item
    size_t i = 0;
    while (i < n_reads) {
        size_t n = std::min(chunk, n_reads - i);
        size_t o0 = cigar_off[i], o1 = cigar_off[i + n];
        if (o1 < o0 || o1 > n_ops) return fail(c, GD_E_INVALID, "cigar_off out of range");
        gd_batch b;
        if (int r = gd_acquire(c, n, o1 - o0, &b)) return r;
        std::vector<FillPool::Item> work;
        pool.bad.store(0);
        const int32_t before = i ? pos[i - 1] : c->contigs[tid].last_pos;
        auto add = [&](void* dst, const void* src, size_t bytes, uint32_t sub, int kind) {
            for (size_t at = 0; at < bytes; at += piece) {
                FillPool::Item it{static_cast<char*>(dst) + at, static_cast<const char*>(src) + at, std::min(piece, bytes - at), sub, kind, 0};
                if (kind == 1) it.prev = at ? reinterpret_cast<const int32_t*>(static_cast<const char*>(src) + at)[-1] : before;
                if (kind == 2) it.prev = at ? reinterpret_cast<const int32_t*>(static_cast<const char*>(src) + at)[-1] : (int32_t)sub;
                work.push_back(it);
            }
        };
        add(b.pos, pos + i, n * sizeof(int32_t), 0, 1);                               // copied and checked: sorted, not negative
        add(b.flag, flag + i, n * sizeof(uint16_t), 0, 0);
        add(b.mapq, mapq + i, n * sizeof(uint8_t), 0, 0);
        add(b.cigar_off, cigar_off + i, (n + 1) * sizeof(uint32_t), (uint32_t)o0, 2);   // block relative; checked: non-decreasing
        if (o1 > o0) add(b.cigar, cigar + o0, (o1 - o0) * sizeof(uint32_t), 0, 0);
        if (workers > 0) pool.run(std::move(work));
        else { for (const auto& it : work) pool.run_item(it); }
        // a block that failed a check goes through gd_commit's own validation, which says where
        if (int r = commit_block(c, &b, tid, n, o1 - o0, pool.bad.load() == 0)) return r;
        i += n;
    }
    return GD_OK;
}

int gd_reserve(gd_ctx* c, int32_t tid, size_t n_reads, size_t n_ops)
{
    if (!c) return GD_E_INVALID;
    if (c->cs.pending) return fail(c, GD_E_STATE, "a compute is in flight: gd_compute_finish first");
    if (int r = set_device(c)) return r;
    if (tid < 0 || (size_t)tid >= c->contigs.size()) return fail(c, GD_E_RANGE, "tid %d out of range", tid);
    ContigHost& h = c->contigs[tid];
    if (h.adopted) return fail(c, GD_E_STATE, "contig %d holds adopted device records", tid);
    if ((uint64_t)h.n_ops + n_ops > 0xffffffffull) return fail(c, GD_E_RANGE, "more than 2^32 CIGAR ops on contig %d", tid);
    if ((uint64_t)h.n_reads + n_reads >= kMaxReadsPerContig) return fail(c, GD_E_RANGE, "more than 2^30 records on contig %d", tid);
    return reserve_records(c, h, n_reads, n_ops);
}

int gd_adopt_device(gd_ctx* c, int32_t tid, const gd_batch* d, size_t n_reads, size_t n_ops)
{
    if (c && c->cs.pending) return fail(c, GD_E_STATE, "a compute is in flight: gd_compute_finish first");
    if (!c || !d) return GD_E_INVALID;
    if (int r = set_device(c)) return r;
    if (tid < 0 || (size_t)tid >= c->contigs.size()) return fail(c, GD_E_RANGE, "tid %d out of range", tid);
    if (n_reads && (!d->pos || !d->flag || !d->mapq || !d->cigar_off)) return GD_E_INVALID;
    if (n_ops > 0xffffffffull) return fail(c, GD_E_RANGE, "more than 2^32 CIGAR ops");
    // depth <= records of the contig; the window reduction adds four depths in 32 bits
    if (n_reads >= kMaxReadsPerContig) return fail(c, GD_E_RANGE, "more than 2^30 records on one contig");
    // The arrays are checked and indexed right away, on this context's stream: whatever
    // stream of the caller produced them must have finished.  A device-wide wait makes that true for
    // any producer (a few microseconds per contig, at ingest time).
    HIPCHK(c, hipDeviceSynchronize());
    ContigHost& h = c->contigs[tid];
    ContigHost t;                                          // the new stream, checked before the old one is let go
    t.length = h.length;
    t.pos = d->pos; t.flag = d->flag; t.mapq = d->mapq; t.off = d->cigar_off; t.cigar = d->cigar;
    t.n_reads = n_reads; t.n_ops = n_ops;
    t.adopted = true;
    if (n_reads) {
        // what gd_commit checks on a host block, here in one pass over pos / cigar_off on the device -- the same pass
        // leaves the position index and the largest span (gd_index_records_kernel)
        uint32_t w[3] = {0, 0, 0};
        int r = GD_OK;
        if (hipMemsetAsync(c->d_ingest, 0, sizeof(uint32_t), c->stream) != hipSuccess) r = fail(c, GD_E_HIP, "hipMemsetAsync failed");
        if (r == GD_OK) r = index_records(c, t, 0, n_reads, -1, true, c->stream);
        if (r == GD_OK) r = read_ingest_words(c, c->stream, w);
        if (r == GD_OK) {
            const uint32_t bad = w[0];
            if (bad & 4u) r = fail(c, GD_E_RANGE, "contig %d: a device record has a negative position (a placed BAM record has POS >= 0)", tid);
            else if (bad & 1u) r = fail(c, GD_E_UNSORTED, "contig %d: device records not coordinate sorted", tid);
            else if (bad & 2u) r = fail(c, GD_E_INVALID, "contig %d: CSR offsets of the device records are not a non-decreasing sequence from 0 to at most %zu", tid, n_ops);
        }
        if (r != GD_OK) { free_contig(t); return r; }      // (the contig keeps what it held)
        t.last_pos = (int32_t)w[2];
        take_ingest_span(c, (int32_t)w[1]);
    }
    free_contig(h);
    h = t;
    c->computed = false;
    return GD_OK;
}

int gd_reset(gd_ctx* c)
{
    if (c && c->cs.pending) return fail(c, GD_E_STATE, "a compute is in flight: gd_compute_finish first");
    if (!c) return GD_E_INVALID;
    if (int r = set_device(c)) return r;
    (void)gd_ingest_abort(c);                              // a device BAM read in progress (its reader thread) ends here
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipStreamSynchronize(c->copy_stream));
    for (auto& h : c->contigs) {
        int64_t len = h.length;
        free_contig(h);
        h.length = len;
    }
    c->bounds.clear();
    for (auto& s : c->ring) s.held = false;                // (blocks handed out before the reset are not part of anything)
    c->commit_checks_pending = false;                      // (... and their verdicts go with the records: d_ingest is cleared below)
    c->computed = false;
    c->lookback = c->params.max_span_hint > 0 ? c->params.max_span_hint : kDefaultLookback;
    c->span_forces_long = false;
    HIPCHK(c, hipMemsetAsync(c->d_ingest, 0, 4 * sizeof(uint32_t), c->stream));   // spans of records that are gone
    HIPCHK(c, hipStreamSynchronize(c->stream));            // (the next block's index pass runs on the copy stream: not before this)
    c->ingest_span = 0;
    c->ingest_span_dirty = false;
    return GD_OK;
}

#include "gd_api_compute.inc"
#include "gd_api_results.inc"
#include "gd_api_aux.inc"
#include "gd_api_ingest.inc"
#include "gd_api_comm.inc"

int gd_device_perbase(gd_ctx* c, int32_t tid, const int32_t** dptr, int64_t* len)
{
    if (!c || !dptr) return GD_E_INVALID;
    if (int r = check_result_tid(c, tid)) return r;
    const ContigHost& h = c->contigs[tid];
    if (!c->d_perbase) return fail(c, GD_E_STATE, "the per-base vector was not kept (gd_set_outputs)");
    *dptr = h.length > 0 ? c->d_perbase + h.base_off : nullptr;
    if (len) *len = h.length;
    return GD_OK;
}

int gd_device_windows(gd_ctx* c, const int64_t** d_sums, const int32_t** d_mins, size_t* n_total)
{
    if (!c) return GD_E_INVALID;
    if (!c->computed) return fail(c, GD_E_STATE, "no results: call gd_compute first");
    if (d_sums) *d_sums = c->d_wsum;
    if (d_mins) *d_mins = c->d_wmin;
    if (n_total) *n_total = (size_t)c->n_win_total;
    return GD_OK;
}

int gd_window_offset(gd_ctx* c, int32_t tid, size_t* off, size_t* n)
{
    if (!c) return GD_E_INVALID;
    if (int r = check_result_tid(c, tid)) return r;
    const ContigHost& h = c->contigs[tid];
    if (off) *off = h.win_off < 0 ? 0 : (size_t)h.win_off;
    if (n) *n = (size_t)h.n_win;
    return GD_OK;
}

int gd_device_runs(gd_ctx* c, const int32_t** d_bounds, size_t* n_bounds)
{
    if (!c) return GD_E_INVALID;
    if (!c->computed) return fail(c, GD_E_STATE, "no results: call gd_compute first");
    if (d_bounds) *d_bounds = reinterpret_cast<const int32_t*>(c->d_ordered);
    if (n_bounds) *n_bounds = c->bounds.size();
    return GD_OK;
}

int gd_set_export(gd_ctx* c, void* device_buf, int64_t max_windows, int64_t cap_bounds)
{
    if (!c || max_windows < 0 || cap_bounds < 0) return GD_E_INVALID;
    c->export_buf = static_cast<int64_t*>(device_buf);
    c->export_max_w = max_windows;
    c->export_cap_b = cap_bounds;
    return GD_OK;
}

int gd_wait_event(gd_ctx* c, void* ev)
{
    if (!c || !ev) return GD_E_INVALID;
    if (int r = set_device(c)) return r;
    HIPCHK(c, hipStreamWaitEvent(c->stream, static_cast<hipEvent_t>(ev), 0));
    return GD_OK;
}

int gd_set_option(gd_ctx* c, int option, int64_t value)
{
    if (c && c->cs.pending) return fail(c, GD_E_STATE, "a compute is in flight: gd_compute_finish first");
    if (!c) return GD_E_INVALID;
#ifndef GD_MEASURE
    // The switches of measurement builds (-DGD_MEASURE; include/goleft_depth.h lists them apart): every one of them was measured
    // neutral or worse (DESIGN.md / HISTORY.md), a release library holds them at their defaults and says so when asked for
    // anything else.
    {
        bool other = false;
        switch (option) {
        case GD_OPT_INGEST_DMA: other = value > 1; break;
        case GD_OPT_INGEST_PIECE_STREAMS: other = value != 1; break;
        case GD_OPT_INGEST_HYBRID: case GD_OPT_INGEST_WALK_CUS: case GD_OPT_INFLATE_LDS_PAD: other = value != 0; break;
        case GD_OPT_INGEST_BATCHES: other = value != 8; break;
        default: break;
        }
        if (other) return fail(c, GD_E_INVALID, "option %d = %lld is a switch of measurement builds (-DGD_MEASURE); this library keeps it at its default", option, (long long)value);
    }
#endif
    switch (option) {
    case GD_OPT_NT_STORES: c->tile_opt = value ? 1 : 0; break;
    case GD_OPT_FAST_KERNEL: c->fast_kernel = value != 0; break;
    case GD_OPT_COPY_THREADS:
        if (value < 1 || value > 16) return fail(c, GD_E_INVALID, "copy threads: 1..16");
        c->ing_copy_threads = (int)value;
        break;
    case GD_OPT_H2D_KERNEL:
        if (value < 0 || value > 4096) return fail(c, GD_E_INVALID, "h2d kernel: 0 (hipMemcpyAsync) or its grid, 1..4096 workgroups");
        c->h2d_kernel = value != 0;
        if (value > 1) c->h2d_grid = (unsigned)value;
        break;
    case GD_OPT_INGEST_CRC: c->ingest_crc = value != 0; break;
    case GD_OPT_INGEST_DMA:
        if (value < 0 || value > 4) return fail(c, GD_E_INVALID, "ingest DMA streams: 1 .. 4 (0: a copy kernel on a high-priority stream)");
        c->ing_dma_n = (int)value;
        break;
    case GD_OPT_BAM_REFS:
        if (value < 0 || value > 0x7fffffff) return fail(c, GD_E_INVALID, "BAM references: 0 (unknown) .. 2^31 - 1");
        c->bam_n_ref = (int32_t)value;
        break;
    case GD_OPT_PUSH_CHUNK:
        if (value < 4096 || value > (1 << 24)) return fail(c, GD_E_INVALID, "push chunk: 4096 .. 2^24 records");
        c->push_chunk = (size_t)value;
        break;
    case GD_OPT_PUSH_THREADS:
        if (value < 1 || value > 64) return fail(c, GD_E_INVALID, "push threads: 1..64");
        c->push_threads = (int)value;
        break;
    case GD_OPT_INGEST_INDEX: c->ingest_index = value != 0; break;
    case GD_OPT_INGEST_COPY_GRID:
        if (value < 1 || value > 4096) return fail(c, GD_E_INVALID, "ingest copy grid: 1 .. 4096 workgroups");
        c->ing_copy_grid = (unsigned)value;
        break;
    case GD_OPT_COMMIT_CHECK:
        if (value != 0 && value != 1) return fail(c, GD_E_INVALID, "commit check: 0 (host, in gd_commit) or 1 (device, deferred)");
        c->commit_check_device = value == 1;
        break;
    case GD_OPT_INGEST_HYBRID: c->ing_hybrid = value != 0; break;
    case GD_OPT_INGEST_CU_SPLIT:
        if (value < 0 || value > 64 || value == 1) return fail(c, GD_E_INVALID, "ingest CU split: 0 (off) or 2 .. 64");
        if (c->ing_stream[0] || c->ing_hp) return fail(c, GD_E_STATE, "the ingest streams exist already: set GD_OPT_INGEST_CU_SPLIT before the first gd_ingest_begin");
        c->ing_cu_split = (int)value;
        break;
    case GD_OPT_INGEST_RANGE_HINT:
        if (value < 0) return fail(c, GD_E_INVALID, "ingest range hint: bytes >= 0");
        c->ing_range_hint = (uint64_t)value;
        break;
    case GD_OPT_INFLATE_LDS_PAD:
        if (value < 0 || value > 120 * 1024) return fail(c, GD_E_INVALID, "inflate LDS pad: 0 .. 122880 bytes");
        c->inflate_pad = (unsigned)value;
        break;
    case GD_OPT_INGEST_BATCHES:
        if (value < 1 || value > 64) return fail(c, GD_E_INVALID, "ingest batches: 1 .. 64 inflate launches per range");
        c->ing_batches = (int)value;
        break;
    case GD_OPT_INGEST_WALK_CUS:
        if (c->ing_walk) return fail(c, GD_E_STATE, "the ingest streams exist already: set GD_OPT_INGEST_WALK_CUS before the first gd_ingest_begin");
        c->ing_walk_cus = value != 0;
        break;
    case GD_OPT_INFLATE_KERNEL:
        if (value < 0 || value > 1) return fail(c, GD_E_INVALID, "inflate kernel: 0 (a lane per member) or 1 (a workgroup per member)");
        c->inflate_kernel = (int)value;
        break;
    case GD_OPT_INGEST_PIECE_STREAMS:
        if (value < 1 || value > 4) return fail(c, GD_E_INVALID, "ingest piece streams: 1 .. 4");
        if (c->ing_n) return fail(c, GD_E_STATE, "a device BAM read is pending");
        c->ing_piece_streams = (int)value;
        break;
    default: return fail(c, GD_E_INVALID, "unknown option %d", option);
    }
    c->computed = false;
    return GD_OK;
}

int gd_get_option(gd_ctx* c, int option, int64_t* value)
{
    if (!c || !value) return GD_E_INVALID;
    switch (option) {
    case GD_OPT_NT_STORES: *value = c->tile_opt & 1; break;
    case GD_OPT_FAST_KERNEL: *value = c->fast_kernel; break;
    case GD_OPT_COPY_THREADS: *value = c->ing_copy_threads; break;
    case GD_OPT_H2D_KERNEL: *value = c->h2d_kernel ? (int64_t)std::max(1u, c->h2d_grid) : 0; break;
    case GD_OPT_INGEST_CRC: *value = c->ingest_crc; break;
    case GD_OPT_INGEST_DMA: *value = c->ing_dma_n; break;
    case GD_OPT_BAM_REFS: *value = c->bam_n_ref; break;
    case GD_OPT_PUSH_CHUNK: *value = (int64_t)c->push_chunk; break;
    case GD_OPT_PUSH_THREADS: *value = c->push_threads; break;
    case GD_OPT_INGEST_INDEX: *value = c->ingest_index; break;
    case GD_OPT_COMMIT_CHECK: *value = c->commit_check_device; break;
    case GD_OPT_INGEST_COPY_GRID: *value = c->ing_copy_grid; break;
    case GD_OPT_INGEST_HYBRID: *value = c->ing_hybrid; break;
    case GD_OPT_INGEST_CU_SPLIT: *value = c->ing_cu_split; break;
    case GD_OPT_INGEST_RANGE_HINT: *value = (int64_t)c->ing_range_hint; break;
    case GD_OPT_INFLATE_LDS_PAD: *value = c->inflate_pad; break;
    case GD_OPT_INGEST_BATCHES: *value = c->ing_batches; break;
    case GD_OPT_INGEST_WALK_CUS: *value = c->ing_walk_cus; break;
    case GD_OPT_INFLATE_KERNEL: *value = c->inflate_kernel; break;
    case GD_OPT_INGEST_PIECE_STREAMS: *value = c->ing_piece_streams; break;
    default: return fail(c, GD_E_INVALID, "unknown option %d", option);
    }
    return GD_OK;
}

int gd_get_stats(gd_ctx* c, gd_stats* out)
{
    if (!c || !out) return GD_E_INVALID;
    *out = c->stats;
    return GD_OK;
}

int gd_compute_timing(gd_ctx* c, double* seconds, int n)
{
    if (!c || !seconds || n < 0) return GD_E_INVALID;
    for (int i = 0; i < n; ++i) seconds[i] = i < 4 ? c->timing[i] : 0.0;
    return GD_OK;
}

int gd_set_profiling(gd_ctx* c, int on)
{
    if (!c) return GD_E_INVALID;
    c->profiling = on != 0;
    c->kernel_ms[GD_K_NORM] = 0;                        // accumulate over the contigs normalised / checkpointed
    c->kernel_ms[GD_K_CKPT] = 0;                        // from now on
    return GD_OK;
}

int gd_kernel_ms(gd_ctx* c, int id, float* ms)
{
    if (!c || !ms || id < 0 || id >= GD_K_COUNT) return GD_E_INVALID;
    *ms = c->kernel_ms[id];
    return GD_OK;
}

}  // extern "C"
