// gd_api_state.hpp -- what a gd_ctx holds and the helpers every part of the C ABI shares: per-contig
// record streams, the pinned staging ring, the device job state, launch helpers of the tile kernels,
// the long-read structures (gd_chunk.hpp) and the state of a device BAM read.  Included by gd_api.hip
// only (one translation unit: the kernels of gd_kernels.hpp are not inline).
#pragma once

#include <memory>
#include <atomic>
#include <mutex>
#include <thread>

namespace {

constexpr int kRingSlots = 4;
constexpr size_t kSpecBounds = 4096;     // boundaries copied to the host before their count is known (one synchronisation)
constexpr int kDefaultLookback = 512;
constexpr uint64_t kMaxReadsPerContig = 1ull << 30;
constexpr int kTileT = 4096;              // reference positions per tile, 256 threads each: the one shape that is built (8192
                                           // positions and 512 threads were measured slower on every workload and retired)
constexpr int kAutoLongSpan = 32768;      // GD_PATH_AUTO leaves the short-read tile path above this read span
constexpr int kMaxSpan = 1 << 27;   // tile-relative byte offsets of the tile kernel stay in 32 bits

// One device allocation shared by the derived arrays of a batch of contigs (long-read structures, position
// indexes, long-read structures): a batch is built with one hipMalloc and no per-contig host round trip; the
// block goes back to HBM when the last contig that points into it drops its share.
struct DevBlock {
    void* p = nullptr;
    size_t bytes = 0;
    ~DevBlock() { if (p) (void)hipFree(p); }
};
typedef std::shared_ptr<DevBlock> BlockRef;

struct ContigHost {
    int64_t length = 0;
    // device record stream (owned unless adopted)
    int32_t*  pos = nullptr;
    uint16_t* flag = nullptr;
    uint8_t*  mapq = nullptr;
    uint32_t* off = nullptr;
    uint32_t* cigar = nullptr;
    size_t n_reads = 0, n_ops = 0;
    size_t cap_reads = 0, cap_ops = 0;
    bool adopted = false;
    int32_t last_pos = -0x7fffffff;
    // long-read path (gd_chunk.hpp): deletion lists, read records and tile indexes, built straight from the records
    // (slices of ck_blk)
    BlockRef ck_blk;
    uint4*    lrec = nullptr;          // n_reads + 2 long-read records {pos, end, list offset, deletions}; [n_reads + 1].x = the largest span
    uint32_t* lfq = nullptr;           // n_reads: flag << 8 | MAPQ
    uint2*    dl = nullptr;            // (ops >> 1) + n_reads + 1 deletions {start, length}
    uint32_t* pck = nullptr;           // tile indexes (filled by gd_dels_raw_kernel)
    uint32_t* ndel = nullptr;          // deletions per read
    int32_t   max_span = 0;
    uint64_t  n_dels = 0;              // deletions in dl
    bool ck_ok = false;                // lrec / lfq / dl / pck describe the current records
    // built as the records arrive (gd_index_records_kernel): the position index, ridx[k] = first read with pos >= 64 k
    uint32_t* ridx = nullptr;          // (length >> 6) + 2 entries; valid up to entry pos[ridx_reads - 1] >> 6
    size_t ridx_reads = 0;             // records it covers (== n_reads: gd_prep_kernel uses it)
    bool ing_left = false;             // device BAM read in parts: a record of another reference has ended this one's records
    // layout in the result arrays of the last compute (-1 = not computed)
    int64_t base_off = -1;
    int64_t win_off = -1;
    int64_t n_win = 0;
    size_t run_beg = 0, run_end = 0;   // slice of ctx->bounds
};

struct RingSlot {
    gd_batch b{};
    hipEvent_t done = nullptr;
    bool busy = false;       // its copy may still be in flight (`done`)
    bool held = false;       // handed out by gd_acquire, not committed yet
};

}  // namespace

// State of a device BAM read between gd_ingest_begin and gd_ingest_finish.
struct IngestState;
namespace { struct FillPool; }

// Device buffers of one pending range (compressed bytes, inflated bytes, member tables): grow-only and
// kept by the context between ranges -- allocating and freeing gigabytes per range cost 0.1-0.2 s.
struct IngestBufs {
    void *in = nullptr, *out = nullptr, *tab = nullptr;
    size_t cap_in = 0, cap_out = 0, cap_tab = 0;
    bool busy = false;
    static bool fit(void** p, size_t* cap, size_t need)
    {
        if (need <= *cap && *p) return true;
        // (a quarter more than asked for when it has to GROW: ranges of about one size -- a reference read in parts --
        // would otherwise free and allocate again for every range a little larger than the last, and a hipFree waits for
        // the whole device, the other range's inflate kernels included)
        const size_t want = *p ? need + need / 4 : (need ? need : 1);
        if (*p) { (void)hipFree(*p); *p = nullptr; *cap = 0; }
        if (hipMalloc(p, want) != hipSuccess) { *p = nullptr; return false; }
        *cap = want;
        return true;
    }
    // The member tables live in PAGE-LOCKED HOST memory that the kernels read (and write: the status words) over the
    // link: a few bytes per member, once.  As device arrays they were five blocking uploads per range -- each of which
    // queued on the copy engine behind the 64 MB pieces of the read in progress (20 ms per gd_ingest_begin measured).
    static bool fit_host(void** p, size_t* cap, size_t need)
    {
        if (need <= *cap && *p) return true;
        if (*p) { (void)hipHostFree(*p); *p = nullptr; *cap = 0; }
        // generously: freeing page-locked memory waits for the device like hipFree does -- ranges of slightly different
        // member counts (a reference read in parts) must not do that between every two of them (measured: 50 ms per
        // gd_ingest_begin, 1.4 s per genome)
        const size_t want = std::max<size_t>(2 * need + 4096, 8u << 20);
        if (hipHostMalloc(p, want, hipHostMallocDefault) != hipSuccess) { *p = nullptr; return false; }
        *cap = want;
        return true;
    }
    void drop()
    {
        if (in) (void)hipFree(in);
        if (out) (void)hipFree(out);
        in = out = nullptr;
        cap_in = cap_out = 0;
    }
    void drop_all()
    {
        drop();
        if (tab) (void)hipHostFree(tab);
        tab = nullptr;
        cap_tab = 0;
    }
};


// What gd_compute carries from its launch to its finish (gd_api_compute.inc).
struct ComputeState {
    std::vector<int32_t> tids;          // contigs of the job
    uint64_t n_reads = 0, n_ops = 0, n_units = 0, n_groups = 0;
    int64_t tile_beg = 0, base_off = 0, win_off = 0, bases = 0;
    bool raw_aligned = true;            // ... arrays aligned for the raw straight-line kernel's vector loads
    int32_t it_kernel = 0;              // GD_TK_*: the kernel of the attempt in flight
    int reruns = 0, used_lookback = 0;
    bool used_scatter = false, used_chunk = false;
    int32_t chunk_span = 0;
    bool it_scatter = false, it_chunk = false;   // the attempt in flight
    size_t spec = 0;                             // boundaries copied speculatively with the counters
    bool pending = false;               // launched, not finished
    bool nothing = false;               // ... a job without tiles
};

struct GdUniqueId { char internal[128]; };   // ncclUniqueId's layout (rccl.h: NCCL_UNIQUE_ID_BYTES = 128)

struct gd_ctx {
    int device = 0;
    hipStream_t stream = nullptr;       // compute stream
    bool own_stream = true;
    hipStream_t copy_stream = nullptr;
    hipEvent_t copy_done = nullptr;
    gd_params params{};
    std::vector<ContigHost> contigs;
    std::vector<int32_t> selected;      // empty = all
    RingSlot ring[kRingSlots];
    int ring_next = 0;
    bool commit_check_device = false;   // GD_OPT_COMMIT_CHECK
    bool commit_checks_pending = false; // blocks whose verdict (d_ingest[3]) nobody has read yet
    int32_t commit_tid_lo = 0, commit_tid_hi = 0;   // ... the contigs they belong to
    std::string err;

    // tuning knobs (gd_set_option; defaults are what the measurements of DESIGN.md section 4 chose)
    bool fast_kernel = true;            // GD_OPT_FAST_KERNEL: the straight-line tile kernel (gd_tile_fast.hpp) for
                                        // ordinary tiles, the generic one for the rest; 0 = generic for every tile
    std::vector<uint8_t> batch_tab, batch_tab_ck;   // host copies of the job tables of the last norm_batch / ck batch
    uint32_t* h_batch = nullptr; size_t cap_h_batch = 0;   // pinned: per-contig totals / status words coming back
    int tile_opt = 1;                   // bit 0: non-temporal per-base stores (2 % faster: the vector is
                                        // never re-read by the kernel)
    bool lookback_pinned = false;       // max_span_hint given: never shrink below it
    int path = GD_PATH_AUTO;            // gd_set_path
    bool keep_perbase = true;           // gd_set_outputs(GD_OUT_PERBASE)
    bool sums_only = false;             // gd_set_outputs(GD_OUT_SUMS_ONLY): window sums, nothing else
    bool ran_sums_only = false;         // what the last gd_compute produced
    bool span_forces_long = false;      // AUTO: the tile path met a read too long for it
    unsigned long long* d_status = nullptr;  size_t cap_status = 0;   // scatter path look-back words
    int lookback = kDefaultLookback;
    // gd_index_records_kernel's words: [0] check flags, [1] the largest reference span of any record that has arrived
    // since gd_reset / gd_set_contigs (INT_MAX: a record it did not walk), [2] the last position of the last launch
    uint32_t* d_ingest = nullptr;
    bool ingest_span_dirty = false;     // [1] may have grown since the host last read it
    int32_t ingest_span = 0;            // the host's copy (0: nothing measured)
    bool ingest_index = true;           // GD_OPT_INGEST_INDEX
    double alloc_in_enqueue = 0;        // seconds of device allocation inside the last enqueue (the long-read block): counted as `prepare`
    double timing[4] = {0, 0, 0, 0};    // gd_compute_timing: allocations + contig table, enqueue, wait, total of the last gd_compute

    // device job state
    gd::ContigDev* d_ctgs = nullptr;  size_t cap_ctgs = 0;
    std::vector<gd::ContigDev> h_ctgs;
    std::vector<gd::ContigDev> up_ctgs;   // what d_ctgs holds
    gd::ContigDev* up_ctgs_dev = nullptr;
    uint32_t slow_grid = 64;             // workgroups of gd_tile_slow_kernel
    std::vector<int32_t> job_tids;      // contig table index -> tid
    gd::TileInfo* d_tiles = nullptr;  size_t cap_tiles = 0;
    gd::TileFast* d_ftiles = nullptr; size_t cap_ftiles = 0;
    int32_t* d_perbase = nullptr;     size_t cap_perbase = 0;
    int64_t* d_wsum = nullptr;        size_t cap_win = 0;
    int32_t* d_wmin = nullptr;
    int2* d_chunks = nullptr;         size_t cap_runs = 0;
    int2* d_ordered = nullptr;
    uint32_t* d_tile_cnt = nullptr;
    uint32_t* d_tile_off = nullptr;
    uint32_t* d_super_cnt = nullptr;
    gd::Counters* d_counters = nullptr;
    gd::Counters* h_counters = nullptr;   // pinned
    int2* h_bounds = nullptr;             // pinned: the first kSpecBounds ordered boundaries travel with the counters
    uint32_t parity = 0;                  // Counters::n_slow in use (alternates per compute)
    bool slow_counter_clean = false;      // the last enqueue ran gd_prep_kernel, which zeroed the other counter
    uint32_t* d_region_cursor = nullptr;

    int64_t* d_wed = nullptr; size_t cap_wed = 0;      // gd_depthwed: tables + the sites x samples matrix
    uint32_t* d_md_bits = nullptr; size_t cap_md = 0;  // gd_md_flags: `any` words then `suf` words
    uint16_t* d_md_cnt = nullptr; size_t cap_md_cnt = 0;   // gd_md_begin .. gd_md_finish: samples >= min_cov per position
    int64_t md_acc_len = -1;                           // gd_md_begin's length (-1: not begun)
    int64_t md_acc_samples = 0;
    int64_t md_len = -1;                               // positions the bitmaps cover (-1: none yet)
    std::vector<int32_t> md_tids;                      // the samples they were built from
    // gd_ingest_begin .. gd_ingest_finish.  Two ranges may be pending: ing_q[0] is the oldest (the one
    // gd_ingest_decode / _finish / _release act on), the last one is being fed -- so the inflate tail of
    // one range overlaps the upload of the next.
    // ranges that may be pending: one being decoded, one inflating, one being fed (include/goleft_depth.h: a fourth
    // gd_ingest_begin is refused).  Four were measured -- 2.42 against 2.44 s per genome read, profiles/r12u_... -- and not kept.
    static constexpr int kIngestDepth = 3;
    IngestState* ing_q[kIngestDepth] = {};
    int ing_n = 0;
    double ing_secs[7] = {0, 0, 0, 0, 0, 0, 0};         // gd_ingest_timing
    std::thread ing_feeder;                             // gd_ingest_feed_fd: the read of the newest range in progress
    int ing_feeder_rc = 0;
    std::string ing_feeder_err;                         // what the feeder thread's failure said (published by ingest_join)
    double ing_feeder_secs[2] = {0, 0};                 // its share of gd_ingest_timing [0], [1] (merged by ingest_join)
    bool ing_stage_used[8] = {false, false, false, false, false, false, false, false};
    int ing_cur = 0;
    IngestBufs ing_bufs[kIngestDepth];
    // staging of the device BAM read, created by the first gd_ingest_begin and kept until gd_destroy
    // (page-locking 128 MB per contig would cost more than many contigs' whole decode)
    uint8_t* ing_stage[8] = {};                        // page-locked staging buffers: two per upload stream
    hipEvent_t ing_staged[8] = {};
    int ing_piece_streams = 1;                         // GD_OPT_INGEST_PIECE_STREAMS: whole pieces alternate over this many streams (copy engines)
    uint64_t ing_piece_seq = 0;
    int ing_cu_split = 0;                              // GD_OPT_INGEST_CU_SPLIT: every n-th CU for the copy kernel, the rest for the inflate launches
    bool ing_hybrid = false;                           // GD_OPT_INGEST_HYBRID: with two piece streams, the second one's pieces leave through a copy kernel
    hipStream_t ing_dma[3] = {nullptr, nullptr, nullptr};   // GD_OPT_INGEST_DMA > 1: a staged piece leaves in slices on several streams (DMA engines)
    hipEvent_t ing_dma_ev[8][3] = {};
    int ing_dma_n = 1;
    hipStream_t ing_walk = nullptr;                     // GD_OPT_INGEST_WALK_CUS: the record walks' stream, masked to the copy kernel's CUs
    int ing_walk_cus = 0;
    hipEvent_t ing_walk_ev = nullptr;
    int ing_batches = 8;                                // GD_OPT_INGEST_BATCHES
    unsigned ing_copy_grid = 16;                        // GD_OPT_INGEST_COPY_GRID: workgroups of the copy kernel that pulls a staged piece over the link
    hipStream_t ing_hp = nullptr;                       // GD_OPT_INGEST_DMA 0: the piece leaves with a copy kernel on a high-priority stream
    hipStream_t ing_stream[8] = {};                     // inflate launches rotate over these (two pending ranges x 4)
    unsigned ing_launch_seq = 0;
    int ing_copy_threads = 1;                          // GD_OPT_COPY_THREADS: threads filling the staging buffer
    bool h2d_kernel = true;                            // GD_OPT_H2D_KERNEL: staging blocks reach HBM through gd_h2d_kernel
    unsigned h2d_grid = 512;                           // ... its workgroups
    bool ingest_crc = true;                            // GD_OPT_INGEST_CRC
    uint64_t ing_range_hint = 0;                       // GD_OPT_INGEST_RANGE_HINT: bytes of the largest range the caller will feed
    int inflate_kernel = 0;                            // GD_OPT_INFLATE_KERNEL
    unsigned inflate_pad = 0;                          // GD_OPT_INFLATE_LDS_PAD: extra LDS per inflate workgroup (occupancy limiter)
    int32_t bam_n_ref = 0;                             // GD_OPT_BAM_REFS: references of the BAM being read (0: unknown)
    int push_threads = 16;
    FillPool* pool = nullptr; int pool_workers = 0;    // host worker threads (gd_push fills, gd_commit validates), created on first use
    size_t push_chunk = 1u << 20;                      // GD_OPT_PUSH_CHUNK: records per staging block of gd_push                             // GD_OPT_PUSH_THREADS: threads of gd_push filling a ring block
    uint32_t* d_scan_tmp = nullptr; size_t cap_scan_tmp = 0;   // launch_scan: block totals
    void* d_rectab = nullptr; size_t cap_rectab = 0;           // ... and the record table the counting walk leaves for the extraction (device, grow-only)
    uint8_t* h_walk = nullptr; size_t cap_walk = 0;            // gd_ingest_decode: per-segment tables of the record walk (page-locked host memory
                                                               // the walk kernels read and write over the link: no copy command)
    uint32_t* h_ingest = nullptr;                              // page-locked: d_ingest's words as the host reads them (gd_copy_words_kernel)
    uint8_t* d_seq = nullptr;  size_t cap_seq = 0;     // gd_seq_load: one contig's bases, zero padded
    int64_t seq_len = -1;
    uint32_t seq_padded = 0;

    int64_t* export_buf = nullptr;     // gd_set_export: caller-owned device buffer, written by every gd_compute
    int64_t export_max_w = 0, export_cap_b = 0;

    // gd_comm_init: one RCCL communicator per context (gd_api_comm.inc)
    void* comm = nullptr;
    int comm_rank = 0, comm_world = 0;
    hipEvent_t comm_ev[3] = {nullptr, nullptr, nullptr};   // [0], [1]: the last two gathers; [2]: "computed so far"
    unsigned comm_seq = 0;
    const int64_t* comm_last_send = nullptr;   // the buffer the last gd_gather_export reads (it may still be reading it)
    unsigned comm_last_ev = 0;                 // ... and which of comm_ev[0 / 1] is recorded behind it

    ComputeState cs;
    bool computed = false;
    int64_t n_tiles = 0, n_win_total = 0, n_bases = 0;
    std::vector<int2> bounds;             // ordered run boundaries of the last compute
    gd_stats stats{};

    bool profiling = false;
    hipEvent_t ev[GD_K_COUNT + 1] = {};
    float kernel_ms[GD_K_COUNT] = {};
};

namespace {

thread_local bool tl_ingest_feeder = false;              // this thread is a context's reader thread (gd_ingest_feed_fd)

int fail(gd_ctx* c, int code, const char* fmt, ...)
{
    if (c) {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        // the read of a pending range runs on a thread of the context (gd_ingest_feed_fd) beside the caller's own
        // calls: its failures are kept apart and published when the read is joined (ingest_join)
        if (tl_ingest_feeder) c->ing_feeder_err = buf;
        else c->err = buf;
    }
    return code;
}

#define HIPCHK(ctx, call)                                                              \
    do {                                                                               \
        hipError_t e_ = (call);                                                        \
        if (e_ != hipSuccess)                                                          \
            return fail((ctx), e_ == hipErrorOutOfMemory ? GD_E_NOMEM : GD_E_HIP,      \
                        "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_),         \
                        __FILE__, __LINE__);                                           \
    } while (0)

template <typename Tp>
int ensure_dev(gd_ctx* c, Tp** p, size_t* cap, size_t need, bool keep = false, size_t used = 0)
{
    if (need <= *cap && *p) return GD_OK;
    size_t ncap = std::max(need, *cap + *cap / 2);
    if (ncap == 0) ncap = 1;
    Tp* np = nullptr;
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&np), ncap * sizeof(Tp)));
    if (*p) {
        if (keep && used)
            HIPCHK(c, hipMemcpyAsync(np, *p, used * sizeof(Tp), hipMemcpyDeviceToDevice,
                                     c->copy_stream));
        HIPCHK(c, hipStreamSynchronize(c->copy_stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, hipFree(*p));
    }
    *p = np;
    *cap = ncap;
    return GD_OK;
}

void drop_ck(ContigHost& h)
{
    h.ck_blk.reset();                  // (hipFree of the block, when this was its last user, waits for the device)
    h.lrec = nullptr; h.lfq = nullptr; h.dl = nullptr; h.pck = nullptr; h.ndel = nullptr;
    h.max_span = 0;
    h.n_dels = 0;
    h.ck_ok = false;
}

void drop_norm(ContigHost& h) { drop_ck(h); }   // everything derived from the records

void free_contig(ContigHost& h)
{
    drop_norm(h);
    if (!h.adopted) {
        if (h.pos) (void)hipFree(h.pos);
        if (h.flag) (void)hipFree(h.flag);
        if (h.mapq) (void)hipFree(h.mapq);
        if (h.off) (void)hipFree(h.off);
        if (h.cigar) (void)hipFree(h.cigar);
    }
    drop_ck(h);
    if (h.ridx) (void)hipFree(h.ridx);
    h.ridx = nullptr; h.ridx_reads = 0;
    h.ing_left = false;
    h.pos = nullptr; h.flag = nullptr; h.mapq = nullptr; h.off = nullptr; h.cigar = nullptr;
    h.n_reads = h.n_ops = h.cap_reads = h.cap_ops = 0;
    h.adopted = false;
    h.last_pos = -0x7fffffff;
    h.base_off = h.win_off = -1;
    h.n_win = 0;
    h.run_beg = h.run_end = 0;
}

int64_t derive_step(const gd_params& p)
{
    if (p.step > 0) return p.step;
    // depth/depth.go:48,:132
    int64_t s = 10000000 / p.window_size;
    if (s < 1) s = 1;
    return s * p.window_size;
}

// (m, s) with floor(x / d) == (x * m) >> s for every x < 2^31 (1 <= d < 2^31):
// s = 31 + ceil(log2 d), m = ceil(2^s / d) < 2^32  (Granlund & Montgomery 1994, N = 31).
void magic_u31(uint32_t d, uint32_t* m, uint32_t* s)
{
    if (d == 0) d = 1;
    uint32_t l = 0;
    while (l < 31 && (1u << l) < d) ++l;
    const unsigned __int128 num = (unsigned __int128)1 << (31 + l);
    *m = (uint32_t)((num + d - 1) / d);
    *s = 31 + l;
}

int set_device(gd_ctx* c)
{
    HIPCHK(c, hipSetDevice(c->device));
    return GD_OK;
}

template <int T>
void launch_prep(gd_ctx* c, const gd::Job& job)
{
    // one thread per tile; the same threads grid-stride over the window arrays (a thread per window made a 30x
    // genome's launch six rounds of workgroups, five of them doing 12 bytes of work per thread)
    int64_t work = std::max<int64_t>(job.n_tiles, std::min<int64_t>(job.n_win_total / 8, 1 << 20));
    int blocks = (int)((work + 255) / 256);
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(gd::gd_prep_kernel<T>, dim3(blocks), dim3(256), 0, c->stream, job);
}

template <int T, int NT>
void launch_tile(gd_ctx* c, const gd::Job& job)
{
    // 8 XCDs: the grid is 8 equal slices of the tile list (see the kernel)
    const unsigned grid = (unsigned)(((job.n_tiles + 7) / 8) * 8);
    if (c->ran_sums_only) {                                 // decided by gd_compute for this run
        hipLaunchKernelGGL((gd::gd_tile_sums_kernel<4096, 256>), dim3(grid), dim3(256), 0, c->stream, job);
        return;
    }
    if (job.fast) {
        // ordinary tiles: the straight-line kernel; the tiles gd_prep_kernel listed as `slow` (clipped at a
        // contig end, deeper than one batch of reads, more ops than the staging area): the generic one
        // The slow list (a few dozen workgroups of a large, cold kernel) goes FIRST: launched after the
        // straight-line kernel it took ~0.1 ms -- as long as all of chr20's ordinary tiles -- behind the
        // write-back of that kernel's per-base stores; in front of it, it costs a few microseconds
        // (profiles/r02d_slow_first.txt; a side stream next to the straight-line kernel bought nothing).
        const unsigned sgrid = c->slow_grid;                 // strides over the slow list (usually one tile per contig)
        if (!c->keep_perbase)
            hipLaunchKernelGGL((gd::gd_tile_slow_kernel<4096, 256, 2>), dim3(sgrid), dim3(256), 0, c->stream, job);
        else if (c->tile_opt & 1)
            hipLaunchKernelGGL((gd::gd_tile_slow_kernel<4096, 256, 1>), dim3(sgrid), dim3(256), 0, c->stream, job);
        else
            hipLaunchKernelGGL((gd::gd_tile_slow_kernel<4096, 256, 0>), dim3(sgrid), dim3(256), 0, c->stream, job);
        if (job.fast == 2u) {                                // the records as they arrived
            if (!c->keep_perbase)
                hipLaunchKernelGGL((gd::fast::gd_tile_fast_kernel<2>), dim3(grid), dim3(256), 0, c->stream, job);
            else if (c->tile_opt & 1)
                hipLaunchKernelGGL((gd::fast::gd_tile_fast_kernel<1>), dim3(grid), dim3(256), 0, c->stream, job);
            else
                hipLaunchKernelGGL((gd::fast::gd_tile_fast_kernel<0>), dim3(grid), dim3(256), 0, c->stream, job);
            return;
        }
        return;
    }
    if (!c->keep_perbase)
        hipLaunchKernelGGL((gd::gd_tile_kernel<T, NT, 2>), dim3(grid), dim3(NT), 0, c->stream, job);
    else if (c->tile_opt & 1)
        hipLaunchKernelGGL((gd::gd_tile_kernel<T, NT, 1>), dim3(grid), dim3(NT), 0, c->stream, job);
    else
        hipLaunchKernelGGL((gd::gd_tile_kernel<T, NT, 0>), dim3(grid), dim3(NT), 0, c->stream, job);
}

template <int T, int NT>
void launch_ltile(gd_ctx* c, const gd::Job& job)
{
    const unsigned grid = (unsigned)(((job.n_tiles + 7) / 8) * 8);
    if (!c->keep_perbase)
        hipLaunchKernelGGL((gd::gd_ltile2_kernel<T, NT, 2>), dim3(grid), dim3(NT), 0, c->stream, job);
    else
        hipLaunchKernelGGL((gd::gd_ltile2_kernel<T, NT, 0>), dim3(grid), dim3(NT), 0, c->stream, job);
}

// In-place exclusive scan of v[0..n) on the compute stream, v[n] = total (v has n + 1 elements).
int launch_scan(gd_ctx* c, uint32_t* v, uint32_t n)
{
    if (n <= 4u * gd::norm::SCAN_BLOCK) {
        hipLaunchKernelGGL(gd::norm::gd_unit_scan_kernel, dim3(1), dim3(1024), 0, c->stream, v, n);
        return GD_OK;
    }
    const uint32_t nb = (n + gd::norm::SCAN_BLOCK - 1u) / gd::norm::SCAN_BLOCK;
    if (int r = ensure_dev(c, &c->d_scan_tmp, &c->cap_scan_tmp, (size_t)nb + 1)) return r;
    hipLaunchKernelGGL(gd::norm::gd_scan_totals_kernel, dim3(nb), dim3(256), 0, c->stream, v, n, c->d_scan_tmp);
    hipLaunchKernelGGL(gd::norm::gd_unit_scan_kernel, dim3(1), dim3(1024), 0, c->stream, c->d_scan_tmp, nb);
    hipLaunchKernelGGL(gd::norm::gd_scan_apply_kernel, dim3(nb), dim3(256), 0, c->stream, v, n, c->d_scan_tmp, nb);
    return GD_OK;
}

// ---- derived record arrays, built per BATCH of contigs ----------------------------------------------------
// Bump allocation inside one DevBlock: every array starts on a 256-byte boundary.
struct Carve {
    size_t at = 0;
    size_t take(size_t bytes) { const size_t o = at; at += (bytes + 255) & ~(size_t)255; return o; }
};

// A block of at least `need` bytes: `keep` (what the batch's contigs held before) when nobody else uses it and it
// is large enough -- a re-normalisation of the same contigs allocates nothing -- else a new allocation.
int batch_block(gd_ctx* c, BlockRef keep, size_t need, BlockRef* out)
{
    if (keep && keep.use_count() == 1 && keep->bytes >= need) { *out = std::move(keep); return GD_OK; }
    keep.reset();
    BlockRef b = std::make_shared<DevBlock>();
    const auto ta = std::chrono::steady_clock::now();
    HIPCHK(c, hipMalloc(&b->p, need ? need : 1));
    // (a first large allocation can wait for the driver to clear memory another process -- or this one -- just released:
    // 0.4 ms on one box, half a second on another; gd_compute_timing reports it with the other device allocations)
    c->alloc_in_enqueue += std::chrono::duration<double>(std::chrono::steady_clock::now() - ta).count();
    b->bytes = need ? need : 1;
    *out = std::move(b);
    return GD_OK;
}

int batch_host(gd_ctx* c, size_t words)
{
    if (words <= c->cap_h_batch && c->h_batch) return GD_OK;
    if (c->h_batch) { (void)hipHostFree(c->h_batch); c->h_batch = nullptr; c->cap_h_batch = 0; }
    const size_t cap = std::max<size_t>(words, 256);
    HIPCHK(c, hipHostMalloc(reinterpret_cast<void**>(&c->h_batch), cap * sizeof(uint32_t), hipHostMallocDefault));
    c->cap_h_batch = cap;
    return GD_OK;
}

// Long-read path: deletion lists, read records and tile indexes (gd_chunk.hpp) of a batch of contigs, straight from the
// records as they arrived: ONE allocation sized from the record counts alone (deletion lists: half the ops + a slot per
// read; tile indexes: an entry per 64 ops + three per read -- gd_chunk.hpp pt_slot), ONE pass (gd_dels_raw_kernel fills the
// index as it walks), one small kernel that puts every contig's deletion total and largest span into page-locked memory.
// Until round 5 the index was three more launches with a host synchronisation and a second allocation between them.
struct CkPending {
    std::vector<ContigHost*> hs;
    BlockRef blk;
    size_t o_jobs = 0, o_tot = 0;
    std::vector<size_t> o_lrec, o_lfq, o_dl, o_ndel, o_pck;
    uint32_t n_units = 0;
    gd::DelBatch B{};
};

int ck_enqueue(gd_ctx* c, const std::vector<ContigHost*>& hs, CkPending* P)
{
    P->hs = hs;
    const size_t nj = hs.size();
    if (nj == 0) return GD_OK;
    BlockRef keep = hs[0]->ck_blk;
    for (ContigHost* h : hs) drop_ck(*h);
    Carve cv;
    uint64_t units = 0;
    P->o_lrec.resize(nj); P->o_lfq.resize(nj); P->o_dl.resize(nj); P->o_ndel.resize(nj); P->o_pck.resize(nj);
    for (size_t k = 0; k < nj; ++k) {
        const ContigHost& h = *hs[k];
        const size_t n = h.n_reads;
        const size_t n_dl = (h.n_ops >> 1) + n + 1;
        const size_t n_pck = (h.n_ops >> 6) + 3 * n + 4;      // (< 2^32: at most 2^30 reads and 2^32 ops per contig)
        if (n_dl > 0xffffffffull) return fail(c, GD_E_RANGE, "too many deletions on one contig");
        P->o_lrec[k] = cv.take((n + 2) * sizeof(uint4));
        P->o_lfq[k] = cv.take((n + 1) * sizeof(uint32_t));
        P->o_ndel[k] = cv.take((n + 1) * sizeof(uint32_t));
        P->o_dl[k] = cv.take(n_dl * sizeof(uint2));
        P->o_pck[k] = cv.take(n_pck * sizeof(uint32_t));
        units += (n + 63) / 64;
    }
    if (units > (1ull << 26) - 64) return fail(c, GD_E_RANGE, "too many records in one batch (internal error)");
    P->n_units = (uint32_t)units;
    P->o_jobs = cv.take(nj * sizeof(gd::DelJob) + (nj + 1) * sizeof(uint32_t));   // the jobs, then ubeg
    P->o_tot = cv.take(3 * nj * sizeof(uint32_t));               // [unused][deletions][largest span] per contig
    if (int r = batch_block(c, std::move(keep), cv.at, &P->blk)) return r;
    char* const base = static_cast<char*>(P->blk->p);
    // job table (host copy kept in the context until the next batch)
    c->batch_tab_ck.assign(nj * sizeof(gd::DelJob) + (nj + 1) * sizeof(uint32_t), 0);
    gd::DelJob* jobs = reinterpret_cast<gd::DelJob*>(c->batch_tab_ck.data());
    uint32_t* ubeg = reinterpret_cast<uint32_t*>(c->batch_tab_ck.data() + nj * sizeof(gd::DelJob));
    uint32_t u = 0;
    for (size_t k = 0; k < nj; ++k) {
        ContigHost& h = *hs[k];
        const uint32_t n = (uint32_t)h.n_reads, nu = (n + 63u) / 64u;
        gd::DelJob& j = jobs[k];
        j.pos = h.pos; j.flag = h.flag; j.mapq = h.mapq;
        j.off = h.off;
        j.cigar = h.cigar;
        j.n_reads = n; j.n_units = nu;
        j.lrec = reinterpret_cast<uint4*>(base + P->o_lrec[k]);
        j.lfq = reinterpret_cast<uint32_t*>(base + P->o_lfq[k]);
        j.dl = reinterpret_cast<uint2*>(base + P->o_dl[k]);
        j.ndel = reinterpret_cast<uint32_t*>(base + P->o_ndel[k]);
        j.del_total = reinterpret_cast<uint32_t*>(base + P->o_tot) + nj + k;
        j.max_span = reinterpret_cast<int32_t*>(j.lrec + n + 1);
        j.pck = reinterpret_cast<uint32_t*>(base + P->o_pck[k]);
        j.total = reinterpret_cast<uint32_t*>(base + P->o_tot) + k;
        ubeg[k] = u;
        u += nu;
    }
    ubeg[nj] = u;
    HIPCHK(c, hipMemcpyAsync(base + P->o_jobs, c->batch_tab_ck.data(), c->batch_tab_ck.size(), hipMemcpyHostToDevice, c->stream));
    static_assert(sizeof(gd::DelJob) % 8 == 0, "the ubeg table follows the jobs");
    gd::DelBatch& B = P->B;
    B.jobs = reinterpret_cast<const gd::DelJob*>(base + P->o_jobs);
    B.ubeg = reinterpret_cast<const uint32_t*>(base + P->o_jobs + nj * sizeof(gd::DelJob));
    B.n_jobs = (uint32_t)nj; B.n_units = P->n_units;
    for (size_t k = 0; k < nj; ++k)                          // lrec[n], lrec[n + 1] (the span accumulator)
        HIPCHK(c, hipMemsetAsync(base + P->o_lrec[k] + hs[k]->n_reads * sizeof(uint4), 0, 2 * sizeof(uint4), c->stream));
    HIPCHK(c, hipMemsetAsync(base + P->o_tot, 0, 2 * nj * sizeof(uint32_t), c->stream));
    if (P->n_units)
        hipLaunchKernelGGL(gd::gd_dels_raw_kernel, dim3((P->n_units + 3u) / 4u), dim3(256), 0, c->stream, B);
    hipLaunchKernelGGL(gd::gd_ptile_totals_kernel, dim3((unsigned)((nj + 255) / 256)), dim3(256), 0, c->stream, B);
    HIPCHK(c, hipGetLastError());
    return GD_OK;
}

// host words the totals of a ck batch take in gd_ctx::h_batch: [n_jobs unused][n_jobs deletion counts][n_jobs spans]
int ck_readback(gd_ctx* c, const CkPending& P, uint32_t* dst)
{
    const size_t nj = P.hs.size();
    if (nj == 0) return GD_OK;
    char* const base = static_cast<char*>(P.blk->p);
    // (dst is page-locked: a one-wave kernel stores the 3 n words there -- no copy commands; there used to be 1 + n of them)
    hipLaunchKernelGGL(gd::gd_copy_words_kernel, dim3(1), dim3(64), 0, c->stream, reinterpret_cast<const uint32_t*>(base + P.o_tot), dst,
                       (uint32_t)(3 * nj));
    HIPCHK(c, hipGetLastError());
    return GD_OK;
}

// after the synchronisation: publish (the spans are the look-back of the tile table)
int ck_finish(gd_ctx* c, CkPending& P, const uint32_t* tot)
{
    const size_t nj = P.hs.size();
    if (nj == 0) return GD_OK;
    gd::DelJob* jobs = reinterpret_cast<gd::DelJob*>(c->batch_tab_ck.data());
    for (size_t k = 0; k < nj; ++k) {
        ContigHost& h = *P.hs[k];
        h.ck_blk = P.blk;
        h.lrec = jobs[k].lrec; h.lfq = jobs[k].lfq; h.dl = jobs[k].dl; h.pck = jobs[k].pck; h.ndel = jobs[k].ndel;
        h.max_span = (int32_t)tot[2 * nj + k];
        h.n_dels = tot[nj + k];
        h.ck_ok = true;
    }
    return GD_OK;
}

// Builds the long-read structures of a batch of contigs.
int ck_batch(gd_ctx* c, const std::vector<ContigHost*>& hs)
{
    if (hs.empty()) return GD_OK;
    HIPCHK(c, hipEventRecord(c->copy_done, c->copy_stream));
    HIPCHK(c, hipStreamWaitEvent(c->stream, c->copy_done, 0));
    if (c->profiling) HIPCHK(c, hipEventRecord(c->ev[6], c->stream));
    CkPending P;
    if (int r = ck_enqueue(c, hs, &P)) return r;
    if (int r = batch_host(c, 3 * hs.size())) return r;
    if (int r = ck_readback(c, P, c->h_batch)) return r;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (int r = ck_finish(c, P, c->h_batch)) return r;
    if (c->profiling) {
        HIPCHK(c, hipEventRecord(c->ev[7], c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        float ms = 0;
        HIPCHK(c, hipEventElapsedTime(&ms, c->ev[6], c->ev[7]));
        c->kernel_ms[GD_K_CKPT] += ms;
    }
    return GD_OK;
}

// What GD_PATH_AUTO sends to the long-read path (contig-wise).
bool long_shaped(const gd_ctx* c, const ContigHost& h)
{
    return c->path == GD_PATH_CHUNK || (c->path == GD_PATH_AUTO && (c->span_forces_long || h.n_ops > 6 * h.n_reads));
}

// The long-read structures of the listed contigs that lack them, straight from the records as they arrived.
int ck_tids(gd_ctx* c, const std::vector<int32_t>& tids)
{
    // A launch holds fewer than 2^32 work-items (the dispatch packet's grid size is 32 bits; a larger grid wraps
    // silently): at one wave per 64-read unit that is 2^26 units, so a very large job is several batches.
    constexpr uint64_t kMaxUnits = 48u << 20;
    std::vector<ContigHost*> part;
    uint64_t units = 0;
    auto flush = [&]() -> int {
        if (part.empty()) return GD_OK;
        const int r = ck_batch(c, part);
        part.clear(); units = 0;
        return r;
    };
    for (int32_t tid : tids) {
        ContigHost& h = c->contigs[tid];
        if (h.length <= 0 || h.ck_ok) continue;
        if (!part.empty() && units + (h.n_reads + 63) / 64 > kMaxUnits)
            if (int r = flush()) return r;
        units += (h.n_reads + 63) / 64;
        part.push_back(&h);
    }
    return flush();
}

// RAII for the scratch device buffers of gd_ingest_bgzf
struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1); }
    template <typename T> T* as() const { return static_cast<T*>(p); }
};

}  // namespace

struct IngestState {
    static constexpr size_t kStage = 64u << 20;        // bytes per page-locked staging buffer
    uint64_t n_bytes = 0, fed = 0, total = 0;           // compressed bytes announced / received, inflated bytes
    size_t nm = 0, next = 0;                            // members, first member not yet handed to the inflate kernel
    std::vector<uint64_t> m_coff, m_end, out_off;       // file offset, end offset in the range, offset in the inflated bytes
    std::vector<uint32_t> out_len;
    IngestBufs* bufs = nullptr;                         // one of gd_ctx::ing_bufs
    uint8_t *d_in = nullptr, *d_out = nullptr;
    uint64_t *t_in_off = nullptr, *t_out_off = nullptr;
    uint32_t *t_in_len = nullptr, *t_out_len = nullptr, *t_status = nullptr, *t_crc = nullptr;
    // One lane inflates one member start to end (~0.04 s whatever the member count), so the members are
    // handed to the kernel in at most kBatches launches, each on its own stream: they overlap each other
    // and the upload of the bytes still to come.
    static constexpr int kBatches = 8;
    int n_launch = 0;
    std::vector<hipEvent_t> inf_done;                   // one per inflate launch of THIS range
    bool inflated = false;                              // every member inflated and its status checked
    ~IngestState()
    {
        for (hipEvent_t e : inf_done) (void)hipEventDestroy(e);
        if (bufs) bufs->busy = false;
    }
};
