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
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>

namespace {

constexpr int kRingSlots = 3;
constexpr int kDefaultLookback = 512;
constexpr uint64_t kMaxReadsPerContig = 1ull << 30;
constexpr int kAutoLongSpan = 32768;      // GD_PATH_AUTO leaves the short-read tile path above this read span
constexpr int kMaxSpan = 1 << 27;   // tile-relative byte offsets of the tile kernel stay in 32 bits

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
    // packed descriptors of the records (gd_tile_v8.hpp), always owned
    uint2* desc = nullptr;
    uint32_t* cxb = nullptr;
    uint32_t* cxc = nullptr;
    bool packed = false;               // desc/cxb/cxc describe the current records
    bool packable = false;             // ... and the v8 kernels may use them
    // layout in the result arrays of the last compute (-1 = not computed)
    int64_t base_off = -1;
    int64_t win_off = -1;
    int64_t n_win = 0;
    size_t run_beg = 0, run_end = 0;   // slice of ctx->bounds
};

struct RingSlot {
    gd_batch b{};
    hipEvent_t done = nullptr;
    bool busy = false;
};

}  // namespace

// State of a device BAM read between gd_ingest_begin and gd_ingest_finish.
struct IngestState;

// Device buffers of one pending range (compressed bytes, inflated bytes, member tables): grow-only and
// kept by the context between ranges -- allocating and freeing gigabytes per range cost 0.1-0.2 s.
struct IngestBufs {
    void *in = nullptr, *out = nullptr, *tab = nullptr;
    size_t cap_in = 0, cap_out = 0, cap_tab = 0;
    bool busy = false;
    static bool fit(void** p, size_t* cap, size_t need)
    {
        if (need <= *cap && *p) return true;
        if (*p) { (void)hipFree(*p); *p = nullptr; *cap = 0; }
        if (hipMalloc(p, need ? need : 1) != hipSuccess) { *p = nullptr; return false; }
        *cap = need ? need : 1;
        return true;
    }
    void drop()
    {
        if (in) (void)hipFree(in);
        if (out) (void)hipFree(out);
        if (tab) (void)hipFree(tab);
        in = out = tab = nullptr;
        cap_in = cap_out = cap_tab = 0;
    }
};


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
    std::string err;

    int tile_T = 4096;
    int tile_NT = 256;
    int kernel_gen = 7;                 // GOLEFT_GD_KERNEL=v8: packed read descriptors (gd_tile_v8.hpp, measured
                                        // equal to v7: DESIGN.md section 4); v6: the previous tile kernel
    bool use_v8 = false;                // this gd_compute: every contig of the job is packed
    int tile_opt = 1;                   // bit 0: non-temporal per-base stores (2 % faster: the vector is
                                        // never re-read by the kernel); GOLEFT_GD_OPT=0 for plain stores
    bool lookback_pinned = false;       // max_span_hint given: never shrink below it
    int path = GD_PATH_AUTO;            // gd_set_path / GOLEFT_GD_PATH
    bool keep_perbase = true;           // gd_set_outputs(GD_OUT_PERBASE)
    bool sums_only = false;             // gd_set_outputs(GD_OUT_SUMS_ONLY): window sums, nothing else
    bool ran_sums_only = false;         // what the last gd_compute produced
    bool span_forces_long = false;      // AUTO: the tile path met a read too long for it
    unsigned long long* d_status = nullptr;  size_t cap_status = 0;   // scatter path look-back words
    uint32_t* d_ck = nullptr;  size_t cap_ck = 0;      // chunk path: CIGAR checkpoints
    int32_t* d_rend = nullptr; size_t cap_rend = 0;    // chunk path: read end positions
    int lookback = kDefaultLookback;

    // device job state
    gd::ContigDev* d_ctgs = nullptr;  size_t cap_ctgs = 0;
    std::vector<gd::ContigDev> h_ctgs;
    std::vector<int32_t> job_tids;      // contig table index -> tid
    gd::TileInfo* d_tiles = nullptr;  size_t cap_tiles = 0;
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
    uint32_t* d_region_cursor = nullptr;

    int64_t* d_wed = nullptr; size_t cap_wed = 0;      // gd_depthwed: tables + the sites x samples matrix
    uint32_t* d_md_bits = nullptr; size_t cap_md = 0;  // gd_md_flags: `any` words then `suf` words
    int64_t md_len = -1;                               // positions the bitmaps cover (-1: none yet)
    std::vector<int32_t> md_tids;                      // the samples they were built from
    // gd_ingest_begin .. gd_ingest_finish.  Two ranges may be pending: ing_q[0] is the oldest (the one
    // gd_ingest_decode / _finish / _release act on), the last one is being fed -- so the inflate tail of
    // one range overlaps the upload of the next.
    IngestState* ing_q[2] = {nullptr, nullptr};
    int ing_n = 0;
    bool ing_stage_used[2] = {false, false};
    int ing_cur = 0;
    IngestBufs ing_bufs[2];
    // staging of the device BAM read, created by the first gd_ingest_begin and kept until gd_destroy
    // (page-locking 128 MB per contig would cost more than many contigs' whole decode)
    uint8_t* ing_stage[2] = {nullptr, nullptr};
    hipEvent_t ing_staged[2] = {nullptr, nullptr};
    hipStream_t ing_stream[8] = {};                     // inflate launches rotate over these (two pending ranges x 4)
    unsigned ing_launch_seq = 0;
    int ing_copy_threads = 1;                          // GOLEFT_GD_COPY_THREADS: threads filling the staging buffer
    uint8_t* d_seq = nullptr;  size_t cap_seq = 0;     // gd_seq_load: one contig's bases, zero padded
    int64_t seq_len = -1;
    uint32_t seq_padded = 0;

    bool computed = false;
    int64_t n_tiles = 0, n_win_total = 0, n_bases = 0;
    std::vector<int2> bounds;             // ordered run boundaries of the last compute
    gd_stats stats{};

    bool profiling = false;
    hipEvent_t ev[GD_K_COUNT + 1] = {};
    float kernel_ms[GD_K_COUNT] = {};
};

namespace {

int fail(gd_ctx* c, int code, const char* fmt, ...)
{
    if (c) {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        c->err = buf;
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

void drop_pack(ContigHost& h)
{
    if (h.desc) (void)hipFree(h.desc);
    if (h.cxb) (void)hipFree(h.cxb);
    if (h.cxc) (void)hipFree(h.cxc);
    h.desc = nullptr; h.cxb = nullptr; h.cxc = nullptr;
    h.packed = h.packable = false;
}

void free_contig(ContigHost& h)
{
    drop_pack(h);
    if (!h.adopted) {
        if (h.pos) (void)hipFree(h.pos);
        if (h.flag) (void)hipFree(h.flag);
        if (h.mapq) (void)hipFree(h.mapq);
        if (h.off) (void)hipFree(h.off);
        if (h.cigar) (void)hipFree(h.cigar);
    }
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
    int64_t work = std::max<int64_t>(job.n_tiles, std::min<int64_t>(job.n_win_total, 1 << 22));
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
        if (c->use_v8) hipLaunchKernelGGL((gd::v8::gd_tile_sums_kernel<4096, 256>), dim3(grid), dim3(256), 0, c->stream, job);
        else           hipLaunchKernelGGL((gd::v7::gd_tile_sums_kernel<4096, 256>), dim3(grid), dim3(256), 0, c->stream, job);
        return;
    }
    if (c->use_v8 && T == 4096 && NT == 256) {              // packed descriptors (default shape only)
        if (!c->keep_perbase)
            hipLaunchKernelGGL((gd::v8::gd_tile_kernel<4096, 256, 2>), dim3(grid), dim3(256), 0, c->stream, job);
        else if (c->tile_opt & 1)
            hipLaunchKernelGGL((gd::v8::gd_tile_kernel<4096, 256, 1>), dim3(grid), dim3(256), 0, c->stream, job);
        else
            hipLaunchKernelGGL((gd::v8::gd_tile_kernel<4096, 256, 0>), dim3(grid), dim3(256), 0, c->stream, job);
        return;
    }
    if (c->kernel_gen >= 7 && T == 4096 && NT == 256) {     // v7 is built for the default shape only
        if (!c->keep_perbase)
            hipLaunchKernelGGL((gd::v7::gd_tile_kernel<4096, 256, 2>), dim3(grid), dim3(256), 0, c->stream, job);
        else if (c->tile_opt & 1)
            hipLaunchKernelGGL((gd::v7::gd_tile_kernel<4096, 256, 1>), dim3(grid), dim3(256), 0, c->stream, job);
        else
            hipLaunchKernelGGL((gd::v7::gd_tile_kernel<4096, 256, 0>), dim3(grid), dim3(256), 0, c->stream, job);
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
    if (c->kernel_gen == 6) {                              // GOLEFT_GD_KERNEL=v6: the first long-read kernel
        if (!c->keep_perbase)
            hipLaunchKernelGGL((gd::gd_ltile_kernel<T, NT, 2>), dim3(grid), dim3(NT), 0, c->stream, job);
        else
            hipLaunchKernelGGL((gd::gd_ltile_kernel<T, NT, 0>), dim3(grid), dim3(NT), 0, c->stream, job);
        return;
    }
    if (!c->keep_perbase)
        hipLaunchKernelGGL((gd::gd_ltile2_kernel<T, NT, 2>), dim3(grid), dim3(NT), 0, c->stream, job);
    else
        hipLaunchKernelGGL((gd::gd_ltile2_kernel<T, NT, 0>), dim3(grid), dim3(NT), 0, c->stream, job);
}

// Builds the packed descriptors of one contig's records (gd_tile_v8.hpp) on the compute stream.
// Afterwards h.packed is set; h.packable says whether the v8 kernels may use them.
int pack_contig(gd_ctx* c, ContigHost& h)
{
    drop_pack(h);
    h.packed = true;
    if (h.n_reads >= (1ull << 29)) return GD_OK;           // descriptor byte offsets stay in 32 bits
    const uint32_t n_reads = (uint32_t)h.n_reads, n_units = (n_reads + 63u) / 64u;
    // records staged on the copy stream must have landed
    HIPCHK(c, hipEventRecord(c->copy_done, c->copy_stream));
    HIPCHK(c, hipStreamWaitEvent(c->stream, c->copy_done, 0));
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&h.desc), std::max<size_t>(n_reads, 1) * sizeof(uint2)));
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&h.cxb), ((size_t)n_units + 2) * sizeof(uint32_t)));   // + total, status
    HIPCHK(c, hipMemsetAsync(h.cxb, 0, ((size_t)n_units + 2) * sizeof(uint32_t), c->stream));
    gd::v8::PackJob j{};
    j.pos = h.pos; j.flag = h.flag; j.mapq = h.mapq; j.off = h.off; j.cigar = h.cigar;
    j.n_reads = n_reads; j.n_units = n_units; j.desc = h.desc; j.cx_base = h.cxb; j.status = h.cxb + n_units + 1;
    uint32_t tail[2] = {0, 0};                              // grand total of compact ops, status bits
    if (c->profiling) HIPCHK(c, hipEventRecord(c->ev[0], c->stream));
    if (n_units) {
        hipLaunchKernelGGL(gd::v8::gd_pack_desc_kernel, dim3((n_units + 3u) / 4u), dim3(256), 0, c->stream, j);
        hipLaunchKernelGGL(gd::v8::gd_pack_scan_kernel, dim3(1), dim3(1024), 0, c->stream, h.cxb, n_units);
        HIPCHK(c, hipMemcpyAsync(tail, h.cxb + n_units, sizeof tail, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&h.cxc), ((size_t)tail[0] + 4) * sizeof(uint32_t)));
    if (tail[0]) {
        j.cx_cigar = h.cxc;
        hipLaunchKernelGGL(gd::v8::gd_pack_ops_kernel, dim3((n_units + 3u) / 4u), dim3(256), 0, c->stream, j);
    }
    HIPCHK(c, hipGetLastError());
    if (c->profiling) {
        float ms = 0;
        HIPCHK(c, hipEventRecord(c->ev[1], c->stream));
        HIPCHK(c, hipEventSynchronize(c->ev[1]));
        HIPCHK(c, hipEventElapsedTime(&ms, c->ev[0], c->ev[1]));
        c->kernel_ms[GD_K_PACK] += ms;
    }
    h.packable = tail[1] == 0;
    return GD_OK;
}

// The v8 kernels exist for the default tile shape; long-read data goes to the chunk path anyway.
bool wants_pack(const gd_ctx* c, uint64_t n_reads, uint64_t n_ops)
{
    if (c->kernel_gen != 8 || c->tile_T != 4096 || c->tile_NT != 256) return false;
    if (c->path == GD_PATH_TILE) return true;
    return c->path == GD_PATH_AUTO && !c->span_forces_long && n_ops <= 6 * n_reads;
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
    // One lane inflates one member start to end (~0.1 s whatever the member count), so the members are
    // handed to the kernel in at most kBatches launches, each on its own stream: they overlap each other
    // and the upload of the bytes still to come.
    static constexpr int kBatches = 4;
    int n_launch = 0;
    std::vector<hipEvent_t> inf_done;                   // one per inflate launch of THIS range
    bool inflated = false;                              // every member inflated and its status checked
    ~IngestState()
    {
        for (hipEvent_t e : inf_done) (void)hipEventDestroy(e);
        if (bufs) bufs->busy = false;
    }
};

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
    if (const char* e = getenv("GOLEFT_GD_TILE")) {
        int t = atoi(e);
        if (t == 4096 || t == 8192) c->tile_T = t;
    }
    if (const char* e = getenv("GOLEFT_GD_KERNEL"))
        c->kernel_gen = (e[0] == 'v' && e[1] == '6') ? 6 : (e[0] == 'v' && e[1] == '8') ? 8 : 7;
    if (const char* e = getenv("GOLEFT_GD_OPT")) c->tile_opt = atoi(e) & 1;
    if (const char* e = getenv("GOLEFT_GD_PATH"))
        c->path = e[0] == 's' ? GD_PATH_SCATTER : e[0] == 't' ? GD_PATH_TILE : e[0] == 'c' ? GD_PATH_CHUNK : GD_PATH_AUTO;
    if (const char* e = getenv("GOLEFT_GD_COPY_THREADS")) { const int t = atoi(e); if (t >= 1 && t <= 16) c->ing_copy_threads = t; }
    if (const char* e = getenv("GOLEFT_GD_THREADS")) {
        int t = atoi(e);
        if (t == 256 || t == 512) c->tile_NT = t;
    }
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
    *out = c;
    return GD_OK;
}

void gd_destroy(gd_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);
    (void)gd_ingest_abort(c);
    for (int k = 0; k < 2; ++k) {
        if (c->ing_stage[k]) (void)hipHostFree(c->ing_stage[k]);
        if (c->ing_staged[k]) (void)hipEventDestroy(c->ing_staged[k]);
    }
    for (hipStream_t st : c->ing_stream) if (st) (void)hipStreamDestroy(st);
    for (auto& b : c->ing_bufs) b.drop();
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
    void* frees[] = {c->d_ctgs, c->d_tiles, c->d_perbase, c->d_wsum, c->d_wmin, c->d_chunks,
                     c->d_ordered, c->d_tile_cnt, c->d_tile_off, c->d_super_cnt, c->d_counters,
                     c->d_region_cursor, c->d_status, c->d_ck, c->d_rend, c->d_seq, c->d_md_bits, c->d_wed};
    for (void* p : frees) if (p) (void)hipFree(p);
    if (c->h_counters) (void)hipHostFree(c->h_counters);
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
    if (!c) return GD_E_INVALID;
    if (path != GD_PATH_AUTO && path != GD_PATH_TILE && path != GD_PATH_SCATTER && path != GD_PATH_CHUNK)
        return fail(c, GD_E_INVALID, "unknown path %d", path);
    c->path = path;
    c->computed = false;
    return GD_OK;
}

int gd_set_outputs(gd_ctx* c, unsigned flags)
{
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
    if (!c || n < 0 || (n > 0 && !lengths)) return GD_E_INVALID;
    if (int r = set_device(c)) return r;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipStreamSynchronize(c->copy_stream));
    for (auto& h : c->contigs) free_contig(h);
    c->contigs.assign(n, ContigHost());
    for (int i = 0; i < n; ++i) {
        if (lengths[i] < 0 || lengths[i] > 0x7fffffffLL)
            return fail(c, GD_E_RANGE, "contig %d length %lld out of range", i, (long long)lengths[i]);
        c->contigs[i].length = lengths[i];
    }
    c->selected.clear();
    c->computed = false;
    // a new data set: forget the look-back learnt from the previous one
    c->lookback = c->params.max_span_hint > 0 ? c->params.max_span_hint : kDefaultLookback;
    c->span_forces_long = false;
    return GD_OK;
}

int gd_select_contigs(gd_ctx* c, int n, const int32_t* tids)
{
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

int gd_acquire(gd_ctx* c, size_t reads_cap, size_t ops_cap, gd_batch* out)
{
    if (!c || !out) return GD_E_INVALID;
    if (int r = set_device(c)) return r;
    RingSlot& s = c->ring[c->ring_next];
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
    *out = s.b;
    return GD_OK;
}

int gd_commit(gd_ctx* c, const gd_batch* b, int32_t tid, size_t n_reads, size_t n_ops)
{
    if (!c || !b) return GD_E_INVALID;
    if (int r = set_device(c)) return r;
    if (b->slot < 0 || b->slot >= kRingSlots || c->ring[b->slot].b.pos != b->pos)
        return fail(c, GD_E_INVALID, "batch was not obtained from gd_acquire");
    if (tid < 0 || (size_t)tid >= c->contigs.size()) return fail(c, GD_E_RANGE, "tid %d out of range", tid);
    if (n_reads > b->reads_cap || n_ops > b->ops_cap) return fail(c, GD_E_INVALID, "batch overflow");
    RingSlot& s = c->ring[b->slot];
    c->ring_next = (b->slot + 1) % kRingSlots;
    if (n_reads == 0) return GD_OK;
    ContigHost& h = c->contigs[tid];
    if (h.adopted) return fail(c, GD_E_STATE, "contig %d holds adopted device records", tid);
    if (b->cigar_off[0] != 0 || b->cigar_off[n_reads] != n_ops)
        return fail(c, GD_E_INVALID, "cigar_off must start at 0 and end at n_ops");
    if ((uint64_t)h.n_ops + n_ops > 0xffffffffull)
        return fail(c, GD_E_RANGE, "more than 2^32 CIGAR ops on contig %d", tid);
    if ((uint64_t)h.n_reads + n_reads >= kMaxReadsPerContig)
        return fail(c, GD_E_RANGE, "more than 2^30 records on contig %d", tid);
    // coordinate order (BAM SO:coordinate) is what makes the tile search valid
    int32_t last = h.last_pos;
    for (size_t i = 0; i < n_reads; ++i) {
        if (b->pos[i] < last) return fail(c, GD_E_UNSORTED, "contig %d record %zu: pos %d < %d", tid, h.n_reads + i, b->pos[i], last);
        if (b->cigar_off[i + 1] < b->cigar_off[i]) return fail(c, GD_E_INVALID, "cigar_off not monotone");
        last = b->pos[i];
    }
    // rebase CSR offsets to the contig stream
    const uint32_t base = (uint32_t)h.n_ops;
    if (base)
        for (size_t i = 0; i <= n_reads; ++i) b->cigar_off[i] += base;
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
    HIPCHK(c, hipMemcpyAsync(h.pos + h.n_reads, b->pos, n_reads * sizeof(int32_t), hipMemcpyHostToDevice, cs));
    HIPCHK(c, hipMemcpyAsync(h.flag + h.n_reads, b->flag, n_reads * sizeof(uint16_t), hipMemcpyHostToDevice, cs));
    HIPCHK(c, hipMemcpyAsync(h.mapq + h.n_reads, b->mapq, n_reads * sizeof(uint8_t), hipMemcpyHostToDevice, cs));
    HIPCHK(c, hipMemcpyAsync(h.off + h.n_reads, b->cigar_off, (n_reads + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, cs));
    if (n_ops)
        HIPCHK(c, hipMemcpyAsync(h.cigar + h.n_ops, b->cigar, n_ops * sizeof(uint32_t), hipMemcpyHostToDevice, cs));
    HIPCHK(c, hipEventRecord(s.done, cs));
    s.busy = true;
    if (h.packed) {                                     // the descriptors no longer cover the stream
        HIPCHK(c, hipStreamSynchronize(c->stream));
        drop_pack(h);
    }
    h.n_reads += n_reads;
    h.n_ops += n_ops;
    h.last_pos = last;
    c->computed = false;
    return GD_OK;
}

int gd_push(gd_ctx* c, int32_t tid, const int32_t* pos, const uint16_t* flag, const uint8_t* mapq,
            const uint32_t* cigar_off, const uint32_t* cigar, size_t n_reads, size_t n_ops)
{
    if (!c) return GD_E_INVALID;
    if (n_reads == 0) return GD_OK;
    if (!pos || !flag || !mapq || !cigar_off || (n_ops && !cigar)) return GD_E_INVALID;
    const size_t chunk = 1u << 21;   // records per staging block (three blocks in flight: fill, copy, copy)
    size_t i = 0;
    while (i < n_reads) {
        size_t n = std::min(chunk, n_reads - i);
        size_t o0 = cigar_off[i], o1 = cigar_off[i + n];
        if (o1 < o0 || o1 > n_ops) return fail(c, GD_E_INVALID, "cigar_off out of range");
        gd_batch b;
        if (int r = gd_acquire(c, n, o1 - o0, &b)) return r;
        // the five arrays are independent: fill the pinned block with a few threads (one core
        // moves ~11 GB/s into pinned memory, well under what the PCIe link takes)
        auto fill_off = [&]() { for (size_t k = 0; k <= n; ++k) b.cigar_off[k] = cigar_off[i + k] - (uint32_t)o0; };
        auto fill_cig = [&]() { if (o1 > o0) memcpy(b.cigar, cigar + o0, (o1 - o0) * sizeof(uint32_t)); };
        auto fill_rec = [&]() {
            memcpy(b.pos, pos + i, n * sizeof(int32_t));
            memcpy(b.flag, flag + i, n * sizeof(uint16_t));
            memcpy(b.mapq, mapq + i, n * sizeof(uint8_t));
        };
        if (n >= (1u << 16)) {
            std::thread t1(fill_off), t2(fill_cig);
            fill_rec();
            t1.join();
            t2.join();
        } else {
            fill_rec(); fill_off(); fill_cig();
        }
        if (int r = gd_commit(c, &b, tid, n, o1 - o0)) return r;
        i += n;
    }
    return GD_OK;
}

int gd_adopt_device(gd_ctx* c, int32_t tid, const gd_batch* d, size_t n_reads, size_t n_ops)
{
    if (!c || !d) return GD_E_INVALID;
    if (int r = set_device(c)) return r;
    if (tid < 0 || (size_t)tid >= c->contigs.size()) return fail(c, GD_E_RANGE, "tid %d out of range", tid);
    if (n_reads && (!d->pos || !d->flag || !d->mapq || !d->cigar_off)) return GD_E_INVALID;
    if (n_ops > 0xffffffffull) return fail(c, GD_E_RANGE, "more than 2^32 CIGAR ops");
    // depth <= records of the contig; the window reduction adds four depths in 32 bits
    if (n_reads >= kMaxReadsPerContig) return fail(c, GD_E_RANGE, "more than 2^30 records on one contig");
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipStreamSynchronize(c->copy_stream));
    ContigHost& h = c->contigs[tid];
    int64_t len = h.length;
    free_contig(h);
    h.length = len;
    h.pos = d->pos; h.flag = d->flag; h.mapq = d->mapq; h.off = d->cigar_off; h.cigar = d->cigar;
    h.n_reads = n_reads; h.n_ops = n_ops;
    h.adopted = true;
    c->computed = false;
    // descriptors are part of taking the records in, not of gd_compute
    if (n_reads && wants_pack(c, n_reads, n_ops))
        if (int r = pack_contig(c, h)) return r;
    return GD_OK;
}

int gd_reset(gd_ctx* c)
{
    if (!c) return GD_E_INVALID;
    if (int r = set_device(c)) return r;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipStreamSynchronize(c->copy_stream));
    for (auto& h : c->contigs) {
        int64_t len = h.length;
        free_contig(h);
        h.length = len;
    }
    c->bounds.clear();
    c->computed = false;
    c->lookback = c->params.max_span_hint > 0 ? c->params.max_span_hint : kDefaultLookback;
    c->span_forces_long = false;
    return GD_OK;
}

int gd_compute(gd_ctx* c)
{
    if (!c) return GD_E_INVALID;
    if (int r = set_device(c)) return r;
    if (c->contigs.empty()) return fail(c, GD_E_STATE, "gd_set_contigs has not been called");
    if (c->ing_n == 0)                                   // the BAM read is over: its buffers go back to HBM
        for (auto& b : c->ing_bufs) b.drop();
    const int T = c->tile_T;
    const gd_params& P = c->params;

    // ---- contig table ------------------------------------------------------
    std::vector<int32_t> tids = c->selected;
    if (tids.empty()) { tids.resize(c->contigs.size()); for (size_t i = 0; i < tids.size(); ++i) tids[i] = (int32_t)i; }
    for (auto& h : c->contigs) { h.base_off = h.win_off = -1; h.n_win = 0; h.run_beg = h.run_end = 0; }
    {
        // packed descriptors: built here only for records that arrived through gd_commit
        // (gd_adopt_device and gd_ingest_finish build them on arrival)
        const float pack_ms = c->kernel_ms[GD_K_PACK];
        memset(c->kernel_ms, 0, sizeof c->kernel_ms);
        c->kernel_ms[GD_K_PACK] = pack_ms;
        uint64_t tr = 0, to = 0;
        for (int32_t tid : tids)
            if (c->contigs[tid].length > 0) { tr += c->contigs[tid].n_reads; to += c->contigs[tid].n_ops; }
        bool v8 = wants_pack(c, tr, to);
        for (size_t i = 0; v8 && i < tids.size(); ++i) {
            ContigHost& h = c->contigs[tids[i]];
            if (h.length <= 0) continue;
            if (!h.packed)
                if (int r = pack_contig(c, h)) return r;
            v8 = h.packable;
        }
        c->use_v8 = v8;
    }
    c->h_ctgs.clear();
    c->job_tids.clear();
    int64_t tile_beg = 0, base_off = 0, win_off = 0, bases = 0;
    uint64_t n_reads = 0, n_ops = 0, n_units = 0, n_ck = 0;
    for (int32_t tid : tids) {
        ContigHost& h = c->contigs[tid];
        if (h.length <= 0) continue;
        gd::ContigDev d{};
        d.pos = h.pos; d.flag = h.flag; d.mapq = h.mapq; d.off = h.off; d.cigar = h.cigar;
        d.n_reads = (uint32_t)h.n_reads;
        d.n_ops = (uint32_t)h.n_ops;
        d.length = (int32_t)h.length;
        d.tile_beg = (int32_t)tile_beg;
        d.n_tiles = (int32_t)((h.length + T - 1) / T);
        d.base_off = base_off;
        d.win_off = win_off;
        d.tid = tid;
        d.unit_beg = (uint32_t)n_units;
        d.ck_off = (int64_t)n_ck;
        d.read_off = (int64_t)n_reads;
        if (c->use_v8) { d.desc = h.desc; d.cx_base = h.cxb; d.cx_cigar = h.cxc; }
        n_units += (h.n_reads + 63) / 64;
        n_ck += (h.n_ops >> 6) + h.n_reads + 1;   // gd_chunk.hpp: slots (off >> 6) + read
        h.base_off = base_off;
        h.win_off = win_off;
        h.n_win = (h.length + P.window_size - 1) / P.window_size;
        tile_beg += d.n_tiles;
        base_off += (int64_t)d.n_tiles * T;
        win_off += h.n_win;
        bases += h.length;
        n_reads += h.n_reads;
        n_ops += h.n_ops;
        c->h_ctgs.push_back(d);
        c->job_tids.push_back(tid);
    }
    if (tile_beg > 0x7fffffffLL) return fail(c, GD_E_RANGE, "too many tiles");
    if (n_units > 0xfffffff0ull) return fail(c, GD_E_RANGE, "too many records");
    c->n_tiles = tile_beg;
    c->n_win_total = win_off;
    c->n_bases = bases;
    c->bounds.clear();
    if (c->n_tiles == 0) { c->computed = true; return GD_OK; }

    // ---- allocations -------------------------------------------------------
    size_t cap;
    if (int r = ensure_dev(c, &c->d_ctgs, &c->cap_ctgs, c->h_ctgs.size())) return r;
    cap = c->cap_tiles;
    if ((size_t)c->n_tiles > c->cap_tiles) {
        size_t c1 = cap, c2 = cap, c3 = cap, c4 = cap;
        if (int r = ensure_dev(c, &c->d_tiles, &c1, (size_t)c->n_tiles)) return r;
        if (int r = ensure_dev(c, &c->d_tile_cnt, &c2, (size_t)c->n_tiles)) return r;
        if (int r = ensure_dev(c, &c->d_tile_off, &c3, (size_t)c->n_tiles)) return r;
        if (int r = ensure_dev(c, &c->d_super_cnt, &c4, (size_t)c->n_tiles / gd::SUPER + 1)) return r;
        c->cap_tiles = c1;
    }
    if (c->keep_perbase) {
        if (int r = ensure_dev(c, &c->d_perbase, &c->cap_perbase, (size_t)base_off)) return r;
    } else if (c->d_perbase) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, hipFree(c->d_perbase));        // windows-only: give the HBM back
        c->d_perbase = nullptr;
        c->cap_perbase = 0;
    }
    if ((size_t)win_off > c->cap_win || !c->d_wsum) {
        size_t c1 = c->cap_win, c2 = c->cap_win;
        if (int r = ensure_dev(c, &c->d_wsum, &c1, (size_t)std::max<int64_t>(win_off, 1))) return r;
        if (int r = ensure_dev(c, &c->d_wmin, &c2, (size_t)std::max<int64_t>(win_off, 1))) return r;
        c->cap_win = c1;
    }
    if (!c->d_chunks) {
        size_t want = std::max<size_t>(1u << 16, (size_t)c->n_tiles * 2);
        size_t c1 = 0, c2 = 0;
        if (int r = ensure_dev(c, &c->d_chunks, &c1, want)) return r;
        if (int r = ensure_dev(c, &c->d_ordered, &c2, want)) return r;
        c->cap_runs = c1;
    }
    HIPCHK(c, hipMemcpyAsync(c->d_ctgs, c->h_ctgs.data(), c->h_ctgs.size() * sizeof(gd::ContigDev),
                             hipMemcpyHostToDevice, c->stream));
    // records staged on the copy stream must have landed
    HIPCHK(c, hipEventRecord(c->copy_done, c->copy_stream));
    HIPCHK(c, hipStreamWaitEvent(c->stream, c->copy_done, 0));

    int reruns = 0;
    int used_lookback = c->lookback;
    bool used_scatter = false, used_chunk = false;
    for (;;) {
        gd::Job job{};
        job.ctgs = c->d_ctgs;
        job.n_ctgs = (int32_t)c->h_ctgs.size();
        job.n_tiles = (int32_t)c->n_tiles;
        job.tiles = c->d_tiles;
        job.perbase = c->d_perbase;
        job.win_sum = c->d_wsum;
        job.win_min = c->d_wmin;
        job.n_win_total = c->n_win_total;
        job.run_chunks = c->d_chunks;
        job.run_cap = (uint32_t)std::min<size_t>(c->cap_runs, 0xffffffffu);
        job.tile_cnt = c->d_tile_cnt;
        job.tile_off = c->d_tile_off;
        job.super_cnt = c->d_super_cnt;
        job.counters = c->d_counters;
        job.W = P.window_size;
        job.Q = P.min_mapq;
        job.mincov = P.min_cov;
        job.maxmean = P.max_mean_depth;
        job.flag_mask = P.flag_mask;
        job.lookback = c->lookback;
        job.step = derive_step(P);
        magic_u31((uint32_t)job.W, &job.w_magic, &job.w_shift);
        magic_u31(job.step > 0x7fffffffLL ? 0x7fffffffu : (uint32_t)job.step, &job.s_magic, &job.s_shift);

        job.n_units = (uint32_t)n_units;
        job.tile_status = c->d_status;

        // which device algorithm (include/goleft_depth.h GD_PATH_*)
        c->ran_sums_only = false;
        const bool scatter = c->path == GD_PATH_SCATTER;
        const bool chunk = c->path == GD_PATH_CHUNK ||
                           (c->path == GD_PATH_AUTO && (c->span_forces_long || n_ops > 6 * n_reads));
        const unsigned runs_grid = (unsigned)((c->n_tiles + gd::SUPER - 1) / gd::SUPER);
        if (scatter && !c->keep_perbase)
            return fail(c, GD_E_INVALID, "windows-only output (gd_set_outputs without GD_OUT_PERBASE) is not "
                                         "available on the scatter path");
        if (chunk) {
            if (int r = ensure_dev(c, &c->d_ck, &c->cap_ck, (size_t)n_ck)) return r;
            if (int r = ensure_dev(c, &c->d_rend, &c->cap_rend, (size_t)std::max<uint64_t>(n_reads, 1))) return r;
            job.ck = c->d_ck;
            job.rend = c->d_rend;
            job.lookback_dev = 1;
            HIPCHK(c, hipMemsetAsync(c->d_counters, 0, sizeof(gd::Counters), c->stream));
            if (c->profiling) HIPCHK(c, hipEventRecord(c->ev[0], c->stream));
            if (n_units) {
                const uint64_t groups = (n_units + 3) / 4;
                const unsigned grid = (unsigned)(((groups + 7) / 8) * 8);
                hipLaunchKernelGGL(gd::gd_ckpt_kernel, dim3(grid), dim3(256), 0, c->stream, job);
            }
            if (c->profiling) HIPCHK(c, hipEventRecord(c->ev[1], c->stream));
            switch (T) {
            case 8192: launch_prep<8192>(c, job); break;
            default: launch_prep<4096>(c, job); break;
            }
            if (c->profiling) HIPCHK(c, hipEventRecord(c->ev[2], c->stream));
            {
                const int key = T + c->tile_NT;
                switch (key) {
                case 4096 + 512: launch_ltile<4096, 512>(c, job); break;
                case 8192 + 512: launch_ltile<8192, 512>(c, job); break;
                case 8192 + 256: launch_ltile<8192, 256>(c, job); break;
                default: launch_ltile<4096, 256>(c, job); break;
                }
            }
            if (c->profiling) HIPCHK(c, hipEventRecord(c->ev[3], c->stream));
            hipLaunchKernelGGL(gd::gd_runs_order_kernel, dim3(runs_grid), dim3(gd::SUPER), 0,
                               c->stream, c->d_chunks, job.run_cap, c->d_tile_cnt, c->d_tile_off,
                               c->d_super_cnt, c->d_ordered, (int)c->n_tiles);
            if (c->profiling) HIPCHK(c, hipEventRecord(c->ev[4], c->stream));
        } else if (!scatter) {
            // sums-only: built for the default tile shape; W < 32 (more windows per tile than the
            // LDS accumulators hold) keeps the regular windows-only kernel
            c->ran_sums_only = c->sums_only && T == 4096 && c->tile_NT == 256 && P.window_size >= 32;
            if (c->profiling) HIPCHK(c, hipEventRecord(c->ev[0], c->stream));
            switch (T) {
            case 8192: launch_prep<8192>(c, job); break;
            default: launch_prep<4096>(c, job); break;
            }
            if (c->profiling) HIPCHK(c, hipEventRecord(c->ev[1], c->stream));
            {
                const int key = T + c->tile_NT;   // (T, NT) variants compiled below
                switch (key) {
                case 4096 + 512: launch_tile<4096, 512>(c, job); break;
                case 8192 + 512: launch_tile<8192, 512>(c, job); break;
                case 8192 + 256: launch_tile<8192, 256>(c, job); break;
                default: launch_tile<4096, 256>(c, job); break;
                }
            }
            if (c->profiling) HIPCHK(c, hipEventRecord(c->ev[2], c->stream));
            if (!c->ran_sums_only)                        // no class runs in sums-only mode
                hipLaunchKernelGGL(gd::gd_runs_order_kernel, dim3(runs_grid), dim3(gd::SUPER), 0,
                                   c->stream, c->d_chunks, job.run_cap, c->d_tile_cnt, c->d_tile_off,
                                   c->d_super_cnt, c->d_ordered, (int)c->n_tiles);
            if (c->profiling) HIPCHK(c, hipEventRecord(c->ev[3], c->stream));
        } else {
            if (int r = ensure_dev(c, &c->d_status, &c->cap_status, (size_t)c->n_tiles)) return r;
            job.tile_status = c->d_status;
            if (c->profiling) HIPCHK(c, hipEventRecord(c->ev[0], c->stream));
            // the per-base array doubles as the difference array: zero it, padding included
            HIPCHK(c, hipMemsetAsync(c->d_perbase, 0, (size_t)base_off * sizeof(int32_t), c->stream));
            {
                int64_t work = std::max<int64_t>(job.n_tiles, std::min<int64_t>(job.n_win_total, 1 << 22));
                hipLaunchKernelGGL(gd::gd_linit_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0,
                                   c->stream, job);
            }
            if (c->profiling) HIPCHK(c, hipEventRecord(c->ev[1], c->stream));
            if (n_units) {
                const uint64_t groups = (n_units + 3) / 4;
                const unsigned grid = (unsigned)(((groups + 7) / 8) * 8);
                hipLaunchKernelGGL(gd::gd_expand_scatter_kernel, dim3(grid), dim3(256), 0, c->stream, job);
            }
            if (c->profiling) HIPCHK(c, hipEventRecord(c->ev[2], c->stream));
            if (T == 8192)
                hipLaunchKernelGGL((gd::gd_scan_kernel<8192, 256>), dim3((unsigned)c->n_tiles), dim3(256), 0, c->stream, job);
            else
                hipLaunchKernelGGL((gd::gd_scan_kernel<4096, 256>), dim3((unsigned)c->n_tiles), dim3(256), 0, c->stream, job);
            if (c->profiling) HIPCHK(c, hipEventRecord(c->ev[3], c->stream));
            hipLaunchKernelGGL(gd::gd_runs_order_kernel, dim3(runs_grid), dim3(gd::SUPER), 0,
                               c->stream, c->d_chunks, job.run_cap, c->d_tile_cnt, c->d_tile_off,
                               c->d_super_cnt, c->d_ordered, (int)c->n_tiles);
            if (c->profiling) HIPCHK(c, hipEventRecord(c->ev[4], c->stream));
        }
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipMemcpyAsync(c->h_counters, c->d_counters, sizeof(gd::Counters), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        used_scatter = scatter;
        used_chunk = chunk;

        const gd::Counters k = *c->h_counters;
        if (chunk) {
            c->stats.max_span_seen = k.max_span > 0 ? k.max_span : 0;
            used_lookback = c->stats.max_span_seen;
        } else if (scatter) {
            if (k.pad1 != 0) return fail(c, GD_E_HIP, "scan look-back timed out (internal error)");
        } else {
            c->stats.max_span_seen = k.max_span > 0 ? k.max_span : 0;
            if (c->path == GD_PATH_AUTO && k.max_span > kAutoLongSpan) {
                // reads this long make the tile path re-examine too much: switch paths
                c->span_forces_long = true;
                ++reruns;
                continue;
            }
            if (k.max_span >= kMaxSpan)
                return fail(c, GD_E_RANGE, "a read spans %d reference bases (tile path limit %d; use GD_PATH_CHUNK)",
                            k.max_span, kMaxSpan - 1);
            if (k.max_span > c->lookback) {
                // a kept read spans more reference than the look-back: redo with the observed maximum
                c->lookback = (k.max_span + 63) & ~63;
                ++reruns;
                continue;
            }
            used_lookback = c->lookback;
            if (!c->lookback_pinned) {
                // the tile kernel reports the true maximum: a look-back far above it only
                // costs re-examined reads, so the next compute uses a tighter (still verified) one
                const int want = std::max(64, (k.max_span + 63) & ~63);
                if (want * 2 <= c->lookback) c->lookback = want;
            }
        }
        if ((size_t)k.run_cursor > c->cap_runs) {
            size_t want = (size_t)k.run_cursor + (size_t)k.run_cursor / 8 + 1024;
            size_t c1 = c->cap_runs, c2 = c->cap_runs;
            if (int r = ensure_dev(c, &c->d_chunks, &c1, want)) return r;
            if (int r = ensure_dev(c, &c->d_ordered, &c2, want)) return r;
            c->cap_runs = c1;
            ++reruns;
            continue;
        }
        // ---- ordered boundaries to the host (small) -------------------------
        c->bounds.resize(k.run_cursor);
        if (k.run_cursor)
            HIPCHK(c, hipMemcpy(c->bounds.data(), c->d_ordered, (size_t)k.run_cursor * sizeof(int2), hipMemcpyDeviceToHost));
        break;
    }
    if (c->profiling) {
        auto ms = [&](int a, int b, float* out) -> int {
            HIPCHK(c, hipEventElapsedTime(out, c->ev[a], c->ev[b]));
            return GD_OK;
        };
        if (used_chunk) {
            if (int r = ms(0, 1, &c->kernel_ms[GD_K_CKPT])) return r;
            if (int r = ms(1, 2, &c->kernel_ms[GD_K_PREP])) return r;
            if (int r = ms(2, 3, &c->kernel_ms[GD_K_TILE])) return r;
            if (int r = ms(3, 4, &c->kernel_ms[GD_K_RUNS])) return r;
        } else if (!used_scatter) {
            if (int r = ms(0, 1, &c->kernel_ms[GD_K_PREP])) return r;
            if (int r = ms(1, 2, &c->kernel_ms[GD_K_TILE])) return r;
            if (int r = ms(2, 3, &c->kernel_ms[GD_K_RUNS])) return r;
        } else {
            if (int r = ms(0, 1, &c->kernel_ms[GD_K_PREP])) return r;
            if (int r = ms(1, 2, &c->kernel_ms[GD_K_EXPAND])) return r;
            if (int r = ms(2, 3, &c->kernel_ms[GD_K_SCAN])) return r;
            if (int r = ms(3, 4, &c->kernel_ms[GD_K_RUNS])) return r;
        }
    }
    // slice the boundary list per contig
    {
        size_t i = 0;
        const size_t n = c->bounds.size();
        for (size_t j = 0; j < c->job_tids.size(); ++j) {
            ContigHost& h = c->contigs[c->job_tids[j]];
            h.run_beg = i;
            while (i < n && (size_t)(c->bounds[i].y >> 2) == j) ++i;
            h.run_end = i;
        }
        if (i != n) return fail(c, GD_E_HIP, "run boundaries out of order (internal error)");
    }
    c->stats.n_reads = n_reads;
    c->stats.n_ops = n_ops;
    c->stats.n_ref_bases = (uint64_t)bases;
    c->stats.n_windows = (uint64_t)c->n_win_total;
    c->stats.n_tiles = (uint64_t)c->n_tiles;
    c->stats.n_runs = c->bounds.size();
    c->stats.tile_positions = T;
    c->stats.lookback = used_lookback;
    c->stats.reruns = reruns;
    c->stats.path = used_chunk ? GD_PATH_CHUNK : used_scatter ? GD_PATH_SCATTER : GD_PATH_TILE;
    c->stats.reserved = 0;
    if (used_scatter) c->stats.max_span_seen = 0;   // not measured on this path
    c->computed = true;
    return GD_OK;
}

static int check_result_tid(gd_ctx* c, int32_t tid)
{
    if (!c->computed) return fail(c, GD_E_STATE, "no results: call gd_compute first");
    if (tid < 0 || (size_t)tid >= c->contigs.size()) return fail(c, GD_E_RANGE, "tid %d out of range", tid);
    if (c->contigs[tid].length > 0 && c->contigs[tid].base_off < 0)
        return fail(c, GD_E_RANGE, "contig %d was not selected in the last gd_compute", tid);
    return GD_OK;
}

int gd_perbase(gd_ctx* c, int32_t tid, int64_t start, int64_t end, int32_t* out)
{
    if (!c || !out) return GD_E_INVALID;
    if (int r = set_device(c)) return r;
    if (int r = check_result_tid(c, tid)) return r;
    const ContigHost& h = c->contigs[tid];
    if (start < 0 || end < start || end > h.length) return fail(c, GD_E_RANGE, "region [%lld,%lld) outside contig of length %lld", (long long)start, (long long)end, (long long)h.length);
    if (end == start) return GD_OK;
    if (!c->d_perbase) return fail(c, GD_E_STATE, "the per-base vector was not kept (gd_set_outputs)");
    HIPCHK(c, hipMemcpy(out, c->d_perbase + h.base_off + start, (size_t)(end - start) * sizeof(int32_t), hipMemcpyDeviceToHost));
    return GD_OK;
}

int gd_windows(gd_ctx* c, int32_t tid, int64_t* sums, int32_t* mins, size_t cap, size_t* n)
{
    if (!c || !n) return GD_E_INVALID;
    if (int r = set_device(c)) return r;
    if (int r = check_result_tid(c, tid)) return r;
    const ContigHost& h = c->contigs[tid];
    *n = (size_t)h.n_win;
    if (h.n_win == 0) return GD_OK;
    if (cap < (size_t)h.n_win || !sums) return fail(c, GD_E_CAPACITY, "need room for %lld windows", (long long)h.n_win);
    if (mins && c->ran_sums_only) return fail(c, GD_E_STATE, "window minima were not produced (GD_OUT_SUMS_ONLY)");
    HIPCHK(c, hipMemcpy(sums, c->d_wsum + h.win_off, (size_t)h.n_win * sizeof(int64_t), hipMemcpyDeviceToHost));
    if (mins)
        HIPCHK(c, hipMemcpy(mins, c->d_wmin + h.win_off, (size_t)h.n_win * sizeof(int32_t), hipMemcpyDeviceToHost));
    return GD_OK;
}

int gd_callable(gd_ctx* c, int32_t tid, gd_run* out, size_t cap, size_t* n)
{
    if (!c || !n) return GD_E_INVALID;
    if (int r = check_result_tid(c, tid)) return r;
    if (c->ran_sums_only) return fail(c, GD_E_STATE, "coverage-class runs were not produced (GD_OUT_SUMS_ONLY)");
    const ContigHost& h = c->contigs[tid];
    const size_t k = h.run_end - h.run_beg;
    *n = k;
    if (k == 0) return GD_OK;
    if (cap < k || !out) return fail(c, GD_E_CAPACITY, "need room for %zu runs", k);
    for (size_t i = 0; i < k; ++i) {
        const int2 b = c->bounds[h.run_beg + i];
        out[i].start = b.x;
        out[i].cls = b.y & 3;
        out[i].end = (i + 1 < k) ? c->bounds[h.run_beg + i + 1].x : (int32_t)h.length;
    }
    return GD_OK;
}

int gd_region_windows(gd_ctx* c, int32_t tid, int64_t start, int64_t end, int64_t* sums,
                      int32_t* mins, size_t cap, size_t* n)
{
    if (!c || !n) return GD_E_INVALID;
    if (int r = set_device(c)) return r;
    if (int r = check_result_tid(c, tid)) return r;
    if (start < 0 || end < start) return fail(c, GD_E_RANGE, "bad region");
    const ContigHost& h = c->contigs[tid];
    const int64_t W = c->params.window_size;
    if (end == start) { *n = 0; return GD_OK; }
    const int64_t first = start / W, last = (end - 1) / W;
    const size_t k = (size_t)(last - first + 1);
    *n = k;
    if (cap < k || !sums) return fail(c, GD_E_CAPACITY, "need room for %zu windows", k);
    if (h.length <= 0) { for (size_t i = 0; i < k; ++i) { sums[i] = 0; if (mins) mins[i] = 0; } return GD_OK; }
    if (!c->d_perbase) return fail(c, GD_E_STATE, "the per-base vector was not kept (gd_set_outputs)");
    int64_t* d_s = nullptr;
    int32_t* d_m = nullptr;
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&d_s), k * sizeof(int64_t)));
    if (hipMalloc(reinterpret_cast<void**>(&d_m), k * sizeof(int32_t)) != hipSuccess) { (void)hipFree(d_s); return fail(c, GD_E_NOMEM, "hipMalloc"); }
    hipLaunchKernelGGL(gd::gd_region_windows_kernel, dim3((unsigned)k), dim3(256), 0, c->stream,
                       c->d_perbase + h.base_off, h.length, start, end, (int32_t)W, first, d_s, d_m);
    hipError_t e1 = hipMemcpyAsync(sums, d_s, k * sizeof(int64_t), hipMemcpyDeviceToHost, c->stream);
    hipError_t e2 = mins ? hipMemcpyAsync(mins, d_m, k * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream) : hipSuccess;
    hipError_t e3 = hipStreamSynchronize(c->stream);
    (void)hipFree(d_s);
    (void)hipFree(d_m);
    if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) return fail(c, GD_E_HIP, "region window reduction failed");
    return GD_OK;
}

int gd_region_callable(gd_ctx* c, int32_t tid, int64_t start, int64_t end, gd_run* out, size_t cap, size_t* n)
{
    if (!c || !n) return GD_E_INVALID;
    if (int r = set_device(c)) return r;
    if (int r = check_result_tid(c, tid)) return r;
    if (start < 0 || end < start || end > 0x7fffffffLL) return fail(c, GD_E_RANGE, "bad region");
    const ContigHost& h = c->contigs[tid];
    if (end == start) { *n = 0; return GD_OK; }
    if (h.length <= 0) {
        *n = 1;
        if (cap < 1 || !out) return fail(c, GD_E_CAPACITY, "need room for 1 run");
        out[0].start = (int32_t)start; out[0].end = (int32_t)end; out[0].cls = GD_NO_COVERAGE;
        return GD_OK;
    }
    if (!c->d_perbase) return fail(c, GD_E_STATE, "the per-base vector was not kept (gd_set_outputs)");
    size_t bcap = std::max<size_t>(cap, 1024);
    for (;;) {
        int2* d_b = nullptr;
        HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&d_b), bcap * sizeof(int2)));
        (void)hipMemsetAsync(c->d_region_cursor, 0, sizeof(uint32_t), c->stream);
        int64_t len = end - start;
        unsigned blocks = (unsigned)std::min<int64_t>((len + 255) / 256, 4096);
        hipLaunchKernelGGL(gd::gd_region_bounds_kernel, dim3(blocks), dim3(256), 0, c->stream,
                           c->d_perbase + h.base_off, h.length, start, end, c->params.min_cov,
                           c->params.max_mean_depth, d_b, (uint32_t)std::min<size_t>(bcap, 0xffffffffu),
                           c->d_region_cursor);
        uint32_t cnt = 0;
        hipError_t e1 = hipMemcpyAsync(&cnt, c->d_region_cursor, sizeof cnt, hipMemcpyDeviceToHost, c->stream);
        hipError_t e2 = hipStreamSynchronize(c->stream);
        if (e1 != hipSuccess || e2 != hipSuccess) { (void)hipFree(d_b); return fail(c, GD_E_HIP, "region class reduction failed"); }
        if (cnt > bcap) { (void)hipFree(d_b); bcap = cnt; continue; }
        std::vector<int2> b(cnt);
        hipError_t e3 = cnt ? hipMemcpy(b.data(), d_b, cnt * sizeof(int2), hipMemcpyDeviceToHost) : hipSuccess;
        (void)hipFree(d_b);
        if (e3 != hipSuccess) return fail(c, GD_E_HIP, "region class copy failed");
        std::sort(b.begin(), b.end(), [](const int2& a, const int2& z) { return a.x < z.x; });
        *n = cnt;
        if (cap < cnt || !out) return fail(c, GD_E_CAPACITY, "need room for %u runs", cnt);
        for (uint32_t i = 0; i < cnt; ++i) {
            out[i].start = b[i].x;
            out[i].cls = b[i].y & 3;
            out[i].end = (i + 1 < cnt) ? b[i + 1].x : (int32_t)end;
        }
        return GD_OK;
    }
}

// Builds the matrix in the context's scratch buffer; *d_cells points at [rows][n_samples].
static int depthwed_on_device(gd_ctx* c, int n_samples, int n_ctg, const int32_t* tids, int64_t size,
                              int32_t* row_ctg, int64_t* row_start, int64_t* row_end, size_t cap_rows,
                              bool need_cap, size_t* n_rows, int64_t** d_cells)
{
    if (!c || !n_rows || n_samples < 1 || n_ctg < 1 || !tids || size < 1) return GD_E_INVALID;
    if (int r = set_device(c)) return r;
    if (!c->computed) return fail(c, GD_E_STATE, "no results: call gd_compute first");
    const int64_t W = c->params.window_size;
    const int64_t group = (size + W - 1) / W;      // depthwed.go:126: rows until span >= size
    std::vector<int64_t> off((size_t)n_samples * n_ctg), nwin(n_ctg), clen(n_ctg), row_beg(n_ctg + 1);
    int64_t rows = 0;
    for (int j = 0; j < n_ctg; ++j) {
        for (int s = 0; s < n_samples; ++s) {
            const int32_t tid = tids[(size_t)s * n_ctg + j];
            if (int r = check_result_tid(c, tid)) return r;
            const ContigHost& h = c->contigs[tid];
            if (s == 0) { clen[j] = h.length; nwin[j] = h.n_win; }
            else if (h.length != clen[j])
                return fail(c, GD_E_INVALID, "sample %d contig %d: length %lld differs from sample 0 (%lld)",
                            s, j, (long long)h.length, (long long)clen[j]);
            off[(size_t)s * n_ctg + j] = h.win_off < 0 ? 0 : h.win_off;
        }
        row_beg[j] = rows;
        rows += (nwin[j] + group - 1) / group;
    }
    row_beg[n_ctg] = rows;
    *n_rows = (size_t)rows;
    *d_cells = nullptr;
    if (rows == 0) return GD_OK;
    if (need_cap && cap_rows < (size_t)rows) return fail(c, GD_E_CAPACITY, "need room for %lld rows", (long long)rows);
    for (int j = 0; j < n_ctg; ++j)
        for (int64_t r = 0; r < row_beg[j + 1] - row_beg[j]; ++r) {
            const int64_t k = row_beg[j] + r;
            if ((size_t)k >= cap_rows) break;
            if (row_ctg) row_ctg[k] = j;
            if (row_start) row_start[k] = r * group * W;
            if (row_end) row_end[k] = std::min<int64_t>((r + 1) * group * W, clen[j]);
        }
    // small tables + the matrix live in one context-owned allocation
    const size_t n_tab = off.size() + nwin.size() + clen.size() + row_beg.size();
    const size_t n_cells = (size_t)rows * (size_t)n_samples;
    if (int r = ensure_dev(c, &c->d_wed, &c->cap_wed, n_tab + n_cells)) return r;
    int64_t* d = c->d_wed;
    std::vector<int64_t> tab;
    tab.reserve(n_tab);
    tab.insert(tab.end(), off.begin(), off.end());
    tab.insert(tab.end(), nwin.begin(), nwin.end());
    tab.insert(tab.end(), clen.begin(), clen.end());
    tab.insert(tab.end(), row_beg.begin(), row_beg.end());
    gd::WedJob j{};
    j.win_sum = c->d_wsum;
    j.off = d; j.nwin = d + off.size(); j.clen = j.nwin + nwin.size(); j.row_beg = j.clen + clen.size();
    j.cells = d + n_tab;
    j.n_samples = n_samples; j.n_ctg = n_ctg; j.n_rows = rows; j.W = W; j.group = group;
    HIPCHK(c, hipMemcpyAsync(d, tab.data(), n_tab * sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(gd::gd_depthwed_kernel, dim3((unsigned)((n_cells + 255) / 256)), dim3(256), 0, c->stream, j);
    HIPCHK(c, hipStreamSynchronize(c->stream));       // `tab` must outlive the copy
    *d_cells = j.cells;
    return GD_OK;
}

int gd_depthwed(gd_ctx* c, int n_samples, int n_ctg, const int32_t* tids, int64_t size,
                int64_t* cells, int32_t* row_ctg, int64_t* row_start, int64_t* row_end,
                size_t cap_rows, size_t* n_rows)
{
    int64_t* d_cells = nullptr;
    if (n_rows && cap_rows && !cells) return GD_E_INVALID;
    if (int r = depthwed_on_device(c, n_samples, n_ctg, tids, size, row_ctg, row_start, row_end, cap_rows, true,
                                   n_rows, &d_cells))
        return r;
    if (!d_cells) return GD_OK;
    HIPCHK(c, hipMemcpy(cells, d_cells, *n_rows * (size_t)n_samples * sizeof(int64_t), hipMemcpyDeviceToHost));
    return GD_OK;
}

int gd_depthwed_device(gd_ctx* c, int n_samples, int n_ctg, const int32_t* tids, int64_t size,
                       const int64_t** d_cells, size_t* n_rows)
{
    if (!d_cells) return GD_E_INVALID;
    int64_t* p = nullptr;
    const int r = depthwed_on_device(c, n_samples, n_ctg, tids, size, nullptr, nullptr, nullptr, 0, false, n_rows, &p);
    *d_cells = p;
    return r;
}

int gd_seq_load(gd_ctx* c, const uint8_t* seq, int64_t len)
{
    if (!c || len < 0 || (len > 0 && !seq)) return GD_E_INVALID;
    if (len >= 0x7fffffffLL - 16) return fail(c, GD_E_RANGE, "sequence of %lld bases (contigs are < 2^31)", (long long)len);
    if (int r = set_device(c)) return r;
    const size_t padded = (((size_t)len + 3) & ~(size_t)3) + 8;
    if (int r = ensure_dev(c, &c->d_seq, &c->cap_seq, padded)) return r;
    if (len) HIPCHK(c, hipMemcpyAsync(c->d_seq, seq, (size_t)len, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_seq + len, 0, padded - (size_t)len, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));          // the caller's buffer is not retained
    c->seq_len = len;
    c->seq_padded = (uint32_t)padded;
    return GD_OK;
}

int gd_seq_stats(gd_ctx* c, size_t n_windows, const int64_t* start, const int64_t* end,
                 uint32_t* n_gc, uint32_t* n_cpg, uint32_t* n_masked)
{
    if (!c || (n_windows && (!start || !end || !n_gc || !n_cpg || !n_masked))) return GD_E_INVALID;
    if (int r = set_device(c)) return r;
    if (c->seq_len < 0) return fail(c, GD_E_STATE, "gd_seq_load has not been called");
    if (n_windows == 0) return GD_OK;
    if (n_windows > 0x7fffffffull * 4) return fail(c, GD_E_RANGE, "too many windows");
    // one scratch allocation: starts, ends (int64), then the three count arrays (uint32)
    const size_t bytes = n_windows * (2 * sizeof(int64_t) + 3 * sizeof(uint32_t));
    uint8_t* d = nullptr;
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&d), bytes));
    int64_t* d_s = reinterpret_cast<int64_t*>(d);
    int64_t* d_e = d_s + n_windows;
    uint32_t* d_c = reinterpret_cast<uint32_t*>(d_e + n_windows);
    gd::SeqStatsJob j{};
    j.seq = c->d_seq; j.len = c->seq_len; j.padded = c->seq_padded;
    j.win_start = d_s; j.win_end = d_e;
    j.gc = d_c; j.cpg = d_c + n_windows; j.masked = d_c + 2 * n_windows;
    j.n_win = (int64_t)n_windows;
    hipError_t e1 = hipMemcpyAsync(d_s, start, n_windows * sizeof(int64_t), hipMemcpyHostToDevice, c->stream);
    hipError_t e2 = hipMemcpyAsync(d_e, end, n_windows * sizeof(int64_t), hipMemcpyHostToDevice, c->stream);
    if (c->profiling) (void)hipEventRecord(c->ev[0], c->stream);
    hipLaunchKernelGGL(gd::gd_seq_stats_kernel, dim3((unsigned)((n_windows + 3) / 4)), dim3(256), 0, c->stream, j);
    if (c->profiling) (void)hipEventRecord(c->ev[1], c->stream);
    hipError_t e3 = hipMemcpyAsync(n_gc, j.gc, n_windows * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream);
    hipError_t e4 = hipMemcpyAsync(n_cpg, j.cpg, n_windows * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream);
    hipError_t e5 = hipMemcpyAsync(n_masked, j.masked, n_windows * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream);
    hipError_t e6 = hipStreamSynchronize(c->stream);
    if (c->profiling && e6 == hipSuccess) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, c->ev[0], c->ev[1]) == hipSuccess) c->kernel_ms[GD_K_SEQSTATS] = ms;
    }
    (void)hipFree(d);
    if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess || e4 != hipSuccess || e5 != hipSuccess ||
        e6 != hipSuccess)
        return fail(c, GD_E_HIP, "sequence statistics kernel failed");
    return GD_OK;
}

// the per-base vectors of n_samples equally long contigs of the last gd_compute
static int md_sample_ptrs(gd_ctx* c, int n_samples, const int32_t* tids, std::vector<const int32_t*>* ptrs,
                          int64_t* len)
{
    if (!c->computed) return fail(c, GD_E_STATE, "no results: call gd_compute first");
    if (!c->d_perbase) return fail(c, GD_E_STATE, "the per-base vector was not kept (gd_set_outputs)");
    ptrs->resize((size_t)n_samples);
    for (int s = 0; s < n_samples; ++s) {
        if (int r = check_result_tid(c, tids[s])) return r;
        const ContigHost& h = c->contigs[tids[s]];
        if (s == 0) *len = h.length;
        else if (h.length != *len)
            return fail(c, GD_E_INVALID, "sample %d: contig length %lld differs from sample 0 (%lld)", s,
                        (long long)h.length, (long long)*len);
        (*ptrs)[(size_t)s] = c->d_perbase + h.base_off;
    }
    return GD_OK;
}

int gd_md_flags(gd_ctx* c, int n_samples, const int32_t* tids, int32_t min_cov, int32_t min_samples,
                uint32_t* any_bits, uint32_t* suf_bits, size_t n_words)
{
    if (!c || n_samples < 1 || !tids) return GD_E_INVALID;
    if (int r = set_device(c)) return r;
    std::vector<const int32_t*> ptrs;
    int64_t len = 0;
    if (int r = md_sample_ptrs(c, n_samples, tids, &ptrs, &len)) return r;
    const size_t need = (size_t)((len + 31) / 32);
    if (n_words < need || ((!any_bits || !suf_bits) && need))
        return fail(c, GD_E_CAPACITY, "need room for %zu bitmap words", need);
    c->md_len = -1;
    if (need == 0) { c->md_len = 0; c->md_tids.assign(tids, tids + n_samples); return GD_OK; }
    if (int r = ensure_dev(c, &c->d_md_bits, &c->cap_md, 2 * need)) return r;
    const int32_t** d_ptrs = nullptr;
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&d_ptrs), ptrs.size() * sizeof(ptrs[0])));
    gd::MdFlagsJob j{};
    j.depth = d_ptrs; j.n_samples = n_samples; j.len = len; j.min_cov = min_cov; j.min_samples = min_samples;
    j.any_bits = c->d_md_bits; j.suf_bits = c->d_md_bits + need;
    hipError_t e1 = hipMemcpyAsync(d_ptrs, ptrs.data(), ptrs.size() * sizeof(ptrs[0]), hipMemcpyHostToDevice, c->stream);
    if (c->profiling) (void)hipEventRecord(c->ev[0], c->stream);
    hipLaunchKernelGGL(gd::gd_md_flags_kernel, dim3((unsigned)((len + 255) / 256)), dim3(256), 0, c->stream, j);
    if (c->profiling) (void)hipEventRecord(c->ev[1], c->stream);
    hipError_t e2 = hipMemcpyAsync(any_bits, j.any_bits, need * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream);
    hipError_t e3 = hipMemcpyAsync(suf_bits, j.suf_bits, need * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream);
    hipError_t e4 = hipStreamSynchronize(c->stream);
    if (c->profiling && e4 == hipSuccess) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, c->ev[0], c->ev[1]) == hipSuccess) c->kernel_ms[GD_K_MDFLAGS] = ms;
    }
    (void)hipFree(d_ptrs);
    if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess || e4 != hipSuccess)
        return fail(c, GD_E_HIP, "multidepth flag kernel failed");
    c->md_len = len;
    c->md_tids.assign(tids, tids + n_samples);
    return GD_OK;
}

int gd_md_sums(gd_ctx* c, size_t n_blocks, const int64_t* start, const int64_t* end, double* sums)
{
    if (!c || (n_blocks && (!start || !end || !sums))) return GD_E_INVALID;
    if (int r = set_device(c)) return r;
    if (c->md_len < 0) return fail(c, GD_E_STATE, "gd_md_flags has not been called");
    if (n_blocks == 0) return GD_OK;
    const int n_samples = (int)c->md_tids.size();
    std::vector<const int32_t*> ptrs;
    int64_t len = 0;
    if (int r = md_sample_ptrs(c, n_samples, c->md_tids.data(), &ptrs, &len)) return r;
    if (len != c->md_len) return fail(c, GD_E_STATE, "results changed since gd_md_flags");
    for (size_t b = 0; b < n_blocks; ++b)
        if (start[b] < 0 || end[b] < start[b] || end[b] > len)
            return fail(c, GD_E_RANGE, "block %zu [%lld, %lld) outside the contig", b, (long long)start[b], (long long)end[b]);
    const size_t n_cells = n_blocks * (size_t)n_samples;
    const size_t bytes = ptrs.size() * sizeof(ptrs[0]) + 2 * n_blocks * sizeof(int64_t) + n_cells * sizeof(double);
    uint8_t* d = nullptr;
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&d), bytes));
    const int32_t** d_ptrs = reinterpret_cast<const int32_t**>(d);
    int64_t* d_s = reinterpret_cast<int64_t*>(d + ptrs.size() * sizeof(ptrs[0]));
    int64_t* d_e = d_s + n_blocks;
    double* d_sums = reinterpret_cast<double*>(d_e + n_blocks);
    gd::MdSumsJob j{};
    j.depth = d_ptrs; j.suf_bits = c->d_md_bits + (size_t)((len + 31) / 32);
    j.start = d_s; j.end = d_e; j.sums = d_sums; j.n_blocks = (int64_t)n_blocks; j.n_samples = n_samples;
    hipError_t e1 = hipMemcpyAsync(d_ptrs, ptrs.data(), ptrs.size() * sizeof(ptrs[0]), hipMemcpyHostToDevice, c->stream);
    hipError_t e2 = hipMemcpyAsync(d_s, start, n_blocks * sizeof(int64_t), hipMemcpyHostToDevice, c->stream);
    hipError_t e3 = hipMemcpyAsync(d_e, end, n_blocks * sizeof(int64_t), hipMemcpyHostToDevice, c->stream);
    hipLaunchKernelGGL(gd::gd_md_sums_kernel, dim3((unsigned)((n_cells + 63) / 64)), dim3(64), 0, c->stream, j);
    hipError_t e4 = hipMemcpyAsync(sums, d_sums, n_cells * sizeof(double), hipMemcpyDeviceToHost, c->stream);
    hipError_t e5 = hipStreamSynchronize(c->stream);
    (void)hipFree(d);
    if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess || e4 != hipSuccess || e5 != hipSuccess)
        return fail(c, GD_E_HIP, "multidepth block-sum kernel failed");
    return GD_OK;
}

int gd_inflate_bgzf(gd_ctx* c, const uint8_t* data, size_t n_bytes, size_t n_members, const uint64_t* in_off,
                    const uint32_t* in_len, const uint64_t* out_off, const uint32_t* out_len, const uint32_t* crc,
                    uint8_t* out, size_t out_bytes, uint32_t* status)
{
    if (!c || !data || !in_off || !in_len || !out_off || !out_len || !out || !status) return GD_E_INVALID;
    if (int r = set_device(c)) return r;
    if (n_members == 0) return GD_OK;
    if (n_members > 0xffffffffull) return fail(c, GD_E_RANGE, "too many BGZF members");
    for (size_t m = 0; m < n_members; ++m)
        if (in_off[m] + in_len[m] > n_bytes || out_off[m] + out_len[m] > out_bytes)
            return fail(c, GD_E_RANGE, "BGZF member %zu outside the given buffers", m);
    uint8_t *d_in = nullptr, *d_out = nullptr, *d_tab = nullptr;
    const size_t tab_bytes = n_members * (2 * sizeof(uint64_t) + 4 * sizeof(uint32_t));
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&d_in), n_bytes);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d_out), out_bytes ? out_bytes : 1);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d_tab), tab_bytes);
    if (e != hipSuccess) {
        if (d_in) (void)hipFree(d_in);
        if (d_out) (void)hipFree(d_out);
        return fail(c, GD_E_NOMEM, "device allocation for BGZF inflate failed");
    }
    uint64_t* t_in_off = reinterpret_cast<uint64_t*>(d_tab);
    uint64_t* t_out_off = t_in_off + n_members;
    uint32_t* t_in_len = reinterpret_cast<uint32_t*>(t_out_off + n_members);
    uint32_t* t_out_len = t_in_len + n_members;
    uint32_t* t_status = t_out_len + n_members;
    uint32_t* t_crc = t_status + n_members;
    if (crc && hipMemcpyAsync(t_crc, crc, n_members * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream) != hipSuccess) crc = nullptr;
    hipError_t e1 = hipMemcpyAsync(d_in, data, n_bytes, hipMemcpyHostToDevice, c->stream);
    hipError_t e2 = hipMemcpyAsync(t_in_off, in_off, n_members * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream);
    hipError_t e3 = hipMemcpyAsync(t_out_off, out_off, n_members * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream);
    hipError_t e4 = hipMemcpyAsync(t_in_len, in_len, n_members * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream);
    hipError_t e5 = hipMemcpyAsync(t_out_len, out_len, n_members * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream);
    gd::InflateJob j{};
    j.comp = d_in; j.in_off = t_in_off; j.in_len = t_in_len; j.out_off = t_out_off; j.out_len = t_out_len;
    j.crc = crc ? t_crc : nullptr;
    j.out = d_out; j.status = t_status; j.n = (uint32_t)n_members;
    if (c->profiling) (void)hipEventRecord(c->ev[0], c->stream);
    hipLaunchKernelGGL(gd::gd_inflate_kernel, dim3((unsigned)((n_members + gd::INF_LANES - 1) / gd::INF_LANES)),
                       dim3(gd::INF_LANES), 0, c->stream, j);
    if (c->profiling) (void)hipEventRecord(c->ev[1], c->stream);
    hipError_t e6 = hipMemcpyAsync(out, d_out, out_bytes, hipMemcpyDeviceToHost, c->stream);
    hipError_t e7 = hipMemcpyAsync(status, t_status, n_members * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream);
    hipError_t e8 = hipStreamSynchronize(c->stream);
    if (c->profiling && e8 == hipSuccess) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, c->ev[0], c->ev[1]) == hipSuccess) c->kernel_ms[GD_K_INFLATE] = ms;
    }
    (void)hipFree(d_in); (void)hipFree(d_out); (void)hipFree(d_tab);
    for (hipError_t x : {e1, e2, e3, e4, e5, e6, e7, e8})
        if (x != hipSuccess) return fail(c, GD_E_HIP, "BGZF inflate kernel failed: %s", hipGetErrorString(x));
    return GD_OK;
}

int gd_host_alloc(gd_ctx* c, size_t bytes, void** out)
{
    if (!c || !out) return GD_E_INVALID;
    if (int r = set_device(c)) return r;
    *out = nullptr;
    if (hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess)
        return fail(c, GD_E_NOMEM, "cannot page-lock %zu bytes", bytes);
    return GD_OK;
}

int gd_host_free(gd_ctx* c, void* p)
{
    if (!c) return GD_E_INVALID;
    if (int r = set_device(c)) return r;
    if (p) HIPCHK(c, hipHostFree(p));
    return GD_OK;
}

// ---- device BAM read: begin (member tables, allocations) / feed (bytes) / finish (records) ----
int gd_ingest_abort(gd_ctx* c)
{
    if (!c) return GD_E_INVALID;
    if (c->ing_n == 0) return GD_OK;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->copy_stream);
    (void)hipStreamSynchronize(c->stream);
    for (hipStream_t s : c->ing_stream)
        if (s) (void)hipStreamSynchronize(s);
    for (int i = 0; i < c->ing_n; ++i) { delete c->ing_q[i]; c->ing_q[i] = nullptr; }
    c->ing_n = 0;
    return GD_OK;
}

// Drops the oldest pending range (its device work must be complete: the caller decoded it).
static void ingest_pop(gd_ctx* c)
{
    if (c->ing_n == 0) return;
    (void)hipStreamSynchronize(c->stream);               // the record walks read its inflated bytes
    delete c->ing_q[0];
    c->ing_q[0] = c->ing_q[1];
    c->ing_q[1] = nullptr;
    --c->ing_n;
}

int gd_ingest_release(gd_ctx* c)
{
    if (!c) return GD_E_INVALID;
    if (int r = set_device(c)) return r;
    if (c->ing_n == 0) return fail(c, GD_E_STATE, "no fed range is pending");
    IngestState* g = c->ing_q[0];
    if (g->next != g->nm) return fail(c, GD_E_STATE, "the oldest range is still being fed (gd_ingest_abort drops it)");
    for (hipEvent_t e : g->inf_done) HIPCHK(c, hipEventSynchronize(e));   // its inflate kernels write its buffers
    ingest_pop(c);
    return GD_OK;
}

int gd_ingest_begin(gd_ctx* c, uint64_t n_bytes, uint64_t base_coffset, size_t n_members, const uint64_t* member_off,
                    const uint32_t* member_size, const uint16_t* header_size, const uint32_t* isize, const uint32_t* crc)
{
    if (!c || n_members == 0 || !member_off || !member_size || !header_size || !isize || !crc) return GD_E_INVALID;
    if (int r = set_device(c)) return r;
    if (n_members > 0xfffffff0ull) return fail(c, GD_E_RANGE, "too many BGZF members");
    // a range that was not fed to its end is an abandoned read: start over.  A completely fed one may
    // wait for its decode while this one is fed.
    if (c->ing_n && c->ing_q[c->ing_n - 1]->next != c->ing_q[c->ing_n - 1]->nm) (void)gd_ingest_abort(c);
    if (c->ing_n == 2) return fail(c, GD_E_STATE, "two fed ranges are pending: decode and release the oldest first");
    IngestState* g = new (std::nothrow) IngestState();
    if (!g) return GD_E_NOMEM;
    c->ing_q[c->ing_n++] = g;
    auto bail = [&](int code, const char* msg) { (void)gd_ingest_abort(c); return fail(c, code, "%s", msg); };
    g->n_bytes = n_bytes;
    g->nm = n_members;
    g->m_coff.resize(n_members); g->m_end.resize(n_members);
    std::vector<uint64_t> in_off(n_members), out_off(n_members);
    std::vector<uint32_t> in_len(n_members);
    g->out_off.resize(n_members); g->out_len.assign(isize, isize + n_members);
    uint64_t total = 0;
    for (size_t m = 0; m < n_members; ++m) {
        if (member_size[m] < (uint32_t)header_size[m] + 8u || member_off[m] + member_size[m] > n_bytes ||
            (m && member_off[m] < member_off[m - 1] + member_size[m - 1]))
            return bail(GD_E_INVALID, "inconsistent BGZF member table");
        g->m_coff[m] = base_coffset + member_off[m];
        g->m_end[m] = member_off[m] + member_size[m];
        in_off[m] = member_off[m] + header_size[m];
        in_len[m] = member_size[m] - header_size[m] - 8u;
        out_off[m] = total;
        g->out_off[m] = total;
        total += isize[m];
    }
    g->total = total;
    const size_t tab_bytes = n_members * (2 * sizeof(uint64_t) + 4 * sizeof(uint32_t));
    g->bufs = c->ing_bufs[0].busy ? &c->ing_bufs[1] : &c->ing_bufs[0];
    g->bufs->busy = true;
    if (!IngestBufs::fit(&g->bufs->in, &g->bufs->cap_in, (size_t)n_bytes) ||
        !IngestBufs::fit(&g->bufs->out, &g->bufs->cap_out, (size_t)total) ||
        !IngestBufs::fit(&g->bufs->tab, &g->bufs->cap_tab, tab_bytes))
        return bail(GD_E_NOMEM, "device allocation for the BAM decode failed");
    g->d_in = static_cast<uint8_t*>(g->bufs->in);
    g->d_out = static_cast<uint8_t*>(g->bufs->out);
    for (int k = 0; k < 2; ++k)
        if ((!c->ing_stage[k] &&
             hipHostMalloc(reinterpret_cast<void**>(&c->ing_stage[k]), IngestState::kStage, hipHostMallocDefault) != hipSuccess) ||
            (!c->ing_staged[k] && hipEventCreateWithFlags(&c->ing_staged[k], hipEventDisableTiming) != hipSuccess))
            return bail(GD_E_NOMEM, "cannot allocate the page-locked staging buffers");
    for (hipStream_t& s : c->ing_stream)
        if (!s && hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return bail(GD_E_HIP, "cannot create a stream");
    g->t_in_off = static_cast<uint64_t*>(g->bufs->tab);
    g->t_out_off = g->t_in_off + n_members;
    g->t_in_len = reinterpret_cast<uint32_t*>(g->t_out_off + n_members);
    g->t_out_len = g->t_in_len + n_members;
    g->t_status = g->t_out_len + n_members;
    g->t_crc = g->t_status + n_members;
    // (pageable sources: each copy is complete on return)
    if (hipMemcpy(g->t_in_off, in_off.data(), n_members * sizeof(uint64_t), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(g->t_out_off, out_off.data(), n_members * sizeof(uint64_t), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(g->t_in_len, in_len.data(), n_members * sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(g->t_out_len, isize, n_members * sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(g->t_crc, crc, n_members * sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess)
        return bail(GD_E_HIP, "uploading the BGZF member table failed");
    return GD_OK;
}

int gd_ingest_feed(gd_ctx* c, const uint8_t* bytes, size_t n)
{
    if (!c || (n && !bytes)) return GD_E_INVALID;
    if (int r = set_device(c)) return r;
    IngestState* g = c->ing_n ? c->ing_q[c->ing_n - 1] : nullptr;
    if (!g) return fail(c, GD_E_STATE, "gd_ingest_begin has not been called");
    if (g->fed + n > g->n_bytes) return fail(c, GD_E_RANGE, "more bytes fed than announced");
    size_t done = 0;
    while (done < n) {
        const size_t piece = std::min(n - done, IngestState::kStage);
        const int k = c->ing_cur;
        if (c->ing_stage_used[k]) HIPCHK(c, hipEventSynchronize(c->ing_staged[k]));   // its previous H2D has left the buffer
        // the caller's pointer is not retained
        if (c->ing_copy_threads > 1 && piece >= (8u << 20)) {
            const int nt = c->ing_copy_threads;
            const size_t slice = ((piece / (size_t)nt) + 4095) & ~(size_t)4095;
            std::vector<std::thread> th;
            for (int t = 1; t < nt; ++t) {
                const size_t b = std::min(piece, slice * t), e = std::min(piece, slice * (t + 1));
                if (e > b) th.emplace_back([=]() { memcpy(c->ing_stage[k] + b, bytes + done + b, e - b); });
            }
            memcpy(c->ing_stage[k], bytes + done, std::min(piece, slice));
            for (auto& t : th) t.join();
        } else {
            memcpy(c->ing_stage[k], bytes + done, piece);
        }
        HIPCHK(c, hipMemcpyAsync(g->d_in + g->fed, c->ing_stage[k], piece, hipMemcpyHostToDevice, c->copy_stream));
        HIPCHK(c, hipEventRecord(c->ing_staged[k], c->copy_stream));
        c->ing_stage_used[k] = true;
        c->ing_cur ^= 1;
        g->fed += piece;
        done += piece;
        // inflate the members that are now completely on the device, behind the copy: a quarter of
        // the range at a time (or whatever is left once every byte is in)
        size_t last = g->next;
        while (last < g->nm && g->m_end[last] <= g->fed) ++last;
        const size_t quota = std::max<size_t>((g->nm + IngestState::kBatches - 1) / IngestState::kBatches, 1);
        if (last > g->next && (last - g->next >= quota || last == g->nm) ) {
            hipStream_t is = c->ing_stream[c->ing_launch_seq++ % 8u];
            ++g->n_launch;
            HIPCHK(c, hipStreamWaitEvent(is, c->ing_staged[k], 0));
            gd::InflateJob ij{};
            ij.comp = g->d_in;
            ij.in_off = g->t_in_off + g->next; ij.in_len = g->t_in_len + g->next;
            ij.out_off = g->t_out_off + g->next; ij.out_len = g->t_out_len + g->next;
            ij.crc = g->t_crc + g->next; ij.out = g->d_out; ij.status = g->t_status + g->next;
            ij.n = (uint32_t)(last - g->next);
            hipLaunchKernelGGL(gd::gd_inflate_kernel, dim3((unsigned)((ij.n + gd::INF_LANES - 1) / gd::INF_LANES)),
                               dim3(gd::INF_LANES), 0, is, ij);
            hipEvent_t done = nullptr;
            HIPCHK(c, hipEventCreateWithFlags(&done, hipEventDisableTiming));
            g->inf_done.push_back(done);
            HIPCHK(c, hipEventRecord(done, is));
            g->next = last;
        }
    }
    return GD_OK;
}

// One reference of the fed range -> contig tid.  release: the range is dropped afterwards (always on error).
static int ingest_decode(gd_ctx* c, int32_t tid, int32_t ref_id, const uint64_t* anchors, size_t n_anchors,
                         uint64_t* n_records, bool release)
{
    if (!c || !anchors || n_anchors == 0) return GD_E_INVALID;
    if (int r = set_device(c)) return r;
    IngestState* g = c->ing_n ? c->ing_q[0] : nullptr;                // the oldest pending range
    if (!g) return fail(c, GD_E_STATE, "gd_ingest_begin has not been called");
    struct Guard { gd_ctx* c; bool on; ~Guard() { if (on) (void)gd_ingest_abort(c); } } guard{c, true};   // everything pending is dropped on an error
    if (tid < 0 || (size_t)tid >= c->contigs.size()) return fail(c, GD_E_RANGE, "tid %d out of range", tid);
    if (n_anchors > 0xfffffff0ull) return fail(c, GD_E_RANGE, "too many anchors");
    if (g->next != g->nm) return fail(c, GD_E_STATE, "only %zu of %zu BGZF members were fed", g->next, g->nm);
    const size_t nm = g->nm;
    const uint64_t total = g->total;
    // ---- anchors (virtual offsets) -> byte offsets in the inflated range ----------------------
    std::vector<uint64_t> seg_beg(n_anchors), seg_end(n_anchors);
    for (size_t i = 0; i < n_anchors; ++i) {
        const uint64_t coff = anchors[i] >> 16, uoff = anchors[i] & 0xffffu;
        const size_t k = (size_t)(std::lower_bound(g->m_coff.begin(), g->m_coff.end(), coff) - g->m_coff.begin());
        if (k >= nm || g->m_coff[k] != coff || uoff > g->out_len[k])
            return fail(c, GD_E_INVALID, "anchor %zu (virtual offset %llu) is not inside a member of the range", i,
                        (unsigned long long)anchors[i]);
        seg_beg[i] = g->out_off[k] + uoff;
        if (i && seg_beg[i] <= seg_beg[i - 1]) return fail(c, GD_E_INVALID, "anchors must be strictly ascending");
        if (i) seg_end[i - 1] = seg_beg[i];
    }
    seg_end[n_anchors - 1] = total;                   // a record of another reference ends the last walk earlier
    std::vector<uint32_t> status;
    if (!g->inflated) {
        for (hipEvent_t e : g->inf_done) HIPCHK(c, hipEventSynchronize(e));      // every member of this range is inflated
        status.resize(nm);
        HIPCHK(c, hipMemcpyAsync(status.data(), g->t_status, nm * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    }

    // ---- count the records of every anchor segment ----------------------------------------------
    DevBuf d_seg;
    const size_t seg_bytes = n_anchors * 5 * sizeof(uint64_t) + n_anchors * 4 * sizeof(uint32_t);
    if (d_seg.alloc(seg_bytes) != hipSuccess) return fail(c, GD_E_NOMEM, "device allocation for the record walk failed");
    uint64_t* s_beg = d_seg.as<uint64_t>();
    uint64_t* s_end = s_beg + n_anchors;
    uint64_t* s_rbase = s_end + n_anchors;
    uint64_t* s_obase = s_rbase + n_anchors;
    uint64_t* s_nops = s_obase + n_anchors;
    uint32_t* s_nrec = reinterpret_cast<uint32_t*>(s_nops + n_anchors);
    int32_t* s_first = reinterpret_cast<int32_t*>(s_nrec + n_anchors);
    int32_t* s_last = s_first + n_anchors;
    uint32_t* s_flags = reinterpret_cast<uint32_t*>(s_last + n_anchors);
    HIPCHK(c, hipMemcpyAsync(s_beg, seg_beg.data(), n_anchors * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(s_end, seg_end.data(), n_anchors * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
    gd::BamSegJob bj{};
    bj.data = g->d_out; bj.n_bytes = total; bj.seg_beg = s_beg; bj.seg_end = s_end; bj.tid = ref_id;
    bj.n_seg = (uint32_t)n_anchors; bj.n_rec = s_nrec; bj.n_ops = s_nops; bj.first_pos = s_first; bj.last_pos = s_last;
    bj.flags = s_flags;
    const unsigned seg_grid = (unsigned)((n_anchors + 63) / 64);
    hipLaunchKernelGGL(gd::gd_bam_walk_kernel<false>, dim3(seg_grid), dim3(64), 0, c->stream, bj);
    std::vector<uint32_t> nrec(n_anchors), flags(n_anchors);
    std::vector<uint64_t> nops(n_anchors);
    std::vector<int32_t> firstp(n_anchors), lastp(n_anchors);
    HIPCHK(c, hipMemcpyAsync(nrec.data(), s_nrec, n_anchors * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(nops.data(), s_nops, n_anchors * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(firstp.data(), s_first, n_anchors * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(lastp.data(), s_last, n_anchors * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(flags.data(), s_flags, n_anchors * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (size_t m = 0; m < status.size(); ++m)
        if (status[m] != 0)
            return fail(c, GD_E_INVALID, "BGZF member at file offset %llu %s (decoder code %u)",
                        (unsigned long long)g->m_coff[m], status[m] == 18 ? "fails its CRC32" : "does not inflate", status[m]);
    g->inflated = true;
    std::vector<uint64_t> rbase(n_anchors), obase(n_anchors);
    uint64_t N = 0, M = 0;
    int32_t prev_last = -0x7fffffff;
    for (size_t i = 0; i < n_anchors; ++i) {
        if (flags[i] & 2u) return fail(c, GD_E_INVALID, "corrupt BAM record in anchor segment %zu", i);
        if (flags[i] & 4u) return fail(c, GD_E_INVALID, "anchor %zu is not a record start (stale or foreign index?)", i + 1);
        if ((flags[i] & 1u) || (nrec[i] && firstp[i] < prev_last))
            return fail(c, GD_E_UNSORTED, "contig %d: records not coordinate sorted (anchor segment %zu)", tid, i);
        if (nrec[i]) prev_last = lastp[i];
        rbase[i] = N;
        obase[i] = M;
        N += nrec[i];
        M += nops[i];
    }
    if (M > 0xffffffffull) return fail(c, GD_E_RANGE, "more than 2^32 CIGAR ops on contig %d", tid);
    if (N >= kMaxReadsPerContig) return fail(c, GD_E_RANGE, "more than 2^30 records on contig %d", tid);

    // ---- extract into the contig's SoA arrays ----------------------------------------------------
    HIPCHK(c, hipStreamSynchronize(c->copy_stream));
    ContigHost& h = c->contigs[tid];
    {
        const int64_t len = h.length;
        free_contig(h);
        h.length = len;
    }
    if (N) {
        size_t c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0;
        if (int r = ensure_dev(c, &h.pos, &c1, (size_t)N)) return r;
        if (int r = ensure_dev(c, &h.flag, &c2, (size_t)N)) return r;
        if (int r = ensure_dev(c, &h.mapq, &c3, (size_t)N)) return r;
        if (int r = ensure_dev(c, &h.off, &c4, (size_t)N + 1)) return r;
        if (int r = ensure_dev(c, &h.cigar, &c5, (size_t)std::max<uint64_t>(M, 1))) return r;
        h.cap_reads = (size_t)N;
        h.cap_ops = c5;
        HIPCHK(c, hipMemcpyAsync(s_rbase, rbase.data(), n_anchors * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(s_obase, obase.data(), n_anchors * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
        bj.rec_base = s_rbase; bj.op_base = s_obase;
        bj.pos = h.pos; bj.flag = h.flag; bj.mapq = h.mapq; bj.cigar_off = h.off; bj.cigar = h.cigar;
        hipLaunchKernelGGL(gd::gd_bam_walk_kernel<true>, dim3(seg_grid), dim3(64), 0, c->stream, bj);
        const uint32_t m32 = (uint32_t)M;
        HIPCHK(c, hipMemcpyAsync(h.off + N, &m32, sizeof m32, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    h.n_reads = (size_t)N;
    h.n_ops = (size_t)M;
    h.last_pos = prev_last;
    c->computed = false;
    if (n_records) *n_records = N;
    if (N && wants_pack(c, N, M))
        if (int r = pack_contig(c, h)) return r;
    guard.on = false;
    if (release) ingest_pop(c);
    return GD_OK;
}

int gd_ingest_decode(gd_ctx* c, int32_t tid, int32_t ref_id, const uint64_t* anchors, size_t n_anchors, uint64_t* n_records)
{
    return ingest_decode(c, tid, ref_id, anchors, n_anchors, n_records, false);
}

int gd_ingest_finish(gd_ctx* c, int32_t tid, int32_t ref_id, const uint64_t* anchors, size_t n_anchors, uint64_t* n_records)
{
    return ingest_decode(c, tid, ref_id, anchors, n_anchors, n_records, true);
}

// The BGZF members of a byte range (SAMv1 4.1: gzip header with a BC extra subfield).
int gd_bgzf_members(const uint8_t* data, size_t n_bytes, size_t cap, uint64_t* member_off, uint32_t* member_size,
                    uint16_t* header_size, uint32_t* isize, uint32_t* crc, size_t* n_members)
{
    if (!data || !n_members) return GD_E_INVALID;
    size_t nm = 0;
    for (size_t p = 0; p + 18 <= n_bytes;) {
        const uint8_t* h = data + p;
        if (h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4)) return GD_E_INVALID;
        const size_t xlen = (size_t)h[10] | ((size_t)h[11] << 8);
        if (p + 12 + xlen > n_bytes) break;
        size_t q = 12, bsize = 0;
        while (q + 4 <= 12 + xlen) {
            const size_t slen = (size_t)h[q + 2] | ((size_t)h[q + 3] << 8);
            if (h[q] == 66 && h[q + 1] == 67 && slen == 2 && q + 6 <= 12 + xlen)
                bsize = ((size_t)h[q + 4] | ((size_t)h[q + 5] << 8)) + 1;
            q += 4 + slen;
        }
        if (bsize < 12 + xlen + 8) return GD_E_INVALID;
        if (p + bsize > n_bytes) break;                     // a trailing partial member is ignored
        if (nm < cap) {
            if (member_off) member_off[nm] = p;
            if (member_size) member_size[nm] = (uint32_t)bsize;
            if (header_size) header_size[nm] = (uint16_t)(12 + xlen);
            if (isize) isize[nm] = (uint32_t)h[bsize - 4] | ((uint32_t)h[bsize - 3] << 8) | ((uint32_t)h[bsize - 2] << 16) |
                                   ((uint32_t)h[bsize - 1] << 24);
            if (crc) crc[nm] = (uint32_t)h[bsize - 8] | ((uint32_t)h[bsize - 7] << 8) | ((uint32_t)h[bsize - 6] << 16) |
                               ((uint32_t)h[bsize - 5] << 24);
        }
        ++nm;
        p += bsize;
    }
    *n_members = nm;
    return nm > cap ? GD_E_CAPACITY : GD_OK;
}

int gd_ingest_bgzf(gd_ctx* c, int32_t tid, int32_t ref_id, const uint8_t* data, size_t n_bytes, uint64_t base_coffset,
                   const uint64_t* anchors, size_t n_anchors, uint64_t* n_records)
{
    if (!c || !data || !anchors || n_anchors == 0) return GD_E_INVALID;
    (void)gd_ingest_abort(c);                              // one-shot form: nothing else may be pending
    size_t nm = 0;
    int rc = gd_bgzf_members(data, n_bytes, 0, nullptr, nullptr, nullptr, nullptr, nullptr, &nm);
    if (rc != GD_OK && rc != GD_E_CAPACITY) return fail(c, GD_E_INVALID, "the range does not start with a BGZF member");
    if (nm == 0) return fail(c, GD_E_INVALID, "no complete BGZF member in the range");
    std::vector<uint64_t> moff(nm);
    std::vector<uint32_t> msize(nm), misize(nm), mcrc(nm);
    std::vector<uint16_t> mhdr(nm);
    if (gd_bgzf_members(data, n_bytes, nm, moff.data(), msize.data(), mhdr.data(), misize.data(), mcrc.data(), &nm) != GD_OK)
        return fail(c, GD_E_INVALID, "corrupt BGZF member table");
    if (int r = gd_ingest_begin(c, n_bytes, base_coffset, nm, moff.data(), msize.data(), mhdr.data(), misize.data(), mcrc.data()))
        return r;
    if (int r = gd_ingest_feed(c, data, n_bytes)) { (void)gd_ingest_abort(c); return r; }
    return gd_ingest_finish(c, tid, ref_id, anchors, n_anchors, n_records);
}

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

int gd_get_stats(gd_ctx* c, gd_stats* out)
{
    if (!c || !out) return GD_E_INVALID;
    *out = c->stats;
    return GD_OK;
}

int gd_set_profiling(gd_ctx* c, int on)
{
    if (!c) return GD_E_INVALID;
    c->profiling = on != 0;
    c->kernel_ms[GD_K_PACK] = 0;                        // accumulates over the contigs packed from now on
    return GD_OK;
}

int gd_kernel_ms(gd_ctx* c, int id, float* ms)
{
    if (!c || !ms || id < 0 || id >= GD_K_COUNT) return GD_E_INVALID;
    *ms = c->kernel_ms[id];
    return GD_OK;
}

}  // extern "C"
