// gd_kernels.hpp -- hand-written CDNA4 (gfx950) kernels of the per-base depth engine.
//
// Replaces the arithmetic that the reference delegates to `samtools depth`
// (call site /root/reference/depth/depth.go:45) plus the per-line reductions of
// its `callback` closure (depth/depth.go:238-364: window mean :293-305, class
// run-length encoding :307-323, getCovClass :223-234).
//
// Data-parallel shape (integer, HBM-bound, no MFMA):
//   K0 gd_prep_kernel   one thread per LDS tile: contig lookup + two binary
//                       searches of the coordinate-sorted `pos` array (first
//                       read that can reach the tile, first read past it);
//                       also initialises the window accumulators.
//   K1 gd_tile_kernel   one workgroup per tile of T reference positions:
//                       (A) lanes walk CIGARs of the tile's reads, merge
//                           adjacent M/=/X ops into reference intervals, clip
//                           to the tile and ds_add +1/-1 into an int32 LDS
//                           difference array (order independent => bit exact);
//                       (B) each wave64 scans its quarter of the tile with a
//                           DPP wavefront scan, rows of 256 positions, and
//                           streams int32x4 per lane to HBM (1 KiB per
//                           wave-instruction, the only large HBM stream);
//                           fused in the same registers: per-window int64
//                           sum / int32 min (flushed with one wave reduction
//                           per window boundary) and the coverage-class
//                           boundary detection;
//                       (C) tiles that contain class boundaries compact them
//                           from an LDS bitmap into a global chunk.
//   K2 gd_runs_scan / gd_runs_gather   order the per-tile chunks.
//   K3 gd_region_* kernels             --bed mode reductions over the resident
//                                      per-base vector.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gd {

constexpr int WAVE = 64;
constexpr int SUPER = 1024;   // tiles per ordering group (one workgroup of gd_runs_order_kernel)

// One contig as the device sees it.
struct ContigDev {
    const int32_t*  pos;
    const uint16_t* flag;
    const uint8_t*  mapq;
    const uint32_t* off;      // n_reads+1 CSR offsets (relative to cigar)
    const uint32_t* cigar;
    uint32_t n_reads;
    uint32_t n_ops;
    int32_t  length;
    int32_t  tile_beg;        // first global tile id of this contig
    int32_t  n_tiles;
    int64_t  base_off;        // element offset of this contig in the per-base array
    int64_t  win_off;         // element offset in the window arrays
    int32_t  tid;             // reference id in the BAM header
    uint32_t unit_beg;        // first 64-read unit of this contig (scatter path)
    uint32_t grp_beg;         // first 4096-read unit of this contig (gd_sums_stream_kernel)
    // long-read path (gd_chunk.hpp), built when the records arrive:
    const uint4*    lrec;     // {pos, end, offset of the deletion list, offset of the tile index} per read
    const uint32_t* lfq;      // flag << 8 | MAPQ per read
    const uint2*    dl;       // deletion lists {start, length}
    const uint32_t* pck;      // tile indexes: deletions starting before every 4096-base boundary a read spans
    const uint32_t* ndel;     // deletions of every read (the tile kernel bisects the list of a read without an index)
    const uint32_t* rec;      // (unused since round 5: the record words of canonical records) always null
    const uint32_t* pidx;     // position index: first read with pos >= 64 k (gd_pidx_kernel, or gd_index_records_kernel as the
                              // records arrived); null: search `pos`
    uint32_t pidx_last;       // entries above this one read as n_reads (an index built block by block has no tail)
    uint32_t pad_;
};

// Everything a workgroup needs for its tile in ONE record (one scalar load
// burst, no dependent contig-table lookup).  Written by gd_prep_kernel.
struct __attribute__((aligned(16))) TileInfo {
    const int32_t*  pos;
    const uint16_t* flag;
    const uint8_t*  mapq;
    const uint32_t* off;
    const uint32_t* cigar;
    int64_t  base_off;        // per-base array offset of the contig
    int64_t  win_off;         // window array offset of the contig
    int32_t  length;          // contig length
    int32_t  ctg;             // index into the ContigDev table
    int32_t  t0;              // first reference position of the tile
    uint32_t lo, hi;          // read index range [lo,hi) that can touch the tile
    uint32_t n_reads;         // records of the contig (bounds for the vector loads)
    uint32_t n_ops;           // CIGAR ops of the contig
    uint32_t clo, chi;        // CIGAR op range [off[lo], off[hi]) of those reads
    int32_t  tile;            // global tile id (the slow list of a fast run is compacted)
    int32_t  pad_;
    const uint32_t* ndel;     // long-read path: deletions per read (ContigDev::ndel)
    const uint4*    lrec;     // long-read path (ContigDev::lrec, ::lfq, ::dl, ::pck)
    const uint32_t* lfq;
    const uint2*    dl;
    const uint32_t* pck;
};

// The same for gd_tile_fast_kernel: one record per ORDINARY tile (full, at most one batch of reads, ops
// fit the staging area), everything resolved -- pointers at the tile's first read / op, window and
// run-break state of each wave's quarter -- so the kernel derives nothing on the scalar unit.
// nrd == 0xffffffff: the tile is on the slow list (job.tiles[0 .. n_slow)) instead.
struct __attribute__((aligned(16))) TileFast {
    const int32_t*  pos;      // at read lo
    const uint32_t* rec;      // record words (flag | MAPQ | op count), at read lo; fast == 2: the CSR offsets, at read lo
    const uint32_t* cig;      // the ops as they arrived, at op clo
    const uint16_t* flag;     // fast == 2: at read lo
    const uint8_t*  mapq;     // fast == 2: at read lo
    int32_t*  out;            // per-base output at t0 (null: windows-only)
    int64_t*  wsum;           // the contig's window sums
    int32_t*  wmin;
    int32_t   t0;
    uint32_t  nrd;            // reads [lo, hi) that can touch the tile
    uint32_t  nst;            // their ops [clo, chi)
    uint32_t  clo;
    int32_t   ctg;
    int32_t   pad_[3];
    uint32_t  win0[4];        // per wave quarter (1024 positions): window of its first position,
    int32_t   wleft[4];       //   positions from there to the next window boundary (1..W; BIG: never),
    int32_t   sleft[4];       //   positions to the next forced run break (0: at the first position; BIG: never)
};
constexpr int FAST_BIG = 0x3fffffff;
constexpr int FAST_FAR = FAST_BIG - 65536;

// Device-side counters, read back once per gd_compute.
struct Counters {
    int32_t  max_span;        // largest reference span of a kept read
    uint32_t run_cursor;      // boundary entries allocated (may exceed capacity)
    uint32_t pad0;            // scatter path: tile ticket of gd_scan_kernel
    uint32_t pad1;            // scatter path: 1 if a look-back ever timed out
    uint32_t n_slow[2];       // fast run: tiles gd_prep_kernel put on the slow list.  Two counters, used
                              // alternately: the prep kernel of compute k counts in [k & 1] and zeroes the
                              // other one for compute k + 1 -- no memset launch between computes
    uint32_t pad2[2];
};

struct Job {
    const ContigDev* ctgs;
    int32_t   n_ctgs;
    int32_t   n_tiles;
    TileInfo* tiles;
    int32_t*  perbase;
    int64_t*  win_sum;
    int32_t*  win_min;
    int64_t   n_win_total;
    int2*     run_chunks;     // unordered per-tile chunks {pos, cls | ctg<<2}
    uint32_t  run_cap;
    uint32_t* tile_cnt;
    uint32_t* tile_off;
    uint32_t* super_cnt;      // boundary entries per group of SUPER consecutive tiles
    Counters* counters;
    int32_t   W;
    int32_t   Q;
    int32_t   mincov;
    int32_t   maxmean;
    uint32_t  flag_mask;
    int32_t   lookback;
    int32_t   reserved0;
    int64_t   step;
    uint32_t  n_units;        // scatter path: 64-read units over all contigs
    uint32_t  n_groups;       // sums-only stream: 4096-read units over all contigs
    unsigned long long* tile_status;   // scatter path: look-back status word per tile
    uint32_t  w_magic, w_shift;   // floor(x / W)    = (x * w_magic) >> w_shift for x < 2^31
    uint32_t  s_magic, s_shift;   // floor(x / step) likewise (step clamped to 2^31-1)
    uint32_t  fast;               // 1: ordinary tiles get a TileFast record, the rest go to the slow list;
                                  // 2: the same over the records as they arrived (gd_tile_fast_kernel<ST, true>)
    uint32_t  parity;             // which Counters::n_slow this compute uses
    TileFast* ftiles;             // n_tiles records (fast run)
};

__device__ __forceinline__ int cov_class(int d, int mincov, int maxmean)
{
    // depth/depth.go:223-234
    int c = 2;
    if (maxmean > 0 && d >= maxmean) c = 3;
    if (d < mincov) c = 1;
    if (d == 0) c = 0;
    return c;
}

__device__ __forceinline__ uint32_t lower_bound_i32(const int32_t* a, uint32_t n, int32_t key)
{
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        uint32_t mid = lo + ((hi - lo) >> 1);
        if (a[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// Two lower_bounds over the same sorted array in ONE latency chain: a 4-ary
// search (three probes per round) for each key, both advanced in the same loop
// iteration, so a round has six independent loads in flight and the chain is
// ~log4(n) rounds instead of 2*log2(n) dependent loads.  keyB >= keyA.
__device__ __forceinline__ void lower_bound_pair(const int32_t* a, uint32_t n, int32_t keyA,
                                                 int32_t keyB, uint32_t& outA, uint32_t& outB)
{
    uint32_t la = 0, ha = n, lb = 0, hb = n;
    if (n == 0) { outA = 0; outB = 0; return; }
    const uint32_t last = n - 1;
    while (ha > la || hb > lb) {
        const uint32_t wa = ha - la, wb = hb - lb;
        uint32_t a1 = la + (wa >> 2), a2 = la + (wa >> 1), a3 = la + wa - (wa >> 2) - (wa != 0);
        uint32_t b1 = lb + (wb >> 2), b2 = lb + (wb >> 1), b3 = lb + wb - (wb >> 2) - (wb != 0);
        a1 = a1 < last ? a1 : last; a2 = a2 < last ? a2 : last; a3 = a3 < last ? a3 : last;
        b1 = b1 < last ? b1 : last; b2 = b2 < last ? b2 : last; b3 = b3 < last ? b3 : last;
        const int32_t va1 = a[a1], va2 = a[a2], va3 = a[a3];
        const int32_t vb1 = a[b1], vb2 = a[b2], vb3 = a[b3];
        if (wa != 0) {
            if (va3 < keyA) la = a3 + 1;
            else if (va2 < keyA) { la = a2 + 1; ha = a3; }
            else if (va1 < keyA) { la = a1 + 1; ha = a2; }
            else ha = a1;
        }
        if (wb != 0) {
            if (vb3 < keyB) lb = b3 + 1;
            else if (vb2 < keyB) { lb = b2 + 1; hb = b3; }
            else if (vb1 < keyB) { lb = b1 + 1; hb = b2; }
            else hb = b1;
        }
    }
    outA = la;
    outB = lb;
}

// lower_bound(a[0..n), key) from a GUESS of where it lies: two probes 8192 elements either side of the guess
// bracket it (a 30x genome's cumulative read count wanders a few thousand reads around the interpolated one), 14
// bisection steps finish; a miss bisects the side the answer is on.  ONE such search per tile: for records without
// a position index (gd_index.hpp builds one as records arrive; without one the pair of plain searches every tile
// ran moved 3.8 GB of 64-byte sectors, most of them TLB misses -- 0.29 ms per genome.
__device__ __forceinline__ uint32_t lower_bound_hint(const int32_t* a, uint32_t n, int32_t key, uint32_t guess)
{
    if (n == 0u) return 0u;
    constexpr uint32_t R = 8192u;
    const uint32_t g = guess < n ? guess : n - 1u;
    const uint32_t pl = g > R ? g - R : 0u;
    const uint32_t ph = n - 1u - g > R ? g + R : n - 1u;
    const int32_t vl = a[pl], vh = a[ph];                  // two independent probes
    uint32_t lo = 0, hi = n;
    if (vl < key) {                                        // the answer is past pl ...
        lo = pl + 1u;
        if (vh >= key) hi = ph; else lo = ph + 1u;         // ... and at or before ph, or past it
    } else {
        hi = pl;
    }
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (a[mid] < key) lo = mid + 1u; else hi = mid;
    }
    return lo;
}

// ---------------------------------------------------------------------------
// K0: tile table + accumulator init
// ---------------------------------------------------------------------------
template <int T>
__global__ void gd_prep_kernel(Job job)
{
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for (int64_t w = gid; w < job.n_win_total; w += gsz) {
        job.win_sum[w] = 0;
        job.win_min[w] = 0x7fffffff;
    }
    for (int64_t g = gid; g < (job.n_tiles + SUPER - 1) / SUPER; g += gsz) job.super_cnt[g] = 0;
    const int32_t lookback = job.lookback;   // long-read path: the exact maximum span, known since the records arrived
    if (gid == 0) {
        job.counters->max_span = 0;
        job.counters->run_cursor = 0;
        job.counters->n_slow[job.parity ^ 1u] = 0;
    }
    if (gid >= job.n_tiles) return;
    const int t = (int)gid;
    // contig of this tile: last c with tile_beg[c] <= t
    int lo = 0, hi = job.n_ctgs;
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (job.ctgs[mid].tile_beg <= t) lo = mid; else hi = mid;
    }
    const ContigDev c = job.ctgs[lo];
    TileInfo ti;
    ti.pos = c.pos; ti.flag = c.flag; ti.mapq = c.mapq; ti.off = c.off; ti.cigar = c.cigar;
    ti.base_off = c.base_off; ti.win_off = c.win_off; ti.length = c.length;
    ti.n_reads = c.n_reads; ti.n_ops = c.n_ops;
    ti.ctg = lo;
    ti.tile = t;
    ti.pad_ = 0;
    ti.ndel = c.ndel;
    ti.lrec = c.lrec; ti.lfq = c.lfq; ti.dl = c.dl; ti.pck = c.pck;
    ti.t0 = (t - c.tile_beg) * T;
    int32_t tend = ti.t0 + T < c.length ? ti.t0 + T : c.length;
    int32_t from = ti.t0 > lookback ? ti.t0 - lookback : 0;
    if (c.pidx) {
        // the index answers both searches: the look-back start rounded DOWN to a multiple of 64 (a few more
        // reads examined, never fewer), the tile end exactly (T is a multiple of 64; the clipped last tile of a
        // contig searches the one 64-position bucket its end lies in)
        const uint32_t kl = (uint32_t)from >> 6, k = (uint32_t)tend >> 6;
        ti.lo = kl > c.pidx_last ? c.n_reads : c.pidx[kl];
        const uint32_t a = k > c.pidx_last ? c.n_reads : c.pidx[k];
        ti.hi = (tend & 63) == 0 ? a : a + lower_bound_i32(c.pos + a, (k + 1u > c.pidx_last ? c.n_reads : c.pidx[k + 1]) - a, tend);
    } else {
        // no index.  ONE search per tile: s = first read at or past the tile's first position.  The tile's last
        // read is the next tile's s (the neighbouring lane has it; the last lane of a wave and the last tile of a
        // contig search for themselves), and the look-back start lies a few dozen reads before s: found by stepping
        // back from it.
        const uint32_t n = c.n_reads;
        const uint32_t g0 = (uint32_t)(((uint64_t)(uint32_t)ti.t0 * n) / (uint32_t)c.length);
        const uint32_t s0 = ti.t0 == 0 ? 0u : lower_bound_hint(c.pos, n, ti.t0, g0);
        const uint32_t nxt = (uint32_t)__shfl_down((int)s0, 1, WAVE);
        const int tnx = __shfl_down(t, 1, WAVE);                  // (lanes past the last tile returned above: not read)
        const bool have = (threadIdx.x & 63u) != 63u && t + 1 < job.n_tiles && tnx == t + 1 &&
                          t + 1 - c.tile_beg < c.n_tiles && ti.t0 + T == tend;
        if (have) ti.hi = nxt;
        else {
            const uint32_t g1 = (uint32_t)(((uint64_t)(uint32_t)tend * n) / (uint32_t)c.length);
            ti.hi = lower_bound_hint(c.pos + s0, n - s0, tend, g1 > s0 ? g1 - s0 : 0u) + s0;
        }
        // first read with pos >= from (from <= t0): step back from s0
        uint32_t lo = 0, hi2 = s0;
        if (from == 0) hi2 = 0u;
        for (uint32_t step = 64u; hi2 > 0u; step <<= 2) {
            const uint32_t p = hi2 > step ? hi2 - step : 0u;
            if (c.pos[p] >= from) hi2 = p; else { lo = p + 1u; break; }
        }
        while (lo < hi2) {
            const uint32_t mid = lo + ((hi2 - lo) >> 1);
            if (c.pos[mid] < from) lo = mid + 1u; else hi2 = mid;
        }
        ti.lo = lo;
    }
    if (job.fast == 2u) ti.lo &= ~3u;   // the raw straight-line kernel loads four reads per lane with aligned vector loads
    ti.clo = c.n_reads ? c.off[ti.lo] : 0u;
    ti.chi = c.n_reads ? c.off[ti.hi] : 0u;
    if (!job.fast) { job.tiles[t] = ti; return; }
    // fast run: an ordinary tile gets its resolved record; anything else joins the slow list
    TileFast tf;
    const uint32_t nrd = ti.hi - ti.lo, nst = ti.chi - ti.clo;
    const bool ordinary = tend - ti.t0 == T && nrd <= 1024u && nst <= (job.fast == 2u ? 1280u : 1024u);
    tf.nrd = 0xffffffffu;
    if (ordinary) {
        tf.pos = c.pos + ti.lo; tf.cig = c.cigar + ti.clo;
        if (job.fast == 2u) { tf.rec = c.off + ti.lo; tf.flag = c.flag + ti.lo; tf.mapq = c.mapq + ti.lo; }
        else { tf.rec = c.rec + ti.lo; tf.flag = nullptr; tf.mapq = nullptr; }
        tf.out = job.perbase ? job.perbase + c.base_off + ti.t0 : nullptr;
        tf.wsum = job.win_sum + c.win_off;
        tf.wmin = job.win_min + c.win_off;
        tf.t0 = ti.t0; tf.nrd = nrd; tf.nst = nst; tf.clo = ti.clo; tf.ctg = lo;
        tf.pad_[0] = tf.pad_[1] = tf.pad_[2] = 0;
        const uint32_t W = (uint32_t)job.W;
        const uint32_t stepc = job.step > 0x7fffffffLL ? 0x7fffffffu : (uint32_t)job.step;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const uint32_t cpos0 = (uint32_t)ti.t0 + (uint32_t)(w * (T / 4));
            const uint32_t cw = cpos0 / W, wleft = W - (cpos0 - cw * W);
            const uint32_t srem = cpos0 % stepc, sleft = srem == 0u ? 0u : stepc - srem;
            tf.win0[w] = cw;
            tf.wleft[w] = wleft > (uint32_t)FAST_FAR ? FAST_BIG : (int32_t)wleft;
            tf.sleft[w] = sleft > (uint32_t)FAST_FAR ? FAST_BIG : (int32_t)sleft;
        }
    } else {
        const uint32_t slot = atomicAdd(&job.counters->n_slow[job.parity], 1u);
        job.tiles[slot] = ti;
    }
    job.ftiles[t] = tf;
}

__global__ __launch_bounds__(64) void gd_copy_words_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, uint32_t n)
{
    for (uint32_t i = threadIdx.x; i < n; i += 64u) dst[i] = src[i];
}

// What the host reads after a compute: the counter block and the boundaries that exist (at most `spec`), stored
// straight into page-locked host memory.
__global__ __launch_bounds__(256) void gd_readback_kernel(const Counters* __restrict__ k, const int2* __restrict__ ordered, uint32_t spec,
                                                          Counters* __restrict__ h_counters, int2* __restrict__ h_bounds)
{
    const uint32_t n = k->run_cursor < spec ? k->run_cursor : spec;
    for (uint32_t i = threadIdx.x; i < n; i += 256u) h_bounds[i] = ordered[i];
    if (threadIdx.x < sizeof(Counters) / 4u)
        reinterpret_cast<uint32_t*>(h_counters)[threadIdx.x] = reinterpret_cast<const uint32_t*>(k)[threadIdx.x];
}

// ---------------------------------------------------------------------------
// wavefront primitives (wave64, DPP)
// ---------------------------------------------------------------------------
// Inclusive prefix sum across the 64 lanes of a wave.  row_shr:1,2,4,8 build
// the scan inside each row of 16 lanes; row_bcast15 / row_bcast31 carry the
// row totals (gfx9-family DPP controls, present on gfx950).
__device__ __forceinline__ int wave_inclusive_scan(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);  // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);  // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);  // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);  // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1,3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2,3
    return v;
}

// Four independent wave64 inclusive scans, advanced together: each DPP step of
// one row is followed by the same step of the other three, so a result is read
// three instructions after it was written (VALU write -> DPP read needs two
// wait states) and no hazard no-op is issued.  One volatile block: the
// compiler would otherwise serialise the four chains again.
__device__ __forceinline__ void scan4(int& a, int& b, int& c, int& d)
{
#define GD_DPP4(ctl)                                  \
    "v_add_u32_dpp %0, %0, %0 " ctl "\n\t"            \
    "v_add_u32_dpp %1, %1, %1 " ctl "\n\t"            \
    "v_add_u32_dpp %2, %2, %2 " ctl "\n\t"            \
    "v_add_u32_dpp %3, %3, %3 " ctl "\n\t"
    asm volatile(
        "s_nop 1\n\t"
        GD_DPP4("row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1")
        GD_DPP4("row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1")
        GD_DPP4("row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1")
        GD_DPP4("row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1")
        GD_DPP4("row_bcast:15 row_mask:0xa bank_mask:0xf")
        GD_DPP4("row_bcast:31 row_mask:0xc bank_mask:0xf")
        : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
#undef GD_DPP4
}

__device__ __forceinline__ int wave_sum(int v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, WAVE);
    return v;
}

__device__ __forceinline__ long long wave_sum64(long long v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, WAVE);
    return v;
}

__device__ __forceinline__ int wave_min(int v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        int o = __shfl_xor(v, m, WAVE);
        v = o < v ? o : v;
    }
    return v;
}

// ---- the same two reductions over MANY regions in one launch each (a --bed file with 10^5 rows) ----
struct RegionTab {
    const int32_t* depth;     // per-base vector of the region's contig
    int64_t  clen, start, end;
    int64_t  first_win;       // start / W
    uint32_t win_off;         // first window of the region among the batch's windows
    uint32_t chunk_off;       // first chunk of the region among the batch's chunks
};
constexpr int REGION_CHUNK = 4096;   // positions per workgroup of gd_regions_bounds_kernel

// One workgroup per window of the batch (win_region[w] = its region).
__global__ __launch_bounds__(256) void gd_regions_windows_kernel(const RegionTab* __restrict__ tab,
                                                                 const uint32_t* __restrict__ win_region, int32_t W,
                                                                 int64_t* __restrict__ sums, int32_t* __restrict__ mins)
{
    __shared__ long long s_sum[4];
    __shared__ int s_min[4];
    const uint32_t w = blockIdx.x;
    const RegionTab t = tab[win_region[w]];
    const int64_t k = t.first_win + (int64_t)(w - t.win_off);
    int64_t ws = k * W, we = ws + W;
    if (ws < t.start) ws = t.start;
    if (we > t.end) we = t.end;
    long long acc = 0;
    int mn = 0x7fffffff;
    for (int64_t p = ws + threadIdx.x; p < we; p += 256) {
        const int d = p < t.clen ? t.depth[p] : 0;
        acc += d;
        mn = d < mn ? d : mn;
    }
    acc = wave_sum64(acc);
    mn = wave_min(mn);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) { s_sum[wv] = acc; s_min[wv] = mn; }
    __syncthreads();
    if (threadIdx.x == 0) {
        int m = s_min[0];
        for (int i = 1; i < 4; ++i) m = s_min[i] < m ? s_min[i] : m;
        sums[w] = s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3];
        mins[w] = m;
    }
}

// One workgroup per chunk of REGION_CHUNK positions of a region: entries {p, class, region, 0} for
// p == start or class(p) != class(p-1), appended unordered (the host sorts by region and position).
__global__ __launch_bounds__(256) void gd_regions_bounds_kernel(const RegionTab* __restrict__ tab,
                                                                const uint32_t* __restrict__ chunk_region, int mincov,
                                                                int maxmean, int4* __restrict__ out, uint32_t cap,
                                                                uint32_t* __restrict__ cursor)
{
    const uint32_t ch = blockIdx.x;
    const uint32_t r = chunk_region[ch];
    const RegionTab t = tab[r];
    const int64_t p0 = t.start + (int64_t)(ch - t.chunk_off) * REGION_CHUNK;
    const int64_t p1 = p0 + REGION_CHUNK < t.end ? p0 + REGION_CHUNK : t.end;
    for (int64_t p = p0 + threadIdx.x; p < p1; p += 256) {
        const int d = p < t.clen ? t.depth[p] : 0;
        const int cl = cov_class(d, mincov, maxmean);
        bool b = (p == t.start);
        if (!b) {
            const int dp = (p - 1) < t.clen ? t.depth[p - 1] : 0;
            b = cov_class(dp, mincov, maxmean) != cl;
        }
        if (b) {
            const uint32_t i = atomicAdd(cursor, 1u);
            if (i < cap) out[i] = make_int4((int)p, cl, (int)r, 0);
        }
    }
}

}  // namespace gd

#include "gd_index.hpp"
#include "gd_tile_common.hpp"
#include "gd_tile_generic.hpp"
#include "gd_tile_fast.hpp"
#include "gd_sums_stream.hpp"
#include "gd_scatter.hpp"
#include "gd_chunk.hpp"
#include "gd_depthwed.hpp"
#include "gd_seqstats.hpp"
#include "gd_multidepth.hpp"
#include "gd_inflate.hpp"
#include "gd_bamdecode.hpp"
#include "gd_stage.hpp"

namespace gd {

// ---------------------------------------------------------------------------
// K2: order the per-tile boundary chunks (tiles are in genome order)
// ---------------------------------------------------------------------------
// One workgroup per group of SUPER tiles.  Its destination base is the sum of
// the group counts before it (a few hundred values), then one block scan of the
// group's tile counts places every chunk.  Groups without boundaries exit at
// once, which is the common case for whole-genome data.
__global__ __launch_bounds__(SUPER) void gd_runs_order_kernel(const int2* __restrict__ chunks,
                                                              uint32_t run_cap,
                                                              const uint32_t* __restrict__ tile_cnt,
                                                              const uint32_t* __restrict__ tile_off,
                                                              const uint32_t* __restrict__ super_cnt,
                                                              int2* __restrict__ ordered, int n_tiles)
{
    __shared__ uint32_t s_w[SUPER / WAVE];
    __shared__ uint32_t s_base;
    const int g = blockIdx.x;
    if (super_cnt[g] == 0) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    uint32_t part = 0;
    for (int i = tid; i < g; i += SUPER) part += super_cnt[i];
    part = (uint32_t)wave_sum((int)part);
    if (lane == 0) s_w[wv] = part;
    __syncthreads();
    if (tid == 0) {
        uint32_t b = 0;
        for (int i = 0; i < SUPER / WAVE; ++i) b += s_w[i];
        s_base = b;
    }
    __syncthreads();
    const uint32_t base = s_base;
    __syncthreads();
    const int t = g * SUPER + tid;
    const uint32_t cnt = t < n_tiles ? tile_cnt[t] : 0;
    const uint32_t incl = (uint32_t)wave_inclusive_scan((int)cnt);
    if (lane == 63) s_w[wv] = incl;
    __syncthreads();
    uint32_t before = 0;
    for (int i = 0; i < wv; ++i) before += s_w[i];
    if (cnt == 0) return;
    const uint32_t dst = base + before + incl - cnt;
    const uint32_t src = tile_off[t];
    for (uint32_t k = 0; k < cnt; ++k)
        if (src + k < run_cap && dst + k < run_cap) ordered[dst + k] = chunks[src + k];
}

// ---------------------------------------------------------------------------
// K2b: packed export block (gd_set_export): the results a merge step needs -- what the
// reference's merge loop (depth/depth.go:394-421) collects from its workers -- in ONE
// caller-owned device buffer (an RCCL send buffer), written before gd_compute's single
// synchronisation.  int64 words: [n_bounds][sums: max_w][mins: ceil(max_w/2)][bounds: cap_b].
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gd_export_kernel(const int64_t* __restrict__ wsum, const int32_t* __restrict__ wmin,
                                                        int64_t n_win, const int2* __restrict__ ordered,
                                                        const Counters* __restrict__ counters, int64_t* __restrict__ dst,
                                                        int64_t max_w, int64_t cap_b, uint32_t run_cap)
{
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    const uint32_t nb = counters->run_cursor;
    if (gid == 0) dst[0] = (int64_t)nb;
    int64_t* const d_sum = dst + 1;
    int32_t* const d_min = reinterpret_cast<int32_t*>(dst + 1 + max_w);
    int2* const d_bnd = reinterpret_cast<int2*>(dst + 1 + max_w + (max_w + 1) / 2);
    for (int64_t w = gid; w < n_win; w += gsz) { d_sum[w] = wsum[w]; d_min[w] = wmin[w]; }
    // `ordered` holds run_cap entries: an attempt that found more (dst[0] says so; gd_compute grows the arrays
    // and runs again, rewriting this block) must not be read past its end
    int64_t m = (int64_t)nb < cap_b ? (int64_t)nb : cap_b;
    m = m < (int64_t)run_cap ? m : (int64_t)run_cap;
    for (int64_t k = gid; k < m; k += gsz) d_bnd[k] = ordered[k];
}

// ---------------------------------------------------------------------------
// K3: --bed mode reductions over the resident per-base vector
// ---------------------------------------------------------------------------
// One workgroup per (clipped) window of the region: sum and min of
// depth[ws..we).  Positions >= contig length count as depth 0.
__global__ __launch_bounds__(256) void gd_region_windows_kernel(const int32_t* __restrict__ depth,
                                                                int64_t clen, int64_t start,
                                                                int64_t end, int32_t W,
                                                                int64_t first_win,
                                                                int64_t* __restrict__ sums,
                                                                int32_t* __restrict__ mins)
{
    __shared__ long long s_sum[4];
    __shared__ int s_min[4];
    const int64_t k = first_win + blockIdx.x;
    int64_t ws = k * W, we = ws + W;
    if (ws < start) ws = start;
    if (we > end) we = end;
    long long acc = 0;
    int mn = 0x7fffffff;
    for (int64_t p = ws + threadIdx.x; p < we; p += 256) {
        const int d = p < clen ? depth[p] : 0;
        acc += d;
        mn = d < mn ? d : mn;
    }
    acc = wave_sum64(acc);
    mn = wave_min(mn);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) { s_sum[wv] = acc; s_min[wv] = mn; }
    __syncthreads();
    if (threadIdx.x == 0) {
        long long t = s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3];
        int m = s_min[0];
        for (int i = 1; i < 4; ++i) m = s_min[i] < m ? s_min[i] : m;
        sums[blockIdx.x] = t;
        mins[blockIdx.x] = m;
    }
}

// Class boundaries inside [start,end): entry for p iff p == start or
// class(p) != class(p-1).  Unordered append + host sort (regions are small).
__global__ void gd_region_bounds_kernel(const int32_t* __restrict__ depth, int64_t clen,
                                        int64_t start, int64_t end, int mincov, int maxmean,
                                        int2* __restrict__ out, uint32_t cap,
                                        uint32_t* __restrict__ cursor)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for (int64_t p = start + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < end; p += gsz) {
        const int d = p < clen ? depth[p] : 0;
        const int cl = cov_class(d, mincov, maxmean);
        bool b = (p == start);
        if (!b) {
            const int dp = (p - 1) < clen ? depth[p - 1] : 0;
            b = cov_class(dp, mincov, maxmean) != cl;
        }
        if (b) {
            const uint32_t i = atomicAdd(cursor, 1u);
            if (i < cap) out[i] = make_int2((int)p, cl);
        }
    }
}

}  // namespace gd
