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
};

// Everything a workgroup needs for its tile in ONE 80-byte record (one scalar
// load burst, no dependent contig-table lookup).  Written by gd_prep_kernel.
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
};

// Device-side counters, read back once per gd_compute.
struct Counters {
    int32_t  max_span;        // largest reference span of a kept read
    uint32_t run_cursor;      // boundary entries allocated (may exceed capacity)
    uint32_t pad0, pad1;
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
    int32_t   ablate;         // debug only (GOLEFT_GD_ABLATE): 1 skip phase A, 2 skip LDS marks,
                              // 4 skip per-base stores, 8 skip window/class reductions
    int64_t   step;
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
    if (gid == 0) {
        job.counters->max_span = 0;
        job.counters->run_cursor = 0;
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
    ti.t0 = (t - c.tile_beg) * T;
    int32_t tend = ti.t0 + T < c.length ? ti.t0 + T : c.length;
    int32_t from = ti.t0 > job.lookback ? ti.t0 - job.lookback : 0;
    ti.lo = lower_bound_i32(c.pos, c.n_reads, from);
    ti.hi = lower_bound_i32(c.pos, c.n_reads, tend);
    ti.clo = c.n_reads ? c.off[ti.lo] : 0u;
    ti.chi = c.n_reads ? c.off[ti.hi] : 0u;
    job.tiles[t] = ti;
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

// ---------------------------------------------------------------------------
// K1: the tile kernel
// ---------------------------------------------------------------------------
// Wave-wide helpers built on DPP (no LDS traffic).
__device__ __forceinline__ int wave_total(int v)     // sum over the wave, valid in every lane
{
    return __builtin_amdgcn_readlane(wave_inclusive_scan(v), 63);
}

// value of lane-1 (lane 0 receives `first`): DPP wave_shr:1
__device__ __forceinline__ int wave_prev_lane(int v, int first)
{
    return __builtin_amdgcn_update_dpp(first, v, 0x138, 0xf, 0xf, false);
}

// Mark the clipped interval [s,e) of a read in the tile's difference array.
__device__ __forceinline__ void mark_interval(int32_t s, int32_t e, int32_t t0, int32_t tend,
                                              int32_t clen, int32_t* s_diff, int& prev_cnt)
{
    if (e > clen) e = clen;
    prev_cnt += (s < t0 && e >= t0) ? 1 : 0;       // covers t0-1
    if (e > t0 && s < tend) {
        const int32_t cs = (s > t0 ? s : t0) - t0;
        atomicAdd(&s_diff[cs], 1);
        if (e < tend) atomicAdd(&s_diff[e - t0], -1);
    }
}

// Walk one CIGAR (generic path): merge adjacent M/=/X ops into reference
// intervals and mark them.  `ops` points at op o0 of the read (LDS staging
// area or global memory).  Returns the reference span of the read.
template <typename OpPtr>
__device__ __forceinline__ int32_t walk_cigar(OpPtr ops, uint32_t n, int32_t p, int32_t t0,
                                              int32_t tend, int32_t clen, int32_t* s_diff,
                                              int& prev_cnt)
{
    int32_t cur = p;
    int32_t rs = -1;                               // open run start, -1 = none
    for (uint32_t k = 0; k <= n; ++k) {
        uint32_t op = 2, len = 0;                  // sentinel: a zero-length D closes the run
        if (k < n) { const uint32_t cg = ops[k]; op = cg & 0xf; len = cg >> 4; }
        const bool counted = (0x181u >> op) & 1u;  // M = X
        const bool consumes = (0x18du >> op) & 1u; // M D N = X
        if (counted) {
            if (rs < 0 && len > 0) rs = cur;
        } else if (consumes && rs >= 0) {
            mark_interval(rs, cur, t0, tend, clen, s_diff, prev_cnt);
            rs = -1;
        }
        if (consumes) cur += (int32_t)len;
    }
    return cur - p;
}

template <int T, int NT>
__global__ __launch_bounds__(NT) void gd_tile_kernel(Job job)
{
    constexpr int NW = NT / WAVE;          // waves per workgroup
    constexpr int CHUNK = T / NW;          // positions per wave
    constexpr int ROWS = CHUNK / 256;      // rows of 256 positions per wave
    constexpr int NWORDS = T / 32;         // bitmap words
    constexpr int BIG = 0x3fffffff;
    constexpr int CQ = (T * 3) / 8;        // staged CIGAR ops (30x/150 bp needs ~T/4)
    static_assert(CHUNK % 256 == 0, "wave chunk must be whole rows");

    __shared__ __attribute__((aligned(16))) int32_t s_diff[T];
    __shared__ uint32_t s_bmap[NWORDS];    // boundary bit per position
    __shared__ uint32_t s_clo[NWORDS];     // class bit 0 at boundary positions
    __shared__ uint32_t s_chi[NWORDS];     // class bit 1 at boundary positions
    __shared__ __attribute__((aligned(16))) uint32_t s_cig[CQ];      // staged CIGAR ops
    __shared__ uint32_t s_wq[NW * 3 * WAVE]; // per-wave queues of multi-op reads
    __shared__ int32_t  s_wtot[NW];
    __shared__ uint32_t s_wcnt[NW];
    __shared__ int32_t  s_prev;            // depth at t0-1
    __shared__ uint32_t s_hasb;
    __shared__ uint32_t s_base;

    const int tid = threadIdx.x;
    const int lane = tid & (WAVE - 1);
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const TileInfo ti = job.tiles[blockIdx.x];
    const TileInfo& c = ti;                // contig fields live in the same record
    const int32_t t0 = ti.t0;
    const int32_t tend = t0 + T < c.length ? t0 + T : c.length;   // clipped tile end
    const int tlen = tend - t0;                                   // valid positions, 1..T

    // ---- loads first: records of the first batch and the tile's CIGAR range --
    // The reads [lo,hi) of a tile are contiguous, so their ops [clo,chi) are one
    // contiguous range: it is fetched with coalesced 16-byte loads at the same
    // time as the record fields (one memory round trip, no dependent second
    // one) and staged in LDS for the CIGAR decode.
    constexpr int U = 4;                          // reads per lane in flight
    constexpr int CCH = (CQ / 4 + NT - 1) / NT;   // 16-byte chunks per thread
    int32_t  p[U];
    uint32_t f[U], o0[U], o1[U], mq[U];
    const bool run_a = !(job.ablate & 1) && ti.lo < ti.hi;
    const uint32_t a0 = ti.clo & ~3u;             // 16-byte aligned start of the op range
    const bool staged = ti.chi - a0 <= (uint32_t)CQ;
    const uint32_t last = ti.hi - 1u;
    if (run_a) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            uint32_t i = ti.lo + u * NT + tid;
            i = i < last ? i : last;              // clamped: lanes past hi redo the last read
            p[u] = c.pos[i];
            f[u] = c.flag[i];
            mq[u] = c.mapq[i];
            o0[u] = c.off[i];
            o1[u] = c.off[i + 1];
        }
        if (staged) {
            uint4 cg4[CCH];
            const uint32_t nst = ti.chi - a0;
#pragma unroll
            for (int k = 0; k < CCH; ++k) {
                const uint32_t j = (uint32_t)(k * NT + tid) * 4u;
                cg4[k] = make_uint4(0, 0, 0, 0);
                if (j < nst) {
                    const uint32_t g = a0 + j;
                    if (g + 4u <= ti.n_ops) {
                        cg4[k] = *reinterpret_cast<const uint4*>(c.cigar + g);
                    } else {
                        if (g + 0u < ti.n_ops) cg4[k].x = c.cigar[g];
                        if (g + 1u < ti.n_ops) cg4[k].y = c.cigar[g + 1];
                        if (g + 2u < ti.n_ops) cg4[k].z = c.cigar[g + 2];
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < CCH; ++k) {
                const uint32_t j = (uint32_t)(k * NT + tid) * 4u;
                if (j < nst) *reinterpret_cast<uint4*>(&s_cig[j]) = cg4[k];
            }
        }
    }

    // ---- zero LDS (overlaps the loads above) -----------------------------
    {
        int4 z = make_int4(0, 0, 0, 0);
        int4* d4 = reinterpret_cast<int4*>(s_diff);
#pragma unroll
        for (int i = tid; i < T / 4; i += NT) d4[i] = z;
        for (int i = tid; i < NWORDS; i += NT) { s_bmap[i] = 0; s_clo[i] = 0; s_chi[i] = 0; }
        if (tid == 0) { s_prev = 0; s_hasb = 0; }
    }
    __syncthreads();

    // ---- phase A: reads -> clipped intervals -> LDS +1/-1 -----------------
    // Single-op reads (the bulk of short-read data) are marked straight away.
    // Multi-op reads are compacted into a per-wave queue and walked afterwards
    // with dense lanes, so the CIGAR loop runs once per wave, not once per slot.
    if (run_a) {
        int prev_cnt = 0;
        int span_max = 0;
        uint32_t* wq = &s_wq[wv * (3 * WAVE)];    // this wave's queue: p | o0 | n
        uint32_t qn = 0;                          // entries queued (wave uniform)
        for (uint32_t base = ti.lo; base < ti.hi; base += NT * U) {
            if (base != ti.lo) {                  // further batches (deep tiles)
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    uint32_t i = base + u * NT + tid;
                    i = i < last ? i : last;
                    p[u] = c.pos[i];
                    f[u] = c.flag[i];
                    mq[u] = c.mapq[i];
                    o0[u] = c.off[i];
                    o1[u] = c.off[i + 1];
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t i = base + u * NT + tid;
                const uint32_t n = o1[u] - o0[u];
                const bool keep = i < ti.hi && (f[u] & job.flag_mask) == 0 &&
                                  (int)mq[u] >= job.Q && n > 0;
                uint32_t cg = 0;
                if (keep) cg = staged ? s_cig[o0[u] - a0] : c.cigar[o0[u]];
                const uint32_t op = cg & 0xf;
                const int32_t len = (int32_t)(cg >> 4);
                const bool simple = keep && n == 1u && ((0x181u >> op) & 1u) && len > 0;
                if (simple) {
                    span_max = len > span_max ? len : span_max;
                    if (!(job.ablate & 2))
                        mark_interval(p[u], p[u] + len, t0, tend, c.length, s_diff, prev_cnt);
                }
                const bool cx = keep && !simple;
                const unsigned long long m = __ballot(cx);
                if (m != 0ull) {                  // wave uniform
                    const uint32_t r = qn + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32),
                                            __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                    if (cx) {
                        if (r < (uint32_t)WAVE) {
                            wq[r] = (uint32_t)p[u]; wq[WAVE + r] = o0[u]; wq[2 * WAVE + r] = n;
                        } else {                  // queue full: walk in place
                            const int32_t span = staged
                                ? walk_cigar(&s_cig[o0[u] - a0], n, p[u], t0, tend, c.length, s_diff, prev_cnt)
                                : walk_cigar(c.cigar + o0[u], n, p[u], t0, tend, c.length, s_diff, prev_cnt);
                            span_max = span > span_max ? span : span_max;
                        }
                    }
                    qn += (uint32_t)__popcll(m);
                    if (qn >= (uint32_t)WAVE) {   // drain a full queue
                        __builtin_amdgcn_wave_barrier();
                        const int32_t qp = (int32_t)wq[lane];
                        const uint32_t qo = wq[WAVE + lane], qk = wq[2 * WAVE + lane];
                        const int32_t span = staged
                            ? walk_cigar(&s_cig[qo - a0], qk, qp, t0, tend, c.length, s_diff, prev_cnt)
                            : walk_cigar(c.cigar + qo, qk, qp, t0, tend, c.length, s_diff, prev_cnt);
                        span_max = span > span_max ? span : span_max;
                        __builtin_amdgcn_wave_barrier();
                        qn = 0;
                    }
                }
            }
        }
        if (qn != 0) {                            // drain the rest
            __builtin_amdgcn_wave_barrier();
            if ((uint32_t)lane < qn) {
                const int32_t qp = (int32_t)wq[lane];
                const uint32_t qo = wq[WAVE + lane], qk = wq[2 * WAVE + lane];
                const int32_t span = staged
                    ? walk_cigar(&s_cig[qo - a0], qk, qp, t0, tend, c.length, s_diff, prev_cnt)
                    : walk_cigar(c.cigar + qo, qk, qp, t0, tend, c.length, s_diff, prev_cnt);
                span_max = span > span_max ? span : span_max;
            }
        }
        // rare: publish look-back violations so the host can re-run
        if (span_max > job.lookback) atomicMax(&job.counters->max_span, span_max);
        const int pc = wave_total(prev_cnt);
        if (lane == 0 && pc != 0) atomicAdd(&s_prev, pc);
    }
    __syncthreads();

    // ---- phase B pass 1: wave chunk totals -------------------------------
    const int chunk0 = wv * CHUNK;
    {
        int tot = 0;
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const int4 v = *reinterpret_cast<const int4*>(&s_diff[chunk0 + r * 256 + lane * 4]);
            tot += v.x + v.y + v.z + v.w;
        }
        tot = wave_total(tot);
        if (lane == 0) s_wtot[wv] = tot;
    }
    __syncthreads();

    // ---- phase B pass 2: scan, store, window reduce, class boundaries ----
    // All positions are tile relative 32-bit ints here; absolute = t0 + rel.
    {
        int carry = 0;                                   // depth at chunk start - 1
        for (int v = 0; v < wv; ++v) carry += s_wtot[v];
        int prev_last = (wv == 0) ? s_prev : carry;      // depth just before this chunk

        const int W = job.W;
        const int64_t cpos0 = (int64_t)t0 + chunk0;      // first position of this chunk
        int64_t cur_win = cpos0 / W;
        const int64_t nb_abs = (cur_win + 1) * (int64_t)W;
        int nb = (nb_abs - t0) > BIG ? BIG : (int)(nb_abs - t0);    // next window boundary (rel)
        const int64_t step = job.step;
        const int64_t nf_abs = ((cpos0 + step - 1) / step) * step;
        int nf = (nf_abs - t0) > BIG ? BIG : (int)(nf_abs - t0);    // next forced run break (rel)
        const int wstep = W > BIG ? BIG : W;
        const int fstep = step > BIG ? BIG : (int)step;
        int64_t* wsum = job.win_sum + c.win_off;
        int32_t* wmin = job.win_min + c.win_off;
        unsigned long long acc = 0;
        int mn = 0x7fffffff;
        bool any_pos = false;
        int32_t* out = job.perbase + c.base_off + t0;
        const int lo_thr = job.mincov > 1 ? job.mincov : 1;          // all depths in [lo_thr, hi_thr)
        const int hi_thr = job.maxmean > 0 ? job.maxmean : 0x7fffffff;  // are CALLABLE

#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const int rb = chunk0 + r * 256;             // row start (rel)
            const int ib = rb + lane * 4;                // this lane's first position (rel)
            if (rb >= tlen) {
                // rows past the (clipped) tile end: keep the padded per-base array zero
                *reinterpret_cast<int4*>(&out[ib]) = make_int4(0, 0, 0, 0);
                continue;
            }
            const int4 v = *reinterpret_cast<const int4*>(&s_diff[ib]);
            const int x0 = v.x, x1 = x0 + v.y, x2 = x1 + v.z, x3 = x2 + v.w;
            const int incl = wave_inclusive_scan(x3);
            const int base = carry + incl - x3;
            carry += __builtin_amdgcn_readlane(incl, 63);
            int nvalid = tlen - ib;
            nvalid = nvalid < 0 ? 0 : (nvalid > 4 ? 4 : nvalid);
            // positions at or past the contig end hold depth 0 (nothing is printed there)
            const int d0 = nvalid > 0 ? base + x0 : 0, d1 = nvalid > 1 ? base + x1 : 0;
            const int d2 = nvalid > 2 ? base + x2 : 0, d3 = nvalid > 3 ? base + x3 : 0;
            if (!(job.ablate & 4)) *reinterpret_cast<int4*>(&out[ib]) = make_int4(d0, d1, d2, d3);
            any_pos = true;
            if (job.ablate & 8) { acc += (uint32_t)d0 ^ (uint32_t)d3; continue; }
            const bool full_row = rb + 256 <= tlen;
            const int m01 = d0 < d1 ? d0 : d1, m23 = d2 < d3 ? d2 : d3;
            const int rmin = m01 < m23 ? m01 : m23;

            // ---- window sum / min (depth/depth.go:181-189, :293-305) -----
            if (nb >= rb + 256 && full_row) {
                // depths are < 2^30 (records per contig are capped), so 4 fit in 32 bits
                acc += (uint32_t)d0 + (uint32_t)d1 + (uint32_t)d2 + (uint32_t)d3;
                mn = rmin < mn ? rmin : mn;
            } else {
                int seg = rb;
                const int dd[4] = {d0, d1, d2, d3};
                while (nb < rb + 256 && nb < tlen) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int pj = ib + j;
                        if (pj >= seg && pj < nb) { acc += (uint32_t)dd[j]; mn = dd[j] < mn ? dd[j] : mn; }
                    }
                    const long long tot = wave_sum64((long long)acc);
                    const int m = wave_min(mn);
                    if (lane == 0) {
                        atomicAdd(reinterpret_cast<unsigned long long*>(&wsum[cur_win]),
                                  (unsigned long long)tot);
                        atomicMin(&wmin[cur_win], m);
                    }
                    acc = 0; mn = 0x7fffffff;
                    cur_win++; seg = nb;
                    nb = nb + wstep > BIG ? BIG : nb + wstep;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int pj = ib + j;
                    if (pj >= seg && j < nvalid) { acc += (uint32_t)dd[j]; mn = dd[j] < mn ? dd[j] : mn; }
                }
            }

            // ---- coverage class boundaries (depth/depth.go:307-323) -----
            const int pl = wave_prev_lane(d3, prev_last);
            prev_last = __builtin_amdgcn_readlane(d3, 63);
            const int x01 = d0 > d1 ? d0 : d1, x23 = d2 > d3 ? d2 : d3;
            int rmax = x01 > x23 ? x01 : x23;
            rmax = pl > rmax ? pl : rmax;
            const int rmin2 = pl < rmin ? pl : rmin;
            const bool quiet = rmin2 >= lo_thr && rmax < hi_thr;   // every class here is CALLABLE
            if (__ballot(!quiet) != 0ull || nf < rb + 256) {
                const int c0 = cov_class(d0, job.mincov, job.maxmean);
                const int c1 = cov_class(d1, job.mincov, job.maxmean);
                const int c2 = cov_class(d2, job.mincov, job.maxmean);
                const int c3 = cov_class(d3, job.mincov, job.maxmean);
                const int cp = cov_class(pl, job.mincov, job.maxmean);
                uint32_t bm = (uint32_t)(c0 != cp) | ((uint32_t)(c1 != c0) << 1) |
                              ((uint32_t)(c2 != c1) << 2) | ((uint32_t)(c3 != c2) << 3);
                while (nf < rb + 256) {                  // forced breaks (quirk Q1), incl. position 0
                    const int o = nf - ib;
                    if (o >= 0 && o < 4) bm |= 1u << o;
                    nf = nf + fstep > BIG ? BIG : nf + fstep;
                }
                bm &= (1u << nvalid) - 1u;
                if (__ballot(bm != 0) != 0ull) {
                    if (bm != 0) {
                        const uint32_t lo = ((uint32_t)(c0 & 1)) | ((uint32_t)(c1 & 1) << 1) |
                                            ((uint32_t)(c2 & 1) << 2) | ((uint32_t)(c3 & 1) << 3);
                        const uint32_t hi = ((uint32_t)(c0 >> 1)) | ((uint32_t)(c1 >> 1) << 1) |
                                            ((uint32_t)(c2 >> 1) << 2) | ((uint32_t)(c3 >> 1) << 3);
                        const int w = ib >> 5, sh = ib & 31;
                        atomicOr(&s_bmap[w], bm << sh);
                        atomicOr(&s_clo[w], (lo & bm) << sh);
                        atomicOr(&s_chi[w], (hi & bm) << sh);
                    }
                    if (lane == 0) s_hasb = 1;
                }
            }
        }
        // flush the open window segment of this wave
        if (any_pos) {
            const long long tot = wave_sum64((long long)acc);
            const int m = wave_min(mn);
            if (lane == 0) {
                atomicAdd(reinterpret_cast<unsigned long long*>(&wsum[cur_win]),
                          (unsigned long long)tot);
                atomicMin(&wmin[cur_win], m);
            }
        }
    }
    __syncthreads();

    // ---- phase C: compact class boundaries of this tile -------------------
    if (s_hasb == 0) {
        if (tid == 0) { job.tile_cnt[blockIdx.x] = 0; job.tile_off[blockIdx.x] = 0; }
        return;
    }
    {
        // blocked word ownership keeps thread order == position order
        static_assert(NWORDS <= NT || NWORDS % NT == 0, "bitmap words vs threads");
        constexpr int WPT = NWORDS <= NT ? 1 : NWORDS / NT;  // words per thread
        uint32_t cnt = 0;
        const int wbeg = tid * WPT;
#pragma unroll
        for (int j = 0; j < WPT; ++j)
            if (wbeg + j < NWORDS) cnt += __popc(s_bmap[wbeg + j]);
        const uint32_t incl = (uint32_t)wave_inclusive_scan((int)cnt);
        if (lane == 63) s_wcnt[wv] = incl;
        __syncthreads();
        uint32_t before = 0, total = 0;
        for (int v = 0; v < NW; ++v) { if (v < wv) before += s_wcnt[v]; total += s_wcnt[v]; }
        if (tid == 0) {
            const uint32_t b = atomicAdd(&job.counters->run_cursor, total);
            s_base = b;
            job.tile_cnt[blockIdx.x] = total;
            job.tile_off[blockIdx.x] = b;
            atomicAdd(&job.super_cnt[blockIdx.x / SUPER], total);
        }
        __syncthreads();
        uint32_t dst = s_base + before + incl - cnt;
#pragma unroll
        for (int j = 0; j < WPT; ++j) {
            const int w = wbeg + j;
            if (w >= NWORDS) break;
            uint32_t bits = s_bmap[w];
            const uint32_t lo = s_clo[w], hi = s_chi[w];
            while (bits) {
                const int b = __ffs(bits) - 1;
                bits &= bits - 1;
                if (dst < job.run_cap) {
                    const int cls = (int)((lo >> b) & 1u) | (int)(((hi >> b) & 1u) << 1);
                    job.run_chunks[dst] = make_int2(t0 + w * 32 + b, cls | (ti.ctg << 2));
                }
                ++dst;
            }
        }
    }
}

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
