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
    int32_t  length;
    int32_t  tile_beg;        // first global tile id of this contig
    int32_t  n_tiles;
    int64_t  base_off;        // element offset of this contig in the per-base array
    int64_t  win_off;         // element offset in the window arrays
    int32_t  tid;             // reference id in the BAM header
    int32_t  pad;
};

struct TileInfo {
    int32_t  ctg;             // index into the ContigDev table
    int32_t  t0;              // first reference position of the tile
    uint32_t lo, hi;          // read index range [lo,hi) that can touch the tile
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
    ti.ctg = lo;
    ti.t0 = (t - c.tile_beg) * T;
    int32_t tend = ti.t0 + T < c.length ? ti.t0 + T : c.length;
    int32_t from = ti.t0 > job.lookback ? ti.t0 - job.lookback : 0;
    ti.lo = lower_bound_i32(c.pos, c.n_reads, from);
    ti.hi = lower_bound_i32(c.pos, c.n_reads, tend);
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

// Walk one CIGAR (generic path): merge adjacent M/=/X ops into reference
// intervals, clip to the tile, mark +1/-1 in the LDS difference array.
// Returns the reference span of the read.
__device__ __forceinline__ int32_t walk_cigar(const uint32_t* __restrict__ cigar, uint32_t o0,
                                              uint32_t o1, int32_t p, int32_t t0, int32_t tend,
                                              int32_t clen, int32_t* s_diff, int& prev_cnt)
{
    int32_t cur = p;
    int32_t rs = -1;                               // open run start, -1 = none
    for (uint32_t k = o0; k <= o1; ++k) {
        uint32_t op = 2, len = 0;                  // sentinel: a zero-length D closes the run
        if (k < o1) { const uint32_t cg = cigar[k]; op = cg & 0xf; len = cg >> 4; }
        const bool counted = (0x181u >> op) & 1u;  // M = X
        const bool consumes = (0x18du >> op) & 1u; // M D N = X
        if (counted) {
            if (rs < 0 && len > 0) rs = cur;
        } else if (consumes && rs >= 0) {
            int32_t s = rs, e = cur;               // close run [rs, cur)
            if (e > clen) e = clen;
            if (s < t0 && e >= t0) prev_cnt++;     // covers t0-1
            if (e > t0 && s < tend) {
                const int32_t cs = (s > t0 ? s : t0) - t0;
                atomicAdd(&s_diff[cs], 1);
                if (e < tend) atomicAdd(&s_diff[e - t0], -1);
            }
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
    constexpr int QCAP = 1024;             // multi-op read queue (indices)
    constexpr int BIG = 0x3fffffff;
    static_assert(CHUNK % 256 == 0, "wave chunk must be whole rows");

    __shared__ __attribute__((aligned(16))) int32_t s_diff[T];
    __shared__ uint32_t s_bmap[NWORDS];    // boundary bit per position
    __shared__ uint32_t s_clo[NWORDS];     // class bit 0 at boundary positions
    __shared__ uint32_t s_chi[NWORDS];     // class bit 1 at boundary positions
    __shared__ uint32_t s_queue[QCAP];
    __shared__ int32_t  s_wtot[NW];
    __shared__ uint32_t s_wcnt[NW];
    __shared__ int32_t  s_prev;            // depth at t0-1
    __shared__ uint32_t s_hasb;
    __shared__ uint32_t s_base;
    __shared__ uint32_t s_qn;

    const int tid = threadIdx.x;
    const int lane = tid & (WAVE - 1);
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const TileInfo ti = job.tiles[blockIdx.x];
    const ContigDev c = job.ctgs[ti.ctg];
    const int32_t t0 = ti.t0;
    const int32_t tend = t0 + T < c.length ? t0 + T : c.length;   // clipped tile end
    const int tlen = tend - t0;                                   // valid positions, 1..T

    // ---- zero LDS -------------------------------------------------------
    {
        int4 z = make_int4(0, 0, 0, 0);
        int4* d4 = reinterpret_cast<int4*>(s_diff);
#pragma unroll
        for (int i = tid; i < T / 4; i += NT) d4[i] = z;
        for (int i = tid; i < NWORDS; i += NT) { s_bmap[i] = 0; s_clo[i] = 0; s_chi[i] = 0; }
        if (tid == 0) { s_prev = 0; s_hasb = 0; s_qn = 0; }
    }
    __syncthreads();

    // ---- phase A: reads -> clipped intervals -> LDS +1/-1 -----------------
    // U reads per lane are in flight at once (the phase is latency bound).
    // Single-op reads (the bulk of short-read data) are handled branch-light;
    // multi-op reads are queued and walked afterwards with dense lanes.
    int prev_cnt = 0;
    int span_max = 0;
    {
        constexpr int U = 4;
        for (uint32_t base = ti.lo; base < ti.hi; base += NT * U) {
            int32_t  p[U];
            uint32_t f[U], o0[U], o1[U], c0[U], idx[U];
            int      mq[U];
            bool     ok[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t i = base + u * NT + tid;
                ok[u] = i < ti.hi;
                idx[u] = ok[u] ? i : ti.lo;                     // any valid index
                p[u] = c.pos[idx[u]];
                f[u] = c.flag[idx[u]];
                mq[u] = c.mapq[idx[u]];
                o0[u] = c.off[idx[u]];
                o1[u] = c.off[idx[u] + 1];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                ok[u] = ok[u] && (f[u] & job.flag_mask) == 0 && mq[u] >= job.Q && o1[u] > o0[u];
                c0[u] = ok[u] ? c.cigar[o0[u]] : 0u;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t op = c0[u] & 0xf;
                const int32_t len = (int32_t)(c0[u] >> 4);
                const bool single = (o1[u] - o0[u]) == 1u;
                const bool simple = ok[u] && single && ((0x181u >> op) & 1u) && len > 0;
                if (simple) {
                    const int32_t s = p[u];
                    int32_t e = s + len;
                    span_max = len > span_max ? len : span_max;
                    if (e > c.length) e = c.length;
                    if (s < t0 && e >= t0) prev_cnt++;
                    if (e > t0 && s < tend) {
                        const int32_t cs = (s > t0 ? s : t0) - t0;
                        atomicAdd(&s_diff[cs], 1);
                        if (e < tend) atomicAdd(&s_diff[e - t0], -1);
                    }
                } else if (ok[u]) {
                    const uint32_t slot = atomicAdd(&s_qn, 1u);
                    if (slot < (uint32_t)QCAP) {
                        s_queue[slot] = idx[u];
                    } else {                                     // queue full: walk in place
                        const int32_t span = walk_cigar(c.cigar, o0[u], o1[u], p[u], t0, tend,
                                                        c.length, s_diff, prev_cnt);
                        span_max = span > span_max ? span : span_max;
                    }
                }
            }
        }
    }
    __syncthreads();
    {
        const uint32_t nq = s_qn < (uint32_t)QCAP ? s_qn : (uint32_t)QCAP;
        for (uint32_t j = tid; j < nq; j += NT) {
            const uint32_t i = s_queue[j];
            const int32_t span = walk_cigar(c.cigar, c.off[i], c.off[i + 1], c.pos[i], t0, tend,
                                            c.length, s_diff, prev_cnt);
            span_max = span > span_max ? span : span_max;
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
            *reinterpret_cast<int4*>(&out[ib]) = make_int4(d0, d1, d2, d3);
            any_pos = true;
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
