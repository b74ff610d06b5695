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
template <int T, int NT>
__global__ __launch_bounds__(NT) void gd_tile_kernel(Job job)
{
    constexpr int NW = NT / WAVE;          // waves per workgroup
    constexpr int CHUNK = T / NW;          // positions per wave
    constexpr int ROWS = CHUNK / 256;      // rows of 256 positions per wave
    constexpr int NWORDS = T / 32;         // bitmap words
    static_assert(CHUNK % 256 == 0, "wave chunk must be whole rows");

    __shared__ __attribute__((aligned(16))) int32_t s_diff[T];
    __shared__ uint32_t s_bmap[NWORDS];    // boundary bit per position
    __shared__ uint32_t s_clo[NWORDS];     // class bit 0 at boundary positions
    __shared__ uint32_t s_chi[NWORDS];     // class bit 1 at boundary positions
    __shared__ int32_t  s_wtot[NW];
    __shared__ uint32_t s_wcnt[NW];
    __shared__ int32_t  s_prev;            // depth at t0-1
    __shared__ uint32_t s_hasb;
    __shared__ uint32_t s_base;

    const int tid = threadIdx.x;
    const int lane = tid & (WAVE - 1);
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const TileInfo ti = job.tiles[blockIdx.x];
    const ContigDev c = job.ctgs[ti.ctg];
    const int32_t t0 = ti.t0;
    const int32_t tend = t0 + T < c.length ? t0 + T : c.length;   // clipped tile end

    // ---- zero LDS -------------------------------------------------------
    {
        int4 z = make_int4(0, 0, 0, 0);
        int4* d4 = reinterpret_cast<int4*>(s_diff);
#pragma unroll
        for (int i = tid; i < T / 4; i += NT) d4[i] = z;
        for (int i = tid; i < NWORDS; i += NT) { s_bmap[i] = 0; s_clo[i] = 0; s_chi[i] = 0; }
        if (tid == 0) { s_prev = 0; s_hasb = 0; }
    }
    __syncthreads();

    // ---- phase A: CIGAR walk -> clipped intervals -> LDS +1/-1 ----------
    {
        int prev_cnt = 0;
        int span_max = 0;
        for (uint32_t i = ti.lo + tid; i < ti.hi; i += NT) {
            const int32_t p = c.pos[i];
            const uint32_t f = c.flag[i];
            const int mq = c.mapq[i];
            const uint32_t o0 = c.off[i];
            const uint32_t o1 = c.off[i + 1];
            if ((f & job.flag_mask) != 0 || mq < job.Q) continue;
            int32_t cur = p;
            int32_t rs = -1;                       // open run start, -1 = none
            for (uint32_t k = o0; k <= o1; ++k) {
                uint32_t op = 2, len = 0;          // sentinel: a zero-length D closes the run
                if (k < o1) { const uint32_t cg = c.cigar[k]; op = cg & 0xf; len = cg >> 4; }
                const bool counted = (0x181u >> op) & 1u;    // M = X
                const bool consumes = (0x18du >> op) & 1u;   // M D N = X
                if (counted) {
                    if (rs < 0 && len > 0) rs = cur;
                } else if (consumes && rs >= 0) {
                    // close run [rs, cur)
                    int32_t s = rs, e = cur;
                    if (e > c.length) e = c.length;
                    if (s < t0 && e >= t0) prev_cnt++;        // covers t0-1
                    if (e > t0 && s < tend) {
                        const int32_t cs = (s > t0 ? s : t0) - t0;
                        atomicAdd(&s_diff[cs], 1);
                        if (e < tend) atomicAdd(&s_diff[e - t0], -1);
                    }
                    rs = -1;
                }
                if (consumes) cur += (int32_t)len;
            }
            const int32_t span = cur - p;
            span_max = span > span_max ? span : span_max;
        }
        // rare: publish look-back violations so the host can re-run
        if (span_max > job.lookback) atomicMax(&job.counters->max_span, span_max);
        const int pc = wave_sum(prev_cnt);
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
        tot = wave_sum(tot);
        if (lane == 0) s_wtot[wv] = tot;
    }
    __syncthreads();

    // ---- phase B pass 2: scan, store, window reduce, class boundaries ----
    {
        int carry = 0;                                   // depth at chunk start - 1
        for (int v = 0; v < wv; ++v) carry += s_wtot[v];
        int prev_last = (wv == 0) ? s_prev : carry;      // depth just before this chunk

        const int W = job.W;
        const int64_t cpos0 = (int64_t)t0 + chunk0;      // first position of this chunk
        int64_t cur_win = cpos0 / W;
        int64_t nb = (cur_win + 1) * (int64_t)W;         // next window boundary
        const int64_t step = job.step;
        int64_t nf = ((cpos0 + step - 1) / step) * step; // next forced run break
        int64_t* wsum = job.win_sum + c.win_off;
        int32_t* wmin = job.win_min + c.win_off;
        long long acc = 0;
        int mn = 0x7fffffff;
        bool any_pos = false;
        int32_t* out = job.perbase + c.base_off + t0;

        for (int r = 0; r < ROWS; ++r) {
            const int ib = chunk0 + r * 256 + lane * 4;  // index inside the tile
            const int64_t rp = cpos0 + r * 256;          // row start (contig position)
            if (rp >= tend) {
                // rows past the (clipped) tile end: keep the padded per-base array zero
                *reinterpret_cast<int4*>(&out[ib]) = make_int4(0, 0, 0, 0);
                continue;
            }
            const int4 v = *reinterpret_cast<const int4*>(&s_diff[ib]);
            const int x0 = v.x, x1 = x0 + v.y, x2 = x1 + v.z, x3 = x2 + v.w;
            const int incl = wave_inclusive_scan(x3);
            const int base = carry + incl - x3;
            carry += __builtin_amdgcn_readlane(incl, 63);
            const int64_t p0 = rp + lane * 4;            // position of d0
            const int nvalid = (int)((tend - p0) < 0 ? 0 : ((tend - p0) > 4 ? 4 : (tend - p0)));
            // positions at or past the contig end hold depth 0 (nothing is printed there)
            const int d0 = nvalid > 0 ? base + x0 : 0, d1 = nvalid > 1 ? base + x1 : 0;
            const int d2 = nvalid > 2 ? base + x2 : 0, d3 = nvalid > 3 ? base + x3 : 0;
            *reinterpret_cast<int4*>(&out[ib]) = make_int4(d0, d1, d2, d3);
            any_pos = true;

            // ---- window sum / min (depth/depth.go:181-189, :293-305) -----
            if (nb >= rp + 256 && rp + 256 <= tend) {
                acc += (long long)d0 + d1 + d2 + d3;
                int m01 = d0 < d1 ? d0 : d1, m23 = d2 < d3 ? d2 : d3;
                int m = m01 < m23 ? m01 : m23;
                mn = m < mn ? m : mn;
            } else {
                int64_t seg = rp;
                const int dd[4] = {d0, d1, d2, d3};
                while (nb < rp + 256 && nb < tend) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int64_t pj = p0 + j;
                        if (pj >= seg && pj < nb) { acc += dd[j]; mn = dd[j] < mn ? dd[j] : mn; }
                    }
                    const long long tot = wave_sum64(acc);
                    const int m = wave_min(mn);
                    if (lane == 0) {
                        atomicAdd(reinterpret_cast<unsigned long long*>(&wsum[cur_win]),
                                  (unsigned long long)tot);
                        atomicMin(&wmin[cur_win], m);
                    }
                    acc = 0; mn = 0x7fffffff;
                    cur_win++; seg = nb; nb += W;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int64_t pj = p0 + j;
                    if (pj >= seg && j < nvalid) { acc += dd[j]; mn = dd[j] < mn ? dd[j] : mn; }
                }
            }

            // ---- coverage class boundaries (depth/depth.go:307-323) -----
            int pl = __shfl_up(d3, 1, WAVE);
            if (lane == 0) pl = prev_last;
            prev_last = __builtin_amdgcn_readlane(d3, 63);
            const int c0 = cov_class(d0, job.mincov, job.maxmean);
            const int c1 = cov_class(d1, job.mincov, job.maxmean);
            const int c2 = cov_class(d2, job.mincov, job.maxmean);
            const int c3 = cov_class(d3, job.mincov, job.maxmean);
            const int cp = cov_class(pl, job.mincov, job.maxmean);
            uint32_t bm = (uint32_t)(c0 != cp) | ((uint32_t)(c1 != c0) << 1) |
                          ((uint32_t)(c2 != c1) << 2) | ((uint32_t)(c3 != c2) << 3);
            while (nf < rp + 256) {                      // forced breaks (quirk Q1), incl. position 0
                const int64_t o = nf - p0;
                if (o >= 0 && o < 4) bm |= 1u << (int)o;
                nf += step;
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
        // flush the open window segment of this wave
        if (any_pos) {
            const long long tot = wave_sum64(acc);
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
// Single workgroup: exclusive scan of tile_cnt -> tile_dst.
__global__ __launch_bounds__(1024) void gd_runs_scan_kernel(const uint32_t* __restrict__ tile_cnt,
                                                            uint32_t* __restrict__ tile_dst,
                                                            int n_tiles)
{
    __shared__ uint32_t s_part[1024];
    const int tid = threadIdx.x;
    const int per = (n_tiles + 1023) / 1024;
    const int beg = tid * per;
    const int end = beg + per < n_tiles ? beg + per : n_tiles;
    uint32_t s = 0;
    for (int i = beg; i < end; ++i) s += tile_cnt[i];
    s_part[tid] = s;
    __syncthreads();
    // Hillis-Steele over 1024 partials
    for (int d = 1; d < 1024; d <<= 1) {
        uint32_t v = tid >= d ? s_part[tid - d] : 0;
        __syncthreads();
        s_part[tid] += v;
        __syncthreads();
    }
    uint32_t run = s_part[tid] - s;
    for (int i = beg; i < end; ++i) { tile_dst[i] = run; run += tile_cnt[i]; }
}

__global__ void gd_runs_gather_kernel(const int2* __restrict__ chunks, uint32_t run_cap,
                                      const uint32_t* __restrict__ tile_cnt,
                                      const uint32_t* __restrict__ tile_off,
                                      const uint32_t* __restrict__ tile_dst,
                                      int2* __restrict__ ordered, int n_tiles)
{
    // one wave per tile; lanes stride the tile's entries
    const int t = (int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const int lane = threadIdx.x & 63;
    if (t >= n_tiles) return;
    const uint32_t cnt = tile_cnt[t];
    if (cnt == 0) return;
    const uint32_t src = tile_off[t], dst = tile_dst[t];
    for (uint32_t k = lane; k < cnt; k += 64)
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
