// gd_multidepth.hpp -- device side of `multidepth` (/root/reference/multidepth/multidepth.go).
//
// The reference runs `samtools depth -q 0 -Q Q -d MaxCov -r chrom:start bam1 bam2 ...`
// per 5 Mb chunk (:203-207), parses one text line per covered position back into
// ints (:148-161), asks whether more than minSamples samples reach MinCov
// (sufficientDepth, :163-171) and, for the blocks its state machine cuts out
// (:217-258), averages every sample's depth over the block's sufficient sites
// (means, :270-283).  Here the S samples are S contigs of one engine whose
// per-base vectors are already in HBM after gd_compute:
//   gd_md_flags_kernel  one thread per position: reads the S depths (coalesced
//       per sample), writes two bitmaps -- `any` (some sample has depth > 0: the
//       positions a multi-file `samtools depth` prints) and `suf`
//       (sufficientDepth).  HBM-bound: 4*S bytes per position read, 2 bits written.
//   gd_md_sums_kernel   one thread per (block, sample): the reference's
//       `dps[i] += float64(d) / 1000.` over the block's sufficient sites, in
//       position order, in IEEE double -- the same operations in the same order,
//       so the "%.2f" the host prints is the reference's to the last digit.
// The block state machine (aggregate / splitBlocks, :188-268) is a sequential loop over the printed
// sites of a 5 Mb chunk in the reference; here it is restated as set operations over the two bitmaps
// (gd_md_runs_kernel .. gd_md_blocks_kernel below) and runs on the device too.  Every value crossing
// the ABI is an integer (bitmaps, block bounds) or the exact double the reference would hold.
// Samples can be brought in GROUPS (gd_md_acc_kernel accumulates the per-position counts): only one
// group's per-base vectors have to be resident at a time.
#pragma once

namespace gd {

struct MdFlagsJob {
    const int32_t* const* depth;   // [n_samples] per-base vectors (device pointers)
    int32_t  n_samples;
    int64_t  len;                  // positions
    int32_t  min_cov;
    int32_t  min_samples;          // sufficient: count(depth >= min_cov) > min_samples
    uint32_t* any_bits;            // [ceil(len/32)] little-endian bit per position
    uint32_t* suf_bits;
};

__global__ __launch_bounds__(256) void gd_md_flags_kernel(MdFlagsJob j)
{
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool valid = p < j.len;
    int cnt = 0;
    int nz = 0;
    int s = 0;
    for (; s + 4 <= j.n_samples; s += 4) {               // four independent loads in flight
        const int32_t* d0 = j.depth[s];
        const int32_t* d1 = j.depth[s + 1];
        const int32_t* d2 = j.depth[s + 2];
        const int32_t* d3 = j.depth[s + 3];
        const int a = valid ? d0[p] : 0, b = valid ? d1[p] : 0, c = valid ? d2[p] : 0, d = valid ? d3[p] : 0;
        cnt += (a >= j.min_cov) + (b >= j.min_cov) + (c >= j.min_cov) + (d >= j.min_cov);
        nz |= a | b | c | d;
    }
    for (; s < j.n_samples; ++s) {
        const int a = valid ? j.depth[s][p] : 0;
        cnt += a >= j.min_cov;
        nz |= a;
    }
    // depths are >= 0, so nz != 0 <=> some sample covers the position
    const unsigned long long am = __builtin_amdgcn_ballot_w64(valid && nz != 0);
    const unsigned long long sm = __builtin_amdgcn_ballot_w64(valid && cnt > j.min_samples);
    const int lane = threadIdx.x & 63;
    const int64_t w = p >> 5;                              // the wave covers words w, w+1 of lane 0
    const int64_t n_words = (j.len + 31) >> 5;
    if (lane == 0) {
        if (w < n_words) { j.any_bits[w] = (uint32_t)am; j.suf_bits[w] = (uint32_t)sm; }
        if (w + 1 < n_words) { j.any_bits[w + 1] = (uint32_t)(am >> 32); j.suf_bits[w + 1] = (uint32_t)(sm >> 32); }
    }
}

// The same over a GROUP of samples, accumulating: cnt[p] += #{samples of the group with depth >= min_cov}
// (16 bits: up to 65535 samples), nz bit p |= some sample of the group covers p.  gd_md_finish_kernel
// turns the accumulators into the two bitmaps once every group has been added.
struct MdAccJob {
    const int32_t* const* depth;   // [n_samples] per-base vectors of this group
    int32_t  n_samples;
    int64_t  len;
    int32_t  min_cov;
    uint16_t* cnt;                 // [len]
    uint32_t* any_bits;            // [ceil(len/32)], OR-ed
};

__global__ __launch_bounds__(256) void gd_md_acc_kernel(MdAccJob j)
{
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool valid = p < j.len;
    int cnt = 0, nz = 0;
    for (int s = 0; s < j.n_samples; ++s) {
        const int a = valid ? j.depth[s][p] : 0;
        cnt += a >= j.min_cov;
        nz |= a;
    }
    if (valid && cnt) j.cnt[p] = (uint16_t)(j.cnt[p] + cnt);
    const unsigned long long am = __builtin_amdgcn_ballot_w64(valid && nz != 0);
    const int lane = threadIdx.x & 63;
    const int64_t w = p >> 5, n_words = (j.len + 31) >> 5;
    if (lane == 0) {                                     // this wave owns words w, w + 1: no atomics needed
        if (w < n_words && (uint32_t)am) j.any_bits[w] |= (uint32_t)am;
        if (w + 1 < n_words && (uint32_t)(am >> 32)) j.any_bits[w + 1] |= (uint32_t)(am >> 32);
    }
}

__global__ __launch_bounds__(256) void gd_md_finish_kernel(const uint16_t* __restrict__ cnt, int64_t len,
                                                           int32_t min_samples, uint32_t* __restrict__ suf_bits)
{
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const unsigned long long sm = __builtin_amdgcn_ballot_w64(p < len && (int)cnt[p < len ? p : 0] > min_samples);
    const int lane = threadIdx.x & 63;
    const int64_t w = p >> 5, n_words = (len + 31) >> 5;
    if (lane == 0) {
        if (w < n_words) suf_bits[w] = (uint32_t)sm;
        if (w + 1 < n_words) suf_bits[w + 1] = (uint32_t)(sm >> 32);
    }
}

// ---------------------------------------------------------------------------------------------------
// The block finder.  What `aggregate` (multidepth.go:203-268) does to one chunk whose 0-based start is i,
// as set operations over A = `any` (printed sites) and S = `suf` (sufficient sites; N = A & ~S):
//   * nothing is looked at before z0 = the first N site >= i (`seen0`, :229-246);
//   * the stream ends at E = the first N site p > i + 1 + chunk with no S site in
//     [max(z0, p - max_skip + 1), p)  (:232-240: the cache is empty or its last site is >= max_skip behind;
//     every S site >= z0 enters the cache, so "its last site" is the last S site before p), or at the contig end;
//   * the caches are the RUNS of S sites in [z0, E): a new run starts at an S site with no S site in the
//     max_skip + 1 positions before it (:247-259) -- N sites in between do not matter;
//   * a run is reported if it has >= min_size sites, or if nothing flushed it: it is the last run of the
//     stream and no printed site q with q - (last + 1) > max_skip follows it before E (:261-266);
//   * splitBlocks (:188-201) cuts a reported run greedily: a block ends with the last S site less than
//     `window` after its first one.
// Runs do not depend on the chunk (only their clipping to [z0, E) does), so they are found ONCE for the
// contig: run starts / ends as bitmaps (one thread per word, bounded look-around), compacted in position
// order with prefix popcounts; each chunk then takes the runs that overlap its stream.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t md_below(int b) { return b >= 32 ? ~0u : ((1u << b) - 1u); }   // bits < b

// first set bit of (a & ~b) at or after `from`, below `lim`; `lim` if none  (b may be null)
__device__ __forceinline__ int64_t md_next_set(const uint32_t* __restrict__ a, const uint32_t* __restrict__ b,
                                               int64_t from, int64_t lim)
{
    if (from < 0) from = 0;
    if (from >= lim) return lim;
    int64_t k = from >> 5;
    const int64_t kl = (lim - 1) >> 5;
    uint32_t cur = (b ? a[k] & ~b[k] : a[k]) & ~md_below((int)(from & 31));
    for (;;) {
        if (cur) { const int64_t p = (k << 5) + (__ffs((int)cur) - 1); return p < lim ? p : lim; }
        if (++k > kl) return lim;
        cur = b ? a[k] & ~b[k] : a[k];
    }
}

// last set bit of a at or before `from`, not below `floor`; -1 if none
__device__ __forceinline__ int64_t md_prev_set(const uint32_t* __restrict__ a, int64_t from, int64_t floor)
{
    if (floor < 0) floor = 0;
    if (from < floor) return -1;
    int64_t k = from >> 5;
    const int64_t kf = floor >> 5;
    uint32_t cur = a[k] & md_below((int)(from & 31) + 1);
    for (;;) {
        if (cur) { const int64_t p = (k << 5) + 31 - __clz((int)cur); return p >= floor ? p : -1; }
        if (--k < kf) return -1;
        cur = a[k];
    }
}

struct MdBlkJob {
    const uint32_t* any_bits;
    const uint32_t* suf_bits;
    int64_t  len, nw;              // positions, words
    int64_t  chunk;
    int32_t  max_skip, min_size, window;
    uint32_t* rs; uint32_t* re;    // [nw] run-start / run-end bitmaps
    uint32_t* ps; uint32_t* prs; uint32_t* pre;   // [nw + 1] popcounts of suf / rs / re, then their exclusive scans
    int32_t* starts; int32_t* ends;               // [n_runs] in position order
    // per chunk
    int64_t  n_chunks;
    int64_t* z0; int64_t* E; int64_t* pa;         // stream begin / end, last printed site before E (-1: none)
    uint32_t* k_lo; uint32_t* pair_off;           // first run of the stream, offset of its (chunk, run) pairs; [n_chunks] = total
    // per (chunk, run) pair
    uint32_t* nblk;                               // [n_pairs + 1] blocks of the pair, then exclusive scan
    int64_t* bstart; int64_t* bend;               // [n_blocks]
};

// B1: run starts and ends of the whole contig.  One thread per word.
__global__ __launch_bounds__(256) void gd_md_runs_kernel(MdBlkJob j)
{
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= j.nw) return;
    const uint32_t sw = j.suf_bits[k];
    uint32_t rsw = 0, rew = 0;
    uint32_t rest = sw;
    int pb = -1;                                           // previous S bit of this word
    while (rest) {
        const int b = __ffs((int)rest) - 1;
        rest &= rest - 1;
        const int64_t p = (k << 5) + b;
        // a start: no S site in [p - max_skip - 1, p)
        bool start;
        if (pb >= 0) start = (b - pb - 1) > j.max_skip;
        else start = md_prev_set(j.suf_bits, p - 1, p - (int64_t)j.max_skip - 1) < 0;
        // an end: no S site in (p, p + max_skip + 1]
        bool end;
        if (rest) end = ((__ffs((int)rest) - 1) - b - 1) > j.max_skip;
        else {
            const int64_t lim = p + (int64_t)j.max_skip + 2 < j.len ? p + (int64_t)j.max_skip + 2 : j.len;
            end = md_next_set(j.suf_bits, nullptr, p + 1, lim) >= lim;
        }
        rsw |= start ? 1u << b : 0u;
        rew |= end ? 1u << b : 0u;
        pb = b;
    }
    j.rs[k] = rsw; j.re[k] = rew;
    j.ps[k] = (uint32_t)__popc(sw); j.prs[k] = (uint32_t)__popc(rsw); j.pre[k] = (uint32_t)__popc(rew);
}

// B2: positions of the run starts / ends in order (after the three scans).
__global__ __launch_bounds__(256) void gd_md_compact_kernel(MdBlkJob j)
{
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= j.nw) return;
    uint32_t a = j.rs[k], o = j.prs[k];
    while (a) { j.starts[o++] = (int32_t)((k << 5) + (__ffs((int)a) - 1)); a &= a - 1; }
    a = j.re[k]; o = j.pre[k];
    while (a) { j.ends[o++] = (int32_t)((k << 5) + (__ffs((int)a) - 1)); a &= a - 1; }
}

// number of set bits of bitmap w (with exclusive word scan pw) at positions < p
__device__ __forceinline__ uint32_t md_rank(const uint32_t* __restrict__ w, const uint32_t* __restrict__ pw,
                                            int64_t p, int64_t len, int64_t nw)
{
    if (p >= len) return pw[nw];
    return pw[p >> 5] + (uint32_t)__popc(w[p >> 5] & md_below((int)(p & 31)));
}

// B3: the stream of every chunk.  One workgroup; a thread per chunk (strided), then one serial scan.
__global__ __launch_bounds__(256) void gd_md_chunks_kernel(MdBlkJob j)
{
    const int64_t L = j.len;
    for (int64_t c = threadIdx.x; c < j.n_chunks; c += 256) {
        const int64_t i = c * j.chunk;
        const int64_t z0 = md_next_set(j.any_bits, j.suf_bits, i, L);
        int64_t E = L, pa = -1;
        uint32_t klo = 0, khi = 0;
        if (z0 < L) {
            int64_t p = z0;
            const int64_t from = i + j.chunk + 2;            // p > rstart + chunk, rstart = i + 1
            if (from > p) p = md_next_set(j.any_bits, j.suf_bits, from, L);
            while (p < L) {
                int64_t lo = p - (int64_t)j.max_skip + 1;
                if (lo < z0) lo = z0;
                if (md_prev_set(j.suf_bits, p - 1, lo) < 0) break;     // nothing cached within max_skip: samtools is killed
                p = md_next_set(j.any_bits, j.suf_bits, p + 1, L);
            }
            E = p;
            klo = md_rank(j.re, j.pre, z0, L, j.nw);          // runs that end before z0 are over
            khi = md_rank(j.rs, j.prs, E, L, j.nw);           // runs that start at or after E are never seen
            pa = md_prev_set(j.any_bits, E - 1, z0);
        }
        j.z0[c] = z0; j.E[c] = E; j.pa[c] = pa; j.k_lo[c] = klo;
        j.pair_off[c] = khi - klo;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (int64_t c = 0; c < j.n_chunks; ++c) { const uint32_t t = j.pair_off[c]; j.pair_off[c] = run; run += t; }
        j.pair_off[j.n_chunks] = run;
    }
}

// the run of one (chunk, run) pair clipped to the chunk's stream, whether it is reported, and its blocks
template <bool WRITE>
__device__ __forceinline__ uint32_t md_pair_blocks(const MdBlkJob& j, uint32_t pair, int64_t* bs, int64_t* be)
{
    // the chunk of this pair: last c with pair_off[c] <= pair
    int64_t lo = 0, hi = j.n_chunks;
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (j.pair_off[mid] <= pair) lo = mid; else hi = mid;
    }
    const int64_t c = lo;
    const uint32_t k = j.k_lo[c] + (pair - j.pair_off[c]);
    const bool last = pair + 1u == j.pair_off[c + 1];
    const int64_t z0 = j.z0[c], E = j.E[c];
    int64_t s = j.starts[k], e = j.ends[k];
    if (s < z0) s = md_next_set(j.suf_bits, nullptr, z0, e + 1);            // the stream begins inside the run
    if (e >= E) e = md_prev_set(j.suf_bits, E - 1, s);                      // ... or ends inside it
    if (e < 0 || s > e) return 0u;
    const uint32_t count = md_rank(j.suf_bits, j.ps, e + 1, j.len, j.nw) - md_rank(j.suf_bits, j.ps, s, j.len, j.nw);
    const bool flushed = !last || j.pa[c] >= e + (int64_t)j.max_skip + 2;
    if (flushed && (int64_t)count < (int64_t)j.min_size) return 0u;
    // splitBlocks
    uint32_t n = 0;
    int64_t b0 = s;
    while (b0 <= e) {
        const int64_t lim = (j.window > 0 && b0 + j.window - 1 < e) ? b0 + j.window - 1 : e;
        int64_t b1 = j.window > 0 ? md_prev_set(j.suf_bits, lim, b0) : b0;
        if (b1 < b0) b1 = b0;
        if (WRITE) { bs[n] = b0; be[n] = b1 + 1; }
        ++n;
        b0 = md_next_set(j.suf_bits, nullptr, b1 + 1, e + 1);
    }
    return n;
}

// B4 / B5: one thread per (chunk, run) pair: count its blocks, then (after the scan) write them.
__global__ __launch_bounds__(256) void gd_md_count_kernel(MdBlkJob j, uint32_t n_pairs)
{
    const uint32_t pair = blockIdx.x * 256u + threadIdx.x;
    if (pair >= n_pairs) return;
    j.nblk[pair] = md_pair_blocks<false>(j, pair, nullptr, nullptr);
}

__global__ __launch_bounds__(256) void gd_md_blocks_kernel(MdBlkJob j, uint32_t n_pairs)
{
    const uint32_t pair = blockIdx.x * 256u + threadIdx.x;
    if (pair >= n_pairs) return;
    const uint32_t o = j.nblk[pair];
    if (j.nblk[pair + 1] == o) return;
    (void)md_pair_blocks<true>(j, pair, j.bstart + o, j.bend + o);
}

struct MdSumsJob {
    const int32_t* const* depth;   // [n_samples]
    const uint32_t* suf_bits;
    const int64_t* start;          // [n_blocks] first position of the block
    const int64_t* end;            // [n_blocks] one past its last position
    double* sums;                  // [n_blocks][n_samples]
    int64_t n_blocks;
    int32_t n_samples;
};

__global__ __launch_bounds__(64) void gd_md_sums_kernel(MdSumsJob j)
{
    const int64_t gid = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (gid >= j.n_blocks * j.n_samples) return;
    const int64_t b = gid / j.n_samples;
    const int s = (int)(gid - b * j.n_samples);
    const int32_t* const d = j.depth[s];
    const int64_t s0 = j.start[b], e0 = j.end[b];
    double acc = 0.0;                                       // multidepth.go:271-277
    for (int64_t w = s0 >> 5; w <= (e0 - 1) >> 5 && e0 > s0; ++w) {
        uint32_t bits = j.suf_bits[w];
        const int64_t base = w << 5;
        if (base < s0) bits &= ~0u << (int)(s0 - base);
        if (base + 32 > e0) bits &= ~0u >> (int)(base + 32 - e0);
        while (bits) {
            const int k = __ffs((int)bits) - 1;
            bits &= bits - 1;
            acc += (double)d[base + k] / 1000.;
        }
    }
    j.sums[gid] = acc;
}

}  // namespace gd
