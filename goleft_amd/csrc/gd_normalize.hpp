// gd_normalize.hpp -- canonical CIGARs, built once when a contig's records arrive.
//
// `samtools depth` (no -J, no -q; /root/reference/depth/depth.go:45) distinguishes exactly two kinds
// of CIGAR op: M/=/X add 1 to every reference position they cover, D/N advance the reference without
// counting; I/S/H/P touch no reference position at all.  The canonical form keeps just that:
//
//   * I, S, H, P and zero-length ops are dropped;
//   * neighbouring M/=/X ops (neighbours once the dropped ops are gone) merge into ONE `M`,
//     neighbouring D/N ops into ONE `N` (a merged length that would not fit the 28-bit BAM length field
//     starts a new op of the same kind instead -- pathological input only);
//   * D/N ops after the last M are dropped (they cover nothing that is counted); a read without any
//     counted base has no ops at all.  Leading D/N ops stay: they shift where the first M begins.
//
// The result uses the BAM encoding (len << 4 | op, op = 0 `M` or 3 `N`), so every kernel that walks
// CIGARs reads it unchanged and produces the identical depth -- but a 150 bp read with soft clips or an
// insertion is now the single op `kM` (98 % of short reads instead of 92 %), which is what the tile
// kernel's straight-line path handles, and a long read carries about half as many ops.
//
// One launch set for ALL contigs of a batch (the 64-read units of the batch's contigs are numbered through; a
// wave finds its contig in a small table), no host round trip in between: (N1) one lane per read counts its
// canonical ops, a wave scan gives the offsets inside each 64-read unit and the unit totals; (N2) the unit
// totals of the whole batch are scanned (a contig's offsets are differences against its first unit's);
// (N3) one lane per read walks its ops again and writes them at unit offset + local offset.  Every array comes
// out of one allocation sized up front -- the canonical array at the size of the original (it can only
// shrink) -- so the host waits once, for the per-contig totals.
#pragma once

namespace gd {
namespace norm {

// The record word of a read, written next to its canonical CIGAR: everything the tile kernel needs
// besides `pos` and the ops themselves, in the 4 bytes the CSR offset would take --
//     flag (12 bits, SAMv1 defines 12) << 20 | MAPQ << 12 | number of canonical ops (0..4094)
// The op offsets are recovered in the kernel by a prefix sum over the tile's reads.  A contig with a FLAG
// above 0xfff or a read of 4095 or more canonical ops is marked (status bit) and runs the generic kernel.
constexpr uint32_t REC_NMAX = 0xfffu;

// last j with beg[j] <= x (beg[0] = 0, n >= 1, entries ascending)
__device__ __forceinline__ uint32_t batch_find(const uint32_t* __restrict__ beg, uint32_t n, uint32_t x)
{
    uint32_t lo = 0, hi = n;
    while (hi - lo > 1u) {
        const uint32_t mid = (lo + hi) >> 1;
        if (beg[mid] <= x) lo = mid; else hi = mid;
    }
    return lo;
}

struct NormJob {
    const int32_t*  pos;      // positions (the index kernel)
    uint32_t* pidx;           // position index, n_idx entries
    uint32_t  n_idx;
    uint32_t  pad_;
    const uint32_t* off;      // original CSR offsets (n_reads + 1)
    const uint32_t* cigar;    // original ops
    const uint16_t* flag;
    const uint8_t*  mapq;
    uint32_t* rec;            // record words (n_reads)
    uint32_t* status;         // bit 0: a record that does not fit its word
    uint32_t  n_reads;
    uint32_t  n_units;        // ceil(n_reads / 64)
    uint32_t* noff;           // canonical CSR offsets (n_reads + 1)
    uint32_t* unit;           // this contig's slice of the batch's unit array: n_units + 1 unit totals, then (after
                              // N2) exclusive offsets over the BATCH -- unit[k] - unit[0] is the contig's own offset,
                              // unit[n_units] - unit[0] its total (all modulo 2^32: a contig holds < 2^32 ops)
    uint32_t* ncig;           // canonical ops
    uint32_t* total;          // out: canonical ops of the contig
    uint32_t  rows;           // fused pass: a workgroup takes rows x 256 consecutive reads (4, 2 or 1: few ops per read -> many reads)
    uint32_t  blk_beg;        // fused pass: first workgroup of this contig
};

// The contigs of one batch: job j owns units [ubeg[j], ubeg[j + 1]) and index entries [ibeg[j], ibeg[j + 1]).
struct NormBatch {
    const NormJob*  jobs;
    const uint32_t* ubeg;     // n_jobs + 1
    const uint32_t* ibeg;     // n_jobs + 1
    uint32_t n_jobs;
    uint32_t n_units;         // ubeg[n_jobs]
    uint32_t n_idx;           // ibeg[n_jobs]
    // fused pass (gd_norm_fused_kernel)
    const uint32_t* bbeg;     // n_jobs + 1: first workgroup of every job
    uint32_t n_blocks;        // bbeg[n_jobs]
    unsigned long long* bstat;   // n_blocks look-back words, zeroed before the launch
    uint32_t* ticket;         // [0] workgroups handed out so far, [1] set when a look-back gave up (never expected)
};

#ifdef GD_WITH_CANONICAL   // canonical records are an optional part of the build (csrc/Makefile: make CANONICAL=1)
constexpr uint32_t LEN_MAX = 0x0fffffffu;

// Calls emit(op, len) for every canonical op of one read, in order; returns their number.
template <typename Emit>
__device__ __forceinline__ uint32_t canonical_walk(const uint32_t* __restrict__ ops, uint32_t n, Emit emit)
{
    uint32_t cnt = 0;
    uint32_t cur_len = 0, cur_kind = 2;        // open run: kind 0 = M, 1 = N, 2 = none
    // An N run is only emitted once an M follows it (trailing N runs are dropped).  When a merged N
    // length overflows the 28-bit field, full-length pieces are split off and counted in `held` (all of
    // length LEN_MAX by construction) until that M arrives.
    uint32_t held = 0;
    const uint32_t held_len = LEN_MAX;
    for (uint32_t k = 0; k < n; ++k) {
        const uint32_t cg = ops[k];
        const uint32_t op = cg & 0xfu, len = cg >> 4;
        const bool counted = (0x181u >> op) & 1u;     // M = X
        const bool consumes = (0x18du >> op) & 1u;    // M D N = X
        if (!consumes || len == 0u) continue;
        const uint32_t kind = counted ? 0u : 1u;
        if (kind == cur_kind && cur_len + len <= LEN_MAX) { cur_len += len; continue; }
        if (kind == cur_kind) {
            // overflow split: close a full-length op of this kind, keep the remainder open
            const uint32_t rest = cur_len + len - LEN_MAX;
            if (kind == 0u) { emit(0u, LEN_MAX); ++cnt; }
            else ++held;
            cur_len = rest;
            continue;
        }
        // the kind changes: close the open run
        if (cur_kind == 0u) { emit(0u, cur_len); ++cnt; }
        else if (cur_kind == 1u) {
            // an N run followed by an M: now it is known to matter
            for (uint32_t h = 0; h < held; ++h) { emit(3u, held_len); ++cnt; }
            held = 0;
            emit(3u, cur_len); ++cnt;
        }
        cur_kind = kind;
        cur_len = len;
    }
    if (cur_kind == 0u) { emit(0u, cur_len); ++cnt; }   // a trailing N run (and its held parts) is dropped
    return cnt;
}


// The same canonical form, built by a whole WAVE for one read: lane k takes op b + k of every group of 64
// ops.  Long reads (ONT / PacBio: 10^3..10^5 ops) make the one-lane walk above serial and uncoalesced;
// here the kept ops (reference consuming, length >= 1) of a group are classified in parallel, a lane
// whose kind differs from the kept op before it (the open run carried in for the first one) is the HEAD
// of a new run and closes the run before it; run lengths are differences of one wave prefix sum.  Trailing
// N runs are never closed by a head, i.e. dropped, as canonical_walk drops them.
// Returns the number of canonical ops (wave uniform); WRITE: stores them at out[0 ..).  `overflow` is set
// (and the result is meaningless) when a merged run would not fit the 28-bit length field or an op is
// longer than 2^22 bases -- input no aligner writes; the caller then walks that read with canonical_walk,
// which splits such runs.  Count and write passes take the same decision, both depend on the ops only.
template <bool WRITE>
__device__ __forceinline__ uint32_t wave_canonical(const uint32_t* __restrict__ ops, uint32_t n, int lane,
                                                   uint32_t* __restrict__ out, bool& overflow)
{
    uint32_t open_kind = 2u, open_len = 0u, w = 0u;            // wave uniform: the run still open, runs closed
    overflow = false;
    const unsigned long long below = (1ull << lane) - 1ull;
    for (uint32_t b = 0; b < n; b += 64u) {
        const uint32_t k = b + (uint32_t)lane;
        const uint32_t cg = k < n ? ops[k] : 0u;
        const uint32_t op = cg & 0xfu, len = cg >> 4;
        const bool kept = ((0x18du >> op) & 1u) && len != 0u;    // M D N = X, not empty
        const uint32_t kind = ((0x181u >> op) & 1u) ? 0u : 1u;   // 0: counted (M = X), 1: skipped (D N)
        const unsigned long long km = __builtin_amdgcn_ballot_w64(kept);
        if (km == 0ull) continue;
        if (__builtin_amdgcn_ballot_w64(kept && len > (1u << 22)) != 0ull) { overflow = true; return 0u; }
        const unsigned long long nm = __builtin_amdgcn_ballot_w64(kept && kind == 1u);
        const unsigned long long pm = km & below;                // kept lanes before this one
        uint32_t pk = open_kind;
        if (pm != 0ull) pk = (uint32_t)((nm >> (63 - __clzll((long long)pm))) & 1ull);
        const bool head = kept && kind != pk;
        const unsigned long long hm = __builtin_amdgcn_ballot_w64(head);
        const uint32_t kl = kept ? len : 0u;
        const uint32_t S = (uint32_t)wave_inclusive_scan((int)kl);   // <= 64 * 2^22
        const uint32_t E = S - kl;
        const uint32_t gtot = (uint32_t)__builtin_amdgcn_readlane((int)S, 63);
        if (hm == 0ull) {                                        // the open run goes on
            open_len += gtot;                                    // both <= 2^28: no wrap
            if (open_len > LEN_MAX) { overflow = true; return 0u; }
            continue;
        }
        const bool open_exists = open_kind != 2u;
        const unsigned long long hb = hm & below;                // heads before this lane
        const uint32_t q = (uint32_t)__popcll(hb);
        const int ph = hb != 0ull ? 63 - __clzll((long long)hb) : 0;
        const uint32_t Eph = (uint32_t)__shfl((int)E, ph, 64);   // where the run this head closes began
        const uint32_t tot = q == 0u ? open_len + E : E - Eph;
        const bool closes = head && (q != 0u || open_exists);
        if (__builtin_amdgcn_ballot_w64(closes && tot > LEN_MAX) != 0ull) { overflow = true; return 0u; }
        if (WRITE && closes)                                     // kinds alternate: the closed run is of the other kind
            out[w + q - (open_exists ? 0u : 1u)] = (tot << 4) | (kind == 0u ? 3u : 0u);
        const int lh = __builtin_amdgcn_readfirstlane(63 - __clzll((long long)hm));
        open_kind = (uint32_t)__builtin_amdgcn_readlane((int)kind, lh);
        open_len = gtot - (uint32_t)__builtin_amdgcn_readlane((int)E, lh);
        w += (uint32_t)__popcll(hm) - (open_exists ? 0u : 1u);
    }
    if (open_kind == 0u) {                                       // a trailing N run is dropped
        if (WRITE && lane == 0) out[w] = open_len << 4;
        ++w;
    }
    return w;
}

constexpr uint32_t WAVE_WALK_MIN = 24;     // reads with more ops than this are walked by the whole wave

// N0: what no other kernel of the batch writes: the four record words past a contig's last read (the tile kernel
// loads 16 bytes per lane), the status word, the offset of a contig without reads.  One thread per contig.
__global__ __launch_bounds__(256) void gd_norm_init_kernel(NormBatch B)
{
    const uint32_t ji = blockIdx.x * 256u + threadIdx.x;
    if (ji >= B.n_jobs) return;
    const NormJob j = B.jobs[ji];
    for (uint32_t k = 0; k < 4u; ++k) j.rec[j.n_reads + k] = 0u;
    *j.status = 0u;
    if (j.n_reads == 0u) { j.noff[0] = 0u; *j.total = 0u; }
}

// N1: count.  One wave per unit of 64 consecutive reads.
__global__ __launch_bounds__(256) void gd_norm_count_kernel(NormBatch B)
{
    const int lane = threadIdx.x & 63;
    const uint32_t gunit = blockIdx.x * 4u + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (gunit >= B.n_units) return;
    const uint32_t ji = (uint32_t)__builtin_amdgcn_readfirstlane((int)batch_find(B.ubeg, B.n_jobs, gunit));
    const NormJob j = B.jobs[ji];
    const uint32_t unit = gunit - B.ubeg[ji];
    const uint32_t r = unit * 64u + (uint32_t)lane;
    uint32_t cnt = 0, o0 = 0, n = 0;
    if (r < j.n_reads) { o0 = j.off[r]; n = j.off[r + 1] - o0; }
    bool serial = n <= WAVE_WALK_MIN;
    unsigned long long todo = __builtin_amdgcn_ballot_w64(!serial);
    while (todo != 0ull) {                                    // long reads: the wave walks one at a time
        const int jl = __ffsll((long long)todo) - 1;
        todo &= todo - 1ull;
        const uint32_t oj = (uint32_t)__builtin_amdgcn_readlane((int)o0, jl);
        const uint32_t nj = (uint32_t)__builtin_amdgcn_readlane((int)n, jl);
        bool ovf;
        const uint32_t cw = wave_canonical<false>(j.cigar + oj, nj, lane, nullptr, ovf);
        if (lane == jl) { cnt = cw; serial = ovf; }
    }
    if (serial && n != 0u) cnt = canonical_walk(j.cigar + o0, n, [](uint32_t, uint32_t) {});
    const uint32_t incl = (uint32_t)wave_inclusive_scan((int)cnt);
    if (r < j.n_reads) {
        j.noff[r] = incl - cnt;                             // offset inside the unit, completed by N3
        const uint32_t f = j.flag[r], q = j.mapq[r];
        const bool fits = f <= 0xfffu && cnt < REC_NMAX;
        j.rec[r] = ((f & 0xfffu) << 20) | (q << 12) | (fits ? cnt : REC_NMAX);
        if (!fits) atomicOr(j.status, 1u);
    }
    if (lane == 63) j.unit[unit] = incl;
}

#endif  // GD_WITH_CANONICAL

// N2: exclusive scan of the unit totals, in place; v[n] = grand total.  One workgroup.
__global__ __launch_bounds__(1024) void gd_unit_scan_kernel(uint32_t* __restrict__ v, uint32_t n)
{
    __shared__ uint32_t s_part[1024];
    const uint32_t tid = threadIdx.x;
    const uint32_t per = (n + 1023u) / 1024u;
    const uint32_t b = tid * per < n ? tid * per : n, e = b + per < n ? b + per : n;
    uint32_t sum = 0;
    for (uint32_t i = b; i < e; ++i) sum += v[i];
    s_part[tid] = sum;
    __syncthreads();
    if (tid < 64) {                                           // 1024 partials: 16 per lane of one wave
        uint32_t loc = 0;
        for (int k = 0; k < 16; ++k) loc += s_part[tid * 16 + k];
        const uint32_t incl = (uint32_t)wave_inclusive_scan((int)loc);
        uint32_t run = incl - loc;
        for (int k = 0; k < 16; ++k) { const uint32_t t = s_part[tid * 16 + k]; s_part[tid * 16 + k] = run; run += t; }
        if (tid == 63) v[n] = incl;
    }
    __syncthreads();
    uint32_t run = s_part[tid];
    for (uint32_t i = b; i < e; ++i) { const uint32_t t = v[i]; v[i] = run; run += t; }
}

// The same scan for large arrays, three launches: totals of blocks of SCAN_BLOCK elements (SB1), the scan of
// those totals by gd_unit_scan_kernel, then every block scans itself from its offset (SB2).  One workgroup alone
// took 0.69 ms for chr1's 780 k units -- 17 of the 28 ms a genome's normalisation took.
constexpr uint32_t SCAN_BLOCK = 4096;      // elements per workgroup: 256 threads x 16

__global__ __launch_bounds__(256) void gd_scan_totals_kernel(const uint32_t* __restrict__ v, uint32_t n,
                                                             uint32_t* __restrict__ btot)
{
    __shared__ uint32_t s_w[4];
    const uint32_t b0 = blockIdx.x * SCAN_BLOCK + threadIdx.x * 16u;
    uint32_t sum = 0;
#pragma unroll
    for (uint32_t k = 0; k < 16u; ++k) sum += b0 + k < n ? v[b0 + k] : 0u;
    const uint32_t incl = (uint32_t)wave_inclusive_scan((int)sum);
    if ((threadIdx.x & 63u) == 63u) s_w[threadIdx.x >> 6] = incl;
    __syncthreads();
    if (threadIdx.x == 0) btot[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

__global__ __launch_bounds__(256) void gd_scan_apply_kernel(uint32_t* __restrict__ v, uint32_t n,
                                                            const uint32_t* __restrict__ boff, uint32_t n_blocks)
{
    __shared__ uint32_t s_w[4];
    const uint32_t b0 = blockIdx.x * SCAN_BLOCK + threadIdx.x * 16u;
    uint32_t x[16];
    uint32_t sum = 0;
#pragma unroll
    for (uint32_t k = 0; k < 16u; ++k) { x[k] = b0 + k < n ? v[b0 + k] : 0u; sum += x[k]; }
    const uint32_t incl = (uint32_t)wave_inclusive_scan((int)sum);
    const uint32_t wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63u) == 63u) s_w[wv] = incl;
    __syncthreads();
    uint32_t run = boff[blockIdx.x] + incl - sum;
    for (uint32_t w = 0; w < wv; ++w) run += s_w[w];
#pragma unroll
    for (uint32_t k = 0; k < 16u; ++k) { if (b0 + k < n) v[b0 + k] = run; run += x[k]; }
    if (blockIdx.x == 0 && threadIdx.x == 0) v[n] = boff[n_blocks];    // grand total (written last by nobody else: slot n is outside every block)
}

#ifdef GD_WITH_CANONICAL
// N3: write.  Same shape as N1.
__global__ __launch_bounds__(256) void gd_norm_write_kernel(NormBatch B)
{
    const int lane = threadIdx.x & 63;
    const uint32_t gunit = blockIdx.x * 4u + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (gunit >= B.n_units) return;
    const uint32_t ji = (uint32_t)__builtin_amdgcn_readfirstlane((int)batch_find(B.ubeg, B.n_jobs, gunit));
    const NormJob j = B.jobs[ji];
    const uint32_t unit = gunit - B.ubeg[ji];
    const uint32_t r = unit * 64u + (uint32_t)lane;
    const bool valid = r < j.n_reads;
    uint32_t dst0 = 0, o0 = 0, n = 0;
    if (valid) {
        const uint32_t ubase = j.unit[0];
        dst0 = j.unit[unit] - ubase + j.noff[r];
        j.noff[r] = dst0;
        if (r + 1u == j.n_reads) {
            const uint32_t tot = j.unit[j.n_units] - ubase;
            j.noff[j.n_reads] = tot;
            *j.total = tot;
        }
        o0 = j.off[r]; n = j.off[r + 1] - o0;
    }
    bool serial = n <= WAVE_WALK_MIN;
    unsigned long long todo = __builtin_amdgcn_ballot_w64(!serial);
    while (todo != 0ull) {
        const int jl = __ffsll((long long)todo) - 1;
        todo &= todo - 1ull;
        const uint32_t oj = (uint32_t)__builtin_amdgcn_readlane((int)o0, jl);
        const uint32_t nj = (uint32_t)__builtin_amdgcn_readlane((int)n, jl);
        const uint32_t dj = (uint32_t)__builtin_amdgcn_readlane((int)dst0, jl);
        bool ovf;
        (void)wave_canonical<true>(j.cigar + oj, nj, lane, j.ncig + dj, ovf);
        if (lane == jl) serial = ovf;                         // (what it wrote before noticing is overwritten below)
    }
    if (serial && n != 0u) {
        uint32_t* out = j.ncig + dst0;
        uint32_t w = 0;
        canonical_walk(j.cigar + o0, n, [&](uint32_t op, uint32_t len) { out[w++] = (len << 4) | op; });
    }
}

// P: the position index of a contig's (coordinate sorted) records, built with the canonical CIGARs when the
// records arrive: pidx[k] = first read with pos >= 64 k, for k = 0 .. (length >> 6) + 1.  gd_prep_kernel
// looks a tile's read range up in it instead of searching `pos` (two dependent chains of ~15 loads per tile,
// 0.29 ms per genome: 7 % of a step).  4 bytes per 64 reference positions (194 MB for hg19).  One thread per
// entry, a plain binary search each: latency bound, but wide, and paid once per ingest.
__global__ __launch_bounds__(256) void gd_pidx_kernel(NormBatch B)
{
    const uint32_t g = blockIdx.x * 256u + threadIdx.x;
    // the contig of this entry: one scalar search per wave when the whole wave lies inside one contig
    const uint32_t g0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(g & ~63u));
    if (g0 >= B.n_idx) return;
    uint32_t ji = (uint32_t)__builtin_amdgcn_readfirstlane((int)batch_find(B.ibeg, B.n_jobs, g0));
    if (g0 + 63u >= B.ibeg[ji + 1]) {
        if (g >= B.n_idx) return;
        ji = batch_find(B.ibeg, B.n_jobs, g);
    }
    const int32_t* __restrict__ const pos = B.jobs[ji].pos;
    uint32_t* __restrict__ const pidx = B.jobs[ji].pidx;
    const uint32_t n_reads = B.jobs[ji].n_reads;
    const uint32_t k = g - B.ibeg[ji];
    const uint64_t key = (uint64_t)k << 6;
    uint32_t lo = 0, hi = n_reads;
    if (key > 0x7fffffffull) lo = n_reads;              // every position is below it
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (pos[mid] < (int32_t)key) lo = mid + 1; else hi = mid;
    }
    pidx[k] = lo;
}

// ---------------------------------------------------------------------------------------------------------
// The same three results -- canonical CIGARs, record words, position index -- in ONE pass (gd_normalize's kernel;
// the count / scan / write / index launches above remain as GD_OPT_FUSED_NORMALIZE = 0 and as the cross-check).
// A workgroup takes rows x 256 consecutive reads of one contig (rows = 4 for short reads: 1024 reads, ~4.5 KB of ops):
//   * every array is read ONCE, coalesced: a read's two CSR offsets, flag, MAPQ, position and the position of the read
//     before it; the workgroup's ops are staged in LDS (up to NF_CAP; a workgroup whose reads hold more -- long reads --
//     walks them from memory, a wave per long read, as the two-pass kernels do);
//   * a lane counts its reads' canonical ops from LDS, wave scans + 16 partial sums give the offsets inside the
//     workgroup, and the workgroup's base comes from a DECOUPLED LOOK-BACK over the workgroups before it (status word =
//     flag | total; a contig's first
//     workgroup starts its own chain) -- no count pass, no scan launch;
//   * the ops are walked a second time FROM LDS and written, with the record words and offsets;
//   * the position index falls out of the positions already loaded: read r is the first with pos >= 64 k for every k in
//     (pos[r - 1] / 64, pos[r] / 64] -- usually none or one entry (12.8 reads per 64 positions at 30x); long gaps and
//     the tail after the last read are filled by the whole wave.
// Algorithmic bytes: 11 + 4 + 4 per read in (offsets twice from cache), 8 per read + 4 per canonical op out, 4 per 64
// positions: what synth.normalise_bytes counts.
constexpr uint32_t NF_CAP = 4096;          // staged ops per workgroup (16 KB)
constexpr int NF_MAXROWS = 4;

__global__ __launch_bounds__(256, 8) void gd_norm_fused_kernel(NormBatch B)
{
    __shared__ uint32_t s_ops[NF_CAP];
    __shared__ uint32_t s_part[NF_MAXROWS * 4 + 1];
    __shared__ uint32_t s_blk;
    __shared__ uint32_t s_base;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Workgroups look back at smaller ids only, and every XCD starts its share of the grid (ids k, k + 8, ...) in
    // order: the smallest unfinished id is always resident, so the chain cannot stall.  (A ticket -- one atomic on
    // one address per workgroup, 6*10^5 per genome -- was as slow as the two-pass kernels all by itself.)
    const uint32_t blk = blockIdx.x;
    (void)s_blk;
    if (blk >= B.n_blocks) return;
    const uint32_t ji = (uint32_t)__builtin_amdgcn_readfirstlane((int)batch_find(B.bbeg, B.n_jobs, blk));
    const NormJob j = B.jobs[ji];
    const uint32_t rows = j.rows;
    const uint32_t lb = blk - B.bbeg[ji];                       // workgroup number inside the contig
    const uint32_t r0 = lb * rows * 256u;
    const uint32_t r_end = r0 + rows * 256u < j.n_reads ? r0 + rows * 256u : j.n_reads;
    const uint32_t ob = j.off[r0], oe = j.off[r_end];
    const bool staged = oe - ob <= NF_CAP;

    // ---- loads: one round trip ----------------------------------------------------------------------
    uint32_t o0[NF_MAXROWS], n[NF_MAXROWS], fl[NF_MAXROWS], mq[NF_MAXROWS];
    int32_t p[NF_MAXROWS], q[NF_MAXROWS];
#pragma unroll
    for (int w = 0; w < NF_MAXROWS; ++w) {
        const uint32_t r = r0 + (uint32_t)w * 256u + (uint32_t)tid;
        o0[w] = 0; n[w] = 0; fl[w] = 0; mq[w] = 0; p[w] = 0; q[w] = -1;
        if ((uint32_t)w < rows && r < r_end) {
            o0[w] = j.off[r];
            n[w] = j.off[r + 1] - o0[w];
            fl[w] = j.flag[r];
            mq[w] = j.mapq[r];
            p[w] = j.pos[r];
            if (r) q[w] = j.pos[r - 1];
        }
    }
    if (staged)
        for (uint32_t i = (uint32_t)tid; i < oe - ob; i += 256u) s_ops[i] = j.cigar[ob + i];

    // ---- the position index, from the positions just loaded ------------------------------------------
#pragma unroll
    for (int w = 0; w < NF_MAXROWS; ++w) {
        const uint32_t r = r0 + (uint32_t)w * 256u + (uint32_t)tid;
        const bool valid = (uint32_t)w < rows && r < r_end;
        // entries (q >> 6, p >> 6] get r; the contig's last read also fills the tail with n_reads
        uint32_t k_lo = 0, k_hi = 0;                               // [k_lo, k_hi)
        if (valid) {
            const int32_t pp = p[w] > 0 ? p[w] : 0, qq = q[w];
            k_lo = qq < 0 ? 0u : ((uint32_t)qq >> 6) + 1u;
            k_hi = ((uint32_t)pp >> 6) + 1u;
            k_hi = k_hi < j.n_idx ? k_hi : j.n_idx;
            if (k_hi < k_lo) k_hi = k_lo;
        }
        const bool last = valid && r + 1u == j.n_reads;
        uint32_t t_lo = 0, t_hi = 0;                               // the tail [t_lo, t_hi) := n_reads
        if (last) { t_lo = k_hi > k_lo ? k_hi : k_lo; t_hi = j.n_idx; if (t_lo > t_hi) t_lo = t_hi; }
        const uint32_t cnt = k_hi - k_lo;
        for (uint32_t k = 0; k < cnt && k < 4u; ++k) j.pidx[k_lo + k] = r;       // the usual case: none or one
        unsigned long long big = __builtin_amdgcn_ballot_w64(cnt > 4u || t_hi > t_lo);
        while (big != 0ull) {                                      // a gap in the coverage, or the tail: the whole wave
            const int l = __ffsll((long long)big) - 1;
            big &= big - 1ull;
            const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)k_lo, l) + 4u, e = (uint32_t)__builtin_amdgcn_readlane((int)k_hi, l);
            const uint32_t rr = (uint32_t)__builtin_amdgcn_readlane((int)r, l);
            for (uint32_t k = a + (uint32_t)lane; k < e; k += 64u) j.pidx[k] = rr;
            const uint32_t ta = (uint32_t)__builtin_amdgcn_readlane((int)t_lo, l), te = (uint32_t)__builtin_amdgcn_readlane((int)t_hi, l);
            for (uint32_t k = ta + (uint32_t)lane; k < te; k += 64u) j.pidx[k] = j.n_reads;
        }
    }
    __syncthreads();                                               // the staged ops are in LDS

    // ---- count --------------------------------------------------------------------------------------
    uint32_t cnt[NF_MAXROWS];
    bool serial[NF_MAXROWS];
#pragma unroll
    for (int w = 0; w < NF_MAXROWS; ++w) {
        cnt[w] = 0;
        serial[w] = true;
        if ((uint32_t)w >= rows) continue;                         // (uniform)
        if (staged) {
            if (n[w] != 0u) cnt[w] = canonical_walk(s_ops + (o0[w] - ob), n[w], [](uint32_t, uint32_t) {});
        } else {
            serial[w] = n[w] <= WAVE_WALK_MIN;
            unsigned long long todo = __builtin_amdgcn_ballot_w64(!serial[w]);
            while (todo != 0ull) {                                 // long reads: the wave walks one at a time
                const int jl = __ffsll((long long)todo) - 1;
                todo &= todo - 1ull;
                const uint32_t oj = (uint32_t)__builtin_amdgcn_readlane((int)o0[w], jl);
                const uint32_t nj = (uint32_t)__builtin_amdgcn_readlane((int)n[w], jl);
                bool ovf;
                const uint32_t cw = wave_canonical<false>(j.cigar + oj, nj, lane, nullptr, ovf);
                if (lane == jl) { cnt[w] = cw; serial[w] = ovf; }
            }
            if (serial[w] && n[w] != 0u) cnt[w] = canonical_walk(j.cigar + o0[w], n[w], [](uint32_t, uint32_t) {});
        }
    }
    // ---- offsets inside the workgroup: reads are numbered row by row ---------------------------------
    uint32_t incl[NF_MAXROWS];
#pragma unroll
    for (int w = 0; w < NF_MAXROWS; ++w) {
        incl[w] = (uint32_t)wave_inclusive_scan((int)cnt[w]);
        if (lane == 63) s_part[w * 4 + wv] = incl[w];
    }
    __syncthreads();
    uint32_t before[NF_MAXROWS], total = 0;
    {
        uint32_t run = 0;
#pragma unroll
        for (int w = 0; w < NF_MAXROWS; ++w) {
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                if (v == wv) before[w] = run;
                run += s_part[w * 4 + v];
            }
        }
        total = run;
    }
    // ---- the workgroup's base: decoupled look-back (wave 0) -------------------------------------------
    constexpr unsigned long long ST_AGG = 1ull << 32, ST_INC = 2ull << 32;
    if (wv == 0) {
        uint32_t prefix = 0;
        if (lb == 0u) {
            if (lane == 0) __hip_atomic_store(&B.bstat[blk], ST_INC | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            if (lane == 0) __hip_atomic_store(&B.bstat[blk], ST_AGG | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t first = B.bbeg[ji];
            int64_t look = (int64_t)blk - 1;
            uint32_t spins = 0;
            for (;;) {
                const int64_t idx = look - lane;
                unsigned long long st = ST_INC;                    // before the contig's first workgroup: inclusive 0
                if (idx >= (int64_t)first) st = __hip_atomic_load(&B.bstat[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t flag = (uint32_t)(st >> 32);
                const uint32_t val = (uint32_t)st;
                const unsigned long long ready = __builtin_amdgcn_ballot_w64(flag != 0u);
                const unsigned long long incm = __builtin_amdgcn_ballot_w64(flag == 2u);
                if (incm != 0ull) {
                    const int fi = __ffsll((long long)incm) - 1;
                    const unsigned long long need = fi == 63 ? ~0ull : ((1ull << (fi + 1)) - 1ull);
                    if ((ready & need) == need) {
                        prefix += (uint32_t)__builtin_amdgcn_readlane(wave_inclusive_scan((int)(lane <= fi ? val : 0u)), 63);
                        break;
                    }
                } else if (ready == ~0ull) {
                    prefix += (uint32_t)__builtin_amdgcn_readlane(wave_inclusive_scan((int)val), 63);
                    look -= 64;
                    continue;
                }
                if (++spins > (1u << 24)) {                        // never expected: report, do not hang
                    if (lane == 0) atomicMax(&B.ticket[1], 1u);
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
            if (lane == 0) __hip_atomic_store(&B.bstat[blk], ST_INC | (prefix + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (lane == 0) s_base = prefix;
    }
    __syncthreads();
    const uint32_t base = s_base;

    // ---- write: offsets, record words, canonical ops ---------------------------------------------------
    uint32_t bad = 0;
#pragma unroll
    for (int w = 0; w < NF_MAXROWS; ++w) {
        if ((uint32_t)w >= rows) continue;
        const uint32_t r = r0 + (uint32_t)w * 256u + (uint32_t)tid;
        const bool valid = r < r_end;
        const uint32_t dst0 = base + before[w] + incl[w] - cnt[w];
        if (valid) {
            j.noff[r] = dst0;
            const bool fits = fl[w] <= 0xfffu && cnt[w] < REC_NMAX;
            j.rec[r] = ((fl[w] & 0xfffu) << 20) | (mq[w] << 12) | (fits ? cnt[w] : REC_NMAX);
            bad |= fits ? 0u : 1u;
        }
        if (staged) {
            if (n[w] != 0u) {
                uint32_t* out = j.ncig + dst0;
                uint32_t k = 0;
                canonical_walk(s_ops + (o0[w] - ob), n[w], [&](uint32_t op, uint32_t len) { out[k++] = (len << 4) | op; });
            }
        } else {
            bool ser = n[w] <= WAVE_WALK_MIN;
            unsigned long long todo = __builtin_amdgcn_ballot_w64(!ser);
            while (todo != 0ull) {
                const int jl = __ffsll((long long)todo) - 1;
                todo &= todo - 1ull;
                const uint32_t oj = (uint32_t)__builtin_amdgcn_readlane((int)o0[w], jl);
                const uint32_t nj = (uint32_t)__builtin_amdgcn_readlane((int)n[w], jl);
                const uint32_t dj = (uint32_t)__builtin_amdgcn_readlane((int)dst0, jl);
                bool ovf;
                (void)wave_canonical<true>(j.cigar + oj, nj, lane, j.ncig + dj, ovf);
                if (lane == jl) ser = ovf;
            }
            if (ser && n[w] != 0u) {
                uint32_t* out = j.ncig + dst0;
                uint32_t k = 0;
                canonical_walk(j.cigar + o0[w], n[w], [&](uint32_t op, uint32_t len) { out[k++] = (len << 4) | op; });
            }
        }
    }
    if (__builtin_amdgcn_ballot_w64(bad != 0u) != 0ull && lane == 0) atomicOr(j.status, 1u);
    if (r_end == j.n_reads && tid == 0) {                          // the contig's last workgroup
        j.noff[j.n_reads] = base + total;
        *j.total = base + total;
    }
}

#endif  // GD_WITH_CANONICAL

// What every way into the engine runs over records once they are resident (gd_adopt_device: the caller's own arrays;
// gd_commit: a staged block that has landed; the device BAM read: a contig the record walk has written) -- ONE pass over
// pos / CSR offsets / ops of the reads [r0, r1) of a contig:
//   * the checks a host block gets in gd_commit (check != 0).  out[0] bits: 0 positions out of order, 1 CSR offsets
//     decreasing / not starting at 0 / ending past the ops, 2 a negative position;
//   * the position index the prep kernel would otherwise search for: ridx[k] = first read with pos >= 64 k, written
//     for every k in (pos[r - 1] >> 6, pos[r] >> 6] by read r (k <= pos[r1 - 1] >> 6 is all that exists afterwards:
//     gd_prep_kernel reads anything above ContigDev::pidx_last as n_reads, so blocks append without a tail to redo);
//   * the largest reference span of ANY record (out[1], atomicMax; unfiltered, so never below what the tile kernel
//     will measure on the kept ones): the first gd_compute starts with the right look-back instead of learning it.
//     Reads of more than 64 ops are not walked here (out[1] = INT_MAX: unknown -- the default look-back and its
//     verification take over; long-read data takes the long-read path, which measures spans itself);
//   * out[2] = pos[r1 - 1].
struct IndexJob {
    const int32_t*  pos;
    const uint32_t* off;
    const uint32_t* cigar;
    uint32_t* ridx;           // null: no index wanted
    uint32_t n_idx;           // its entries ((length >> 6) + 2); records placed past the contig's end write none
    uint32_t* out;            // [3]
    uint32_t r0, r1;
    uint32_t n_reads_total;   // records of the contig once this block is in (offset checks at its last record)
    uint32_t n_ops_total;
    int32_t  prev_pos;        // pos[r0 - 1]; r0 == 0: -1
    uint32_t check;
    uint32_t walk_ops;        // 0: spans not measured (out[1] untouched)
};

__global__ __launch_bounds__(256) void gd_index_records_kernel(IndexJob j)
{
    const uint32_t r = j.r0 + blockIdx.x * 256u + threadIdx.x;
    const bool in = r < j.r1;
    uint32_t bad = 0;
    int32_t ka = 0, kb = -1;
    int32_t span = 0;
    if (in) {
        const int32_t p = j.pos[r];
        const int32_t q = r > j.r0 ? j.pos[r - 1u] : j.prev_pos;
        if (q > p) bad |= 1u;
        if (p < 0) bad |= 4u;
        const uint32_t a = j.off[r], b = j.off[r + 1u];
        if (a > b || (r == 0u && a != 0u) || (r + 1u == j.n_reads_total && b > j.n_ops_total)) bad |= 2u;
        if (bad == 0u) { ka = (q >> 6) + 1; kb = p >> 6; kb = kb < (int32_t)j.n_idx ? kb : (int32_t)j.n_idx - 1; }        // (q = -1 in front of the first record: from entry 0)
        if (j.walk_ops && bad == 0u && b <= j.n_ops_total) {
            if (b - a > 64u) span = 0x7fffffff;
            else {
                uint32_t sp = 0;
                for (uint32_t o = a; o < b; ++o) {
                    const uint32_t op = j.cigar[o];
                    sp += ((0x18du >> (op & 15u)) & 1u) ? (op >> 4) : 0u;     // M D N = X consume the reference
                    sp = sp > 0x7fffffffu ? 0x7fffffffu : sp;
                }
                span = (int32_t)sp;
            }
        }
        if (r + 1u == j.r1) j.out[2] = (uint32_t)p;
    }
    if (j.ridx) {
        // nearly every read owns zero or one entry; a read behind a gap (a centromere: 10^5 entries) hands it to its wave
        int32_t k = ka;
        for (int n = 0; n < 2 && k <= kb; ++n, ++k) j.ridx[k] = r;
        unsigned long long m = __builtin_amdgcn_ballot_w64(k <= kb);
        const int lane = (int)(threadIdx.x & 63u);
        while (m) {
            const int l = __builtin_ctzll(m);
            m &= m - 1ull;
            const int32_t k0 = __shfl(k, l, 64), k1 = __shfl(kb, l, 64);
            const uint32_t rr = (uint32_t)__shfl((int)r, l, 64);
            for (int32_t x = k0 + lane; x <= k1; x += 64) j.ridx[x] = rr;
        }
    }
    if (j.check) {
        const uint32_t any = (__builtin_amdgcn_ballot_w64((bad & 1u) != 0u) != 0ull ? 1u : 0u) |
                             (__builtin_amdgcn_ballot_w64((bad & 2u) != 0u) != 0ull ? 2u : 0u) |
                             (__builtin_amdgcn_ballot_w64((bad & 4u) != 0u) != 0ull ? 4u : 0u);
        if (any != 0u && (threadIdx.x & 63u) == 0u) atomicOr(j.out, any);
    }
    if (j.walk_ops) {
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) { const int32_t o = __shfl_xor(span, m, 64); span = o > span ? o : span; }
        // one address for the whole launch: ask first, most waves have nothing new to say
        if ((threadIdx.x & 63u) == 0u && span > 0 && span > (int32_t)__atomic_load_n(j.out + 1, __ATOMIC_RELAXED))
            atomicMax(reinterpret_cast<int32_t*>(j.out + 1), span);
    }
}

}  // namespace norm
}  // namespace gd
