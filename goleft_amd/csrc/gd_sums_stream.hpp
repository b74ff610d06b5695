// gd_sums_stream.hpp -- the sums-only output (GD_OUT_SUMS_ONLY: all that depth.bed's mean column and the
// depthwed matrix need, BASELINE.json config 4) as ONE STREAMING PASS over the records as they arrived.
//
// The sum of the depth over a window equals the sum over reads of their overlap with the window
// (/root/reference/depth/depth.go:181-189, :293-305: `mean` of the per-base lines of a window).  The tile
// kernel for this output (gd_tile_sums_kernel) inherited the tile machinery it no longer needs: a tile
// table, a verified look-back that re-examines 5 % of the reads, ops staged in LDS, and per-read 64-bit LDS
// atomics onto the 16 window accumulators of a tile -- 64 lanes hitting 2-3 addresses.  Nothing here is
// positional except the window index, and the records are coordinate sorted, so:
//   * one wave takes 256 CONSECUTIVE reads of one contig (4 per lane: 16 bytes of `pos`, 16 + 4 bytes of CSR offsets,
//     four flags, four MAPQs, the reads' first ops), every read exactly once: no tiles, no look-back;
//   * a read adds to its start window and -- when it crosses the boundary -- the next; a lane folds its four
//     reads into three consecutive windows kb, kb+1, kb+2 in registers;
//   * the lanes' kb are non-decreasing, so a window's total over the wave is a difference of ONE plain wave
//     prefix sum at the segment ends (ballot of "my kb differs from the next lane's" + one bpermute): about
//     six adds per accumulator and group instead of ~330 contended ones -- into the wave's own LDS accumulators
//     (the 256 windows from its first read's on), which go to memory with one 64-bit atomic each at its end;
//   * whatever does not fit that shape (multi-op reads: 2 %; a read that ends past the lane's third window --
//     long or sparse reads) is parked in a per-wave LDS queue and walked 64 reads at a time, one add per
//     (interval, window) into the same accumulators.
// Integer adds commute: the result is bit-identical to the per-base sums.
#pragma once

namespace gd {
namespace sums {

constexpr uint32_t NACC = 256;             // LDS window accumulators per wave

// Where a wave's sums go.  Global 64-bit atomics are the expensive ingredient of this kernel (tools/probe/
// stream_mix.hip: a read-only stream of this shape runs at 6.4 TB/s, with ~24 lane-atomics per KB of records at
// 4.8): every one is an L2 request, and the sorted records of a wave keep hitting the same ~80 windows.  So a
// wave accumulates the NACC windows from its first read's on in LDS and adds them to memory ONCE at its end;
// only a window outside that range (sparse records, tiny windows) goes to memory directly.
struct Acc {
    unsigned long long* lds;               // NACC accumulators of this wave
    uint32_t kw0;                          // window of accumulator 0
    int64_t* wsum;
    uint32_t nwin;
};

__device__ __forceinline__ void add_win(const Acc& A, uint32_t k, uint32_t v)
{
    if (v != 0u && k < A.nwin) {
        const uint32_t rel = k - A.kw0;
        if (rel < NACC) atomicAdd(&A.lds[rel], (unsigned long long)v);
        else atomicAdd(reinterpret_cast<unsigned long long*>(&A.wsum[k]), (unsigned long long)v);
    }
}

// a counted interval [s, e) of contig positions (e <= contig length < 2^31) -> every window it touches, one add
// each.  32-bit throughout: the next boundary nb = (k + 1) W <= s + W < 2^32, and it only advances while nb < e.
__device__ __forceinline__ void add_interval_direct(const Acc& A, uint32_t W, uint32_t wm, uint32_t ws, uint32_t s, uint32_t e)
{
    uint32_t k = div_magic(s, wm, ws);
    uint32_t nb = (k + 1u) * W;
    while (s < e) {
        const uint32_t c = (nb < e ? nb : e) - s;
        add_win(A, k, c);
        s += c; ++k; nb += W;
    }
}

constexpr int U = 4;                       // reads per lane and group
constexpr uint32_t GROUP = 64u * U;        // reads per group: one pass of a wave
constexpr uint32_t GPW = 16;               // consecutive groups per wave: 4096 reads (64 or 256: slower, r02o)
constexpr int SQ_CAP = 256;                // queued odd reads per wave (a round adds at most 64)
#ifndef GD_SUMS_DRAIN_AT
#define GD_SUMS_DRAIN_AT 256
#endif
constexpr int SQ_DRAIN_AT = GD_SUMS_DRAIN_AT;   // the queue is walked when it would hold more than this
#ifndef GD_SUMS_WAVES
#define GD_SUMS_WAVES 1
#endif

// pointers read from the contig table are generic to the compiler; as GLOBAL ones their loads return in order with
// the buffer loads and the waits between pipeline stages can be partial (one FLAT load outstanding forces every
// wait to vmcnt(0))
typedef const __attribute__((address_space(1))) uint32_t* gptr_u32;

// The wave's queue of odd reads {POS, index of the first op, op count}, 64 at a time: walk the ops -- any BAM op: M = X
// counted, D N advance, I S H P nothing -- every counted interval on its own.  (Inlined, at its one call site and the final
// one: as a function call it cost the kernel 25 vector registers, one resident wave per SIMD in six and a tenth of its speed.)
__device__ __forceinline__ void drain_queue(const uint4* Q, uint32_t cnt, int lane, gptr_u32 cigar, uint32_t length,
                                         Acc acc, uint32_t W, uint32_t wm, uint32_t ws)
{
    __builtin_amdgcn_wave_barrier();
    for (uint32_t i = (uint32_t)lane; i < cnt; i += 64u) {
        const uint4 it = Q[i];
        if (it.x & 0x80000000u) {
            // a read of THREE ops whose ops came with it (bit 31 of POS, which is non-negative): `M D M`, `M I M`, `S M S` -- the
            // odd reads of a short-read sample.  Fetched when the group's ops were (round 6): walked from memory at the wave's
            // end, 60 us later, every one of them was a line that had left L2 long ago -- 27 GB of 181 fetched for a cohort's
            // 154 GB of records, ALL of the kernel's over-fetch (profiles/r13k_cohort_overfetch.txt: the same kernel on data
            // without such reads fetches 1.00 x its bytes)
            uint32_t x = it.x & 0x7fffffffu;
            const uint32_t o3[3] = {it.y, it.z, it.w};
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const uint32_t o = o3[k], ol = o >> 4, op = o & 0xfu;
                if (((0x18du >> op) & 1u) && x < length) {
                    const uint32_t xe = x + ol;
                    if (((0x181u >> op) & 1u) && ol != 0u) add_interval_direct(acc, W, wm, ws, x, xe < length ? xe : length);
                    x = xe;
                }
            }
            continue;
        }
        const int32_t pu = (int32_t)it.x;
        const gptr_u32 ops = cigar + it.y;
        const uint32_t nu = it.z;
        if (pu >= 0) {
            // 32 bits: x stays below the contig length (the walk stops there), an op is < 2^28
            uint32_t x = (uint32_t)pu;
            for (uint32_t k = 0; k < nu && x < length; ++k) {
                const uint32_t o = ops[k], ol = o >> 4;
                const uint32_t op = o & 0xfu;
                if (!((0x18du >> op) & 1u)) continue;
                const uint32_t xe = x + ol;
                if (((0x181u >> op) & 1u) && ol != 0u) add_interval_direct(acc, W, wm, ws, x, xe < length ? xe : length);
                x = xe;
            }
        } else {
            long long x = pu;                                  // a negative POS (no aligner writes one): the long form
            for (uint32_t k = 0; k < nu; ++k) {
                const uint32_t o = ops[k], ol = o >> 4;
                if (!((0x18du >> (o & 0xfu)) & 1u)) continue;
                if ((0x181u >> (o & 0xfu)) & 1u) {
                    const long long e64 = x + (long long)ol;
                    const uint32_t s = x > 0 ? (x < (long long)length ? (uint32_t)x : length) : 0u;
                    const uint32_t e = e64 < (long long)length ? (e64 > 0 ? (uint32_t)e64 : 0u) : length;
                    if (e > s) add_interval_direct(acc, W, wm, ws, s, e);
                }
                x += (long long)ol;
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
}

// one group of 256 reads on its way through the wave's pipeline
struct Stage {
    typedef unsigned int v4u __attribute__((ext_vector_type(4)));
    v4u pv, rv;            // POS and CSR offsets of the lane's four reads
    uint32_t ob;           // the CSR offset after the lane's last read
    uint32_t obase;        // bit u = read u is kept
    uint32_t cg[U];        // the ONE counted op of a read that has just one (else 0xffffffff)
    unsigned int fw0, fw1, mq;   // the four flags (two per word) and MAPQs as loaded
    uint32_t g0;           // first read of the group (wave uniform)
};

// A wave that took one group per launch slot spent its life in dependent round trips (which contig? its
// pointers? the group's first op? the records? the ops? -- a dozen for 3 KB of records, 2 TB/s at full
// occupancy).  So: a wave takes GPW consecutive groups of one contig, finds its contig with ONE vector load of
// the group table (64 contigs per ballot), and runs them through a three-stage pipeline: while group g is
// worked on, the first ops of group g+1 (their addresses need that group's record words) and the records of
// group g+2 are in flight.  The loop is unrolled three times so that a stage's registers are never copied while
// their loads are outstanding (a copy is a wait).
//
// The pass reads the records AS THEY ARRIVED (pos / flag / MAPQ / CSR offsets / BAM ops): a cohort's samples are computed
// once each, and building canonical records first (rounds 2-4 kept that as an option) moved 28 bytes per read to save 3
// here (181 ms for 200 x chr1 in front of a 25 ms kernel).  A read FITS the lane's three windows when it has ONE counted
// op and nothing before it that consumes the reference: 150M, 20S130M, 100M50S, 5H145M, 70M3I, 150M2D (97 % of
// short reads); everything else takes the queue and the general op walk.
__global__ __launch_bounds__(256, GD_SUMS_WAVES) void gd_sums_stream_kernel(Job job)
{
    __shared__ uint4 s_q[4 * SQ_CAP];
    __shared__ unsigned long long s_acc[4 * NACC];
    const int lane = threadIdx.x & 63;
    const uint32_t unit = blockIdx.x * 4u + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (unit >= job.n_groups) return;
    // contig of this unit: the last c with grp_beg[c] <= unit, 64 contigs per round
    int ci = 0;
    for (int base = 0; base < job.n_ctgs; base += 64) {
        const int k = base + lane;
        const bool le = k < job.n_ctgs && job.ctgs[k].grp_beg <= unit;
        const int cnt = __popcll(__builtin_amdgcn_ballot_w64(le));
        ci = base + cnt - 1;
        if (cnt < 64) break;                                     // grp_beg is non-decreasing
    }
    ci = __builtin_amdgcn_readfirstlane(ci);
    const ContigDev& c = job.ctgs[ci];
    const uint32_t n_reads = c.n_reads;
    const uint32_t length = (uint32_t)c.length;
    const uint32_t W = (uint32_t)job.W, wm = job.w_magic, ws = job.w_shift;
    const uint32_t nwin = div_magic(length - 1u, wm, ws) + 1u;     // length >= 1: a contig with reads
    int64_t* const wsum = job.win_sum + c.win_off;
    const gptr_u32 off = (gptr_u32)c.off;
    const gptr_u32 cigar = (gptr_u32)c.cigar;
    const int Q_ = job.Q;

    const uint32_t r_first = (unit - c.grp_beg) * (GROUP * GPW);   // first read of this wave
    uint32_t r_end = r_first + GROUP * GPW;
    r_end = r_end < n_reads ? r_end : n_reads;
    // one descriptor pair for the wave's whole range: reads past it load 0 = no ops
    const rsrc_t r_pos = make_rsrc(c.pos + r_first, (r_end - r_first) * 4u);
    const rsrc_t r_rec = make_rsrc(c.off + r_first, (r_end - r_first + 1u) * 4u);       // (+ the end of the last read)
    // flags (16 bits) and MAPQs (8 bits), four per lane in one load each; the ranges are rounded up to whole
    // dwords (at most 2 / 3 bytes past the wave's last read, inside the same aligned word: gd_tile_fast.hpp)
    const rsrc_t r_flag = make_rsrc((const void*)(c.flag + r_first), ((r_end - r_first + 1u) & ~1u) * 2u);
    const rsrc_t r_mapq = make_rsrc((const void*)(c.mapq + r_first), ((r_end - r_first + 3u) & ~3u));

    uint4* const Q = &s_q[(threadIdx.x >> 6) * SQ_CAP];
    uint32_t qn = 0;                                             // queued odd reads (wave uniform)
    // the wave's LDS accumulators: the NACC windows from its first read's on (sorted records: nothing before it;
    // kw0 is set when the first records have arrived)
    Acc acc;
    acc.lds = &s_acc[(threadIdx.x >> 6) * NACC];
    acc.kw0 = 0;
    acc.wsum = wsum;
    acc.nwin = nwin;
#pragma unroll
    for (uint32_t i = 0; i < NACC; i += 64u) acc.lds[i + (uint32_t)lane] = 0ull;

    // stage 1: the records of group g (nothing past the wave's range: zeros = reads without ops)
    auto load_records = [&](Stage& S, uint32_t g) {
        S.pv = __builtin_amdgcn_raw_buffer_load_b128(r_pos, (int)(g - r_first) * 4 + lane * 16, 0, 0);
        S.rv = __builtin_amdgcn_raw_buffer_load_b128(r_rec, (int)(g - r_first) * 4 + lane * 16, 0, 0);
        typedef unsigned int v2u_t __attribute__((ext_vector_type(2)));
        S.g0 = g;
        S.ob = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(r_rec, (int)(g - r_first) * 4 + lane * 16 + 16, 0, 0);
        const v2u_t fv = __builtin_amdgcn_raw_buffer_load_b64(r_flag, (int)(g - r_first) * 2 + lane * 8, 0, 0);
        S.fw0 = fv.x; S.fw1 = fv.y;
        S.mq = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(r_mapq, (int)(g - r_first) + lane * 4, 0, 0);
    };
    // stage 2: where the group's ops are (the CSR offsets say it) and the first two ops of each read
    auto fetch_ops = [&](Stage& S) {
        {
            const uint32_t o[U + 1] = {S.rv.x, S.rv.y, S.rv.z, S.rv.w, S.ob};
            uint32_t n[U];
            bool keep[U];
            const uint32_t fl[U] = {S.fw0 & 0xffffu, S.fw0 >> 16, S.fw1 & 0xffffu, S.fw1 >> 16};
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool valid = S.g0 + (uint32_t)lane * U + (uint32_t)u < r_end;
                n[u] = valid ? o[u + 1] - o[u] : 0u;
                keep[u] = ((fl[u] & job.flag_mask) == 0) & ((int)((S.mq >> (8 * u)) & 0xffu) >= Q_) & (n[u] != 0u);
            }
            typedef unsigned int v4u_t __attribute__((ext_vector_type(4)));
            typedef const __attribute__((address_space(1))) v4u_t* gptr_v4;
            // four reads of one op each: their ops lie side by side, ONE 16-byte load
            const bool four = ((n[0] & n[1] & n[2] & n[3]) == 1u) & ((n[0] | n[1] | n[2] | n[3]) == 1u);
            v4u_t o4 = {0u, 0u, 0u, 0u};
            if (four) o4 = *(gptr_v4)(cigar + o[0]);
            uint32_t c0[U] = {o4.x, o4.y, o4.z, o4.w}, c1[U] = {0u, 0u, 0u, 0u}, c2[U] = {0u, 0u, 0u, 0u};
            const int32_t pq[U] = {(int32_t)S.pv.x, (int32_t)S.pv.y, (int32_t)S.pv.z, (int32_t)S.pv.w};
            uint32_t three = 0;                                  // bit u: read u has three ops and goes to the queue WITH them
            if (__builtin_amdgcn_ballot_w64(!four) != 0ull) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const bool t3 = !four & keep[u] & (n[u] == 3u) & (pq[u] >= 0);
                    if (!four & keep[u]) c0[u] = cigar[o[u]];
                    if (!four & keep[u] & ((n[u] == 2u) | t3)) c1[u] = cigar[o[u] + 1u];
                    if (t3) c2[u] = cigar[o[u] + 2u];
                    three |= t3 ? 1u << u : 0u;
                }
            }
            // (one queue slot per lane and round, as in `work` below.  No drain here: when the queue has no room -- a wave of
            // nothing but such reads -- the reads left over stay ordinary odd reads and take `work`'s way, ops by index; the
            // drain's code exists once per `work` and once at the end, and four more copies of it cost more than they saved)
            uint32_t odd3 = three;
            while (__builtin_amdgcn_ballot_w64(odd3 != 0u) != 0ull) {
                const bool mine = odd3 != 0u;
                const int u = mine ? __ffs((int)odd3) - 1 : 0;
                const unsigned long long m = __builtin_amdgcn_ballot_w64(mine);
                const uint32_t np = (uint32_t)__popcll(m);
                if (qn + np > (uint32_t)SQ_CAP) break;
                odd3 &= odd3 - 1u;
                if (mine) {
                    const uint32_t rk = qn + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                    const uint32_t pu = (uint32_t)(u == 0 ? pq[0] : u == 1 ? pq[1] : u == 2 ? pq[2] : pq[3]);
                    const uint32_t a0 = u == 0 ? c0[0] : u == 1 ? c0[1] : u == 2 ? c0[2] : c0[3];
                    const uint32_t a1 = u == 0 ? c1[0] : u == 1 ? c1[1] : u == 2 ? c1[2] : c1[3];
                    const uint32_t a2 = u == 0 ? c2[0] : u == 1 ? c2[1] : u == 2 ? c2[2] : c2[3];
                    Q[rk] = make_uint4(pu | 0x80000000u, a0, a1, a2);
                }
                qn += np;
            }
            three &= ~odd3;
            uint32_t kb = 0;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t oa = c0[u] & 0xfu, ob2 = c1[u] & 0xfu;
                const bool ca = (0x181u >> oa) & 1u, ka = (0x18du >> oa) & 1u;       // counted (M = X), consumes (M D N = X)
                const bool cb = (n[u] == 2u) & (bool)((0x181u >> ob2) & 1u);
                // ONE counted op and nothing that consumes the reference before it: that op is the read
                uint32_t eff = 0xffffffffu;                                          // everything else: the queue
                if (n[u] <= 2u) {
                    if (ca & !cb) eff = c0[u] >> 4;                                  // M | M S | M I | M D (a trailing D covers nothing counted)
                    else if (!ka & cb) eff = c1[u] >> 4;                             // S M | H M | I M
                    else if (!ca & !cb) eff = 0u;                                    // nothing counted at all
                }
                const bool k = keep[u] & (eff != 0u) & !((three >> u) & 1u);   // (a queued three-op read is done with)
                S.cg[u] = eff == 0xffffffffu ? eff : eff << 4;
                kb |= k ? 1u << u : 0u;
            }
            S.obase = kb;
        }
    };
    // stage 3: the group's reads onto their windows
    auto work = [&](const Stage& S, uint32_t g0) {
        const int32_t p[U] = {(int32_t)S.pv.x, (int32_t)S.pv.y, (int32_t)S.pv.z, (int32_t)S.pv.w};
        const uint32_t rec[U] = {S.rv.x, S.rv.y, S.rv.z, S.rv.w};
        uint32_t n[U], ex[U];
        bool keep[U];
        {
            const uint32_t oe[U] = {S.rv.y, S.rv.z, S.rv.w, S.ob};
#pragma unroll
            for (int u = 0; u < U; ++u) {
                keep[u] = (S.obase >> u) & 1u;
                n[u] = keep[u] ? oe[u] - rec[u] : 0u;
                ex[u] = rec[u];                                    // the read's first op, an index into the contig's ops
            }
        }

        // this lane's window base: the start window of its first read (sorted records: non-decreasing over the
        // lanes; lanes past the contig's last read sort last and add nothing)
        const bool inb = g0 + (uint32_t)lane * U < r_end;
        const uint32_t kb = inb ? div_magic((uint32_t)(p[0] > 0 ? p[0] : 0), wm, ws) : 0xffffffffu;
        // the lane's three windows [kb W, nb0), [nb0, nb1), [nb1, nb2): a read's share of each is an interval overlap
        // (min, max, subtract) -- no division, no multiplication, no case analysis per read
        const uint32_t nb0 = inb ? (kb + 1u) * W : 0u;             // <= start of the lane's first read + W < 2^32
        const uint32_t nb1 = nb0 + W >= nb0 ? nb0 + W : 0xffffffffu;
        const uint32_t nb2 = nb1 + W >= nb1 ? nb1 + W : 0xffffffffu;
        uint32_t a0 = 0, a1 = 0, a2 = 0;
        uint32_t odd = 0;                                          // bit u: read u of this lane is queued
#pragma unroll
        for (int u = 0; u < U; ++u) {
            // [s, e) clipped to the contig, in 32 bits.  A kept
            // read FITS when it is one counted op of fewer than 2^22 bases (the wave's prefix sums stay below 2^32 whatever
            // the window) at a non-negative POS that ends inside the lane's three windows; every other read --
            // several ops, a long or sparse read, a negative POS -- is an EMPTY interval here and goes to the queue
            const uint32_t len = S.cg[u] >> 4;
            const uint32_t s = p[u] > 0 ? (uint32_t)p[u] : 0u;
            const uint32_t eu = s + len;                           // < 2^31 + 2^28
            const uint32_t ec = eu < length ? eu : length;
            const bool fits = keep[u] & (S.cg[u] != 0xffffffffu) & (len < (1u << 22)) & (p[u] >= 0) & (ec <= nb2);
            const uint32_t e = fits ? ec : s;
            const uint32_t l1 = s > nb0 ? s : nb0, l2 = s > nb1 ? s : nb1;
            const uint32_t h0 = e < nb0 ? e : nb0, h1 = e < nb1 ? e : nb1;
            a0 += (h0 > s ? h0 : s) - s;
            a1 += (h1 > l1 ? h1 : l1) - l1;
            a2 += (e > l2 ? e : l2) - l2;
            odd |= (keep[u] & !fits) ? 1u << u : 0u;
        }
        // Odd reads (2 % of a short-read sample) do not interrupt the stream: they are parked in the wave's queue and
        // walked 64 at a time.  Walked where they occur -- a loop per group, a handful of lanes each round -- they
        // were a quarter of this kernel's vector instructions (PMC, profiles/r02l_cohort_pmc.txt).
        // (one queue slot per lane and round: every lane parks its next flagged read, a select over the four slots --
        // ONE instance of the insertion code and of the drain)
        while (__builtin_amdgcn_ballot_w64(odd != 0u) != 0ull) {
            const bool mine = odd != 0u;
            const int u = mine ? __ffs((int)odd) - 1 : 0;
            odd &= odd - 1u;
            const unsigned long long m = __builtin_amdgcn_ballot_w64(mine);
            const uint32_t np = (uint32_t)__popcll(m);
            if (qn + np > (uint32_t)SQ_DRAIN_AT) { drain_queue(Q, qn, lane, cigar, length, acc, W, wm, ws); qn = 0; }
            if (mine) {
                const uint32_t rk = qn + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32),
                                             __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                const int32_t pu = u == 0 ? p[0] : u == 1 ? p[1] : u == 2 ? p[2] : p[3];
                const uint32_t nu = u == 0 ? n[0] : u == 1 ? n[1] : u == 2 ? n[2] : n[3];
                const uint32_t xu = u == 0 ? ex[0] : u == 1 ? ex[1] : u == 2 ? ex[2] : ex[3];
                Q[rk] = make_uint4((uint32_t)pu, xu, nu, 0u);
            }
            qn += np;
        }

        // segment totals: kb is non-decreasing over the lanes, so a window's total is a difference of one plain
        // prefix sum taken at the segment ends
        const uint32_t kb_next = (uint32_t)__builtin_amdgcn_update_dpp((int)0xfffffffeu, (int)kb, 0x130, 0xf, 0xf, false);   // wave_shl:1
        const bool tail = inb && (lane == 63 || kb_next != kb);
        const unsigned long long tm = __builtin_amdgcn_ballot_w64(tail);
        const unsigned long long before = tm & ((1ull << lane) - 1ull);
        const int pt = before != 0ull ? 63 - __clzll((long long)before) : 0;
        const uint32_t s0 = (uint32_t)wave_inclusive_scan((int)a0);
        const uint32_t s1 = (uint32_t)wave_inclusive_scan((int)a1);
        const uint32_t s2 = (uint32_t)wave_inclusive_scan((int)a2);
        const uint32_t q0 = (uint32_t)__shfl((int)s0, pt, 64);
        const uint32_t q1 = (uint32_t)__shfl((int)s1, pt, 64);
        const uint32_t q2 = (uint32_t)__shfl((int)s2, pt, 64);
        if (tail) {
            const bool first = before == 0ull;
            add_win(acc, kb, s0 - (first ? 0u : q0));
            add_win(acc, kb + 1u, s1 - (first ? 0u : q1));
            add_win(acc, kb + 2u, s2 - (first ? 0u : q2));
        }
    };

    Stage A, B, C;
    load_records(A, r_first);
    load_records(B, r_first + GROUP);
    {
        const int32_t p0 = __builtin_amdgcn_readfirstlane((int)A.pv.x);
        acc.kw0 = div_magic((uint32_t)(p0 > 0 ? p0 : 0), wm, ws);
    }
    fetch_ops(A);
    for (uint32_t g = r_first; g < r_end; g += 3u * GROUP) {
        load_records(C, g + 2u * GROUP); fetch_ops(B); work(A, g);
        if (g + GROUP >= r_end) break;
        load_records(A, g + 3u * GROUP); fetch_ops(C); work(B, g + GROUP);
        if (g + 2u * GROUP >= r_end) break;
        load_records(B, g + 4u * GROUP); fetch_ops(A); work(C, g + 2u * GROUP);
    }
    if (qn != 0u) drain_queue(Q, qn, lane, cigar, length, acc, W, wm, ws);
    // the wave's accumulators -> memory, once
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (uint32_t i = 0; i < NACC; i += 64u) {
        const unsigned long long v = acc.lds[i + (uint32_t)lane];
        const uint32_t k = acc.kw0 + i + (uint32_t)lane;
        if (v != 0ull && k < nwin) atomicAdd(reinterpret_cast<unsigned long long*>(&wsum[k]), v);
    }
}

}  // namespace sums
}  // namespace gd
