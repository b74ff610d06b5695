// gd_sums_stream.hpp -- the sums-only output (GD_OUT_SUMS_ONLY: all that depth.bed's mean column and the
// depthwed matrix need, BASELINE.json config 4) as ONE STREAMING PASS over the canonical records.
//
// The sum of the depth over a window equals the sum over reads of their overlap with the window
// (/root/reference/depth/depth.go:181-189, :293-305: `mean` of the per-base lines of a window).  The tile
// kernel for this output (gd_tile_sums_kernel) inherited the tile machinery it no longer needs: a tile
// table, a verified look-back that re-examines 5 % of the reads, ops staged in LDS, and per-read 64-bit LDS
// atomics onto the 16 window accumulators of a tile -- 64 lanes hitting 2-3 addresses.  Nothing here is
// positional except the window index, and the records are coordinate sorted, so:
//   * one wave takes 256 CONSECUTIVE reads of one contig (4 per lane: 16 bytes of `pos`, 16 bytes of record
//     words, the reads' first canonical op each), every read exactly once: no tiles, no look-back, no LDS;
//   * a read adds to its start window and -- when it crosses the boundary -- the next; a lane folds its four
//     reads into three consecutive windows kb, kb+1, kb+2 in registers;
//   * the lanes' kb are non-decreasing, so a window's total over the wave is a difference of ONE plain wave
//     prefix sum at the segment ends (ballot of "my kb differs from the next lane's" + one bpermute): about
//     six 64-bit global atomics per accumulator and wave instead of ~330 contended LDS atomics;
//   * whatever does not fit that shape (multi-op reads: 2 %; a read longer than two windows; reads so sparse
//     that a lane spans more than three windows) adds its intervals with direct global atomics.
// Integer adds commute: the result is bit-identical to the per-base sums.
#pragma once

namespace gd {
namespace sums {

__device__ __forceinline__ void add_win(int64_t* wsum, uint32_t nwin, uint32_t k, uint32_t v)
{
    if (v != 0u && k < nwin) atomicAdd(reinterpret_cast<unsigned long long*>(&wsum[k]), (unsigned long long)v);
}

// a counted interval [s, e) of contig positions (e <= contig length < 2^31) -> every window it touches, one atomic
// each.  32-bit throughout: the next boundary nb = (k + 1) W <= s + W < 2^32, and it only advances while nb < e.
__device__ __forceinline__ void add_interval_direct(int64_t* wsum, uint32_t nwin, uint32_t W, uint32_t wm, uint32_t ws,
                                                    uint32_t s, uint32_t e)
{
    uint32_t k = div_magic(s, wm, ws);
    uint32_t nb = (k + 1u) * W;
    while (s < e) {
        const uint32_t c = (nb < e ? nb : e) - s;
        add_win(wsum, nwin, k, c);
        s += c; ++k; nb += W;
    }
}

constexpr int U = 4;                       // reads per lane and group
constexpr uint32_t GROUP = 64u * U;        // reads per group: one pass of a wave
constexpr uint32_t GPW = 16;               // consecutive groups per wave: 4096 reads

// A wave that took one group per launch slot spent its life in dependent round trips (which contig? its
// pointers? the group's first op? the records? the ops? -- a dozen for 3 KB of records, 2 TB/s at full
// occupancy).  So: a wave takes GPW consecutive groups of one contig, finds its contig with ONE vector load of
// the group table (64 contigs per ballot), and has the next group's records and op offset in flight while it
// works on the current one -- per group only the op fetch (which needs the record words) is exposed.
__global__ __launch_bounds__(256) void gd_sums_stream_kernel(Job job)
{
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
    const uint32_t* const off = c.off;
    const uint32_t* const cigar = c.cigar;
    const uint32_t fmask = job.flag_mask << 20;

    const uint32_t r_first = (unit - c.grp_beg) * (GROUP * GPW);   // first read of this wave
    uint32_t r_end = r_first + GROUP * GPW;
    r_end = r_end < n_reads ? r_end : n_reads;
    // one descriptor pair for the wave's whole range: reads past it load 0 = no ops
    typedef unsigned int v4u __attribute__((ext_vector_type(4)));
    const rsrc_t r_pos = make_rsrc(c.pos + r_first, (r_end - r_first) * 4u);
    const rsrc_t r_rec = make_rsrc(c.rec + r_first, (r_end - r_first) * 4u);

    v4u pv = __builtin_amdgcn_raw_buffer_load_b128(r_pos, lane * 16, 0, 0);
    v4u rv = __builtin_amdgcn_raw_buffer_load_b128(r_rec, lane * 16, 0, 0);
    uint32_t ob = off[r_first];                                  // first canonical op of the group (uniform)
    for (uint32_t g0 = r_first; g0 < r_end; g0 += GROUP) {
        // the next group's records and op offset: in flight while this one is processed
        const uint32_t gn = g0 + GROUP;
        v4u pvn = pv, rvn = rv;
        uint32_t obn = ob;
        if (gn < r_end) {
            pvn = __builtin_amdgcn_raw_buffer_load_b128(r_pos, (int)(gn - r_first) * 4 + lane * 16, 0, 0);
            rvn = __builtin_amdgcn_raw_buffer_load_b128(r_rec, (int)(gn - r_first) * 4 + lane * 16, 0, 0);
            obn = off[gn];
        }
        const uint32_t* const cig = cigar + ob;                    // canonical ops of this group's reads, in read order
        const int32_t p[U] = {(int32_t)pv.x, (int32_t)pv.y, (int32_t)pv.z, (int32_t)pv.w};
        const uint32_t rec[U] = {rv.x, rv.y, rv.z, rv.w};

        // where each read's ops are: prefix sum of the op counts
        uint32_t n[U], ex[U];
#pragma unroll
        for (int u = 0; u < U; ++u) n[u] = rec[u] & norm::REC_NMAX;
        ex[0] = 0; ex[1] = n[0]; ex[2] = ex[1] + n[1]; ex[3] = ex[2] + n[2];
        const uint32_t ltot = ex[3] + n[3];
        const uint32_t obase = (uint32_t)wave_inclusive_scan((int)ltot) - ltot;

        // the first op of every read (the only one of 98 % of short reads) and, for the few multi-op reads, the two
        // after it (a deletion is M N M) -- ONE round trip: a lane walking its ops load by load held the whole wave
        // for three more
        bool keep[U];
        uint32_t cg[U], cg1[U], cg2[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            keep[u] = ((rec[u] & fmask) == 0) & ((int)((rec[u] >> 12) & 0xffu) >= job.Q) & (n[u] != 0u);
            const uint32_t* q = cig + obase + ex[u];
            cg[u] = keep[u] ? q[0] : 0u;
            cg1[u] = (keep[u] & (n[u] > 1u)) ? q[1] : 0u;
            cg2[u] = (keep[u] & (n[u] > 2u)) ? q[2] : 0u;
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
        uint32_t s_[U], e_[U];
        uint32_t odd = 0;                                          // bit u: read u of this lane goes the long way
#pragma unroll
        for (int u = 0; u < U; ++u) {
            // canonical: a single op is an M of 1 <= len < 2^28.  [s, e) clipped to the contig, in 32 bits; everything
            // that is not a kept single-op read at a non-negative POS is an EMPTY interval here
            const uint32_t len = cg[u] >> 4;
            const bool one = keep[u] & (n[u] == 1u);
            const uint32_t s = p[u] > 0 ? (uint32_t)p[u] : 0u;
            const uint32_t eu = s + len;                           // < 2^31 + 2^28
            const uint32_t e = (one & (p[u] >= 0)) ? (eu < length ? eu : length) : s;
            const uint32_t l1 = s > nb0 ? s : nb0, l2 = s > nb1 ? s : nb1;
            const uint32_t h0 = e < nb0 ? e : nb0, h1 = e < nb1 ? e : nb1, h2 = e < nb2 ? e : nb2;
            a0 += (h0 > s ? h0 : s) - s;
            a1 += (h1 > l1 ? h1 : l1) - l1;
            a2 += (h2 > l2 ? h2 : l2) - l2;
            // what sticks out past the third window (a long read, sparse reads), a negative POS, several ops: the long way
            s_[u] = s > nb2 ? s : nb2; e_[u] = e;
            odd |= ((one & (e > nb2)) | (keep[u] & (n[u] > 1u)) | (one & (p[u] < 0))) ? 1u << u : 0u;
        }
        // The long way, ONE instance of the code for all four slots: nearly every group of 256 reads has a few
        // multi-op reads, in different slots of different lanes -- a block per slot ran three of the four blocks
        // with a handful of lanes each (PMC: more than half of this kernel's vector instructions).  Every lane
        // takes its next flagged read (a select over the four slots); usually one round.
        while (__builtin_amdgcn_ballot_w64(odd != 0u) != 0ull) {
            if (odd != 0u) {
                const int u = __ffs((int)odd) - 1;
                odd &= odd - 1u;
                const int32_t pu = u == 0 ? p[0] : u == 1 ? p[1] : u == 2 ? p[2] : p[3];
                const uint32_t nu = u == 0 ? n[0] : u == 1 ? n[1] : u == 2 ? n[2] : n[3];
                if (nu == 1u && pu >= 0) {
                    // the part of a single-op read past the lane's third window
                    const uint32_t su = u == 0 ? s_[0] : u == 1 ? s_[1] : u == 2 ? s_[2] : s_[3];
                    const uint32_t eu = u == 0 ? e_[0] : u == 1 ? e_[1] : u == 2 ? e_[2] : e_[3];
                    if (eu > su) add_interval_direct(wsum, nwin, W, wm, ws, su, eu);
                } else {
                    // walk the canonical ops: M (0) counted, N (3) skipped, every M interval on its own
                    const uint32_t c0 = u == 0 ? cg[0] : u == 1 ? cg[1] : u == 2 ? cg[2] : cg[3];
                    const uint32_t c1 = u == 0 ? cg1[0] : u == 1 ? cg1[1] : u == 2 ? cg1[2] : cg1[3];
                    const uint32_t c2 = u == 0 ? cg2[0] : u == 1 ? cg2[1] : u == 2 ? cg2[2] : cg2[3];
                    const uint32_t xu = u == 0 ? ex[0] : u == 1 ? ex[1] : u == 2 ? ex[2] : ex[3];
                    const uint32_t* ops = cig + obase + xu;
                    if (pu >= 0) {
                        // 32 bits: x stays below the contig length (the walk stops there), an op is < 2^28
                        uint32_t x = (uint32_t)pu;
                        for (uint32_t k = 0; k < nu && x < length; ++k) {
                            const uint32_t o = k == 0u ? c0 : k == 1u ? c1 : k == 2u ? c2 : ops[k], ol = o >> 4;
                            const uint32_t xe = x + ol;
                            if ((o & 0xfu) == 0u) add_interval_direct(wsum, nwin, W, wm, ws, x, xe < length ? xe : length);
                            x = xe;
                        }
                    } else {
                        long long x = pu;                      // a negative POS (no aligner writes one): the long form
                        for (uint32_t k = 0; k < nu; ++k) {
                            const uint32_t o = k == 0u ? c0 : k == 1u ? c1 : k == 2u ? c2 : ops[k], ol = o >> 4;
                            if ((o & 0xfu) == 0u) {
                                const long long e64 = x + (long long)ol;
                                const uint32_t s = x > 0 ? (x < (long long)length ? (uint32_t)x : length) : 0u;
                                const uint32_t e = e64 < (long long)length ? (e64 > 0 ? (uint32_t)e64 : 0u) : length;
                                if (e > s) add_interval_direct(wsum, nwin, W, wm, ws, s, e);
                            }
                            x += (long long)ol;
                        }
                    }
                }
            }
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
            add_win(wsum, nwin, kb, s0 - (first ? 0u : q0));
            add_win(wsum, nwin, kb + 1u, s1 - (first ? 0u : q1));
            add_win(wsum, nwin, kb + 2u, s2 - (first ? 0u : q2));
        }
        pv = pvn; rv = rvn; ob = obn;
    }
}

}  // namespace sums
}  // namespace gd
