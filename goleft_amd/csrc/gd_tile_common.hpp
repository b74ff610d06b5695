// gd_tile_common.hpp -- what the tile kernels of the per-base depth engine share: buffer descriptors,
// wavefront primitives, the generic CIGAR walk, the generic (any tile shape, clipped tiles, any depth,
// any window size) phase B and phase C.
//
// One workgroup (NT threads, NW = NT/64 waves) handles one tile of T reference positions.  Replaces, for
// the reads of one tile, the per-read CIGAR walk and per-position counting that `samtools depth`
// performs (/root/reference/depth/depth.go:45) and the per-line window / class reductions of the
// callback (depth/depth.go:293-323).
//
//   * record fields come through raw buffer loads: the descriptors are bound to the tile's read range
//     [lo,hi), so out-of-range lanes read 0 (n_ops = 0 => dropped) and no index clamping / 64-bit
//     address arithmetic is issued; the 4 slots of a lane share one VGPR offset (immediate offsets);
//   * all interval arithmetic is tile relative and pre-multiplied by 4 (LDS byte addresses); the depth
//     at t0-1 (needed for the class boundary at the tile start) lives at LDS index -1: reads that start
//     before the tile are clipped to -1, so the scan's carry-in IS that depth and no separate
//     "covers t0-1" counting exists;
//   * phase B keeps the depth in registers: per row of 256 positions one ds_read_b128, a 6-step DPP
//     scan, one global_store_dwordx4, v_add3/v_min3 window accumulation in 32 bits (exact: depth <=
//     reads of the tile < 2^22, else the 64-bit path runs), DPP wave reductions at window boundaries,
//     and a one-compare "row is all CALLABLE" shortcut.
#pragma once

#include <type_traits>

namespace gd {

typedef __amdgpu_buffer_rsrc_t rsrc_t;

// Raw buffer descriptor (gfx9 family word 3: DATA_FORMAT = 32-bit) over `bytes`
// bytes starting at p.  Loads past `bytes` return 0.
__device__ __forceinline__ rsrc_t make_rsrc(const void* p, uint32_t bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), (short)0, (int)bytes, 0x00020000);
}

__device__ __forceinline__ int wave_total(int v)     // sum over the wave, valid in every lane
{
    return __builtin_amdgcn_readlane(wave_inclusive_scan(v), 63);
}

// value of lane-1 (lane 0 receives `first`): DPP wave_shr:1
__device__ __forceinline__ int wave_prev_lane(int v, int first)
{
    return __builtin_amdgcn_update_dpp(first, v, 0x138, 0xf, 0xf, false);
}

// Wave minimum via DPP (lanes without a DPP source keep their own value).
__device__ __forceinline__ int wave_min_dpp(int v)
{
    int o;
    o = __builtin_amdgcn_update_dpp(v, v, 0x111, 0xf, 0xf, false); v = o < v ? o : v;  // row_shr:1
    o = __builtin_amdgcn_update_dpp(v, v, 0x112, 0xf, 0xf, false); v = o < v ? o : v;  // row_shr:2
    o = __builtin_amdgcn_update_dpp(v, v, 0x114, 0xf, 0xf, false); v = o < v ? o : v;  // row_shr:4
    o = __builtin_amdgcn_update_dpp(v, v, 0x118, 0xf, 0xf, false); v = o < v ? o : v;  // row_shr:8
    o = __builtin_amdgcn_update_dpp(v, v, 0x142, 0xa, 0xf, false); v = o < v ? o : v;  // row_bcast:15
    o = __builtin_amdgcn_update_dpp(v, v, 0x143, 0xc, 0xf, false); v = o < v ? o : v;  // row_bcast:31
    return __builtin_amdgcn_readlane(v, 63);
}

// Wave maximum via DPP, valid in every lane.
__device__ __forceinline__ uint32_t wave_max_dpp(uint32_t v)
{
    uint32_t o;
    o = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x111, 0xf, 0xf, false); v = o > v ? o : v;  // row_shr:1
    o = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x112, 0xf, 0xf, false); v = o > v ? o : v;  // row_shr:2
    o = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x114, 0xf, 0xf, false); v = o > v ? o : v;  // row_shr:4
    o = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x118, 0xf, 0xf, false); v = o > v ? o : v;  // row_shr:8
    o = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x142, 0xa, 0xf, false); v = o > v ? o : v;  // row_bcast:15
    o = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x143, 0xc, 0xf, false); v = o > v ? o : v;  // row_bcast:31
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// Publishes the largest reference span a wave has seen (the host re-runs when it exceeds the look-back
// and tightens the look-back when it is far below).  Gated on the value read at kernel entry: once any
// wave has published the data set's typical span, ONE compare + ballot per wave is all that is left; and a
// wave that does publish sends one atomic, not one per lane -- at kernel start thousands of workgroups
// find max_span == 0, and a quarter of a million same-address atomics was ~0.1 ms of every launch.
__device__ __forceinline__ void publish_span(int32_t* max_span, uint32_t smax, int seen0, int lane)
{
    if (__builtin_amdgcn_ballot_w64(smax > (uint32_t)seen0) != 0ull) {
        const uint32_t m = wave_max_dpp(smax);
        if (lane == 0) atomicMax(max_span, (int32_t)m);
    }
}

// LDS word at byte offset `b4` from `base` (b4 is already a multiple of 4).
__device__ __forceinline__ int32_t* lds_at(int32_t* base, int b4)
{
    return reinterpret_cast<int32_t*>(reinterpret_cast<char*>(base) + b4);
}

// +1 at the clipped start, -1 at the end of [s4, e4) (tile relative byte
// offsets, s4 < T4 guaranteed by the caller).  Index -1 (byte -4) holds the
// depth at t0-1.
__device__ __forceinline__ void mark4(int32_t* s_diff, int s4, int e4, int T4)
{
    if (e4 >= 0) {                                    // reaches t0-1 or beyond
        const int cs4 = s4 > -4 ? s4 : -4;
        atomicAdd(lds_at(s_diff, cs4), 1);
        if (e4 < T4) atomicAdd(lds_at(s_diff, e4), -1);
    }
}

constexpr uint32_t SPAN_SAT = 1u << 28;               // spans saturate here (host rejects)

// Generic CIGAR walk of one read: every M/=/X op is one interval (adjacent
// intervals simply cancel at the shared edge).  `ops` is the LDS staging area
// or global memory.  Returns the (saturated) reference span of the read.
template <typename OpPtr>
__device__ __forceinline__ uint32_t walk_cigar4(OpPtr ops, uint32_t n, int ps4, int T4, int32_t* s_diff)
{
    uint32_t span = 0;
    for (uint32_t k = 0; k < n; ++k) {
        const uint32_t cg = ops[k];
        const uint32_t op = cg & 0xf, len = cg >> 4;
        const bool counted = (0x181u >> op) & 1u;     // M = X
        const bool consumes = (0x18du >> op) & 1u;    // M D N = X
        if (counted && len != 0 && span < SPAN_SAT) {
            const int s4 = ps4 + (int)(span << 2);
            if (s4 < T4) mark4(s_diff, s4, s4 + (int)(len << 2), T4);
        }
        if (consumes) { span += len; span = span < SPAN_SAT ? span : SPAN_SAT; }
    }
    return span;
}

// Scalars of phase A.
struct PhaseA {
    const int32_t* pos; const uint16_t* flag; const uint8_t* mapq; const uint32_t* off;   // at read `lo`
    int32_t* s_diff;
    const uint32_t* s_cig;       // staged ops of the tile (index: op - clo)
    uint32_t* wq;                // this wave's queue: ps4 | o0 | n
    const uint32_t* gcig;        // the contig's CIGAR array
    uint32_t clo, nrd;
    int neg4t0, T4;
    uint32_t flag_mask;
    int Q, tid, lane;
};

// Scalars shared by the phase-B instantiations.
struct PhaseB {
    int32_t* s_diff;
    uint32_t* s_bmap; uint32_t* s_clo; uint32_t* s_chi; uint32_t* s_hasb;
    int32_t* out;            // per-base output of this tile (t0 applied)
    int64_t* wsum; int32_t* wmin;
    int t0, tlen, chunk0, carry, lane;
    int W, mincov, maxmean;
    int64_t step;
};

// Phase B pass 2 for one wave: scan rows of 256 positions, store depth, reduce
// windows, detect class boundaries.
//   FULL  every position of the tile is inside the contig (no masking)
//   WIDE  depths may reach 2^22: 64-bit window accumulation everywhere
//   ST    per-base stores: 0 plain, 1 non-temporal, 2 none (windows-only output)
template <int ROWS, bool FULL, bool WIDE, int ST>
__device__ __forceinline__ void phase_b_rows(const PhaseB& B)
{
    constexpr int BIG = 0x3fffffff;
    typedef typename std::conditional<WIDE, unsigned long long, uint32_t>::type acc_t;
    const int lane = B.lane, t0 = B.t0, tlen = B.tlen, chunk0 = B.chunk0;
    const int W = B.W;
    int carry = B.carry;                                  // depth at (row start - 1)

    const uint32_t cpos0 = (uint32_t)t0 + (uint32_t)chunk0;
    uint32_t cur_win = cpos0 / (uint32_t)W;
    const int64_t nb_abs = ((int64_t)cur_win + 1) * (int64_t)W;
    int nb = (nb_abs - t0) > BIG ? BIG : (int)(nb_abs - t0);      // next window boundary (rel)
    const uint32_t stepc = B.step > 0x7fffffffLL ? 0x7fffffffu : (uint32_t)B.step;
    const int64_t nf_abs = (int64_t)((cpos0 + stepc - 1) / stepc) * (int64_t)stepc;
    int nf = (nf_abs - t0) > BIG ? BIG : (int)(nf_abs - t0);      // next forced run break (rel)
    const int wstep = W > BIG ? BIG : W;
    const int fstep = stepc > (uint32_t)BIG ? BIG : (int)stepc;
    acc_t acc = 0;
    int mn = 0x7fffffff;
    const int lo_thr = B.mincov > 1 ? B.mincov : 1;               // depths in [lo_thr, hi_thr)
    const int hi_thr = B.maxmean > 0 ? B.maxmean : 0x7fffffff;    // are CALLABLE
    const bool has_max = B.maxmean > 0;

#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const int rb = chunk0 + r * 256;                  // row start (rel)
        const int ib = rb + lane * 4;                     // this lane's first position (rel)
        int4* dst = reinterpret_cast<int4*>(&B.out[ib]);
        if (!FULL && rb >= tlen) {
            // rows past the (clipped) tile end: keep the padded per-base array zero
            if (ST != 2) *dst = make_int4(0, 0, 0, 0);
            continue;
        }
        const int4 v = *reinterpret_cast<const int4*>(&B.s_diff[ib]);
        const int x1 = v.x + v.y, x2 = x1 + v.z, x3 = x2 + v.w;
        const int incl = wave_inclusive_scan(x3);
        const int base = carry + (incl - x3);
        int d0 = base + v.x, d1 = base + x1, d2 = base + x2, d3 = base + x3;
        const int carry_before = carry;
        carry += __builtin_amdgcn_readlane(incl, 63);
        int nvalid = 4;
        if (!FULL) {
            nvalid = tlen - ib;
            nvalid = nvalid < 0 ? 0 : (nvalid > 4 ? 4 : nvalid);
            // positions at or past the contig end hold depth 0 (nothing is printed there)
            d0 = nvalid > 0 ? d0 : 0; d1 = nvalid > 1 ? d1 : 0;
            d2 = nvalid > 2 ? d2 : 0; d3 = nvalid > 3 ? d3 : 0;
        }
        if (ST == 2) {
            // windows-only output: the depth never leaves the registers
        } else if (ST == 1) {
            typedef int v4i32 __attribute__((ext_vector_type(4)));
            v4i32 dv; dv.x = d0; dv.y = d1; dv.z = d2; dv.w = d3;
            __builtin_nontemporal_store(dv, reinterpret_cast<v4i32*>(dst));
        } else {
            *dst = make_int4(d0, d1, d2, d3);
        }

        const uint32_t s4 = (uint32_t)d0 + (uint32_t)d1 + (uint32_t)d2 + (uint32_t)d3;  // depth < 2^30
        int t = d0 < d1 ? d0 : d1;
        t = d2 < t ? d2 : t;
        t = d3 < t ? d3 : t;                              // min of the lane's 4 positions

        // ---- window sum / min (depth/depth.go:181-189, :293-305) ---------
        if (FULL && nb >= rb + 256) {
            acc += s4;
            mn = t < mn ? t : mn;
        } else if (FULL && !WIDE && nb + wstep >= rb + 256) {
            // exactly one boundary in this row: split at lane granularity, fix
            // the straddling lane with scalar arithmetic
            const int rel = nb - rb;                      // 0..255
            const int L = rel >> 2, k = rel & 3;
            const bool lt = lane < L;
            const uint32_t a_old = (uint32_t)acc + (lt ? s4 : 0u);
            const int t_old = lt ? t : 0x7fffffff;
            const int m_old = t_old < mn ? t_old : mn;
            const int e0 = __builtin_amdgcn_readlane(d0, L), e1 = __builtin_amdgcn_readlane(d1, L);
            const int e2 = __builtin_amdgcn_readlane(d2, L), e3 = __builtin_amdgcn_readlane(d3, L);
            const uint32_t ps = (k > 0 ? (uint32_t)e0 : 0u) + (k > 1 ? (uint32_t)e1 : 0u) +
                                (k > 2 ? (uint32_t)e2 : 0u);
            int pm = 0x7fffffff;
            if (k > 0) pm = e0 < pm ? e0 : pm;
            if (k > 1) pm = e1 < pm ? e1 : pm;
            if (k > 2) pm = e2 < pm ? e2 : pm;
            const uint32_t qs = (uint32_t)e0 + (uint32_t)e1 + (uint32_t)e2 + (uint32_t)e3 - ps;
            int qm = e3;
            if (k <= 0) qm = e0 < qm ? e0 : qm;
            if (k <= 1) qm = e1 < qm ? e1 : qm;
            if (k <= 2) qm = e2 < qm ? e2 : qm;
            const uint32_t tot = (uint32_t)wave_total((int)a_old) + ps;   // < 2^32 (depth < 2^22)
            int m = wave_min_dpp(m_old);
            m = pm < m ? pm : m;
            if (lane == 0) {
                atomicAdd(reinterpret_cast<unsigned long long*>(&B.wsum[cur_win]),
                          (unsigned long long)tot);
                atomicMin(&B.wmin[cur_win], m);
            }
            const bool gt = lane > L;
            acc = gt ? s4 : 0u;
            mn = gt ? t : 0x7fffffff;
            if (lane == L) { acc = qs; mn = qm; }
            cur_win++;
            nb = nb + wstep > BIG ? BIG : nb + wstep;
        } else {
            // generic: any number of boundaries, clipped rows, 64-bit sums
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
                    atomicAdd(reinterpret_cast<unsigned long long*>(&B.wsum[cur_win]),
                              (unsigned long long)tot);
                    atomicMin(&B.wmin[cur_win], m);
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

        // ---- coverage class boundaries (depth/depth.go:307-323) ----------
        // carry_before is the depth just before this row.  A row whose
        // positions (and predecessor) are all CALLABLE has no boundary.
        bool noisy = __ballot(t < lo_thr) != 0ull;
        if (has_max) {
            int tx = d0 > d1 ? d0 : d1;
            tx = d2 > tx ? d2 : tx;
            tx = d3 > tx ? d3 : tx;
            noisy = noisy || __ballot(tx >= hi_thr) != 0ull;
        }
        noisy = noisy || carry_before < lo_thr || carry_before >= hi_thr;
        if (noisy || nf < rb + 256) {
            const int pl = wave_prev_lane(d3, carry_before);
            const int c0 = cov_class(d0, B.mincov, B.maxmean);
            const int c1 = cov_class(d1, B.mincov, B.maxmean);
            const int c2 = cov_class(d2, B.mincov, B.maxmean);
            const int c3 = cov_class(d3, B.mincov, B.maxmean);
            const int cp = cov_class(pl, B.mincov, B.maxmean);
            uint32_t bm = (uint32_t)(c0 != cp) | ((uint32_t)(c1 != c0) << 1) |
                          ((uint32_t)(c2 != c1) << 2) | ((uint32_t)(c3 != c2) << 3);
            while (nf < rb + 256) {                       // forced breaks (quirk Q1), incl. position 0
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
                    atomicOr(&B.s_bmap[w], bm << sh);
                    atomicOr(&B.s_clo[w], (lo & bm) << sh);
                    atomicOr(&B.s_chi[w], (hi & bm) << sh);
                }
                if (lane == 0) *B.s_hasb = 1;
            }
        }
    }
    // flush the open window segment of this wave
    if (chunk0 < tlen) {
        unsigned long long tot;
        int m;
        if (WIDE) { tot = (unsigned long long)wave_sum64((long long)acc); m = wave_min(mn); }
        else      { tot = (uint32_t)wave_total((int)(uint32_t)acc);       m = wave_min_dpp(mn); }
        if (lane == 0) {
            atomicAdd(reinterpret_cast<unsigned long long*>(&B.wsum[cur_win]), tot);
            atomicMin(&B.wmin[cur_win], m);
        }
    }
}

// Phase C (rare): tiles whose bitmap holds class boundaries compact them into a
// chunk allocated with one global atomicAdd and add their count to the
// per-SUPER-tiles group counter (ordering happens in gd_runs_order_kernel).
// Must be called by every thread of the workgroup after a barrier.
template <int T, int NT>
__device__ __forceinline__ void phase_c(const Job& job, int tile, int32_t t0, int ctg, int tid, int lane,
                                        int wv, const uint32_t* s_bmap, const uint32_t* s_clo,
                                        const uint32_t* s_chi, uint32_t* s_wcnt, const uint32_t* s_hasb,
                                        uint32_t* s_base)
{
    constexpr int NW = NT / WAVE;
    constexpr int NWORDS = T / 32;
    if (*s_hasb == 0) {
        if (tid == 0) { job.tile_cnt[tile] = 0; job.tile_off[tile] = 0; }
        return;
    }
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
        *s_base = b;
        job.tile_cnt[tile] = total;
        job.tile_off[tile] = b;
        atomicAdd(&job.super_cnt[tile / SUPER], total);
    }
    __syncthreads();
    uint32_t dst = *s_base + before + incl - cnt;
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
                job.run_chunks[dst] = make_int2(t0 + w * 32 + b, cls | (ctg << 2));
            }
            ++dst;
        }
    }
}

}  // namespace gd
