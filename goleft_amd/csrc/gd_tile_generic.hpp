// gd_tile_generic.hpp -- K1, the generic tile kernel (short-read path): any tile shape (T, NT), tiles
// clipped at a contig end, tiles deeper than one batch of reads, CIGARs in any form,
// any window size.  It runs every tile when the straight-line kernel is switched off
// (GD_OPT_FAST_KERNEL = 0), the tiles gd_prep_kernel lists as `slow` otherwise, and -- as
// gd_tile_sums_kernel -- the sums-only output.  Same algorithm and results as gd_tile_fast.hpp: phase A
// marks +1/-1 per counted interval in an LDS difference array, phase B scans, stores and reduces,
// phase C compacts class boundaries; it replaces the per-read CIGAR walk of `samtools depth`
// (/root/reference/depth/depth.go:45) and the per-line window / class reductions of the callback
// (depth/depth.go:293-323).
//
//   * phase A is branch-free up to the LDS atomics: the record filter is mask arithmetic, the first
//     CIGAR op of all four slots is fetched with one LDS round trip, the two marks of a read share one
//     EXEC region, and the multi-op reads of a whole batch are queued with one test;
//   * phase B scans the four rows of a wave in one basic block (four independent DPP chains
//     interleave, no hazard no-ops, one LDS wait), decides "no class boundary in this quarter tile"
//     with ONE ballot, and gets its window / step indices from host-computed multiplicative inverses.
#pragma once

namespace gd {

// floor(x / d) for x < 2^31 with the (m, s) pair of magic_u31() (host side):
// m = ceil(2^s / d), s = 31 + ceil(log2 d)  (Granlund & Montgomery, N = 31).
__device__ __forceinline__ uint32_t div_magic(uint32_t x, uint32_t m, uint32_t s)
{
    return (uint32_t)(((unsigned long long)x * (unsigned long long)m) >> s);
}

// ---- sums-only output (GD_OUT_SUMS_ONLY: what depth.bed means and the depthwed matrix need) ---
// The sum of the depth over a window equals the sum over reads of their overlap with the window,
// so no per-base vector has to exist at all: every counted interval adds its overlap lengths to
// the (one or two) windows it touches, in 64-bit LDS accumulators of the tile.
struct SumSink {
    unsigned long long* acc;     // [T / 32 + 2] window accumulators of the tile (LDS)
    uint32_t r0;                 // position of t0 inside its window
    uint32_t W, w_magic, w_shift;
    int tlen;
};

// counted interval [s, e) in tile-relative positions (may stick out on both sides)
__device__ __forceinline__ void add_interval(const SumSink& S, int s, int e)
{
    const int cs = s > 0 ? s : 0, ce = e < S.tlen ? e : S.tlen;
    if (ce <= cs) return;
    const uint32_t a = (uint32_t)cs + S.r0, b = (uint32_t)(ce - 1) + S.r0;      // window-space offsets, < 2^31
    const uint32_t ks = div_magic(a, S.w_magic, S.w_shift), ke = div_magic(b, S.w_magic, S.w_shift);
    if (ks == ke) {
        atomicAdd(&S.acc[ks], (unsigned long long)(uint32_t)(ce - cs));
    } else {
        for (uint32_t k = ks; k <= ke; ++k) {
            const long long lo_w = (long long)k * S.W - S.r0, hi_w = lo_w + S.W;   // the window in tile positions
            const int lo = lo_w > cs ? (int)lo_w : cs, hi = hi_w < ce ? (int)hi_w : ce;
            atomicAdd(&S.acc[k], (unsigned long long)(uint32_t)(hi - lo));
        }
    }
}

// walk_cigar4 for the sums-only sink (ps4 is the read's start, tile relative, times 4)
template <typename OpPtr>
__device__ __forceinline__ uint32_t walk_cigar_sums(OpPtr ops, uint32_t n, int ps4, const SumSink& S)
{
    uint32_t span = 0;
    const int ps = ps4 >> 2;
    for (uint32_t k = 0; k < n; ++k) {
        const uint32_t cg = ops[k];
        const uint32_t op = cg & 0xf, len = cg >> 4;
        const bool counted = (0x181u >> op) & 1u;     // M = X
        const bool consumes = (0x18du >> op) & 1u;    // M D N = X
        if (counted && len != 0 && span < SPAN_SAT) {
            const long long s = (long long)ps + span;
            if (s < S.tlen) {
                const long long e = s + len;
                add_interval(S, s < -1 ? -1 : (int)s, e > S.tlen ? S.tlen : (int)e);
            }
        }
        if (consumes) { span += len; span = span < SPAN_SAT ? span : SPAN_SAT; }
    }
    return span;
}

// Phase A for one wave, U = 4 reads per lane and batch of NT*4.  SUMS: intervals go to a SumSink
// (window accumulators) instead of +1/-1 marks.
template <int NT, bool STAGED, bool SUMS = false>
__device__ __forceinline__ uint32_t phase_a(const PhaseA& A, int32_t (&p)[4], uint32_t (&f)[4],
                                            uint32_t (&mq)[4], uint32_t (&o0)[4], uint32_t (&o1)[4],
                                            const SumSink* sink = nullptr)
{
    constexpr int U = 4;
    const int tid = A.tid, lane = A.lane;
    const int tid4 = tid * 4, tid2 = tid * 2;
    uint32_t smax = 0;
    uint32_t qn = 0;                              // entries queued (wave uniform)
    uint32_t* const wq = A.wq;
    const uint32_t wave0 = (uint32_t)__builtin_amdgcn_readfirstlane(tid - lane);   // first thread of this wave (scalar)

    auto drain = [&](uint32_t cnt) {
        __builtin_amdgcn_wave_barrier();
        if ((uint32_t)lane < cnt) {
            const int qp = (int)wq[lane];
            const uint32_t qo = wq[WAVE + lane], qk = wq[2 * WAVE + lane];
            uint32_t span;
            if constexpr (SUMS)
                span = STAGED ? walk_cigar_sums(A.s_cig + (qo - A.clo), qk, qp, *sink) : walk_cigar_sums(A.gcig + qo, qk, qp, *sink);
            else
                span = STAGED ? walk_cigar4(A.s_cig + (qo - A.clo), qk, qp, A.T4, A.s_diff)
                              : walk_cigar4(A.gcig + qo, qk, qp, A.T4, A.s_diff);
            smax = span > smax ? span : smax;
        }
        __builtin_amdgcn_wave_barrier();
    };

    for (uint32_t base = 0; base < A.nrd; base += NT * U) {
        if (base != 0) {                          // further batches (deep tiles)
            const uint32_t rem = A.nrd - base;
            const rsrc_t r_pos = make_rsrc(A.pos + base, rem * 4u);
            const rsrc_t r_flag = make_rsrc(A.flag + base, rem * 2u);
            const rsrc_t r_mapq = make_rsrc(A.mapq + base, rem);
            const rsrc_t r_off0 = make_rsrc(A.off + base, rem * 4u);
            const rsrc_t r_off1 = make_rsrc(A.off + base + 1, rem * 4u);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                p[u]  = __builtin_amdgcn_raw_buffer_load_b32(r_pos, tid4 + u * NT * 4, 0, 0);
                f[u]  = __builtin_amdgcn_raw_buffer_load_b16(r_flag, tid2 + u * NT * 2, 0, 0);
                mq[u] = __builtin_amdgcn_raw_buffer_load_b8(r_mapq, tid + u * NT, 0, 0);
                o0[u] = __builtin_amdgcn_raw_buffer_load_b32(r_off0, tid4 + u * NT * 4, 0, 0);
                o1[u] = __builtin_amdgcn_raw_buffer_load_b32(r_off1, tid4 + u * NT * 4, 0, 0);
            }
        }
        // ---- filter, then the first op of all four slots in one LDS round trip
        uint32_t n[U], r[U];
        int ps4[U];
        bool keep[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            n[u] = o1[u] - o0[u];                                 // 0 for lanes past the range
            keep[u] = ((f[u] & A.flag_mask) == 0) & ((int)mq[u] >= A.Q) & (n[u] != 0);
        }
        uint32_t cg[U];
        if (STAGED) {
#pragma unroll
            for (int u = 0; u < U; ++u) cg[u] = A.s_cig[keep[u] ? o0[u] - A.clo : 0u];   // slot 0 is always addressable
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) { cg[u] = 0; if (keep[u]) cg[u] = A.gcig[o0[u]]; }
        }
        // ---- single-M reads: two marks in one EXEC region ------------------
        bool cx[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            r[u] = __builtin_rotateright32(cg[u], 4);             // op<<28 | len
            const bool simple = keep[u] & (n[u] == 1u) & ((r[u] - 1u) < 0x0fffffffu);   // one M op, len >= 1
            cx[u] = keep[u] & !simple;
            ps4[u] = (int)(((uint32_t)p[u] << 2) + (uint32_t)A.neg4t0);
            if (base + (uint32_t)(u * NT) + wave0 >= A.nrd) continue;   // wave uniform: slot past the range
            const uint32_t rs = simple ? r[u] : 0u;
            smax = rs > smax ? rs : smax;
            const int e4 = ps4[u] + (int)(r[u] << 2);
            if constexpr (SUMS) {
                if (simple & (e4 > 0)) add_interval(*sink, ps4[u] >> 2, e4 >> 2);   // overlaps [0, tlen)
            } else if (simple & (e4 >= 0)) {                      // reaches t0-1 or beyond
                const int cs4 = ps4[u] > -4 ? ps4[u] : -4;
                atomicAdd(lds_at(A.s_diff, cs4), 1);
                if (e4 < A.T4) atomicAdd(lds_at(A.s_diff, e4), -1);
            }
        }
        // ---- every other kept read: queue for the dense-lane walk ----------
        unsigned long long m[U];
        uint32_t cnt[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { m[u] = __builtin_amdgcn_ballot_w64(cx[u]); cnt[u] = (uint32_t)__popcll(m[u]); }
        const uint32_t tot = cnt[0] + cnt[1] + cnt[2] + cnt[3];
        if (tot != 0u) {                          // wave uniform
            if (qn + tot > (uint32_t)WAVE) { drain(qn); qn = 0; }
            if (tot <= (uint32_t)WAVE) {
                uint32_t b = qn;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (cx[u]) {
                        const uint32_t rk = b + __builtin_amdgcn_mbcnt_hi((uint32_t)(m[u] >> 32),
                                                    __builtin_amdgcn_mbcnt_lo((uint32_t)m[u], 0u));
                        wq[rk] = (uint32_t)ps4[u]; wq[WAVE + rk] = o0[u]; wq[2 * WAVE + rk] = n[u];
                    }
                    b += cnt[u];
                }
                qn += tot;
            } else {
                // more multi-op reads in one batch than the queue holds (not short-read
                // shaped data): slot by slot through the queue.  The slot's fields are
                // fetched again (cache hits) so that no register array is indexed at run time.
                const uint32_t rem = A.nrd - base;
                const rsrc_t q_pos = make_rsrc(A.pos + base, rem * 4u);
                const rsrc_t q_flag = make_rsrc(A.flag + base, rem * 2u);
                const rsrc_t q_mapq = make_rsrc(A.mapq + base, rem);
                const rsrc_t q_off0 = make_rsrc(A.off + base, rem * 4u);
                const rsrc_t q_off1 = make_rsrc(A.off + base + 1, rem * 4u);
#pragma unroll 1
                for (int u = 0; u < U; ++u) {
                    const int vo = tid + u * NT;
                    const int32_t pp = __builtin_amdgcn_raw_buffer_load_b32(q_pos, vo * 4, 0, 0);
                    const uint32_t ff = __builtin_amdgcn_raw_buffer_load_b16(q_flag, vo * 2, 0, 0);
                    const uint32_t mm = __builtin_amdgcn_raw_buffer_load_b8(q_mapq, vo, 0, 0);
                    const uint32_t a0 = __builtin_amdgcn_raw_buffer_load_b32(q_off0, vo * 4, 0, 0);
                    const uint32_t a1 = __builtin_amdgcn_raw_buffer_load_b32(q_off1, vo * 4, 0, 0);
                    const uint32_t nn = a1 - a0;
                    const bool kp = ((ff & A.flag_mask) == 0) & ((int)mm >= A.Q) & (nn != 0);
                    uint32_t c0 = 0;
                    if (kp) c0 = STAGED ? A.s_cig[a0 - A.clo] : A.gcig[a0];
                    const uint32_t rr = __builtin_rotateright32(c0, 4);
                    const bool c = kp & !((nn == 1u) & ((rr - 1u) < 0x0fffffffu));
                    const unsigned long long mu = __builtin_amdgcn_ballot_w64(c);
                    if (c) {
                        const uint32_t rk = __builtin_amdgcn_mbcnt_hi((uint32_t)(mu >> 32),
                                                __builtin_amdgcn_mbcnt_lo((uint32_t)mu, 0u));
                        wq[rk] = ((uint32_t)pp << 2) + (uint32_t)A.neg4t0; wq[WAVE + rk] = a0;
                        wq[2 * WAVE + rk] = nn;
                    }
                    drain((uint32_t)__popcll(mu));
                }
            }
        }
    }
    if (qn != 0) drain(qn);
    return smax;
}

// Phase B pass 2 for one wave on the fast configuration: every position of the
// tile inside the contig, depths below 2^22 (32-bit window accumulation is
// exact).  Other tiles take gd_tile_v6.hpp's generic phase_b_rows.
//   ST    per-base stores: 0 plain, 1 non-temporal, 2 none (windows-only output)
// The rows of a wave's quarter after their scans: what phase_b_scan leaves in registers for phase_b_finish.
struct PhaseBRows {
    int4 v[4];
    int x1[4], x2[4], x3[4], incl[4];
};

// First half: load the wave's four rows of 256 positions and run the four wave scans.  Returns the quarter's
// total (wave uniform): a kernel without a separate totals pass publishes it, waits for the other waves and
// passes the carry to phase_b_finish in B.carry.
template <int ROWS>
__device__ __forceinline__ int phase_b_scan(const PhaseB& B, PhaseBRows& R)
{
    static_assert(ROWS == 4, "scan4 interleaves exactly four rows");
#pragma unroll
    for (int r = 0; r < ROWS; ++r)
        R.v[r] = *reinterpret_cast<const int4*>(&B.s_diff[B.chunk0 + r * 256 + B.lane * 4]);
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        R.x1[r] = R.v[r].x + R.v[r].y; R.x2[r] = R.x1[r] + R.v[r].z; R.x3[r] = R.x2[r] + R.v[r].w;
        R.incl[r] = R.x3[r];
    }
    scan4(R.incl[0], R.incl[1], R.incl[2], R.incl[3]);
    int tot = 0;
#pragma unroll
    for (int r = 0; r < ROWS; ++r) tot += __builtin_amdgcn_readlane(R.incl[r], 63);
    return tot;
}

template <int ROWS, int ST>
__device__ __forceinline__ void phase_b_finish(const PhaseB& B, const PhaseBRows& R, uint32_t w_magic, uint32_t w_shift,
                                               uint32_t s_magic, uint32_t s_shift);

template <int ROWS, int ST>
__device__ __forceinline__ void phase_b_rows_full(const PhaseB& B, uint32_t w_magic, uint32_t w_shift,
                                             uint32_t s_magic, uint32_t s_shift)
{
    PhaseBRows R;
    (void)phase_b_scan<ROWS>(B, R);
    phase_b_finish<ROWS, ST>(B, R, w_magic, w_shift, s_magic, s_shift);
}

template <int ROWS, int ST>
__device__ __forceinline__ void phase_b_finish(const PhaseB& B, const PhaseBRows& R, uint32_t w_magic, uint32_t w_shift,
                                               uint32_t s_magic, uint32_t s_shift)
{
    constexpr int BIG = 0x3fffffff;
    constexpr int FAR = BIG - 65536;                     // anything at or past this is "never"
    const int lane = B.lane, t0 = B.t0, chunk0 = B.chunk0;
    const int W = B.W;
    const int4 (&v)[4] = R.v;
    const int (&x1)[4] = R.x1;
    const int (&x2)[4] = R.x2;
    const int (&x3)[4] = R.x3;
    const int (&incl)[4] = R.incl;

    // ---- stage 1: all rows of this wave: carry in, store ----------------------
    int cin[ROWS + 1];                                    // depth just before each row
    cin[0] = B.carry;
#pragma unroll
    for (int r = 0; r < ROWS; ++r) cin[r + 1] = cin[r] + __builtin_amdgcn_readlane(incl[r], 63);
    int d[ROWS][4];
    uint32_t s4[ROWS];
    int t[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const int base = cin[r] + (incl[r] - x3[r]);
        d[r][0] = base + v[r].x; d[r][1] = base + x1[r]; d[r][2] = base + x2[r]; d[r][3] = base + x3[r];
        int4* dst = reinterpret_cast<int4*>(&B.out[chunk0 + r * 256 + lane * 4]);
        if (ST == 1) {
            typedef int v4i32 __attribute__((ext_vector_type(4)));
            v4i32 dv; dv.x = d[r][0]; dv.y = d[r][1]; dv.z = d[r][2]; dv.w = d[r][3];
            __builtin_nontemporal_store(dv, reinterpret_cast<v4i32*>(dst));
        } else if (ST == 0) {
            *dst = make_int4(d[r][0], d[r][1], d[r][2], d[r][3]);
        }
        s4[r] = (uint32_t)d[r][0] + (uint32_t)d[r][1] + (uint32_t)d[r][2] + (uint32_t)d[r][3];
        int m = d[r][0] < d[r][1] ? d[r][0] : d[r][1];
        m = d[r][2] < m ? d[r][2] : m;
        t[r] = d[r][3] < m ? d[r][3] : m;                 // min of the lane's 4 positions
    }

    // ---- window / forced-break positions (scalar) ----------------------------
    const uint32_t cpos0 = (uint32_t)t0 + (uint32_t)chunk0;          // < 2^31
    uint32_t cur_win = div_magic(cpos0, w_magic, w_shift);
    const uint32_t wrem = cpos0 - cur_win * (uint32_t)W;              // 0..W-1
    const uint32_t wleft = (uint32_t)W - wrem;                        // 1..W: to the next boundary
    int nb = wleft > (uint32_t)FAR ? BIG : chunk0 + (int)wleft;       // next window boundary (rel)
    const uint32_t stepc = B.step > 0x7fffffffLL ? 0x7fffffffu : (uint32_t)B.step;
    const uint32_t srem = cpos0 - div_magic(cpos0, s_magic, s_shift) * stepc;
    const uint32_t sleft = srem == 0u ? 0u : stepc - srem;
    int nf = sleft > (uint32_t)FAR ? BIG : chunk0 + (int)sleft;       // next forced run break (rel)
    const int wstep = W > FAR ? BIG : W;
    const int fstep = stepc > (uint32_t)FAR ? BIG : (int)stepc;
    uint32_t acc = 0;
    int mn = 0x7fffffff;
    const int lo_thr = B.mincov > 1 ? B.mincov : 1;               // depths in [lo_thr, hi_thr)
    const int hi_thr = B.maxmean > 0 ? B.maxmean : 0x7fffffff;    // are CALLABLE
    const bool has_max = B.maxmean > 0;

    // ---- one test for the whole quarter tile: any position (or the one before
    // it) outside CALLABLE, or a forced break inside?  (depth/depth.go:307-323)
    bool any_noisy;
    {
        int tm = t[0];
#pragma unroll
        for (int r = 1; r < ROWS; ++r) tm = t[r] < tm ? t[r] : tm;
        any_noisy = __builtin_amdgcn_ballot_w64(tm < lo_thr) != 0ull;
        if (has_max) {
            int tx = d[0][0];
#pragma unroll
            for (int r = 0; r < ROWS; ++r) {
#pragma unroll
                for (int j = 0; j < 4; ++j) tx = d[r][j] > tx ? d[r][j] : tx;
            }
            any_noisy = any_noisy || __builtin_amdgcn_ballot_w64(tx >= hi_thr) != 0ull;
        }
        any_noisy = any_noisy || B.carry < lo_thr || B.carry >= hi_thr || nf < chunk0 + ROWS * 256;
    }

#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const int rb = chunk0 + r * 256;                  // row start (rel)
        const int ib = rb + lane * 4;                     // this lane's first position (rel)
        const int d0 = d[r][0], d1 = d[r][1], d2 = d[r][2], d3 = d[r][3];

        // ---- window sum / min (depth/depth.go:181-189, :293-305) ---------
        if (nb >= rb + 256) {
            acc += s4[r];
            mn = t[r] < mn ? t[r] : mn;
        } else if (nb + wstep >= rb + 256) {
            // exactly one boundary in this row: split at lane granularity, fix
            // the straddling lane with scalar arithmetic
            const int rel = nb - rb;                      // 0..255
            const int L = rel >> 2, k = rel & 3;
            const bool lt = lane < L;
            const uint32_t a_old = acc + (lt ? s4[r] : 0u);
            const int t_old = lt ? t[r] : 0x7fffffff;
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
            acc = gt ? s4[r] : 0u;
            mn = gt ? t[r] : 0x7fffffff;
            if (lane == L) { acc = qs; mn = qm; }
            cur_win++;
            nb = nb + wstep > BIG ? BIG : nb + wstep;
        } else {
            // several boundaries in one row (W < 256)
            int seg = rb;
            const int dd[4] = {d0, d1, d2, d3};
            while (nb < rb + 256) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int pj = ib + j;
                    if (pj >= seg && pj < nb) { acc += (uint32_t)dd[j]; mn = dd[j] < mn ? dd[j] : mn; }
                }
                const uint32_t tot = (uint32_t)wave_total((int)acc);
                const int m = wave_min_dpp(mn);
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
                if (pj >= seg) { acc += (uint32_t)dd[j]; mn = dd[j] < mn ? dd[j] : mn; }
            }
        }

        // ---- coverage class boundaries (depth/depth.go:307-323) ----------
        if (any_noisy) {
            const int carry_before = cin[r];
            bool noisy = __ballot(t[r] < lo_thr) != 0ull;
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
                while (nf < rb + 256) {                   // forced breaks (quirk Q1), incl. position 0
                    const int o = nf - ib;
                    if (o >= 0 && o < 4) bm |= 1u << o;
                    nf = nf + fstep > BIG ? BIG : nf + fstep;
                }
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
    }
    // flush the open window segment of this wave
    {
        const uint32_t tot = (uint32_t)wave_total((int)acc);
        const int m = wave_min_dpp(mn);
        if (lane == 0) {
            atomicAdd(reinterpret_cast<unsigned long long*>(&B.wsum[cur_win]), (unsigned long long)tot);
            atomicMin(&B.wmin[cur_win], m);
        }
    }
}

// One tile.  OPT: per-base stores 0 plain, 1 non-temporal, 2 none (gd_set_outputs without GD_OUT_PERBASE).
// Called by every thread of the workgroup.
template <int T, int NT, int OPT>
__device__ __forceinline__ void tile_body(const Job& job, const TileInfo& ti, const int tile)
{
    constexpr int NW = NT / WAVE;          // waves per workgroup
    constexpr int CHUNK = T / NW;          // positions per wave
    constexpr int ROWS = CHUNK / 256;      // rows of 256 positions per wave
    constexpr int NWORDS = T / 32;         // bitmap words
    constexpr int CQ = (T * 3) / 8;        // staged CIGAR ops (30x/150 bp needs ~T/4)
    constexpr int U = 4;                   // reads per lane in flight
    constexpr int CCH = (CQ + NT - 1) / NT;   // staged ops per thread
    static_assert(CHUNK % 256 == 0, "wave chunk must be whole rows");

    __shared__ __attribute__((aligned(16))) int32_t s_diffp[T + 4];  // [3] = index -1
    __shared__ uint32_t s_bmap[NWORDS];    // boundary bit per position
    __shared__ uint32_t s_clo[NWORDS];     // class bit 0 at boundary positions
    __shared__ uint32_t s_chi[NWORDS];     // class bit 1 at boundary positions
    __shared__ uint32_t s_cig[CQ];         // staged CIGAR ops of the tile's reads
    __shared__ uint32_t s_wq[NW * 3 * WAVE];  // per-wave queues of multi-op reads
    __shared__ int32_t  s_wtot[NW];
    __shared__ uint32_t s_wcnt[NW];
    __shared__ uint32_t s_hasb;
    __shared__ uint32_t s_base;
    int32_t* const s_diff = s_diffp + 4;

    const int tid = threadIdx.x;
    const int lane = tid & (WAVE - 1);
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int seen0 = __hip_atomic_load(&job.counters->max_span, __ATOMIC_RELAXED,
                                        __HIP_MEMORY_SCOPE_AGENT);
    const int32_t t0 = ti.t0;
    const int32_t tend = t0 + T < ti.length ? t0 + T : ti.length;   // clipped tile end
    const int tlen = tend - t0;                                      // valid positions, 1..T
    const int T4 = tlen * 4;

    // ---- loads first: record fields of the first batch + the tile's CIGAR range
    const uint32_t nrd = ti.hi - ti.lo;
    const uint32_t nst = ti.chi - ti.clo;
    const bool staged = nst <= (uint32_t)CQ;
    const rsrc_t r_pos = make_rsrc(ti.pos + ti.lo, nrd * 4u);
    const rsrc_t r_flag = make_rsrc(ti.flag + ti.lo, nrd * 2u);
    const rsrc_t r_mapq = make_rsrc(ti.mapq + ti.lo, nrd);
    const rsrc_t r_off0 = make_rsrc(ti.off + ti.lo, nrd * 4u);
    const rsrc_t r_off1 = make_rsrc(ti.off + ti.lo + 1, nrd * 4u);
    const rsrc_t r_cig = make_rsrc(ti.cigar + ti.clo, staged ? nst * 4u : 0u);
    const int tid4 = tid * 4, tid2 = tid * 2;
    int32_t  p[U];
    uint32_t f[U], mq[U], o0[U], o1[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        p[u]  = __builtin_amdgcn_raw_buffer_load_b32(r_pos, tid4 + u * NT * 4, 0, 0);
        f[u]  = __builtin_amdgcn_raw_buffer_load_b16(r_flag, tid2 + u * NT * 2, 0, 0);
        mq[u] = __builtin_amdgcn_raw_buffer_load_b8(r_mapq, tid + u * NT, 0, 0);
        o0[u] = __builtin_amdgcn_raw_buffer_load_b32(r_off0, tid4 + u * NT * 4, 0, 0);
        o1[u] = __builtin_amdgcn_raw_buffer_load_b32(r_off1, tid4 + u * NT * 4, 0, 0);
    }
    // out-of-range op loads return 0 (descriptor bound), so no per-chunk test is needed
    uint32_t cgv[CCH];
#pragma unroll
    for (int k = 0; k < CCH; ++k)
        cgv[k] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(r_cig, tid4 + k * NT * 4, 0, 0);

    // ---- zero LDS (overlaps the loads above) -----------------------------
    {
        const int4 z = make_int4(0, 0, 0, 0);
        int4* d4 = reinterpret_cast<int4*>(s_diffp);
#pragma unroll
        for (int i = 0; i < T / 4 / NT; ++i) d4[tid + i * NT] = z;       // (T / 4 is a multiple of NT)
        if (tid == 0) d4[T / 4] = z;
        for (int i = tid; i < NWORDS; i += NT) { s_bmap[i] = 0; s_clo[i] = 0; s_chi[i] = 0; }
        if (tid == 0) s_hasb = 0;
#pragma unroll
        for (int k = 0; k < CCH; ++k)
            if (k * NT + tid < CQ) s_cig[k * NT + tid] = cgv[k];
    }
    __syncthreads();

    // ---- phase A: reads -> clipped intervals -> LDS +1/-1 -----------------
    if (nrd != 0) {
        PhaseA A;
        A.pos = ti.pos + ti.lo; A.flag = ti.flag + ti.lo; A.mapq = ti.mapq + ti.lo; A.off = ti.off + ti.lo;
        A.s_diff = s_diff; A.s_cig = s_cig; A.wq = &s_wq[wv * (3 * WAVE)];
        A.gcig = ti.cigar; A.clo = ti.clo; A.nrd = nrd;
        A.neg4t0 = (int)(0u - ((uint32_t)t0 << 2));       // (p<<2) + neg4t0 = 4*(p - t0)
        A.T4 = T4; A.flag_mask = job.flag_mask; A.Q = job.Q; A.tid = tid; A.lane = lane;
        const uint32_t smax = staged ? phase_a<NT, true>(A, p, f, mq, o0, o1)
                                     : phase_a<NT, false>(A, p, f, mq, o0, o1);
        // publish the largest span seen
        publish_span(&job.counters->max_span, smax, seen0, lane);
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
    {
        PhaseB B;
        B.s_diff = s_diff; B.s_bmap = s_bmap; B.s_clo = s_clo; B.s_chi = s_chi; B.s_hasb = &s_hasb;
        B.out = job.perbase + ti.base_off + t0;
        B.wsum = job.win_sum + ti.win_off;
        B.wmin = job.win_min + ti.win_off;
        B.t0 = t0; B.tlen = tlen; B.chunk0 = chunk0; B.lane = lane;
        B.W = job.W; B.mincov = job.mincov; B.maxmean = job.maxmean; B.step = job.step;
        int carry = s_diff[-1];                            // depth at t0-1
#pragma unroll
        for (int v = 0; v < NW - 1; ++v) carry += v < wv ? s_wtot[v] : 0;
        B.carry = carry;
        // depth <= reads examined for the tile: below 2^22 the 32-bit window
        // accumulation is exact (1024 positions x depth < 2^32)
        const bool wide = nrd >= (1u << 22);
        if constexpr (ROWS == 4) {
            if (tlen == T && !wide) phase_b_rows_full<ROWS, OPT>(B, job.w_magic, job.w_shift, job.s_magic, job.s_shift);
            else                    phase_b_rows<ROWS, false, true, OPT>(B);   // clipped or very deep tiles
        } else {
            if (tlen == T && !wide) phase_b_rows<ROWS, true, false, OPT>(B);   // other tile shapes: row by row
            else                    phase_b_rows<ROWS, false, true, OPT>(B);
        }
    }
    __syncthreads();

    // ---- phase C: compact class boundaries of this tile -------------------
    phase_c<T, NT>(job, tile, t0, ti.ctg, tid, lane, wv, s_bmap, s_clo, s_chi, s_wcnt, &s_hasb, &s_base);
}

// Every tile, one workgroup each (GD_OPT_FAST_KERNEL = 0, or contig arrays the straight-line kernel's vector loads cannot take).
template <int T, int NT, int OPT>
__global__ __launch_bounds__(NT) void gd_tile_kernel(Job job)
{
    // XCD-aware order: workgroup b runs on XCD b % 8; every XCD gets a contiguous
    // eighth of the genome so the look-back reads of neighbouring tiles hit the same L2.
    const int per = (job.n_tiles + 7) >> 3;
    const int tile = (int)(blockIdx.x & 7u) * per + (int)(blockIdx.x >> 3);
    if (tile >= job.n_tiles) return;
    const TileInfo ti = job.tiles[tile];
    tile_body<T, NT, OPT>(job, ti, tile);
}

// The `slow` tiles of a fast run: gd_prep_kernel compacted their descriptors to the front of job.tiles
// (tile id in TileInfo::tile) and counted them in Counters::n_slow[job.parity]; a fixed grid strides over the list.
template <int T, int NT, int OPT>
__global__ __launch_bounds__(NT) void gd_tile_slow_kernel(Job job)
{
    const uint32_t n = __hip_atomic_load(&job.counters->n_slow[job.parity], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (uint32_t i = blockIdx.x; i < n; i += gridDim.x) {
        const TileInfo ti = job.tiles[i];
        tile_body<T, NT, OPT>(job, ti, ti.tile);
        __syncthreads();                       // the next tile reuses the LDS arrays
    }
}


// K1s: the tile kernel for GD_OUT_SUMS_ONLY.  Same read selection, filter, staging and look-back
// verification as gd_tile_kernel; phase A feeds window accumulators, phases B and C do not exist.
template <int T, int NT>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(8, 8))) void gd_tile_sums_kernel(Job job)
{
    constexpr int NW = NT / WAVE;
    constexpr int CQ = (T * 3) / 8;
    constexpr int U = 4;
    constexpr int CCH = (CQ + NT - 1) / NT;
    constexpr int NACC = T / 32 + 2;       // windows a tile can touch when W >= 32

    __shared__ unsigned long long s_acc[NACC];
    __shared__ uint32_t s_cig[CQ];
    __shared__ uint32_t s_wq[NW * 3 * WAVE];

    const int per = (job.n_tiles + 7) >> 3;
    const int tile = (int)(blockIdx.x & 7u) * per + (int)(blockIdx.x >> 3);
    if (tile >= job.n_tiles) return;

    const int tid = threadIdx.x;
    const int lane = tid & (WAVE - 1);
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const TileInfo ti = job.tiles[tile];
    const int seen0 = __hip_atomic_load(&job.counters->max_span, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int32_t t0 = ti.t0;
    const int32_t tend = t0 + T < ti.length ? t0 + T : ti.length;
    const int tlen = tend - t0;

    const uint32_t nrd = ti.hi - ti.lo;
    const uint32_t nst = ti.chi - ti.clo;
    const bool staged = nst <= (uint32_t)CQ;
    const rsrc_t r_pos = make_rsrc(ti.pos + ti.lo, nrd * 4u);
    const rsrc_t r_flag = make_rsrc(ti.flag + ti.lo, nrd * 2u);
    const rsrc_t r_mapq = make_rsrc(ti.mapq + ti.lo, nrd);
    const rsrc_t r_off0 = make_rsrc(ti.off + ti.lo, nrd * 4u);
    const rsrc_t r_off1 = make_rsrc(ti.off + ti.lo + 1, nrd * 4u);
    const rsrc_t r_cig = make_rsrc(ti.cigar + ti.clo, staged ? nst * 4u : 0u);
    const int tid4 = tid * 4, tid2 = tid * 2;
    int32_t  p[U];
    uint32_t f[U], mq[U], o0[U], o1[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        p[u]  = __builtin_amdgcn_raw_buffer_load_b32(r_pos, tid4 + u * NT * 4, 0, 0);
        f[u]  = __builtin_amdgcn_raw_buffer_load_b16(r_flag, tid2 + u * NT * 2, 0, 0);
        mq[u] = __builtin_amdgcn_raw_buffer_load_b8(r_mapq, tid + u * NT, 0, 0);
        o0[u] = __builtin_amdgcn_raw_buffer_load_b32(r_off0, tid4 + u * NT * 4, 0, 0);
        o1[u] = __builtin_amdgcn_raw_buffer_load_b32(r_off1, tid4 + u * NT * 4, 0, 0);
    }
    uint32_t cgv[CCH];
#pragma unroll
    for (int k = 0; k < CCH; ++k)
        cgv[k] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(r_cig, tid4 + k * NT * 4, 0, 0);
    for (int i = tid; i < NACC; i += NT) s_acc[i] = 0ull;
#pragma unroll
    for (int k = 0; k < CCH; ++k)
        if (k * NT + tid < CQ) s_cig[k * NT + tid] = cgv[k];
    __syncthreads();

    const uint32_t w_first = div_magic((uint32_t)t0, job.w_magic, job.w_shift);
    SumSink S;
    S.acc = s_acc; S.r0 = (uint32_t)t0 - w_first * (uint32_t)job.W; S.W = (uint32_t)job.W;
    S.w_magic = job.w_magic; S.w_shift = job.w_shift; S.tlen = tlen;
    if (nrd != 0) {
        PhaseA A;
        A.pos = ti.pos + ti.lo; A.flag = ti.flag + ti.lo; A.mapq = ti.mapq + ti.lo; A.off = ti.off + ti.lo;
        A.s_diff = nullptr; A.s_cig = s_cig; A.wq = &s_wq[wv * (3 * WAVE)];
        A.gcig = ti.cigar; A.clo = ti.clo; A.nrd = nrd;
        A.neg4t0 = (int)(0u - ((uint32_t)t0 << 2));
        A.T4 = tlen * 4; A.flag_mask = job.flag_mask; A.Q = job.Q; A.tid = tid; A.lane = lane;
        const uint32_t smax = staged ? phase_a<NT, true, true>(A, p, f, mq, o0, o1, &S)
                                     : phase_a<NT, false, true>(A, p, f, mq, o0, o1, &S);
        publish_span(&job.counters->max_span, smax, seen0, lane);
    }
    __syncthreads();
    // the tile's share of every window it touches
    const uint32_t n_touch = div_magic((uint32_t)(tlen - 1) + S.r0, job.w_magic, job.w_shift) + 1u;
    unsigned long long* const wsum = reinterpret_cast<unsigned long long*>(job.win_sum + ti.win_off) + w_first;
    for (uint32_t k = (uint32_t)tid; k < n_touch && k < (uint32_t)NACC; k += NT) {
        const unsigned long long v = s_acc[k];
        if (v) atomicAdd(&wsum[k], v);
    }
}

}  // namespace gd
