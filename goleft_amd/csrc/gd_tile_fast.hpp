// gd_tile_fast.hpp -- K1, the tile kernel of the short-read path: the straight-line kernel for ORDINARY
// tiles (all T = 4096 positions inside the contig, at most one batch of 1024 candidate reads, at most
// 1280 CIGAR ops -- 99.99 % of a 30x genome; gd_prep_kernel lists every other tile for
// gd_tile_slow_kernel).  Same algorithm and results as gd_tile_generic.hpp; it replaces, like it, the
// per-read CIGAR walk of `samtools depth` (/root/reference/depth/depth.go:45) and the per-line window /
// class reductions of the callback (depth/depth.go:293-323).
//
// The generic kernel was bound by instruction issue, not bytes (~456 VALU + ~400
// SALU wave-instructions per wave and tile, 5 workgroups per CU).  What is different here:
//   * it reads the records AS THEY ARRIVED (pos, flag, MAPQ, CSR offsets, BAM-encoded ops in any form) -- nothing
//     derived has to exist before the first gd_compute: a `goleft depth` run computes every input exactly once, so
//     a pass that rewrites the records first costs more than it saves (rounds 2-4 kept such a pass, "canonical
//     records", as an option; round 5 removed it).  A lane's four reads bring their CSR offsets (no prefix sum
//     over op counts), flag and MAPQ come as one 8-byte and one 4-byte load per lane, reads of ONE OR TWO ops of
//     any kind (150M, 20S130M, 100M50S, 5H145M ... 97 % of short reads) are one interval computed inline: no op
//     decode loop, no zero-length / non-M special cases;
//   * everything longer goes to ONE workgroup queue (an LDS counter, one atomic per wave that has any) and is
//     walked after a barrier by as many lanes as there are entries -- one dense walk per TILE instead of one
//     sparse walk per WAVE;
//   * no pass 1: a wave scans its quarter, publishes the quarter's total, and only adds the carry of
//     the quarters before it after the barrier;
//   * nothing is derived on the (CU-shared) scalar unit: pointers at the tile's first read / op and
//     every wave's window / run-break state come resolved from gd_prep_kernel (TileFast);
//   * LDS: 16 KB difference array + 5 KB op staging (reused as the class-boundary bitmaps of phases B
//     and C) + 1.4 KB queue = 23 004 bytes => 7 workgroups (28 waves) per CU instead of 5.
// gd_prep_kernel rounds the tile's first read down to a multiple of four so that every vector load is naturally
// aligned (the contig's arrays must be: 16 / 16 / 8 / 4 bytes for pos / offsets / flag / MAPQ, else the generic
// kernel runs).
#pragma once

#ifndef GD_LOAD_AUX
#define GD_LOAD_AUX 0
#endif

namespace gd {
namespace fast {

constexpr int T = 4096;
constexpr int NT = 256;
constexpr int NW = NT / WAVE;          // 4 waves
constexpr int CHUNK = T / NW;          // 1024 positions per wave
constexpr int ROWS = CHUNK / 256;      // 4 rows of 256 positions per wave
constexpr int CQ = 1280;               // staged ops (1024 would put 1-2 % of a 30x genome's tiles on the slow list)
constexpr int U = 4;                   // reads per lane
constexpr int QCAP = 120;              // queued multi-op reads per tile (more: walked in place); 23 004 bytes of LDS = seven workgroups per CU at any allocation granularity up to 512
constexpr int NWORDS = T / 32;

// ST: per-base stores 0 plain, 1 non-temporal, 2 none (windows-only output).
template <int ST>
__global__ __launch_bounds__(NT) void gd_tile_fast_kernel(Job job)
{
    constexpr int CQN = CQ;
    constexpr int QCAPN = QCAP;
    __shared__ __attribute__((aligned(16))) int32_t s_diffp[T + 4];   // [3] = index -1
    __shared__ __attribute__((aligned(16))) uint32_t s_cig[CQN];      // phase A: staged ops; B, C: boundary bitmaps
    __shared__ uint32_t s_q[3 * QCAPN];                                // ps4 | first op (staged index) | n ops
    __shared__ int32_t  s_wtot[NW];
    __shared__ uint32_t s_wcnt[NW];
    __shared__ uint32_t s_qn;
    __shared__ uint32_t s_hasb;
    __shared__ uint32_t s_base;
    int32_t* const s_diff = s_diffp + 4;
    uint32_t* const s_bmap = s_cig;
    uint32_t* const s_clo = s_cig + NWORDS;
    uint32_t* const s_chi = s_cig + 2 * NWORDS;

    // XCD-aware order: workgroup b runs on XCD b % 8; every XCD gets a contiguous
    // eighth of the genome so the look-back reads of neighbouring tiles hit the same L2.
    const int per = (job.n_tiles + 7) >> 3;
    const int tile = (int)(blockIdx.x & 7u) * per + (int)(blockIdx.x >> 3);
    if (tile >= job.n_tiles) return;
    // field by field through the (uniform) pointer: scalar loads; a local copy of the record would be
    // indexed by the wave number below and end up in scratch memory
    const TileFast* __restrict__ const tp = job.ftiles + tile;
    struct {
        const int32_t* pos; const uint32_t* rec; const uint32_t* cig; const uint16_t* flag; const uint8_t* mapq;
        int32_t* out; int64_t* wsum; int32_t* wmin; int32_t t0; uint32_t nrd, nst, clo; int32_t ctg;
    } tf;
    tf.nrd = tp->nrd;
    if ((int32_t)tf.nrd < 0) return;                   // on the slow list
    tf.pos = tp->pos; tf.rec = tp->rec; tf.cig = tp->cig;
    tf.flag = tp->flag; tf.mapq = tp->mapq; tf.clo = tp->clo;
    tf.out = tp->out; tf.wsum = tp->wsum; tf.wmin = tp->wmin;
    tf.t0 = tp->t0; tf.nst = tp->nst; tf.ctg = tp->ctg;

    const int tid = threadIdx.x;
    const int lane = tid & (WAVE - 1);
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int seen0 = __hip_atomic_load(&job.counters->max_span, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    constexpr int T4 = T * 4;

    // ---- loads first: every lane takes FOUR CONSECUTIVE reads -- 16 bytes of `pos`, 16 + 4 bytes of CSR
    // offsets, their flags and MAPQs -- and the workgroup the tile's op range: one memory round trip.  The descriptors end at the tile's last read: dwords past it
    // read 0 (gfx950 checks the range of a multi-dword raw buffer load per dword: tools/probe/oob_x4.hip),
    // i.e. no ops => dropped.
    const uint32_t nrd = tf.nrd;
    const rsrc_t r_pos = make_rsrc(tf.pos, nrd * 4u);
    // tf.rec points at the CSR offsets of the tile's first read; entry nrd (the end of the last read) is read too
    const rsrc_t r_rec = make_rsrc(tf.rec, (nrd != 0u ? nrd + 1u : nrd) * 4u);   // (no reads: no array either)
    const rsrc_t r_cig = make_rsrc(tf.cig, tf.nst * 4u);
    const int tid4 = tid * 4;
    typedef unsigned int v4u __attribute__((ext_vector_type(4)));
    typedef unsigned int v2u __attribute__((ext_vector_type(2)));
    const v4u pv = __builtin_amdgcn_raw_buffer_load_b128(r_pos, tid * 16, 0, GD_LOAD_AUX);
    const v4u rv = __builtin_amdgcn_raw_buffer_load_b128(r_rec, tid * 16, 0, GD_LOAD_AUX);
    // flag: four 16-bit values, MAPQ: four bytes per lane.  The ranges are rounded up to whole dwords (the range
    // check is per dword): at most 2 / 3 bytes past the tile's last read, inside the same aligned word.
    const rsrc_t r_flag = make_rsrc(tf.flag, ((nrd + 1u) & ~1u) * 2u);
    const rsrc_t r_mapq = make_rsrc(tf.mapq, (nrd + 3u) & ~3u);
    const uint32_t o4 = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(r_rec, tid * 16 + 16, 0, GD_LOAD_AUX);
    const v2u fv = __builtin_amdgcn_raw_buffer_load_b64(r_flag, tid * 8, 0, GD_LOAD_AUX);
    const uint32_t mv = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(r_mapq, tid4, 0, GD_LOAD_AUX);
    uint32_t cgv[CQN / NT];
#pragma unroll
    for (int k = 0; k < CQN / NT; ++k)
        cgv[k] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(r_cig, tid4 + k * NT * 4, 0, GD_LOAD_AUX);

    // ---- zero the difference array (overlaps the loads) -----------------------------------------
    {
        const int4 z = make_int4(0, 0, 0, 0);
        int4* d4 = reinterpret_cast<int4*>(s_diffp);
#pragma unroll
        for (int i = 0; i < T / 4 / NT; ++i) d4[tid + i * NT] = z;
        if (tid == 0) { d4[T / 4] = z; s_qn = 0; s_hasb = 0; }
    }
    // ---- where each read's ops are: the CSR offsets say it
    const uint32_t rec[U] = {rv.x, rv.y, rv.z, rv.w};
    uint32_t n[U], ex[U];                               // op count, staged index of the read's first op
    {
        const uint32_t oe[U] = {rv.y, rv.z, rv.w, o4};
#pragma unroll
        for (int u = 0; u < U; ++u) {
            n[u] = (uint32_t)(tid4 + u) < nrd ? oe[u] - rec[u] : 0u;
            ex[u] = rec[u] - tf.clo;
        }
    }
    // stage the ops
#pragma unroll
    for (int k = 0; k < CQN / NT; ++k) s_cig[k * NT + tid] = cgv[k];
    __syncthreads();

    // ---- phase A: reads -> clipped intervals -> LDS +1/-1 ---------------------------------------
    uint32_t smax = 0;
    {
        const int neg4t0 = (int)(0u - ((uint32_t)tf.t0 << 2));     // (p << 2) + neg4t0 = 4 * (p - t0)
        const int32_t p[U] = {(int32_t)pv.x, (int32_t)pv.y, (int32_t)pv.z, (int32_t)pv.w};
        uint32_t cg[U], idx[U];
        bool keep[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            idx[u] = ex[u];
            const uint32_t fw = u < 2 ? fv.x : fv.y;
            const uint32_t f = (u & 1) ? fw >> 16 : fw & 0xffffu;
            const uint32_t mq = (mv >> (8 * u)) & 0xffu;
            keep[u] = ((f & job.flag_mask) == 0) & ((int)mq >= job.Q) & (n[u] != 0);   // n = 0 past the tile's reads
        }
#pragma unroll
        for (int u = 0; u < U; ++u) cg[u] = s_cig[keep[u] ? idx[u] : 0u];   // slot 0 is always addressable
        uint32_t cg2[U];                                           // the second op of a two-op read (else 0 = "0M": neutral)
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t x = s_cig[(keep[u] & (n[u] == 2u)) ? idx[u] + 1u : 0u];
            cg2[u] = n[u] == 2u ? x : 0u;
        }
        int ps4[U];
        bool cx[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            ps4[u] = (int)(((uint32_t)p[u] << 2) + (uint32_t)neg4t0);
            {
                // one or two ops of any kind are at most ONE counted interval: M/=/X lengths add up (two of them
                // are adjacent), a leading D/N shifts the start, I/S/H/P and a trailing D/N cover nothing
                const uint32_t oa = cg[u] & 0xfu, la = cg[u] >> 4, ob = cg2[u] & 0xfu, lb = cg2[u] >> 4;
                const bool ca = (0x181u >> oa) & 1u, ka = (0x18du >> oa) & 1u;     // counted (M = X), consumes (M D N = X)
                const bool cb = (0x181u >> ob) & 1u, kb = (0x18du >> ob) & 1u;
                const bool simple = keep[u] & (n[u] <= 2u);
                cx[u] = keep[u] & (n[u] > 2u);
                uint32_t L = (ca ? la : 0u) + (cb ? lb : 0u);
                const uint32_t so = (ka & !ca) ? la : 0u;
                const uint32_t span = (ka ? la : 0u) + (kb ? lb : 0u);
                const uint32_t rs = simple ? (span < SPAN_SAT ? span : SPAN_SAT) : 0u;
                smax = rs > smax ? rs : smax;
                L = L < (1u << 27) ? L : (1u << 27);               // (a longer read is refused by the host: its span says so)
                const int s4 = ps4[u] + (int)(so << 2);
                const int e4 = s4 + (int)(L << 2);
                if (simple & (L != 0u) & (e4 >= 0) & (s4 < T4)) {
                    const int cs4 = s4 > -4 ? s4 : -4;
                    atomicAdd(lds_at(s_diff, cs4), 1);
                    if (e4 < T4) atomicAdd(lds_at(s_diff, e4), -1);
                }
            }
        }
        // multi-op reads -> the workgroup queue
        unsigned long long m[U];
        uint32_t cnt[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { m[u] = __builtin_amdgcn_ballot_w64(cx[u]); cnt[u] = (uint32_t)__popcll(m[u]); }
        const uint32_t tot = cnt[0] + cnt[1] + cnt[2] + cnt[3];
        if (tot != 0u) {                                           // wave uniform
            uint32_t b = 0;
            if (lane == 0) b = atomicAdd(&s_qn, tot);
            b = (uint32_t)__builtin_amdgcn_readfirstlane((int)b);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (cx[u]) {
                    const uint32_t rk = b + __builtin_amdgcn_mbcnt_hi((uint32_t)(m[u] >> 32),
                                                __builtin_amdgcn_mbcnt_lo((uint32_t)m[u], 0u));
                    if (rk < (uint32_t)QCAPN) {
                        s_q[rk] = (uint32_t)ps4[u]; s_q[QCAPN + rk] = idx[u]; s_q[2 * QCAPN + rk] = n[u];
                    } else {
                        // did not fit the queue (not short-read shaped data): walked by its own lane
                        const uint32_t sp = walk_cigar4(s_cig + idx[u], n[u], ps4[u], T4, s_diff);
                        smax = sp > smax ? sp : smax;
                    }
                }
                b += cnt[u];
            }
        }
    }
    __syncthreads();

    // ---- the queued multi-op reads: one lane each ------------------------------------------------
    {
        const uint32_t nq = s_qn < (uint32_t)QCAPN ? s_qn : (uint32_t)QCAPN;
        if ((uint32_t)(wv * WAVE) < nq) {                          // wave uniform: usually wave 0 only
            if ((uint32_t)tid < nq) {
                const uint32_t sp = walk_cigar4(s_cig + s_q[QCAPN + tid], s_q[2 * QCAPN + tid], (int)s_q[tid], T4, s_diff);
                smax = sp > smax ? sp : smax;
            }
        }
        publish_span(&job.counters->max_span, smax, seen0, lane);
    }
    __syncthreads();

    // ---- phase B, first half: scan this wave's quarter in registers, publish its total -----------
    const int chunk0 = wv * CHUNK;
    int4 v[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r)
        v[r] = *reinterpret_cast<const int4*>(&s_diff[chunk0 + r * 256 + lane * 4]);
    // the staging area becomes the boundary bitmaps (its last readers passed the barrier above)
    s_cig[tid] = 0;
    if (tid < 3 * NWORDS - NT) s_cig[NT + tid] = 0;
    int x1[ROWS], x2[ROWS], x3[ROWS], incl[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        x1[r] = v[r].x + v[r].y; x2[r] = x1[r] + v[r].z; x3[r] = x2[r] + v[r].w;
        incl[r] = x3[r];
    }
    scan4(incl[0], incl[1], incl[2], incl[3]);
    int rtot[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) rtot[r] = __builtin_amdgcn_readlane(incl[r], 63);
    if (lane == 0) s_wtot[wv] = rtot[0] + rtot[1] + rtot[2] + rtot[3];
    __syncthreads();

    // ---- phase B, second half: carry, store, window reduce, class boundaries ----------------------
    {
        constexpr int BIG = FAST_BIG;
        int carry = s_diff[-1];                                    // depth at t0-1
#pragma unroll
        for (int w = 0; w < NW - 1; ++w) carry += w < wv ? s_wtot[w] : 0;
        int cin[ROWS + 1];                                         // depth just before each row
        cin[0] = carry;
#pragma unroll
        for (int r = 0; r < ROWS; ++r) cin[r + 1] = cin[r] + rtot[r];
        int d[ROWS][4];
        uint32_t s4[ROWS];
        int t[ROWS];
        int32_t* const out = tf.out;
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const int base = cin[r] + (incl[r] - x3[r]);
            d[r][0] = base + v[r].x; d[r][1] = base + x1[r]; d[r][2] = base + x2[r]; d[r][3] = base + x3[r];
            if (ST != 2) {
                int4* dst = reinterpret_cast<int4*>(&out[chunk0 + r * 256 + lane * 4]);
                if (ST == 1) {
                    typedef int v4i32 __attribute__((ext_vector_type(4)));
                    v4i32 dv; dv.x = d[r][0]; dv.y = d[r][1]; dv.z = d[r][2]; dv.w = d[r][3];
                    __builtin_nontemporal_store(dv, reinterpret_cast<v4i32*>(dst));
                } else {
                    *dst = make_int4(d[r][0], d[r][1], d[r][2], d[r][3]);
                }
            }
            s4[r] = (uint32_t)d[r][0] + (uint32_t)d[r][1] + (uint32_t)d[r][2] + (uint32_t)d[r][3];
            int m = d[r][0] < d[r][1] ? d[r][0] : d[r][1];
            m = d[r][2] < m ? d[r][2] : m;
            t[r] = d[r][3] < m ? d[r][3] : m;                      // min of the lane's 4 positions
        }

        // window / forced-break state of this quarter, resolved by gd_prep_kernel
        const int W = job.W;
        uint32_t cur_win = tp->win0[wv];
        const int wl = tp->wleft[wv], sl = tp->sleft[wv];
        int nb = wl >= BIG ? BIG : chunk0 + wl;                    // next window boundary (tile relative)
        int nf = sl >= BIG ? BIG : chunk0 + sl;                    // next forced run break (tile relative)
        const int wstep = W > FAST_FAR ? BIG : W;
        const uint32_t stepc = job.step > 0x7fffffffLL ? 0x7fffffffu : (uint32_t)job.step;
        const int fstep = stepc > (uint32_t)FAST_FAR ? BIG : (int)stepc;
        uint32_t acc = 0;
        int mn = 0x7fffffff;
        const int lo_thr = job.mincov > 1 ? job.mincov : 1;        // depths in [lo_thr, hi_thr)
        const int hi_thr = job.maxmean > 0 ? job.maxmean : 0x7fffffff;   // are CALLABLE
        const bool has_max = job.maxmean > 0;
        int64_t* const wsum = tf.wsum;
        int32_t* const wmin = tf.wmin;

        // one test for the whole quarter tile: any position (or the one before it) outside CALLABLE,
        // or a forced break inside?  (depth/depth.go:307-323)
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
            any_noisy = any_noisy || carry < lo_thr || carry >= hi_thr || nf < chunk0 + ROWS * 256;
        }

#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const int rb = chunk0 + r * 256;                       // row start (rel)
            const int ib = rb + lane * 4;                          // this lane's first position (rel)
            const int d0 = d[r][0], d1 = d[r][1], d2 = d[r][2], d3 = d[r][3];

            // ---- window sum / min (depth/depth.go:181-189, :293-305) ---------
            if (nb >= rb + 256) {
                acc += s4[r];
                mn = t[r] < mn ? t[r] : mn;
            } else if (nb + wstep >= rb + 256) {
                // exactly one boundary in this row: split at lane granularity, fix
                // the straddling lane with scalar arithmetic
                const int rel = nb - rb;                           // 0..255
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
                const uint32_t tot = (uint32_t)wave_total((int)a_old) + ps;   // < 2^32 (depth <= 1024 reads)
                int m = wave_min_dpp(m_old);
                m = pm < m ? pm : m;
                if (lane == 0) {
                    atomicAdd(reinterpret_cast<unsigned long long*>(&wsum[cur_win]), (unsigned long long)tot);
                    atomicMin(&wmin[cur_win], m);
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
                        atomicAdd(reinterpret_cast<unsigned long long*>(&wsum[cur_win]), (unsigned long long)tot);
                        atomicMin(&wmin[cur_win], m);
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
                    const int c0 = cov_class(d0, job.mincov, job.maxmean);
                    const int c1 = cov_class(d1, job.mincov, job.maxmean);
                    const int c2 = cov_class(d2, job.mincov, job.maxmean);
                    const int c3 = cov_class(d3, job.mincov, job.maxmean);
                    const int cp = cov_class(pl, job.mincov, job.maxmean);
                    uint32_t bm = (uint32_t)(c0 != cp) | ((uint32_t)(c1 != c0) << 1) |
                                  ((uint32_t)(c2 != c1) << 2) | ((uint32_t)(c3 != c2) << 3);
                    while (nf < rb + 256) {                        // forced breaks (quirk Q1), incl. position 0
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
                            atomicOr(&s_bmap[w], bm << sh);
                            atomicOr(&s_clo[w], (lo & bm) << sh);
                            atomicOr(&s_chi[w], (hi & bm) << sh);
                        }
                        if (lane == 0) s_hasb = 1;
                    }
                }
            }
        }
        // flush the open window segment of this wave
        {
            const uint32_t tot = (uint32_t)wave_total((int)acc);
            const int m = wave_min_dpp(mn);
            if (lane == 0) {
                atomicAdd(reinterpret_cast<unsigned long long*>(&wsum[cur_win]), (unsigned long long)tot);
                atomicMin(&wmin[cur_win], m);
            }
        }
    }
    __syncthreads();

    // ---- phase C: compact class boundaries of this tile -------------------------------------------
    phase_c<T, NT>(job, tile, tf.t0, tf.ctg, tid, lane, wv, s_bmap, s_clo, s_chi, s_wcnt, &s_hasb, &s_base);
}

}  // namespace fast
}  // namespace gd
