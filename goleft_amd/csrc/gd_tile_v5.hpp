// gd_tile_v5.hpp -- previous generation of the tile kernel, kept ONLY for A/B timing
// (GOLEFT_GD_KERNEL=v5); remove once v6 is confirmed on hardware.
#pragma once
namespace gd {
namespace v5 {
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

}  // namespace v5
}  // namespace gd
