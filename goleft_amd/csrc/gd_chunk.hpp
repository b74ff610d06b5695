// gd_chunk.hpp -- the long-read tile path (GD_PATH_CHUNK) of the per-base depth engine.
//
// Long reads (ONT / PacBio: ~10^3..10^5 CIGAR ops per record, spans of tens of
// kilobases up to megabases) defeat both other paths: the short-read tile kernel
// would re-walk a read's whole CIGAR for every tile under it, and the scatter
// path pays one device-scope atomic pair per deletion plus a serial walk per
// read.  This path keeps the tile kernel's shape -- one workgroup per tile of T
// reference positions, +-1 marks in an LDS difference array, fused scan / store
// / window / class reductions, every per-base value written to HBM exactly once
// -- and uses
//     depth(read, x) = [pos <= x < end] - [x inside one of the read's D/N ops]
// (every reference-consuming op is either counted, M/=/X, or a D/N;
// `samtools depth` semantics, /root/reference/depth/depth.go:45): per tile a read
// contributes ONE +1/-1 pair plus one -1/+1 pair per deletion or skip that reaches
// the tile.  What the tile kernel needs is therefore not the CIGAR but the read's
// DELETION LIST in reference coordinates:
//
//   DL  gd_dels_kernel   one pass over a contig's canonical CIGARs (gd_normalize.hpp: M and N
//       alternate) WHEN ITS RECORDS ARRIVE, not in gd_compute: every N op becomes {start, length}
//       (8 bytes -- what the (M, N) op pair took) at a dense per-read offset, every 64th start
//       is also a checkpoint (4 bytes per 64 deletions), and each read gets one 16-byte record
//       {pos, end, offset of its list, its length}.  Independent of the read filter (-Q, flag
//       mask), which the tile kernel applies.  Offsets need no prefix sum: the canonical CSR
//       offset o of read r gives list offset (o >> 1) + r and checkpoint offset
//       (list offset >> 6) + r, neither of which ever overlaps the next read's.
//   LT2 gd_ltile2_kernel per tile: the candidate reads (start within one maximum span before
//       the tile) are tested lane-parallel from their records; the checkpoints of up to four
//       overlapping reads are fetched in one round trip and the 64-deletion chunks that reach
//       the tile queued; lanes then load one deletion each (8 bytes, 512 bytes per chunk and
//       wave instruction) and mark it -- no op decode, no position scan, no alignment waste.
//       Then the tile kernel's phase B/C (depth/depth.go:293-323).
//
// History: walking M runs (19.7 ms on the 20x ONT genome) -> deletions of
// 64-op chunks with a checkpoint pass per gd_compute (15.4 ms) -> canonical op pairs, checkpoints at
// ingest (11.0 ms; PMC: 4.0e9 VALU + 2.9e9 SALU wave-instructions per launch, issue bound, two
// thirds of them decoding and scanning ops) -> deletion lists.
#pragma once

namespace gd {

constexpr uint32_t DL_CHUNK = 64;             // deletions per checkpoint
constexpr int DL_UNROLL = 4;                  // 64-op groups in flight per wave in gd_dels_kernel

__device__ __forceinline__ uint32_t sat_pos(uint32_t v) { return v < POS_CAP ? v : POS_CAP; }

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const uint32_t o = (uint32_t)__shfl_xor((int)v, m, WAVE);
        v = o > v ? o : v;
    }
    return v;
}

// DL: one wave per unit of 64 consecutive reads of ONE contig.
struct DelJob {
    const int32_t*  pos;
    const uint32_t* off;      // CSR offsets of the CANONICAL ops (gd_normalize.hpp)
    const uint32_t* cigar;    // canonical ops: M (0) and N (3) alternate, lengths >= 1, the last one is an M
    const uint16_t* flag;
    const uint8_t*  mapq;
    uint32_t  n_reads;
    uint32_t  n_units;        // ceil(n_reads / 64)
    uint4*    lrec;           // n_reads + 1 long-read records {pos, end, list offset, deletions}: everything the
                              // tile kernel asks about a candidate read's geometry in ONE 16-byte load
                              // (end = reference position after the last op, == pos without ops)
    uint32_t* lfq;            // n_reads: flag << 8 | MAPQ (the filter is applied per tile)
    uint2*    dl;             // deletion lists {start, length}: (n_ops >> 1) + n_reads + 1 entries
    uint32_t* dck;            // start of every 64th deletion of a read: (entries of dl >> 6) + n_reads + 1
    int32_t*  max_span;       // atomicMax of end - pos
};

__global__ __launch_bounds__(256) void gd_dels_kernel(DelJob job)
{
    const int lane = threadIdx.x & 63;
    const uint32_t unit = blockIdx.x * 4u + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (unit >= job.n_units) return;
    const uint32_t n_reads = job.n_reads;
    const uint32_t* const cigar = job.cigar;

    const uint32_t r = unit * 64u + (uint32_t)lane;
    const bool valid = r < n_reads;
    uint32_t p = 0, o0 = 0, n = 0, fq = 0;
    if (valid) {
        p = (uint32_t)job.pos[r];
        o0 = job.off[r];
        n = job.off[r + 1] - o0;
        fq = ((uint32_t)job.flag[r] << 8) | (uint32_t)job.mapq[r];
    }
    const uint32_t doff = (o0 >> 1) + r;                  // this read's deletion list ...
    const uint32_t koff = (doff >> 6) + r;                // ... and its checkpoints
    uint32_t endp = p;                                    // reference position after the last op

    // short CIGARs: lane serial
    if (n != 0u && n <= SHORT_OPS) {
        for (uint32_t k = 0; k < n; ++k) {
            const uint32_t cg = cigar[o0 + k];
            const uint32_t len = cg >> 4;
            if ((cg & 0xfu) != 0u) {                      // N: deletion number k >> 1
                job.dl[doff + (k >> 1)] = make_uint2(endp, len);
                if ((k >> 1) == 0u) job.dck[koff] = endp;
            }
            endp = sat_pos(endp + len);
        }
    }

    // long CIGARs: the wave walks one read at a time, DL_UNROLL groups of 64 ops in flight
    unsigned long long todo = __ballot(n > SHORT_OPS);
    while (todo != 0ull) {
        const int j = __ffsll((long long)todo) - 1;
        todo &= todo - 1ull;
        const uint32_t pj = (uint32_t)__builtin_amdgcn_readlane((int)p, j);
        const uint32_t oj = (uint32_t)__builtin_amdgcn_readlane((int)o0, j);
        const uint32_t nj = (uint32_t)__builtin_amdgcn_readlane((int)n, j);
        const uint32_t dj = (uint32_t)__builtin_amdgcn_readlane((int)doff, j);
        const uint32_t kj = (uint32_t)__builtin_amdgcn_readlane((int)koff, j);
        uint32_t run = pj;                                // reference position at the start of the group
        for (uint32_t b = 0; b < nj; b += DL_UNROLL * 64u) {
            uint32_t cg[DL_UNROLL];
#pragma unroll
            for (int u = 0; u < DL_UNROLL; ++u) {
                const uint32_t k = b + (uint32_t)u * 64u + (uint32_t)lane;
                cg[u] = k < nj ? cigar[oj + k] : 0u;
            }
            uint32_t len[DL_UNROLL], mx = 0;
#pragma unroll
            for (int u = 0; u < DL_UNROLL; ++u) { len[u] = cg[u] >> 4; mx |= len[u]; }
            // where each op starts: four interleaved plain 32-bit scans unless an op consumes more than
            // 2^24 bases (64 * 2^24 < 2^31: no wrap), then saturating scans
            uint32_t excl[DL_UNROLL], tot[DL_UNROLL];
            if (__builtin_amdgcn_ballot_w64(mx > (1u << 24)) == 0ull) {
                int t[DL_UNROLL];
#pragma unroll
                for (int u = 0; u < DL_UNROLL; ++u) t[u] = (int)len[u];
                static_assert(DL_UNROLL == 4, "one scan4 group");
                scan4(t[0], t[1], t[2], t[3]);
#pragma unroll
                for (int u = 0; u < DL_UNROLL; ++u) {
                    excl[u] = (uint32_t)t[u] - len[u];
                    tot[u] = (uint32_t)__builtin_amdgcn_readlane(t[u], 63);
                }
            } else {
#pragma unroll
                for (int u = 0; u < DL_UNROLL; ++u) {
                    const uint32_t inc = wave_inclusive_scan_sat(sat_pos(len[u]));
                    excl[u] = (uint32_t)wave_prev_lane((int)inc, 0);
                    tot[u] = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
                }
            }
#pragma unroll
            for (int u = 0; u < DL_UNROLL; ++u) {
                const uint32_t k = b + (uint32_t)u * 64u + (uint32_t)lane;
                if (k < nj && (cg[u] & 0xfu) != 0u) {     // N: deletion number k >> 1 (M and N alternate)
                    const uint32_t s = sat_pos(run + excl[u]);
                    const uint32_t di = k >> 1;
                    job.dl[dj + di] = make_uint2(s, len[u]);
                    if ((di & (DL_CHUNK - 1u)) == 0u) job.dck[kj + (di >> 6)] = s;
                }
                run = sat_pos(run + tot[u]);
            }
        }
        if (lane == j) endp = run;
    }
    if (valid) {
        job.lrec[r] = make_uint4(p, endp, doff, n >> 1);  // M and N alternate and the last op is an M: n >> 1 deletions
        job.lfq[r] = fq;
    }
    const uint32_t smax = wave_max_u32(n != 0u ? endp - p : 0u);
    if (lane == 0 && smax != 0u) atomicMax(job.max_span, (int32_t)smax);
}

// ---------------------------------------------------------------------------
// LT2: the long-read tile kernel.
// ---------------------------------------------------------------------------
constexpr int LQ_CAP = 128;                   // queue items per wave (a read pushes at most T / 128 + 2 per round)

// One queue item: {index of the chunk's first deletion in the contig's list, deletions in it (1..64)}
typedef uint2 LItem;

// -1 over [s, s+len) of a D/N op (absolute, saturated positions)
__device__ __forceinline__ void del_mark(int32_t* s_diff, bool del, uint32_t s, uint32_t len, int t0, int tlen)
{
    const int rs = (int)s - t0;
    const int re = (int)sat_pos(s + len) - t0;
    if (del & (re >= 0) & (rs < tlen)) {                 // reaches t0-1 or beyond, starts before the tile end
        atomicAdd(&s_diff[rs > -1 ? rs : -1], -1);
        if (re < tlen) atomicAdd(&s_diff[re], 1);
    }
}

template <int T, int NT, int OPT>
__global__ __launch_bounds__(NT) void gd_ltile2_kernel(Job job)
{
    constexpr int NW = NT / WAVE;
    constexpr int CHUNK = T / NW;
    constexpr int ROWS = CHUNK / 256;
    constexpr int NWORDS = T / 32;
    constexpr int G = 4;                   // overlapping reads whose checkpoints are fetched together
    static_assert(CHUNK % 256 == 0, "wave chunk must be whole rows");
    static_assert(T / 128 + 2 <= LQ_CAP, "one read's chunks of a round fit the queue");

    __shared__ __attribute__((aligned(16))) int32_t s_diffp[T + 4];  // [3] = index -1
    __shared__ uint32_t s_bmap[NWORDS];
    __shared__ uint32_t s_clo[NWORDS];
    __shared__ uint32_t s_chi[NWORDS];
    __shared__ __attribute__((aligned(8))) LItem s_q[NW * LQ_CAP];
    __shared__ int32_t  s_wtot[NW];
    __shared__ uint32_t s_wcnt[NW];
    __shared__ uint32_t s_hasb;
    __shared__ uint32_t s_base;
    int32_t* const s_diff = s_diffp + 4;

    const int per = (job.n_tiles + 7) >> 3;                // XCD-contiguous tile order
    const int tile = (int)(blockIdx.x & 7u) * per + (int)(blockIdx.x >> 3);
    if (tile >= job.n_tiles) return;

    const int tid = threadIdx.x;
    const int lane = tid & (WAVE - 1);
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const TileInfo ti = job.tiles[tile];
    const int32_t t0 = ti.t0;
    const int32_t tend = t0 + T < ti.length ? t0 + T : ti.length;
    const int tlen = tend - t0;

    const uint32_t nrd = ti.hi - ti.lo;
    const uint4* const grec = ti.lrec + ti.lo;
    const uint32_t* const gfq = ti.lfq + ti.lo;
    const uint2* const dl = ti.dl;
    const uint32_t* const dck = ti.dck;

    {
        const int4 z = make_int4(0, 0, 0, 0);
        int4* d4 = reinterpret_cast<int4*>(s_diffp);
#pragma unroll
        for (int i = tid; i < T / 4 + 1; i += NT) d4[i] = z;
        for (int i = tid; i < NWORDS; i += NT) { s_bmap[i] = 0; s_clo[i] = 0; s_chi[i] = 0; }
        if (tid == 0) s_hasb = 0;
    }
    __syncthreads();

    LItem* const Q = &s_q[wv * LQ_CAP];
    uint32_t qn = 0;                                       // items queued (wave uniform)

    // four queued chunks: deletion k of each in lane k ({0, 0} past the chunk)
    auto fetch4 = [&](uint32_t i, uint32_t cnt, uint2 (&d)[4]) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            d[g] = make_uint2(0u, 0u);
            if (i + (uint32_t)g < cnt) {                   // wave uniform
                const LItem it = Q[i + g];                 // same address in every lane: one broadcast read
                const uint32_t ci = (uint32_t)__builtin_amdgcn_readfirstlane((int)it.x);
                const uint32_t no = (uint32_t)__builtin_amdgcn_readfirstlane((int)it.y);
                if ((uint32_t)lane < no) d[g] = dl[ci + (uint32_t)lane];
            }
        }
    };
    auto drain = [&](uint32_t cnt) {
        __builtin_amdgcn_wave_barrier();
        uint2 dA[4];
        fetch4(0, cnt, dA);
        for (uint32_t i = 0; i < cnt; i += 4u) {
            uint2 dB[4];
            fetch4(i + 4u, cnt, dB);                       // next four are in flight while these are marked
#pragma unroll
            for (int g = 0; g < 4; ++g) del_mark(s_diff, dA[g].y != 0u, dA[g].x, dA[g].y, t0, tlen);
#pragma unroll
            for (int g = 0; g < 4; ++g) dA[g] = dB[g];
        }
        __builtin_amdgcn_wave_barrier();
    };

    // ---- phase A: candidate reads -> deletion chunks -> LDS +1/-1 -------------
    // Candidate i of a batch of NT belongs to wave i % NW: the overlapping reads (mostly
    // the latest starters) spread evenly over the waves.  G "slots" each hold one
    // overlapping read and the block of 64 chunks under examination; a read with more
    // than 64 chunks (4096 deletions) keeps its slot until its chunks pass the tile end.
    for (uint32_t base = 0; base < nrd; base += (uint32_t)NT) {
        const uint32_t idx = base + (uint32_t)(lane * NW + wv);
        const bool inb = idx < nrd;
        // one round trip per candidate: its record and its filter word
        uint4 rc = make_uint4(0x7fffffffu, 0u, 0u, 0u);
        uint32_t fq = 0;
        if (inb) { rc = grec[idx]; fq = gfq[idx]; }
        const int32_t p = (int32_t)rc.x;
        const int32_t e = inb ? (int32_t)rc.y : -1;
        // reaches t0-1 or beyond, and passes the read filter of `samtools depth`
        const bool hit = e >= t0 && p < tend && ((fq >> 8) & job.flag_mask) == 0u && (int)(fq & 0xffu) >= job.Q;
        if (hit) {
            // the read's own +1 / -1 (its D/N ops subtract below)
            const int rs = p - t0;
            atomicAdd(&s_diff[rs > -1 ? rs : -1], 1);
            if (e < tend) atomicAdd(&s_diff[e - t0], -1);
        }
        unsigned long long m = __builtin_amdgcn_ballot_w64(hit && rc.w != 0u);   // reads with deletions

        bool act[G];
        uint32_t dj[G], nj[G], ej[G], nch[G], cb[G];
        const uint32_t* ckj[G];
#pragma unroll
        for (int g = 0; g < G; ++g) { act[g] = false; dj[g] = nj[g] = ej[g] = nch[g] = cb[g] = 0; ckj[g] = dck; }
        bool any = false;
        while (m != 0ull || any) {
            // free slots take the next overlapping reads
#pragma unroll
            for (int g = 0; g < G; ++g) {
                if (act[g] || m == 0ull) continue;            // wave uniform
                const int j = __ffsll((long long)m) - 1;
                m &= m - 1ull;
                dj[g] = (uint32_t)__builtin_amdgcn_readlane((int)rc.z, j);
                nj[g] = (uint32_t)__builtin_amdgcn_readlane((int)rc.w, j);
                ej[g] = (uint32_t)__builtin_amdgcn_readlane(e, j);
                const uint32_t rj = ti.lo + (uint32_t)__builtin_amdgcn_readlane((int)idx, j);
                nch[g] = (nj[g] + DL_CHUNK - 1u) >> 6;
                ckj[g] = dck + ((dj[g] >> 6) + rj);
                cb[g] = 0;
                act[g] = true;
                if (nch[g] > 64u) {
                    // more than 4096 deletions: a strided probe of the (monotone) checkpoints finds
                    // the block of 64 chunks where the tile begins
                    const uint32_t stride = (nch[g] + 63u) >> 6;
                    const uint32_t pq = (uint32_t)lane * stride;
                    const uint32_t pv = pq < nch[g] ? ckj[g][pq] : POS_CAP;
                    // chunks before the last probe that starts before t0 end before t0
                    const int pc = __popcll(__builtin_amdgcn_ballot_w64(pq < nch[g] && (int)pv < t0));
                    cb[g] = pc > 1 ? (uint32_t)(pc - 1) * stride : 0u;
                }
            }
            // the checkpoints of all slots in one round trip: lane q owns chunk cb + q, which starts at
            // c0 and whose deletions all end before c1 = the start of the next chunk (or the read's end)
            uint32_t c0[G], c1[G];
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const uint32_t q = cb[g] + (uint32_t)lane;
                c0[g] = (act[g] && q < nch[g]) ? ckj[g][q] : POS_CAP;
            }
#pragma unroll
            for (int g = 0; g < G; ++g) {                  // c0 of the next lane (wave_shl:1); lane 63: the chunk after
                const uint32_t q = cb[g] + (uint32_t)lane;
                uint32_t nx = (uint32_t)__builtin_amdgcn_update_dpp((int)POS_CAP, (int)c0[g], 0x130, 0xf, 0xf, false);
                if (lane == 63) nx = (act[g] && q + 1u < nch[g]) ? ckj[g][q + 1u] : POS_CAP;
                c1[g] = (q + 1u < nch[g]) ? nx : ej[g];
            }
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const uint32_t q = cb[g] + (uint32_t)lane;
                const unsigned long long cm = __builtin_amdgcn_ballot_w64(act[g] && q < nch[g] &&
                                                                          (int)c0[g] < tend && (int)c1[g] >= t0);
                const uint32_t cnt = (uint32_t)__popcll(cm);
                if (cnt == 0u) continue;                     // wave uniform
                if (qn + cnt > (uint32_t)LQ_CAP) { drain(qn); qn = 0; }
                if ((cm >> lane) & 1ull) {
                    const uint32_t rk = qn + __builtin_amdgcn_mbcnt_hi((uint32_t)(cm >> 32),
                                                 __builtin_amdgcn_mbcnt_lo((uint32_t)cm, 0u));
                    const uint32_t first = q * DL_CHUNK, left = nj[g] - first;
                    Q[rk] = make_uint2(dj[g] + first, left < DL_CHUNK ? left : DL_CHUNK);
                }
                qn += cnt;
            }
            // a slot is done when its chunks are exhausted or start at/after the tile end
            // (checkpoints only grow)
            any = false;
#pragma unroll
            for (int g = 0; g < G; ++g) {
                if (!act[g]) continue;
                const bool past = (int)(uint32_t)__builtin_amdgcn_readlane((int)c0[g], 63) >= tend;
                if (cb[g] + 64u >= nch[g] || past) act[g] = false;
                else { cb[g] += 64u; any = true; }
            }
        }
    }
    if (qn != 0u) drain(qn);
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
        // a read covers a position at most once: depth <= candidate reads
        const bool wide = nrd >= (1u << 22);
        if (tlen == T && !wide) {
            if constexpr (ROWS == 4) phase_b_rows_full<ROWS, OPT>(B, job.w_magic, job.w_shift, job.s_magic, job.s_shift);
            else                     phase_b_rows<ROWS, true, false, OPT>(B);
        } else {
            phase_b_rows<ROWS, false, true, OPT>(B);
        }
    }
    __syncthreads();

    phase_c<T, NT>(job, tile, t0, ti.ctg, tid, lane, wv, s_bmap, s_clo, s_chi, s_wcnt, &s_hasb, &s_base);
}

}  // namespace gd
