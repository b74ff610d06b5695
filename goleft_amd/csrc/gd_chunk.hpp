// gd_chunk.hpp -- the long-read tile path (GD_PATH_CHUNK) of the per-base depth engine.
//
// Long reads (ONT / PacBio: ~10^3..10^5 CIGAR ops per record, spans of tens of
// kilobases up to megabases) defeat both other paths: the short-read tile kernel
// would re-walk a read's whole CIGAR for every tile under it, and the scatter
// path pays one device-scope atomic pair per deletion plus a serial walk per
// read.  This path keeps the tile kernel's shape -- one workgroup per tile of T
// reference positions, +-1 marks in an LDS difference array, fused scan / store
// / window / class reductions, every per-base value written to HBM exactly once
// -- and makes a read's CIGAR addressable by reference position:
//
//   CK  gd_ckpt_kernel   one pass over a contig's CIGARs WHEN ITS RECORDS ARRIVE (with the
//       canonical CIGARs of gd_normalize.hpp, which it reads: about half the ops of an
//       ONT-like read; not part of gd_compute): for every chunk of 64 ops of a read, the
//       reference position at which the chunk starts (a checkpoint, 4 bytes per 64 ops), the
//       read's end position, and the largest span.  Independent of the read filter (-Q, flag
//       mask), which the tile kernel applies.  Checkpoint slots need no prefix sum: read r
//       with CSR offset o uses slots (o >> 6) + r ... which never overlap
//       (ceil(n/64) <= (n >> 6) + 1).
//   LT2 gd_ltile2_kernel per tile: the candidate reads (start within one maximum
//       span before the tile; 8 bytes each: start, end) are tested lane-parallel
//       and contribute one +1/-1 pair; the 64-op chunks that reach the tile are
//       expanded for their D/N ops only (`samtools depth` semantics,
//       /root/reference/depth/depth.go:45; details at the kernel).  Then the tile
//       kernel's phase B/C (depth/depth.go:293-323).
#pragma once

namespace gd {

constexpr uint32_t CK_OPS = 64;               // CIGAR ops per checkpoint chunk
constexpr int CK_UNROLL = 8;                  // chunks in flight per wave in gd_ckpt_kernel

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

// CK: one wave per unit of 64 consecutive reads of ONE contig.
struct CkJob {
    const int32_t*  pos;
    const uint32_t* off;      // CSR offsets of `cigar` (the canonical arrays when the contig has them)
    const uint32_t* cigar;
    uint32_t  n_reads;
    uint32_t  n_units;        // ceil(n_reads / 64)
    const uint16_t* flag;
    const uint8_t*  mapq;
    uint32_t* ck;             // (n_ops >> 6) + n_reads + 1 slots
    uint4*    lrec;           // n_reads + 1 long-read records {pos, end, off, flag << 8 | MAPQ}: everything the
                              // tile kernel asks about a candidate read in ONE 16-byte load (end = reference
                              // position after the last op, == pos without ops; [n_reads] = {0, 0, n_ops, 0})
    int32_t*  max_span;       // atomicMax of end - pos
};

__global__ __launch_bounds__(256) void gd_ckpt_kernel(CkJob job)
{
    const int lane = threadIdx.x & 63;
    const uint32_t unit = blockIdx.x * 4u + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (unit >= job.n_units) return;
    const uint32_t n_reads = job.n_reads;
    const uint32_t* const cigar = job.cigar;
    uint32_t* const ck = job.ck;

    const uint32_t r = unit * 64u + (uint32_t)lane;
    const bool valid = r < n_reads;
    uint32_t p = 0, o0 = 0, n = 0, fq = 0;
    if (valid) {
        p = (uint32_t)job.pos[r];
        o0 = job.off[r];
        n = job.off[r + 1] - o0;
        fq = ((uint32_t)job.flag[r] << 8) | (uint32_t)job.mapq[r];
    }
    const bool keep = n != 0u;
    uint32_t endp = p;                                    // reference position after the last op

    // short CIGARs: lane serial, a single chunk
    if (keep && n <= SHORT_OPS) {
        for (uint32_t k = 0; k < n; ++k) {
            const uint32_t cg = cigar[o0 + k];
            const uint32_t op = cg & 0xf, len = cg >> 4;
            if ((0x18du >> op) & 1u) endp = sat_pos(endp + len);      // M D N = X
        }
    }
    if (keep) ck[(o0 >> 6) + r] = p;                      // chunk 0 starts at POS

    // long CIGARs: the wave walks one read at a time, CK_UNROLL chunks in flight
    unsigned long long todo = __ballot(keep && n > SHORT_OPS);
    while (todo != 0ull) {
        const int j = __ffsll((long long)todo) - 1;
        todo &= todo - 1ull;
        const uint32_t pj = (uint32_t)__builtin_amdgcn_readlane((int)p, j);
        const uint32_t oj = (uint32_t)__builtin_amdgcn_readlane((int)o0, j);
        const uint32_t nj = (uint32_t)__builtin_amdgcn_readlane((int)n, j);
        const uint32_t rj = (uint32_t)__builtin_amdgcn_readlane((int)r, j);
        uint32_t* const ckj = ck + ((oj >> 6) + rj);
        uint32_t run = pj;
        for (uint32_t b = 0; b < nj; b += CK_UNROLL * CK_OPS) {
            uint32_t cg[CK_UNROLL];
#pragma unroll
            for (int u = 0; u < CK_UNROLL; ++u) {
                const uint32_t k = b + (uint32_t)u * CK_OPS + (uint32_t)lane;
                cg[u] = k < nj ? cigar[oj + k] : 0u;
            }
            // reference bases each op consumes; the chunk totals come from eight interleaved plain
            // 32-bit scans (scan4 twice) unless an op consumes more than 2^24 bases (64 * 2^24 <
            // 2^31: no wrap), then from saturating scans
            uint32_t cons[CK_UNROLL];
            uint32_t mx = 0;
#pragma unroll
            for (int u = 0; u < CK_UNROLL; ++u) {
                const uint32_t op = cg[u] & 0xf, len = cg[u] >> 4;
                cons[u] = ((0x18du >> op) & 1u) ? len : 0u;
                mx |= cons[u];
            }
            uint32_t tot[CK_UNROLL];
            if (__builtin_amdgcn_ballot_w64(mx > (1u << 24)) == 0ull) {
                int t[CK_UNROLL];
#pragma unroll
                for (int u = 0; u < CK_UNROLL; ++u) t[u] = (int)cons[u];
                static_assert(CK_UNROLL == 8, "two scan4 groups");
                scan4(t[0], t[1], t[2], t[3]);
                scan4(t[4], t[5], t[6], t[7]);
#pragma unroll
                for (int u = 0; u < CK_UNROLL; ++u) tot[u] = (uint32_t)__builtin_amdgcn_readlane(t[u], 63);
            } else {
#pragma unroll
                for (int u = 0; u < CK_UNROLL; ++u)
                    tot[u] = (uint32_t)__builtin_amdgcn_readlane((int)wave_inclusive_scan_sat(sat_pos(cons[u])), 63);
            }
#pragma unroll
            for (int u = 0; u < CK_UNROLL; ++u) {
                if (b + (uint32_t)u * CK_OPS >= nj) break;            // uniform
                run = sat_pos(run + tot[u]);
                const uint32_t nxt = (b >> 6) + (uint32_t)u + 1u;     // the chunk that starts here
                if (nxt * CK_OPS < nj && lane == 0) ckj[nxt] = run;
            }
        }
        if (lane == j) endp = run;
    }
    if (valid) {
        job.lrec[r] = make_uint4(p, endp, o0, fq);
        if (r + 1u == n_reads) job.lrec[n_reads] = make_uint4(0u, 0u, o0 + n, 0u);
    }
    const uint32_t smax = wave_max_u32(keep ? endp - p : 0u);
    if (lane == 0 && smax != 0u) atomicMax(job.max_span, (int32_t)smax);
}

// ---------------------------------------------------------------------------
// LT2: the long-read tile kernel.
//
// Its predecessor walked one overlapping read at a time per wave, marking M runs, and
// spent ~120 wave-instructions per 64 CIGAR ops; at ONT shape (4.8e9 ops,
// every chunk expanded ~1.2 times) that was instruction-issue bound.  Here:
//   * depth of a read = [pos <= x < end] - [x inside one of its D/N ops]
//     (every reference-consuming op is either counted, M/=/X, or a D/N): the
//     read contributes ONE +1/-1 pair per tile (lane-parallel, from the start /
//     end arrays the checkpoint pass wrote) and each D/N op one -1/+1 pair.
//     Only a quarter of ONT-like ops are deletions, the M runs need no marks at
//     all, and no neighbour-op logic is left;
//   * work is split into producing and consuming a per-wave queue of ITEMS
//     {first op index, reference position of the first op, op count <= 128}
//     (two consecutive 64-op checkpoint chunks): the checkpoints of up to four
//     overlapping reads are fetched in one round trip, lanes whose chunk pair
//     reaches the tile push an item (ballot + mbcnt, no atomics); items are
//     expanded four at a time, lane k holding the PAIR of ops 2k, 2k+1 of the item (one
//     8-byte load; 512 bytes per item and wave instruction), the next four items' loads
//     in flight while the current ones expand, and ONE wave scan per item over the pairs'
//     reference-consuming lengths (four interleave in scan4), seeded with the item's
//     checkpoint.  Canonical CIGARs alternate M and N, so a pair holds exactly one D/N op:
//     every lane has one mark to make, none idles on an M (a pair of two D/N ops -- only
//     when normalisation is off -- takes a second, wave-uniformly skipped, mark).
// Reference positions use plain 32-bit scans whenever every op of the four
// items consumes <= 2^23 bases (64 * 2^24 + 2^31 < 2^32: no wrap); items with
// a longer D/N op take a saturating scan.  Integer adds commute, so the
// per-base result equals M-run marking bit for bit.
// ---------------------------------------------------------------------------
constexpr int LQ_CAP = 128;                   // queue items per wave: one round of 4 slots pushes <= 4 * 32
constexpr uint32_t LQ_OPS = 2 * CK_OPS;       // ops per item

// One queue item: {index of the first op in the contig's CIGAR array, reference position
// of op 0, reference position of op 64, ops in the item (1..128)}
typedef uint4 LItem;

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

// Reference positions of the op pairs of four items (lane k = ops 2k, 2k+1; cons = bases the pair consumes).
__device__ __forceinline__ void chunk_pos4(const uint32_t (&cons)[4], const uint32_t (&st)[4], bool big,
                                           uint32_t (&pos)[4])
{
    if (!big) {
        int incl[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) incl[g] = (int)cons[g];
        scan4(incl[0], incl[1], incl[2], incl[3]);
#pragma unroll
        for (int g = 0; g < 4; ++g) pos[g] = sat_pos(st[g] + ((uint32_t)incl[g] - cons[g]));
    } else {
        // rare: saturating scans (positions stay exact up to POS_CAP, then stick there)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const uint32_t inc = wave_inclusive_scan_sat(sat_pos(cons[g]));
            pos[g] = sat_pos(st[g] + (uint32_t)wave_prev_lane((int)inc, 0));
        }
    }
}

// Four items: lane k holds ops 2k (a) and 2k+1 (b) of each; 0 = nothing.  sa: reference position of op 0.
__device__ __forceinline__ void expand4_lds(const uint32_t (&a)[4], const uint32_t (&b)[4],
                                            const uint32_t (&sa)[4], int t0, int tlen, int32_t* s_diff)
{
    uint32_t la[4], lb[4], ca[4], cp[4], pa[4];
    bool da[4], db[4];
    uint32_t mx = 0;
    bool two = false;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const uint32_t oa = a[g] & 0xf, ob = b[g] & 0xf;
        la[g] = a[g] >> 4; lb[g] = b[g] >> 4;
        ca[g] = ((0x18du >> oa) & 1u) ? la[g] : 0u;                       // M D N = X
        const uint32_t cb = ((0x18du >> ob) & 1u) ? lb[g] : 0u;
        da[g] = ((0xcu >> oa) & 1u) && la[g] != 0;                        // D N
        db[g] = ((0xcu >> ob) & 1u) && lb[g] != 0;
        mx |= ca[g] | cb;
        cp[g] = sat_pos(ca[g]) + sat_pos(cb);                             // < 2^32
        two = two || (da[g] && db[g]);
    }
    const bool big = __builtin_amdgcn_ballot_w64(mx > (1u << 23)) != 0ull;   // wave uniform
    chunk_pos4(cp, sa, big, pa);
    // one mark per pair: the pair's D/N op (canonical CIGARs alternate, so there is exactly one)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const uint32_t ps = da[g] ? pa[g] : sat_pos(pa[g] + ca[g]);
        del_mark(s_diff, da[g] | db[g], ps, da[g] ? la[g] : lb[g], t0, tlen);
    }
    if (__builtin_amdgcn_ballot_w64(two) != 0ull) {                      // original CIGARs only: D next to N
#pragma unroll
        for (int g = 0; g < 4; ++g)
            del_mark(s_diff, da[g] && db[g], sat_pos(pa[g] + ca[g]), lb[g], t0, tlen);
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

    __shared__ __attribute__((aligned(16))) int32_t s_diffp[T + 4];  // [3] = index -1
    __shared__ uint32_t s_bmap[NWORDS];
    __shared__ uint32_t s_clo[NWORDS];
    __shared__ uint32_t s_chi[NWORDS];
    __shared__ __attribute__((aligned(16))) LItem s_q[NW * LQ_CAP];
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
    const uint32_t* const ck = ti.ck;
    const int32_t t0 = ti.t0;
    const int32_t tend = t0 + T < ti.length ? t0 + T : ti.length;
    const int tlen = tend - t0;

    const uint32_t nrd = ti.hi - ti.lo;
    const uint4* const grec = ti.lrec + ti.lo;
    const uint32_t* const cigar = ti.cigar;

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

    // four queued items: ops 2k, 2k+1 of each in lane k (0 past the item) and the items' start positions
    typedef uint32_t __attribute__((ext_vector_type(2), aligned(4))) pair_u;   // a read's ops start at any dword
    auto fetch4 = [&](uint32_t i, uint32_t cnt, uint32_t (&a)[4], uint32_t (&b)[4], uint32_t (&sa)[4]) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            a[g] = 0; b[g] = 0; sa[g] = POS_CAP;
            if (i + (uint32_t)g < cnt) {                   // wave uniform
                const LItem it = Q[i + g];                 // same address in every lane: one broadcast read
                const uint32_t ci = (uint32_t)__builtin_amdgcn_readfirstlane((int)it.x);
                const uint32_t no = (uint32_t)__builtin_amdgcn_readfirstlane((int)it.w);
                sa[g] = (uint32_t)__builtin_amdgcn_readfirstlane((int)it.y);
                const uint32_t* src = cigar + ci + 2u * (uint32_t)lane;
                if (2u * (uint32_t)lane + 1u < no) {
                    const pair_u v = *reinterpret_cast<const pair_u*>(src);
                    a[g] = v.x; b[g] = v.y;
                } else if (2u * (uint32_t)lane < no) {
                    a[g] = src[0];
                }
            }
        }
    };
    auto drain = [&](uint32_t cnt) {
        __builtin_amdgcn_wave_barrier();
        uint32_t aA[4], bA[4], saA[4];
        fetch4(0, cnt, aA, bA, saA);
        for (uint32_t i = 0; i < cnt; i += 4u) {
            uint32_t aB[4], bB[4], saB[4];
            fetch4(i + 4u, cnt, aB, bB, saB);              // next four are in flight while these expand
            expand4_lds(aA, bA, saA, t0, tlen, s_diff);
#pragma unroll
            for (int g = 0; g < 4; ++g) { aA[g] = aB[g]; bA[g] = bB[g]; saA[g] = saB[g]; }
        }
        __builtin_amdgcn_wave_barrier();
    };

    // ---- phase A: candidate reads -> chunk items -> LDS +1/-1 ----------------
    // Candidate i of a batch of NT belongs to wave i % NW: the overlapping reads (mostly
    // the latest starters) spread evenly over the waves.  G "slots" each hold one
    // overlapping read and the block of 64 chunks under examination; a read with more
    // than 64 chunks keeps its slot until its chunks pass the tile end.
    for (uint32_t base = 0; base < nrd; base += (uint32_t)NT) {
        const uint32_t idx = base + (uint32_t)(lane * NW + wv);
        const bool inb = idx < nrd;
        // one round trip per candidate: its record and the next one's CIGAR offset
        uint4 rc = make_uint4(0x7fffffffu, 0u, 0u, 0u);
        uint32_t o1 = 0;
        if (inb) { rc = grec[idx]; o1 = grec[idx + 1u].z; }
        const int32_t p = (int32_t)rc.x;
        const int32_t e = inb ? (int32_t)rc.y : -1;
        // reaches t0-1 or beyond, and passes the read filter of `samtools depth`
        const bool hit = e >= t0 && p < tend && ((rc.w >> 8) & job.flag_mask) == 0u && (int)(rc.w & 0xffu) >= job.Q;
        uint32_t o0 = 0, n = 0;
        if (hit) {
            o0 = rc.z; n = o1 - o0;
            // the read's own +1 / -1 (its D/N ops subtract below)
            const int rs = p - t0;
            atomicAdd(&s_diff[rs > -1 ? rs : -1], 1);
            if (e < tend) atomicAdd(&s_diff[e - t0], -1);
        }
        unsigned long long m = __builtin_amdgcn_ballot_w64(hit);

        bool act[G];
        uint32_t oj[G], nj[G], ej[G], nch[G], cb[G];
        const uint32_t* ckj[G];
#pragma unroll
        for (int g = 0; g < G; ++g) { act[g] = false; oj[g] = nj[g] = ej[g] = nch[g] = cb[g] = 0; ckj[g] = ck; }
        bool any = false;
        while (m != 0ull || any) {
            // free slots take the next overlapping reads
#pragma unroll
            for (int g = 0; g < G; ++g) {
                if (act[g] || m == 0ull) continue;            // wave uniform
                const int j = __ffsll((long long)m) - 1;
                m &= m - 1ull;
                oj[g] = (uint32_t)__builtin_amdgcn_readlane((int)o0, j);
                nj[g] = (uint32_t)__builtin_amdgcn_readlane((int)n, j);
                ej[g] = (uint32_t)__builtin_amdgcn_readlane(e, j);
                const uint32_t rj = ti.lo + (uint32_t)__builtin_amdgcn_readlane((int)idx, j);
                nch[g] = (nj[g] + CK_OPS - 1u) >> 6;
                ckj[g] = ck + ((oj[g] >> 6) + rj);
                cb[g] = 0;
                act[g] = true;
                if (nch[g] > 64u) {
                    // more than 4096 ops: a strided probe of the (monotone) checkpoints finds
                    // the block of 64 chunks where the tile begins
                    const uint32_t stride = (nch[g] + 63u) >> 6;
                    const uint32_t pq = (uint32_t)lane * stride;
                    const uint32_t pv = pq < nch[g] ? ckj[g][pq] : POS_CAP;
                    // chunks before the last probe that starts before t0 end before t0
                    const int pc = __popcll(__builtin_amdgcn_ballot_w64(pq < nch[g] && (int)pv < t0));
                    cb[g] = pc > 1 ? (uint32_t)(pc - 1) * stride : 0u;
                }
            }
            // the checkpoints of all slots in one round trip; even lanes own the pair of
            // chunks (q, q+1): c0 = where it starts, c2 = where it ends
            uint32_t c0[G], c1[G], c2[G];
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const uint32_t q = cb[g] + (uint32_t)lane;
                c0[g] = (act[g] && q < nch[g]) ? ckj[g][q] : POS_CAP;
                c2[g] = (act[g] && q + 2u < nch[g]) ? ckj[g][q + 2u] : ej[g];
            }
#pragma unroll
            for (int g = 0; g < G; ++g)                    // start of chunk q+1 = c0 of the next lane (wave_shl:1)
                c1[g] = (uint32_t)__builtin_amdgcn_update_dpp((int)POS_CAP, (int)c0[g], 0x130, 0xf, 0xf, false);
            unsigned long long cm[G];
            uint32_t cnt[G], tot = 0;
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const uint32_t q = cb[g] + (uint32_t)lane;
                cm[g] = __builtin_amdgcn_ballot_w64(act[g] && (lane & 1) == 0 && q < nch[g] &&
                                                    (int)c0[g] < tend && (int)c2[g] >= t0);
                cnt[g] = (uint32_t)__popcll(cm[g]);
                tot += cnt[g];
            }
            if (qn + tot > (uint32_t)LQ_CAP) { drain(qn); qn = 0; }   // tot <= 4 * 32 = LQ_CAP
#pragma unroll
            for (int g = 0; g < G; ++g) {
                if ((cm[g] >> lane) & 1ull) {
                    const uint32_t rk = qn + __builtin_amdgcn_mbcnt_hi((uint32_t)(cm[g] >> 32),
                                                 __builtin_amdgcn_mbcnt_lo((uint32_t)cm[g], 0u));
                    const uint32_t first = (cb[g] + (uint32_t)lane) * CK_OPS, left = nj[g] - first;
                    Q[rk] = make_uint4(oj[g] + first, c0[g], c1[g], left < LQ_OPS ? left : LQ_OPS);
                }
                qn += cnt[g];
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
