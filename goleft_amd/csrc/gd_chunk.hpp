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
//   DL  gd_dels_raw_kernel   ONE pass over a contig's CIGARs as they arrived, at the first compute of the records:
//       every D/N op becomes {start, length} (8 bytes) at a dense per-read offset, each read gets one 16-byte record
//       {pos, end, offset of its list, offset of its tile index} and -- in the same pass -- its TILE INDEX: for every
//       4096-base boundary b the read spans (from the one at or before pos to the one after end), the number of its
//       deletions that start before b: 4 bytes per (read, 4 kb of reference), 80 MB for a 20x ONT genome.  A tile finds
//       the deletions of an overlapping read that can touch it with TWO lookups (no checkpoint search, no chunk that
//       merely brushes the tile).  Independent of the read filter (-Q, flag mask), which the tile kernel applies.  Neither
//       the list offsets nor the index offsets need a prefix sum: the CSR offset o of read r gives (o >> 1) + r and
//       (o >> 6) + 3 r, which never overlap the next read's.
//   LT2 gd_ltile2_kernel per tile: the candidate reads (start within one maximum span before
//       the tile) are tested lane-parallel from their records; a lane whose read overlaps looks
//       its deletion range up and queues it in pieces of 64; lanes then load one deletion each
//       (8 bytes, 512 bytes per piece and wave instruction) and mark it -- no op decode, no
//       position scan.  Then the tile kernel's phase B/C (depth/depth.go:293-323).
//
// History: walking M runs (19.7 ms on the 20x ONT genome) -> deletions of
// 64-op chunks with a checkpoint pass per gd_compute (15.4 ms) -> canonical op pairs, checkpoints at
// ingest (11.0 ms; PMC: 4.0e9 VALU + 2.9e9 SALU wave-instructions per launch, issue bound, two
// thirds of them decoding and scanning ops) -> deletion lists found through 64-deletion checkpoints
// (9.4 ms) -> deletion lists found through the per-read tile index.
#pragma once

namespace gd {

constexpr uint32_t DL_CHUNK = 64;             // deletions per queued piece
constexpr int PT_SHIFT = 12;                  // the tile index has one entry per 4096 reference positions
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
    const uint32_t* off;      // CSR offsets of the ops as they arrived
    const uint32_t* cigar;    // BAM-encoded ops, any form
    const uint16_t* flag;
    const uint8_t*  mapq;
    uint32_t  n_reads;
    uint32_t  n_units;        // ceil(n_reads / 64)
    uint4*    lrec;           // n_reads + 1 long-read records {pos, end, list offset, tile index offset}: everything
                              // the tile kernel asks about a candidate read's geometry in ONE 16-byte load
                              // (end = reference position after the last op, == pos without ops; the last field
                              // is filled by gd_ptile_fill_kernel, PT_NONE: the read has no deletion)
    uint32_t* lfq;            // n_reads: flag << 8 | MAPQ (the filter is applied per tile)
    uint2*    dl;             // deletion lists {start, length}: (n_ops >> 1) + n_reads + 1 entries
    uint32_t* ndel;           // n_reads: deletions of each read
    uint32_t* del_total;      // out: deletions of the contig
    int32_t*  max_span;       // atomicMax of end - pos
    // the tile index (PT kernels below)
    uint32_t* pck;            // the index: (n_ops >> 6) + 3 n_reads + 4 entries, read r's at pt_slot(off[r], r)
    uint32_t* total;          // out: [n_jobs] deletions of the contig, [2 n_jobs] its largest span ([0 .. n_jobs) is reserved: the layout is the host's)
};
constexpr uint32_t PT_NONE = 0xffffffffu;
constexpr uint32_t PT_SEARCH = 0xfffffffeu;   // the read has deletions but no tile index (its slots were too few): the tile kernel bisects its list

// Round 5: the tile index is filled BY THE PASS THAT WRITES THE DELETION LISTS.  It used to be three more launches (entries
// per read, a scan over the 64-read units, a fill pass in which every read bisected its own list per boundary) with a host
// synchronisation and an allocation between them -- 1.6-2.3 ms of an 8.2 ms preparation, and the only reason the first
// compute's enqueue could not run through.  The walk knows where every deletion starts as it goes: entry k of a read =
// deletions that start before boundary ((pos >> 12) + k) << 12, and a group of ops that covers the reference positions
// (cur, cur + tot] fixes exactly the boundaries inside that range -- deletions of earlier groups start before them, later
// ones behind.  A group crosses a boundary once in five (4096 bases, ~13 per op): one scalar compare per group otherwise.
// Where the entries live needs no prefix sum either: read r of a contig owns the slots (o >> 6) + 3 r .. ((o + n) >> 6) + 3 r + 2
// (o its CSR offset, n its ops) -- enough whenever its ops average <= 64 reference bases; a read that needs more (a
// spliced read: three ops, 100 kb) is marked PT_SEARCH and the tile kernel bisects its (short) list.
// entries of a read's tile index: boundaries ((pos >> 12) + k) << 12, k = 0 .. K - 1, from the one at or before pos to the one
// behind end (reads without deletions have none)
__device__ __forceinline__ uint32_t pt_entries(uint32_t p, uint32_t e, uint32_t n_del)
{
    return (n_del != 0u && e >= p) ? (e >> PT_SHIFT) - (p >> PT_SHIFT) + 2u : 0u;   // (e < p: a negative POS)
}
__device__ __forceinline__ uint32_t pt_slot(uint32_t o0, uint32_t r) { return (o0 >> 6) + 3u * r; }
__device__ __forceinline__ uint32_t pt_cap(uint32_t o0, uint32_t n) { return ((o0 + n) >> 6) - (o0 >> 6) + 3u; }

// The long-read contigs of one batch: job j owns the 64-read units [ubeg[j], ubeg[j + 1]) of the batch.
struct DelBatch {
    const DelJob*   jobs;
    const uint32_t* ubeg;     // n_jobs + 1
    uint32_t n_jobs;
    uint32_t n_units;
};


// ---- the same structures straight from the records AS THEY ARRIVED ------------------------------------------------
// One pass over the original CIGARs (round 2 went through a rewritten, "canonical" copy of the CIGARs first: a 20x ONT
// genome's 19 GB of ops read twice and 10 GB written and read again before the first deletion list existed -- 24 + 3 ms in
// front of a 4.6 ms tile kernel; a run computes its input once).  The merging walk below keeps that copy's rules without
// writing it: I/S/H/P and zero-length ops vanish, neighbouring D/N ops merge into ONE
// deletion, one that no M follows is dropped, and a read ends where its last M ends.

// one lane, op by op (short CIGARs; ops longer than 2^22 bases or runs that overflow)
__device__ __forceinline__ uint32_t serial_dels(const uint32_t* __restrict__ ops, uint32_t n, uint32_t pos,
                                                uint2* __restrict__ out, uint32_t& endp)
{
    uint32_t x = pos, w = 0, last_end = pos;
    uint32_t run_kind = 2u, run_start = pos, run_len = 0u;     // the open run: 0 counted (M = X), 1 skipped (D N), 2 none yet
    for (uint32_t k = 0; k < n; ++k) {
        const uint32_t cg = ops[k], op = cg & 0xfu, len = cg >> 4;
        if (!((0x18du >> op) & 1u) || len == 0u) continue;
        const uint32_t kind = ((0x181u >> op) & 1u) ? 0u : 1u;
        if (kind == run_kind) run_len = sat_pos(run_len + len);
        else {
            if (run_kind == 1u) out[w++] = make_uint2(run_start, run_len);   // a deletion an M follows
            if (run_kind == 0u) last_end = x;
            run_kind = kind; run_start = x; run_len = len;
        }
        x = sat_pos(x + len);
    }
    if (run_kind == 0u) last_end = x;
    endp = last_end;
    return w;
}

// a whole wave, 64 ops at a time (wave_canonical's run logic: heads from ballots, run lengths as differences of one
// wave prefix sum).  `overflow`: an op longer than 2^22 bases or a run past the 28-bit range -- the caller walks that
// read with serial_dels (what it wrote before noticing is overwritten).  A run's length is a plain 32-bit sum: it cannot
// pass the position cap the walk checks (a deletion is {start, length} in 32 bits each; no 28-bit op field to fit).
__device__ __forceinline__ uint32_t wave_dels(const uint32_t* __restrict__ ops, uint32_t n, int lane, uint32_t pos,
                                              uint2* __restrict__ out, uint32_t& endp, bool& overflow)
{
    uint32_t open_kind = 2u, open_len = 0u, open_start = pos, cur = pos, w = 0u;   // wave uniform
    overflow = false;
    const unsigned long long below = (1ull << lane) - 1ull;
    // DL_UNROLL groups of 64 ops per batch, and the NEXT batch's loads are issued before this one is worked on: one
    // group at a time left the wave waiting a memory round trip per 256 bytes, one batch at a time still one per KB
    // (10.4 ms for a 20x genome's 19 GB of ops: latency, not bytes)
    uint32_t nxt[DL_UNROLL];
#pragma unroll
    for (int u = 0; u < DL_UNROLL; ++u) {
        const uint32_t k = (uint32_t)u * 64u + (uint32_t)lane;
        nxt[u] = k < n ? ops[k] : 0u;
    }
    for (uint32_t b0 = 0; b0 < n; b0 += DL_UNROLL * 64u) {
        uint32_t cgv[DL_UNROLL];
#pragma unroll
        for (int u = 0; u < DL_UNROLL; ++u) cgv[u] = nxt[u];
        if (b0 + DL_UNROLL * 64u < n) {
#pragma unroll
            for (int u = 0; u < DL_UNROLL; ++u) {
                const uint32_t k = b0 + (uint32_t)(DL_UNROLL + u) * 64u + (uint32_t)lane;
                nxt[u] = k < n ? ops[k] : 0u;
            }
        }
        // what does not depend on the run carried from group to group, for all four groups at once: kept / kind masks,
        // one overflow test, and the four prefix sums INTERLEAVED (scan4: no DPP hazard stalls between the steps)
        uint32_t lenv[DL_UNROLL], klv[DL_UNROLL], kindv[DL_UNROLL];
        bool keptv[DL_UNROLL];
        unsigned long long kmv[DL_UNROLL], nmv[DL_UNROLL];
        bool big = false;
        int sc[DL_UNROLL];
#pragma unroll
        for (int u = 0; u < DL_UNROLL; ++u) {
            const uint32_t op = cgv[u] & 0xfu;
            lenv[u] = cgv[u] >> 4;
            keptv[u] = ((0x18du >> op) & 1u) && lenv[u] != 0u;
            kindv[u] = ((0x181u >> op) & 1u) ? 0u : 1u;
            kmv[u] = __builtin_amdgcn_ballot_w64(keptv[u]);
            nmv[u] = __builtin_amdgcn_ballot_w64(keptv[u] && kindv[u] == 1u);
            big = big || (keptv[u] && lenv[u] > (1u << 22));
            klv[u] = keptv[u] ? lenv[u] : 0u;
            sc[u] = (int)klv[u];
        }
        if (__builtin_amdgcn_ballot_w64(big) != 0ull) { overflow = true; return 0u; }
        static_assert(DL_UNROLL == 4, "one scan4 group");
        scan4(sc[0], sc[1], sc[2], sc[3]);                             // each <= 64 * 2^22
#pragma unroll
        for (int u = 0; u < DL_UNROLL; ++u) {
        const unsigned long long km = kmv[u];
        if (km == 0ull) continue;
        if (cur >= POS_CAP - (1u << 28)) { overflow = true; return 0u; }
        const bool kept = keptv[u];
        const uint32_t kind = kindv[u], kl = klv[u];
        const unsigned long long nm = nmv[u];
        const unsigned long long pm = km & below;
        uint32_t pk = open_kind;
        if (pm != 0ull) pk = (uint32_t)((nm >> (63 - __clzll((long long)pm))) & 1ull);
        const bool head = kept && kind != pk;
        const unsigned long long hm = __builtin_amdgcn_ballot_w64(head);
        const uint32_t S = (uint32_t)sc[u];
        const uint32_t E = S - kl;
        const uint32_t gtot = (uint32_t)__builtin_amdgcn_readlane((int)S, 63);
        if (hm == 0ull) {                                              // the open run goes on
            open_len += gtot;
            cur += gtot;
            continue;
        }
        const bool open_exists = open_kind != 2u;
        const unsigned long long hb = hm & below;
        const uint32_t q = (uint32_t)__popcll(hb);
        const int ph = hb != 0ull ? 63 - __clzll((long long)hb) : 0;
        const uint32_t Eph = (uint32_t)__shfl((int)E, ph, 64);         // where the run this head closes began
        const uint32_t tot = q == 0u ? open_len + E : E - Eph;        // (32 bits: a run stays below the position cap checked above)
        const uint32_t start = q == 0u ? open_start : cur + Eph;
        const bool closes = head && (q != 0u || open_exists);
        // an M head closes a D/N run: a deletion; its number = the deletions closed by the heads before this lane
        const unsigned long long dm = __builtin_amdgcn_ballot_w64(closes && kind == 0u);
        if (closes && kind == 0u) out[w + (uint32_t)__popcll(dm & below)] = make_uint2(start, tot);
        w += (uint32_t)__popcll(dm);
        const int lh = __builtin_amdgcn_readfirstlane(63 - __clzll((long long)hm));
        const uint32_t Elh = (uint32_t)__builtin_amdgcn_readlane((int)E, lh);
        open_kind = (uint32_t)__builtin_amdgcn_readlane((int)kind, lh);
        open_len = gtot - Elh;
        open_start = cur + Elh;
        cur += gtot;
        }
    }
    endp = open_kind == 0u ? cur : open_start;                         // (no run, or only a D/N run: open_start == pos)
    return w;
}

constexpr uint32_t WAVE_DELS_MIN = 24;     // reads with more ops than this are walked by the whole wave

// ---- the walk without any run logic ---------------------------------------------------------------------------------
//     depth(read, x) = [pos <= x < end] - [x inside one of the read's D/N ops]
// holds for ANY way of cutting the D/N bases into pieces, and for end = the position after the LAST reference-consuming
// op whatever its kind: a trailing deletion lies inside [pos, end) and cancels itself, neighbouring D/N ops are two
// pieces that share an edge.  So every D/N op of length >= 1 is its own list entry at its reference position -- one
// prefix sum of the consumed lengths per 64 ops and one ballot, nothing carried from group to group but the position and
// the count.  (The run-merging walk above spends ~1 us per 64 ops on heads, closing runs and their bookkeeping, most
// of it on the scalar unit the four SIMDs of a CU share: 10.4 ms for a 20x genome.)  What merging bought is the bound
// "at most half the ops are deletions" behind the dense list offset (op offset >> 1) + r; a read with more D/N ops
// than its slots hold (cap) -- D D D ..., no aligner's output -- is walked again by the merging walk, which fits.
// ix / ixcap: the read's tile index slots (entries 1 .. K - 2 are written here, the caller adds the first and the last)
__device__ __forceinline__ uint32_t serial_dels_plain(const uint32_t* __restrict__ ops, uint32_t n, uint32_t pos,
                                                      uint2* __restrict__ out, uint32_t cap, uint32_t& endp,
                                                      uint32_t* __restrict__ ix, uint32_t ixcap)
{
    uint32_t x = pos, w = 0;
    const uint32_t pk = pos >> PT_SHIFT;
    for (uint32_t k = 0; k < n; ++k) {
        const uint32_t cg = ops[k], op = cg & 0xfu, len = cg >> 4;
        if (!((0x18du >> op) & 1u) || len == 0u) continue;
        if (!((0x181u >> op) & 1u)) {                                  // D / N
            if (w < cap) out[w] = make_uint2(x, len);
            ++w;
        }
        const uint32_t y = sat_pos(x + len);
        // boundaries in (x, y]: every deletion so far (this op too, if it is one) starts before them
        for (uint32_t kk = (x >> PT_SHIFT) + 1u; kk <= (y >> PT_SHIFT) && kk - pk < ixcap; ++kk) ix[kk - pk] = w;
        x = y;
    }
    endp = x;
    return w;                                                          // > cap: the caller walks the read again
}

// Round 6: FOUR CONSECUTIVE ops per lane.  The walk above it replaced took a batch of 256 ops as four groups of 64 -- one op per
// lane and group: four wave prefix sums, four ballots, four rank computations and four boundary tests per batch, ~175
// instructions -- and the pass is bound by exactly that (18.7 M batches per genome x 175 instructions is 5.4 of its 7.0 ms at
// one vector instruction per four cycles and SIMD; halving its list bytes changed nothing, HISTORY.md).  Here a lane loads its
// four ops with ONE 16-byte load through a buffer descriptor over the read's ops (past the end: zeros = ops that consume
// nothing), adds them up locally, and the wave needs two prefix sums per batch: of the consumed lengths and of the D/N counts.
__device__ __forceinline__ uint32_t wave_dels_plain(const uint32_t* __restrict__ ops, uint32_t n, int lane, uint32_t pos,
                                                    uint2* __restrict__ out, uint32_t cap, uint32_t& endp, bool& again,
                                                    uint32_t* __restrict__ ix, uint32_t ixcap)
{
    typedef unsigned int v4u_t __attribute__((ext_vector_type(4)));
    const uint32_t pk = pos >> PT_SHIFT;
    uint32_t cur = pos, w = 0u;                                         // wave uniform
    again = false;
    const rsrc_t rs = make_rsrc(ops, n * 4u);
    // THREE batches in flight.  With one (what every form of this walk had) a wave holds 1 KB of loads in the air, the
    // device's 8 192 resident waves 8 MB -- and 8 MB per ~2 us of loaded-memory latency IS the 4.2 TB/s the pass ran at,
    // whatever its instruction count or the bytes it wrote (both were halved in turn this round and changed nothing).  Three
    // slots, the loop unrolled by three so that no register with a load outstanding is ever copied.
    auto load = [&](uint32_t b) -> v4u_t { return __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(b * 4u) + lane * 16, 0, 0); };
    bool stop = false;
    auto batch = [&](v4u_t& slot, uint32_t b0) {
        const v4u_t cg = slot;
        if (b0 + 768u < n) slot = load(b0 + 768u);
        const uint32_t c4[4] = {cg.x, cg.y, cg.z, cg.w};
        uint32_t len[4], cons[4];
        bool dn[4], big = false;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint32_t op = c4[e] & 0xfu;
            len[e] = c4[e] >> 4;
            const bool c = ((0x18du >> op) & 1u) && len[e] != 0u;
            dn[e] = c && !((0x181u >> op) & 1u);
            big = big || (c && len[e] > (1u << 22));
            cons[e] = c ? len[e] : 0u;
        }
        if (__builtin_amdgcn_ballot_w64(big) != 0ull || cur >= POS_CAP - (1u << 30)) { again = true; stop = true; return; }
        const uint32_t p1 = cons[0], p2 = p1 + cons[1], p3 = p2 + cons[2], tot = p3 + cons[3];   // <= 4 * 2^22
        const uint32_t nd4 = (uint32_t)dn[0] + (uint32_t)dn[1] + (uint32_t)dn[2] + (uint32_t)dn[3];
        const uint32_t incl = (uint32_t)wave_inclusive_scan((int)tot);     // the batch advances by < 2^30
        const uint32_t cincl = (uint32_t)wave_inclusive_scan((int)nd4);
        const uint32_t gtot = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        const uint32_t cnt = (uint32_t)__builtin_amdgcn_readlane((int)cincl, 63);
        if (w + cnt > cap) { again = true; stop = true; return; }
        const uint32_t s0 = cur + incl - tot, s1 = s0 + p1, s2 = s0 + p2, s3 = s0 + p3;
        uint32_t k = w + cincl - nd4;
        if (dn[0]) out[k++] = make_uint2(s0, len[0]);
        if (dn[1]) out[k++] = make_uint2(s1, len[1]);
        if (dn[2]) out[k++] = make_uint2(s2, len[2]);
        if (dn[3]) out[k] = make_uint2(s3, len[3]);
        const uint32_t nxt_cur = cur + gtot;
        if ((nxt_cur >> PT_SHIFT) != (cur >> PT_SHIFT)) {               // (wave uniform) the tile index:
            // boundaries in (cur, nxt_cur]: the deletions before this batch, and those of it that start before the boundary
            for (uint32_t kk = (cur >> PT_SHIFT) + 1u; kk <= (nxt_cur >> PT_SHIFT) && kk - pk < ixcap; ++kk) {
                const uint32_t bnd = kk << PT_SHIFT;
                const uint32_t mine = (uint32_t)(dn[0] && s0 < bnd) + (uint32_t)(dn[1] && s1 < bnd) + (uint32_t)(dn[2] && s2 < bnd) +
                                      (uint32_t)(dn[3] && s3 < bnd);
                const uint32_t before = (uint32_t)wave_total((int)mine);
                if (lane == 0) ix[kk - pk] = w + before;
            }
        }
        w += cnt;
        cur = nxt_cur;
    };
    v4u_t A = load(0u), B = {0u, 0u, 0u, 0u}, C = {0u, 0u, 0u, 0u};
    if (n > 256u) B = load(256u);
    if (n > 512u) C = load(512u);
    for (uint32_t b0 = 0; b0 < n && !stop; b0 += 768u) {
        batch(A, b0);
        if (stop || b0 + 256u >= n) break;
        batch(B, b0 + 256u);
        if (stop || b0 + 512u >= n) break;
        batch(C, b0 + 512u);
    }
    if (stop) return 0u;
    endp = cur;
    return w;
}

__global__ __launch_bounds__(256) void gd_dels_raw_kernel(DelBatch B)
{
    const int lane = threadIdx.x & 63;
    const uint32_t gunit = blockIdx.x * 4u + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (gunit >= B.n_units) return;
    const uint32_t ji = (uint32_t)__builtin_amdgcn_readfirstlane((int)norm::batch_find(B.ubeg, B.n_jobs, gunit));
    const DelJob job = B.jobs[ji];
    const uint32_t unit = gunit - B.ubeg[ji];
    const uint32_t r = unit * 64u + (uint32_t)lane;
    const bool valid = r < job.n_reads;
    uint32_t p = 0, o0 = 0, n = 0, fq = 0;
    if (valid) {
        p = (uint32_t)job.pos[r];
        o0 = job.off[r];
        n = job.off[r + 1] - o0;
        fq = ((uint32_t)job.flag[r] << 8) | (uint32_t)job.mapq[r];
    }
    const uint32_t doff = (o0 >> 1) + r;                  // the read's slots: up to the next read's (o1 >> 1) + r + 1
    const uint32_t cap = ((o0 + n) >> 1) - (o0 >> 1) + 1u;
    const uint32_t pb = pt_slot(o0, r), pcap = pt_cap(o0, n);   // ... and its tile index slots
    uint32_t endp = p, nd = 0;
    bool serial = n <= WAVE_DELS_MIN;
    bool merged = false;                                  // the list holds MERGED deletions (the fallback walks): no index
    unsigned long long todo = __builtin_amdgcn_ballot_w64(!serial);
    while (todo != 0ull) {                                // long reads: the wave walks one at a time
        const int j = __ffsll((long long)todo) - 1;
        todo &= todo - 1ull;
        const uint32_t pj = (uint32_t)__builtin_amdgcn_readlane((int)p, j);
        const uint32_t oj = (uint32_t)__builtin_amdgcn_readlane((int)o0, j);
        const uint32_t nj = (uint32_t)__builtin_amdgcn_readlane((int)n, j);
        const uint32_t dj = (uint32_t)__builtin_amdgcn_readlane((int)doff, j);
        const uint32_t cj = (uint32_t)__builtin_amdgcn_readlane((int)cap, j);
        const uint32_t bj = (uint32_t)__builtin_amdgcn_readlane((int)pb, j);
        const uint32_t kj = (uint32_t)__builtin_amdgcn_readlane((int)pcap, j);
        uint32_t ej = pj;
        bool ovf, again;
        uint32_t wj = wave_dels_plain(job.cigar + oj, nj, lane, pj, job.dl + dj, cj, ej, again, job.pck + bj, kj);
        ovf = false;
        if (again) wj = wave_dels(job.cigar + oj, nj, lane, pj, job.dl + dj, ej, ovf);   // merged runs always fit
        if (lane == j) { nd = wj; endp = ej; serial = ovf; merged = again; }
    }
    if (serial && n != 0u) {
        nd = serial_dels_plain(job.cigar + o0, n, p, job.dl + doff, cap, endp, job.pck + pb, pcap);
        merged = nd > cap;
        if (merged) nd = serial_dels(job.cigar + o0, n, p, job.dl + doff, endp);
    }
    if (valid) {
        // the index's first entry (nothing starts before the boundary at or before pos) and its last (everything starts
        // before the one behind the end); the walk wrote the ones between
        uint32_t iw = PT_NONE;
        const uint32_t K = pt_entries(p, endp, nd);
        if (K != 0u) {
            iw = PT_SEARCH;
            if (!merged && K <= pcap) {
                job.pck[pb] = 0u;
                job.pck[pb + K - 1u] = nd;
                iw = pb;
            }
        }
        job.lrec[r] = make_uint4(p, endp, doff, iw);
        job.lfq[r] = fq;
        job.ndel[r] = nd;
    }
    const uint32_t smax = wave_max_u32(endp - p);
    if (lane == 0 && smax != 0u) atomicMax(job.max_span, (int32_t)smax);
    const uint32_t dsum = (uint32_t)wave_total((int)nd);
    if (lane == 0 && dsum != 0u) atomicAdd(job.del_total, dsum);
}

// Every contig's largest span next to its deletion total ([unused x n][deletions x n][spans x n]): ONE read-back for the
// batch instead of one 4-byte copy per contig.
__global__ __launch_bounds__(256) void gd_ptile_totals_kernel(DelBatch B)
{
    const uint32_t ji = blockIdx.x * 256u + threadIdx.x;
    if (ji >= B.n_jobs) return;
    const DelJob job = B.jobs[ji];
    *job.total = 0u;
    job.total[2u * B.n_jobs] = (uint32_t)*job.max_span;
}

// deletions of a sorted list that start before `bound` (a read without a tile index: PT_SEARCH)
__device__ __forceinline__ uint32_t pt_count_before(const uint2* __restrict__ d, uint32_t n, uint32_t bound)
{
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (d[mid].x < bound) lo = mid + 1u; else hi = mid;
    }
    return lo;
}

// ---------------------------------------------------------------------------
// LT2: the long-read tile kernel.
// ---------------------------------------------------------------------------
constexpr int LQ_CAP = 128;                   // queue items per wave (every lane pushes at most one per round)

// One queue item: {index of the piece's first deletion in the contig's list, deletions in it (1..64)}
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
    static_assert(CHUNK % 256 == 0, "wave chunk must be whole rows");
    static_assert(T % (1 << PT_SHIFT) == 0 && WAVE <= LQ_CAP, "tiles start on index boundaries; a round fits the queue");

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
    const uint32_t* const pck = ti.pck;

    {
        const int4 z = make_int4(0, 0, 0, 0);
        int4* d4 = reinterpret_cast<int4*>(s_diffp);
#pragma unroll
        for (int i = 0; i < T / 4 / NT; ++i) d4[tid + i * NT] = z;       // (T / 4 is a multiple of NT)
        if (tid == 0) d4[T / 4] = z;
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

    // ---- phase A: candidate reads -> their deletions in this tile -> LDS +1/-1 -------------
    // Candidate i of a batch of NT belongs to wave i % NW: the overlapping reads (mostly
    // the latest starters) spread evenly over the waves.
    for (uint32_t base = 0; base < nrd; base += (uint32_t)NT) {
        const uint32_t idx = base + (uint32_t)(lane * NW + wv);
        const bool inb = idx < nrd;
        // one round trip per candidate: its record and its filter word
        uint4 rc = make_uint4(0x7fffffffu, 0u, 0u, PT_NONE);
        uint32_t fq = 0;
        if (inb) { rc = grec[idx]; fq = gfq[idx]; }
        const int32_t p = (int32_t)rc.x;
        const int32_t e = inb ? (int32_t)rc.y : -1;
        // reaches t0-1 or beyond, and passes the read filter of `samtools depth`
        const bool hit = e >= t0 && p < tend && ((fq >> 8) & job.flag_mask) == 0u && (int)(fq & 0xffu) >= job.Q;
        uint32_t cur = 0, rem = 0;                            // this lane's deletions still to queue
        if (hit) {
            // the read's own +1 / -1 (its D/N ops subtract below)
            const int rs = p - t0;
            atomicAdd(&s_diff[rs > -1 ? rs : -1], 1);
            if (e < tend) atomicAdd(&s_diff[e - t0], -1);
            if (rc.w != PT_NONE) {
                // deletions that can touch the tile: the last one starting before t0 (it may reach in) up to the
                // last one starting before the tile's end -- two entries of the read's tile index
                const uint32_t pk = rc.x >> PT_SHIFT;
                const uint32_t K = (rc.y >> PT_SHIFT) - pk + 2u;
                const uint32_t k0 = (uint32_t)t0 > rc.x ? ((uint32_t)t0 >> PT_SHIFT) - pk : 0u;   // t0 <= end: k0 <= K - 2
                uint32_t k1 = k0 + (uint32_t)(T >> PT_SHIFT);
                k1 = k1 < K - 1u ? k1 : K - 1u;
                uint32_t a, b;
                if (rc.w != PT_SEARCH) { a = pck[rc.w + k0]; b = pck[rc.w + k1]; }
                else {                                         // no index (a read whose slots were too few): its list is bisected
                    const uint32_t nd = ti.ndel[ti.lo + idx];
                    a = k0 != 0u ? pt_count_before(dl + rc.z, nd, (pk + k0) << PT_SHIFT) : 0u;
                    b = k1 == K - 1u ? nd : pt_count_before(dl + rc.z, nd, (pk + k1) << PT_SHIFT);
                }
                const uint32_t j0 = a != 0u ? a - 1u : 0u;
                cur = rc.z + j0;
                rem = b - j0;
            }
        }
        // queue them in pieces of 64, one piece per lane and round
        for (;;) {
            const unsigned long long pm = __builtin_amdgcn_ballot_w64(rem != 0u);
            if (pm == 0ull) break;
            const uint32_t np = (uint32_t)__popcll(pm);
            if (qn + np > (uint32_t)LQ_CAP) { drain(qn); qn = 0; }
            if (rem != 0u) {
                const uint32_t rk = qn + __builtin_amdgcn_mbcnt_hi((uint32_t)(pm >> 32),
                                             __builtin_amdgcn_mbcnt_lo((uint32_t)pm, 0u));
                const uint32_t c = rem < DL_CHUNK ? rem : DL_CHUNK;
                Q[rk] = make_uint2(cur, c);
                cur += c; rem -= c;
            }
            qn += np;
        }
    }
    if (qn != 0u) drain(qn);
    __syncthreads();

    // ---- phase B: scan the wave's quarter, publish its total, then carry / store / reduce ----
    const int chunk0 = wv * CHUNK;
    {
        PhaseB B;
        B.s_diff = s_diff; B.s_bmap = s_bmap; B.s_clo = s_clo; B.s_chi = s_chi; B.s_hasb = &s_hasb;
        B.out = job.perbase + ti.base_off + t0;
        B.wsum = job.win_sum + ti.win_off;
        B.wmin = job.win_min + ti.win_off;
        B.t0 = t0; B.tlen = tlen; B.chunk0 = chunk0; B.lane = lane;
        B.W = job.W; B.mincov = job.mincov; B.maxmean = job.maxmean; B.step = job.step;
        // a read covers a position at most once: depth <= candidate reads
        const bool wide = nrd >= (1u << 22);
        if constexpr (ROWS == 4) {
            if (tlen == T && !wide) {                     // workgroup uniform: the single-pass form
                PhaseBRows R;
                const int tot = phase_b_scan<ROWS>(B, R);
                if (lane == 0) s_wtot[wv] = tot;
                __syncthreads();
                int carry = s_diff[-1];                    // depth at t0-1
#pragma unroll
                for (int v = 0; v < NW - 1; ++v) carry += v < wv ? s_wtot[v] : 0;
                B.carry = carry;
                phase_b_finish<ROWS, OPT>(B, R, job.w_magic, job.w_shift, job.s_magic, job.s_shift);
                __syncthreads();
                phase_c<T, NT>(job, tile, t0, ti.ctg, tid, lane, wv, s_bmap, s_clo, s_chi, s_wcnt, &s_hasb, &s_base);
                return;
            }
        }
        // clipped / very deep tiles and other tile shapes: totals pass, then row by row
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
        int carry = s_diff[-1];                            // depth at t0-1
#pragma unroll
        for (int v = 0; v < NW - 1; ++v) carry += v < wv ? s_wtot[v] : 0;
        B.carry = carry;
        if (tlen == T && !wide) phase_b_rows<ROWS, true, false, OPT>(B);
        else                    phase_b_rows<ROWS, false, true, OPT>(B);
    }
    __syncthreads();

    phase_c<T, NT>(job, tile, t0, ti.ctg, tid, lane, wv, s_bmap, s_clo, s_chi, s_wcnt, &s_hasb, &s_base);
}

}  // namespace gd
