// gd_tile_v8.hpp -- K1, the tile kernel (short-read path), generation 8: packed read descriptors.
//
// Same algorithm, LDS difference array, phases B and C and results as gd_tile_v7.hpp -- it
// replaces, like it, the per-read CIGAR walk of `samtools depth`
// (/root/reference/depth/depth.go:45) and the per-line window / class reductions of the callback
// (depth/depth.go:293-323).  What changed is what phase A reads.
//
// v7 moved its real traffic at ~88 % of copy bandwidth, so the bytes themselves were what was
// left: per read it fetched pos (4) + flag (2) + MAPQ (1) + two CSR offsets (4) + the first CIGAR
// op (4) from five arrays, although a short read is almost always ONE counted op.  v8 reads one
// 8-byte descriptor per read, built once when the records arrive (gd_pack_*_kernel below, run by
// gd_adopt_device / gd_ingest_finish, or by the first gd_compute after gd_commit):
//
//   word 0   bit 31      0: `simple` (at most one counted op)   1: `complex`
//            bits 0-30   pos
//   word 1   bits 0-7    MAPQ
//            bits 8-19   FLAG (SAMv1 defines 12 bits; a contig with a higher bit set, or a
//                        negative pos, is not packed and takes the v7 kernel)
//            bits 20-31  simple:  length of the M/=/X op, 0..4095 (0: nothing to mark: no CIGAR, or a
//                                 zero-length op)
//                        complex: (n_ops - 1) | delta << 5 -- the read's ops sit in a compact side
//                                 array at cx_base[read >> 6] + delta (n_ops <= 32, delta <= 126);
//                                 delta == 127: `far` -- fetched from the original CSR arrays.
//
// Complex reads (indels, clips: a few percent of short-read data) therefore cost one more 4-byte
// load per 64 reads (cx_base, issued together with the descriptor) and their ops, which lie
// contiguously in the side array in read order: the first four ops of every complex read of a
// batch are fetched with ONE 16-byte load per lane as soon as the descriptors are in (all lanes in
// parallel, one extra hop), and travel through the per-wave queue to the dense-lane walk.  Nothing
// is staged in LDS any more.
#pragma once

namespace gd {
namespace v8 {

constexpr uint32_t CX_FAR = 127u;          // delta value of a far complex read
constexpr uint32_t CX_MAX_OPS = 32u;       // ops of a near complex read
constexpr uint32_t SIMPLE_MAX = 4095u;     // longest inlined op

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// ---- packing (ingest side) -----------------------------------------------------------------
struct PackJob {
    const int32_t*  pos;
    const uint16_t* flag;
    const uint8_t*  mapq;
    const uint32_t* off;
    const uint32_t* cigar;
    uint32_t  n_reads;
    uint32_t  n_units;        // ceil(n_reads / 64)
    uint2*    desc;           // n_reads descriptors
    uint32_t* cx_base;        // n_units + 1: per-unit op totals, then (after the scan) exclusive offsets
    uint32_t* cx_cigar;       // compact ops of the near complex reads (third kernel)
    uint32_t* status;         // bit 0: a FLAG above 0xfff, bit 1: a negative pos
};

// P1: one wave per unit of 64 consecutive reads: classify, write descriptors, count the unit's ops.
__global__ __launch_bounds__(256) void gd_pack_desc_kernel(PackJob j)
{
    const int lane = threadIdx.x & 63;
    const uint32_t unit = blockIdx.x * 4u + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (unit >= j.n_units) return;
    const uint32_t r = unit * 64u + (uint32_t)lane;
    const bool valid = r < j.n_reads;
    int32_t p = 0;
    uint32_t f = 0, mq = 0, n = 0, cg = 0;
    if (valid) {
        p = j.pos[r]; f = j.flag[r]; mq = j.mapq[r];
        const uint32_t o0 = j.off[r];
        n = j.off[r + 1] - o0;
        if (n) cg = j.cigar[o0];
    }
    const uint32_t op = cg & 0xfu, len = cg >> 4;
    const bool counted = (0x181u >> op) & 1u;                 // M = X
    const bool inl = n == 0u || (n == 1u && counted && len <= SIMPLE_MAX);
    const bool cand = valid && !inl && n <= CX_MAX_OPS;
    const uint32_t mine = cand ? n : 0u;
    const uint32_t ex = (uint32_t)wave_inclusive_scan((int)mine) - mine;
    const bool near = cand && ex < CX_FAR;
    uint32_t payload;
    if (inl) payload = n ? len : 0u;
    else if (near) payload = (n - 1u) | (ex << 5);
    else payload = CX_FAR << 5;
    if (valid) {
        uint32_t bad = 0;
        if (f >> 12) bad |= 1u;
        if (p < 0) bad |= 2u;
        if (bad) atomicOr(j.status, bad);
        j.desc[r] = make_uint2((uint32_t)p | (inl ? 0u : 0x80000000u), mq | ((f & 0xfffu) << 8) | (payload << 20));
    }
    const uint32_t tot = (uint32_t)wave_total((int)(near ? n : 0u));
    if (lane == 0) j.cx_base[unit] = tot;
}

// P2: exclusive scan of the unit totals, in place; cx_base[n_units] = grand total.  One workgroup.
__global__ __launch_bounds__(1024) void gd_pack_scan_kernel(uint32_t* __restrict__ v, uint32_t n)
{
    __shared__ uint32_t s_part[1024];
    const uint32_t tid = threadIdx.x;
    const uint32_t per = (n + 1023u) / 1024u;
    const uint32_t b = tid * per, e = b + per < n ? b + per : n;
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

// P3: the ops of the near complex reads into the side array.
__global__ __launch_bounds__(256) void gd_pack_ops_kernel(PackJob j)
{
    const int lane = threadIdx.x & 63;
    const uint32_t unit = blockIdx.x * 4u + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (unit >= j.n_units) return;
    const uint32_t r = unit * 64u + (uint32_t)lane;
    if (r >= j.n_reads) return;
    const uint2 d = j.desc[r];
    const uint32_t payload = d.y >> 20, delta = payload >> 5;
    if (!(d.x >> 31) || delta == CX_FAR) return;
    const uint32_t n = (payload & 31u) + 1u;
    const uint32_t src = j.off[r], dst = j.cx_base[unit] + delta;
    for (uint32_t k = 0; k < n; ++k) j.cx_cigar[dst + k] = j.cigar[src + k];
}

// ---- phase A -------------------------------------------------------------------------------
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr int QW = 7;                      // queue words per entry: ps4 | offset | n | four prefetched ops

// The ops of one queued read: the first four from registers, the rest from memory.
struct OpSrc {
    uint32_t r0, r1, r2, r3;
    const uint32_t* g;
    __device__ __forceinline__ uint32_t operator[](uint32_t k) const
    {
        return k == 0u ? r0 : k == 1u ? r1 : k == 2u ? r2 : k == 3u ? r3 : g[k];
    }
};

struct PhaseA8 {
    const uint2*    desc;        // the contig's descriptors
    const uint32_t* cxb;         // the contig's cx_base
    const uint32_t* cxc;         // the contig's compact ops
    const uint32_t* off;         // original CSR arrays of the contig (far reads)
    const uint32_t* gcig;
    uint32_t lo, nrd, n_units, n_cx;   // n_cx: ops in cxc
    int32_t* s_diff;
    uint32_t* wq;                // this wave's queue, QW rows of 64: ps4 | op offset or read index | n (0: far) | ops
    int neg4t0, T4;
    uint32_t fmask;              // (flag_mask & 0xfff) << 8
    int Q, tid, lane;
};

// Phase A for one wave, U = 4 reads per lane and batch of NT*4 (see gd_tile_v7.hpp).  d[]/cb[] hold
// the first batch on entry.  SUMS: intervals go to a SumSink instead of +1/-1 marks.
template <int NT, bool SUMS>
__device__ __forceinline__ uint32_t phase_a(const PhaseA8& A, u32x2 (&d)[4], uint32_t (&cb)[4],
                                            const v7::SumSink* sink = nullptr)
{
    constexpr int U = 4;
    const int tid = A.tid, lane = A.lane;
    uint32_t smax = 0;
    uint32_t qn = 0;                              // entries queued (wave uniform)
    uint32_t* const wq = A.wq;
    const uint32_t wave0 = (uint32_t)__builtin_amdgcn_readfirstlane(tid - lane);
    const rsrc_t r_cxb = make_rsrc(A.cxb, (A.n_units + 1u) * 4u);
    const rsrc_t r_cxc = make_rsrc(A.cxc, (A.n_cx + 4u) * 4u);   // the array is padded by four entries

    auto drain = [&](uint32_t cnt) {
        __builtin_amdgcn_wave_barrier();
        if ((uint32_t)lane < cnt) {
            const int qp = (int)wq[lane];
            const uint32_t qa = wq[WAVE + lane];
            const uint32_t qk = wq[2 * WAVE + lane];
            uint32_t span;
            if (qk != 0u) {
                OpSrc ops;
                ops.r0 = wq[3 * WAVE + lane]; ops.r1 = wq[4 * WAVE + lane];
                ops.r2 = wq[5 * WAVE + lane]; ops.r3 = wq[6 * WAVE + lane];
                ops.g = A.cxc + qa;
                if constexpr (SUMS) span = v7::walk_cigar_sums(ops, qk, qp, *sink);
                else                span = walk_cigar4(ops, qk, qp, A.T4, A.s_diff);
            } else {                              // far: the original CSR arrays
                const uint32_t o0 = A.off[qa];
                const uint32_t* ops = A.gcig + o0;
                const uint32_t k = A.off[qa + 1] - o0;
                if constexpr (SUMS) span = v7::walk_cigar_sums(ops, k, qp, *sink);
                else                span = walk_cigar4(ops, k, qp, A.T4, A.s_diff);
            }
            smax = span > smax ? span : smax;
        }
        __builtin_amdgcn_wave_barrier();
    };

    for (uint32_t base = 0; base < A.nrd; base += NT * U) {
        if (base != 0) {                          // further batches (deep tiles)
            const rsrc_t r_desc = make_rsrc(A.desc + A.lo + base, (A.nrd - base) * 8u);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                d[u] = __builtin_amdgcn_raw_buffer_load_b64(r_desc, (tid + u * NT) * 8, 0, 0);
                cb[u] = __builtin_amdgcn_raw_buffer_load_b32(r_cxb, (int)(((A.lo + base + (uint32_t)(tid + u * NT)) >> 6) << 2), 0, 0);
            }
        }
        int ps4[U];
        bool cx[U], keep[U];
        uint32_t pl[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t w = d[u].y;
            keep[u] = ((w & A.fmask) == 0u) & ((int)(w & 0xffu) >= A.Q);
            pl[u] = w >> 20;
            cx[u] = keep[u] & ((d[u].x >> 31) != 0u);             // lanes past the range hold 0: neither
            ps4[u] = (int)((d[u].x << 2) + (uint32_t)A.neg4t0);   // the shift drops bit 31
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (base + (uint32_t)(u * NT) + wave0 >= A.nrd) continue;   // wave uniform: slot past the range
            const bool simple = keep[u] & !cx[u] & ((d[u].x >> 31) == 0u) & (pl[u] != 0u);
            const uint32_t rs = simple ? pl[u] : 0u;
            smax = rs > smax ? rs : smax;
            const int e4 = ps4[u] + (int)(pl[u] << 2);
            if constexpr (SUMS) {
                if (simple & (e4 > 0)) v7::add_interval(*sink, ps4[u] >> 2, e4 >> 2);
            } else if (simple & (e4 >= 0)) {                      // reaches t0-1 or beyond
                const int cs4 = ps4[u] > -4 ? ps4[u] : -4;
                atomicAdd(lds_at(A.s_diff, cs4), 1);
                if (e4 < A.T4) atomicAdd(lds_at(A.s_diff, e4), -1);
            }
        }
        // ---- complex reads: queue for the dense-lane walk ------------------
        unsigned long long m[U];
        uint32_t cnt[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { m[u] = __builtin_amdgcn_ballot_w64(cx[u]); cnt[u] = (uint32_t)__popcll(m[u]); }
        const uint32_t tot = cnt[0] + cnt[1] + cnt[2] + cnt[3];
        if (tot != 0u) {                          // wave uniform
            if (qn + tot > (uint32_t)WAVE) { drain(qn); qn = 0; }
            if (tot <= (uint32_t)WAVE) {
                // first four ops of every near complex read, all slots in flight together; every
                // other lane points past the buffer (no access)
                u32x4 o4[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const bool near = cx[u] & ((pl[u] >> 5) != CX_FAR);
                    o4[u] = __builtin_amdgcn_raw_buffer_load_b128(r_cxc, near ? (int)((cb[u] + (pl[u] >> 5)) << 2) : -16, 0, 0);
                }
                uint32_t b = qn;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (cx[u]) {
                        const uint32_t rk = b + __builtin_amdgcn_mbcnt_hi((uint32_t)(m[u] >> 32),
                                                    __builtin_amdgcn_mbcnt_lo((uint32_t)m[u], 0u));
                        const uint32_t delta = pl[u] >> 5;
                        const bool far = delta == CX_FAR;
                        wq[rk] = (uint32_t)ps4[u];
                        wq[WAVE + rk] = far ? A.lo + base + (uint32_t)(tid + u * NT) : cb[u] + delta;
                        wq[2 * WAVE + rk] = far ? 0u : (pl[u] & 31u) + 1u;
                        wq[3 * WAVE + rk] = o4[u].x; wq[4 * WAVE + rk] = o4[u].y;
                        wq[5 * WAVE + rk] = o4[u].z; wq[6 * WAVE + rk] = o4[u].w;
                    }
                    b += cnt[u];
                }
                qn += tot;
            } else {
                // more complex reads in one batch than the queue holds (not short-read shaped
                // data): slot by slot through the queue, fields fetched again (cache hits) so
                // that no register array is indexed at run time
                const rsrc_t q_desc = make_rsrc(A.desc + A.lo + base, (A.nrd - base) * 8u);
#pragma unroll 1
                for (int u = 0; u < U; ++u) {
                    const int vo = tid + u * NT;
                    const uint32_t ridx = A.lo + base + (uint32_t)vo;
                    const u32x2 dd = __builtin_amdgcn_raw_buffer_load_b64(q_desc, vo * 8, 0, 0);
                    const uint32_t cc = __builtin_amdgcn_raw_buffer_load_b32(r_cxb, (int)((ridx >> 6) << 2), 0, 0);
                    const uint32_t w = dd.y, pp = w >> 20;
                    const bool c = ((w & A.fmask) == 0u) & ((int)(w & 0xffu) >= A.Q) & ((dd.x >> 31) != 0u);
                    const bool far = (pp >> 5) == CX_FAR;
                    const u32x4 oo = __builtin_amdgcn_raw_buffer_load_b128(r_cxc, (c & !far) ? (int)((cc + (pp >> 5)) << 2) : -16, 0, 0);
                    const unsigned long long mu = __builtin_amdgcn_ballot_w64(c);
                    if (c) {
                        const uint32_t rk = __builtin_amdgcn_mbcnt_hi((uint32_t)(mu >> 32),
                                                __builtin_amdgcn_mbcnt_lo((uint32_t)mu, 0u));
                        wq[rk] = (dd.x << 2) + (uint32_t)A.neg4t0;
                        wq[WAVE + rk] = far ? ridx : cc + (pp >> 5);
                        wq[2 * WAVE + rk] = far ? 0u : (pp & 31u) + 1u;
                        wq[3 * WAVE + rk] = oo.x; wq[4 * WAVE + rk] = oo.y; wq[5 * WAVE + rk] = oo.z; wq[6 * WAVE + rk] = oo.w;
                    }
                    drain((uint32_t)__popcll(mu));
                }
            }
        }
    }
    if (qn != 0) drain(qn);
    return smax;
}

// OPT: per-base stores 0 plain, 1 non-temporal, 2 none (gd_set_outputs without GD_OUT_PERBASE).
template <int T, int NT, int OPT>
__global__ __launch_bounds__(NT) void gd_tile_kernel(Job job)
{
    constexpr int NW = NT / WAVE;          // waves per workgroup
    constexpr int CHUNK = T / NW;          // positions per wave
    constexpr int ROWS = CHUNK / 256;      // rows of 256 positions per wave
    constexpr int NWORDS = T / 32;         // bitmap words
    constexpr int U = 4;                   // reads per lane in flight
    static_assert(CHUNK % 256 == 0, "wave chunk must be whole rows");

    __shared__ __attribute__((aligned(16))) int32_t s_diffp[T + 4];  // [3] = index -1
    __shared__ uint32_t s_bmap[NWORDS];    // boundary bit per position
    __shared__ uint32_t s_clo[NWORDS];     // class bit 0 at boundary positions
    __shared__ uint32_t s_chi[NWORDS];     // class bit 1 at boundary positions
    __shared__ uint32_t s_wq[NW * QW * WAVE]; // per-wave queues of complex reads
    __shared__ int32_t  s_wtot[NW];
    __shared__ uint32_t s_wcnt[NW];
    __shared__ uint32_t s_hasb;
    __shared__ uint32_t s_base;
    int32_t* const s_diff = s_diffp + 4;

    // XCD-aware order: workgroup b runs on XCD b % 8; every XCD gets a contiguous
    // eighth of the genome so the look-back reads of neighbouring tiles hit the same L2.
    const int per = (job.n_tiles + 7) >> 3;
    const int tile = (int)(blockIdx.x & 7u) * per + (int)(blockIdx.x >> 3);
    if (tile >= job.n_tiles) return;

    const int tid = threadIdx.x;
    const int lane = tid & (WAVE - 1);
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const TileInfo ti = job.tiles[tile];
    const int seen0 = __hip_atomic_load(&job.counters->max_span, __ATOMIC_RELAXED,
                                        __HIP_MEMORY_SCOPE_AGENT);
    const int32_t t0 = ti.t0;
    const int32_t tend = t0 + T < ti.length ? t0 + T : ti.length;   // clipped tile end
    const int tlen = tend - t0;                                      // valid positions, 1..T
    const int T4 = tlen * 4;

    // ---- loads first: the descriptors of the first batch ----------------------
    const uint32_t nrd = ti.hi - ti.lo;
    const uint32_t n_units = (ti.n_reads + 63u) >> 6;
    const rsrc_t r_desc = make_rsrc(ti.desc + ti.lo, nrd * 8u);
    const rsrc_t r_cxb = make_rsrc(ti.cxb, (n_units + 1u) * 4u);
    u32x2 d[U];
    uint32_t cb[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        d[u] = __builtin_amdgcn_raw_buffer_load_b64(r_desc, (tid + u * NT) * 8, 0, 0);
        cb[u] = __builtin_amdgcn_raw_buffer_load_b32(r_cxb, (int)(((ti.lo + (uint32_t)(tid + u * NT)) >> 6) << 2), 0, 0);
    }

    // ---- zero LDS (overlaps the loads above) -----------------------------
    {
        const int4 z = make_int4(0, 0, 0, 0);
        int4* d4 = reinterpret_cast<int4*>(s_diffp);
#pragma unroll
        for (int i = tid; i < T / 4 + 1; i += NT) d4[i] = z;
        for (int i = tid; i < NWORDS; i += NT) { s_bmap[i] = 0; s_clo[i] = 0; s_chi[i] = 0; }
        if (tid == 0) s_hasb = 0;
    }
    __syncthreads();

    // ---- phase A: reads -> clipped intervals -> LDS +1/-1 -----------------
    if (nrd != 0) {
        PhaseA8 A;
        A.desc = ti.desc; A.cxb = ti.cxb; A.cxc = ti.cxc; A.off = ti.off; A.gcig = ti.cigar;
        A.lo = ti.lo; A.nrd = nrd; A.n_units = n_units; A.n_cx = ti.cxb[n_units];
        A.s_diff = s_diff; A.wq = &s_wq[wv * (QW * WAVE)];
        A.neg4t0 = (int)(0u - ((uint32_t)t0 << 2));       // (p<<2) + neg4t0 = 4*(p - t0)
        A.T4 = T4; A.fmask = (job.flag_mask & 0xfffu) << 8; A.Q = job.Q; A.tid = tid; A.lane = lane;
        const uint32_t smax = phase_a<NT, false>(A, d, cb);
        // publish the largest span seen (see gd_tile_v6.hpp)
        if (smax > (uint32_t)seen0) atomicMax(&job.counters->max_span, (int32_t)smax);
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
        if (tlen == T && !wide) v7::phase_b_rows<ROWS, OPT>(B, job.w_magic, job.w_shift, job.s_magic, job.s_shift);
        else                    gd::phase_b_rows<ROWS, false, true, OPT>(B);   // clipped or very deep tiles
    }
    __syncthreads();

    // ---- phase C: compact class boundaries of this tile -------------------
    phase_c<T, NT>(job, tile, t0, ti.ctg, tid, lane, wv, s_bmap, s_clo, s_chi, s_wcnt, &s_hasb, &s_base);
}

// K1s: GD_OUT_SUMS_ONLY on packed descriptors (see v7::gd_tile_sums_kernel).
template <int T, int NT>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(8, 8))) void gd_tile_sums_kernel(Job job)
{
    constexpr int NW = NT / WAVE;
    constexpr int U = 4;
    constexpr int NACC = T / 32 + 2;       // windows a tile can touch when W >= 32

    __shared__ unsigned long long s_acc[NACC];
    __shared__ uint32_t s_wq[NW * QW * WAVE];

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
    const uint32_t n_units = (ti.n_reads + 63u) >> 6;
    const rsrc_t r_desc = make_rsrc(ti.desc + ti.lo, nrd * 8u);
    const rsrc_t r_cxb = make_rsrc(ti.cxb, (n_units + 1u) * 4u);
    u32x2 d[U];
    uint32_t cb[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        d[u] = __builtin_amdgcn_raw_buffer_load_b64(r_desc, (tid + u * NT) * 8, 0, 0);
        cb[u] = __builtin_amdgcn_raw_buffer_load_b32(r_cxb, (int)(((ti.lo + (uint32_t)(tid + u * NT)) >> 6) << 2), 0, 0);
    }
    for (int i = tid; i < NACC; i += NT) s_acc[i] = 0ull;
    __syncthreads();

    const uint32_t w_first = v7::div_magic((uint32_t)t0, job.w_magic, job.w_shift);
    v7::SumSink S;
    S.acc = s_acc; S.r0 = (uint32_t)t0 - w_first * (uint32_t)job.W; S.W = (uint32_t)job.W;
    S.w_magic = job.w_magic; S.w_shift = job.w_shift; S.tlen = tlen;
    if (nrd != 0) {
        PhaseA8 A;
        A.desc = ti.desc; A.cxb = ti.cxb; A.cxc = ti.cxc; A.off = ti.off; A.gcig = ti.cigar;
        A.lo = ti.lo; A.nrd = nrd; A.n_units = n_units; A.n_cx = ti.cxb[n_units];
        A.s_diff = nullptr; A.wq = &s_wq[wv * (QW * WAVE)];
        A.neg4t0 = (int)(0u - ((uint32_t)t0 << 2));
        A.T4 = tlen * 4; A.fmask = (job.flag_mask & 0xfffu) << 8; A.Q = job.Q; A.tid = tid; A.lane = lane;
        const uint32_t smax = phase_a<NT, true>(A, d, cb, &S);
        if (smax > (uint32_t)seen0) atomicMax(&job.counters->max_span, (int32_t)smax);
    }
    __syncthreads();
    // the tile's share of every window it touches
    const uint32_t n_touch = v7::div_magic((uint32_t)(tlen - 1) + S.r0, job.w_magic, job.w_shift) + 1u;
    unsigned long long* const wsum = reinterpret_cast<unsigned long long*>(job.win_sum + ti.win_off) + w_first;
    for (uint32_t k = (uint32_t)tid; k < n_touch && k < (uint32_t)NACC; k += NT) {
        const unsigned long long v = s_acc[k];
        if (v) atomicAdd(&wsum[k], v);
    }
}

}  // namespace v8
}  // namespace gd
