// gd_scatter.hpp -- the long-read path of the per-base depth engine.
//
// The tile kernel (gd_tile_v6.hpp) re-examines, for every tile, all reads that
// START within one maximum read span before it; that is the right trade for
// short reads (span ~150) and hopeless for long reads (ONT, N50 ~20 kb, up to
// megabases; spliced RNA-seq with 100 kb introns).  This path has no look-back:
//
//   LK1 gd_expand_scatter_kernel  every CIGAR op is expanded exactly once
//       (short CIGARs lane-serial, long CIGARs cooperatively with a saturating
//       wavefront scan of the reference-consuming lengths) and each counted
//       op (M/=/X, what `samtools depth` counts, /root/reference/depth/depth.go:45)
//       adds +1 / -1 to a global int32 difference array -- the per-base result
//       array itself -- with device-scope integer atomics (order independent,
//       hence bit exact).
//   LK2 gd_scan_kernel  one workgroup per tile of T positions turns the
//       difference array into depth IN PLACE: the tile is staged in LDS, its
//       carry-in comes from a decoupled look-back over per-tile status words
//       (single pass, 8 bytes of HBM traffic per base), then the same
//       phase-B rows as the tile kernel scan, store, reduce windows and detect
//       class boundaries (depth/depth.go:293-323).
#pragma once

namespace gd {

constexpr uint32_t POS_CAP = 0x7fffffffu;     // positions saturate here (contigs are < 2^31)
constexpr uint32_t SHORT_OPS = 8;             // CIGARs up to this many ops are walked by one lane

// Saturating (at POS_CAP) inclusive prefix sum across the wave; inputs <= POS_CAP.
__device__ __forceinline__ uint32_t wave_inclusive_scan_sat(uint32_t v)
{
    uint32_t o;
    o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true); v += o; v = v < POS_CAP ? v : POS_CAP;
    o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true); v += o; v = v < POS_CAP ? v : POS_CAP;
    o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true); v += o; v = v < POS_CAP ? v : POS_CAP;
    o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true); v += o; v = v < POS_CAP ? v : POS_CAP;
    o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false); v += o; v = v < POS_CAP ? v : POS_CAP;
    o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false); v += o; v = v < POS_CAP ? v : POS_CAP;
    return v;
}

// One +-1 mark of the difference array: a device-scope atomic (units of different workgroups, on
// different XCDs, mark the same cache lines; workgroup scope would not be coherent across XCDs).
__device__ __forceinline__ void scatter_mark(int32_t* diff, uint32_t p, int v, uint32_t clen)
{
    if (p < clen) __hip_atomic_fetch_add(&diff[p], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// LK0: accumulators, counters, look-back status words.
__global__ void gd_linit_kernel(Job job)
{
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for (int64_t w = gid; w < job.n_win_total; w += gsz) {
        job.win_sum[w] = 0;
        job.win_min[w] = 0x7fffffff;
    }
    for (int64_t g = gid; g < (job.n_tiles + SUPER - 1) / SUPER; g += gsz) job.super_cnt[g] = 0;
    for (int64_t t = gid; t < job.n_tiles; t += gsz) job.tile_status[t] = 0ull;
    if (gid == 0) {
        job.counters->max_span = 0;
        job.counters->run_cursor = 0;
        job.counters->pad0 = 0;      // scan ticket
        job.counters->pad1 = 0;      // look-back timeout flag
    }
}

// LK1: one wave per unit of 64 consecutive reads of one contig.
//
// A read contributes +1 on every maximal run of counted ops (M/=/X) that is
// not interrupted by reference-consuming uncounted ops (D/N of positive
// length): insertions, clips and pads between two matches leave the intervals
// adjacent, and adjacent intervals need no marks at the shared edge.  So a
// read costs 2 x (number of deletions/skips + 1) atomics, not 2 per match op.
__global__ __launch_bounds__(256) void gd_expand_scatter_kernel(Job job)
{
    // XCD-contiguous order (workgroup b runs on XCD b % 8): neighbouring units
    // touch neighbouring cache lines of the difference array
    const uint32_t n_groups = (job.n_units + 3u) >> 2;
    const uint32_t per = (n_groups + 7u) >> 3;
    const uint32_t grp = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
    const int lane = threadIdx.x & 63;
    const uint32_t unit = grp * 4u + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (grp >= n_groups || unit >= job.n_units) return;

    // contig of this unit: last c with unit_beg[c] <= unit
    int lo = 0, hi = job.n_ctgs;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (job.ctgs[mid].unit_beg <= unit) lo = mid; else hi = mid;
    }
    const ContigDev& c = job.ctgs[lo];
    const uint32_t n_reads = c.n_reads;
    const uint32_t clen = (uint32_t)c.length;
    int32_t* const diff = job.perbase + c.base_off;
    const uint32_t* const cigar = c.cigar;

    const uint32_t r = (unit - c.unit_beg) * 64u + (uint32_t)lane;
    const bool valid = r < n_reads;
    const uint32_t rc = valid ? r : (n_reads ? n_reads - 1u : 0u);
    uint32_t p = 0, o0 = 0, n = 0;
    bool keep = false;
    if (n_reads) {
        p = (uint32_t)c.pos[rc];
        const uint32_t f = c.flag[rc], mq = c.mapq[rc];
        o0 = c.off[rc];
        n = c.off[rc + 1] - o0;
        keep = valid && (f & job.flag_mask) == 0 && (int)mq >= job.Q && n != 0;
    }

    // ---- short CIGARs: each lane walks its own read ---------------------
    if (keep && n <= SHORT_OPS) {
        uint32_t cur = p;
        bool open = false;                                // inside a run of counted ops
        for (uint32_t k = 0; k < n; ++k) {
            const uint32_t cg = cigar[o0 + k];
            const uint32_t op = cg & 0xf, len = cg >> 4;
            const bool counted = (0x181u >> op) & 1u;     // M = X
            const bool consumes = (0x18du >> op) & 1u;    // M D N = X
            if (len != 0) {
                if (counted && !open) { scatter_mark(diff, cur, 1, clen); open = true; }
                if (consumes && !counted && open) { scatter_mark(diff, cur, -1, clen); open = false; }
                if (consumes) { cur += len; cur = cur < POS_CAP ? cur : POS_CAP; }
            }
        }
        if (open) scatter_mark(diff, cur, -1, clen);
    }

    // ---- long CIGARs: the wave expands one read at a time, 64 ops per round
    unsigned long long todo = __ballot(keep && n > SHORT_OPS);
    const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane));     // lanes < lane
    const unsigned long long above = lane == 63 ? 0ull : (~0ull << (lane + 1));     // lanes > lane
    while (todo != 0ull) {
        const int j = __ffsll((long long)todo) - 1;
        todo &= todo - 1ull;
        const uint32_t pj = (uint32_t)__builtin_amdgcn_readlane((int)p, j);
        const uint32_t oj = (uint32_t)__builtin_amdgcn_readlane((int)o0, j);
        const uint32_t nj = (uint32_t)__builtin_amdgcn_readlane((int)n, j);
        uint32_t carry = pj;                               // reference position of the round's first op
        for (uint32_t b = 0; b < nj; b += 64u) {
            const uint32_t k = b + (uint32_t)lane;
            const uint32_t cg = k < nj ? cigar[oj + k] : 0u;   // a zero-length M: contributes nothing
            const uint32_t op = cg & 0xf, len = cg >> 4;
            const bool counted = ((0x181u >> op) & 1u) && len != 0;
            const bool consumes = ((0x18du >> op) & 1u) && len != 0;
            const uint32_t cons = consumes ? len : 0u;
            const uint32_t incl = wave_inclusive_scan_sat(cons);
            uint32_t s = carry + (incl - cons);            // <= 2 * POS_CAP: no wrap
            s = s < POS_CAP ? s : POS_CAP;
            // runs of counted ops inside this round; a run is closed at the round's end
            // (the +1 of its continuation lands on the same position and cancels)
            const unsigned long long cm = __ballot(counted);
            const unsigned long long nn = cm | __ballot(consumes);       // non-neutral ops
            if (counted) {
                const unsigned long long pm = nn & below;
                const bool prev_counted = pm != 0ull && ((cm >> (63 - __builtin_clzll(pm))) & 1ull);
                const unsigned long long nm = nn & above;
                const bool next_counted = nm != 0ull && ((cm >> (__builtin_ffsll((long long)nm) - 1)) & 1ull);
                if (!prev_counted) scatter_mark(diff, s, 1, clen);
                if (!next_counted) scatter_mark(diff, s + len, -1, clen);
            }
            carry += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            carry = carry < POS_CAP ? carry : POS_CAP;
            if (carry >= clen) break;                      // the rest of the read lies past the contig
        }
    }
}

// LK2: difference array -> depth in place, windows, class boundaries.
template <int T, int NT>
__global__ __launch_bounds__(NT) void gd_scan_kernel(Job job)
{
    constexpr int NW = NT / WAVE;
    constexpr int CHUNK = T / NW;
    constexpr int ROWS = CHUNK / 256;
    constexpr int NWORDS = T / 32;
    constexpr unsigned long long ST_AGG = 1ull << 32, ST_INC = 2ull << 32;
    static_assert(CHUNK % 256 == 0, "wave chunk must be whole rows");

    __shared__ __attribute__((aligned(16))) int32_t s_diffp[T + 4];
    __shared__ uint32_t s_bmap[NWORDS];
    __shared__ uint32_t s_clo[NWORDS];
    __shared__ uint32_t s_chi[NWORDS];
    __shared__ int32_t  s_wtot[NW];
    __shared__ int32_t  s_wpos[NW];
    __shared__ uint32_t s_wcnt[NW];
    __shared__ uint32_t s_hasb;
    __shared__ uint32_t s_base;
    __shared__ int32_t  s_tile;
    __shared__ int32_t  s_prefix;
    int32_t* const s_diff = s_diffp + 4;

    const int tid = threadIdx.x;
    const int lane = tid & (WAVE - 1);
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);

    // tiles are handed out in start order, so every predecessor of a tile is
    // running or finished when the tile looks back (no dispatch-order assumption)
    if (tid == 0) s_tile = (int32_t)atomicAdd(&job.counters->pad0, 1u);
    __syncthreads();
    const int tile = s_tile;
    if (tile >= job.n_tiles) return;

    int lo = 0, hi = job.n_ctgs;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (job.ctgs[mid].tile_beg <= tile) lo = mid; else hi = mid;
    }
    const ContigDev& c = job.ctgs[lo];
    const int32_t t0 = (tile - c.tile_beg) * T;
    const int32_t tend = t0 + T < c.length ? t0 + T : c.length;
    const int tlen = tend - t0;
    int32_t* const gtile = job.perbase + c.base_off + t0;

    // ---- stage the tile's differences in LDS -----------------------------
    {
        const int4* src = reinterpret_cast<const int4*>(gtile);
        int4* d4 = reinterpret_cast<int4*>(s_diff);
#pragma unroll
        for (int i = tid; i < T / 4; i += NT) d4[i] = src[i];
        for (int i = tid; i < NWORDS; i += NT) { s_bmap[i] = 0; s_clo[i] = 0; s_chi[i] = 0; }
        if (tid == 0) s_hasb = 0;
    }
    __syncthreads();

    // ---- pass 1: wave totals (and the sum of the positive differences, a
    // bound on how far the depth can rise inside the tile) -----------------
    const int chunk0 = wv * CHUNK;
    {
        int tot = 0, pos = 0;
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const int4 v = *reinterpret_cast<const int4*>(&s_diff[chunk0 + r * 256 + lane * 4]);
            tot += v.x + v.y + v.z + v.w;
            pos += (v.x > 0 ? v.x : 0) + (v.y > 0 ? v.y : 0) + (v.z > 0 ? v.z : 0) + (v.w > 0 ? v.w : 0);
        }
        tot = wave_total(tot);
        pos = wave_total(pos);
        if (lane == 0) { s_wtot[wv] = tot; s_wpos[wv] = pos; }
    }
    __syncthreads();

    // ---- decoupled look-back (wave 0): exclusive prefix of the tile totals
    if (wv == 0) {
        int total = 0;
#pragma unroll
        for (int v = 0; v < NW; ++v) total += s_wtot[v];
        int prefix = 0;
        unsigned long long* const status = job.tile_status;
        if (tile == c.tile_beg) {
            if (lane == 0)
                __hip_atomic_store(&status[tile], ST_INC | (uint32_t)total, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        } else {
            if (lane == 0)
                __hip_atomic_store(&status[tile], ST_AGG | (uint32_t)total, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            int look = tile - 1;
            uint32_t spins = 0;
            for (;;) {
                const int idx = look - lane;
                unsigned long long st = ST_INC;            // before the contig: inclusive 0
                if (idx >= c.tile_beg)
                    st = __hip_atomic_load(&status[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t flag = (uint32_t)(st >> 32);
                const int val = (int)(uint32_t)st;
                const unsigned long long ready = __ballot(flag != 0u);
                const unsigned long long incm = __ballot(flag == 2u);
                if (incm != 0ull) {
                    const int fi = __ffsll((long long)incm) - 1;
                    const unsigned long long need = fi == 63 ? ~0ull : ((1ull << (fi + 1)) - 1ull);
                    if ((ready & need) == need) {
                        prefix += wave_total(lane <= fi ? val : 0);
                        break;
                    }
                } else if (ready == ~0ull) {
                    prefix += wave_total(val);
                    look -= 64;
                    continue;
                }
                if (++spins > (1u << 24)) {                // never expected: report, do not hang
                    if (lane == 0) atomicMax(&job.counters->pad1, 1u);
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
            if (lane == 0)
                __hip_atomic_store(&status[tile], ST_INC | (uint32_t)(prefix + total), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        }
        if (lane == 0) s_prefix = prefix;
    }
    __syncthreads();

    // ---- pass 2: the tile kernel's rows, carry-in from the look-back -----
    {
        PhaseB B;
        B.s_diff = s_diff; B.s_bmap = s_bmap; B.s_clo = s_clo; B.s_chi = s_chi; B.s_hasb = &s_hasb;
        B.out = gtile;
        B.wsum = job.win_sum + c.win_off;
        B.wmin = job.win_min + c.win_off;
        B.t0 = t0; B.tlen = tlen; B.chunk0 = chunk0; B.lane = lane;
        B.W = job.W; B.mincov = job.mincov; B.maxmean = job.maxmean; B.step = job.step;
        const int prefix = s_prefix;                       // depth at t0-1
        int carry = prefix, rise = 0;
#pragma unroll
        for (int v = 0; v < NW; ++v) { carry += v < wv ? s_wtot[v] : 0; rise += s_wpos[v]; }
        B.carry = carry;
        // depth inside the tile <= prefix + sum of positive differences
        const bool wide = (long long)prefix + rise >= (1 << 22);
        if (tlen == T && !wide) phase_b_rows<ROWS, true, false, 0>(B);
        else                    phase_b_rows<ROWS, false, true, 0>(B);
    }
    __syncthreads();

    phase_c<T, NT>(job, tile, t0, lo, tid, lane, wv, s_bmap, s_clo, s_chi, s_wcnt, &s_hasb, &s_base);
}

}  // namespace gd
