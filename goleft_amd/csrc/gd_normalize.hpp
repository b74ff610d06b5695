// gd_normalize.hpp -- canonical CIGARs, built once when a contig's records arrive.
//
// `samtools depth` (no -J, no -q; /root/reference/depth/depth.go:45) distinguishes exactly two kinds
// of CIGAR op: M/=/X add 1 to every reference position they cover, D/N advance the reference without
// counting; I/S/H/P touch no reference position at all.  The canonical form keeps just that:
//
//   * I, S, H, P and zero-length ops are dropped;
//   * neighbouring M/=/X ops (neighbours once the dropped ops are gone) merge into ONE `M`,
//     neighbouring D/N ops into ONE `N` (a merged length that would not fit the 28-bit BAM length field
//     starts a new op of the same kind instead -- pathological input only);
//   * D/N ops after the last M are dropped (they cover nothing that is counted); a read without any
//     counted base has no ops at all.  Leading D/N ops stay: they shift where the first M begins.
//
// The result uses the BAM encoding (len << 4 | op, op = 0 `M` or 3 `N`), so every kernel that walks
// CIGARs reads it unchanged and produces the identical depth -- but a 150 bp read with soft clips or an
// insertion is now the single op `kM` (98 % of short reads instead of 92 %), which is what the tile
// kernel's straight-line path handles, and a long read carries about half as many ops.
//
// Three launches per contig, no host round trip: (N1) one lane per read counts its canonical ops, a wave
// scan gives the offsets inside each 64-read unit and the unit totals; (N2) one workgroup scans the
// unit totals; (N3) one lane per read walks its ops again and writes them at unit offset + local offset.
// The canonical array is allocated at the size of the original (it can only shrink).
#pragma once

namespace gd {
namespace norm {

// The record word of a read, written next to its canonical CIGAR: everything the tile kernel needs
// besides `pos` and the ops themselves, in the 4 bytes the CSR offset would take --
//     flag (12 bits, SAMv1 defines 12) << 20 | MAPQ << 12 | number of canonical ops (0..4094)
// The op offsets are recovered in the kernel by a prefix sum over the tile's reads.  A contig with a FLAG
// above 0xfff or a read of 4095 or more canonical ops is marked (status bit) and runs the generic kernel.
constexpr uint32_t REC_NMAX = 0xfffu;

struct NormJob {
    const uint32_t* off;      // original CSR offsets (n_reads + 1)
    const uint32_t* cigar;    // original ops
    const uint16_t* flag;
    const uint8_t*  mapq;
    uint32_t* rec;            // record words (n_reads)
    uint32_t* status;         // bit 0: a record that does not fit its word
    uint32_t  n_reads;
    uint32_t  n_units;        // ceil(n_reads / 64)
    uint32_t* noff;           // canonical CSR offsets (n_reads + 1)
    uint32_t* unit;           // n_units + 1: unit totals, then (after N2) exclusive offsets; [n_units] = grand total
    uint32_t* ncig;           // canonical ops
};

constexpr uint32_t LEN_MAX = 0x0fffffffu;

// Calls emit(op, len) for every canonical op of one read, in order; returns their number.
template <typename Emit>
__device__ __forceinline__ uint32_t canonical_walk(const uint32_t* __restrict__ ops, uint32_t n, Emit emit)
{
    uint32_t cnt = 0;
    uint32_t cur_len = 0, cur_kind = 2;        // open run: kind 0 = M, 1 = N, 2 = none
    // An N run is only emitted once an M follows it (trailing N runs are dropped).  When a merged N
    // length overflows the 28-bit field, full-length pieces are split off and counted in `held` (all of
    // length LEN_MAX by construction) until that M arrives.
    uint32_t held = 0;
    const uint32_t held_len = LEN_MAX;
    for (uint32_t k = 0; k < n; ++k) {
        const uint32_t cg = ops[k];
        const uint32_t op = cg & 0xfu, len = cg >> 4;
        const bool counted = (0x181u >> op) & 1u;     // M = X
        const bool consumes = (0x18du >> op) & 1u;    // M D N = X
        if (!consumes || len == 0u) continue;
        const uint32_t kind = counted ? 0u : 1u;
        if (kind == cur_kind && cur_len + len <= LEN_MAX) { cur_len += len; continue; }
        if (kind == cur_kind) {
            // overflow split: close a full-length op of this kind, keep the remainder open
            const uint32_t rest = cur_len + len - LEN_MAX;
            if (kind == 0u) { emit(0u, LEN_MAX); ++cnt; }
            else ++held;
            cur_len = rest;
            continue;
        }
        // the kind changes: close the open run
        if (cur_kind == 0u) { emit(0u, cur_len); ++cnt; }
        else if (cur_kind == 1u) {
            // an N run followed by an M: now it is known to matter
            for (uint32_t h = 0; h < held; ++h) { emit(3u, held_len); ++cnt; }
            held = 0;
            emit(3u, cur_len); ++cnt;
        }
        cur_kind = kind;
        cur_len = len;
    }
    if (cur_kind == 0u) { emit(0u, cur_len); ++cnt; }   // a trailing N run (and its held parts) is dropped
    return cnt;
}

// N1: count.  One wave per unit of 64 consecutive reads.
__global__ __launch_bounds__(256) void gd_norm_count_kernel(NormJob j)
{
    const int lane = threadIdx.x & 63;
    const uint32_t unit = blockIdx.x * 4u + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (unit >= j.n_units) return;
    const uint32_t r = unit * 64u + (uint32_t)lane;
    uint32_t cnt = 0;
    if (r < j.n_reads) {
        const uint32_t o0 = j.off[r], n = j.off[r + 1] - o0;
        cnt = canonical_walk(j.cigar + o0, n, [](uint32_t, uint32_t) {});
    }
    const uint32_t incl = (uint32_t)wave_inclusive_scan((int)cnt);
    if (r < j.n_reads) {
        j.noff[r] = incl - cnt;                             // offset inside the unit, completed by N3
        const uint32_t f = j.flag[r], q = j.mapq[r];
        const bool fits = f <= 0xfffu && cnt < REC_NMAX;
        j.rec[r] = ((f & 0xfffu) << 20) | (q << 12) | (fits ? cnt : REC_NMAX);
        if (!fits) atomicOr(j.status, 1u);
    }
    if (lane == 63) j.unit[unit] = incl;
}

// N2: exclusive scan of the unit totals, in place; v[n] = grand total.  One workgroup.
__global__ __launch_bounds__(1024) void gd_unit_scan_kernel(uint32_t* __restrict__ v, uint32_t n)
{
    __shared__ uint32_t s_part[1024];
    const uint32_t tid = threadIdx.x;
    const uint32_t per = (n + 1023u) / 1024u;
    const uint32_t b = tid * per < n ? tid * per : n, e = b + per < n ? b + per : n;
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

// N3: write.  Same shape as N1.
__global__ __launch_bounds__(256) void gd_norm_write_kernel(NormJob j)
{
    const int lane = threadIdx.x & 63;
    const uint32_t unit = blockIdx.x * 4u + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (unit >= j.n_units) return;
    const uint32_t r = unit * 64u + (uint32_t)lane;
    if (r >= j.n_reads) return;
    const uint32_t dst0 = j.unit[unit] + j.noff[r];
    j.noff[r] = dst0;
    if (r + 1u == j.n_reads) j.noff[j.n_reads] = j.unit[j.n_units];
    const uint32_t o0 = j.off[r], n = j.off[r + 1] - o0;
    uint32_t* out = j.ncig + dst0;
    uint32_t w = 0;
    canonical_walk(j.cigar + o0, n, [&](uint32_t op, uint32_t len) { out[w++] = (len << 4) | op; });
}

}  // namespace norm
}  // namespace gd
