// gd_multidepth.hpp -- device side of `multidepth` (/root/reference/multidepth/multidepth.go).
//
// The reference runs `samtools depth -q 0 -Q Q -d MaxCov -r chrom:start bam1 bam2 ...`
// per 5 Mb chunk (:203-207), parses one text line per covered position back into
// ints (:148-161), asks whether more than minSamples samples reach MinCov
// (sufficientDepth, :163-171) and, for the blocks its state machine cuts out
// (:217-258), averages every sample's depth over the block's sufficient sites
// (means, :270-283).  Here the S samples are S contigs of one engine whose
// per-base vectors are already in HBM after gd_compute:
//   gd_md_flags_kernel  one thread per position: reads the S depths (coalesced
//       per sample), writes two bitmaps -- `any` (some sample has depth > 0: the
//       positions a multi-file `samtools depth` prints) and `suf`
//       (sufficientDepth).  HBM-bound: 4*S bytes per position read, 2 bits written.
//   gd_md_sums_kernel   one thread per (block, sample): the reference's
//       `dps[i] += float64(d) / 1000.` over the block's sufficient sites, in
//       position order, in IEEE double -- the same operations in the same order,
//       so the "%.2f" the host prints is the reference's to the last digit.
// The block state machine itself runs on the host over the two bitmaps
// (host/multidepth_host.cpp); every value crossing the ABI is an integer bitmap
// or the exact double the reference would hold.
#pragma once

namespace gd {

struct MdFlagsJob {
    const int32_t* const* depth;   // [n_samples] per-base vectors (device pointers)
    int32_t  n_samples;
    int64_t  len;                  // positions
    int32_t  min_cov;
    int32_t  min_samples;          // sufficient: count(depth >= min_cov) > min_samples
    uint32_t* any_bits;            // [ceil(len/32)] little-endian bit per position
    uint32_t* suf_bits;
};

__global__ __launch_bounds__(256) void gd_md_flags_kernel(MdFlagsJob j)
{
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool valid = p < j.len;
    int cnt = 0;
    int nz = 0;
    int s = 0;
    for (; s + 4 <= j.n_samples; s += 4) {               // four independent loads in flight
        const int32_t* d0 = j.depth[s];
        const int32_t* d1 = j.depth[s + 1];
        const int32_t* d2 = j.depth[s + 2];
        const int32_t* d3 = j.depth[s + 3];
        const int a = valid ? d0[p] : 0, b = valid ? d1[p] : 0, c = valid ? d2[p] : 0, d = valid ? d3[p] : 0;
        cnt += (a >= j.min_cov) + (b >= j.min_cov) + (c >= j.min_cov) + (d >= j.min_cov);
        nz |= a | b | c | d;
    }
    for (; s < j.n_samples; ++s) {
        const int a = valid ? j.depth[s][p] : 0;
        cnt += a >= j.min_cov;
        nz |= a;
    }
    // depths are >= 0, so nz != 0 <=> some sample covers the position
    const unsigned long long am = __builtin_amdgcn_ballot_w64(valid && nz != 0);
    const unsigned long long sm = __builtin_amdgcn_ballot_w64(valid && cnt > j.min_samples);
    const int lane = threadIdx.x & 63;
    const int64_t w = p >> 5;                              // the wave covers words w, w+1 of lane 0
    const int64_t n_words = (j.len + 31) >> 5;
    if (lane == 0) {
        if (w < n_words) { j.any_bits[w] = (uint32_t)am; j.suf_bits[w] = (uint32_t)sm; }
        if (w + 1 < n_words) { j.any_bits[w + 1] = (uint32_t)(am >> 32); j.suf_bits[w + 1] = (uint32_t)(sm >> 32); }
    }
}

struct MdSumsJob {
    const int32_t* const* depth;   // [n_samples]
    const uint32_t* suf_bits;
    const int64_t* start;          // [n_blocks] first position of the block
    const int64_t* end;            // [n_blocks] one past its last position
    double* sums;                  // [n_blocks][n_samples]
    int64_t n_blocks;
    int32_t n_samples;
};

__global__ __launch_bounds__(64) void gd_md_sums_kernel(MdSumsJob j)
{
    const int64_t gid = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (gid >= j.n_blocks * j.n_samples) return;
    const int64_t b = gid / j.n_samples;
    const int s = (int)(gid - b * j.n_samples);
    const int32_t* const d = j.depth[s];
    const int64_t s0 = j.start[b], e0 = j.end[b];
    double acc = 0.0;                                       // multidepth.go:271-277
    for (int64_t w = s0 >> 5; w <= (e0 - 1) >> 5 && e0 > s0; ++w) {
        uint32_t bits = j.suf_bits[w];
        const int64_t base = w << 5;
        if (base < s0) bits &= ~0u << (int)(s0 - base);
        if (base + 32 > e0) bits &= ~0u >> (int)(base + 32 - e0);
        while (bits) {
            const int k = __ffs((int)bits) - 1;
            bits &= bits - 1;
            acc += (double)d[base + k] / 1000.;
        }
    }
    j.sums[gid] = acc;
}

}  // namespace gd
