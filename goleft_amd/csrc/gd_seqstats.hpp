// gd_seqstats.hpp -- `goleft depth --stats` columns on the device.
//
// The reference appends "\t%.3g\t%.3g\t%.3g" of faidx.Stats(chrom, s, e) --
// GC, CpG and masked fraction of the window's reference bases -- to every
// depth.bed row (/root/reference/depth/depth.go:191-200, called from the
// callback at :299-301, :333, :356; FASTA opened at :244-252).  faidx is an
// external module (github.com/brentp/faidx @c39eb85, go.mod:12) whose source is
// not under /root/reference and whose values no reference test asserts:
// PARITY UNPINNED.  The semantics restated in oracle/pyoracle.py::seq_stats
// (and host/fasta_stats.hpp) are the contract here:
//   gc     = #{i in [s,e) : seq[i] in "GCgc"}
//   masked = #{i in [s,e) : 'a' <= seq[i] <= 'z'}
//   cpg    = #{i in [s,e) : seq[i] in "Cc" and i+1 < len and seq[i+1] in "Gg"}
// with [s,e) clipped to the contig.  Because the contract is a restatement from memory, the kernel also
// returns what the OTHER plausible reading of faidx.Stats needs (include/goleft_depth_host.h, GDH_STATS_*):
//   acgt        = #{i : seq[i] in "ACGTacgt"}            (a denominator that skips N and IUPAC codes)
//   masked_acgt = #{i : seq[i] in "acgt"}
// and can ignore a CpG whose C is the last base of a FASTA line (line_bases > 0): a scan of the raw, line
// broken file -- which is what a memory-mapped faidx does -- sees "C\nG" there, not "CG".
// The kernel returns integer counts; the divisions and the %.3g stay on the host.
//
// One wave per window.  HBM-bound: 1 byte per reference base read once, 12 bytes
// per window written.  Lanes read aligned 32-bit words through a buffer
// descriptor bound to the contig (bytes past the end read as 0) and classify
// four bases at a time with SWAR byte masks; no LDS, one DPP reduction per counter.
#pragma once

namespace gd {

struct SeqStatsJob {
    const uint8_t* seq;        // one contig, newline free, zero padded to `padded` bytes
    int64_t  len;
    uint32_t padded;           // multiple of 4, >= len + 4 (whole-word loads never leave it)
    const int64_t* win_start;  // [n_win]
    const int64_t* win_end;    // [n_win]
    uint32_t* gc;              // [n_win]
    uint32_t* cpg;
    uint32_t* masked;
    uint32_t* acgt;            // null: not wanted
    uint32_t* masked_acgt;     // null: not wanted
    int64_t  n_win;
    uint32_t line_bases;       // > 0: bases per FASTA line; a C at the end of a line never starts a CpG
};

// bit 7 of every byte of x that is zero (exact: no carries between bytes)
__device__ __forceinline__ uint32_t zero_bytes(uint32_t x)
{
    const uint32_t t = (x & 0x7f7f7f7fu) + 0x7f7f7f7fu;
    return ~(t | x | 0x7f7f7f7fu);
}

// bit 7 of bytes lo..hi-1 (0 <= lo, hi <= 4)
__device__ __forceinline__ uint32_t byte_range_mask(int lo, int hi)
{
    uint32_t m = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) m |= (j >= lo && j < hi) ? (0x80u << (8 * j)) : 0u;
    return m;
}

__global__ __launch_bounds__(256) void gd_seq_stats_kernel(SeqStatsJob job)
{
    const int lane = threadIdx.x & 63;
    const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= job.n_win) return;
    int64_t s = job.win_start[w], e = job.win_end[w];
    s = s < 0 ? 0 : s;
    e = e > job.len ? job.len : e;
    uint32_t n_gc = 0, n_cpg = 0, n_low = 0, n_acgt = 0, n_lacgt = 0;
    const uint32_t lb = job.line_bases;
    if (e > s) {
        // bytes past the contig are 0 (never a base), so the look-ahead at the contig end
        // needs no special case
        const rsrc_t r = make_rsrc(job.seq, job.padded);
        const int64_t a0 = s & ~(int64_t)3;                     // first aligned word
        const int64_t nwords = ((e + 3) >> 2) - (a0 >> 2);
        for (int64_t k = lane; k < nwords; k += 64) {
            const int64_t p0 = a0 + 4 * k;                      // position of byte 0 of this word
            const uint32_t x = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(r, (int)p0, 0, 0);
            const uint32_t nx = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(r, (int)p0 + 4, 0, 0);
            const int lo = (int)(s - p0 > 0 ? s - p0 : 0), hi = (int)(e - p0 < 4 ? e - p0 : 4);
            const uint32_t in = (lo == 0 && hi == 4) ? 0x80808080u : byte_range_mask(lo, hi);
            const uint32_t y = x | 0x20202020u;                 // fold case: only C/c -> 'c', only G/g -> 'g'
            const uint32_t is_c = zero_bytes(y ^ 0x63636363u);
            const uint32_t is_g = zero_bytes(y ^ 0x67676767u);
            const uint32_t ng = zero_bytes(((nx & 0xffu) | 0x20u) ^ 0x67u) & 0x80u;   // first base of the next word
            const uint32_t g_next = (is_g >> 8) | (ng << 24);
            const uint32_t x7 = x & 0x7f7f7f7fu;
            const uint32_t ge_a = (x7 + 0x1f1f1f1fu) & 0x80808080u;                  // low 7 bits >= 'a'
            const uint32_t gt_z = (x7 + 0x05050505u) & 0x80808080u;                  // low 7 bits >  'z'
            const uint32_t low = ge_a & ~gt_z & ~x;                                  // and bit 7 clear
            const uint32_t is_at = zero_bytes(y ^ 0x61616161u) | zero_bytes(y ^ 0x74747474u);
            const uint32_t acgt = (is_c | is_g | is_at) & in;
            uint32_t not_eol = 0x80808080u;                                          // bases that are not the last of a line
            if (lb != 0u) {
                const uint32_t r0 = (uint32_t)(p0 % (int64_t)lb);                    // column of byte 0
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if ((r0 + (uint32_t)j) % lb == lb - 1u) not_eol &= ~(0x80u << (8 * j));
            }
            n_gc += __popc((is_c | is_g) & in);
            // line aware (the raw-scan contract): a scan of the window's own bytes cannot see the base after its last
            // one either -- a C at e - 1 starts no CpG
            uint32_t in_cpg = in;
            if (lb != 0u && e - 1 >= p0 && e - 1 < p0 + 4) in_cpg &= ~(0x80u << (8 * (int)(e - 1 - p0)));
            n_cpg += __popc(is_c & g_next & in_cpg & not_eol);
            n_low += __popc(low & in);
            n_acgt += __popc(acgt);
            n_lacgt += __popc(acgt & ((x & 0x20202020u) << 2));                      // bit 5: lower case
        }
    }
    n_gc = (uint32_t)wave_total((int)n_gc);
    n_cpg = (uint32_t)wave_total((int)n_cpg);
    n_low = (uint32_t)wave_total((int)n_low);
    n_acgt = (uint32_t)wave_total((int)n_acgt);
    n_lacgt = (uint32_t)wave_total((int)n_lacgt);
    if (lane == 0) {
        job.gc[w] = n_gc; job.cpg[w] = n_cpg; job.masked[w] = n_low;
        if (job.acgt) job.acgt[w] = n_acgt;
        if (job.masked_acgt) job.masked_acgt[w] = n_lacgt;
    }
}

}  // namespace gd
