// gd_index.hpp -- what is built as records ARRIVE, and two helpers the batch kernels share.
//
//   gd_index_records_kernel   one pass over what just became resident (gd_adopt_device's check pass, every committed
//                             block once its copy has landed, what the device BAM walk wrote): coordinate order and CSR
//                             offsets are checked, the POSITION INDEX ridx[k] = first read with pos >= 64 k is written
//                             (gd_prep_kernel looks the tiles' read ranges up there instead of searching), and the
//                             largest reference span of any record of <= 64 ops is kept, so that the first gd_compute
//                             runs with the data's look-back (no re-run);
//   batch_find                the job a 64-read unit of a batch belongs to (gd_chunk.hpp);
//   gd_unit_scan / gd_scan_*  exclusive prefix sums over small device arrays (multidepth's block finder, gd_api_aux.inc).
// (Until round 5 this file was gd_normalize.hpp and also held the canonical-record kernels: a rewritten copy of the
// records for a host that computes the same records many times.  No caller in the reference does -- a `goleft depth` run
// computes each input once, /root/reference/depth/depth.go:392-421 -- and every path reads the records as they arrived;
// they were removed.)
#pragma once

namespace gd {
namespace norm {

// last j with beg[j] <= x (beg[0] = 0, n >= 1, entries ascending)
__device__ __forceinline__ uint32_t batch_find(const uint32_t* __restrict__ beg, uint32_t n, uint32_t x)
{
    uint32_t lo = 0, hi = n;
    while (hi - lo > 1u) {
        const uint32_t mid = (lo + hi) >> 1;
        if (beg[mid] <= x) lo = mid; else hi = mid;
    }
    return lo;
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

// The same scan for large arrays, three launches: totals of blocks of SCAN_BLOCK elements (SB1), the scan of
// those totals by gd_unit_scan_kernel, then every block scans itself from its offset (SB2).  One workgroup alone
// took 0.69 ms for chr1's 780 k units -- 17 of the 28 ms a genome's normalisation took.
constexpr uint32_t SCAN_BLOCK = 4096;      // elements per workgroup: 256 threads x 16

__global__ __launch_bounds__(256) void gd_scan_totals_kernel(const uint32_t* __restrict__ v, uint32_t n,
                                                             uint32_t* __restrict__ btot)
{
    __shared__ uint32_t s_w[4];
    const uint32_t b0 = blockIdx.x * SCAN_BLOCK + threadIdx.x * 16u;
    uint32_t sum = 0;
#pragma unroll
    for (uint32_t k = 0; k < 16u; ++k) sum += b0 + k < n ? v[b0 + k] : 0u;
    const uint32_t incl = (uint32_t)wave_inclusive_scan((int)sum);
    if ((threadIdx.x & 63u) == 63u) s_w[threadIdx.x >> 6] = incl;
    __syncthreads();
    if (threadIdx.x == 0) btot[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

__global__ __launch_bounds__(256) void gd_scan_apply_kernel(uint32_t* __restrict__ v, uint32_t n,
                                                            const uint32_t* __restrict__ boff, uint32_t n_blocks)
{
    __shared__ uint32_t s_w[4];
    const uint32_t b0 = blockIdx.x * SCAN_BLOCK + threadIdx.x * 16u;
    uint32_t x[16];
    uint32_t sum = 0;
#pragma unroll
    for (uint32_t k = 0; k < 16u; ++k) { x[k] = b0 + k < n ? v[b0 + k] : 0u; sum += x[k]; }
    const uint32_t incl = (uint32_t)wave_inclusive_scan((int)sum);
    const uint32_t wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63u) == 63u) s_w[wv] = incl;
    __syncthreads();
    uint32_t run = boff[blockIdx.x] + incl - sum;
    for (uint32_t w = 0; w < wv; ++w) run += s_w[w];
#pragma unroll
    for (uint32_t k = 0; k < 16u; ++k) { if (b0 + k < n) v[b0 + k] = run; run += x[k]; }
    if (blockIdx.x == 0 && threadIdx.x == 0) v[n] = boff[n_blocks];    // grand total (written last by nobody else: slot n is outside every block)
}


// What every way into the engine runs over records once they are resident (gd_adopt_device: the caller's own arrays;
// gd_commit: a staged block that has landed; the device BAM read: a contig the record walk has written) -- ONE pass over
// pos / CSR offsets / ops of the reads [r0, r1) of a contig:
//   * the checks a host block gets in gd_commit (check != 0).  out[0] bits: 0 positions out of order, 1 CSR offsets
//     decreasing / not starting at 0 / ending past the ops, 2 a negative position;
//   * the position index the prep kernel would otherwise search for: ridx[k] = first read with pos >= 64 k, written
//     for every k in (pos[r - 1] >> 6, pos[r] >> 6] by read r (k <= pos[r1 - 1] >> 6 is all that exists afterwards:
//     gd_prep_kernel reads anything above ContigDev::pidx_last as n_reads, so blocks append without a tail to redo);
//   * the largest reference span of ANY record (out[1], atomicMax; unfiltered, so never below what the tile kernel
//     will measure on the kept ones): the first gd_compute starts with the right look-back instead of learning it.
//     Reads of more than 64 ops are not walked here (out[1] = INT_MAX: unknown -- the default look-back and its
//     verification take over; long-read data takes the long-read path, which measures spans itself);
//   * out[2] = pos[r1 - 1].
struct IndexJob {
    const int32_t*  pos;
    const uint32_t* off;
    const uint32_t* cigar;
    uint32_t* ridx;           // null: no index wanted
    uint32_t n_idx;           // its entries ((length >> 6) + 2); records placed past the contig's end write none
    uint32_t* out;            // [3]
    uint32_t* bad_out;        // where the check bits are ORed: out (gd_adopt_device) or the word of the committed blocks
    uint32_t r0, r1;
    uint32_t n_reads_total;   // records of the contig once this block is in (offset checks at its last record)
    uint32_t n_ops_total;
    int32_t  prev_pos;        // pos[r0 - 1]; r0 == 0: -1
    uint32_t check;
    uint32_t walk_ops;        // 0: spans not measured (out[1] untouched)
};

__global__ __launch_bounds__(256) void gd_index_records_kernel(IndexJob j)
{
    const uint32_t r = j.r0 + blockIdx.x * 256u + threadIdx.x;
    const bool in = r < j.r1;
    uint32_t bad = 0;
    int32_t ka = 0, kb = -1;
    int32_t span = 0;
    if (in) {
        const int32_t p = j.pos[r];
        const int32_t q = r > j.r0 ? j.pos[r - 1u] : j.prev_pos;
        if (q > p) bad |= 1u;
        if (p < 0) bad |= 4u;
        const uint32_t a = j.off[r], b = j.off[r + 1u];
        if (a > b || (r == 0u && a != 0u) || (r + 1u == j.n_reads_total && b > j.n_ops_total)) bad |= 2u;
        if (bad == 0u) { ka = (q >> 6) + 1; kb = p >> 6; kb = kb < (int32_t)j.n_idx ? kb : (int32_t)j.n_idx - 1; }        // (q = -1 in front of the first record: from entry 0)
        if (j.walk_ops && bad == 0u && b <= j.n_ops_total) {
            if (b - a > 64u) span = 0x7fffffff;
            else {
                uint32_t sp = 0;
                for (uint32_t o = a; o < b; ++o) {
                    const uint32_t op = j.cigar[o];
                    sp += ((0x18du >> (op & 15u)) & 1u) ? (op >> 4) : 0u;     // M D N = X consume the reference
                    sp = sp > 0x7fffffffu ? 0x7fffffffu : sp;
                }
                span = (int32_t)sp;
            }
        }
        if (r + 1u == j.r1) j.out[2] = (uint32_t)p;
    }
    if (j.ridx) {
        // nearly every read owns zero or one entry; a read behind a gap (a centromere: 10^5 entries) hands it to its wave
        int32_t k = ka;
        for (int n = 0; n < 2 && k <= kb; ++n, ++k) j.ridx[k] = r;
        unsigned long long m = __builtin_amdgcn_ballot_w64(k <= kb);
        const int lane = (int)(threadIdx.x & 63u);
        while (m) {
            const int l = __builtin_ctzll(m);
            m &= m - 1ull;
            const int32_t k0 = __shfl(k, l, 64), k1 = __shfl(kb, l, 64);
            const uint32_t rr = (uint32_t)__shfl((int)r, l, 64);
            for (int32_t x = k0 + lane; x <= k1; x += 64) j.ridx[x] = rr;
        }
    }
    if (j.check) {
        const uint32_t any = (__builtin_amdgcn_ballot_w64((bad & 1u) != 0u) != 0ull ? 1u : 0u) |
                             (__builtin_amdgcn_ballot_w64((bad & 2u) != 0u) != 0ull ? 2u : 0u) |
                             (__builtin_amdgcn_ballot_w64((bad & 4u) != 0u) != 0ull ? 4u : 0u);
        if (any != 0u && (threadIdx.x & 63u) == 0u) atomicOr(j.bad_out, any);
    }
    if (j.walk_ops) {
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) { const int32_t o = __shfl_xor(span, m, 64); span = o > span ? o : span; }
        // one address for the whole launch: ask first, most waves have nothing new to say
        if ((threadIdx.x & 63u) == 0u && span > 0 && span > (int32_t)__atomic_load_n(j.out + 1, __ATOMIC_RELAXED))
            atomicMax(reinterpret_cast<int32_t*>(j.out + 1), span);
    }
}

}  // namespace norm
}  // namespace gd
