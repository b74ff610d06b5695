// gd_inflate.hpp -- BGZF (RFC 1951 DEFLATE) decompression on the device.
//
// The BAM read that every `samtools depth` child of the reference performs
// (/root/reference/depth/depth.go:45) starts with inflating BGZF members; on the host that
// is the end-to-end limiter (scope iii).  A BGZF file is a sequence of
// independent <= 64 KiB deflate streams, so the file offers tens of thousands to millions
// of independent decode jobs: ONE LANE inflates ONE member, start to end.
//
// Round 3 rewrote the symbol loop as a per-lane STATE MACHINE with a fixed memory schedule.  The first
// kernel (one `if` per symbol kind, loads and stores inside the branches) left the compiler no way to
// count what is in flight, so every use of a loaded value became `s_waitcnt vmcnt(0)` -- the wave then
// paid a full store round trip per symbol (7 000 cycles per step measured, 72 % of wave cycles waiting).
// Here every iteration of every lane does the same things in the same order -- issue its two loads (the
// 16-byte chunk of a match in progress, the next 8 input bytes), decode one symbol while they are in
// flight, retire them, issue one 16-byte store -- so no load crosses the loop's back edge, the waits are
// counted (`vmcnt(1)`, then `vmcnt(0)` with nothing else outstanding) and a whole iteration of Huffman
// decoding hides each round trip:
//   * input: the branch-free 64-bit bit buffer (`buf |= next << cnt; p += (63 - cnt) >> 3; cnt |= 56`);
//     the word for the next refill is loaded right after this one, its address does not depend on what
//     the decode consumes;
//   * output: the last 16 bytes of the member live in registers (T).  Literals shift into T and are
//     stored 16 at a time (or when a match starts); a match is copied in 16-byte chunks, one per
//     iteration: a chunk is planned at the end of one iteration, its source loaded at the top of the next
//     and stored at that one's end, AFTER it has decoded its own symbol.  An iteration's store precedes the
//     next iteration's load, so a chunk may read what the previous chunk wrote (any distance >= 16);
//     distances below 16 take their first chunk
//     from T with two byte permutes (selectors from a table shared by the workgroup) and continue at
//     the next multiple of the distance that is >= 16;
//   * lanes with nothing to load or store issue nothing (the loads are waited for inside the iteration that
//     issued them, so the counts hold either way).
// Huffman decoding is canonical and branch free: the 15-bit peek is compared with the left-aligned end
// of every code length (15 compares against packed registers) and the symbol index is one add and two
// LDS reads ([entry][lane] tables: 408 bytes per lane).  Block headers (dynamic
// tables: ~300 serial code lengths per lane) are a divergent side path; lanes that reach one wait a
// few iterations so that the wave builds its tables together.  CRC32 is a second, converged kernel.
//
// Round 4: a lane's INPUT and OUTPUT go through LDS (608 bytes per lane in all: four workgroups per CU, measured faster
// than the six the tables alone allowed).  With ~100 000 members in flight every lane kept an input line, an output line
// and match sources alive in an L2 that holds a third of that: the 8 bytes the refill loaded per iteration came from a
// line evicted since the last iteration, and each 16-byte store was a partial write of a line that left L2 before its
// neighbours arrived -- FETCH_SIZE 126 GB and WRITE_SIZE 66 GB for 1.9 GB of input and 7.0 GB of output (108 k members).
//   * input: a 64-byte window per lane; a 16-byte slot is loaded once, when everything in it has been consumed, and the
//     refill reads three dwords of LDS;
//   * output: a 128-byte ring per lane, addressed like memory (offset + the member's address mod 64); what an iteration
//     produces is written there (five aligned dwords around the 16-byte register window), and the 64-byte block the
//     output has just passed leaves for memory as four 16-byte stores to one aligned block, back to back; the source of
//     a match is loaded from memory when it lies completely below what has been stored, else it lies completely inside
//     the ring's last 96 bytes and is read from there.
// 82.8 -> 54.3 ms for those 108 k members, WRITE_SIZE 66 -> 8.1 GB (1.15 x the output), FETCH_SIZE 126 -> 92 GB
// (profiles/r10u_…, r10v_…, r10w_inflate_output_ring.txt).
//
// Round 5: what bounds it (profiles/r12j_..., r12k_..., r12l_..., r12m_...).  The section counters of the measurement build
// (-DGD_MEASURE) put an iteration at ~4 400 cycles, 0.4 % of them waiting for the two loads: a wave issues one
// instruction per four cycles whatever its kind, and the ~750 instructions of an iteration (450 vector, 250 scalar, 50 LDS /
// memory) plus the dependent LDS look-ups of three Huffman decodes are those cycles -- at ONE wave per SIMD, because 608 bytes
// of LDS per member allow four waves per CU.  A kernel that split a member's work over two waves with the same LDS (a decoder
// wave that turns the stream into 32-bit tokens, a writer wave that owns T, the ring, the chunk loads and the block stores;
// a six-token queue per lane between them; commits 886121a, fdd3ff3) was bit exact in the emulation and on the GPU and NOT
// faster: 50.2 against 49.6 ms at deflate level 1, 69.4 against 72.2 at level 6 -- although its decoder alone runs at
// 40.0 ms and the pair without the writer's match-source loads at 41.5.  With issue slots to spare the next bound is right
// behind the first: the 16-byte source loads of ~100 000 members in flight each pull a line that no cache still holds
// (FETCH_SIZE 92 GB for 7 GB of output = 1.85 TB/s of scattered 64-byte fetches), and asking for them a whole step ahead
// changed nothing -- it is their NUMBER, not their latency.  One lane per member ends at ~145 GB/s of output on this part;
// the two-wave kernel was removed again.
#pragma once

// A probe for the host emulation (tests/emul/inflate_stats.cpp counts what the lanes of a wave do in every iteration);
// nothing on the device.
#ifndef GD_INFLATE_PROBE
#define GD_INFLATE_PROBE(what, value)
#endif

// MEASUREMENT BUILDS ONLY (-DGD_MEASURE; tools/r12_inflate_sections.sh): the cycles a wave spends in each section of
// the symbol loop, summed over all waves into g_inflate_sections (read back through gd_debug_inflate_sections).  The
// product is compiled without it.
#ifdef GD_MEASURE
__device__ unsigned long long g_inflate_sections[16];
#define GD_INF_T(k) do { const uint64_t t_ = __builtin_readcyclecounter(); tsum[k] += t_ - tlast; tlast = t_; } while (0)
#else
#define GD_INF_T(k)
#endif

namespace gd {

struct InflateJob {
    const uint8_t* comp;           // the compressed bytes (readable up to 64 bytes past the last member)
    const uint64_t* in_off;        // [n] offset of each member's deflate payload in comp
    const uint32_t* in_len;        // [n] payload bytes
    const uint64_t* out_off;       // [n] offset of the member's data in out
    const uint32_t* out_len;       // [n] ISIZE
    const uint32_t* crc;           // [n] CRC32 of the member's data (gzip trailer), or nullptr: not checked
    uint8_t* out;                  // (readable up to 64 bytes past the last member)
    uint32_t* status;              // [n] 0 ok, else an error code (18: CRC32 mismatch)
    uint32_t n;
    uint32_t only_status;          // 0: every member.  Else the lane-per-member kernel inflates only the members whose status word
                                   // holds this value (WV_FALLBACK: what the workgroup-per-member kernel left to it) and leaves the rest alone
};

constexpr size_t INF_SLACK = 256;  // bytes the inflate buffers are allocated beyond their contents

constexpr int INF_LANES = 64;      // lanes (= members in flight) per workgroup
constexpr int INF_MAXL = 288, INF_MAXD = 30;
// LDS of a workgroup: per-lane tables in an [entry][lane] layout (the 64 lanes of the wave read the same
// entry of 64 tables without bank conflicts when they are in step)
constexpr int INF_LDN = 0;                               // u32 [15][64]: lit/len, per code length: index delta | (first length symbol index << 16)
constexpr int INF_DDN = INF_LDN + 15 * 64 * 4;           // u16 [15][64]: distance, per code length: index delta
constexpr int INF_LSYM = INF_DDN + 15 * 64 * 2;          // u8 [288][64]: low 8 bits of the lit/len symbols in code order
constexpr int INF_DSYM = INF_LSYM + INF_MAXL * 64;       // u8 [30][64]
constexpr int INF_PERM = INF_DSYM + INF_MAXD * 64;       // u32 [16][8]: byte-permute selectors of a period-d chunk (shared)
constexpr int INF_INWIN = INF_PERM + 16 * 32;            // u32 [16][64]: every lane's 64-byte window of its member's input
constexpr int INF_RING = INF_INWIN + 16 * 64 * 4;        // u32 [32][64]: every lane's last 128 bytes of output
constexpr int INF_LDS_BYTES = INF_RING + 32 * 64 * 4;    // 38 912: four workgroups per CU

typedef uint32_t inf_v4 __attribute__((ext_vector_type(4)));

// Canonical Huffman tables from code lengths.  A code of length l and value v (MSB first) is decoded from the
// 15-bit peek x (first stream bit = MSB) as: l = 1 + #{k : x >= E_k}, E_k = (first_k + count_k) << (15 - k)
// (non-decreasing); index in code order = (x >> (15 - l)) + (offset_l - first_l).  The ends go to E (two per
// register), the deltas to LDS.  LIT: a lit/len symbol needs nine bits and a byte is stored -- within one length
// the canonical order is ascending, so indexes from `offset_l + #literals of length l` on are 256 + the byte.
// false: over-subscribed lengths.
template <bool LIT>
__device__ __forceinline__ bool inf_build(const uint8_t* lens, int n, uint8_t* s_tbl, int lane, uint32_t (&E)[8])
{
    uint32_t* const t32 = reinterpret_cast<uint32_t*>(s_tbl + INF_LDN) + lane;
    uint16_t* const t16 = reinterpret_cast<uint16_t*>(s_tbl + INF_DDN) + lane;
    uint8_t* const sym = s_tbl + (LIT ? INF_LSYM : INF_DSYM) + lane;
    auto rd = [&](int l) -> uint32_t { return LIT ? t32[(l - 1) * 64] : (uint32_t)t16[(l - 1) * 64]; };
    auto wr = [&](int l, uint32_t v) { if (LIT) t32[(l - 1) * 64] = v; else t16[(l - 1) * 64] = (uint16_t)v; };
    for (int l = 1; l <= 15; ++l) wr(l, 0);
    for (int i = 0; i < n; ++i) {                        // codes per length (bits 0-8), of those literals (bits 9-17)
        const int l = lens[i];
        if (l) wr(l, rd(l) + 1u + ((LIT && i < 256) ? 512u : 0u));
    }
    int left = 1;
    uint32_t first = 0, offs = 0;
    bool ok = true;
    uint32_t fin[15];
#pragma unroll
    for (int k = 0; k < 8; ++k) E[k] = 0;
#pragma unroll
    for (int l = 1; l <= 15; ++l) {
        const uint32_t v = rd(l), c = v & 511u, nl = (v >> 9) & 511u;
        left = (left << 1) - (int)c;
        ok = ok && left >= 0;
        const uint32_t e = ((first + c) << (15 - l)) & 0xffffu;          // <= 32768 while ok
        E[(l - 1) >> 1] |= e << (((l - 1) & 1) * 16);
        const uint32_t delta = (offs - first) & 0xffffu;
        fin[l - 1] = LIT ? delta | ((offs + nl) << 16) : delta;
        wr(l, offs);
        offs += c;
        first = (first + c) << 1;
    }
    if (!ok) return false;
    for (int i = 0; i < n; ++i) {
        const int l = lens[i];
        if (l) {
            const uint32_t pos = rd(l);
            wr(l, pos + 1u);
            sym[pos * 64] = (uint8_t)i;
        }
    }
#pragma unroll
    for (int l = 1; l <= 15; ++l) wr(l, fin[l - 1]);
    return true;
}

// One symbol from the 15-bit peek x.  len: its code length (1..15); bad: x is no code of this table.
template <bool LIT>
__device__ __forceinline__ uint32_t inf_decode(uint32_t x, const uint32_t (&E)[8], const uint8_t* s_tbl, int lane, uint32_t& len,
                                               bool& bad)
{
    uint32_t n = 1;
#pragma unroll
    for (int l = 1; l <= 15; ++l) {
        const uint32_t e = (l - 1) & 1 ? E[(l - 1) >> 1] >> 16 : E[(l - 1) >> 1] & 0xffffu;
        n += x >= e ? 1u : 0u;
    }
    bad = n > 15u;
    n = n > 15u ? 15u : n;
    len = n;
    int delta;
    uint32_t thr = 0;
    if (LIT) {
        const uint32_t w = reinterpret_cast<const uint32_t*>(s_tbl + INF_LDN)[(n - 1) * 64 + lane];
        delta = (int)(int16_t)(w & 0xffffu);
        thr = w >> 16;
    } else {
        delta = (int)(int16_t)reinterpret_cast<const uint16_t*>(s_tbl + INF_DDN)[(n - 1) * 64 + lane];
    }
    int idx = (int)(x >> (15u - n)) + delta;
    idx = idx < 0 ? 0 : idx;
    idx = idx > (LIT ? INF_MAXL : INF_MAXD) - 1 ? (LIT ? INF_MAXL : INF_MAXD) - 1 : idx;
    uint32_t v = s_tbl[(LIT ? INF_LSYM : INF_DSYM) + idx * 64 + lane];
    if (LIT) v |= (uint32_t)idx >= thr ? 256u : 0u;
    return v;
}

__device__ __forceinline__ inf_v4 inf_load16(const uint8_t* p)
{
    inf_v4 v;
    __builtin_memcpy(&v, p, 16);
    return v;
}
// a match source is read once: it should not push the output lines that are still being filled out of L2
__device__ __forceinline__ inf_v4 inf_load16_stream(const uint8_t* p)
{
    uint32_t w[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) w[k] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(p + 4 * k));
    return inf_v4{w[0], w[1], w[2], w[3]};
}
__device__ __forceinline__ void inf_store16(uint8_t* p, inf_v4 v) { __builtin_memcpy(p, &v, 16); }
__device__ __forceinline__ uint64_t inf_load8(const uint8_t* p)
{
    uint64_t v;
    __builtin_memcpy(&v, p, 8);
    return v;
}

// The last 16 bytes of (t followed by the first n bytes of c), 0 <= n <= 16.
__device__ __forceinline__ inf_v4 inf_append(inf_v4 t, inf_v4 c, uint32_t n)
{
    uint32_t a0 = t.x, a1 = t.y, a2 = t.z, a3 = t.w, a4 = c.x, a5 = c.y, a6 = c.z, a7 = c.w;
    if (n & 16u) { a0 = a4; a1 = a5; a2 = a6; a3 = a7; }
    if (n & 8u) { a0 = a2; a1 = a3; a2 = a4; a3 = a5; a4 = a6; a5 = a7; }
    if (n & 4u) { a0 = a1; a1 = a2; a2 = a3; a3 = a4; a4 = a5; }
    const uint32_t r = n & 3u;
    inf_v4 o;
    o.x = __builtin_amdgcn_alignbyte(a1, a0, r);
    o.y = __builtin_amdgcn_alignbyte(a2, a1, r);
    o.z = __builtin_amdgcn_alignbyte(a3, a2, r);
    o.w = __builtin_amdgcn_alignbyte(a4, a3, r);
    return o;
}

// The last 20 bytes of (e t followed by the first n bytes of c), 0 <= n <= 16: e' (the oldest four) and t'.
__device__ __forceinline__ void inf_append5(uint32_t& e, inf_v4& t, inf_v4 c, uint32_t n)
{
    uint32_t a0 = e, a1 = t.x, a2 = t.y, a3 = t.z, a4 = t.w, a5 = c.x, a6 = c.y, a7 = c.z, a8 = c.w;
    if (n & 16u) { a0 = a4; a1 = a5; a2 = a6; a3 = a7; a4 = a8; }
    if (n & 8u) { a0 = a2; a1 = a3; a2 = a4; a3 = a5; a4 = a6; a5 = a7; a6 = a8; }
    if (n & 4u) { a0 = a1; a1 = a2; a2 = a3; a3 = a4; a4 = a5; a5 = a6; }
    const uint32_t r = n & 3u;
    e = __builtin_amdgcn_alignbyte(a1, a0, r);
    t.x = __builtin_amdgcn_alignbyte(a2, a1, r);
    t.y = __builtin_amdgcn_alignbyte(a3, a2, r);
    t.z = __builtin_amdgcn_alignbyte(a4, a3, r);
    t.w = __builtin_amdgcn_alignbyte(a5, a4, r);
}

__global__ __launch_bounds__(INF_LANES) void gd_inflate_kernel(InflateJob job)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_tbl[INF_LDS_BYTES];
    const int lane = threadIdx.x;
    // selectors of a chunk with period d, taken from the last d of 16 bytes: byte b of the chunk is byte
    // 16 - d + b % d; one permute reads bytes 0-7 (selector 0x0c: zero), a second one bytes 8-15
    for (int i = lane; i < 16 * 8; i += INF_LANES) {
        const int d = i >> 3, half = (i >> 2) & 1, j = i & 3;
        uint32_t w = 0;
        for (int t = 0; t < 4; ++t) {
            const int b = 4 * j + t, s = d ? 16 - d + b % d : 0;
            const uint32_t sel = half == 0 ? (s < 8 ? (uint32_t)s : 0x0cu) : (s >= 8 ? (uint32_t)(s - 8) : 0x0cu);
            w |= sel << (8 * t);
        }
        reinterpret_cast<uint32_t*>(s_tbl + INF_PERM)[i] = w;
    }
    __syncthreads();
    const uint32_t m = blockIdx.x * INF_LANES + threadIdx.x;
    const bool mine = m < job.n && (job.only_status == 0u || job.status[m] == job.only_status);
    if (job.only_status != 0u && __ballot(mine) == 0) return;   // (a wave with nothing left to it -- nearly all of them)
    const uint32_t mm = mine ? m : 0u;
    const uint8_t* const in_beg = job.comp + job.in_off[mm];
    const uint8_t* const in_end = in_beg + job.in_len[mm];
    uint8_t* const out = job.out + job.out_off[mm];
    const uint32_t olen = job.out_len[mm];

    enum : uint32_t { DECODE = 0, COPY = 1, HDR = 2, DONE = 3 };
    uint32_t mode = mine ? HDR : DONE;
    uint32_t err = 0;
    uint32_t o = 0;                                        // bytes of the member produced (some still in T / C)
    uint32_t rem = 0, deff = 16;                           // a match in progress: bytes left, source distance (>= 16)
    uint32_t pend = 0;                                     // trailing bytes of T not stored yet (literals)
    bool cpend = false, csmall = false;                    // a chunk waits for its store: [co, co + cn), from cl (loaded) or cs (built from T)
    uint32_t co = 0, cn = 0;
    bool lastblk = false;
    inf_v4 T = {0, 0, 0, 0}, cs = {0, 0, 0, 0};
    uint32_t LE[8], DE[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) LE[k] = DE[k] = 0;
    const uint8_t* p = in_beg;
    uint64_t buf = 0;
    uint32_t cnt = 0;
    const uint8_t* ld_addr = out;                          // the chunk source the next iteration loads (or any readable address)
    // The member's INPUT goes through a 64-byte window in LDS ([dword][lane]: four 16-byte slots, a ring).  The refill of
    // the bit buffer used to load its 8 bytes from memory in every iteration -- from a line that advances by a byte or two
    // per iteration and, with ~100 000 members in flight (each with an input line, an output line and match sources:
    // more than L2 holds), was evicted in between: FETCH_SIZE 126 GB per 1.9 GB of input + 7 GB of output.  Now a
    // 16-byte slot is loaded ONCE, when everything in it has been consumed (about every tenth iteration), and the
    // refill reads three dwords of LDS.  win_hi: payload offset (a multiple of 16) where the window's newest slot ends;
    // it always reaches at least 32 bytes past the byte the bit buffer is at.
    uint32_t* const s_win = reinterpret_cast<uint32_t*>(s_tbl + INF_INWIN) + lane;
    uint32_t win_hi = 0;
    auto win_put = [&](uint32_t at, inf_v4 v) {            // payload bytes [at, at + 16), at % 16 == 0
        const uint32_t j = (at >> 2) & 12u;
        s_win[(j + 0u) * 64u] = v.x; s_win[(j + 1u) * 64u] = v.y; s_win[(j + 2u) * 64u] = v.z; s_win[(j + 3u) * 64u] = v.w;
    };
    auto win_restart = [&](uint32_t poff) {                // (start of the member; behind a block header) four slots from poff's on
        const uint32_t b = poff & ~15u;
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k) win_put(b + 16u * k, inf_load16(in_beg + b + 16u * k));
        win_hi = b + 64u;
    };
    win_restart(0);
    // The member's OUTPUT goes through a 128-byte ring in LDS and leaves for memory in whole 64-byte blocks, aligned in
    // MEMORY (ring and blocks are addressed by ao = the output offset + the low six bits of the member's address): a
    // lane's 16-byte pieces at scattered addresses -- the chunk of a match, the window of the last 16 literals -- were
    // each a partial write of a line that ~100 000 members in flight kept evicting from L2 before its neighbours
    // arrived (WRITE_SIZE 9.4 x the output at six workgroups per CU, 1.7 x at two).  fl: everything below it has been
    // stored; a block is stored as soon as the output has passed its end, so at most 79 bytes are ever only in the ring --
    // and the source of a match that is not completely below fl lies completely inside the ring's last 96 bytes.
    uint32_t* const s_ring = reinterpret_cast<uint32_t*>(s_tbl + INF_RING) + lane;
    const uint32_t obase = (uint32_t)(reinterpret_cast<uintptr_t>(out) & 63u);
    uint8_t* const out_al = out - obase;                   // 64-byte aligned: block [fl, fl + 64) lives at out_al + fl
    uint32_t fl = 0;
    uint32_t E0 = 0;                                       // the four bytes before T: bytes [o - 20, o - 16)
#ifdef GD_MEASURE
    uint64_t tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter(), titer = 0;
#endif
    auto ring_byte = [&](uint32_t a) -> uint32_t { return (s_ring[((a >> 2) & 31u) * 64u] >> (8u * (a & 3u))) & 0xffu; };
    // the bytes [lo, hi) of the ring to memory, one by one (a member's first block when the member starts inside it, its
    // last bytes, what is pending when a stored block starts)
    auto ring_bytes_out = [&](uint32_t lo, uint32_t hi) { for (uint32_t a = lo; a < hi; ++a) out_al[a] = (uint8_t)ring_byte(a); };

    for (uint32_t it = 0;; ++it) {
        const uint64_t live = __ballot(mode != DONE);
        if (live == 0) break;
        // A backstop, not a budget: a member is at most 65 536 output bytes (one iteration each at least, 16 per chunk of a
        // match) plus its block headers -- a 64 KB payload of nothing but empty stored / fixed blocks is ~50 000 of them,
        // each of which may wait up to 32 iterations for the other lanes: 2^22 iterations cover that; a lane that runs off
        // its input ends long before (err 1).
        if (it >= (1u << 22)) {
            if (mode != DONE) err = 19;
            break;
        }
        // ---- (0) the iteration's two loads -- the source of the chunk the previous iteration planned (cl; from_mem: it is
        //      completely below fl, else it lies inside the ring), the 16 input bytes behind the window (in16, want_in) --
        //      issued before anything else (but the loop's exits: a load that a path around the body leaves pending is waited
        //      for at the top of the loop): behind the stores of the iteration before (a chunk may read what the previous
        //      chunk wrote), in flight across the loop's bookkeeping, a block header and the decode.  Round 5: they used to
        //      follow the header path (51.3 -> 48.8 ms on 108 k members, profiles/r12j_...) ----
        inf_v4 cl = {0, 0, 0, 0}, in16 = {0, 0, 0, 0};
        const bool cload = cpend && !csmall;
        const uint32_t sa = obase + co - deff;             // where the chunk's source begins (ao); it ends at or before co
        const bool from_mem = cload && sa + 16u <= fl;     // completely stored -- else completely inside the ring
        if (from_mem) cl = inf_load16_stream(ld_addr);
        // (the slot 64 bytes behind win_hi has been consumed; a lane that waits for its block header may ask too: the header
        // path drops the request when it restarts the window)
        bool want_in = mode != DONE && (uint32_t)(p - in_beg) + 48u >= win_hi;
        if (want_in) in16 = inf_load16(in_beg + win_hi);
        // ---- block header (a divergent side path; lanes wait for each other to build together) ----
        const uint64_t hm = __ballot(mode == HDR);
        GD_INFLATE_PROBE(0, mode);
        if (hm != 0 && (hm == live || __popcll(hm) >= 32 || (it & 31u) == 31u)) {
            GD_INFLATE_PROBE(1, mode == HDR);
            if (mode == HDR) {
                auto need = [&](uint32_t nb) {             // nb <= 32
                    if (cnt < nb) {
                        buf |= inf_load8(p) << cnt;
                        p += (63u - cnt) >> 3;
                        cnt |= 56u;
                    }
                };
                auto bits = [&](uint32_t nb) -> uint32_t { // nb <= 16
                    need(nb);
                    const uint32_t v = (uint32_t)buf & ((1u << nb) - 1u);
                    buf >>= nb;
                    cnt -= nb;
                    return v;
                };
                auto past_end = [&]() { return (int64_t)(p - in_beg) * 8 - (int64_t)cnt > (int64_t)(in_end - in_beg) * 8; };
                lastblk = bits(1) != 0;
                const uint32_t type = bits(2);
                uint8_t lens[INF_MAXL + INF_MAXD + 4];
                if (type == 0) {                           // stored: byte loop, then T is read back
                    const uint32_t drop = cnt & 7u;
                    buf >>= drop;
                    cnt -= drop;
                    const uint32_t len = bits(16), nlen = bits(16);
                    if ((len ^ 0xffffu) != nlen) err = 2;
                    else if (o + len > olen || past_end()) err = 3;
                    else {
                        // (a stored block's bytes go straight to memory: what is still only in the ring goes first, and the
                        // ring restarts behind the block -- with the bytes of the block's last, incomplete 64-byte block)
                        ring_bytes_out(fl > obase ? fl : obase, obase + o);
                        for (uint32_t k = 0; k < len && err == 0; ++k) {
                            out[o++] = (uint8_t)bits(8);
                            if (p > in_end + 16) err = 1;
                        }
                        uint32_t w[5] = {0, 0, 0, 0, 0};
                        for (uint32_t k = 0; k < 20 && k < o; ++k)
                            w[4 - (k >> 2)] |= (uint32_t)out[o - 1 - k] << (8 * (3 - (k & 3)));
                        E0 = w[0];
                        T = inf_v4{w[1], w[2], w[3], w[4]};
                        fl = (obase + o) & ~63u;
                        // (from sixteen bytes below fl: a later chunk whose 16-byte source straddles fl -- sa < fl < sa + 16 --
                        // is read from the ring, so the ring must hold [fl - 15, fl) of what this block put into memory)
                        for (uint32_t a = fl >= obase + 16u ? fl - 16u : obase; a < obase + o; ++a) {
                            uint32_t& d = s_ring[((a >> 2) & 31u) * 64u];
                            d = (d & ~(0xffu << (8u * (a & 3u)))) | ((uint32_t)out_al[a] << (8u * (a & 3u)));
                        }
                        if (err == 0 && past_end()) err = 1;
                    }
                    // the next header follows (or the member ends)
                    if (err == 0 && lastblk) mode = DONE;
                } else if (type == 3) {
                    err = 4;
                } else {
                    int nlen = 288, ndist = 30;
                    if (type == 1) {                       // fixed codes
                        int s = 0;
                        for (; s < 144; ++s) lens[s] = 8;
                        for (; s < 256; ++s) lens[s] = 9;
                        for (; s < 280; ++s) lens[s] = 7;
                        for (; s < 288; ++s) lens[s] = 8;  // (286 and 287 complete the code; a stream must not use them)
                        for (s = 0; s < 30; ++s) lens[nlen + s] = 5;
                    } else {                               // dynamic codes
                        nlen = (int)bits(5) + 257;
                        ndist = (int)bits(5) + 1;
                        const int ncode = (int)bits(4) + 4;
                        if (nlen > 286 || ndist > INF_MAXD) err = 5;
                        static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
                        uint8_t cl_len[19];
                        for (int k = 0; k < 19; ++k) cl_len[order[k]] = k < ncode ? (uint8_t)bits(3) : (uint8_t)0;
                        // the code-length code borrows the distance tables
                        if (err == 0 && !inf_build<false>(cl_len, 19, s_tbl, lane, DE)) err = 6;
                        int idx = 0;
                        while (err == 0 && idx < nlen + ndist) {
                            need(32);
                            uint32_t l = 0;
                            bool bad = false;
                            const uint32_t sym = inf_decode<false>(__brev((uint32_t)buf) >> 17, DE, s_tbl, lane, l, bad);
                            if (bad || sym > 18u) { err = 7; break; }
                            buf >>= l;
                            cnt -= l;
                            if (sym < 16u) { lens[idx++] = (uint8_t)sym; continue; }
                            uint32_t prev = 0, rep;
                            if (sym == 16u) {
                                if (idx == 0) { err = 8; break; }
                                prev = lens[idx - 1];
                                rep = 3u + bits(2);
                            } else if (sym == 17u) rep = 3u + bits(3);
                            else rep = 11u + bits(7);
                            if (idx + (int)rep > nlen + ndist) { err = 9; break; }
                            while (rep--) lens[idx++] = (uint8_t)prev;
                            if (p > in_end + 16) err = 1;
                        }
                        if (err == 0 && lens[256] == 0) err = 10;
                    }
                    if (err == 0 && past_end()) err = 1;
                    if (err == 0 && !inf_build<true>(lens, nlen, s_tbl, lane, LE)) err = 11;
                    if (err == 0 && !inf_build<false>(lens + nlen, ndist, s_tbl, lane, DE)) err = 12;
                    if (err == 0) mode = DECODE;
                }
                if (err != 0) { mode = DONE; p = in_beg; }
                if (mode == DONE && err == 0 && o != olen) err = 17;
                need(56);                                  // what the decode below may consume
                win_restart((uint32_t)(p - in_beg));       // the header was read past the window
                want_in = false;                           // (16 bytes asked for before the header belong behind the OLD window)
            }
        }
        GD_INF_T(0);

        // ---- (1) what the loads issued at (0) do not cover: a chunk whose source lies in the ring, the window's dwords ----
        GD_INFLATE_PROBE(2, from_mem ? 1u : (cload ? 2u : 0u));
        GD_INFLATE_PROBE(4, from_mem ? sa : 0xffffffffu);
        GD_INFLATE_PROBE(5, from_mem ? deff : 0u);
        inf_v4 cr = {0, 0, 0, 0};
        if (cload && !from_mem) {
            const uint32_t rj = sa >> 2, rs = sa & 3u;
            const uint32_t r0 = s_ring[((rj + 0u) & 31u) * 64u], r1 = s_ring[((rj + 1u) & 31u) * 64u], r2 = s_ring[((rj + 2u) & 31u) * 64u],
                           r3 = s_ring[((rj + 3u) & 31u) * 64u], r4 = s_ring[((rj + 4u) & 31u) * 64u];
            cr.x = __builtin_amdgcn_alignbyte(r1, r0, rs);
            cr.y = __builtin_amdgcn_alignbyte(r2, r1, rs);
            cr.z = __builtin_amdgcn_alignbyte(r3, r2, rs);
            cr.w = __builtin_amdgcn_alignbyte(r4, r3, rs);
        }
        const uint32_t poff = (uint32_t)(p - in_beg);
        GD_INFLATE_PROBE(3, want_in);
        const uint32_t wj = poff >> 2;
        const uint32_t wd0 = s_win[((wj + 0u) & 15u) * 64u], wd1 = s_win[((wj + 1u) & 15u) * 64u], wd2 = s_win[((wj + 2u) & 15u) * 64u];

        // ---- (2) decode one symbol: everything a match needs, for every lane (no branches) ----
        const uint32_t lo = (uint32_t)buf;
        uint32_t l1 = 0, l2 = 0;
        bool bad1 = false, bad2 = false;
        const uint32_t sym = inf_decode<true>(__brev(lo) >> 17, LE, s_tbl, lane, l1, bad1);
        const uint32_t ls = sym - 257u;                    // length symbols 257..285
        uint32_t e1 = ls < 8u || ls >= 28u ? 0u : (ls >> 2) - 1u;
        uint32_t mlen = ls < 8u ? 3u + ls : ls == 28u ? 258u : 3u + ((4u + (ls & 3u)) << e1) + ((lo >> l1) & ((1u << e1) - 1u));
        const uint32_t used1 = l1 + e1;                    // <= 20
        const uint32_t lo2 = (uint32_t)(buf >> used1);
        const uint32_t x2 = __brev(lo2) >> 17;
        const uint32_t ds = inf_decode<false>(x2, DE, s_tbl, lane, l2, bad2);
        // ... and, should `sym` be a literal, the symbol behind it (the same bits: a literal has no extra bits): a second
        // literal goes out in the same iteration.  A stream of literals -- packed bases, and what libdeflate's parser makes
        // of short runs that zlib codes as matches -- is one iteration per BYTE otherwise: on the members of a
        // libdeflate-written BAM 19 200 iterations per 64 KB member against 13 900 for zlib's, before this.
        uint32_t l3 = 0;
        bool bad3 = false;
        const uint32_t sym2 = inf_decode<true>(x2, LE, s_tbl, lane, l3, bad3);
        const uint32_t e2 = ds < 4u ? 0u : (ds >> 1) - 1u;
        const uint32_t mdist = ds < 4u ? 1u + ds : 1u + ((2u + (ds & 1u)) << e2) + ((lo2 >> l2) & ((1u << e2) - 1u));
        const uint32_t used2 = used1 + l2 + e2;            // <= 48
        GD_INF_T(1);
        // Both loads are waited for HERE, by every lane: their uses below sit in branches (a lane without a chunk to append,
        // without a slot to fill, skips them), and a load the compiler cannot prove finished on every path costs a
        // `s_waitcnt vmcnt(0)` at the top of the next iteration -- in front of that iteration's loads, behind the block
        // stores of this one.  (gfx9 encoding: vmcnt 0, expcnt and lgkmcnt untouched.)
        __builtin_amdgcn_s_waitcnt(0x0F70);
        GD_INF_T(2);

        // ---- (3) the chunk loaded (or built) in the previous iteration goes into T ----
        const bool cp = cpend;
        const inf_v4 c = csmall ? cs : (from_mem ? cl : cr);
        if (cp) inf_append5(E0, T, c, cn);
        cpend = false;

        // ---- (4) this iteration's symbol ----
        bool flush = false;
        bool start = false;
        if (mode == DECODE) {
            if (bad1) { err = 13; mode = DONE; }
            else if (sym < 256u) {
                if (o >= olen) { err = 3; mode = DONE; }
                else {
                    // (not on top of a chunk appended in this iteration: the ring write below carries seventeen new bytes at most)
                    const bool two = !bad3 && sym2 < 256u && o + 2u <= olen && !cp;
                    const uint32_t nb = two ? 2u : 1u;
                    E0 = __builtin_amdgcn_alignbyte(T.x, E0, nb);
                    T.x = __builtin_amdgcn_alignbyte(T.y, T.x, nb);
                    T.y = __builtin_amdgcn_alignbyte(T.z, T.y, nb);
                    T.z = __builtin_amdgcn_alignbyte(T.w, T.z, nb);
                    T.w = two ? (T.w >> 16) | (sym << 16) | (sym2 << 24) : (T.w >> 8) | (sym << 24);
                    o += nb;
                    pend += nb;
                    flush = pend >= 15u;                   // (E0 and T hold twenty bytes: a pair on top of fourteen pending ones fits)
                    const uint32_t used = two ? l1 + l3 : l1;   // <= 30
                    buf >>= used;
                    cnt -= used;
                }
            } else if (sym == 256u) {
                buf >>= l1;
                cnt -= l1;
                flush = pend != 0u;
                mode = lastblk ? DONE : HDR;
                if (lastblk) {
                    if (o != olen) err = 17;
                    else if ((int64_t)(p - in_beg) * 8 - (int64_t)cnt > (int64_t)(in_end - in_beg) * 8) err = 1;
                }
            } else {
                if (ls >= 29u) { err = 14; mode = DONE; }
                else if (bad2 || ds >= 30u) { err = 15; mode = DONE; }
                else if (mdist > o || o + mlen > olen) { err = 16; mode = DONE; }
                else {
                    buf >>= used2;
                    cnt -= used2;
                    flush = pend != 0u;
                    start = true;
                    mode = COPY;
                    rem = mlen;
                }
            }
            if (mode == DONE) p = in_beg;
        }
        GD_INF_T(3);
        // ---- (6) refill from the word loaded at the top (before the store: nothing else is in flight then) ----
        if (want_in) { win_put(win_hi, in16); win_hi += 16u; }   // (the slot's first byte is >= 32 bytes ahead: nobody reads it yet)
        if (mode != DONE) {
            const uint32_t sh = poff & 3u;
            const uint64_t nw = (uint64_t)__builtin_amdgcn_alignbyte(wd1, wd0, sh) | ((uint64_t)__builtin_amdgcn_alignbyte(wd2, wd1, sh) << 32);
            buf |= nw << cnt;
            p += (63u - cnt) >> 3;
            cnt |= 56u;
            if (p > in_end + 16) { err = 1; mode = DONE; p = in_beg; }   // ran off the member's input
        }

        GD_INF_T(4);
        // what this iteration produced goes into the ring: the 20 bytes [o - 20, o) (E0, T) as five aligned dwords from the
        // dword that holds byte o - 20 on -- up to three bytes more than T needs on either side: in front bytes that are
        // there already, behind bytes that the next write replaces before anything reads them
        if (flush || cp) {
            const uint32_t ao = obase + o;
            const uint32_t sh = (4u - (ao & 3u)) & 3u;
            const uint32_t jw = (ao - 20u + sh) >> 2;
            s_ring[((jw + 0u) & 31u) * 64u] = __builtin_amdgcn_alignbyte(T.x, E0, sh);
            s_ring[((jw + 1u) & 31u) * 64u] = __builtin_amdgcn_alignbyte(T.y, T.x, sh);
            s_ring[((jw + 2u) & 31u) * 64u] = __builtin_amdgcn_alignbyte(T.z, T.y, sh);
            s_ring[((jw + 3u) & 31u) * 64u] = __builtin_amdgcn_alignbyte(T.w, T.z, sh);
            s_ring[((jw + 4u) & 31u) * 64u] = __builtin_amdgcn_alignbyte(0u, T.w, sh);
            // ... and the 64-byte block the output has just passed the end of leaves for memory: four 16-byte stores to ONE
            // aligned 64-byte block, back to back (at most one block per iteration: the output grows by 16 bytes at most)
            if (ao >= fl + 64u) {
                const uint32_t j0 = (fl >> 2) & 31u;       // 0 or 16
                if (fl >= obase) {
#pragma unroll
                    for (uint32_t q = 0; q < 4u; ++q) {
                        inf_v4 v;
                        v.x = s_ring[(j0 + 4u * q + 0u) * 64u]; v.y = s_ring[(j0 + 4u * q + 1u) * 64u];
                        v.z = s_ring[(j0 + 4u * q + 2u) * 64u]; v.w = s_ring[(j0 + 4u * q + 3u) * 64u];
                        inf_store16(out_al + fl + 16u * q, v);
                    }
                } else {
                    ring_bytes_out(obase, fl + 64u);       // the member starts inside this block: the bytes in front of it are another member's
                }
                fl += 64u;
            }
        }
        if (flush) pend = 0;
        GD_INF_T(5);

        // ---- (5) a match in progress: its next chunk, loaded (next iteration) after the store above ----
        csmall = false;
        if (mode == COPY) {
            const uint32_t n = rem < 16u ? rem : 16u;
            if (start && mdist < 16u) {
                // period mdist, from the last mdist bytes of T
                const inf_v4 s0 = *reinterpret_cast<const inf_v4*>(s_tbl + INF_PERM + mdist * 32);
                const inf_v4 s1 = *reinterpret_cast<const inf_v4*>(s_tbl + INF_PERM + mdist * 32 + 16);
                cs.x = __builtin_amdgcn_perm(T.y, T.x, s0.x) | __builtin_amdgcn_perm(T.w, T.z, s1.x);
                cs.y = __builtin_amdgcn_perm(T.y, T.x, s0.y) | __builtin_amdgcn_perm(T.w, T.z, s1.y);
                cs.z = __builtin_amdgcn_perm(T.y, T.x, s0.z) | __builtin_amdgcn_perm(T.w, T.z, s1.z);
                cs.w = __builtin_amdgcn_perm(T.y, T.x, s0.w) | __builtin_amdgcn_perm(T.w, T.z, s1.w);
                csmall = true;
                // the following chunks repeat with the next multiple of the period that is >= 16
                deff = 16u + (uint32_t)((0xECA8642052402000ull >> (4u * mdist)) & 15u);
            } else {
                if (start) deff = mdist;
                ld_addr = out + (o - deff);
            }
            co = o;
            cn = n;
            o += n;
            rem -= n;
            cpend = true;
            if (rem == 0u) mode = DECODE;
        }
        GD_INF_T(6);
#ifdef GD_MEASURE
        ++titer;
#endif
    }
#ifdef GD_MEASURE
    if (lane == 0) {
        for (int k = 0; k < 7; ++k) atomicAdd(&::g_inflate_sections[k], (unsigned long long)tsum[k]);
        atomicAdd(&::g_inflate_sections[7], (unsigned long long)titer);
    }
#endif
    // the member's last bytes: what never completed a 64-byte block
    if (mine && err == 0u) ring_bytes_out(fl > obase ? fl : obase, obase + olen);
    if (mine) job.status[m] = err;
}

// CRC32 of every member that inflated cleanly against its gzip trailer: one lane per member, all lanes of a wave
// in step.  Slicing-by-8: eight bytes per step through eight tables -- eight INDEPENDENT look-ups instead of a chain
// of dependent ones (a lane's CRC is a latency chain: the byte-at-a-time form took a quarter of the inflate's time).
__global__ __launch_bounds__(256) void gd_inflate_crc_kernel(InflateJob job)
{
    __shared__ uint32_t s_crc[8][256];                     // CRC-32 (IEEE 802.3, reflected): s_crc[k][b] = b followed by k zero bytes
    {
        uint32_t c = threadIdx.x;
        for (int k = 0; k < 8; ++k) c = (c >> 1) ^ (0xedb88320u & (0u - (c & 1u)));
        s_crc[0][threadIdx.x] = c;
    }
    __syncthreads();
    {
        uint32_t c = s_crc[0][threadIdx.x];
        for (int k = 1; k < 8; ++k) {
            c = (c >> 8) ^ s_crc[0][c & 0xffu];
            s_crc[k][threadIdx.x] = c;
        }
    }
    __syncthreads();
    const uint32_t m = blockIdx.x * 256u + threadIdx.x;
    if (m >= job.n || job.status[m] != 0) return;
    const uint8_t* const out = job.out + job.out_off[m];
    const uint32_t olen = job.out_len[m];
    uint32_t c = 0xffffffffu, k = 0;
    for (; k + 8 <= olen; k += 8) {
        uint32_t w0, w1;
        __builtin_memcpy(&w0, out + k, 4);
        __builtin_memcpy(&w1, out + k + 4, 4);
        w0 ^= c;
        c = s_crc[7][w0 & 0xffu] ^ s_crc[6][(w0 >> 8) & 0xffu] ^ s_crc[5][(w0 >> 16) & 0xffu] ^ s_crc[4][w0 >> 24] ^
            s_crc[3][w1 & 0xffu] ^ s_crc[2][(w1 >> 8) & 0xffu] ^ s_crc[1][(w1 >> 16) & 0xffu] ^ s_crc[0][w1 >> 24];
    }
    for (; k < olen; ++k) c = s_crc[0][(c ^ out[k]) & 0xffu] ^ (c >> 8);
    if ((c ^ 0xffffffffu) != job.crc[m]) job.status[m] = 18;
}

// The same check with ONE WAVE PER MEMBER, every byte fetched ONCE (round 5).  With a lane per member every lane streams
// through its own 64 KB: the lines a workgroup's 256 lanes are in the middle of stay neither in L1 nor, with ~100 000 members in
// flight, in L2 -- each 8-byte step fetches its line again; that kernel took a third of the inflate kernel's own time.  Round 4's
// first wave-per-member form gave lane l the 1 KB slice that ends (63 - l) KB before the member's end: 3.5 x faster (13.7 ms
// against 48 ms for 108 k members, profiles/r12a_*), but a wave instruction still touched 64 different lines and FETCH_SIZE
// said 102 GB for those members' 7 GB -- a line came back from memory for every 8-byte step that used it.  Here a wave
// instruction reads ONE KILOBYTE OF CONTIGUOUS MEMORY: the member is cut into rows of 1 KB aligned to its END, lane l takes
// the 16 bytes at l * 16 of every row (one 16-byte load), so a lane's bytes are 16 of every 1024.  CRC is linear over GF(2):
// a lane carries its register over the 1008 bytes between two of its pieces with a FIXED operator (Z^1008 as four byte
// tables: four look-ups), runs its 16 bytes through two slicing-by-8 steps, and at the end the member's register is the XOR of
// every lane's register advanced over the (63 - l) * 16 bytes behind its last piece -- the product of the 32 x 32 bit
// matrices Z^(16 * 2^j) for the set bits j of 63 - l.  Rows aligned to the END: the first row is simply shorter (bytes in
// front of the member are zeros to a register that starts at zero), and the CRC's initial value 0xffffffff is the same as
// inverting the member's first four bytes.  Workgroups of four waves walk the members in a grid-stride loop.
__global__ __launch_bounds__(256) void gd_inflate_crc_wave_kernel(InflateJob job)
{
    __shared__ uint32_t s_crc[8][256];                     // as above: s_crc[k][b] = b followed by k zero bytes
    __shared__ uint32_t s_adv[4][256];                     // Z^1008 by bytes: s_adv[k][b] = (b << 8k) advanced over 1008 zero bytes
    __shared__ uint32_t s_z[2][32];                        // squaring scratch: Z^(2^k) bytes, column b = image of bit b
    __shared__ uint32_t s_zk[6][32];                       // Z^(16 * 2^j), j = 0 .. 5
    __shared__ uint32_t s_z1008[32];
    __shared__ uint32_t s_red[4][64];
    {
        uint32_t c = threadIdx.x;
        for (int k = 0; k < 8; ++k) c = (c >> 1) ^ (0xedb88320u & (0u - (c & 1u)));
        s_crc[0][threadIdx.x] = c;
    }
    __syncthreads();
    {
        uint32_t c = s_crc[0][threadIdx.x];
        for (int k = 1; k < 8; ++k) {
            c = (c >> 8) ^ s_crc[0][c & 0xffu];
            s_crc[k][threadIdx.x] = c;
        }
        // one zero byte: c -> (c >> 8) ^ table[c & 0xff]
        if (threadIdx.x < 32) {
            const uint32_t v = 1u << threadIdx.x;
            s_z[0][threadIdx.x] = (v >> 8) ^ s_crc[0][v & 0xffu];
            s_z1008[threadIdx.x] = v;                        // the identity: the product of Z^(2^k) over the set bits k of 1008 grows here
        }
    }
    __syncthreads();
    auto mat_vec = [](const uint32_t* M, uint32_t v) { uint32_t y = 0; for (int b = 0; b < 32; ++b) y ^= M[b] & (0u - ((v >> b) & 1u)); return y; };
    for (int k = 0; k < 10; ++k) {                         // s_z[k & 1] = Z^(2^k)
        const uint32_t* const M = s_z[k & 1];
        if (threadIdx.x < 32) {
            if (k >= 4) s_zk[k - 4][threadIdx.x] = M[threadIdx.x];               // Z^16 .. Z^512
            if ((1008u >> k) & 1u) s_z1008[threadIdx.x] = mat_vec(M, s_z1008[threadIdx.x]);   // 1008 = 2^4 + 2^5 + 2^6 + 2^7 + 2^8 + 2^9
            s_z[(k + 1) & 1][threadIdx.x] = mat_vec(M, M[threadIdx.x]);          // its square
        }
        __syncthreads();
    }
    for (int k = 0; k < 4; ++k) s_adv[k][threadIdx.x] = mat_vec(s_z1008, (uint32_t)threadIdx.x << (8 * k));
    __syncthreads();
    const uint32_t w = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    for (uint32_t base = blockIdx.x * 4u; base < job.n; base += gridDim.x * 4u) {
        const uint32_t m = base + w;
        const bool live = m < job.n && job.status[m] == 0;
        uint32_t c = 0;
        if (live) {
            const uint8_t* const out = job.out + job.out_off[m];
            const int64_t n = (int64_t)job.out_len[m];
            if (n < 4) {                                     // (shorter than the initial value: byte by byte, one lane)
                if (lane == 63u) {
                    c = 0xffffffffu;
                    for (int64_t k = 0; k < n; ++k) c = s_crc[0][(c ^ out[k]) & 0xffu] ^ (c >> 8);
                }
            } else {
                const int64_t rows = (n + 1023) >> 10;
                // this lane's piece of row r: [n - (rows - r) * 1024 + 16 * lane, + 16)
                int64_t at = n - rows * 1024 + 16 * (int64_t)lane;
                for (int64_t r = 0; r < rows; ++r, at += 1024) {
                    if (at + 16 <= 0) continue;              // in front of the member: zeros to a register that is still zero
                    inf_v4 q = {0, 0, 0, 0};
                    if (at >= 0) q = inf_load16(out + at);
                    if (at < 4) {
                        // the piece that holds byte 0 (the bytes in front of it are another member's: loaded one by one), and
                        // the CRC's initial value: the member's first four bytes inverted (they may lie in two lanes' pieces)
                        uint32_t d[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                        for (int k = 0; k < 16; ++k) {
                            const int64_t pos = at + k;
                            if (at < 0 && pos >= 0) d[k >> 2] |= (uint32_t)out[pos] << (8 * (k & 3));
                            if (pos >= 0 && pos < 4) d[k >> 2] ^= 0xffu << (8 * (k & 3));
                        }
                        q = inf_v4{d[0], d[1], d[2], d[3]};
                    }
                    // over the 1008 bytes since this lane's last piece, then through the piece's 16 bytes
                    c = s_adv[0][c & 0xffu] ^ s_adv[1][(c >> 8) & 0xffu] ^ s_adv[2][(c >> 16) & 0xffu] ^ s_adv[3][c >> 24];
                    uint32_t w0 = q.x ^ c;
                    c = s_crc[7][w0 & 0xffu] ^ s_crc[6][(w0 >> 8) & 0xffu] ^ s_crc[5][(w0 >> 16) & 0xffu] ^ s_crc[4][w0 >> 24] ^
                        s_crc[3][q.y & 0xffu] ^ s_crc[2][(q.y >> 8) & 0xffu] ^ s_crc[1][(q.y >> 16) & 0xffu] ^ s_crc[0][q.y >> 24];
                    w0 = q.z ^ c;
                    c = s_crc[7][w0 & 0xffu] ^ s_crc[6][(w0 >> 8) & 0xffu] ^ s_crc[5][(w0 >> 16) & 0xffu] ^ s_crc[4][w0 >> 24] ^
                        s_crc[3][q.w & 0xffu] ^ s_crc[2][(q.w >> 8) & 0xffu] ^ s_crc[1][(q.w >> 16) & 0xffu] ^ s_crc[0][q.w >> 24];
                }
                // ... advanced over the (63 - lane) * 16 bytes that follow this lane's last piece
                const uint32_t after = 63u - lane;
                for (int j = 0; j < 6; ++j)
                    if ((after >> j) & 1u) c = mat_vec(s_zk[j], c);
            }
        }
        s_red[w][lane] = c;
        __syncthreads();
        if (lane < 8u) {
            uint32_t x = 0;
            for (int q = 0; q < 8; ++q) x ^= s_red[w][lane * 8u + q];
            s_red[w][lane * 8u] = x;                       // (only this lane reads or writes elements lane*8 .. lane*8+7 here)
        }
        __syncthreads();
        if (lane == 0u && live) {
            uint32_t x = 0;
            for (int q = 0; q < 8; ++q) x ^= s_red[w][q * 8];
            if ((x ^ 0xffffffffu) != job.crc[m]) job.status[m] = 18;
        }
        __syncthreads();                                    // s_red is written again in the next round
    }
}

constexpr bool INF_CRC_WAVE = true;                        // which of the two CRC kernels inflate_launch uses

}  // namespace gd

#include "gd_inflate_wave.hpp"

namespace gd {

#ifndef GD_INF_WAVE_NW
#define GD_INF_WAVE_NW 4
#endif
constexpr int INF_WAVE_NW = GD_INF_WAVE_NW;                             // waves per member of the workgroup-per-member kernel

// The kernels of one inflate on one stream: a workgroup per member (gd_inflate_wave.hpp), the lane-per-member kernel for what
// that one left (WV_FALLBACK), the CRC check.
// kernel (GD_OPT_INFLATE_KERNEL): 0 -- the lane-per-member kernel alone, the default; 1 -- the workgroup-per-member kernel first
// (round 6: a sixth of the memory traffic, but slower on an MI355X -- DESIGN.md 3.5), the lane-per-member kernel for what it left.
// lds_pad: bytes of LDS a lane-per-member workgroup claims on top of its own 38 KB -- an occupancy limiter for measurements.
inline void inflate_launch(const InflateJob& job_in, hipStream_t stream, unsigned lds_pad = 0, int kernel = 0)
{
    InflateJob job = job_in;
    if (job.n == 0) return;
    if (kernel == 1) {
        hipLaunchKernelGGL(gd_inflate_wave_kernel<INF_WAVE_NW>, dim3(job.n), dim3(64 * INF_WAVE_NW), 0, stream, job);
        job.only_status = WV_FALLBACK;
    } else {
        job.only_status = 0;
    }
    hipLaunchKernelGGL(gd_inflate_kernel, dim3((job.n + INF_LANES - 1) / INF_LANES), dim3(INF_LANES), lds_pad, stream, job);
    if (job.crc && INF_CRC_WAVE) {
        const unsigned groups = (job.n + 3u) / 4u;
        hipLaunchKernelGGL(gd_inflate_crc_wave_kernel, dim3(groups < 8192u ? groups : 8192u), dim3(256), 0, stream, job);
    } else if (job.crc) {
        hipLaunchKernelGGL(gd_inflate_crc_kernel, dim3((job.n + 255u) / 256u), dim3(256), 0, stream, job);
    }
}

}  // namespace gd
