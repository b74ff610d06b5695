// gd_inflate_wave.hpp -- BGZF (RFC 1951 DEFLATE) decompression, ONE WORKGROUP PER MEMBER, the member's output in LDS.
//
// The BAM read of every `samtools depth` child of the reference (/root/reference/depth/depth.go:45) starts with this.  The
// lane-per-member kernel (gd_inflate.hpp) ends at ~145 GB/s of output: its Huffman state machine is issue bound at one wave per
// SIMD, and the 16-byte match-source loads of ~100 000 members in flight each fetch a line no cache still holds (FETCH_SIZE 6-7 x
// the useful bytes).  Here a member's 64 KB of output never leave the CU before they are complete:
//
//   header    wave 0 reads the block header; a dynamic block's ~290 code-length symbols are decoded on the scalar unit (the
//             128-entry code-length table lives in two VGPRs and is read with v_readlane);
//   tables    all lanes: canonical codes -> a two-level look-up table per alphabet (10 / 8 root bits, 16-bit entries);
//   pass A    the block's remaining bits are cut into NL equal subsequences, one per lane.  Every lane decodes its subsequence
//             from its (wrong) boundary to the first symbol that starts in the next lane's subsequence; a DEFLATE stream
//             self-synchronises (median 7 symbols, tools/huffman_sync.py), so most crossings are right at once; lanes restart
//             from their left neighbour's crossing until nothing changes (2-3 passes, tools/inflate_wave_sim.cpp).  Result:
//             every lane's true start and the bytes its symbols produce -> a prefix sum gives its output offset;
//   pass B1   the same decode once more: literals go to their place in LDS, a match is cut into pieces of <= 16 bytes, each left
//             as a 3-byte token in the first bytes of its own destination, its start marked in a bitmap (1 bit per output byte);
//   pass B2   one wave resolves the pieces in OUTPUT order, 64 at a time: a piece is copied (one unaligned ds_read_b128, one
//             write) once its source holds no unresolved piece -- decided with two cheap sufficient rules (the source ends below
//             the first unresolved piece of the batch; the source lies in the literal gap right behind the previous piece).  A
//             piece of a match with a period below 16 is built from the bytes in front of the MATCH, so a long run is not a
//             chain of pieces (that alone halves the rounds: tools/inflate_wave_sim.cpp);
//   store     LDS -> memory in 16-byte stores, the only bytes this kernel writes.
//
// Anything this kernel does not handle (a table that does not fit, a match reaching in front of the member, a stream that does
// not end where it should, any invalid code) is NOT diagnosed here: the member's status becomes WV_FALLBACK and the lane-per-
// member kernel, launched behind this one for exactly those members, inflates or refuses it with its own codes.  So a status of
// zero from here must mean the bytes are right, and every doubt is a fallback.
#pragma once

namespace gd {

constexpr uint32_t WV_FALLBACK = 100u;                     // status: left to the lane-per-member kernel

constexpr int WV_RL = 10, WV_RD = 8;                       // root bits of the lit/len and the distance table
constexpr int WV_LSUB = 256, WV_DSUB = 160;                // second-level entries (checked when the tables are built)
constexpr int WV_OUT = 0;                                  // [16 zero bytes][65536][16]: the member's output
constexpr int WV_OUT_BYTES = 16 + 65536 + 16;
constexpr int WV_LIT = WV_OUT + WV_OUT_BYTES;              // u16 [1024 + WV_LSUB]
constexpr int WV_DIST = WV_LIT + 2 * ((1 << WV_RL) + WV_LSUB);   // u16 [256 + WV_DSUB]
constexpr int WV_MISC = WV_DIST + 2 * ((1 << WV_RD) + WV_DSUB);  // u32 [32]
constexpr int WV_LANE = WV_MISC + 128;                     // u32 cross[NL], cnt[NL]
constexpr int WV_X_BYTES = 8192 + 512;                     // region X: scratch of header / tables / scan, then the bitmap; selectors
template <int NW> struct WvLayout {
    static constexpr int NL = 64 * NW;
    static constexpr int X = WV_LANE + (8 * NL > 1408 ? 8 * NL : 1408);   // (pass B2 keeps 704 piece starts there)
    static constexpr int BYTES = X + WV_X_BYTES;
};
// region X before pass B1
constexpr int WX_HEAD = 0;                                 // 1088 bytes of the payload from the block's first byte on
constexpr int WX_LENS = 1088;                              // u8 [320]: code lengths, lit/len then distance
constexpr int WX_RANK = 1408;                              // u8 [6][64]: rank of a symbol among the equally long ones of its chunk
constexpr int WX_SORTL = 1792;                             // u16 [288]: lit/len symbols in code order
constexpr int WX_SORTD = 2368;                             // u16 [32]
constexpr int WX_CCNT = 2432;                              // u16 [6][16]: codes of each length per chunk of 64 symbols
constexpr int WX_BASE = 2624;                              // u16 [6][16]: ... in the chunks before
constexpr int WX_CNT = 2816;                               // u16 [2][16]: codes of each length (lit/len, distance)
constexpr int WX_OFFS = 2880;                              // u16 [2][16]: index of the first code of each length
constexpr int WX_DELTA = 2944;                             // i16 [2][16]: offs - first code
constexpr int WX_SCAN = 3072;                              // u32 [2][NL]
// region X from pass B1 on
constexpr int WX_BITMAP = 0;                               // u32 [2048]: bit p = a piece starts at output byte p
constexpr int WX_PERM = 8192;                              // u32 [16][8]: byte-permute selectors of a chunk with period d (all blocks)
enum : int { WM_FLAG0 = 0, WM_FLAG1, WM_TYPE, WM_FINAL, WM_NLEN, WM_NDIST, WM_HDREND, WM_ERR, WM_STORED, WM_EOB, WM_TOTAL, WM_ENDPOS, WM_FAIL };
enum : uint32_t { WS_RUN = 0, WS_CROSSED = 1, WS_EOB = 2, WS_BAD = 3, WS_INACTIVE = 4 };

#ifdef GD_EMUL_HOST
// (tests/emul: the lanes of a wave are fibers; emul_machine.hpp supplies these)
#define WV_READLANE(v, l) emul_readlane((uint32_t)(v), (uint32_t)(l))
#define WV_SHFL_UP(v, d) emul_shfl_up((uint32_t)(v), (uint32_t)(d))
#define WV_UNIFORM(v) (v)
#define WV_WAVE_SYNC() emul::wave_barrier()
#define WV_LDS_OR(p, v) (*(p) |= (v))
#define WV_LANE_IN(mask, lane) ((((uint64_t)(mask)) >> (lane)) & 1ull)
#else
#define WV_READLANE(v, l) ((uint32_t)__builtin_amdgcn_readlane((int)(v), (int)(l)))
#define WV_SHFL_UP(v, d) ((uint32_t)__shfl_up((int)(v), (unsigned)(d), 64))
#define WV_UNIFORM(v) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(v)))
// LDS operations of one wave execute in order: a lane reads what another lane of its wave wrote before -- the compiler must not
// move them across this point
#define WV_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
#define WV_LDS_OR(p, v) ((void)__hip_atomic_fetch_or((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))
// is this lane's bit set in a lane mask that lives in scalar registers: the mask becomes the execution mask, no vector compare
#define WV_LANE_IN(mask, lane) __builtin_amdgcn_inverse_ballot_w64((uint64_t)(mask))
#endif

__device__ __forceinline__ uint64_t wv_load8(const uint8_t* p)
{
    uint64_t v;
    __builtin_memcpy(&v, p, 8);
    return v;
}
__device__ __forceinline__ uint32_t wv_load4(const uint8_t* p)
{
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}

// Inclusive prefix sum over the 64 lanes of a wave.
__device__ __forceinline__ uint32_t wv_wave_incl_scan(uint32_t v, int lane)
{
#ifdef GD_EMUL_HOST
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = WV_SHFL_UP(v, d);
        if (lane >= d) v += up;
    }
    return v;
#else
    // row_shr within rows of 16 (bound_ctrl: zero comes in), then the row totals across rows through permlane-free broadcasts
    (void)lane;
    int x = (int)v;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true);   // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true);   // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true);   // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true);   // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, true);   // row_bcast:15 into rows 1 and 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, true);   // row_bcast:31 into rows 2 and 3
    return (uint32_t)x;
#endif
}

// Canonical code: E[l - 1] = end of the codes of length <= l, left-aligned in 15 bits (non-decreasing, <= 32768).  The code of
// the left-aligned 15-bit value x has length n = 1 + #{l : x >= E[l - 1]} (n > 15: none) and index (x >> (15 - n)) + delta[n].
struct WvCanon {
    uint32_t E[15];
    uint32_t lmax;                                         // the longest code
    bool over, complete;
};
__device__ __forceinline__ WvCanon wv_canon_setup(const uint16_t* cnt, uint16_t* offs, int16_t* delta, bool write)
{
    WvCanon c;
    int left = 1;
    uint32_t first = 0, o = 0;
    c.lmax = 0;
    c.over = false;
#pragma unroll
    for (int l = 1; l <= 15; ++l) {
        const uint32_t n = WV_UNIFORM(cnt[l]);
        left = (left << 1) - (int)n;
        if (left < 0) c.over = true;
        if (n) c.lmax = (uint32_t)l;
        const uint32_t e = (first + n) << (15 - l);
        c.E[l - 1] = e > 32768u ? 32768u : e;
        if (write) { offs[l] = (uint16_t)o; delta[l] = (int16_t)((int)o - (int)first); }
        o += n;
        first = (first + n) << 1;
    }
    c.complete = left == 0;
    return c;
}
__device__ __forceinline__ uint32_t wv_canon_len(const WvCanon& c, uint32_t x)
{
    uint32_t n = 1;
#pragma unroll
    for (int l = 0; l < 15; ++l) n += x >= c.E[l] ? 1u : 0u;
    return n;
}

// ---- table entries (16 bits) ----
// lit/len:   bits 0-3 code length; bit 4 = 0: literal, byte in bits 8-15
//            bit 4 = 1: bits 5-7 = t: 0-5 a length with t extra bits, base - 3 in bits 8-15; 6 end of block;
//                       7 special: code length 0 invalid, else second level: bits 0-3 = its index bits, bits 8-15 = offset / 2
// distance:  bits 0-3 code length; bit 4 = 0: bits 5-9 the distance symbol; bit 4 = 1 special as above (offset in bits 5-12)
__device__ __forceinline__ uint32_t wv_lit_entry(uint32_t sym, uint32_t len)
{
    if (sym < 256u) return (sym << 8) | len;
    if (sym == 256u) return (6u << 5) | 16u | len;
    if (sym > 285u) return (7u << 5) | 16u;                // 286, 287: not in a valid stream
    const uint32_t ls = sym - 257u;
    const uint32_t eb = ls < 8u || ls == 28u ? 0u : (ls >> 2) - 1u;
    const uint32_t base = ls < 8u ? 3u + ls : ls == 28u ? 258u : 3u + ((4u + (ls & 3u)) << eb);
    return ((base - 3u) << 8) | (eb << 5) | 16u | len;
}

// The second level of a table with R root bits: a root prefix P (R bits, MSB first) whose codes are longer than R gets
// 2^(maxlen(P) - R) entries, maxlen(P) = the longest code under P.  With B[L] = E[L - 1] >> (15 - R) (prefixes completely covered
// by codes of length <= L) the prefixes with maxlen L are [B[L - 1], B[L]), so every offset follows from the fifteen ends.
template <int R> struct WvSub {
    uint32_t B[16 - R];                                    // B[k] = prefixes covered by codes of length <= R + k
    uint32_t cum[16 - R];                                  // cum[k] = second-level entries of prefixes with maxlen <= R + k
};
template <int R> __device__ __forceinline__ WvSub<R> wv_sub_setup(const WvCanon& c)
{
    WvSub<R> s;
    s.B[0] = c.E[R - 1] >> (15 - R);
    s.cum[0] = 0;
#pragma unroll
    for (int k = 1; k <= 15 - R; ++k) {
        s.B[k] = c.E[R + k - 1] >> (15 - R);
        s.cum[k] = s.cum[k - 1] + ((s.B[k] - s.B[k - 1]) << k);
    }
    return s;
}

// One run of a lane over its subsequence: symbols from bit `pos` until one starts at or behind `bound`, the block ends, or the
// stream is invalid.
// WRITE = false (pass A): the payload is read from LDS -- `in` points into the output area, which nothing else uses yet -- one
// unaligned 8-byte read per symbol (>= 57 bits behind the position; a symbol takes 48 at most).
// WRITE = true (pass B1): literals go to their place, matches are left as pieces; the output area is being written, so the
// payload comes from memory: a two-word window in registers, the word behind it asked for at the top of EVERY iteration and
// taken at its bottom -- a loaded value that crosses the loop's back edge is waited for where the compiler copies it, i.e. at
// once (the lane-per-member kernel learnt that in round 3), and in lock step the wave pays the slowest lane's round trip in
// every iteration; asked for again and again, the word comes from L1 while the symbol is decoded.
struct WvRun { uint32_t pos, cnt, state; };

template <bool WRITE>
__device__ __forceinline__ WvRun wv_run(const uint8_t* in, uint32_t start, uint32_t bound, uint32_t endbits, bool go, uint8_t* sm, int xoff,
                                         uint32_t o, uint32_t& fail)
{
    const uint16_t* const lit = reinterpret_cast<const uint16_t*>(sm + WV_LIT);
    const uint16_t* const dis = reinterpret_cast<const uint16_t*>(sm + WV_DIST);
    uint8_t* const out = sm + WV_OUT + 16;
    uint32_t* const bitmap = reinterpret_cast<uint32_t*>(sm + xoff + WX_BITMAP);
    WvRun r;
    r.pos = start;
    r.cnt = 0;
    r.state = go ? WS_RUN : WS_INACTIVE;
    uint32_t qi = start >> 6;
    uint64_t q0 = 0, q1 = 0;
    if (WRITE && go) { q0 = wv_load8(in + 8u * qi); q1 = wv_load8(in + 8u * qi + 8u); }
    for (;;) {
        const bool act = r.state == WS_RUN;
        if (__ballot(act) == 0) break;
        uint64_t nx = 0;
        if (WRITE && act) nx = wv_load8(in + 8u * qi + 16u);
        // One symbol, every lane the same instructions: 32 bits at the position (w0: a lit/len code and its extra bits are
        // 20 at most), the lit/len entry, 32 bits behind code + extra bits (w1: a distance code and its extra bits, 28 at
        // most), the distance entry -- whether the symbol is a match or not; branches only where a second-level entry is
        // needed (a ballot: rare) and, in pass B1, where bytes are written.
        uint32_t c0, c1, c2 = 0, sh;
        if (WRITE) {
            const bool up = (r.pos & 32u) != 0u;
            c0 = up ? (uint32_t)(q0 >> 32) : (uint32_t)q0;
            c1 = up ? (uint32_t)q1 : (uint32_t)(q0 >> 32);
            c2 = up ? (uint32_t)(q1 >> 32) : (uint32_t)q1;
            sh = r.pos & 31u;
        } else {
            const uint64_t raw = act ? wv_load8(in + (r.pos >> 3)) : 0ull;
            c0 = (uint32_t)raw;
            c1 = (uint32_t)(raw >> 32);
            sh = r.pos & 7u;
        }
        const uint32_t w0 = __builtin_amdgcn_alignbit(c1, c0, sh);
        uint32_t e = lit[w0 & ((1u << WV_RL) - 1u)];
        {
            const bool two = act && (e & 0xf0u) == 0xf0u && (e & 15u) != 0u;
            if (__ballot(two)) { if (two) e = lit[(1u << WV_RL) + ((e >> 8) << 1) + __builtin_amdgcn_ubfe(w0, WV_RL, e & 15u)]; }
        }
        const uint32_t nb = e & 15u, t = (e >> 5) & 7u;
        const bool flag = (e & 16u) != 0u, ismatch = flag && t < 6u;
        const uint32_t used = nb + t;                      // (a literal: t = 0)
        const uint32_t mlen = 3u + (e >> 8) + __builtin_amdgcn_ubfe(w0, nb, t);
        uint32_t w1;
        if (WRITE) {
            const uint32_t s2 = sh + used;                 // <= 51
            w1 = s2 >= 32u ? __builtin_amdgcn_alignbit(c2, c1, s2) : __builtin_amdgcn_alignbit(c1, c0, s2);
        } else {
            w1 = __builtin_amdgcn_alignbit(c1, c0, sh + used);   // <= 27
        }
        uint32_t d = dis[w1 & ((1u << WV_RD) - 1u)];
        {
            const bool two = act && ismatch && (d & 16u) != 0u && (d & 15u) != 0u;
            if (__ballot(two)) { if (two) d = dis[(1u << WV_RD) + (d >> 5) + __builtin_amdgcn_ubfe(w1, WV_RD, d & 15u)]; }
        }
        const uint32_t nd = d & 15u, ds = (d >> 5) & 31u;
        const uint32_t de = ds < 4u ? 0u : (ds >> 1) - 1u;
        const bool bad = flag && (t == 7u || (ismatch && (d & 16u) != 0u));
        const bool eob = flag && t == 6u;
        if (act) {
            if (WRITE) {
                if (!flag) out[o] = (uint8_t)(e >> 8);
                else if (ismatch && !bad) {
                    const uint32_t mdist = ds < 4u ? 1u + ds : 1u + ((2u + (ds & 1u)) << de) + __builtin_amdgcn_ubfe(w1, nd, de);
                    if (mdist > o) { fail = 1u; }
                    else {
                        // pieces of <= 16 bytes, each a 3-byte token in its own first bytes: bit 0 = periodic;
                        // bits 1-4 length - 1; plain: bits 5-19 distance - 1 (>= 16); periodic (distance p < 16): bits 5-8 p,
                        // bits 9-16 c = how far behind the ANCHOR the piece starts -- the anchor is the start of the match
                        // (the piece is the pattern in front of the anchor repeated from phase 0: c is a multiple of p), or
                        // the piece itself (c = 0) for a tail that cannot start at phase 0
                        uint32_t rem = mlen, p = o;
                        if (mdist >= 16u) {
                            while (rem) {
                                const uint32_t n = rem > 16u ? (rem - 16u < 3u ? 13u : 16u) : rem;
                                const uint32_t tok = ((n - 1u) << 1) | ((mdist - 1u) << 5);
                                out[p] = (uint8_t)tok; out[p + 1u] = (uint8_t)(tok >> 8); out[p + 2u] = (uint8_t)(tok >> 16);
                                WV_LDS_OR(&bitmap[p >> 5], 1u << (p & 31u));
                                p += n; rem -= n;
                            }
                        } else {
                            // whole periods in 16 bytes: mdist * (16 / mdist)
                            const uint32_t lp = mdist >= 9u ? mdist : (uint32_t)((0xFDBEFEFF0ull >> (4u * mdist)) & 15ull) + 1u;
                            uint32_t c = 0;
                            bool offphase = false;
                            while (rem) {
                                uint32_t n = rem < lp ? rem : lp;
                                const uint32_t cc = offphase ? 0u : c;     // (a tail that does not start at phase 0: its own anchor)
                                const uint32_t left = rem - n;
                                if (left == 1u || left == 2u) {
                                    if (rem <= 16u) n = rem;           // the tail joins this piece
                                    else { n -= 3u - left; offphase = true; }   // ... or this piece leaves it three bytes
                                }
                                const uint32_t tok = 1u | ((n - 1u) << 1) | (mdist << 5) | (cc << 9);
                                out[p] = (uint8_t)tok; out[p + 1u] = (uint8_t)(tok >> 8); out[p + 2u] = (uint8_t)(tok >> 16);
                                WV_LDS_OR(&bitmap[p >> 5], 1u << (p & 31u));
                                p += n; rem -= n; c += n;
                            }
                        }
                    }
                }
            }
            const uint32_t inc = ismatch ? mlen : (flag ? 0u : 1u);
            r.pos += ismatch ? used + nd + de : nb;
            r.cnt += inc;
            o += inc;
            r.state = bad ? WS_BAD : eob ? (r.pos > endbits ? WS_BAD : WS_EOB) : (r.pos > endbits || r.cnt > 65536u) ? WS_BAD : r.pos >= bound ? WS_CROSSED : WS_RUN;
            if (WRITE && fail) r.state = WS_BAD;
        }
        if (WRITE) {
#ifndef GD_EMUL_HOST
            __builtin_amdgcn_s_waitcnt(0x0F70);            // vmcnt(0): the word asked for at the top (gfx9 encoding; lgkmcnt untouched)
#endif
            if (act && (r.pos >> 6) != qi) { q0 = q1; q1 = nx; ++qi; }
        }
    }
    return r;
}

// MEASUREMENT BUILDS ONLY (-DGD_MEASURE): the cycles a workgroup spends in each phase, summed into g_inflate_sections[8..15]
#ifdef GD_MEASURE
#define WV_T(k) do { const uint64_t t_ = __builtin_readcyclecounter(); wsum[k] += t_ - wlast; wlast = t_; } while (0)
// ... and inside pass B2 (g_inflate_b2): cycles of the bitmap expansion, of a batch's set-up, of its rounds; windows, batches, rounds
__device__ unsigned long long g_inflate_b2[8];
#define WV_B(k) do { const uint64_t t_ = __builtin_readcyclecounter(); bsum[k] += t_ - blast; blast = t_; } while (0)
#define WV_BN(k) (++bsum[k])
#else
#define WV_T(k)
#define WV_B(k)
#define WV_BN(k)
#endif

template <int NW>
__global__ __launch_bounds__(64 * NW, (NW + 1) / 2) void gd_inflate_wave_kernel(InflateJob job)
{
    constexpr int NL = 64 * NW;
    constexpr int XO = WvLayout<NW>::X;
    __shared__ __attribute__((aligned(16))) uint8_t sm[WvLayout<NW>::BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t m = blockIdx.x;
    const uint8_t* const in = job.comp + job.in_off[m];
    const uint32_t ilen = job.in_len[m], olen = job.out_len[m], endbits = ilen * 8u;
    uint8_t* const out = sm + WV_OUT + 16;
    uint32_t* const misc = reinterpret_cast<uint32_t*>(sm + WV_MISC);
    uint32_t* const l_cross = reinterpret_cast<uint32_t*>(sm + WV_LANE);
    uint32_t* const l_cnt = l_cross + NL;                  // bytes produced | state << 24
    uint8_t* const X = sm + XO;
    uint16_t* const lit = reinterpret_cast<uint16_t*>(sm + WV_LIT);
    uint16_t* const dis = reinterpret_cast<uint16_t*>(sm + WV_DIST);

    if (tid < 8) reinterpret_cast<uint32_t*>(sm + WV_OUT)[tid < 4 ? tid : (16 + 65536) / 4 + tid - 4] = 0;
    if (tid < 32) misc[tid] = 0;
    // selectors of a chunk with period d, taken from the last d of 16 bytes: byte b of the chunk is byte 16 - d + b % d; one
    // permute reads bytes 0-7 (selector 0x0c: zero), a second one bytes 8-15
    for (int i = tid; i < 16 * 8; i += NL) {
        const int d = i >> 3, half = (i >> 2) & 1, j = i & 3;
        uint32_t w = 0;
        for (int t = 0; t < 4; ++t) {
            const int b = 4 * j + t, sb = d ? 16 - d + b % d : 0;
            const uint32_t sel = half == 0 ? (sb < 8 ? (uint32_t)sb : 0x0cu) : (sb >= 8 ? (uint32_t)(sb - 8) : 0x0cu);
            w |= sel << (8 * t);
        }
        reinterpret_cast<uint32_t*>(X + WX_PERM)[i] = w;
    }
#ifdef GD_MEASURE
    uint64_t wsum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, wlast = __builtin_readcyclecounter();
    uint64_t bsum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, blast = 0;
#endif
    uint32_t bitpos = 0, opos = 0;
    bool fallback = olen > 65536u || ilen > (1u << 20);
    bool done = false;
    __syncthreads();

    uint32_t nblocks = 0;
    while (!fallback && !done) {
        // (a member of very many blocks -- tens of thousands of empty ones are valid -- pays this kernel's per-block set-up every
        // time: the other kernel's)
        if (++nblocks > 48u) { fallback = true; break; }
        // ================= the block header =================
        const uint32_t hb = bitpos >> 3;
        for (int i = tid; i < 1088 / 4; i += NL) {
            const uint32_t at = hb + 4u * (uint32_t)i;
            reinterpret_cast<uint32_t*>(X + WX_HEAD)[i] = at + 4u <= ilen + 64u ? wv_load4(in + at) : 0u;
        }
        __syncthreads();
        if (wave == 0) {
            const uint8_t* const head = X + WX_HEAD;
            uint8_t* const lens = X + WX_LENS;
            uint32_t hp = bitpos & 7u, err = 0, type, fin, nl = 0, nd = 0, stored = 0;
            auto peek = [&](uint32_t p) -> uint64_t {       // >= 57 bits from bit p of the staged bytes
                const uint64_t w = wv_load8(head + (p >> 3)) >> (p & 7u);
                return (uint64_t)WV_UNIFORM((uint32_t)w) | ((uint64_t)WV_UNIFORM((uint32_t)(w >> 32)) << 32);
            };
            {
                const uint32_t w = (uint32_t)peek(hp);
                fin = w & 1u;
                type = (w >> 1) & 3u;
                hp += 3u;
            }
            if (type == 0u) {
                hp = (hp + 7u) & ~7u;
                const uint32_t w = (uint32_t)peek(hp);
                stored = w & 0xffffu;
                if ((stored ^ 0xffffu) != (w >> 16)) err = 2;
                hp += 32u;
            } else if (type == 3u) {
                err = 4;
            } else if (type == 1u) {
                nl = 288; nd = 30;
                for (int s = lane; s < 320; s += 64) lens[s] = (uint8_t)(s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : s < 288 ? 8 : 5);
            } else {
                uint32_t w = (uint32_t)peek(hp);
                nl = (w & 31u) + 257u;
                nd = ((w >> 5) & 31u) + 1u;
                const uint32_t nc = ((w >> 10) & 15u) + 4u;
                hp += 14u;
                if (nl > 286u || nd > 30u) err = 5;
                uint64_t clb = peek(hp) & ((1ull << (3u * nc)) - 1ull);   // nc * 3 <= 57 bits
                hp += 3u * nc;
                // the code-length code: 19 symbols, up to 7 bits; entry r of its table (r = the next 7 bits of the stream) is
                // held by lane r & 63 in t0 (r < 64) or t1: symbol | length << 5, 0 = no code
                uint32_t cl[19];
                {
                    constexpr uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
#pragma unroll
                    for (int k = 0; k < 19; ++k) cl[k] = 0;
#pragma unroll
                    for (int k = 0; k < 19; ++k) cl[order[k]] = (uint32_t)(clb >> (3 * k)) & 7u;
                }
                uint32_t ccnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int s = 0; s < 19; ++s)
#pragma unroll
                    for (int l = 1; l <= 7; ++l) ccnt[l] += cl[s] == (uint32_t)l ? 1u : 0u;
                uint32_t cfirst[8], left = 1, code = 0;
                bool cover = false;
                cfirst[0] = 0;
#pragma unroll
                for (int l = 1; l <= 7; ++l) {
                    code = (code + ccnt[l - 1]) << 1;
                    if (l == 1) code = 0;
                    cfirst[l] = code;
                    left = (left << 1);
                    if (left < ccnt[l]) cover = true;
                    left -= ccnt[l];
                }
                if (cover || left != 0u) err = 6;         // (zlib refuses an incomplete code-length code too)
                uint32_t tt[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const uint32_t r7 = (uint32_t)lane + 64u * (uint32_t)h;
                    const uint32_t x = __brev(r7) >> 25;   // the 7 bits, first stream bit = MSB
                    uint32_t ent = 0;
#pragma unroll
                    for (int l = 1; l <= 7; ++l) {
                        const uint32_t cd = x >> (7 - l);
                        if (ent == 0u && cd >= cfirst[l] && cd - cfirst[l] < ccnt[l]) {
                            // the (cd - cfirst[l])-th symbol of length l in symbol order
                            uint32_t want = cd - cfirst[l], sym = 0;
#pragma unroll
                            for (int s = 0; s < 19; ++s) {
                                if (cl[s] == (uint32_t)l) { if (want == 0u) sym = (uint32_t)s; --want; }
                            }
                            ent = sym | ((uint32_t)l << 5);
                        }
                    }
                    tt[h] = ent;
                }
                uint32_t idx = 0;
                const uint32_t total = nl + nd;
                uint64_t wb = 0;
                uint32_t have = 0;                         // valid bits in wb
                while (err == 0u && idx < total) {
                    if (have < 14u) { wb = peek(hp); have = 57u; }
                    const uint32_t r7 = (uint32_t)wb & 127u;
                    const uint32_t a = WV_READLANE(tt[0], r7 & 63u), b = WV_READLANE(tt[1], r7 & 63u);
                    const uint32_t ent = r7 < 64u ? a : b;
                    const uint32_t l = ent >> 5, sym = ent & 31u;
                    if (l == 0u) { err = 7; break; }
                    wb >>= l; have -= l; hp += l;
                    if (sym < 16u) {
                        if (lane == 0) lens[idx] = (uint8_t)sym;
                        ++idx;
                        continue;
                    }
                    uint32_t prev = 0, rep;
                    if (sym == 16u) {
                        if (idx == 0u) { err = 8; break; }
                        WV_WAVE_SYNC();
                        prev = WV_UNIFORM(lens[idx - 1u]);
                        rep = 3u + ((uint32_t)wb & 3u); wb >>= 2; have -= 2u; hp += 2u;
                    } else if (sym == 17u) {
                        rep = 3u + ((uint32_t)wb & 7u); wb >>= 3; have -= 3u; hp += 3u;
                    } else {
                        rep = 11u + ((uint32_t)wb & 127u); wb >>= 7; have -= 7u; hp += 7u;
                    }
                    if (idx + rep > total) { err = 9; break; }
                    for (uint32_t k = (uint32_t)lane; k < rep; k += 64u) lens[idx + k] = (uint8_t)prev;
                    idx += rep;
                    if (hp > 8u * 1080u) err = 1;
                }
                WV_WAVE_SYNC();
                if (err == 0u && WV_UNIFORM(lens[256]) == 0u) err = 10;
                // (the distance lengths behind the lit/len lengths: move them to a fixed place)
                if (err == 0u) {
                    const uint32_t dl = lane < 32 && (uint32_t)lane < nd ? lens[nl + (uint32_t)lane] : 0u;
                    WV_WAVE_SYNC();
                    for (uint32_t s = nl + (uint32_t)lane; s < 288u; s += 64u) lens[s] = 0;
                    WV_WAVE_SYNC();
                    if (lane < 32) lens[288 + lane] = (uint8_t)dl;
                }
            }
            if (type == 1u && lane < 2) lens[318 + lane] = 0;
            if (lane == 0) {
                misc[WM_TYPE] = type; misc[WM_FINAL] = fin; misc[WM_NLEN] = nl; misc[WM_NDIST] = nd;
                misc[WM_HDREND] = (bitpos & ~7u) + hp; misc[WM_ERR] = err; misc[WM_STORED] = stored;
                misc[WM_EOB] = 0; misc[WM_FAIL] = 0; misc[WM_FLAG0] = 0; misc[WM_FLAG1] = 0;
            }
        }
        __syncthreads();
        WV_T(0);
        const uint32_t type = misc[WM_TYPE], fin = misc[WM_FINAL], nlen = misc[WM_NLEN];
        bitpos = misc[WM_HDREND];
        if (misc[WM_ERR] != 0u || bitpos > endbits) { fallback = true; break; }
        if (type == 0u) {
            // a stored block: its bytes straight from the payload
            const uint32_t n = misc[WM_STORED], from = bitpos >> 3;
            if (opos + n > olen || from + n > ilen) { fallback = true; break; }
            for (uint32_t k = (uint32_t)tid; k < n; k += NL) out[opos + k] = in[from + k];
            opos += n;
            bitpos += 8u * n;
            __syncthreads();
            if (fin) done = true;
            continue;
        }
        // ================= the tables =================
        // chunks of 64 symbols (five of lit/len, one of distance): the rank of every symbol among the equally long ones of its
        // chunk, the codes of each length per chunk
        for (int c = wave; c < 6; c += NW) {
            const uint32_t s = (uint32_t)(c < 5 ? c * 64 + lane : 288 + lane);
            const bool valid = c < 5 ? s < 288u : lane < 32;
            const uint32_t l = valid ? X[WX_LENS + s] : 0u;
            uint32_t rank = 0;
            for (uint32_t q = 1; q <= 15u; ++q) {
                const uint64_t mk = __ballot(l == q);
                if (l == q) rank = (uint32_t)__popcll(mk & ((1ull << lane) - 1ull));
                if (lane == 0) reinterpret_cast<uint16_t*>(X + WX_CCNT)[c * 16 + (int)q] = (uint16_t)__popcll(mk);
            }
            X[WX_RANK + c * 64 + lane] = (uint8_t)rank;
        }
        __syncthreads();
        if (tid < 16) {
            const uint16_t* const cc = reinterpret_cast<const uint16_t*>(X + WX_CCNT);
            uint16_t* const base = reinterpret_cast<uint16_t*>(X + WX_BASE);
            uint32_t a = 0;
            for (int c = 0; c < 5; ++c) { base[c * 16 + tid] = (uint16_t)a; a += cc[c * 16 + tid]; }
            base[5 * 16 + tid] = 0;
            reinterpret_cast<uint16_t*>(X + WX_CNT)[tid] = tid ? (uint16_t)a : (uint16_t)0;
            reinterpret_cast<uint16_t*>(X + WX_CNT)[16 + tid] = tid ? cc[5 * 16 + tid] : (uint16_t)0;
        }
        __syncthreads();
        const WvCanon CL = wv_canon_setup(reinterpret_cast<const uint16_t*>(X + WX_CNT), reinterpret_cast<uint16_t*>(X + WX_OFFS),
                                          reinterpret_cast<int16_t*>(X + WX_DELTA), tid == 0);
        const WvCanon CD = wv_canon_setup(reinterpret_cast<const uint16_t*>(X + WX_CNT) + 16, reinterpret_cast<uint16_t*>(X + WX_OFFS) + 16,
                                          reinterpret_cast<int16_t*>(X + WX_DELTA) + 16, tid == 0);
        const WvSub<WV_RL> SL = wv_sub_setup<WV_RL>(CL);
        const WvSub<WV_RD> SD = wv_sub_setup<WV_RD>(CD);
        // (an incomplete lit/len code is the other kernel's to judge; an incomplete distance code is common -- one distance
        // code, or none -- and only means entries without a code)
        if (CL.over || !CL.complete || CD.over || SL.cum[15 - WV_RL] > (uint32_t)WV_LSUB || SD.cum[15 - WV_RD] > (uint32_t)WV_DSUB) { fallback = true; break; }
        __syncthreads();
        for (int c = wave; c < 6; c += NW) {
            const uint32_t s = (uint32_t)(c < 5 ? c * 64 + lane : 288 + lane);
            const bool valid = c < 5 ? s < 288u : lane < 32;
            const uint32_t l = valid ? X[WX_LENS + s] : 0u;
            if (l) {
                const int t = c < 5 ? 0 : 1;
                const uint32_t at = reinterpret_cast<const uint16_t*>(X + WX_OFFS)[t * 16 + (int)l] + reinterpret_cast<const uint16_t*>(X + WX_BASE)[c * 16 + (int)l] +
                                    X[WX_RANK + c * 64 + lane];
                if (t == 0) reinterpret_cast<uint16_t*>(X + WX_SORTL)[at] = (uint16_t)s;
                else reinterpret_cast<uint16_t*>(X + WX_SORTD)[at] = (uint16_t)lane;
            }
        }
        __syncthreads();
        {
            const int16_t* const dl = reinterpret_cast<const int16_t*>(X + WX_DELTA);
            const uint16_t* const sortl = reinterpret_cast<const uint16_t*>(X + WX_SORTL);
            const uint16_t* const sortd = reinterpret_cast<const uint16_t*>(X + WX_SORTD);
            // lit/len: the root, then the second level
            for (uint32_t r = (uint32_t)tid; r < (1u << WV_RL); r += NL) {
                const uint32_t x = __brev(r) >> 17;        // the 10 bits left-aligned in 15
                const uint32_t n = wv_canon_len(CL, x);
                uint32_t e;
                if (n > 15u) e = (7u << 5) | 16u;
                else if (n <= (uint32_t)WV_RL) e = wv_lit_entry(sortl[(int)(x >> (15u - n)) + dl[n]], n);
                else {
                    // the longest code under this prefix: the code of the prefix's LAST value
                    const uint32_t P = x >> (15 - WV_RL);
                    const uint32_t nl2 = wv_canon_len(CL, x | ((1u << (15 - WV_RL)) - 1u));
                    uint32_t off = 0;
#pragma unroll
                    for (int q = 1; q <= 15 - WV_RL; ++q)
                        if (nl2 == (uint32_t)(WV_RL + q)) off = SL.cum[q - 1] + ((P - SL.B[q - 1]) << q);
                    e = nl2 > 15u ? ((7u << 5) | 16u) : (((off >> 1) << 8) | (7u << 5) | 16u | (nl2 - (uint32_t)WV_RL));
                }
                lit[r] = (uint16_t)e;
            }
            for (uint32_t j = (uint32_t)tid; j < SL.cum[15 - WV_RL]; j += NL) {
                uint32_t e = (7u << 5) | 16u;
#pragma unroll
                for (int q = 1; q <= 15 - WV_RL; ++q) {
                    if (j >= SL.cum[q - 1] && j < SL.cum[q]) {
                        const uint32_t jj = j - SL.cum[q - 1], P = SL.B[q - 1] + (jj >> q), sub = jj & ((1u << q) - 1u);
                        const uint32_t x = (P << (15 - WV_RL)) | ((__brev(sub) >> (32 - q)) << (15 - WV_RL - q));
                        const uint32_t n = wv_canon_len(CL, x);
                        if (n <= 15u && n > (uint32_t)WV_RL) e = wv_lit_entry(sortl[(int)(x >> (15u - n)) + dl[n]], n);
                    }
                }
                lit[(1u << WV_RL) + j] = (uint16_t)e;
            }
            // distance
            for (uint32_t r = (uint32_t)tid; r < (1u << WV_RD); r += NL) {
                const uint32_t x = __brev(r) >> 17;
                const uint32_t n = wv_canon_len(CD, x);
                uint32_t e;
                if (n > 15u) e = 16u;
                else if (n <= (uint32_t)WV_RD) { const uint32_t s = sortd[(int)(x >> (15u - n)) + dl[16 + n]]; e = s < 30u ? (s << 5) | n : 16u; }
                else {
                    const uint32_t P = x >> (15 - WV_RD);
                    const uint32_t nl2 = wv_canon_len(CD, x | ((1u << (15 - WV_RD)) - 1u));
                    uint32_t off = 0;
#pragma unroll
                    for (int q = 1; q <= 15 - WV_RD; ++q)
                        if (nl2 == (uint32_t)(WV_RD + q)) off = SD.cum[q - 1] + ((P - SD.B[q - 1]) << q);
                    // (an incomplete code: the prefix's last value may have no code -- then the whole prefix is refused, which
                    // the other kernel sorts out)
                    e = nl2 > 15u ? 16u : ((off << 5) | 16u | (nl2 - (uint32_t)WV_RD));
                }
                dis[r] = (uint16_t)e;
            }
            for (uint32_t j = (uint32_t)tid; j < SD.cum[15 - WV_RD]; j += NL) {
                uint32_t e = 16u;
#pragma unroll
                for (int q = 1; q <= 15 - WV_RD; ++q) {
                    if (j >= SD.cum[q - 1] && j < SD.cum[q]) {
                        const uint32_t jj = j - SD.cum[q - 1], P = SD.B[q - 1] + (jj >> q), sub = jj & ((1u << q) - 1u);
                        const uint32_t x = (P << (15 - WV_RD)) | ((__brev(sub) >> (32 - q)) << (15 - WV_RD - q));
                        const uint32_t n = wv_canon_len(CD, x);
                        if (n <= 15u && n > (uint32_t)WV_RD) { const uint32_t s = sortd[(int)(x >> (15u - n)) + dl[16 + n]]; if (s < 30u) e = (s << 5) | n; }
                    }
                }
                dis[(1u << WV_RD) + j] = (uint16_t)e;
            }
        }
        (void)nlen;
        __syncthreads();
        WV_T(1);

        // ================= pass A: where every lane's subsequence really starts, and what it produces =================
        const uint32_t body = bitpos;
        uint32_t S = (endbits - body + NL - 1u) / NL;
        if (S < 64u) S = 64u;
        const uint32_t b0 = body + (uint32_t)tid * S, bound = b0 + S;
        uint32_t nofail = 0;
        uint32_t mystart = b0;
        // the rest of the payload (and 16 bytes behind it: a run reads 8 bytes at a position up to 48 bits past the end) goes
        // into the part of the output area that is still free -- behind what the blocks before produced, which pass B2 will
        // read; it fits whenever what is left of the payload is not longer than what it still has to produce
        const uint32_t sfrom = body >> 3, sbytes = ilen + 16u - sfrom, sat = (opos + 15u) & ~15u;
        if (sat + sbytes > 65536u + 16u) { fallback = true; break; }
        for (uint32_t i = (uint32_t)tid * 16u; i < sbytes; i += 16u * NL) {
            inf_v4 v;
            __builtin_memcpy(&v, in + sfrom + i, 16);
            __builtin_memcpy(out + sat + i, &v, 16);
        }
        const uint8_t* const lin = out + sat - sfrom;      // payload byte k at lin[k]
        __syncthreads();
        {
            const WvRun r = wv_run<false>(lin, b0, bound, endbits, b0 < endbits, sm, XO, 0, nofail);
            l_cross[tid] = r.pos; l_cnt[tid] = r.cnt | (r.state << 24);
        }
        __syncthreads();
        for (uint32_t it = 0;; ++it) {
            bool redo = false;
            uint32_t from = 0;
            if (tid > 0) {
                const uint32_t ps = l_cnt[tid - 1] >> 24;
                if (ps == WS_CROSSED) { from = l_cross[tid - 1]; redo = from != mystart; }
            }
            if (redo) misc[WM_FLAG0 + (it & 1u)] = 1u;
            __syncthreads();
            const uint32_t any = misc[WM_FLAG0 + (it & 1u)];
            if (tid == 0) misc[WM_FLAG0 + ((it + 1u) & 1u)] = 0u;
            if (!any) break;
            if (it > (uint32_t)NL + 2u) { fallback = true; break; }   // (cannot happen: a pass fixes at least one lane for good)
            // every lane takes part (the run loop ballots); lanes with nothing to redo are done at once
            const WvRun r = wv_run<false>(lin, from, bound, endbits, redo, sm, XO, 0, nofail);
            if (redo) { mystart = from; l_cross[tid] = r.pos; l_cnt[tid] = r.cnt | (r.state << 24); }
            __syncthreads();
        }
        if (fallback) break;
        WV_T(2);
        // ================= the lanes' output offsets: an exclusive scan =================
        // a lane counts when every lane in front of it crossed into its successor; value: bytes | (blocks what follows) << 31
        {
            uint32_t* const sc = reinterpret_cast<uint32_t*>(X + WX_SCAN);
            const uint32_t mine = l_cnt[tid];
            const uint32_t st = mine >> 24;
            uint32_t v = (mine & 0xffffffu) | (st != WS_CROSSED ? 0x80000000u : 0u);
            sc[tid] = v;
            __syncthreads();
            int src = 0;
            for (int d = 1; d < NL; d <<= 1) {
                uint32_t a = sc[src * NL + tid];
                if (tid >= d) {
                    const uint32_t b = sc[src * NL + tid - d];
                    a = (((a & 0x7fffffffu) + (b & 0x7fffffffu)) & 0x7fffffffu) | ((a | b) & 0x80000000u);
                }
                sc[(src ^ 1) * NL + tid] = a;
                src ^= 1;
                __syncthreads();
            }
            const uint32_t incl = sc[src * NL + tid];
            const uint32_t excl_v = tid ? sc[src * NL + tid - 1] : 0u;
            const bool valid = !(excl_v & 0x80000000u);
            const uint32_t excl = excl_v & 0x7fffffffu;
            (void)incl;
            if (valid && st == WS_EOB) { misc[WM_EOB] = 1u; misc[WM_TOTAL] = excl + (mine & 0xffffffu); misc[WM_ENDPOS] = l_cross[tid]; }
            if (valid && (st == WS_BAD || st == WS_INACTIVE)) misc[WM_FAIL] = 1u;
            __syncthreads();
            const uint32_t total = misc[WM_TOTAL];
            if (!misc[WM_EOB] || misc[WM_FAIL] || opos + total > olen) { fallback = true; break; }
            // ================= pass B1: literals to their place, matches as pieces =================
            for (int i = tid; i < 2048 / 4; i += NL) reinterpret_cast<inf_v4*>(X + WX_BITMAP)[i] = inf_v4{0, 0, 0, 0};
            __syncthreads();
            WV_T(3);
            uint32_t fail = 0;
            const bool go = valid && (st == WS_CROSSED || st == WS_EOB);
            const WvRun r = wv_run<true>(in, mystart, bound, endbits, go, sm, XO, opos + excl, fail);
            if (fail || (go && (r.cnt != (mine & 0xffffffu) || r.state != st))) misc[WM_FAIL] = 1u;
            __syncthreads();
            if (misc[WM_FAIL]) { fallback = true; break; }
            WV_T(4);
            // ================= pass B2: the pieces in output order =================
            // (the lanes' arrays are free until the next block's pass A: they hold the piece starts of the window being resolved)
            if (wave == 0) {
                const uint32_t* const bitmap = reinterpret_cast<const uint32_t*>(X + WX_BITMAP);
                uint16_t* const stage = reinterpret_cast<uint16_t*>(sm + WV_LANE);   // [704]: pieces are >= 3 bytes, a window is 2 KB
                const uint32_t w_lo = opos >> 5, w_hi = (opos + total + 31u) >> 5;
                uint32_t carry_end = 0;                    // end of the last piece of the batch before
#ifdef GD_MEASURE
                blast = __builtin_readcyclecounter();
#endif
                for (uint32_t w0 = w_lo; w0 < w_hi; w0 += 64u) {
                    WV_BN(3);
                    uint32_t word = w0 + (uint32_t)lane < w_hi ? bitmap[w0 + (uint32_t)lane] : 0u;
                    const uint32_t c = (uint32_t)__popc(word);
                    const uint32_t incl2 = wv_wave_incl_scan(c, lane);
                    const uint32_t npc = WV_READLANE(incl2, 63);
                    uint32_t at = incl2 - c;
                    while (__ballot(word != 0u)) {
                        if (word) {
                            const uint32_t bq = (uint32_t)__builtin_ctz(word);
                            stage[at++] = (uint16_t)(((w0 + (uint32_t)lane) << 5) + bq);
                            word &= word - 1u;
                        }
                    }
                    WV_WAVE_SYNC();
                    WV_B(0);
                    for (uint32_t s0 = 0; s0 < npc; s0 += 64u) {
                        WV_BN(4);
                        const bool act = s0 + (uint32_t)lane < npc;
                        const uint32_t dst = act ? stage[s0 + (uint32_t)lane] : 0xfffffu;
                        const uint32_t tok = act ? wv_load4(out + dst) & 0xffffffu : 0u;
                        const uint32_t n = ((tok >> 1) & 15u) + 1u;
                        const bool per = (tok & 1u) != 0u;
                        const uint32_t p = (tok >> 5) & 15u, cc = (tok >> 9) & 255u;
                        const uint32_t src = per ? dst - cc - 16u : dst - (((tok >> 5) & 0x7fffu) + 1u);   // where the 16 bytes are read
                        const uint32_t s_lo = per ? dst - cc - p : src, s_hi = per ? dst - cc : src + n;   // what of them matters
                        const uint32_t endv = dst + n;
                        uint32_t prev_end = WV_SHFL_UP(endv, 1);
                        if (lane == 0) prev_end = carry_end;
                        // the selectors of a periodic piece, once per batch
                        inf_v4 sel0 = {0, 0, 0, 0}, sel1 = {0, 0, 0, 0};
                        if (__ballot(per)) {
                            if (per) {
                                sel0 = *reinterpret_cast<const inf_v4*>(X + WX_PERM + p * 32u);
                                sel1 = *reinterpret_cast<const inf_v4*>(X + WX_PERM + p * 32u + 16u);
                            }
                        }
                        // ready at once: the source lies in the literals right behind the previous piece; ready when
                        // everything below the first unresolved piece of the batch is resolved and the source ends there.
                        // The round loop runs on lane MASKS (scalar registers): which lanes are periodic, which store 16 / 8 / 4 /
                        // 2 / 1 bytes is fixed per batch, so a round is one compare, a few scalar operations, one 16-byte read
                        // and the stores that some ready lane needs -- the first form of this loop tested every lane's bits again
                        // in every round and took ~100 instructions (profiles/r13f_*).
                        const bool gap = s_lo >= prev_end;
                        const bool part = act && n != 16u;
                        const uint64_t MG = __ballot(act && gap), MP = __ballot(act && per), M16 = __ballot(act && n == 16u);
                        const uint64_t B8 = __ballot((n & 8u) != 0u), B4 = __ballot((n & 4u) != 0u), B2 = __ballot((n & 2u) != 0u);
                        const uint64_t M8 = __ballot(part && (n & 8u)), M4 = __ballot(part && (n & 4u)), M2 = __ballot(part && (n & 2u)),
                                       M1 = __ballot(part && (n & 1u));
                        const uint8_t* const s8 = out + (int)src;
                        uint8_t* const a8 = out + dst;
                        uint8_t* const a4 = a8 + (n & 8u);
                        uint8_t* const a2 = a4 + (n & 4u);
                        uint8_t* const a1 = a2 + (n & 2u);
                        uint64_t U = __ballot(act);
                        WV_B(1);
                        while (U) {
                            WV_BN(5);
                            const uint32_t first = (uint32_t)__builtin_ctzll(U);
                            const uint32_t F = WV_READLANE(dst, first);
                            const uint64_t R = (__ballot(s_hi <= F) | MG | (1ull << first)) & U;
                            inf_v4 v = {0, 0, 0, 0};
                            if (WV_LANE_IN(R, lane)) __builtin_memcpy(&v, s8, 16);
                            if (R & MP) {
                                if (WV_LANE_IN(R & MP, lane)) {
                                    // the last p bytes of v, repeated from phase 0
                                    inf_v4 q;
                                    q.x = __builtin_amdgcn_perm(v.y, v.x, sel0.x) | __builtin_amdgcn_perm(v.w, v.z, sel1.x);
                                    q.y = __builtin_amdgcn_perm(v.y, v.x, sel0.y) | __builtin_amdgcn_perm(v.w, v.z, sel1.y);
                                    q.z = __builtin_amdgcn_perm(v.y, v.x, sel0.z) | __builtin_amdgcn_perm(v.w, v.z, sel1.z);
                                    q.w = __builtin_amdgcn_perm(v.y, v.x, sel0.w) | __builtin_amdgcn_perm(v.w, v.z, sel1.w);
                                    v = q;
                                }
                            }
                            if (WV_LANE_IN(R & M16, lane)) __builtin_memcpy(a8, &v, 16);
                            if (R & ~M16) {
                                if (WV_LANE_IN(R & M8, lane)) { const uint64_t lo = (uint64_t)v.x | ((uint64_t)v.y << 32); __builtin_memcpy(a8, &lo, 8); }
                                const uint32_t x4 = WV_LANE_IN(B8, lane) ? v.z : v.x, y4 = WV_LANE_IN(B8, lane) ? v.w : v.y;
                                if (WV_LANE_IN(R & M4, lane)) __builtin_memcpy(a4, &x4, 4);
                                const uint32_t x2 = WV_LANE_IN(B4, lane) ? y4 : x4;
                                if (WV_LANE_IN(R & M2, lane)) { const uint16_t h = (uint16_t)x2; __builtin_memcpy(a2, &h, 2); }
                                const uint32_t x1 = WV_LANE_IN(B2, lane) ? x2 >> 16 : x2;
                                if (WV_LANE_IN(R & M1, lane)) *a1 = (uint8_t)x1;
                            }
                            U &= ~R;
                            WV_WAVE_SYNC();
                        }
                        WV_B(2);
                        carry_end = WV_READLANE(endv, (npc - s0 < 64u ? npc - s0 : 64u) - 1u);
                    }
                }
            }
            __syncthreads();
            WV_T(5);
            opos += total;
            bitpos = misc[WM_ENDPOS];
            if (fin) done = true;
        }
    }
    if (!fallback && (opos != olen || bitpos > endbits)) fallback = true;
    if (!fallback) {
        // ================= store: LDS -> memory =================
        uint8_t* const g = job.out + job.out_off[m];
        const uint32_t head = (uint32_t)((16u - (reinterpret_cast<uintptr_t>(g) & 15u)) & 15u);
        const uint32_t h = head < olen ? head : olen;
        if ((uint32_t)tid < h) g[tid] = out[tid];
        const uint32_t body16 = (olen - h) >> 4;
        for (uint32_t i = (uint32_t)tid; i < body16; i += NL) {
            inf_v4 v;
            __builtin_memcpy(&v, out + h + 16u * i, 16);
            *reinterpret_cast<inf_v4*>(g + h + 16u * i) = v;
        }
        const uint32_t tail0 = h + 16u * body16;
        if (tail0 + (uint32_t)tid < olen && tid < 16) g[tail0 + tid] = out[tail0 + tid];
    }
    if (tid == 0) job.status[m] = fallback ? WV_FALLBACK : 0u;
#ifdef GD_MEASURE
    WV_T(6);
    if (tid == 0) {
        for (int k = 0; k < 7; ++k) atomicAdd(&::g_inflate_sections[8 + k], (unsigned long long)wsum[k]);
        atomicAdd(&::g_inflate_sections[15], 1ull);
        for (int k = 0; k < 6; ++k) atomicAdd(&gd::g_inflate_b2[k], (unsigned long long)bsum[k]);
    }
#endif
}

}  // namespace gd
