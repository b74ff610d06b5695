// gd_inflate.hpp -- BGZF (RFC 1951 DEFLATE) decompression on the device.
//
// The BAM read that every `samtools depth` child of the reference performs
// (/root/reference/depth/depth.go:45) starts with inflating BGZF members; on the host that
// is the end-to-end limiter (scope iii).  A BGZF file is a sequence of
// independent <= 64 KiB deflate streams, so the file offers tens of thousands to millions
// of independent decode jobs: here ONE LANE inflates ONE member, start to end, with the
// classic canonical-Huffman bit-serial decoder (count[] / symbol[] per code length, as in
// zlib's contrib/puff) -- no shared state between lanes, no host involvement beyond
// listing the members.  The Huffman tables of a lane live in LDS ([entry][lane] layout: the
// 64 lanes of a wave read/write the same entry of 64 different tables without bank
// conflicts when they are in step), the code lengths of a dynamic block in the lane's
// private memory, input is fetched 4 bytes at a time, output goes to the member's own
// region of the inflated buffer (back-references read that same region); the member's CRC32
// is verified at the end (table in LDS).
#pragma once

namespace gd {

struct InflateJob {
    const uint8_t* comp;           // the compressed bytes
    const uint64_t* in_off;        // [n] offset of each member's deflate payload in comp
    const uint32_t* in_len;        // [n] payload bytes
    const uint64_t* out_off;       // [n] offset of the member's data in out
    const uint32_t* out_len;       // [n] ISIZE
    const uint32_t* crc;           // [n] CRC32 of the member's data (gzip trailer), or nullptr: not checked
    uint8_t* out;
    uint32_t* status;              // [n] 0 ok, else an error code (18: CRC32 mismatch)
    uint32_t n;
};

constexpr int INF_LANES = 64;      // lanes (= members in flight) per workgroup
constexpr int INF_MAXL = 288, INF_MAXD = 30;
// Per-lane LDS tables, 384 bytes: cnt[16] (uint16, shared by the two builds: the counts live in
// registers while a block is decoded), nlit[16] (uint16), litlen symbols [288] and distance symbols
// [32] as BYTES.  A lit/len symbol needs 9 bits; the ninth is not stored: within one code length the
// canonical order is ascending symbol value, so the first nlit[len] entries are literals (< 256) and
// the rest are 256 + the stored byte.  (uint16 symbols were 700 bytes per lane = 3 workgroups per CU;
// one lane inflates one member and is latency bound, so members in flight are the throughput.)
constexpr int INF_TBL_BYTES = 32 + 32 + INF_MAXL + 32;

struct BitReader {
    const uint8_t* p;              // next byte not yet fetched into `ahead`
    const uint8_t* end;
    uint64_t buf;                  // bits not yet consumed, LSB first
    uint64_t ahead;                // the next 8 input bytes, fetched one refill early (hides the load latency)
    int cnt, acnt;                 // valid bits in buf / valid BYTES in ahead
    bool bad;
    __device__ __forceinline__ void fetch()
    {
        ahead = 0;
        acnt = 0;
        if (p + 8 <= end) { __builtin_memcpy(&ahead, p, 8); acnt = 8; p += 8; }   // unaligned 64-bit global load
        else while (p < end) { ahead |= (uint64_t)(*p++) << (8 * acnt); ++acnt; }
    }
    __device__ __forceinline__ void init(const uint8_t* b, const uint8_t* e)
    {
        p = b; end = e; buf = 0; cnt = 0; bad = false;
        fetch();
    }
    __device__ __forceinline__ void refill()              // afterwards cnt >= 32 unless the input is exhausted
    {
        if (cnt <= 32 && acnt > 0) {
            const int take = acnt < 4 ? acnt : 4;         // whole bytes that fit: 32 + 32 <= 64
            buf |= (ahead & (take == 4 ? 0xffffffffull : ((1ull << (8 * take)) - 1))) << cnt;
            cnt += 8 * take;
            ahead >>= 8 * take;
            acnt -= take;
            if (acnt == 0) fetch();
        }
    }
    __device__ __forceinline__ uint32_t bits(int n)       // n <= 16
    {
        if (cnt < n) { refill(); if (cnt < n) { bad = true; return 0; } }
        const uint32_t v = (uint32_t)(buf & ((1ull << n) - 1));
        buf >>= n;
        cnt -= n;
        return v;
    }
};

// Canonical Huffman tables of one lane in LDS: t[entry * INF_LANES + lane]
struct HuffLds {
    uint16_t* cnt;                 // [16] codes of each length
    uint16_t* nlit;                // [16] of those, symbols below 256 (lit/len table only)
    uint8_t* sym;                  // low 8 bits of the symbols, ordered by code
    __device__ __forceinline__ uint16_t& c(int i) const { return cnt[i * INF_LANES]; }
    __device__ __forceinline__ uint16_t& nl(int i) const { return nlit[i * INF_LANES]; }
    __device__ __forceinline__ uint8_t& s(int i) const { return sym[i * INF_LANES]; }
};

// Builds the tables from code lengths (puff.c construct()); returns false for an over-subscribed set.
// LIT: also count the symbols below 256 per length (what restores the ninth symbol bit).
template <bool LIT>
__device__ __forceinline__ bool huff_build(const HuffLds& h, const uint8_t* lengths, int n)
{
    for (int l = 0; l <= 15; ++l) { h.c(l) = 0; if (LIT) h.nl(l) = 0; }
    for (int i = 0; i < n; ++i) {
        h.c(lengths[i]) = (uint16_t)(h.c(lengths[i]) + 1);
        if (LIT && i < 256) h.nl(lengths[i]) = (uint16_t)(h.nl(lengths[i]) + 1);
    }
    int left = 1;
    for (int l = 1; l <= 15; ++l) {
        left <<= 1;
        left -= h.c(l);
        if (left < 0) return false;
    }
    uint16_t offs[16];
    offs[1] = 0;
    for (int l = 1; l < 15; ++l) offs[l + 1] = (uint16_t)(offs[l] + h.c(l));
    for (int i = 0; i < n; ++i)
        if (lengths[i] != 0) h.s(offs[lengths[i]]++) = (uint8_t)i;
    return true;
}

// The 15 per-length code counts of a table, held in registers while a block is decoded (the
// bit-serial walk below then touches LDS once per symbol instead of once per code bit).
struct HuffCnt {
    uint32_t w[8];                 // counts of lengths 2k, 2k+1 in the halves of w[k]
    __device__ __forceinline__ void load(const HuffLds& h)
    {
#pragma unroll
        for (int k = 0; k < 8; ++k) w[k] = (uint32_t)h.c(2 * k) | ((uint32_t)h.c(2 * k + 1) << 16);
    }
    __device__ __forceinline__ void load_nlit(const HuffLds& h)
    {
#pragma unroll
        for (int k = 0; k < 8; ++k) w[k] = (uint32_t)h.nl(2 * k) | ((uint32_t)h.nl(2 * k + 1) << 16);
    }
    __device__ __forceinline__ int at(int len) const { return (int)((w[len >> 1] >> ((len & 1) * 16)) & 0xffffu); }
};

// One symbol (puff.c decode(), bit serial over the code lengths).  LIT: a lit/len table, nl holds its
// per-length literal counts.
template <bool LIT>
__device__ __forceinline__ int huff_decode(BitReader& br, const HuffLds& h, const HuffCnt& hc, const HuffCnt& nl)
{
    if (br.cnt < 15) br.refill();
    int code = 0, first = 0, index = 0;
    uint64_t b = br.buf;
#pragma unroll
    for (int len = 1; len <= 15; ++len) {
        code |= (int)(b & 1);
        b >>= 1;
        const int count = hc.at(len);
        if (code - count < first) {
            if (br.cnt < len) { br.bad = true; return -1; }
            br.buf >>= len;
            br.cnt -= len;
            const int k = code - first;
            const int v = h.s(index + k);
            return LIT ? v | ((int)(k >= nl.at(len)) << 8) : v;
        }
        index += count;
        first += count;
        first <<= 1;
        code <<= 1;
    }
    br.bad = true;
    return -1;
}

__global__ __launch_bounds__(INF_LANES) void gd_inflate_kernel(InflateJob job)
{
    __shared__ __attribute__((aligned(4))) uint8_t s_tbl[INF_TBL_BYTES * INF_LANES];
    __shared__ uint32_t s_crc[256];                        // CRC-32 (IEEE 802.3, reflected) byte table
    for (int i = threadIdx.x; i < 256; i += INF_LANES) {
        uint32_t c = (uint32_t)i;
        for (int k = 0; k < 8; ++k) c = (c >> 1) ^ (0xedb88320u & (0u - (c & 1u)));
        s_crc[i] = c;
    }
    __syncthreads();
    const uint32_t m = blockIdx.x * INF_LANES + threadIdx.x;
    if (m >= job.n) return;
    const int lane = threadIdx.x;
    HuffLds hl, hd;
    hl.cnt = reinterpret_cast<uint16_t*>(s_tbl) + lane;
    hl.nlit = reinterpret_cast<uint16_t*>(s_tbl + 32 * INF_LANES) + lane;
    hl.sym = s_tbl + 64 * INF_LANES + lane;
    hd.cnt = hl.cnt;                                       // built after the lit/len counts are in registers
    hd.nlit = nullptr;
    hd.sym = s_tbl + (64 + INF_MAXL) * INF_LANES + lane;

    BitReader br;
    br.init(job.comp + job.in_off[m], job.comp + job.in_off[m] + job.in_len[m]);
    uint8_t* const out = job.out + job.out_off[m];
    const uint32_t olen = job.out_len[m];
    uint32_t o = 0;
    uint32_t err = 0;

    static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

    bool last = false;
    while (!last && err == 0) {
        last = br.bits(1) != 0;
        const uint32_t type = br.bits(2);
        if (br.bad) { err = 1; break; }
        if (type == 0) {                                   // stored
            br.buf >>= (br.cnt & 7);                       // to the byte boundary
            br.cnt &= ~7;
            const uint32_t len = br.bits(16), nlen = br.bits(16);
            if (br.bad || (len ^ 0xffffu) != nlen) { err = 2; break; }
            if (o + len > olen) { err = 3; break; }
            for (uint32_t k = 0; k < len; ++k) {
                const uint32_t v = br.bits(8);
                out[o++] = (uint8_t)v;
            }
            if (br.bad) { err = 1; break; }
            continue;
        }
        if (type == 3) { err = 4; break; }
        uint8_t lengths[INF_MAXL + INF_MAXD + 2];
        HuffCnt cl, cd, nl;
        if (type == 1) {                                   // fixed codes
            int s = 0;
            for (; s < 144; ++s) lengths[s] = 8;
            for (; s < 256; ++s) lengths[s] = 9;
            for (; s < 280; ++s) lengths[s] = 7;
            for (; s < 288; ++s) lengths[s] = 8;
            huff_build<true>(hl, lengths, 288);
            cl.load(hl);
            nl.load_nlit(hl);
            for (s = 0; s < 30; ++s) lengths[s] = 5;
            huff_build<false>(hd, lengths, 30);
            cd.load(hd);
        } else {                                           // dynamic codes
            const int nlen = (int)br.bits(5) + 257, ndist = (int)br.bits(5) + 1, ncode = (int)br.bits(4) + 4;
            if (br.bad || nlen > INF_MAXL || ndist > INF_MAXD) { err = 5; break; }
            int idx = 0;
            for (; idx < ncode; ++idx) lengths[order[idx]] = (uint8_t)br.bits(3);
            for (; idx < 19; ++idx) lengths[order[idx]] = 0;
            if (!huff_build<false>(hl, lengths, 19)) { err = 6; break; }   // the code-length code borrows the lit/len table
            HuffCnt cc;
            cc.load(hl);
            idx = 0;
            while (idx < nlen + ndist) {
                int sym = huff_decode<false>(br, hl, cc, cc);
                if (sym < 0) { err = 7; break; }
                if (sym < 16) { lengths[idx++] = (uint8_t)sym; continue; }
                int prev = 0, rep;
                if (sym == 16) {
                    if (idx == 0) { err = 8; break; }
                    prev = lengths[idx - 1];
                    rep = 3 + (int)br.bits(2);
                } else if (sym == 17) rep = 3 + (int)br.bits(3);
                else rep = 11 + (int)br.bits(7);
                if (idx + rep > nlen + ndist) { err = 9; break; }
                while (rep--) lengths[idx++] = (uint8_t)prev;
            }
            if (err) break;
            if (lengths[256] == 0) { err = 10; break; }
            uint8_t dl[INF_MAXD];
            for (int k = 0; k < ndist; ++k) dl[k] = lengths[nlen + k];
            if (!huff_build<true>(hl, lengths, nlen)) { err = 11; break; }
            cl.load(hl);
            nl.load_nlit(hl);
            if (!huff_build<false>(hd, dl, ndist)) { err = 12; break; }
            cd.load(hd);
        }
        // ---- the block's symbols ---------------------------------------------
        // Short matches far enough back (the common case) are DEFERRED: their 16 source bytes are
        // loaded now and stored only after the next symbol has been decoded, so the global load
        // latency overlaps that decode instead of stalling the wave (with 64 lanes in 64 different
        // places, nearly every step has some lane copying).  Bytes past the match length that the
        // 16-byte store also writes are overwritten by the output that follows.
        uint32_t pw0 = 0, pw1 = 0, pw2 = 0, pw3 = 0, pend_o = 0;
        int pend_n = 0;
        auto flush = [&]() {
            if (pend_n == 0) return;
            if (pend_o + 16u <= olen) {
                __builtin_memcpy(out + pend_o, &pw0, 4); __builtin_memcpy(out + pend_o + 4, &pw1, 4);
                __builtin_memcpy(out + pend_o + 8, &pw2, 4); __builtin_memcpy(out + pend_o + 12, &pw3, 4);
            } else {                                       // next to the member's end: exactly the match
                const uint32_t w[4] = {pw0, pw1, pw2, pw3};
                for (int k = 0; k < pend_n; ++k) out[pend_o + k] = (uint8_t)(w[k >> 2] >> (8 * (k & 3)));
            }
            pend_n = 0;
        };
        for (;;) {
            const int sym = huff_decode<true>(br, hl, cl, nl);
            flush();
            if (sym < 0) { err = 13; break; }
            if (sym < 256) {
                if (o >= olen) { err = 3; break; }
                out[o++] = (uint8_t)sym;
            } else if (sym == 256) {
                break;
            } else {
                // base value and extra-bit count of length / distance symbols in closed form
                // (RFC 1951 3.2.5) -- table lookups would be dependent global loads on the
                // critical path of every match
                const int ls = sym - 257;
                if (ls >= 29) { err = 14; break; }
                uint32_t len;
                if (ls < 8) len = 3u + (uint32_t)ls;
                else if (ls == 28) len = 258u;
                else { const int e = (ls >> 2) - 1; len = 3u + ((4u + (uint32_t)(ls & 3)) << e) + br.bits(e); }
                const int ds = huff_decode<false>(br, hd, cd, cd);
                if (ds < 0 || ds >= 30) { err = 15; break; }
                uint32_t dist;
                if (ds < 4) dist = 1u + (uint32_t)ds;
                else { const int e = (ds >> 1) - 1; dist = 1u + ((2u + (uint32_t)(ds & 1)) << e) + br.bits(e); }
                if (br.bad) { err = 1; break; }
                if (dist > o || o + len > olen) { err = 16; break; }
                if (dist >= 16 && len <= 16) {
                    const uint8_t* src = out + o - dist;    // src + 16 <= out + o: only finished bytes are read
                    __builtin_memcpy(&pw0, src, 4); __builtin_memcpy(&pw1, src + 4, 4);
                    __builtin_memcpy(&pw2, src + 8, 4); __builtin_memcpy(&pw3, src + 12, 4);
                    pend_o = o;
                    pend_n = (int)len;
                    o += len;
                    continue;
                }
                uint32_t k = 0;
                if (dist < 4) {
                    // run-length style matches (distance 1..3, up to 258 bytes): the pattern is read
                    // once and replayed from registers -- a byte loop here would chain a
                    // store -> load round trip through memory per output byte
                    const uint32_t b0 = out[o - dist];
                    const uint32_t b1 = dist > 1 ? out[o - dist + 1] : b0;
                    const uint32_t b2 = dist > 2 ? out[o - dist + 2] : (dist == 2 ? b0 : b0);
                    uint32_t w[3];
                    if (dist == 1) { w[0] = w[1] = w[2] = b0 * 0x01010101u; }
                    else if (dist == 2) { w[0] = w[1] = w[2] = b0 | (b1 << 8) | (b0 << 16) | (b1 << 24); }
                    else {
                        w[0] = b0 | (b1 << 8) | (b2 << 16) | (b0 << 24);
                        w[1] = b1 | (b2 << 8) | (b0 << 16) | (b1 << 24);
                        w[2] = b2 | (b0 << 8) | (b1 << 16) | (b2 << 24);
                    }
                    uint32_t ph = 0;
                    for (; k + 4 <= len; k += 4, o += 4) {
                        const uint32_t v = ph == 0 ? w[0] : ph == 1 ? w[1] : w[2];
                        __builtin_memcpy(out + o, &v, 4);
                        ph = ph == 2 ? 0 : ph + 1;
                    }
                    uint32_t v = ph == 0 ? w[0] : ph == 1 ? w[1] : w[2];
                    for (; k < len; ++k, ++o, v >>= 8) out[o] = (uint8_t)v;
                }
                if (dist >= 16)                            // 16-byte groups: four loads in flight, then four stores
                    for (; k + 16 <= len; k += 16, o += 16) {
                        uint32_t w0, w1, w2, w3;
                        const uint8_t* src = out + o - dist;
                        __builtin_memcpy(&w0, src, 4); __builtin_memcpy(&w1, src + 4, 4);
                        __builtin_memcpy(&w2, src + 8, 4); __builtin_memcpy(&w3, src + 12, 4);
                        __builtin_memcpy(out + o, &w0, 4); __builtin_memcpy(out + o + 4, &w1, 4);
                        __builtin_memcpy(out + o + 8, &w2, 4); __builtin_memcpy(out + o + 12, &w3, 4);
                    }
                if (dist >= 4)                             // source and destination words do not overlap
                    for (; k + 4 <= len; k += 4, o += 4) {
                        uint32_t w;
                        __builtin_memcpy(&w, out + o - dist, 4);
                        __builtin_memcpy(out + o, &w, 4);
                    }
                for (; k < len; ++k, ++o) out[o] = out[o - dist];
            }
        }
        flush();                                           // a match deferred by the block's last step
    }
    if (err == 0 && (o != olen || br.bad)) err = 17;
    if (err == 0 && job.crc) {
        // the gzip trailer's CRC32 over what was just written (all lanes are here together again:
        // a converged byte loop, ~2 % of the decode time)
        uint32_t c = 0xffffffffu;
        uint32_t k = 0;
        for (; k + 4 <= olen; k += 4) {
            uint32_t w;
            __builtin_memcpy(&w, out + k, 4);
            c = s_crc[(c ^ w) & 0xffu] ^ (c >> 8);
            c = s_crc[(c ^ (w >> 8)) & 0xffu] ^ (c >> 8);
            c = s_crc[(c ^ (w >> 16)) & 0xffu] ^ (c >> 8);
            c = s_crc[(c ^ (w >> 24)) & 0xffu] ^ (c >> 8);
        }
        for (; k < olen; ++k) c = s_crc[(c ^ out[k]) & 0xffu] ^ (c >> 8);
        if ((c ^ 0xffffffffu) != job.crc[m]) err = 18;
    }
    job.status[m] = err;
}

}  // namespace gd
