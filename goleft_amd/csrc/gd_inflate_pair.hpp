// gd_inflate_pair.hpp -- the inflate kernel as TWO waves per 64 members: a decoder wave and a writer wave.
//
// gd_inflate_kernel (gd_inflate.hpp) is bound by instruction issue, not by memory: a wave issues one instruction per four
// cycles whatever its kind, its ~750 instructions per iteration (450 vector, 250 scalar, 50 LDS / memory) take the ~4 400
// cycles the section counters of the measurement build show (profiles/r12j_...: the two loads of an iteration are not
// waited for at all), and the 608 bytes of LDS a member needs (tables, input window, output ring) allow four waves per CU:
// ONE wave per SIMD, nobody to share the issue slots with.  More members per CU do not fit.  But the work of a member
// splits in two halves that touch different state:
//   * the DECODER (wave 0 of the workgroup) owns the bit buffer, the input window and the Huffman tables: block headers,
//     three symbols decoded per iteration, every check of the stream -- and turns what it decodes into 32-bit TOKENS
//     (one or two literals; a match; end of block; a stored block; the end of the member);
//   * the WRITER (wave 1) owns the output: the 16-byte register window T, the 128-byte ring, the chunks of a match and
//     their source loads, the 64-byte block stores.  It executes tokens that are valid by construction.
// Lane l of both waves works on member 64 * block + l; the two meet in a queue of INF_QCAP tokens per lane in LDS
// ([slot][lane], one producer and one consumer per lane: a tail counter the decoder writes, a head counter the writer
// writes; LDS operations of a wave execute in order, so a token is written before the tail that publishes it and read
// after the tail that announced it).  Nobody spins: a decoder lane whose queue is full decodes nothing in that iteration,
// a writer lane whose queue is empty executes nothing.  The same 40 KB of LDS per 64 members (38 912 + 2 048 for the
// queues: four workgroups per CU as before), but EIGHT waves per CU: two per SIMD, each with half of the instructions.
#pragma once

namespace gd {

constexpr int INF_QCAP = 6;                                // tokens a decoder lane may be ahead of its writer lane
constexpr int INF_Q = INF_LDS_BYTES;                       // u32 [INF_QCAP][64]
constexpr int INF_QTAIL = INF_Q + INF_QCAP * 64 * 4;       // u32 [64]: tokens pushed so far (INF_QABORT: the decoder gave up)
constexpr int INF_QHEAD = INF_QTAIL + 64 * 4;              // u32 [64]: tokens popped so far
constexpr int INF_PAIR_LDS_BYTES = INF_QHEAD + 64 * 4;     // 40 960: four workgroups per CU
constexpr uint32_t INF_QABORT = 0xffffffffu;

// A token: bits 1:0 its kind.
//   LIT1 / LIT2   byte(s) at bits 15:8 (23:16)
//   MATCH         length - 3 at bits 9:2, distance - 1 at bits 24:10
//   SPECIAL       bits 3:2: END (of the member: flush and finish), FLUSH (end of a block: what is pending goes into the
//                 ring), STORED_OFF (a stored block's bytes begin at payload offset `bits 31:4`), STORED_LEN (... and are
//                 `bits 31:4` many: copy them)
enum : uint32_t { INF_TK_LIT1 = 0, INF_TK_LIT2 = 1, INF_TK_MATCH = 2, INF_TK_SPECIAL = 3 };
enum : uint32_t { INF_TS_END = 0, INF_TS_FLUSH = 1, INF_TS_STORED_OFF = 2, INF_TS_STORED_LEN = 3 };
__device__ __forceinline__ uint32_t inf_special(uint32_t sub, uint32_t arg) { return INF_TK_SPECIAL | (sub << 2) | (arg << 4); }

// (the queue's words are read and written in program order: volatile, and nothing the compiler may move across them)
__device__ __forceinline__ uint32_t inf_q_read(const uint32_t* p) { return *reinterpret_cast<const volatile uint32_t*>(p); }
__device__ __forceinline__ void inf_q_write(uint32_t* p, uint32_t v)
{
    __atomic_signal_fence(__ATOMIC_SEQ_CST);
    *reinterpret_cast<volatile uint32_t*>(p) = v;
    __atomic_signal_fence(__ATOMIC_SEQ_CST);
}

// ---- the decoder wave -------------------------------------------------------------------------------------------------
__device__ __forceinline__ void inflate_pair_decode(const InflateJob& job, uint8_t* s_tbl, const int lane, const uint32_t m, const bool mine)
{
    const uint32_t mm = mine ? m : 0u;
    const uint8_t* const in_beg = job.comp + job.in_off[mm];
    const uint8_t* const in_end = in_beg + job.in_len[mm];
    const uint32_t olen = job.out_len[mm];
    uint32_t* const q = reinterpret_cast<uint32_t*>(s_tbl + INF_Q) + lane;
    uint32_t* const q_tail = reinterpret_cast<uint32_t*>(s_tbl + INF_QTAIL) + lane;
    const uint32_t* const q_head = reinterpret_cast<const uint32_t*>(s_tbl + INF_QHEAD) + lane;

    enum : uint32_t { DECODE = 0, HDR = 2, DONE = 3 };
    uint32_t mode = mine ? HDR : DONE;
    uint32_t err = 0;
    uint32_t o = 0;                                        // bytes of the member the tokens pushed (or waiting) so far produce
    bool lastblk = false;
    bool after_match = false;                              // the previous token was a match: a pair of literals must not follow it
                                                           // (the writer appends the match's last chunk in the iteration that
                                                           // executes the next token: its ring write carries 17 new bytes at most)
    uint32_t ptok0 = 0, ptok1 = 0, ptok2 = 0, npend = 0;   // tokens of the header path that wait for room
    uint32_t tail = 0, tslot = 0;
    uint32_t LE[8], DE[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) LE[k] = DE[k] = 0;
    const uint8_t* p = in_beg;
    uint64_t buf = 0;
    uint32_t cnt = 0;
    // the member's input window (see gd_inflate_kernel)
    uint32_t* const s_win = reinterpret_cast<uint32_t*>(s_tbl + INF_INWIN) + lane;
    uint32_t win_hi = 0;
    auto win_put = [&](uint32_t at, inf_v4 v) {
        const uint32_t j = (at >> 2) & 12u;
        s_win[(j + 0u) * 64u] = v.x; s_win[(j + 1u) * 64u] = v.y; s_win[(j + 2u) * 64u] = v.z; s_win[(j + 3u) * 64u] = v.w;
    };
    auto win_restart = [&](uint32_t poff) {
        const uint32_t b = poff & ~15u;
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k) win_put(b + 16u * k, inf_load16(in_beg + b + 16u * k));
        win_hi = b + 64u;
    };
    win_restart(0);
    auto pend_push = [&](uint32_t t) {                     // (three at most: the two of a stored block and the member's end)
        if (npend == 0u) ptok0 = t; else if (npend == 1u) ptok1 = t; else ptok2 = t;
        ++npend;
    };

    for (uint32_t it = 0;; ++it) {
        const uint64_t live = __ballot(mode != DONE || npend != 0u);
        if (live == 0) break;
        if (it >= (1u << 22)) {                            // a backstop (see gd_inflate_kernel); the writer is told to stop
            if (mode != DONE || npend != 0u) {
                if (err == 0u) err = 19;
                inf_q_write(q_tail, INF_QABORT);
            }
            break;
        }
        inf_v4 in16 = {0, 0, 0, 0};
        bool want_in = mode != DONE && (uint32_t)(p - in_beg) + 48u >= win_hi;
        if (want_in) in16 = inf_load16(in_beg + win_hi);

        // ---- block header (a divergent side path; lanes wait for each other to build together) ----
        const uint64_t decoding = __ballot(mode != DONE);
        const uint64_t hm = __ballot(mode == HDR && npend == 0u);
        if (hm != 0 && (hm == decoding || __popcll(hm) >= 32 || (it & 31u) == 31u)) {
            if (mode == HDR && npend == 0u) {
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
                if (type == 0) {                           // stored: the writer copies the bytes, the decoder steps over them
                    const uint32_t drop = cnt & 7u;
                    buf >>= drop;
                    cnt -= drop;
                    const uint32_t len = bits(16), nlen = bits(16);
                    if ((len ^ 0xffffu) != nlen) err = 2;
                    else if (o + len > olen || past_end()) err = 3;
                    else {
                        const uint32_t soff = (uint32_t)(p - in_beg) - (cnt >> 3);   // (the bit buffer holds whole bytes here)
                        if ((uint64_t)soff + len > (uint64_t)(in_end - in_beg)) err = 1;
                        else {
                            if (len != 0u) {
                                pend_push(inf_special(INF_TS_STORED_OFF, soff));
                                pend_push(inf_special(INF_TS_STORED_LEN, len));
                            }
                            o += len;
                            p = in_beg + soff + len;
                            buf = 0;
                            cnt = 0;
                        }
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
                if (mode == DONE) pend_push(inf_special(INF_TS_END, 0));
                need(56);                                  // what the decode below may consume
                win_restart((uint32_t)(p - in_beg));       // the header was read past the window
                want_in = false;                           // (16 bytes asked for before the header belong behind the OLD window)
            }
        }

        // ---- room for a token?  (the head only grows: a stale value errs on the side of waiting) ----
        const bool room = tail - inf_q_read(q_head) < (uint32_t)INF_QCAP;
        const uint32_t poff = (uint32_t)(p - in_beg);
        const uint32_t wj = poff >> 2;
        const uint32_t wd0 = s_win[((wj + 0u) & 15u) * 64u], wd1 = s_win[((wj + 1u) & 15u) * 64u], wd2 = s_win[((wj + 2u) & 15u) * 64u];

        // ---- decode one symbol: everything a match needs, for every lane (no branches); see gd_inflate_kernel ----
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
        uint32_t l3 = 0;
        bool bad3 = false;
        const uint32_t sym2 = inf_decode<true>(x2, LE, s_tbl, lane, l3, bad3);
        const uint32_t e2 = ds < 4u ? 0u : (ds >> 1) - 1u;
        const uint32_t mdist = ds < 4u ? 1u + ds : 1u + ((2u + (ds & 1u)) << e2) + ((lo2 >> l2) & ((1u << e2) - 1u));
        const uint32_t used2 = used1 + l2 + e2;            // <= 48
        __builtin_amdgcn_s_waitcnt(0x0F70);                // the input load (see gd_inflate_kernel: vmcnt 0 on every path)

        // ---- this iteration's token ----
        uint32_t tok = 0;
        bool have = false;
        if (mode == DECODE && npend == 0u && room) {
            have = true;
            if (bad1) { err = 13; mode = DONE; }
            else if (sym < 256u) {
                if (o >= olen) { err = 3; mode = DONE; }
                else {
                    const bool two = !bad3 && sym2 < 256u && o + 2u <= olen && !after_match;
                    tok = two ? INF_TK_LIT2 | (sym << 8) | (sym2 << 16) : INF_TK_LIT1 | (sym << 8);
                    o += two ? 2u : 1u;
                    const uint32_t used = two ? l1 + l3 : l1;   // <= 30
                    buf >>= used;
                    cnt -= used;
                    after_match = false;
                }
            } else if (sym == 256u) {
                buf >>= l1;
                cnt -= l1;
                after_match = false;
                tok = inf_special(INF_TS_FLUSH, 0);
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
                    tok = INF_TK_MATCH | ((mlen - 3u) << 2) | ((mdist - 1u) << 10);
                    o += mlen;
                    after_match = true;
                }
            }
            if (mode == DONE) { p = in_beg; tok = inf_special(INF_TS_END, 0); }
        } else if (npend != 0u && room) {
            have = true;
            tok = ptok0;
            ptok0 = ptok1;
            ptok1 = ptok2;
            --npend;
        }
        if (have) {
            inf_q_write(q + tslot * 64u, tok);
            tslot = tslot + 1u == (uint32_t)INF_QCAP ? 0u : tslot + 1u;
            ++tail;
            inf_q_write(q_tail, tail);
        }
        // ---- refill from the window ----
        if (want_in) { win_put(win_hi, in16); win_hi += 16u; }
        if (mode != DONE) {
            const uint32_t sh = poff & 3u;
            const uint64_t nw = (uint64_t)__builtin_amdgcn_alignbyte(wd1, wd0, sh) | ((uint64_t)__builtin_amdgcn_alignbyte(wd2, wd1, sh) << 32);
            buf |= nw << cnt;
            p += (63u - cnt) >> 3;
            cnt |= 56u;
            if (p > in_end + 16) {                         // ran off the member's input
                err = 1; mode = DONE; p = in_beg;
                pend_push(inf_special(INF_TS_END, 0));
            }
        }
    }
    if (mine) job.status[m] = err;
}

// ---- the writer wave --------------------------------------------------------------------------------------------------
// One step = one iteration of gd_inflate_kernel's output half, with ONE difference: the source of the chunk a step plans is
// asked for at the TOP of that step -- what the plan needs (the token, the bytes left of a match, its distance, the output
// position) is known before anything touches T -- and used by the NEXT step, so a load has a whole step of appending, ring
// writes and block stores to arrive in (without the decode in front of it, the writer would otherwise wait for every load
// it has just issued: 50.5 ms against the decoder's own 39.8, profiles/r12k_..., r12l_...).  `fl` is then the one of the
// step before: a source that is completely stored only after this step's block store is read from the ring instead, where
// it still is (it begins less than 112 bytes below what the ring holds when it is read).  The steps alternate between two
// sets of load registers (cl_in: asked for by the step before; cl_out: asked for now): a loaded value that had to be COPIED
// at the loop's back edge would be waited for there.
__device__ __forceinline__ void inflate_pair_write(const InflateJob& job, uint8_t* s_tbl, const int lane, const uint32_t m, const bool mine)
{
    const uint32_t mm = mine ? m : 0u;
    const uint8_t* const in_beg = job.comp + job.in_off[mm];
    uint8_t* const out = job.out + job.out_off[mm];
    const uint32_t* const q = reinterpret_cast<const uint32_t*>(s_tbl + INF_Q) + lane;
    const uint32_t* const q_tail = reinterpret_cast<const uint32_t*>(s_tbl + INF_QTAIL) + lane;
    uint32_t* const q_head = reinterpret_cast<uint32_t*>(s_tbl + INF_QHEAD) + lane;

    enum : uint32_t { IDLE = 0, COPY = 1, FIN = 3 };
    uint32_t wmode = mine ? IDLE : FIN;
    uint32_t o = 0;                                        // bytes of the member produced (some still in T / C)
    uint32_t rem = 0, deff = 16;                           // a match in progress: bytes left, source distance (>= 16)
    uint32_t pend = 0;                                     // trailing bytes of T not in the ring yet (literals)
    // a chunk waits for its append: [co, co + cn); csmall: built from T (cs); cmem: its source was asked for from memory (else
    // it is read from the ring, at ring offset csa)
    bool cpend = false, csmall = false, cmem = false;
    uint32_t co = 0, cn = 0, csa = 0;
    inf_v4 T = {0, 0, 0, 0}, cs = {0, 0, 0, 0};
    uint32_t E0 = 0;                                       // the four bytes before T: bytes [o - 20, o - 16)
    uint32_t head = 0, hslot = 0, st_off = 0;
    uint32_t it = 0;
    // the member's output ring (see gd_inflate_kernel)
    uint32_t* const s_ring = reinterpret_cast<uint32_t*>(s_tbl + INF_RING) + lane;
    const uint32_t obase = (uint32_t)(reinterpret_cast<uintptr_t>(out) & 63u);
    uint8_t* const out_al = out - obase;
    uint32_t fl = 0;
    auto ring_byte = [&](uint32_t a) -> uint32_t { return (s_ring[((a >> 2) & 31u) * 64u] >> (8u * (a & 3u))) & 0xffu; };
    auto ring_bytes_out = [&](uint32_t lo, uint32_t hi) { for (uint32_t a = lo; a < hi; ++a) out_al[a] = (uint8_t)ring_byte(a); };

    // false: every lane has finished
    auto step = [&](const inf_v4& cl_in, inf_v4& cl_out) -> bool {
        const uint64_t live = __ballot(wmode != FIN);
        if (live == 0) return false;
        if (++it >= (1u << 23)) return false;              // (the decoder's backstop ends the writer through INF_QABORT long before)
        __builtin_amdgcn_s_waitcnt(0x0F70);                // cl_in has arrived (vmcnt 0: and nothing older is in flight)
        // ---- a token, if this lane may execute one ----
        const uint32_t tail = inf_q_read(q_tail);
        const bool avail = wmode == IDLE && head != tail && tail != INF_QABORT;
        if (tail == INF_QABORT && wmode == IDLE) wmode = FIN;
        const uint32_t tok = avail ? inf_q_read(q + hslot * 64u) : 0u;
        if (avail) {                                       // the slot is free again (the token is in a register)
            hslot = hslot + 1u == (uint32_t)INF_QCAP ? 0u : hslot + 1u;
            ++head;
            inf_q_write(q_head, head);
        }
        if (job.probe & 8u) {                              // MEASUREMENT ONLY: tokens are thrown away -- the decoder wave's own pace
            if (avail && (tok & 15u) == inf_special(INF_TS_END, 0)) wmode = FIN;
            return true;
        }
        // ---- the chunk this step is going to plan (below, once T is up to date), and the load of its source ----
        const uint32_t kind = tok & 3u;
        const bool start = avail && kind == INF_TK_MATCH;
        const uint32_t mdist = start ? ((tok >> 10) & 0x7fffu) + 1u : 0u;
        const bool p_copy = wmode == COPY || start;
        const bool p_small = start && mdist < 16u;
        // (a period below 16 repeats with its next multiple that is >= 16)
        const uint32_t p_deff = start ? (p_small ? 16u + (uint32_t)((0xECA8642052402000ull >> (4u * mdist)) & 15u) : mdist) : deff;
        const uint32_t p_sa = obase + o - p_deff;          // where the chunk's source begins (ao); it ends at or before o
        const bool p_mem = p_copy && !p_small && p_sa + 16u <= fl && !(job.probe & 1u);   // completely stored
        cl_out = inf_v4{0, 0, 0, 0};
        if (p_mem) cl_out = inf_load16_stream(out + (o - p_deff));

        // ---- the chunk planned by the step before goes into T ----
        inf_v4 cr = {0, 0, 0, 0};
        if (cpend && !csmall && !cmem) {
            const uint32_t rj = csa >> 2, rs = csa & 3u;
            const uint32_t r0 = s_ring[((rj + 0u) & 31u) * 64u], r1 = s_ring[((rj + 1u) & 31u) * 64u], r2 = s_ring[((rj + 2u) & 31u) * 64u],
                           r3 = s_ring[((rj + 3u) & 31u) * 64u], r4 = s_ring[((rj + 4u) & 31u) * 64u];
            cr.x = __builtin_amdgcn_alignbyte(r1, r0, rs);
            cr.y = __builtin_amdgcn_alignbyte(r2, r1, rs);
            cr.z = __builtin_amdgcn_alignbyte(r3, r2, rs);
            cr.w = __builtin_amdgcn_alignbyte(r4, r3, rs);
        }
        const bool cp = cpend;
        const inf_v4 c = csmall ? cs : (cmem ? cl_in : cr);
        if (cp) inf_append5(E0, T, c, cn);
        cpend = false;

        // ---- this step's token ----
        bool flush = false;
        if (avail) {
            if (kind <= INF_TK_LIT2) {
                const bool two = kind == INF_TK_LIT2;
                const uint32_t nb = two ? 2u : 1u, b1 = (tok >> 8) & 0xffu, b2 = (tok >> 16) & 0xffu;
                E0 = __builtin_amdgcn_alignbyte(T.x, E0, nb);
                T.x = __builtin_amdgcn_alignbyte(T.y, T.x, nb);
                T.y = __builtin_amdgcn_alignbyte(T.z, T.y, nb);
                T.z = __builtin_amdgcn_alignbyte(T.w, T.z, nb);
                T.w = two ? (T.w >> 16) | (b1 << 16) | (b2 << 24) : (T.w >> 8) | (b1 << 24);
                o += nb;
                pend += nb;
                flush = pend >= 15u;                       // (E0 and T hold twenty bytes: a pair on top of fourteen pending ones fits)
            } else if (kind == INF_TK_MATCH) {
                rem = ((tok >> 2) & 0xffu) + 3u;
                flush = pend != 0u;
                wmode = COPY;
            } else {
                const uint32_t sub = (tok >> 2) & 3u, arg = tok >> 4;
                if (sub == INF_TS_END) { flush = pend != 0u; wmode = FIN; }
                else if (sub == INF_TS_FLUSH) flush = pend != 0u;
                else if (sub == INF_TS_STORED_OFF) st_off = arg;
                else {
                    // a stored block (behind a FLUSH or at the member's start: nothing is pending, no chunk in flight): its
                    // bytes go straight to memory -- what is still only in the ring goes first -- and T, E0 and the ring
                    // restart behind it (see gd_inflate_kernel)
                    ring_bytes_out(fl > obase ? fl : obase, obase + o);
                    for (uint32_t k = 0; k < arg; ++k) out[o++] = in_beg[st_off + k];
                    uint32_t w[5] = {0, 0, 0, 0, 0};
                    for (uint32_t k = 0; k < 20 && k < o; ++k)
                        w[4 - (k >> 2)] |= (uint32_t)out[o - 1 - k] << (8 * (3 - (k & 3)));
                    E0 = w[0];
                    T = inf_v4{w[1], w[2], w[3], w[4]};
                    fl = (obase + o) & ~63u;
                    for (uint32_t a = fl >= obase + 16u ? fl - 16u : obase; a < obase + o; ++a) {
                        uint32_t& d = s_ring[((a >> 2) & 31u) * 64u];
                        d = (d & ~(0xffu << (8u * (a & 3u)))) | ((uint32_t)out_al[a] << (8u * (a & 3u)));
                    }
                }
            }
        }

        // ---- what this step produced goes into the ring; the 64-byte block the output has passed leaves for memory ----
        if (flush || cp) {
            const uint32_t ao = obase + o;
            const uint32_t sh = (4u - (ao & 3u)) & 3u;
            const uint32_t jw = (ao - 20u + sh) >> 2;
            s_ring[((jw + 0u) & 31u) * 64u] = __builtin_amdgcn_alignbyte(T.x, E0, sh);
            s_ring[((jw + 1u) & 31u) * 64u] = __builtin_amdgcn_alignbyte(T.y, T.x, sh);
            s_ring[((jw + 2u) & 31u) * 64u] = __builtin_amdgcn_alignbyte(T.z, T.y, sh);
            s_ring[((jw + 3u) & 31u) * 64u] = __builtin_amdgcn_alignbyte(T.w, T.z, sh);
            s_ring[((jw + 4u) & 31u) * 64u] = __builtin_amdgcn_alignbyte(0u, T.w, sh);
            if (ao >= fl + 64u) {
                const uint32_t j0 = (fl >> 2) & 31u;       // 0 or 16
                if (job.probe & 2u) {
                } else if (fl >= obase) {
#pragma unroll
                    for (uint32_t qd = 0; qd < 4u; ++qd) {
                        inf_v4 v;
                        v.x = s_ring[(j0 + 4u * qd + 0u) * 64u]; v.y = s_ring[(j0 + 4u * qd + 1u) * 64u];
                        v.z = s_ring[(j0 + 4u * qd + 2u) * 64u]; v.w = s_ring[(j0 + 4u * qd + 3u) * 64u];
                        inf_store16(out_al + fl + 16u * qd, v);
                    }
                } else {
                    ring_bytes_out(obase, fl + 64u);       // the member starts inside this block: the bytes in front of it are another member's
                }
                fl += 64u;
            }
        }
        if (flush) pend = 0;

        // ---- the plan: the chunk whose source was asked for above (o has not moved since: a step with a plan has no literals) ----
        csmall = false;
        if (p_copy) {
            const uint32_t n = rem < 16u ? rem : 16u;
            if (p_small) {
                // period mdist, from the last mdist bytes of T
                const inf_v4 s0 = *reinterpret_cast<const inf_v4*>(s_tbl + INF_PERM + mdist * 32);
                const inf_v4 s1 = *reinterpret_cast<const inf_v4*>(s_tbl + INF_PERM + mdist * 32 + 16);
                cs.x = __builtin_amdgcn_perm(T.y, T.x, s0.x) | __builtin_amdgcn_perm(T.w, T.z, s1.x);
                cs.y = __builtin_amdgcn_perm(T.y, T.x, s0.y) | __builtin_amdgcn_perm(T.w, T.z, s1.y);
                cs.z = __builtin_amdgcn_perm(T.y, T.x, s0.z) | __builtin_amdgcn_perm(T.w, T.z, s1.z);
                cs.w = __builtin_amdgcn_perm(T.y, T.x, s0.w) | __builtin_amdgcn_perm(T.w, T.z, s1.w);
                csmall = true;
            }
            deff = p_deff;
            cmem = p_mem;
            csa = p_sa;
            co = o;
            cn = n;
            o += n;
            rem -= n;
            cpend = true;
            wmode = rem == 0u ? IDLE : COPY;
        }
        return true;
    };
    inf_v4 cl_a = {0, 0, 0, 0}, cl_b = {0, 0, 0, 0};
    for (;;) {
        if (!step(cl_a, cl_b)) break;
        if (!step(cl_b, cl_a)) break;
    }
    // the member's last bytes: what never completed a 64-byte block (after an error: of what was produced)
    if (mine) ring_bytes_out(fl > obase ? fl : obase, obase + o);
}

__global__ __launch_bounds__(2 * INF_LANES, 2) void gd_inflate_pair_kernel(InflateJob job)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_tbl[INF_PAIR_LDS_BYTES];
    const int lane = threadIdx.x & (INF_LANES - 1);
    // selectors of a chunk with period d (see gd_inflate_kernel)
    for (int i = threadIdx.x; i < 16 * 8; i += 2 * INF_LANES) {
        const int d = i >> 3, half = (i >> 2) & 1, j = i & 3;
        uint32_t w = 0;
        for (int t = 0; t < 4; ++t) {
            const int b = 4 * j + t, s = d ? 16 - d + b % d : 0;
            const uint32_t sel = half == 0 ? (s < 8 ? (uint32_t)s : 0x0cu) : (s >= 8 ? (uint32_t)(s - 8) : 0x0cu);
            w |= sel << (8 * t);
        }
        reinterpret_cast<uint32_t*>(s_tbl + INF_PERM)[i] = w;
    }
    if (threadIdx.x < INF_LANES) {
        reinterpret_cast<uint32_t*>(s_tbl + INF_QTAIL)[lane] = 0;
        reinterpret_cast<uint32_t*>(s_tbl + INF_QHEAD)[lane] = 0;
    }
    __syncthreads();
    const uint32_t m = blockIdx.x * INF_LANES + lane;
    const bool mine = m < job.n;
    if (threadIdx.x < INF_LANES) inflate_pair_decode(job, s_tbl, lane, m, mine);
    else inflate_pair_write(job, s_tbl, lane, m, mine);
}

}  // namespace gd
