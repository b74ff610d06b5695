// inflate_wave_sim.cpp -- MEASUREMENT TOOL (CPU, no GPU): what a workgroup-per-member inflate would do, step by step, on the
// members of a BAM file -- the numbers the design of goleft_amd/csrc/gd_inflate_wave.hpp stands on.
//
//   g++ -O2 -std=c++17 -o /tmp/inflate_wave_sim tools/inflate_wave_sim.cpp -lz
//   /tmp/inflate_wave_sim x.bam [members=64] [skip=20]
//
// For every Huffman block of the sampled members, with NL lanes (64 .. 512) that each take 1/NL of the block's remaining bits:
//   * pass A: every lane decodes its subsequence from its (wrong) boundary start to the first symbol that starts in the next
//     lane's subsequence; then lanes restart from their left neighbour's crossing until nothing changes: passes needed,
//     lock-step symbol steps per pass (the longest lane's);
//   * pass B1: the same decode once more with the output offsets known (literals written, matches left as tokens);
//   * pass B2: the matches in OUTPUT order in batches of BW (64 / 256): a match is copied once every earlier match of its batch
//     whose destination overlaps its source has been copied -- the rounds a batch needs are the depth of that dependency
//     graph; 16-byte chunk iterations per batch (the longest match's).
#include <zlib.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

static const int LEN_BASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const int LEN_EXTRA[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const int DIST_BASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const int DIST_EXTRA[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
static const int ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

struct Bits {
    const uint8_t* d; uint64_t n;                           // n: bits
    int bit(uint64_t p) const { return p < n ? (d[p >> 3] >> (p & 7)) & 1 : 0; }
    uint32_t bits(uint64_t p, int k) const { uint32_t v = 0; for (int i = 0; i < k; ++i) v |= (uint32_t)bit(p + i) << i; return v; }
};
struct Canon {                                              // canonical code: first code / first index / count per length
    int count[16] = {0}, first[16] = {0}, offs[16] = {0}; std::vector<int> sorted;
    void build(const std::vector<int>& lens)
    {
        for (int l : lens) count[l]++;
        count[0] = 0;
        int code = 0, o = 0;
        for (int l = 1; l < 16; ++l) { code = (code + count[l - 1]) << 1; first[l] = code; offs[l] = o; o += count[l]; }
        sorted.assign(o, 0);
        std::vector<int> nx(offs, offs + 16);
        for (size_t s = 0; s < lens.size(); ++s) if (lens[s]) sorted[nx[lens[s]]++] = (int)s;
    }
    int sym(const Bits& b, uint64_t& p) const              // -1: no code
    {
        int code = 0;
        for (int l = 1; l < 16; ++l) {
            if (p + l > b.n) return -1;
            code = (code << 1) | b.bit(p + l - 1);
            if (code - first[l] < count[l] && code >= first[l]) { p += l; return sorted[offs[l] + code - first[l]]; }
        }
        return -1;
    }
};
struct Step { int kind; int n; int dist; };                 // kind: 0 literal, 1 match, 2 eob, -1 invalid
static Step step(const Bits& b, uint64_t& p, const Canon& L, const Canon& D)
{
    uint64_t q = p;
    int s = L.sym(b, q);
    if (s < 0 || s > 285) return {-1, 0, 0};
    if (s < 256) { p = q; return {0, 1, 0}; }
    if (s == 256) { p = q; return {2, 0, 0}; }
    int ls = s - 257;
    if (q + LEN_EXTRA[ls] > b.n) return {-1, 0, 0};
    int n = LEN_BASE[ls] + (int)b.bits(q, LEN_EXTRA[ls]);
    q += LEN_EXTRA[ls];
    int d = D.sym(b, q);
    if (d < 0 || d > 29 || q + DIST_EXTRA[d] > b.n) return {-1, 0, 0};
    int dist = DIST_BASE[d] + (int)b.bits(q, DIST_EXTRA[d]);
    p = q + DIST_EXTRA[d];
    return {1, n, dist};
}

struct Match { uint32_t dst, len, dist; };
struct Acc {
    double members = 0, blocks = 0, symbols = 0, matches = 0, out = 0, in = 0, hdr_syms = 0;
    std::map<int, double> passes, stepsA, stepsB1;          // by NL
    std::map<int, double> b2_batches, b2_rounds, b2_chunks; // by BW
    std::map<int, double> b2_rounds_frontier, b2_r1, b2_r2;
    std::map<int, double> hist_len, hist_dist;
    double local64 = 0, local256 = 0, pieces = 0, pc_batches = 0, pc_exact = 0, pc_r1 = 0, pc_r2 = 0, pc_r3 = 0;
};

int main(int argc, char** argv)
{
    if (argc < 2) { fprintf(stderr, "usage: %s x.bam [members] [skip]\n", argv[0]); return 2; }
    const int want = argc > 2 ? atoi(argv[2]) : 64, skip = argc > 3 ? atoi(argv[3]) : 20;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 1;
    fseek(f, 0, SEEK_END); long sz = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> data(sz);
    if (fread(data.data(), 1, sz, f) != (size_t)sz) return 1;
    fclose(f);
    Acc A;
    const int NLS[4] = {64, 128, 256, 512};
    size_t off = 0; int idx = 0;
    while (off + 18 < data.size() && A.members < want) {
        const uint32_t xlen = data[off + 10] | (data[off + 11] << 8), bsize = data[off + 16] | (data[off + 17] << 8);
        const uint8_t* pay = &data[off + 12 + xlen];
        const uint32_t plen = bsize + 1 - 8 - 12 - xlen;
        uint32_t isize; memcpy(&isize, &data[off + bsize + 1 - 4], 4);
        off += bsize + 1;
        if (idx++ % skip || isize < 1000) continue;
        Bits b{pay, (uint64_t)plen * 8};
        A.members++; A.in += plen; A.out += isize;
        uint64_t pos = 0; uint32_t opos = 0;
        std::vector<Match> all;                             // every match of the member, output order
        for (;;) {
            const int final = b.bit(pos), typ = (int)b.bits(pos + 1, 2);
            pos += 3;
            if (typ == 0) { pos = (pos + 7) & ~7ull; const uint32_t n = b.bits(pos, 16); pos += 32 + 8ull * n; opos += n; if (final) break; continue; }
            Canon L, D;
            if (typ == 1) {
                std::vector<int> l(288, 8); for (int i = 144; i < 256; ++i) l[i] = 9; for (int i = 256; i < 280; ++i) l[i] = 7;
                L.build(l); D.build(std::vector<int>(30, 5));
            } else {
                const int nl = (int)b.bits(pos, 5) + 257, nd = (int)b.bits(pos + 5, 5) + 1, nc = (int)b.bits(pos + 10, 4) + 4;
                pos += 14;
                std::vector<int> cl(19, 0);
                for (int k = 0; k < nc; ++k) { cl[ORDER[k]] = (int)b.bits(pos, 3); pos += 3; }
                Canon C; C.build(cl);
                std::vector<int> lens;
                while ((int)lens.size() < nl + nd) {
                    const int s = C.sym(b, pos);
                    A.hdr_syms++;
                    if (s < 16) lens.push_back(s);
                    else if (s == 16) { int r = 3 + (int)b.bits(pos, 2); pos += 2; int pv = lens.back(); while (r--) lens.push_back(pv); }
                    else if (s == 17) { int r = 3 + (int)b.bits(pos, 3); pos += 3; while (r--) lens.push_back(0); }
                    else { int r = 11 + (int)b.bits(pos, 7); pos += 7; while (r--) lens.push_back(0); }
                }
                L.build(std::vector<int>(lens.begin(), lens.begin() + nl));
                D.build(std::vector<int>(lens.begin() + nl, lens.end()));
            }
            A.blocks++;
            // the truth: the block's symbols
            const uint64_t body = pos;
            std::vector<uint64_t> starts; std::vector<Step> syms;
            for (;;) {
                starts.push_back(pos);
                const Step s = step(b, pos, L, D);
                if (s.kind < 0) { fprintf(stderr, "bad stream\n"); return 1; }
                syms.push_back(s);
                if (s.kind == 2) break;
                if (s.kind == 1) { all.push_back({opos, (uint32_t)s.n, (uint32_t)s.dist}); A.hist_len[s.n <= 16 ? 16 : s.n <= 32 ? 32 : s.n <= 64 ? 64 : 258]++;
                    A.hist_dist[s.dist < 16 ? 16 : s.dist < 64 ? 64 : s.dist < 256 ? 256 : s.dist < 1024 ? 1024 : s.dist < 4096 ? 4096 : 32768]++; }
                opos += s.n;
            }
            const uint64_t eob_end = pos;
            A.symbols += syms.size();
            // ---- pass A with NL lanes over [body, end of the member's payload) ----
            for (int NL : NLS) {
                const uint64_t total = b.n - body;
                uint64_t S = (total + NL - 1) / NL; if (S < 64) S = 64;
                std::vector<uint64_t> start(NL), cross(NL), want_s(NL);
                std::vector<int> nsym(NL, 0), state(NL, 0);   // state: 0 crossed, 1 eob, 2 bad, 3 inactive
                auto run = [&](int k) {
                    uint64_t p = start[k]; const uint64_t bound = body + (uint64_t)(k + 1) * S;
                    int n = 0;
                    state[k] = 0;
                    while (p < bound) {
                        if (p >= b.n) { state[k] = 2; break; }
                        const Step s = step(b, p, L, D);
                        ++n;
                        if (s.kind < 0) { state[k] = 2; break; }
                        if (s.kind == 2) { state[k] = 1; break; }
                    }
                    cross[k] = p; nsym[k] = n;
                };
                int passes = 0; double steps = 0;
                for (int k = 0; k < NL; ++k) { start[k] = body + (uint64_t)k * S; if (start[k] >= b.n) { state[k] = 3; nsym[k] = 0; } else run(k); }
                { int mx = 0; for (int k = 0; k < NL; ++k) mx = std::max(mx, nsym[k]); steps += mx; passes = 1; }
                for (;;) {
                    bool any = false; int mx = 0;
                    std::vector<int> redo;
                    for (int k = 1; k < NL; ++k) {
                        // (a lane behind one that met the end of the block or an invalid code -- with a wrong start that proves
                        // nothing -- keeps its own chain; which lanes count is decided once nothing changes any more)
                        if (state[k - 1] != 0) continue;
                        if (state[k] == 3 || start[k] != cross[k - 1]) { want_s[k] = cross[k - 1]; redo.push_back(k); }
                    }
                    // (all lanes of a pass restart from the crossings of the PREVIOUS pass: lock step)
                    for (int k : redo) { start[k] = want_s[k]; }
                    for (int k : redo) { run(k); mx = std::max(mx, nsym[k]); any = true; }
                    if (!any) break;
                    if (!redo.empty()) { steps += mx; ++passes; }
                }
                // the chain must be the truth
                { size_t t = 0; for (int k = 0; k < NL && state[k] != 3 && (k == 0 || state[k - 1] == 0); ++k) { while (t < starts.size() && starts[t] < start[k]) ++t; if (t >= starts.size() || starts[t] != start[k]) { fprintf(stderr, "chain is not the truth (NL %d lane %d)\n", NL, k); return 1; } } }
                A.passes[NL] += passes; A.stepsA[NL] += steps;
                int mx = 0; for (int k = 0; k < NL && (k == 0 || state[k - 1] == 0); ++k) if (state[k] != 3) mx = std::max(mx, nsym[k]);
                A.stepsB1[NL] += mx;
            }
            (void)eob_end;
            if (final) break;
        }
        A.matches += all.size();
        // ---- the same with every match cut into pieces of at most 16 bytes (a lane copies ONE chunk): 64 pieces per batch ----
        {
            std::vector<Match> pc; std::vector<std::pair<uint32_t, uint32_t>> srcr;
            for (const Match& m : all) {
                uint32_t rem = m.len, p = m.dst;
                while (rem) { const uint32_t n = rem > 16 ? (rem - 16 < 3 ? 13 : 16) : rem; pc.push_back({p, n, m.dist});
                    // a piece of a match with a period below 16 is built from the bytes in front of the MATCH (its pieces do not
                    // depend on each other); any other piece reads dst - dist
                    if (m.dist < 16 && getenv("PERIODIC")) srcr.push_back({m.dst - m.dist, m.dst}); else srcr.push_back({p - m.dist, std::min(p, p - m.dist + n)});
                    p += n; rem -= n; }
            }
            A.pieces += pc.size();
            for (size_t i0 = 0; i0 < pc.size(); i0 += 64) {
                const size_t i1 = std::min(pc.size(), i0 + 64), n = i1 - i0;
                // exact depth
                std::vector<int> level(n, 1); int depth = 1;
                for (size_t i = 0; i < n; ++i) {
                    const uint32_t s = srcr[i0 + i].first, e = srcr[i0 + i].second;
                    for (size_t j = 0; j < i; ++j) { const Match& q = pc[i0 + j]; if (q.dst < e && q.dst + q.len > s) level[i] = std::max(level[i], level[j] + 1); }
                    depth = std::max(depth, level[i]);
                }
                A.pc_batches++; A.pc_exact += depth;
                for (int rule = 1; rule <= 3; ++rule) {
                    std::vector<char> done(n, 0); size_t open_ = n; int rounds = 0;
                    while (open_) {
                        ++rounds;
                        size_t first = 0; while (done[first]) ++first;
                        const uint32_t F = pc[i0 + first].dst;
                        std::vector<char> go(n, 0); long h = -1;
                        for (size_t i = 0; i < n; ++i) if (!done[i]) {
                            const uint32_t s = srcr[i0 + i].first, e = srcr[i0 + i].second;
                            bool ready = e <= F || i == first;
                            if (!ready && rule == 2 && (h < 0 || s >= pc[i0 + h].dst + pc[i0 + h].len)) ready = true;
                            if (!ready && rule == 3 && (i == 0 || s >= pc[i0 + i - 1].dst + pc[i0 + i - 1].len)) ready = true;
                            go[i] = ready; h = (long)i;
                        }
                        for (size_t i = 0; i < n; ++i) if (go[i]) { done[i] = 1; --open_; }
                    }
                    (rule == 1 ? A.pc_r1 : rule == 2 ? A.pc_r2 : A.pc_r3) += rounds;
                }
            }
        }
        // ---- pass B2: batches of BW matches in output order ----
        for (int BW : {64, 256}) {
            for (size_t i0 = 0; i0 < all.size(); i0 += BW) {
                const size_t i1 = std::min(all.size(), i0 + BW);
                std::vector<int> level(i1 - i0, 1);
                int depth = 1, chunks = 1, fr_rounds = 0;
                for (size_t i = i0; i < i1; ++i) {
                    const uint32_t s = all[i].dst - all[i].dist, e = std::min(all[i].dst, s + all[i].len);
                    int lv = 1;
                    for (size_t j = i0; j < i; ++j)
                        if (all[j].dst < e && all[j].dst + all[j].len > s) lv = std::max(lv, level[j - i0] + 1);
                    level[i - i0] = lv; depth = std::max(depth, lv);
                    chunks = std::max(chunks, (int)(all[i].len + 15) / 16);
                }
                // the conservative frontier rule: ready when the source ends at or below the first unresolved destination
                { std::vector<char> done(i1 - i0, 0); size_t left = i1 - i0;
                  while (left) { ++fr_rounds; uint32_t F = 0xffffffffu; for (size_t i = i0; i < i1; ++i) if (!done[i - i0]) { F = all[i].dst; break; }
                      for (size_t i = i0; i < i1; ++i) if (!done[i - i0]) { const uint32_t s = all[i].dst - all[i].dist, e = std::min(all[i].dst, s + all[i].len);
                          if (e <= F || all[i].dst == F) { done[i - i0] = 2; } }
                      for (size_t i = i0; i < i1; ++i) if (done[i - i0] == 2) { done[i - i0] = 1; --left; } } }
                // one 16-byte chunk per lane and round; a lane starts when rule 1 (its source ends at or below the first
                // unresolved destination) or rule 2 (its source begins at or above the end of the highest unresolved lane below
                // it) holds, and counts as resolved once its last chunk is written
                for (int rule = 1; rule <= 2; ++rule) {
                    const size_t n = i1 - i0; std::vector<int> left(n), started(n, 0); size_t open_ = n; int rounds = 0;
                    for (size_t i = 0; i < n; ++i) left[i] = (int)(all[i0 + i].len + 15) / 16;
                    while (open_) {
                        ++rounds;
                        size_t first = 0; while (left[first] == 0) ++first;
                        const uint32_t F = all[i0 + first].dst;
                        std::vector<char> go(n, 0);
                        long h = -1;
                        for (size_t i = 0; i < n; ++i) {
                            if (left[i]) {
                                const Match& m = all[i0 + i]; const uint32_t s = m.dst - m.dist, e = std::min(m.dst, s + m.len);
                                bool ready = started[i] || e <= F || i == first;
                                if (!ready && rule == 2 && h >= 0 && s >= all[i0 + h].dst + all[i0 + h].len) ready = true;
                                if (!ready && rule == 2 && h < 0) ready = true;
                                go[i] = ready;
                                h = (long)i;
                            }
                        }
                        for (size_t i = 0; i < n; ++i) if (go[i]) { started[i] = 1; if (--left[i] == 0) --open_; }
                    }
                    (rule == 1 ? A.b2_r1 : A.b2_r2)[BW] += rounds;
                }
                A.b2_batches[BW]++; A.b2_rounds[BW] += depth; A.b2_chunks[BW] += chunks; A.b2_rounds_frontier[BW] += fr_rounds;
            }
        }
    }
    printf("{\n \"file\": \"%s\", \"members\": %.0f, \"in_bytes_per_member\": %.0f, \"out_bytes_per_member\": %.0f,\n", argv[1], A.members, A.in / A.members, A.out / A.members);
    printf(" \"huffman_blocks_per_member\": %.2f, \"symbols_per_member\": %.0f, \"matches_per_member\": %.0f, \"code_length_symbols_per_member\": %.0f,\n",
           A.blocks / A.members, A.symbols / A.members, A.matches / A.members, A.hdr_syms / A.members);
    printf(" \"match_length_share\": {"); { bool c = false; for (auto& kv : A.hist_len) { printf("%s\"<=%d\": %.3f", c ? ", " : "", kv.first, kv.second / A.matches); c = true; } } printf("},\n");
    printf(" \"match_distance_share\": {"); { bool c = false; for (auto& kv : A.hist_dist) { printf("%s\"<%d\": %.3f", c ? ", " : "", kv.first, kv.second / A.matches); c = true; } } printf("},\n");
    for (int NL : NLS)
        printf(" \"lanes_%d\": {\"passes_A_per_block\": %.2f, \"lockstep_steps_A_per_member\": %.0f, \"lockstep_steps_B1_per_member\": %.0f},\n", NL,
               A.passes[NL] / A.blocks, A.stepsA[NL] / A.members, A.stepsB1[NL] / A.members);
    for (int BW : {64, 256})
        printf(" \"b2_batches_of_%d\": {\"batches_per_member\": %.1f, \"rounds_per_batch_exact\": %.2f, \"rounds_per_batch_frontier_rule\": %.2f, \"chunk_iterations_per_batch\": %.2f, \"chunk_rounds_rule1\": %.2f, \"chunk_rounds_rule1or2\": %.2f}%s\n", BW,
               A.b2_batches[BW] / A.members, A.b2_rounds[BW] / A.b2_batches[BW], A.b2_rounds_frontier[BW] / A.b2_batches[BW], A.b2_chunks[BW] / A.b2_batches[BW], A.b2_r1[BW] / A.b2_batches[BW], A.b2_r2[BW] / A.b2_batches[BW], BW == 64 ? "," : "");
    printf(" ,\"pieces_of_16\": {\"pieces_per_member\": %.0f, \"batches_per_member\": %.1f, \"rounds_exact\": %.2f, \"rounds_rule1\": %.2f, \"rounds_rule1or2\": %.2f, \"rounds_rule1_or_static_gap\": %.2f}\n", A.pieces / A.members, A.pc_batches / A.members, A.pc_exact / A.pc_batches, A.pc_r1 / A.pc_batches, A.pc_r2 / A.pc_batches, A.pc_r3 / A.pc_batches);
    printf("}\n");
    return 0;
}
