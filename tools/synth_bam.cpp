// synth_bam.cpp -- writes a realistic single-contig BAM for the end-to-end timing
// (SURVEY.md section 8d scope iii: BAM file -> BED).  MEASUREMENT TOOL, not product code.
//
//   synth-bam OUT.bam CONTIG LENGTH[,LENGTH...] COVERAGE [SEED] [THREADS]
// (several comma-separated lengths: contigs CONTIG, CONTIG_2, CONTIG_3 ... in one coordinate-sorted file)
//
// 150 bp reads of the SURVEY 8d short-read model (92 % 150M, 5 % soft clip, 2 % one deletion,
// 1 % one insertion; 5 % DUP, 0.3 % other filtered flags, 1 % MAPQ 0), coordinate sorted, WITH
// SEQ and QUAL (random bases, run-structured qualities) so that the file has the size and the
// inflate cost of a real one (~250 B per record before BGZF).  BGZF members are deflated in
// parallel (level 1; libdeflate when the system has it, else zlib; SYNTH_BAM_LEVEL changes the level).  Also writes a "<contig>\t<length>\t..." line to OUT.fa.fai and
// SYNTH_BAM_AUX=1: records as an aligner + duplicate marker leave them -- Illumina-style read names
// (`A00741:188:HGTMNDSX2:3:2437:23511:17018`), mate fields (same reference, a position a fragment away, TLEN) and the
// tags RG:Z NM:C MD:Z AS:C XS:C MC:Z (~ +75 B per record before BGZF): the inflate and the record walk of a real 30x file,
// where the default records (a 14-byte name, no tags) are the cheapest a BAM can hold.
// OUT.bam.bai (exact 16 kb linear index; the binning index is collapsed into bin 0, enough
// for this repository's readers, not for region queries by other tools).
#include <dlfcn.h>
#include <fcntl.h>
#include <unistd.h>
#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <algorithm>
#include <thread>
#include <utility>
#include <vector>

static inline uint64_t mix(uint64_t x)
{
    x += 0x9e3779b97f4a7c15ull;
    x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
    x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
    return x ^ (x >> 31);
}

static void put32(std::vector<uint8_t>& v, uint32_t x) { for (int i = 0; i < 4; ++i) v.push_back((uint8_t)(x >> (8 * i))); }
static void put16(std::vector<uint8_t>& v, uint16_t x) { v.push_back((uint8_t)x); v.push_back((uint8_t)(x >> 8)); }

static int reg2bin(int64_t beg, int64_t end)
{
    --end;
    if (beg >> 14 == end >> 14) return (int)(((1 << 15) - 1) / 7 + (beg >> 14));
    if (beg >> 17 == end >> 17) return (int)(((1 << 12) - 1) / 7 + (beg >> 17));
    if (beg >> 20 == end >> 20) return (int)(((1 << 9) - 1) / 7 + (beg >> 20));
    if (beg >> 23 == end >> 23) return (int)(((1 << 6) - 1) / 7 + (beg >> 23));
    if (beg >> 26 == end >> 26) return (int)(((1 << 3) - 1) / 7 + (beg >> 26));
    return 0;
}

// libdeflate, when the system has its shared library (what htslib itself compresses BAM files with when built against it,
// and three times as fast as zlib here: the genome's 170 GB are deflated on the 16 CPUs' worth of time a GPU box gives a
// container, two thirds of bench.py's run time with zlib); zlib otherwise.  SYNTH_BAM_ZLIB=1 forces zlib.
struct LibDeflate {
    void* (*alloc)(int) = nullptr;
    size_t (*compress)(void*, const void*, size_t, void*, size_t) = nullptr;
    uint32_t (*crc)(uint32_t, const void*, size_t) = nullptr;
    LibDeflate()
    {
        if (getenv("SYNTH_BAM_ZLIB")) return;
        void* h = dlopen("libdeflate.so.0", RTLD_NOW);
        if (!h) return;
        alloc = reinterpret_cast<void* (*)(int)>(dlsym(h, "libdeflate_alloc_compressor"));
        compress = reinterpret_cast<size_t (*)(void*, const void*, size_t, void*, size_t)>(dlsym(h, "libdeflate_deflate_compress"));
        crc = reinterpret_cast<uint32_t (*)(uint32_t, const void*, size_t)>(dlsym(h, "libdeflate_crc32"));
        if (!alloc || !compress || !crc) alloc = nullptr;
    }
    bool ok() const { return alloc != nullptr; }
};
static const LibDeflate g_ld;
static int g_level = 1;
static bool g_aux = false;                     // SYNTH_BAM_AUX=1

// what one worker keeps from flush to flush (the threads themselves are started per flush): its compressor and its scratch
struct Worker {
    void* comp = nullptr;
    std::vector<uint8_t> c, one;
};

static void bgzf_member(const uint8_t* data, size_t n, std::vector<uint8_t>* out, Worker* wk)
{
    out->clear();
    std::vector<uint8_t>& c = wk->c;                     // (3 million members otherwise allocate and zero 74 KB each)
    if (c.size() < n + n / 8 + 128) c.resize(n + n / 8 + 128);
    size_t clen = 0;
    uint32_t crc = 0;
    if (g_ld.ok()) {
        if (!wk->comp) wk->comp = g_ld.alloc(g_level);
        clen = wk->comp ? g_ld.compress(wk->comp, data, n, c.data(), c.size()) : 0;
        crc = g_ld.crc(0, data, n);
    }
    if (clen == 0) {
        z_stream zs;
        memset(&zs, 0, sizeof zs);
        deflateInit2(&zs, g_level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
        zs.next_in = const_cast<uint8_t*>(data);
        zs.avail_in = (uInt)n;
        zs.next_out = c.data();
        zs.avail_out = (uInt)c.size();
        deflate(&zs, Z_FINISH);
        clen = c.size() - zs.avail_out;
        deflateEnd(&zs);
        crc = (uint32_t)crc32(crc32(0L, nullptr, 0), data, (uInt)n);
    }
    static const uint8_t hdr[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
    out->insert(out->end(), hdr, hdr + 16);
    put16(*out, (uint16_t)(clen + 25));
    out->insert(out->end(), c.begin(), c.begin() + (long)clen);
    put32(*out, crc);
    put32(*out, (uint32_t)n);
}

int main(int argc, char** argv)
{
    if (argc < 5) { fprintf(stderr, "usage: synth-bam OUT.bam CONTIG LENGTH COVERAGE [SEED] [THREADS]\n"); return 2; }
    const std::string path = argv[1], contig0 = argv[2];
    std::vector<int64_t> lens;
    for (const char* q = argv[3]; *q;) { lens.push_back(atoll(q)); while (*q && *q != ',') ++q; if (*q == ',') ++q; }
    std::vector<std::string> names;
    for (size_t k = 0; k < lens.size(); ++k) names.push_back(k == 0 ? contig0 : contig0 + "_" + std::to_string(k + 1));
    const double cov = atof(argv[4]);
    const uint64_t seed = argc > 5 ? strtoull(argv[5], nullptr, 10) : 1;
    int threads = argc > 6 ? atoi(argv[6]) : (int)std::thread::hardware_concurrency();
    if (threads < 1) threads = 1;
    if (const char* lv = getenv("SYNTH_BAM_LEVEL")) g_level = std::max(1, std::min(9, atoi(lv)));
    if (const char* ax = getenv("SYNTH_BAM_AUX")) g_aux = atoi(ax) != 0;
    const int RL = 150;
    const int fd = open(path.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) { perror("open"); return 1; }

    // uncompressed BAM bytes not yet written: a plain buffer (a std::vector zero-fills what resize() adds -- 4 GB per
    // batch on one thread, a third of the tool's run time on a 256-core host)
    struct Raw {
        uint8_t* p = nullptr; size_t n = 0, cap = 0;
        size_t size() const { return n; }
        uint8_t* data() { return p; }
        void grow_to(size_t m)
        {
            if (m > cap) {
                size_t nc = std::max(m, cap + cap / 2);
                uint8_t* q = static_cast<uint8_t*>(malloc(nc));
                if (!q) { perror("malloc"); exit(1); }
                if (n) memcpy(q, p, n);
                free(p);
                p = q; cap = nc;
            }
            n = m;
        }
        void drop_front(size_t k) { if (k < n) memmove(p, p + k, n - k); n -= k; }
    } raw;
    std::vector<uint8_t> hdr;
    std::string text = "@HD\tVN:1.6\tSO:coordinate\n";
    for (size_t k = 0; k < lens.size(); ++k) text += "@SQ\tSN:" + names[k] + "\tLN:" + std::to_string(lens[k]) + "\n";
    text += "@RG\tID:rg1\tSM:synth\n";
    hdr.insert(hdr.end(), {'B', 'A', 'M', 1});
    put32(hdr, (uint32_t)text.size());
    hdr.insert(hdr.end(), text.begin(), text.end());
    put32(hdr, (uint32_t)lens.size());
    for (size_t k = 0; k < lens.size(); ++k) {
        put32(hdr, (uint32_t)names[k].size() + 1);
        hdr.insert(hdr.end(), names[k].begin(), names[k].end());
        hdr.push_back(0);
        put32(hdr, (uint32_t)lens[k]);
    }

    raw.grow_to(hdr.size());
    memcpy(raw.data(), hdr.data(), hdr.size());

    const size_t BLK = 0xff00;
    uint64_t out_bytes = 0;
    std::vector<uint32_t> csize;               // compressed size of every data member, in file order
    std::vector<std::vector<uint64_t>> lins(lens.size());      // per contig: uncompressed offset of the first record per 16 kb window
    for (size_t k = 0; k < lens.size(); ++k) lins[k].assign((size_t)(lens[k] >> 14) + 1, ~0ull);
    uint64_t stream_off = 0;                   // uncompressed bytes written to `raw` so far (whole file)
    std::vector<uint64_t> first_rec(lens.size(), ~0ull), after_last(lens.size(), 0);
    int64_t n_total = 0;
    auto flush = [&](bool final) {
        // every thread deflates a CONTIGUOUS range of members into one buffer and writes it with one pwrite at the offset
        // the sizes before it give (one thread's fwrite of 46 GB was a third of the tool's run time)
        const size_t nblk = raw.size() / BLK + ((final && raw.size() % BLK) ? 1 : 0);
        const size_t nt = (size_t)threads;
        // kept from flush to flush: fresh gigabytes per batch mean page faults of 256 threads on one address space
        static std::vector<std::vector<uint8_t>> comp;
        static std::vector<std::vector<uint32_t>> sizes;
        static std::vector<Worker> workers;
        comp.resize(nt); sizes.resize(nt); workers.resize(nt);
        for (size_t t = 0; t < nt; ++t) { comp[t].clear(); sizes[t].clear(); }
        std::vector<std::thread> pool;
        for (size_t t = 0; t < nt; ++t)
            pool.emplace_back([&, t]() {
                std::vector<uint8_t>& one = workers[t].one;
                for (size_t b = nblk * t / nt; b < nblk * (t + 1) / nt; ++b) {
                    const size_t off = b * BLK, len = std::min(BLK, raw.size() - off);
                    bgzf_member(raw.data() + off, len, &one, &workers[t]);
                    comp[t].insert(comp[t].end(), one.begin(), one.end());
                    sizes[t].push_back((uint32_t)one.size());
                }
            });
        for (auto& th : pool) th.join();
        pool.clear();
        std::vector<uint64_t> at(nt + 1, out_bytes);
        for (size_t t = 0; t < nt; ++t) at[t + 1] = at[t] + comp[t].size();
        bool ok = true;
        for (size_t t = 0; t < nt; ++t)
            pool.emplace_back([&, t]() {
                size_t done = 0;
                while (done < comp[t].size()) {
                    const ssize_t w = pwrite(fd, comp[t].data() + done, comp[t].size() - done, (off_t)(at[t] + done));
                    if (w <= 0) { ok = false; return; }
                    done += (size_t)w;
                }
            });
        for (auto& th : pool) th.join();
        if (!ok) { perror("pwrite"); exit(1); }
        out_bytes = at[nt];
        for (size_t t = 0; t < nt; ++t) csize.insert(csize.end(), sizes[t].begin(), sizes[t].end());
        const size_t used = std::min(raw.size(), nblk * BLK);
        raw.drop_front(used);
        stream_off += used;
    };

    // Records are a pure function of (contig, rank): chunks of reads are generated by all threads at once into
    // their own buffers, then laid end to end in `raw` (the stream the BGZF members are cut from) -- the same bytes
    // the one-thread loop wrote, which for a 46 GB genome took six minutes.
    struct Chunk {
        std::vector<uint8_t> bytes;
        std::vector<std::pair<uint32_t, uint64_t>> lin;     // (16 kb window, chunk-local offset of the first record that reaches it)
        uint64_t after_last = 0;
    };
    const int64_t CH = 1 << 16;                              // reads per chunk (~17 MB)
    auto gen_chunk = [&](size_t ctg, int64_t i0, int64_t i1, int64_t n, int64_t span, Chunk* c) {
        // (written through a pointer into a buffer of the largest possible size: a push_back per byte was most of the tool's
        // CPU time once the deflate was libdeflate's)
        std::vector<uint8_t>& buf = c->bytes;
        const size_t worst = (size_t)(i1 - i0) * (g_aux ? 448 : 320);
        if (buf.capacity() < worst) { buf.clear(); buf.reserve(worst); }
        buf.resize(worst);                                   // (no-op after the first chunk of this size)
        uint8_t* const w0 = buf.data();
        uint8_t* w = w0;
        auto p32 = [&](uint32_t x) { memcpy(w, &x, 4); w += 4; };
        auto p16 = [&](uint16_t x) { memcpy(w, &x, 2); w += 2; };
        c->lin.clear();
        uint32_t last_w = 0xffffffffu;
        const int64_t stride = span / n > 0 ? span / n : 1;
        for (int64_t i = i0; i < i1; ++i) {
            const uint64_t h = mix((seed + ctg * 7919) * 0x100000001b3ull + (uint64_t)i);
            int64_t pos = (int64_t)(((__int128)i * span) / n) + (int64_t)(h % (uint64_t)stride);
            if (pos > span) pos = span;
            const uint32_t kind = (uint32_t)((h >> 20) % 10000);
            uint32_t cig[3];
            int nc = 1, ref = RL;
            if (kind < 9200) { cig[0] = (uint32_t)RL << 4; }
            else if (kind < 9700) { const uint32_t k = 1 + (uint32_t)((h >> 40) % 30); cig[0] = k << 4 | 4; cig[1] = (RL - k) << 4; nc = 2; ref = RL - (int)k; }
            else if (kind < 9900) { const uint32_t a = 20 + (uint32_t)((h >> 40) % 100), d = 1 + (uint32_t)((h >> 50) % 10);
                                    cig[0] = a << 4; cig[1] = d << 4 | 2; cig[2] = (RL - a) << 4; nc = 3; ref = RL + (int)d; }
            else { const uint32_t a = 20 + (uint32_t)((h >> 40) % 100), ins = 1 + (uint32_t)((h >> 50) % 10);
                   cig[0] = a << 4; cig[1] = ins << 4 | 1; cig[2] = (RL - a - ins) << 4; nc = 3; ref = RL - (int)ins; }
            const uint32_t fr = (uint32_t)((h >> 8) % 1000);
            uint16_t flag = (h & 1) ? 99 : 147;
            if (fr < 50) flag |= 0x400; else if (fr < 51) flag |= 0x100; else if (fr < 52) flag |= 0x200; else if (fr < 57) flag |= 0x800;
            const uint8_t mapq = ((h >> 12) % 100) == 0 ? 0 : 60;
            char name[64] = "synth.";
            int ln = 6;
            auto put_dec = [&](uint64_t v) {
                char dig[24];
                int nd = 0;
                do { dig[nd++] = (char)('0' + v % 10); v /= 10; } while (v);
                while (nd) name[ln++] = dig[--nd];
            };
            uint8_t aux[96];
            int la = 0;
            uint32_t mate_ref = 0xffffffffu, mate_pos = 0xffffffffu;
            int32_t tlen = 0;
            if (!g_aux) {
                put_dec((uint64_t)i);
                name[ln++] = 0;
            } else {
                // instrument : run : flowcell : lane : tile : x : y (what bcl2fastq writes; the pair shares it)
                const uint64_t g = mix(h ^ 0x5851f42d4c957f2dull);
                ln = 0;
                memcpy(name, "A00741:188:HGTMNDSX2:", 21); ln = 21;
                put_dec(1 + (g & 3)); name[ln++] = ':';
                put_dec(1101 + (g >> 2) % 1578); name[ln++] = ':';
                put_dec(1000 + (g >> 16) % 31000); name[ln++] = ':';
                put_dec(1000 + (g >> 36) % 36000);
                name[ln++] = 0;
                const int32_t frag = 300 + (int32_t)((g >> 52) % 200);
                mate_ref = (uint32_t)ctg;
                const bool fwd = (h & 1) != 0;                   // flag 99: forward, mate downstream
                int64_t mp = fwd ? pos + frag - RL : pos - (frag - RL);
                if (mp < 0) mp = 0;
                mate_pos = (uint32_t)mp;
                tlen = fwd ? frag : -frag;
                auto tagZ = [&](char a, char b, const char* v) { aux[la++] = (uint8_t)a; aux[la++] = (uint8_t)b; aux[la++] = 'Z'; while (*v) aux[la++] = (uint8_t)*v++; aux[la++] = 0; };
                auto tagC = [&](char a, char b, uint8_t v) { aux[la++] = (uint8_t)a; aux[la++] = (uint8_t)b; aux[la++] = 'C'; aux[la++] = v; };
                const uint32_t nm = (uint32_t)((g >> 8) % 8) < 5 ? 0u : (uint32_t)((g >> 11) % 4);     // most reads match
                char md[24];
                int lm = 0;
                auto md_dec = [&](uint32_t v) { char dg[8]; int nd = 0; do { dg[nd++] = (char)('0' + v % 10); v /= 10; } while (v); while (nd) md[lm++] = dg[--nd]; };
                const uint32_t mlen = (uint32_t)(nc == 2 ? (int)(cig[1] >> 4) : nc == 3 && (cig[1] & 15) == 1 ? RL - (int)(cig[1] >> 4) : RL);
                if (nm == 0) md_dec(mlen);
                else { const uint32_t a = 1 + (uint32_t)((g >> 20) % (mlen - 2)); md_dec(a); md[lm++] = "ACGT"[(g >> 30) & 3]; md_dec(mlen - a - 1); }
                md[lm] = 0;
                char mc[8] = "150M";
                tagC('N', 'M', (uint8_t)nm);
                tagZ('M', 'D', md);
                tagZ('M', 'C', mc);
                tagC('A', 'S', (uint8_t)(RL - 5 * nm));
                tagC('X', 'S', (uint8_t)((g >> 40) % 8 < 6 ? 0 : 19 + (g >> 44) % 100));
                tagZ('R', 'G', "rg1");
            }
            const uint32_t block = 32 + (uint32_t)ln + 4u * (uint32_t)nc + (RL + 1) / 2 + RL + (uint32_t)la;
            {
                const uint64_t off = (uint64_t)(w - w0);         // chunk local
                c->after_last = off + 4 + block;
                // windows are met in ascending order (positions ascend): only a window beyond the last one noted is new
                for (int64_t win = pos >> 14; win <= (pos + ref - 1) >> 14; ++win)
                    if (last_w == 0xffffffffu || (uint32_t)win > last_w) { c->lin.emplace_back((uint32_t)win, off); last_w = (uint32_t)win; }
            }
            p32(block);
            p32((uint32_t)ctg);                              // refID
            p32((uint32_t)pos);
            *w++ = (uint8_t)ln;
            *w++ = mapq;
            p16((uint16_t)reg2bin(pos, pos + ref));
            p16((uint16_t)nc);
            p16(flag);
            p32(RL);
            p32(mate_ref);                                   // next refID
            p32(mate_pos);                                   // next pos
            p32((uint32_t)tlen);
            memcpy(w, name, (size_t)ln); w += ln;
            for (int k = 0; k < nc; ++k) p32(cig[k]);
            uint64_t r = h;
            // two random bases per byte (A C G T = 1 2 4 8): byte k of a group of 16 is made of bits 4k .. 4k + 3 of r
            static const uint8_t two_bases[16] = {0x11, 0x21, 0x41, 0x81, 0x12, 0x22, 0x42, 0x82, 0x14, 0x24, 0x44, 0x84, 0x18, 0x28, 0x48, 0x88};
            for (int k = 0; k < (RL + 1) / 2; k += 16) {
                r = mix(r);
                const int m = std::min(16, (RL + 1) / 2 - k);
                for (int j = 0; j < m; ++j) w[j] = two_bases[(r >> (4 * j)) & 15];
                w += m;
            }
            for (int k = 0; k < RL; k += 8) {                // qualities: runs of 8, 37 with occasional dips
                r = mix(r);
                const uint64_t q8 = ((r & 7) == 0 ? (uint64_t)(2 + (r >> 8) % 35) : 37) * 0x0101010101010101ull;
                const int m = std::min(8, RL - k);
                memcpy(w, &q8, (size_t)m);
                w += m;
            }
            if (la) { memcpy(w, aux, (size_t)la); w += la; }
        }
        buf.resize((size_t)(w - w0));
    };

    std::vector<Chunk> chunks((size_t)threads);
    for (size_t ctg = 0; ctg < lens.size(); ++ctg) {
        const int64_t L = lens[ctg];
        const int64_t n = (int64_t)((double)L * cov / RL);
        std::vector<uint64_t>& lin = lins[ctg];
        n_total += n;
        const int64_t span = L - RL > 0 ? L - RL : 1;
        for (int64_t base = 0; base < n; base += CH * threads) {
            const int nch = (int)std::min<int64_t>(threads, (n - base + CH - 1) / CH);
            std::vector<std::thread> pool;
            for (int t = 0; t < nch; ++t)
                pool.emplace_back([&, t]() { gen_chunk(ctg, base + t * CH, std::min(n, base + (t + 1) * CH), n, span, &chunks[(size_t)t]); });
            for (auto& th : pool) th.join();
            pool.clear();
            std::vector<size_t> at((size_t)nch + 1, raw.size());
            for (int t = 0; t < nch; ++t) at[(size_t)t + 1] = at[(size_t)t] + chunks[(size_t)t].bytes.size();
            raw.grow_to(at[(size_t)nch]);
            for (int t = 0; t < nch; ++t)
                pool.emplace_back([&, t]() { memcpy(raw.data() + at[(size_t)t], chunks[(size_t)t].bytes.data(), chunks[(size_t)t].bytes.size()); });
            for (auto& th : pool) th.join();
            for (int t = 0; t < nch; ++t) {
                const Chunk& c = chunks[(size_t)t];
                if (c.bytes.empty()) continue;
                const uint64_t off0 = stream_off + at[(size_t)t];
                if (first_rec[ctg] == ~0ull) first_rec[ctg] = off0;
                after_last[ctg] = off0 + c.after_last;
                for (const auto& e : c.lin)
                    if ((size_t)e.first < lin.size() && lin[e.first] == ~0ull) lin[e.first] = off0 + e.second;
            }
            if (raw.size() >= (256u << 20)) flush(false);
        }
    }
    flush(true);
    static const uint8_t eof[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (pwrite(fd, eof, 28, (off_t)out_bytes) != 28) { perror("pwrite"); return 1; }
    close(fd);
    {
        std::vector<uint64_t> coff(csize.size() + 1, 0);
        for (size_t k = 0; k < csize.size(); ++k) coff[k + 1] = coff[k] + csize[k];
        auto voff = [&](uint64_t o) { return (coff[o / BLK] << 16) | (o % BLK); };
        std::vector<uint8_t> b;
        b.insert(b.end(), {'B', 'A', 'I', 1});
        put32(b, (uint32_t)lens.size());
        for (size_t ctg = 0; ctg < lens.size(); ++ctg) {
            const std::vector<uint64_t>& lin = lins[ctg];
            if (first_rec[ctg] != ~0ull) {
                put32(b, 1); put32(b, 0); put32(b, 1);              // one bin (0) with one chunk
                const uint64_t v0 = voff(first_rec[ctg]), v1 = voff(after_last[ctg]);
                for (int i = 0; i < 8; ++i) b.push_back((uint8_t)(v0 >> (8 * i)));
                for (int i = 0; i < 8; ++i) b.push_back((uint8_t)(v1 >> (8 * i)));
            } else put32(b, 0);
            size_t n_intv = lin.size();
            while (n_intv && lin[n_intv - 1] == ~0ull) --n_intv;
            put32(b, (uint32_t)n_intv);
            uint64_t last = 0;
            for (size_t w = 0; w < n_intv; ++w) {
                if (lin[w] != ~0ull) last = voff(lin[w]);
                for (int i = 0; i < 8; ++i) b.push_back((uint8_t)(last >> (8 * i)));
            }
        }
        FILE* fb = fopen((path + ".bai").c_str(), "wb");
        if (fb) { fwrite(b.data(), 1, b.size(), fb); fclose(fb); }
    }
    FILE* fai = fopen((path.substr(0, path.size() - 4) + ".fa.fai").c_str(), "w");
    if (fai) {
        for (size_t k = 0; k < lens.size(); ++k) fprintf(fai, "%s\t%lld\t6\t60\t61\n", names[k].c_str(), (long long)lens[k]);
        fclose(fai);
    }
    printf("{\"reads\": %lld, \"bam_bytes\": %llu, \"deflate\": \"%s level %d\", \"records\": \"%s\", \"inflated_bytes\": %llu}\n", (long long)n_total,
           (unsigned long long)(out_bytes + 28), g_ld.ok() ? "libdeflate" : "zlib", g_level,
           g_aux ? "Illumina-style names, mate fields, RG NM MD AS XS MC tags" : "short names, no tags",
           (unsigned long long)(stream_off + 0));
    return 0;
}
