// bam_reader.cpp -- see bam_reader.hpp.
#include "bam_reader.hpp"

#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cstring>
#include <thread>

namespace gdh {

namespace {

inline uint16_t rd16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }
inline uint32_t rd32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
inline uint64_t rd64(const uint8_t* p) { return (uint64_t)rd32(p) | ((uint64_t)rd32(p + 4) << 32); }

struct Member {
    size_t off;      // offset of the gzip member in raw
    size_t size;     // total member size (BSIZE+1)
    size_t xlen;
    size_t out_off;  // offset of its payload in the decoded buffer
    uint32_t isize;
};

// Parses one BGZF member header at raw[off..]; returns 1 ok, 0 need more bytes, -1 corrupt.
int parse_member(const std::vector<uint8_t>& raw, size_t off, Member* m)
{
    if (raw.size() - off < 18) return 0;
    const uint8_t* p = raw.data() + off;
    if (p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || !(p[3] & 4)) return -1;
    const size_t xlen = rd16(p + 10);
    if (raw.size() - off < 12 + xlen) return 0;
    size_t q = 12, bsize = 0;
    bool found = false;
    while (q + 4 <= 12 + xlen) {
        const uint8_t si1 = p[q], si2 = p[q + 1];
        const size_t slen = rd16(p + q + 2);
        if (si1 == 66 && si2 == 67 && slen == 2) { bsize = rd16(p + q + 4); found = true; }
        q += 4 + slen;
    }
    if (!found) return -1;
    if (raw.size() - off < bsize + 1) return 0;
    if (bsize + 1 < 12 + xlen + 8) return -1;
    m->off = off;
    m->size = bsize + 1;
    m->xlen = xlen;
    m->isize = rd32(p + bsize + 1 - 4);
    return 1;
}

bool inflate_member(const uint8_t* raw, const Member& m, uint8_t* out)
{
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (inflateInit2(&zs, -15) != Z_OK) return false;
    zs.next_in = const_cast<Bytef*>(raw + m.off + 12 + m.xlen);
    zs.avail_in = (uInt)(m.size - 12 - m.xlen - 8);
    zs.next_out = out + m.out_off;
    zs.avail_out = m.isize;
    const int rc = inflate(&zs, Z_FINISH);
    inflateEnd(&zs);
    if (rc != Z_STREAM_END || zs.avail_out != 0) return false;
    const uint32_t crc = rd32(raw + m.off + m.size - 8);
    return (uint32_t)crc32(crc32(0L, Z_NULL, 0), out + m.out_off, m.isize) == crc;
}

}  // namespace

BamReader::~BamReader()
{
    if (fp_) fclose(fp_);
}

bool BamReader::fill(std::string* err)
{
    // drop consumed bytes
    if (cur_ > 0) {
        buf_.erase(buf_.begin(), buf_.begin() + (ptrdiff_t)cur_);
        cur_ = 0;
    }
    if (eof_ && raw_.empty()) return false;
    const size_t kChunk = 16u << 20;
    if (!eof_) {
        const size_t old = raw_.size();
        raw_.resize(old + kChunk);
        const size_t got = fread(raw_.data() + old, 1, kChunk, fp_);
        raw_.resize(old + got);
        if (got < kChunk) eof_ = true;
    }
    std::vector<Member> ms;
    size_t off = 0, out = 0;
    for (;;) {
        Member m;
        const int rc = parse_member(raw_, off, &m);
        if (rc < 0) { if (err) *err = "corrupt BGZF member in " + path_; return false; }
        if (rc == 0) break;
        m.out_off = out;
        out += m.isize;
        off += m.size;
        ms.push_back(m);
    }
    if (ms.empty()) {
        if (eof_) {
            if (!raw_.empty()) { if (err) *err = "truncated BGZF file " + path_; raw_.clear(); }
            return false;
        }
        return true;   // need more compressed bytes; caller loops
    }
    const size_t base = buf_.size();
    buf_.resize(base + out);
    std::atomic<size_t> next{0};
    std::atomic<bool> bad{false};
    auto work = [&]() {
        for (;;) {
            const size_t i = next.fetch_add(1);
            if (i >= ms.size() || bad.load()) return;
            if (ms[i].isize == 0) continue;
            if (!inflate_member(raw_.data(), ms[i], buf_.data() + base)) bad.store(true);
        }
    };
    const int nt = (int)std::min<size_t>((size_t)threads_, ms.size());
    if (nt <= 1) {
        work();
    } else {
        std::vector<std::thread> th;
        for (int t = 0; t < nt; ++t) th.emplace_back(work);
        for (auto& t : th) t.join();
    }
    if (bad.load()) { if (err) *err = "BGZF inflate/CRC failure in " + path_; return false; }
    raw_.erase(raw_.begin(), raw_.begin() + (ptrdiff_t)off);
    return true;
}

bool BamReader::need(size_t n, std::string* err)
{
    while (buf_.size() - cur_ < n) {
        const size_t before = buf_.size() - cur_;
        std::string e;
        if (!fill(&e)) {
            if (!e.empty() && err) *err = e;
            return false;
        }
        if (buf_.size() - cur_ == before && eof_ && raw_.empty()) return false;
    }
    return true;
}

bool BamReader::open(const std::string& path, int threads, std::string* err)
{
    path_ = path;
    threads_ = threads > 0 ? threads : (int)std::max(1u, std::thread::hardware_concurrency());
    fp_ = fopen(path.c_str(), "rb");
    if (!fp_) { if (err) *err = "cannot open " + path; return false; }
    if (!need(12, err) || memcmp(buf_.data() + cur_, "BAM\1", 4) != 0) {
        if (err && err->empty()) *err = path + " is not a BAM file";
        return false;
    }
    const uint32_t l_text = rd32(buf_.data() + cur_ + 4);
    if (!need(12 + (size_t)l_text, err)) { if (err && err->empty()) *err = "truncated BAM header"; return false; }
    text_.assign(reinterpret_cast<const char*>(buf_.data() + cur_ + 8), l_text);
    const uint32_t n_ref = rd32(buf_.data() + cur_ + 8 + l_text);
    cur_ += 12 + l_text;
    contigs_.clear();
    for (uint32_t i = 0; i < n_ref; ++i) {
        if (!need(4, err)) { if (err && err->empty()) *err = "truncated BAM reference table"; return false; }
        const uint32_t l_name = rd32(buf_.data() + cur_);
        if (!need(8 + (size_t)l_name, err)) { if (err && err->empty()) *err = "truncated BAM reference table"; return false; }
        BamContig c;
        c.name.assign(reinterpret_cast<const char*>(buf_.data() + cur_ + 4), l_name ? l_name - 1 : 0);
        c.length = (int32_t)rd32(buf_.data() + cur_ + 4 + l_name);
        contigs_.push_back(c);
        cur_ += 8 + l_name;
    }
    return true;
}

bool BamReader::seek_contig(int32_t tid, std::string* err)
{
    FILE* fi = fopen((path_ + ".bai").c_str(), "rb");
    if (!fi) {
        std::string alt = path_;
        if (alt.size() > 4 && alt.substr(alt.size() - 4) == ".bam") alt = alt.substr(0, alt.size() - 4) + ".bai";
        fi = fopen(alt.c_str(), "rb");
        if (!fi) return false;
    }
    std::vector<uint8_t> d;
    uint8_t tmp[65536];
    size_t g;
    while ((g = fread(tmp, 1, sizeof tmp, fi)) > 0) d.insert(d.end(), tmp, tmp + g);
    fclose(fi);
    if (d.size() < 8 || memcmp(d.data(), "BAI\1", 4) != 0) { if (err) *err = "bad BAI magic"; return false; }
    const int32_t n_ref = (int32_t)rd32(d.data() + 4);
    size_t p = 8;
    uint64_t best = ~0ull;
    for (int32_t r = 0; r < n_ref; ++r) {
        if (p + 4 > d.size()) return false;
        const int32_t n_bin = (int32_t)rd32(d.data() + p);
        p += 4;
        for (int32_t b = 0; b < n_bin; ++b) {
            if (p + 8 > d.size()) return false;
            const uint32_t bin = rd32(d.data() + p);
            const int32_t n_chunk = (int32_t)rd32(d.data() + p + 4);
            p += 8;
            if (p + 16ull * (size_t)n_chunk > d.size()) return false;
            if (r == tid && bin != 37450)
                for (int32_t k = 0; k < n_chunk; ++k) best = std::min(best, rd64(d.data() + p + 16 * (size_t)k));
            p += 16 * (size_t)n_chunk;
        }
        if (p + 4 > d.size()) return false;
        const int32_t n_intv = (int32_t)rd32(d.data() + p);
        p += 4 + 8 * (size_t)n_intv;
        if (r == tid) break;
    }
    if (best == ~0ull) return false;
    const uint64_t coff = best >> 16, uoff = best & 0xffff;
    if (fseeko(fp_, (off_t)coff, SEEK_SET) != 0) { if (err) *err = "seek failed"; return false; }
    raw_.clear();
    buf_.clear();
    cur_ = 0;
    eof_ = false;
    if (!need(uoff + 1, err)) return false;
    cur_ = uoff;
    return true;
}

int BamReader::next_block(RecordBlock& out, size_t max_reads, std::string* err)
{
    out.clear();
    for (;;) {
        std::string e;
        if (!need(4, &e)) {
            if (!e.empty()) { if (err) *err = e; return -1; }
            if (buf_.size() - cur_ != 0) { if (err) *err = "truncated BAM record"; return -1; }
            return out.size() ? 1 : 0;
        }
        const uint32_t block_size = rd32(buf_.data() + cur_);
        if (block_size < 32) { if (err) *err = "corrupt BAM record"; return -1; }
        if (!need(4 + (size_t)block_size, &e)) { if (err) *err = e.empty() ? "truncated BAM record" : e; return -1; }
        const uint8_t* r = buf_.data() + cur_ + 4;
        const int32_t ref_id = (int32_t)rd32(r);
        if (out.size() && (ref_id != out.tid || out.size() >= max_reads)) return 1;   // leave it for the next call
        const int32_t pos = (int32_t)rd32(r + 4);
        const uint32_t l_read_name = r[8];
        const uint8_t mq = r[9];
        uint32_t n_cigar = rd16(r + 12);
        const uint16_t flag = rd16(r + 14);
        const uint32_t l_seq = rd32(r + 16);
        cur_ += 4 + block_size;
        ++n_records_;
        if (ref_id < 0) { ++n_unplaced_; continue; }
        if (32 + l_read_name + 4ull * n_cigar > block_size) { if (err) *err = "corrupt BAM record"; return -1; }
        const uint8_t* cg = r + 32 + l_read_name;
        const uint8_t* end = r + block_size;
        // long-CIGAR convention: <l_seq>S<ref_len>N placeholder, real CIGAR in CG:B,I
        const uint8_t* real = nullptr;
        uint32_t n_real = 0;
        if (n_cigar == 2 && (rd32(cg) & 0xf) == 4 && (rd32(cg) >> 4) == l_seq && (rd32(cg + 4) & 0xf) == 3) {
            const uint8_t* t = cg + 8 + (l_seq + 1) / 2 + l_seq;
            while (t + 3 <= end) {
                const uint8_t t0 = t[0], t1 = t[1], ty = t[2];
                t += 3;
                size_t sz = 0;
                if (ty == 'A' || ty == 'c' || ty == 'C') sz = 1;
                else if (ty == 's' || ty == 'S') sz = 2;
                else if (ty == 'i' || ty == 'I' || ty == 'f') sz = 4;
                else if (ty == 'Z' || ty == 'H') { while (t < end && *t) ++t; ++t; continue; }
                else if (ty == 'B') {
                    if (t + 5 > end) break;
                    const uint8_t sub = t[0];
                    const uint32_t cnt = rd32(t + 1);
                    t += 5;
                    const size_t es = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : 4;
                    if (t0 == 'C' && t1 == 'G' && sub == 'I' && t + 4ull * cnt <= end) { real = t; n_real = cnt; break; }
                    t += es * (size_t)cnt;
                    continue;
                } else break;
                t += sz;
            }
        }
        if (real) { cg = real; n_cigar = n_real; }
        if (out.size() == 0) out.tid = ref_id;
        out.pos.push_back(pos);
        out.flag.push_back(flag);
        out.mapq.push_back(mq);
        const size_t c0 = out.cigar.size();
        out.cigar.resize(c0 + n_cigar);
        for (uint32_t k = 0; k < n_cigar; ++k) out.cigar[c0 + k] = rd32(cg + 4 * (size_t)k);
        out.cigar_off.push_back((uint32_t)out.cigar.size());
    }
}

}  // namespace gdh
