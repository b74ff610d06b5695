// bam_reader.cpp -- see bam_reader.hpp.
#include "bam_reader.hpp"

#include <dlfcn.h>
#include <fcntl.h>
#include <unistd.h>
#include <zlib.h>

#include <cerrno>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>

#include <chrono>
namespace gdh {

// Worker threads that live as long as the reader: a batch used to start (and join) one thread per core for the
// inflate and two more sets for the record passes -- on a 256-core host more time than the work itself.
struct Workers {
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv, idle_cv;
    std::function<void(size_t)> fn;
    size_t n_tasks = 0;
    std::atomic<size_t> next{0};
    uint64_t gen = 0;
    int active = 0;
    bool quit = false;
    explicit Workers(int n)
    {
        for (int k = 0; k < n; ++k)
            th.emplace_back([this] {
                uint64_t seen = 0;
                std::unique_lock<std::mutex> lk(mu);
                for (;;) {
                    cv.wait(lk, [&] { return quit || gen != seen; });
                    if (quit) return;
                    seen = gen;
                    ++active;
                    lk.unlock();
                    drain();
                    lk.lock();
                    if (--active == 0) idle_cv.notify_all();
                }
            });
    }
    ~Workers()
    {
        { std::lock_guard<std::mutex> lk(mu); quit = true; }
        cv.notify_all();
        for (auto& t : th) t.join();
    }
    void drain()
    {
        for (;;) {
            const size_t i = next.fetch_add(1);
            if (i >= n_tasks) return;
            fn(i);
        }
    }
    // f(0) .. f(n - 1) on the workers and the calling thread; returns when all have returned
    void run(size_t n, std::function<void(size_t)> f)
    {
        {
            std::unique_lock<std::mutex> lk(mu);
            idle_cv.wait(lk, [&] { return active == 0; });     // (a worker that woke late for the previous call is through)
            fn = std::move(f);
            n_tasks = n;
            next.store(0);
            ++gen;
        }
        cv.notify_all();
        drain();
        std::unique_lock<std::mutex> lk(mu);
        idle_cv.wait(lk, [&] { return active == 0; });
        n_tasks = 0;                                         // a worker that wakes from now on finds nothing to do
    }
};

namespace {
struct Tm {                                   // GOLEFT_BAM_TIMING=1: phase wall times on stderr at close
    double read = 0, inflate = 0, wait = 0, append = 0, hop = 0, pass1 = 0, pass2 = 0;
    static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
};
Tm g_tm;

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
int parse_member(const Bytes& raw, size_t off, Member* m)
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
        if (si1 == 66 && si2 == 67 && slen == 2 && q + 6 <= 12 + xlen) { bsize = rd16(p + q + 4); found = true; }
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

// libdeflate when the system has its shared library (htslib's own choice when built against it; 2-3x zlib's inflate
// and a carry-less-multiply CRC32 -- the inflate is what this reader's time is made of), zlib otherwise or when
// GOLEFT_HOST_ZLIB is set.  Same contract either way: exactly `isize` bytes and the trailer's CRC, or false.
struct LibDeflate {
    void* (*alloc)() = nullptr;
    void (*release)(void*) = nullptr;
    int (*decompress)(void*, const void*, size_t, void*, size_t, size_t*) = nullptr;
    uint32_t (*crc)(uint32_t, const void*, size_t) = nullptr;
    LibDeflate()
    {
        if (getenv("GOLEFT_HOST_ZLIB")) return;
        void* h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
        if (!h) return;
        alloc = reinterpret_cast<void* (*)()>(dlsym(h, "libdeflate_alloc_decompressor"));
        release = reinterpret_cast<void (*)(void*)>(dlsym(h, "libdeflate_free_decompressor"));
        decompress = reinterpret_cast<int (*)(void*, const void*, size_t, void*, size_t, size_t*)>(dlsym(h, "libdeflate_deflate_decompress"));
        crc = reinterpret_cast<uint32_t (*)(uint32_t, const void*, size_t)>(dlsym(h, "libdeflate_crc32"));
        if (!alloc || !release || !decompress || !crc) alloc = nullptr;
    }
};
const LibDeflate& libdeflate()
{
    static const LibDeflate ld;
    return ld;
}
struct Decompressor {                                    // one per thread, freed with it
    void* d = nullptr;
    ~Decompressor() { if (d) libdeflate().release(d); }
};

bool inflate_member(const uint8_t* raw, const Member& m, uint8_t* out)
{
    const LibDeflate& ld = libdeflate();
    if (ld.alloc) {
        static thread_local Decompressor td;
        if (!td.d) td.d = ld.alloc();
        if (td.d) {
            // (no actual-size pointer: anything but exactly isize bytes is an error)
            if (ld.decompress(td.d, raw + m.off + 12 + m.xlen, m.size - 12 - m.xlen - 8, out + m.out_off, m.isize, nullptr) != 0) return false;
            return ld.crc(0, out + m.out_off, m.isize) == rd32(raw + m.off + m.size - 8);
        }
    }
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

BamReader::BamReader() = default;

BamReader::~BamReader()
{
    drop_prefetch();
    if (getenv("GOLEFT_BAM_TIMING"))
        fprintf(stderr, "bam_reader: read %.3f inflate %.3f (producer thread) | wait %.3f append %.3f hop %.3f pass1 %.3f pass2 %.3f s\n",
                g_tm.read, g_tm.inflate, g_tm.wait, g_tm.append, g_tm.hop, g_tm.pass1, g_tm.pass2);
    if (fd_ >= 0) close(fd_);
}

// Producer side: reads the next kChunk compressed bytes, inflates every complete BGZF member
// in it (in parallel) and returns the decoded bytes.  Runs on a background thread one chunk
// ahead of the record decoder (prefetch_), so file I/O + inflate overlap the decode.
// Every batch of inflated bytes starts kHead bytes into its buffer: the consumer copies the record that straddles two
// batches (hundreds of bytes; megabytes for an ultra-long read) in FRONT of the new batch and goes on in that buffer,
// instead of appending the new batch (a quarter of a gigabyte) to the old one.
static size_t kHead = 8u << 20;                        // (GOLEFT_BAM_HEAD_KB, read when a file is opened: tests shrink it)
static size_t kChunkBytes = 64u << 20;                 // compressed bytes per batch (GOLEFT_BAM_CHUNK_KB)

BamReader::Chunk BamReader::produce(Bytes spare)
{
    Chunk c;
    c.data = std::move(spare);                           // recycled pages: no fresh page faults per batch
    c.data.clear();
    if (eof_ && raw_.empty()) { c.end = true; return c; }
    // the first batch after open / seek is small: whoever only wants the header (goleft-depth with the device decoder)
    // does not pay for 64 MB of BGZF, and the first records arrive sooner
    const size_t kChunk = n_batches_++ == 0 ? std::min<size_t>(kChunkBytes, 1u << 20) : kChunkBytes;
    for (;;) {
        double t0 = Tm::now();
        if (!eof_) {
            // every worker reads its slice of the batch at its own offset (one thread's read of 64 MB from the page cache
            // took as long as all threads' inflate of it); a pipe is read in order by this thread
            const size_t old = raw_.size();
            raw_.resize(old + kChunk);
            uint8_t* const dst = raw_.data() + old;
            // A read ERROR is not the end of the file: taken for one, an error that lands on a member boundary would
            // make a truncated depth look complete.
            std::atomic<int> read_errno{0};
            auto read_at = [&](size_t b, size_t e) -> size_t {           // bytes read of [b, e): short only at the end of the file
                size_t done = b;
                while (done < e) {
                    const ssize_t r = seekable_ ? pread(fd_, dst + done, e - done, (off_t)(file_off_ + done)) : read(fd_, dst + done, e - done);
                    if (r < 0 && errno == EINTR) continue;
                    if (r < 0) { read_errno.store(errno ? errno : EIO); break; }
                    if (r == 0) break;
                    done += (size_t)r;
                }
                return done - b;
            };
            size_t got = 0;
            const size_t slice = 4u << 20;
            const size_t ns = (kChunk + slice - 1) / slice;
            if (seekable_ && threads_ > 1 && ns > 1) {
                std::vector<size_t> part(ns, 0);
                std::atomic<size_t> next{0};
                if (!inflate_workers_) inflate_workers_.reset(new Workers(threads_ - 1));
                inflate_workers_->run(std::min<size_t>((size_t)threads_, ns), [&](size_t) {
                    for (;;) {
                        const size_t i = next.fetch_add(1);
                        if (i >= ns) return;
                        part[i] = read_at(i * slice, std::min(kChunk, (i + 1) * slice));
                    }
                });
                // the bytes in front of the first short slice are the file's; nothing follows a short read
                for (size_t i = 0; i < ns; ++i) {
                    got += part[i];
                    if (part[i] < std::min(kChunk, (i + 1) * slice) - i * slice) break;
                }
            } else {
                got = read_at(0, kChunk);
            }
            if (read_errno.load() != 0) {
                c.err = "read error in " + path_ + ": " + strerror(read_errno.load());
                raw_.clear();
                c.data.clear();
                c.end = true;
                eof_ = true;
                return c;
            }
            file_off_ += got;
            raw_.resize(old + got);
            if (got < kChunk) eof_ = true;
        }
        g_tm.read += Tm::now() - t0;
        t0 = Tm::now();
        std::vector<Member> ms;
        size_t off = 0, out = 0;
        for (;;) {
            Member m;
            const int rc = parse_member(raw_, off, &m);
            if (rc < 0) { c.err = "corrupt BGZF member in " + path_; c.end = true; return c; }
            if (rc == 0) break;
            m.out_off = out;
            out += m.isize;
            off += m.size;
            ms.push_back(m);
        }
        if (ms.empty()) {
            if (eof_) {
                if (!raw_.empty()) { c.err = "truncated BGZF file " + path_; raw_.clear(); }
                c.end = true;
                return c;
            }
            continue;                                    // a member larger than what is buffered: read on
        }
        c.data.resize(kHead + out);
        std::atomic<size_t> next{0};
        std::atomic<bool> bad{false};
        auto work = [&]() {
            for (;;) {
                const size_t i = next.fetch_add(8);      // eight members (~0.5 MB) per grab
                if (i >= ms.size() || bad.load()) return;
                for (size_t k = i; k < std::min(i + 8, ms.size()); ++k) {
                    if (ms[k].isize == 0) continue;
                    if (!inflate_member(raw_.data(), ms[k], c.data.data() + kHead)) bad.store(true);
                }
            }
        };
        const int nt = (int)std::min<size_t>((size_t)threads_, (ms.size() + 15) / 16);
        if (nt <= 1) {
            work();
        } else {
            if (!inflate_workers_) inflate_workers_.reset(new Workers(threads_ - 1));
            inflate_workers_->run((size_t)nt, [&](size_t) { work(); });
        }
        if (bad.load()) { c.err = "BGZF inflate/CRC failure in " + path_; c.data.clear(); c.end = true; return c; }
        raw_.erase_front(off);
        g_tm.inflate += Tm::now() - t0;
        return c;
    }
}

void BamReader::drop_prefetch()
{
    if (prefetch_.valid()) (void)prefetch_.get();
}

bool BamReader::fill(std::string* err)
{
    if (done_) return false;
    double t0 = Tm::now();
    Chunk c = prefetch_.valid() ? prefetch_.get() : produce(std::move(spare_));
    g_tm.wait += Tm::now() - t0;
    t0 = Tm::now();
    if (!c.err.empty()) { if (err) *err = c.err; done_ = true; return false; }
    if (c.end) { done_ = true; return false; }
    const size_t rest = buf_.size() - cur_;              // decoded bytes not consumed yet: normally one partial record
    Bytes spare;
    if (rest <= kHead) {
        if (rest) memcpy(c.data.data() + kHead - rest, buf_.data() + cur_, rest);
        spare = std::move(buf_);
        buf_ = std::move(c.data);
        cur_ = kHead - rest;
    } else {                                             // (a record larger than the headroom, or a caller that holds on to a block's bytes)
        if (cur_ > 0) {
            buf_.erase_front(cur_);
            cur_ = 0;
        }
        if (buf_.capacity() < buf_.size() + c.data.size()) buf_.reserve(3 * c.data.size() + buf_.size());
        buf_.append(c.data.data() + kHead, c.data.size() - kHead);
        spare = std::move(c.data);
    }
    g_tm.append += Tm::now() - t0;
    // one batch ahead, into the buffer just emptied -- from the second batch on: nothing is read ahead of the first
    // (small) one, so that opening a file for its header leaves the cores alone
    if (n_fills_++ == 0) { spare_ = std::move(spare); return true; }
    prefetch_ = std::async(std::launch::async,
                           [this](Bytes sp) { return produce(std::move(sp)); }, std::move(spare));
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
        if (buf_.size() - cur_ == before && done_) return false;
    }
    return true;
}

// The CPUs this process may really use: the hardware's, or fewer when a container's CPU quota says so (cgroup v2
// cpu.max, v1 cfs_quota_us / cfs_period_us).  256 threads under a 16-CPU quota spend each 100 ms period's budget in its
// first 6 ms and stand still for the rest -- every thread of the process with them, the ones feeding the device included.
int usable_cpus()
{
    const int hw = (int)std::max(1u, std::thread::hardware_concurrency());
    long long quota = -1, period = 100000;
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[32] = {0};
        if (fscanf(f, "%31s %lld", q, &period) >= 1 && strcmp(q, "max") != 0) quota = atoll(q);
        fclose(f);
    } else if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
        if (fscanf(g, "%lld", &quota) != 1) quota = -1;
        fclose(g);
        if (FILE* h = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
            if (fscanf(h, "%lld", &period) != 1) period = 100000;
            fclose(h);
        }
    }
    if (quota <= 0 || period <= 0) return hw;
    return (int)std::max<long long>(1, std::min<long long>(hw, (quota + period - 1) / period));
}

bool BamReader::open(const std::string& path, int threads, std::string* err)
{
    path_ = path;
    threads_ = threads > 0 ? threads : usable_cpus();
    inflate_workers_.reset();                            // (started by the first batch that has work for them)
    parse_workers_.reset();
    if (const char* e = getenv("GOLEFT_BAM_CHUNK_KB")) kChunkBytes = (size_t)std::max(64, atoi(e)) << 10;
    if (const char* e = getenv("GOLEFT_BAM_HEAD_KB")) kHead = (size_t)std::max(0, atoi(e)) << 10;
    drop_prefetch();                                     // (a reader that is opened again: nobody still reads the old file)
    if (fd_ >= 0) close(fd_);
    fd_ = ::open(path.c_str(), O_RDONLY | O_CLOEXEC);
    if (fd_ < 0) { if (err) *err = "cannot open " + path; return false; }
    file_off_ = 0;
    seekable_ = lseek(fd_, 0, SEEK_CUR) != (off_t)-1;
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
    left_.assign(contigs_.size(), false);
    last_ref_ = -2;
    return true;
}

namespace {
bool load_bai(const std::string& bam_path, std::vector<uint8_t>* d)
{
    FILE* fi = fopen((bam_path + ".bai").c_str(), "rb");
    if (!fi) {
        std::string alt = bam_path;
        if (alt.size() > 4 && alt.substr(alt.size() - 4) == ".bam") alt = alt.substr(0, alt.size() - 4) + ".bai";
        fi = fopen(alt.c_str(), "rb");
        if (!fi) return false;
    }
    uint8_t tmp[65536];
    size_t g;
    while ((g = fread(tmp, 1, sizeof tmp, fi)) > 0) d->insert(d->end(), tmp, tmp + g);
    fclose(fi);
    return d->size() >= 8 && memcmp(d->data(), "BAI\1", 4) == 0;
}
}  // namespace

bool BamReader::linear_index(const std::string& bam_path, std::vector<std::vector<uint64_t>>* per_ref, std::string* err,
                             std::vector<char>* has_chunks, std::vector<uint64_t>* chunk_end)
{
    std::vector<uint8_t> d;
    if (!load_bai(bam_path, &d)) return false;
    const int32_t n_ref = (int32_t)rd32(d.data() + 4);
    if (n_ref < 0 || (size_t)n_ref > (d.size() - 8) / 8) {     // every reference takes at least n_bin + n_intv
        if (err) *err = "corrupt BAI";
        return false;
    }
    per_ref->assign((size_t)n_ref, std::vector<uint64_t>());
    if (has_chunks) has_chunks->assign((size_t)n_ref, 0);
    if (chunk_end) chunk_end->assign((size_t)n_ref, 0);
    size_t p = 8;
    for (int32_t r = 0; r < n_ref; ++r) {
        if (p + 4 > d.size()) { if (err) *err = "truncated BAI"; return false; }
        const int32_t n_bin = (int32_t)rd32(d.data() + p);
        p += 4;
        for (int32_t b = 0; b < n_bin; ++b) {
            if (p + 8 > d.size()) { if (err) *err = "truncated BAI"; return false; }
            const uint32_t bin = rd32(d.data() + p);
            const int32_t n_chunk = (int32_t)rd32(d.data() + p + 4);
            if (n_chunk < 0 || (d.size() - p - 8) / 16 < (size_t)n_chunk) { if (err) *err = "corrupt BAI"; return false; }
            if (bin != 37450 && n_chunk > 0) {                  // 37450: the metadata pseudo-bin (SAMv1 5.2)
                if (has_chunks) (*has_chunks)[(size_t)r] = 1;
                if (chunk_end)
                    for (int32_t k = 0; k < n_chunk; ++k) {
                        const uint64_t e = rd64(d.data() + p + 8 + 16 * (size_t)k + 8);
                        if (e > (*chunk_end)[(size_t)r]) (*chunk_end)[(size_t)r] = e;
                    }
            }
            p += 8 + 16 * (size_t)n_chunk;
        }
        if (p + 4 > d.size()) { if (err) *err = "truncated BAI"; return false; }
        const int32_t n_intv = (int32_t)rd32(d.data() + p);
        p += 4;
        if (n_intv < 0 || (d.size() - p) / 8 < (size_t)n_intv) { if (err) *err = "truncated BAI"; return false; }
        std::vector<uint64_t>& v = (*per_ref)[(size_t)r];
        for (int32_t i = 0; i < n_intv; ++i) {
            const uint64_t x = rd64(d.data() + p + 8 * (size_t)i);
            if (x != 0 && (v.empty() || x > v.back())) v.push_back(x);   // the index is non-decreasing
        }
        p += 8 * (size_t)n_intv;
    }
    return true;
}

bool BamReader::seek_contig(int32_t tid, std::string* err) { return seek_contig_ex(tid, err) == 1; }

// 1: positioned at the reference's first record; 0: the index says the reference holds no records (nothing was touched);
// -1: no usable index (nothing was touched: the caller may scan from where it is); -2: the index pointed somewhere the
// file could not be read -- the reader's position is gone, the caller must give up.
int BamReader::seek_contig_ex(int32_t tid, std::string* err)
{
    std::vector<uint8_t> d;
    if (!load_bai(path_, &d)) return -1;
    const int32_t n_ref = (int32_t)rd32(d.data() + 4);
    if (tid < 0 || tid >= n_ref) return -1;
    size_t p = 8;
    uint64_t best = ~0ull;
    for (int32_t r = 0; r < n_ref; ++r) {
        if (p + 4 > d.size()) return -1;
        const int32_t n_bin = (int32_t)rd32(d.data() + p);
        p += 4;
        for (int32_t b = 0; b < n_bin; ++b) {
            if (p + 8 > d.size()) return -1;
            const uint32_t bin = rd32(d.data() + p);
            const int32_t n_chunk = (int32_t)rd32(d.data() + p + 4);
            p += 8;
            if (n_chunk < 0 || (d.size() - p) / 16 < (size_t)n_chunk) return -1;
            if (r == tid && bin != 37450)
                for (int32_t k = 0; k < n_chunk; ++k) best = std::min(best, rd64(d.data() + p + 16 * (size_t)k));
            p += 16 * (size_t)n_chunk;
        }
        if (p + 4 > d.size()) return -1;
        const int32_t n_intv = (int32_t)rd32(d.data() + p);
        if (n_intv < 0 || (d.size() - p - 4) / 8 < (size_t)n_intv) return -1;
        p += 4 + 8 * (size_t)n_intv;
        if (r == tid) break;
    }
    if (best == ~0ull) return 0;                    // a well-formed index without a chunk for this reference: no records
    const uint64_t coff = best >> 16, uoff = best & 0xffff;
    drop_prefetch();                              // the producer owns the file offset and raw_ while a chunk is in flight
    if (!seekable_) { if (err) *err = "seek failed"; return -2; }
    file_off_ = coff;
    raw_.clear();
    buf_.clear();
    cur_ = 0;
    eof_ = false;
    done_ = false;
    n_batches_ = 0;
    n_fills_ = 0;
    if (!need(uoff + 1, err)) { if (err && err->empty()) *err = "the index points past the end of the file"; return -2; }
    cur_ += uoff;
    left_.assign(contigs_.size(), false);         // a deliberate jump: the run rule starts over
    last_ref_ = -2;
    return 1;
}

namespace {

// The CIGAR of the record at r (block_size bytes): the stored one, or the CG:B,I tag's when the
// stored one is the <l_seq>S<ref_len>N placeholder of the long-CIGAR convention (SAMv1 4.2.2).
// Returns false on a corrupt record.
bool record_cigar(const uint8_t* r, uint32_t block_size, const uint8_t** cg_out, uint32_t* n_out)
{
    const uint32_t l_read_name = r[8];
    uint32_t n_cigar = rd16(r + 12);
    const uint32_t l_seq = rd32(r + 16);
    if (32 + l_read_name + 4ull * n_cigar > block_size) return false;
    const uint8_t* cg = r + 32 + l_read_name;
    const uint8_t* end = r + block_size;
    if (n_cigar == 2 && (rd32(cg) & 0xf) == 4 && (rd32(cg) >> 4) == l_seq && (rd32(cg + 4) & 0xf) == 3 &&
        (uint64_t)(end - (cg + 8)) >= ((uint64_t)l_seq + 1) / 2 + (uint64_t)l_seq) {
        const uint8_t* t = cg + 8 + (l_seq + 1) / 2 + l_seq;
        while (end - t >= 3) {
            const uint8_t t0 = t[0], t1 = t[1], ty = t[2];
            t += 3;
            size_t sz = 0;
            if (ty == 'A' || ty == 'c' || ty == 'C') sz = 1;
            else if (ty == 's' || ty == 'S') sz = 2;
            else if (ty == 'i' || ty == 'I' || ty == 'f') sz = 4;
            else if (ty == 'Z' || ty == 'H') { while (t < end && *t) ++t; ++t; continue; }
            else if (ty == 'B') {
                if (end - t < 5) break;
                const uint8_t sub = t[0];
                const uint32_t cnt = rd32(t + 1);
                t += 5;
                const size_t es = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : 4;
                if ((uint64_t)(end - t) < (uint64_t)es * cnt) break;          // the array runs past the record
                if (t0 == 'C' && t1 == 'G' && sub == 'I') { cg = t; n_cigar = cnt; break; }
                t += es * (size_t)cnt;
                continue;
            } else break;
            if ((size_t)(end - t) < sz) break;
            t += sz;
        }
    }
    *cg_out = cg;
    *n_out = n_cigar;
    return true;
}

template <typename F>
void parallel_for(size_t n, int threads, Workers* pool, F f)   // f(begin, end) over [0, n)
{
    const size_t grain = 1u << 15;
    const size_t nt = std::min<size_t>((size_t)std::max(1, threads), (n + grain - 1) / grain);
    if (nt <= 1 || !pool) { f((size_t)0, n); return; }
    pool->run(nt, [&](size_t t) { f(n * t / nt, n * (t + 1) / nt); });
}

}  // namespace

// Record boundaries are found with a sequential hop over the block_size fields (cheap); the
// fields and CIGARs of the records found are then extracted by all threads in two passes
// (sizes -> prefix sum -> contents), so decode no longer runs at one core's speed.
int BamReader::next_block(RecordBlock& out, size_t max_reads, std::string* err)
{
    out.clear();
    const double t_hop0 = Tm::now();
    const double wait0 = g_tm.wait + g_tm.append;
    std::vector<size_t>& at = at_;                        // offset of every record's body, relative to cur_ (kept: no growth after the first block)
    at.clear();
    size_t p = 0;                                         // bytes hopped over, relative to cur_
    for (;;) {
        if (!at.empty()) {
            // the common case, without the calls below: a placed record of the block's reference that lies wholly in the
            // decoded bytes.  The hop is a chain of dependent loads from lines other cores have just written; the line a
            // few records ahead is asked for early, guessed from this record's size (records of one file are of a size).
            const uint8_t* const b = buf_.data() + cur_;
            const size_t have = buf_.size() - cur_;
            while (at.size() < max_reads && p + 12 <= have) {
                const uint32_t bs = rd32(b + p);
                if (bs < 32 || p + 4 + (size_t)bs > have) break;
                if ((int32_t)rd32(b + p + 4) != out.tid || (int32_t)rd32(b + p + 8) < 0) break;
                __builtin_prefetch(b + p + 8 * (size_t)(4 + bs));
                __builtin_prefetch(b + p + 16 * (size_t)(4 + bs));
                ++n_records_;
                at.push_back(p + 4);
                p += 4 + (size_t)bs;
            }
        }
        const size_t have = buf_.size() - cur_;
        if (at.empty() && p) { cur_ += p; p = 0; continue; }           // records skipped so far are consumed
        // a block ends where the decoded bytes end: the record that straddles two batches opens the next block
        if (!at.empty() && (have < p + 4 || have < p + 4 + (size_t)rd32(buf_.data() + cur_ + p))) break;
        std::string e;
        if (!need(p + 4, &e)) {
            if (!e.empty()) { if (err) *err = e; return -1; }
            if (buf_.size() - cur_ != p) { if (err) *err = "truncated BAM record"; return -1; }
            break;
        }
        const uint32_t block_size = rd32(buf_.data() + cur_ + p);
        if (block_size < 32) { if (err) *err = "corrupt BAM record"; return -1; }
        if (!need(p + 4 + (size_t)block_size, &e)) { if (err) *err = e.empty() ? "truncated BAM record" : e; return -1; }
        const int32_t ref_id = (int32_t)rd32(buf_.data() + cur_ + p + 4);
        if (!at.empty() && (ref_id != out.tid || at.size() >= max_reads)) break;   // leave it for the next call
        // a coordinate-sorted BAM holds a reference's records in ONE run (`samtools depth -r` stops at the first
        // record of another reference): records of a reference that was left already are refused, not appended
        if (ref_id != last_ref_) {
            if (last_ref_ >= 0 && (size_t)last_ref_ < left_.size()) left_[(size_t)last_ref_] = true;
            const int32_t before = last_ref_;
            last_ref_ = ref_id;
            if (ref_id < -1 || (ref_id >= 0 && (size_t)ref_id >= left_.size())) {
                if (err) *err = "corrupt BAM record (reference id out of range)";
                return -1;
            }
            // sorted by (reference, position): references ascend, the unplaced records (-1) come last
            if (ref_id >= 0 && (left_[(size_t)ref_id] || before == -1 || (before >= 0 && ref_id < before))) {
                if (err) *err = "BAM not sorted: records of a reference after a later reference's (or after the unplaced ones)";
                return -1;
            }
        }
        ++n_records_;
        // unplaced records: no reference, or filed under one with POS -1 ("no position", dropped by samtools depth
        // through the 0x4 flag they carry)
        if (ref_id < 0 || (int32_t)rd32(buf_.data() + cur_ + p + 8) < 0) { ++n_unplaced_; p += 4 + block_size; continue; }
        if (at.empty()) out.tid = ref_id;
        at.push_back(p + 4);
        p += 4 + block_size;
    }
    const size_t n = at.size();
    g_tm.hop += Tm::now() - t_hop0 - (g_tm.wait + g_tm.append - wait0);
    if (n) {
        double t1 = Tm::now();
        const uint8_t* base = buf_.data() + cur_;
        out.pos.resize(n); out.flag.resize(n); out.mapq.resize(n);
        out.cigar_off.assign(n + 1, 0);
        std::atomic<bool> bad{false};
        if (!parse_workers_ && threads_ > 1 && n > (1u << 15)) parse_workers_.reset(new Workers(threads_ - 1));
        parallel_for(n, threads_, parse_workers_.get(), [&](size_t b, size_t e) {
            for (size_t i = b; i < e; ++i) {
                const uint8_t* r = base + at[i];
                const uint8_t* cg;
                uint32_t nc;
                if (!record_cigar(r, rd32(r - 4), &cg, &nc)) { bad.store(true); return; }
                out.cigar_off[i + 1] = nc;
                out.pos[i] = (int32_t)rd32(r + 4);
                out.mapq[i] = r[9];
                out.flag[i] = rd16(r + 14);
            }
        });
        if (bad.load()) { if (err) *err = "corrupt BAM record"; return -1; }
        g_tm.pass1 += Tm::now() - t1;
        t1 = Tm::now();
        uint64_t tot = 0;
        for (size_t i = 0; i < n; ++i) { tot += out.cigar_off[i + 1]; out.cigar_off[i + 1] = (uint32_t)tot; }
        if (tot > 0xffffffffull) { if (err) *err = "more than 2^32 CIGAR ops in one block"; return -1; }
        out.cigar.resize((size_t)tot);
        parallel_for(n, threads_, parse_workers_.get(), [&](size_t b, size_t e) {
            for (size_t i = b; i < e; ++i) {
                const uint8_t* r = base + at[i];
                const uint8_t* cg;
                uint32_t nc;
                record_cigar(r, rd32(r - 4), &cg, &nc);
                uint32_t* dst = out.cigar.data() + out.cigar_off[i];
                for (uint32_t k = 0; k < nc; ++k) dst[k] = rd32(cg + 4 * (size_t)k);
            }
        });
        g_tm.pass2 += Tm::now() - t1;
    }
    cur_ += p;
    return n ? 1 : 0;
}

}  // namespace gdh
