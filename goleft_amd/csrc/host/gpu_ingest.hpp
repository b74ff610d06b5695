// gpu_ingest.hpp -- host side of the device BAM read: which bytes of the file hold one
// reference's records (from the .bai linear index), and streaming them to the device decoder
// (gd_ingest_begin / _feed_fd / _decode, include/goleft_depth.h: inflate + record decode on the GPU).
// Shared by `goleft depth` and `multidepth`.
#pragma once

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdint>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <string>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#include "../../../include/goleft_depth.h"

namespace gdh {

// A read-only mapping of a whole file.
struct FileMap {
    const uint8_t* p = nullptr;
    size_t size = 0;
    int fd = -1;                                           // stays open: the bulk of the bytes is read with pread (gd_ingest_feed_fd)
    FileMap() = default;
    FileMap(const FileMap&) = delete;
    FileMap& operator=(const FileMap&) = delete;
    bool keep = false;                                     // a process about to exit leaves the unmapping (0.15 s for a touched 7 GB mapping) to the kernel
    ~FileMap()
    {
        if (keep) return;
        if (p) munmap(const_cast<uint8_t*>(p), size);
        if (fd >= 0) ::close(fd);
    }
    bool open(const std::string& path)
    {
        fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) return false;
        struct stat st;
        if (fstat(fd, &st) != 0 || st.st_size <= 0) { ::close(fd); fd = -1; return false; }
        void* m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) { ::close(fd); fd = -1; return false; }
        (void)madvise(m, (size_t)st.st_size, MADV_SEQUENTIAL);
        p = static_cast<const uint8_t*>(m);
        size = (size_t)st.st_size;
        return true;
    }
};

// One device pass of the BAM read: references refs[first..last] (indices into the wanted list) and the
// byte range [beg, end) of the file that holds all their records.
struct IngestPass {
    size_t first, last;
    uint64_t beg, end;
    // a reference whose byte range exceeds the pass size is read in PARTS cut at anchors of its linear index:
    // first == last, the part holds its anchors [a_lo, a_hi) and ends at anchor a_hi (a_hi == n_anchors: the last part)
    bool part = false;
    size_t a_lo = 0, a_hi = 0;
    double scale = 0;                  // the reference's byte range over this part's (first part: sizes the contig's arrays)
};

// Plans the passes for the wanted references refs (ascending reference ids).  start[r] is the file offset
// of the BGZF member in which reference r's first record lies (the first linear-index entry >> 16) and
// has[r] whether r has records at all.  Wanted references that follow each other in the file -- nothing
// with records between them -- share a pass while it stays under group_bytes; a pass ends with the member
// in which the next reference with records starts (records may straddle it), or at the end of the file.
// ref_end (optional): the largest chunk end of every reference's .bai bins (virtual offsets).  The last
// reference with records then ends one member past it instead of at the end of the file -- the unmapped
// tail of a WGS BAM (often gigabytes) is neither uploaded nor inflated.
// lin + part_bytes (optional): a pass that would hold ONE reference and more than 1.5 x part_bytes of the file is cut
// into parts of about part_bytes at anchors of that reference's linear index (lin[r]: strictly ascending virtual
// offsets of record starts); part k ends with the member that holds the first anchor of part k + 1 (that member is read
// and inflated by both parts: the records before the anchor belong to part k, the rest to part k + 1).
inline std::vector<IngestPass> plan_ingest_passes(const std::vector<uint64_t>& start, const std::vector<char>& has,
                                                  const std::vector<int32_t>& refs, uint64_t file_size,
                                                  uint64_t group_bytes, const std::vector<uint64_t>* ref_end = nullptr,
                                                  const std::vector<std::vector<uint64_t>>* lin = nullptr, uint64_t part_bytes = 0)
{
    std::vector<IngestPass> out;
    auto next_with_records = [&](size_t r) {
        size_t u = r + 1;
        while (u < has.size() && !has[u]) ++u;
        return u;
    };
    size_t i = 0;
    while (i < refs.size()) {
        if (!has[(size_t)refs[i]]) { ++i; continue; }
        const uint64_t beg = start[(size_t)refs[i]];
        size_t j = i;
        for (;;) {
            const size_t nx = next_with_records((size_t)refs[j]);
            size_t k = j + 1;
            while (k < refs.size() && !has[(size_t)refs[k]]) ++k;
            if (k >= refs.size() || (size_t)refs[k] != nx) break;
            if (start[nx] - beg > group_bytes) break;
            j = k;
        }
        uint64_t end = ~0ull;
        const size_t after = next_with_records((size_t)refs[j]);
        if (after < has.size()) end = start[after] + 65536 + 26;
        else if (ref_end && (size_t)refs[j] < ref_end->size() && (*ref_end)[(size_t)refs[j]] != 0)
            end = ((*ref_end)[(size_t)refs[j]] >> 16) + 2 * (65536 + 26);   // the member holding the last record's end, whole
        if (end > file_size) end = file_size;
        if (i == j && lin && part_bytes > 0 && end > beg && end - beg > part_bytes + part_bytes / 2 && (*lin)[(size_t)refs[i]].size() > 1) {
            const std::vector<uint64_t>& an = (*lin)[(size_t)refs[i]];
            const uint64_t total = end - beg;
            const size_t want = (size_t)((total + part_bytes - 1) / part_bytes);
            std::vector<size_t> cut{0};                    // anchor indices where parts begin
            for (size_t k = 1; k < want; ++k) {
                const uint64_t target = beg + (uint64_t)((unsigned __int128)total * k / want);
                // the first anchor whose member starts at or after the target
                size_t lo = cut.back() + 1, hi = an.size();
                while (lo < hi) { const size_t mid = (lo + hi) / 2; if ((an[mid] >> 16) < target) lo = mid + 1; else hi = mid; }
                if (lo >= an.size()) break;
                if ((an[lo] >> 16) > (an[cut.back()] >> 16)) cut.push_back(lo);   // (a part must start in a later member than the one before)
            }
            cut.push_back(an.size());
            for (size_t k = 0; k + 1 < cut.size(); ++k) {
                IngestPass ps{i, i, an[cut[k]] >> 16, end};
                if (k == 0) ps.beg = beg;
                if (k + 2 < cut.size()) ps.end = std::min<uint64_t>((an[cut[k + 1]] >> 16) + 65536 + 26, end);
                ps.part = true;
                ps.a_lo = cut[k]; ps.a_hi = cut[k + 1];
                ps.scale = ps.end > ps.beg ? (double)total / (double)(ps.end - ps.beg) : 0.0;
                out.push_back(ps);
            }
            if (cut.size() == 2) out.back().part = false;  // one part after all: the ordinary whole-reference pass
            i = j + 1;
            continue;
        }
        out.push_back(IngestPass{i, j, beg, end});
        i = j + 1;
    }
    return out;
}

// The BGZF members of a byte range: offsets (relative to the range), sizes, header sizes, ISIZE and CRC of every
// complete member.  A member's header says where the next one starts, so the walk is serial -- 0.2 s for the
// 230 k members of a 3.7 GB chromosome, a quarter of that file's whole read.  The .bai breaks the chain: the
// upper 48 bits of every linear-index entry are the file offset of a member, so the range is cut at ~16 of them
// and the pieces are walked by as many threads (each must end exactly where the next begins; anything else --
// a stale index -- falls back to the serial walk).
struct MemberTable {
    std::vector<uint64_t> off;
    std::vector<uint32_t> size, isize, crc;
    std::vector<uint16_t> hdr;
    size_t n = 0;
    void swap_into(MemberTable* o) { off.swap(o->off); size.swap(o->size); isize.swap(o->isize); crc.swap(o->crc); hdr.swap(o->hdr); std::swap(n, o->n); }
};

inline bool list_members_serial(const uint8_t* base, size_t nb, MemberTable* t, size_t guess)
{
    for (int attempt = 0; attempt < 2; ++attempt) {
        t->off.resize(guess); t->size.resize(guess); t->isize.resize(guess); t->crc.resize(guess); t->hdr.resize(guess);
        size_t nm = 0;
        const int rc = gd_bgzf_members(base, nb, guess, t->off.data(), t->size.data(), t->hdr.data(), t->isize.data(),
                                       t->crc.data(), &nm);
        if (rc == GD_OK) {
            t->n = nm;
            t->off.resize(nm); t->size.resize(nm); t->isize.resize(nm); t->crc.resize(nm); t->hdr.resize(nm);
            return true;
        }
        if (rc != GD_E_CAPACITY) return false;
        guess = nm;                                          // the exact count: second walk
    }
    return false;
}

// The same walk with pread: one read per member -- its trailer and the header that follows -- instead of one
// mapped page per member (a 7 GB file: 460 k pages faulted in, and 0.15 s to unmap them again at exit).
inline bool list_members_serial_fd(int fd, uint64_t file_beg, size_t nb, MemberTable* t)
{
    auto rd = [&](uint64_t off, uint8_t* dst, size_t n) {
        size_t got = 0;
        while (got < n) {
            const ssize_t r = pread(fd, dst + got, n - got, (off_t)(off + got));
            if (r <= 0) return false;
            got += (size_t)r;
        }
        return true;
    };
    const size_t guess = nb / 12000 + 64;
    t->off.clear(); t->size.clear(); t->isize.clear(); t->crc.clear(); t->hdr.clear();
    t->off.reserve(guess); t->size.reserve(guess); t->isize.reserve(guess); t->crc.reserve(guess); t->hdr.reserve(guess);
    constexpr size_t kHead = 96;
    uint8_t hdr[kHead], tl[8 + kHead];
    std::vector<uint8_t> big;
    size_t p = 0, have = 0;
    while (p + 18 <= nb) {
        if (have < 18) {
            have = std::min(kHead, nb - p);
            if (!rd(file_beg + p, hdr, have)) return false;
        }
        const uint8_t* h = hdr;
        if (h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4)) return false;
        const size_t xlen = (size_t)h[10] | ((size_t)h[11] << 8);
        if (p + 12 + xlen > nb) break;
        if (12 + xlen > have) {                             // an unusually long extra field
            big.resize(12 + xlen);
            if (!rd(file_beg + p, big.data(), big.size())) return false;
            h = big.data();
        }
        size_t q = 12, bsize = 0;
        while (q + 4 <= 12 + xlen) {
            const size_t slen = (size_t)h[q + 2] | ((size_t)h[q + 3] << 8);
            if (h[q] == 66 && h[q + 1] == 67 && slen == 2 && q + 6 <= 12 + xlen)
                bsize = ((size_t)h[q + 4] | ((size_t)h[q + 5] << 8)) + 1;
            q += 4 + slen;
        }
        if (bsize < 12 + xlen + 8) return false;
        if (p + bsize > nb) break;                          // a trailing partial member is ignored
        const size_t next = std::min(kHead, nb - (p + bsize));
        if (!rd(file_beg + p + bsize - 8, tl, 8 + next)) return false;
        t->off.push_back(p);
        t->size.push_back((uint32_t)bsize);
        t->hdr.push_back((uint16_t)(12 + xlen));
        t->crc.push_back((uint32_t)tl[0] | ((uint32_t)tl[1] << 8) | ((uint32_t)tl[2] << 16) | ((uint32_t)tl[3] << 24));
        t->isize.push_back((uint32_t)tl[4] | ((uint32_t)tl[5] << 8) | ((uint32_t)tl[6] << 16) | ((uint32_t)tl[7] << 24));
        p += bsize;
        have = next;
        memcpy(hdr, tl + 8, next);
    }
    t->n = t->off.size();
    return true;
}

inline bool list_members(const uint8_t* base, size_t nb, uint64_t beg, const std::vector<uint64_t>& member_starts,
                         MemberTable* out, unsigned max_threads = 16, size_t kMin = 64u << 20, int fd = -1)
{
    // fd >= 0: `base` is not dereferenced, the bytes come from offset beg of the file
    auto serial = [&](uint64_t b, size_t n, MemberTable* t) {
        return fd >= 0 ? list_members_serial_fd(fd, beg + b, n, t) : list_members_serial(base + b, n, t, n / 12000 + 64);
    };
    // (ranges under kMin bytes: not worth the threads)
    std::vector<uint64_t> cuts;                              // range-relative member starts, strictly inside the range
    if (nb >= kMin && max_threads > 1) {
        std::vector<uint64_t> c;
        for (uint64_t m : member_starts)
            if (m > beg && m - beg < nb) c.push_back(m - beg);
        std::sort(c.begin(), c.end());
        c.erase(std::unique(c.begin(), c.end()), c.end());
        const unsigned pieces = (unsigned)std::min<size_t>(max_threads, nb / std::max<size_t>(kMin / 4, 1));
        for (unsigned k = 1; k < pieces && !c.empty(); ++k) {
            const uint64_t target = (uint64_t)((unsigned __int128)nb * k / pieces);
            auto it = std::lower_bound(c.begin(), c.end(), target);
            if (it == c.end()) break;
            if (cuts.empty() || *it > cuts.back()) cuts.push_back(*it);
        }
    }
    if (cuts.empty()) return serial(0, nb, out);
    std::vector<uint64_t> seg_beg{0};
    seg_beg.insert(seg_beg.end(), cuts.begin(), cuts.end());
    const size_t ns = seg_beg.size();
    std::vector<MemberTable> part(ns);
    std::vector<char> ok(ns, 0);
    std::vector<std::thread> th;
    for (size_t k = 0; k < ns; ++k)
        th.emplace_back([&, k]() {
            const uint64_t b = seg_beg[k], e = k + 1 < ns ? seg_beg[k + 1] : (uint64_t)nb;
            ok[k] = serial(b, (size_t)(e - b), &part[k]);
            if (ok[k] && k + 1 < ns) {                         // an inner piece must be tiled exactly by its members
                const MemberTable& p = part[k];
                ok[k] = p.n != 0 && p.off[p.n - 1] + p.size[p.n - 1] == e - b;
            }
        });
    for (auto& t : th) t.join();
    for (size_t k = 0; k < ns; ++k)
        if (!ok[k]) return serial(0, nb, out);              // the index lied: walk the chain
    size_t total = 0;
    for (const MemberTable& p : part) total += p.n;
    out->off.resize(total); out->size.resize(total); out->isize.resize(total); out->crc.resize(total); out->hdr.resize(total);
    size_t w = 0;
    for (size_t k = 0; k < ns; ++k) {
        const MemberTable& p = part[k];
        for (size_t i = 0; i < p.n; ++i, ++w) {
            out->off[w] = p.off[i] + seg_beg[k]; out->size[w] = p.size[i]; out->isize[w] = p.isize[i];
            out->crc[w] = p.crc[i]; out->hdr[w] = p.hdr[i];
        }
    }
    out->n = total;
    return true;
}

// Records of the BAM references refs[0..n) (ascending reference ids that have records) -> engine
// contigs tids[0..n), decoded on the device.  lin = BamReader::linear_index() of the file.
// References that follow each other in the file share one pass while the pass stays under
// `group_bytes` of BGZF: an inflate pass costs ~0.05 s however small it is, and an assembly with
// thousands of small contigs would otherwise pay it per contig.  The passes form a pipeline that keeps
// the link busy: a lister thread walks the BGZF members of the next pass (pread: headers and trailers
// only); gd_ingest_begin uploads the table -- while the pass before is still being read;
// gd_ingest_feed_fd hands the byte range to a thread of the context (pread straight into page-locked
// buffers, upload, inflate launches) and returns at once; the pass before that is decoded meanwhile.
// Returns GD_OK (with *io_ok = false when the file does not look as the index says: the caller
// falls back to the host decoder) or a gd_* error.
inline int ingest_references_on_device(gd_ctx* ctx, const FileMap& fm, const std::vector<std::vector<uint64_t>>& lin,
                                       const std::vector<int32_t>& refs, const std::vector<int32_t>& tids,
                                       uint64_t* n_records, bool* io_ok, uint64_t group_bytes = 512ull << 20,
                                       const std::vector<uint64_t>* ref_end = nullptr, uint64_t part_bytes = 2ull << 30)
{
    *io_ok = true;
    *n_records = 0;
    const double t_enter = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    std::vector<uint64_t> start(lin.size(), 0);
    std::vector<char> has(lin.size(), 0);
    for (size_t r = 0; r < lin.size(); ++r)
        if (!lin[r].empty()) { has[r] = 1; start[r] = lin[r].front() >> 16; }
    const std::vector<IngestPass> passes = plan_ingest_passes(start, has, refs, fm.size, group_bytes, ref_end, &lin, part_bytes);
    {
        uint64_t largest = 0;
        for (const IngestPass& ps : passes) largest = std::max(largest, ps.end > ps.beg ? ps.end - ps.beg : 0);
        (void)gd_set_option(ctx, GD_OPT_INGEST_RANGE_HINT, (int64_t)largest);   // both buffer sets are allocated once, for the largest pass
    }
    // decode + release of the oldest pending pass: references refs[a..b], or one part of one reference
    auto decode_pass = [&](const IngestPass& ps) -> int {
        const size_t a = ps.first, b = ps.last;
        if (ps.part) {
            const std::vector<uint64_t>& an = lin[(size_t)refs[a]];
            uint64_t n = 0;
            const int rc = gd_ingest_decode_part(ctx, tids[a], refs[a], an.data() + ps.a_lo, ps.a_hi - ps.a_lo,
                                                 ps.a_hi < an.size() ? an[ps.a_hi] : 0, (ps.a_lo ? (unsigned)GD_PART_APPEND : 0u) | (unsigned)GD_PART_RELEASE,
                                                 ps.a_lo ? 0.0 : ps.scale, &n);
            if (rc != GD_OK) return rc;
            *n_records += n;
            return GD_OK;
        }
        for (size_t k = a; k <= b; ++k) {
            const std::vector<uint64_t>& an = lin[(size_t)refs[k]];
            if (an.empty()) continue;
            uint64_t n = 0;
            const int rc = gd_ingest_decode(ctx, tids[k], refs[k], an.data(), an.size(), &n);   // drops everything on error
            if (rc != GD_OK) return rc;
            *n_records += n;
        }
        return gd_ingest_release(ctx);
    };
    // GOLEFT_INGEST_TIMING=1: where a pass spends its wall clock (stderr; measurement only)
    const bool timing = getenv("GOLEFT_INGEST_TIMING") != nullptr;
    const bool from_mapping = getenv("GOLEFT_INGEST_MMAP") != nullptr;   // (measurement: the bytes through the mapping instead of pread)
    auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_list = 0, t_begin = 0, t_feed = 0, t_decode = 0;
    // The member tables are listed up to two passes ahead on a thread of their own.
    struct Listed { MemberTable mt; bool ok = false; double secs = 0; };
    std::vector<Listed> listed(passes.size());
    std::mutex mu;
    std::condition_variable cv;
    size_t n_listed = 0, n_taken = 0;                      // the lister stays at most two passes ahead
    bool stop = false;
    // threads that walk a pass's BGZF members (one pread per member: all system time).  Four keep two passes ahead of a
    // 45 GB/s read; sixteen spent 7 CPU-seconds per genome more than four (they contend inside the kernel) and, under a
    // container's CPU quota, got the whole process throttled
    const unsigned list_threads = getenv("GOLEFT_LIST_THREADS") ? (unsigned)std::max(1, atoi(getenv("GOLEFT_LIST_THREADS")))
                                                                : (usable_cpus() <= 16 ? 4u : 8u);
    std::thread lister([&]() {
        for (size_t k = 0; k < passes.size(); ++k) {
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return stop || k < n_taken + 2; });
                if (stop) return;
            }
            const IngestPass& ps = passes[k];
            const double t0 = now();
            if (ps.beg < ps.end) {
                std::vector<uint64_t> member_starts;       // what the .bai knows about where members begin
                for (size_t r = ps.first; r <= ps.last; ++r)
                    for (uint64_t v : lin[(size_t)refs[r]])
                        if ((v >> 16) >= ps.beg && (v >> 16) < ps.end) member_starts.push_back(v >> 16);
                listed[k].ok = list_members(fm.p + ps.beg, (size_t)(ps.end - ps.beg), ps.beg, member_starts, &listed[k].mt, list_threads,
                                            64u << 20, from_mapping ? -1 : fm.fd) &&
                               listed[k].mt.n != 0;
            }
            listed[k].secs = now() - t0;
            { std::lock_guard<std::mutex> lk(mu); n_listed = k + 1; }
            cv.notify_all();
        }
    });
    struct Joiner {
        std::thread& t; std::mutex& mu; std::condition_variable& cv; bool& stop;
        ~Joiner() { { std::lock_guard<std::mutex> lk(mu); stop = true; } cv.notify_all(); if (t.joinable()) t.join(); }
    } joiner{lister, mu, cv, stop};
    // Three passes are in flight: the one being fed (its bytes on the link, its first members inflating), the one before it
    // (its last members inflating) and the one before that, which the caller decodes meanwhile -- with only two, the decode
    // had to wait for the inflate tail of the pass that had JUST been fed before the next pass could be announced, and the
    // link stood still for that long (it matters once the link is fast: GD_OPT_INGEST_CU_SPLIT).
    std::vector<size_t> fed;                               // passes fed and not yet decoded, oldest first
    // (GOLEFT_INGEST_DEPTH: 2 or 3 passes in flight, measurement switch; the library holds up to three -- a build that held
    // four was measured, profiles/r12u_...: no difference)
    size_t depth = 3;
    if (const char* e = getenv("GOLEFT_INGEST_DEPTH")) depth = (size_t)std::min(3, std::max(2, atoi(e)));
    for (size_t pk = 0; pk < passes.size(); ++pk) {
        const IngestPass& ps = passes[pk];
        const uint64_t beg = ps.beg, end = ps.end;
        auto bad_file = [&]() { (void)gd_ingest_abort(ctx); *io_ok = false; *n_records = 0; return GD_OK; };
        if (beg >= end) return bad_file();
        const double t0 = now();
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return n_listed > pk; });
            n_taken = pk + 1;
        }
        cv.notify_all();
        if (!listed[pk].ok) return bad_file();
        MemberTable& mt = listed[pk].mt;
        const size_t nm = mt.n;
        std::vector<uint64_t>& moff = mt.off;
        std::vector<uint32_t>&msize = mt.size, &misize = mt.isize, &mcrc = mt.crc;
        std::vector<uint16_t>& mhdr = mt.hdr;
        int rc = GD_OK;
        const size_t used = (size_t)(moff[nm - 1] + msize[nm - 1]);   // a trailing partial member is not fed
        const double t1 = now();
        rc = gd_ingest_begin(ctx, used, beg, nm, moff.data(), msize.data(), mhdr.data(), misize.data(), mcrc.data());
        if (rc != GD_OK) { (void)gd_ingest_abort(ctx); return rc; }
        const double t2 = now();
        if (!from_mapping) {
            rc = gd_ingest_feed_fd(ctx, fm.fd, beg, used); // pread on the context's workers, straight into the staging buffers
        } else {
            const size_t piece = 32u << 20;
            for (size_t off = 0; off < used && rc == GD_OK; off += piece)
                rc = gd_ingest_feed(ctx, fm.p + beg + off, used - off < piece ? used - off : piece);
        }
        if (rc != GD_OK) { (void)gd_ingest_abort(ctx); return rc; }
        MemberTable().swap_into(&mt);                      // (the table is page-locked memory of the context now)
        const double t3 = now();
        fed.push_back(pk);
        // this pass is on its way (upload + inflate are asynchronous): now decode the oldest one, once two are behind it
        if (fed.size() > depth - 1) {
            rc = decode_pass(passes[fed.front()]);
            if (rc != GD_OK) return rc;
            fed.erase(fed.begin());
        }
        t_list += t1 - t0; t_begin += t2 - t1; t_feed += t3 - t2; t_decode += now() - t3;
    }
    int rc_last = GD_OK;
    while (!fed.empty() && rc_last == GD_OK) {
        const double t = now();
        rc_last = decode_pass(passes[fed.front()]);
        fed.erase(fed.begin());
        t_decode += now() - t;
    }
    if (rc_last != GD_OK) return rc_last;
    if (timing) {
        double t_listing = 0;
        for (const Listed& l : listed) t_listing += l.secs;
        fprintf(stderr, "{\"listing_thread_s\": %.4f, \"ingest_call_s\": %.4f}\n", t_listing, now() - t_enter);
        double lib[7] = {0, 0, 0, 0, 0, 0, 0};
        if (gd_ingest_timing(ctx, lib, 7) == GD_OK)
            fprintf(stderr, "{\"lib_read_s\": %.4f, \"lib_wait_link_s\": %.4f, \"lib_begin_s\": %.4f, \"lib_wait_inflate_s\": %.4f, "
                            "\"lib_count_walk_s\": %.4f, \"lib_alloc_s\": %.4f, \"lib_alloc_and_extract_walk_s\": %.4f}\n",
                    lib[0], lib[1], lib[2], lib[3], lib[4], lib[5], lib[6]);
    }
    if (timing)
        fprintf(stderr, "{\"ingest_list_members_s\": %.4f, \"begin_s\": %.4f, \"feed_s\": %.4f, \"decode_s\": %.4f}\n",
                t_list, t_begin, t_feed, t_decode);
    return rc_last;
}

// One reference (multidepth: one contig of one BAM).
inline int ingest_reference_on_device(gd_ctx* ctx, const FileMap& fm, const std::vector<std::vector<uint64_t>>& lin,
                                      int32_t ref_id, int32_t engine_tid, uint64_t* n_records, bool* io_ok)
{
    return ingest_references_on_device(ctx, fm, lin, std::vector<int32_t>{ref_id}, std::vector<int32_t>{engine_tid},
                                       n_records, io_ok);
}

}  // namespace gdh
