// bam_reader.hpp -- BGZF/BAM decode to the engine's SoA record blocks.
//
// Replaces the BAM read that every `samtools depth` child of the reference
// performs (/root/reference/depth/depth.go:45): BGZF members are inflated in
// parallel (libdeflate when the system has it, else zlib), records are decoded to {pos, flag, mapq, CIGAR} -- the only
// fields `samtools depth -Q q` without -q ever consults.
// Format: SAMv1 section 4 (BGZF, BAM), long-CIGAR convention 4.2.2 (CG:B,I).
#pragma once

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <future>
#include <memory>
#include <new>
#include <string>
#include <vector>

namespace gdh {

struct Workers;                                 // persistent worker threads (bam_reader.cpp)

int usable_cpus();                              // hardware threads, capped by the container's CPU quota

// A byte buffer that does NOT zero what resize() adds: a batch of inflated members is a quarter of a gigabyte that the
// inflate workers are about to overwrite, and std::vector's resize() fills it with zeros first -- on ONE thread, as long
// as all threads together then take to inflate into it (and 64 MB more per batch for the compressed bytes).
class Bytes {
public:
    Bytes() = default;
    ~Bytes() { free(p_); }
    Bytes(Bytes&& o) noexcept : p_(o.p_), n_(o.n_), cap_(o.cap_) { o.p_ = nullptr; o.n_ = o.cap_ = 0; }
    Bytes& operator=(Bytes&& o) noexcept
    {
        if (this != &o) { free(p_); p_ = o.p_; n_ = o.n_; cap_ = o.cap_; o.p_ = nullptr; o.n_ = o.cap_ = 0; }
        return *this;
    }
    Bytes(const Bytes&) = delete;
    Bytes& operator=(const Bytes&) = delete;
    size_t size() const { return n_; }
    size_t capacity() const { return cap_; }
    bool empty() const { return n_ == 0; }
    uint8_t* data() { return p_; }
    const uint8_t* data() const { return p_; }
    void clear() { n_ = 0; }
    void reserve(size_t c)                      // (exact: the batches are all of one size)
    {
        if (c <= cap_) return;
        uint8_t* q = static_cast<uint8_t*>(malloc(c));
        if (!q) throw std::bad_alloc();
        if (n_) memcpy(q, p_, n_);
        free(p_);
        p_ = q; cap_ = c;
    }
    void resize(size_t n) { reserve(n); n_ = n; }           // new bytes are NOT initialised
    void erase_front(size_t k)
    {
        if (k >= n_) { n_ = 0; return; }
        memmove(p_, p_ + k, n_ - k);
        n_ -= k;
    }
    void append(const uint8_t* src, size_t k)
    {
        if (n_ + k > cap_) reserve(std::max(n_ + k, cap_ + cap_ / 2));
        memcpy(p_ + n_, src, k);
        n_ += k;
    }

private:
    uint8_t* p_ = nullptr;
    size_t n_ = 0, cap_ = 0;
};

struct BamContig {
    std::string name;
    int64_t length;
};

// One block of decoded records of a single contig, coordinate order.
struct RecordBlock {
    int32_t tid = -1;
    std::vector<int32_t> pos;
    std::vector<uint16_t> flag;
    std::vector<uint8_t> mapq;
    std::vector<uint32_t> cigar_off;   // n+1, starts at 0
    std::vector<uint32_t> cigar;
    void clear()
    {
        tid = -1;
        pos.clear(); flag.clear(); mapq.clear(); cigar.clear();
        cigar_off.assign(1, 0);
    }
    size_t size() const { return pos.size(); }
};

class BamReader {
public:
    BamReader();
    ~BamReader();
    BamReader(const BamReader&) = delete;
    BamReader& operator=(const BamReader&) = delete;

    // Opens the file, inflates and parses the header.  threads <= 0: usable_cpus().
    bool open(const std::string& path, int threads, std::string* err);
    const std::vector<BamContig>& contigs() const { return contigs_; }
    const std::string& header_text() const { return text_; }

    // Position the stream at the first record of contig `tid` using the .bai
    // (path + ".bai"); returns false (stream untouched) when there is no usable
    // index or the contig has no records.
    bool seek_contig(int32_t tid, std::string* err);
    int seek_contig_ex(int32_t tid, std::string* err);   // 1 positioned, 0 no records (index), -1 no usable index, -2 I/O error

    // The .bai linear index (SAMv1 5.2): for every reference the sorted, distinct, non-zero
    // virtual offsets of record starts (one per 16 kb window that holds reads).  false when
    // there is no usable index next to the file.  Static: needs no open reader.
    // Optional: has_chunks[r] = reference r has chunks in a real bin (so "no linear index" cannot mean
    // "no records"), chunk_end[r] = the largest chunk end (virtual offset) of its bins: no record of r
    // lies past the BGZF member that holds it.
    static bool linear_index(const std::string& bam_path, std::vector<std::vector<uint64_t>>* per_ref,
                             std::string* err, std::vector<char>* has_chunks = nullptr,
                             std::vector<uint64_t>* chunk_end = nullptr);

    // Fills `out` with up to max_reads records, all of one contig (a block ends
    // at a contig change).  Records with refID < 0 are skipped and counted.
    // Returns 1 on success, 0 at end of file, -1 on error.
    int next_block(RecordBlock& out, size_t max_reads, std::string* err);

    uint64_t n_records() const { return n_records_; }
    uint64_t n_unplaced() const { return n_unplaced_; }

private:
    struct Chunk {                           // one batch of inflated BGZF members
        Bytes data;
        std::string err;
        bool end = false;                    // nothing more to read (or an error)
    };
    Chunk produce(Bytes spare);              // read + inflate the next batch (runs one batch ahead)
    void drop_prefetch();
    bool fill(std::string* err);             // append the next batch to buf_
    bool need(size_t n, std::string* err);   // make n decoded bytes available at cur_

    int fd_ = -1;                             // the file; read with pread at file_off_ by the inflate workers
    uint64_t file_off_ = 0;
    bool seekable_ = true;                    // (a pipe is read in order, by one thread)
    std::string path_;
    int threads_ = 1;
    // two sets: the producer inflates the next batch (on the prefetch thread) while the consumer extracts the records of this one
    std::unique_ptr<Workers> inflate_workers_, parse_workers_;
    bool eof_ = false;                        // producer: the file is exhausted
    bool done_ = false;                       // consumer: the last batch has been appended
    size_t n_fills_ = 0;                      // batches taken over by the consumer since open / seek
    size_t n_batches_ = 0;                    // batches produced since open / seek: the first one is small and nothing is read ahead of it
    Bytes spare_;                             // the buffer the next batch will be inflated into
    std::future<Chunk> prefetch_;
    Bytes raw_;                               // compressed batch
    Bytes buf_;                               // decoded bytes not yet consumed
    size_t cur_ = 0;
    std::vector<size_t> at_;                  // next_block: record offsets of the block being cut
    std::vector<BamContig> contigs_;
    std::string text_;
    uint64_t n_records_ = 0, n_unplaced_ = 0;
    std::vector<bool> left_;          // references whose run of records has ended (a sorted BAM never returns to one)
    int32_t last_ref_ = -2;           // reference of the last record seen (-2: none yet, or just after a seek)
};

}  // namespace gdh
