// gpu_ingest.hpp -- host side of the device BAM read: which bytes of the file hold one
// reference's records (from the .bai linear index), read them into page-locked memory and
// hand them to gd_ingest_bgzf (include/goleft_depth.h), which inflates and decodes on the GPU.
// Shared by `goleft depth` and `multidepth`.
#pragma once

#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../../include/goleft_depth.h"

namespace gdh {

// Page-locked staging buffer, grown as needed, released with the object.
struct PinnedStage {
    gd_ctx* ctx = nullptr;
    uint8_t* p = nullptr;
    size_t cap = 0;
    explicit PinnedStage(gd_ctx* c) : ctx(c) {}
    PinnedStage(const PinnedStage&) = delete;
    PinnedStage& operator=(const PinnedStage&) = delete;
    ~PinnedStage() { if (p) (void)gd_host_free(ctx, p); }
    int reserve(size_t n)
    {
        if (n <= cap) return GD_OK;
        if (p) { (void)gd_host_free(ctx, p); p = nullptr; cap = 0; }
        void* q = nullptr;
        const int rc = gd_host_alloc(ctx, n + n / 4, &q);
        if (rc != GD_OK) return rc;
        p = static_cast<uint8_t*>(q);
        cap = n + n / 4;
        return GD_OK;
    }
};

// Records of BAM reference ref_id -> engine contig engine_tid, decoded on the device.
// lin = BamReader::linear_index() of the file.  Returns GD_OK (with *io_ok = false when the file
// could not be read as expected: the caller falls back to the host decoder) or a gd_* error.
inline int ingest_reference_on_device(gd_ctx* ctx, FILE* fb, const std::vector<std::vector<uint64_t>>& lin,
                                      int32_t ref_id, int32_t engine_tid, PinnedStage* stage, uint64_t* n_records,
                                      bool* io_ok)
{
    *io_ok = true;
    *n_records = 0;
    const std::vector<uint64_t>& a = lin[(size_t)ref_id];
    if (a.empty()) return GD_OK;                                  // no records on this reference
    const uint64_t beg = a.front() >> 16;
    // up to the member in which the next reference with records starts (inclusive), or EOF
    uint64_t end = ~0ull;
    for (size_t u = (size_t)ref_id + 1; u < lin.size(); ++u)
        if (!lin[u].empty()) { end = (lin[u].front() >> 16) + 65536 + 26; break; }
    if (fseeko(fb, 0, SEEK_END) != 0) { *io_ok = false; return GD_OK; }
    const uint64_t fsize = (uint64_t)ftello(fb);
    if (end > fsize) end = fsize;
    if (beg >= end || fseeko(fb, (off_t)beg, SEEK_SET) != 0) { *io_ok = false; return GD_OK; }
    const size_t nb = (size_t)(end - beg);
    if (int rc = stage->reserve(nb)) return rc;
    if (fread(stage->p, 1, nb, fb) != nb) { *io_ok = false; return GD_OK; }
    return gd_ingest_bgzf(ctx, engine_tid, ref_id, stage->p, nb, beg, a.data(), a.size(), n_records);
}

}  // namespace gdh
