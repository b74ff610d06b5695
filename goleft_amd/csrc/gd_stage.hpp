// gd_stage.hpp -- records host -> HBM: one staging block (pos / flag / MAPQ / CSR offsets / ops in page-locked host
// memory, gd_acquire) appended to a contig's device arrays by ONE kernel that reads the host memory over the link.
// hipMemcpyAsync would be five copies per block through one DMA engine (measured 19 GB/s for the whole feed of a
// 30x chromosome); a grid of workgroups keeps enough 16-byte reads in flight to fill a Gen5 x16 link, and the five
// arrays travel in one launch.  Replaces, with the ring of gd_acquire / gd_commit, the BGZF / BAM read every
// `samtools depth` child performs (/root/reference/depth/depth.go:45) as the way records reach the arithmetic.
#pragma once

namespace gd {

struct H2DSeg { void* dst; const void* src; uint64_t bytes; };
struct H2DJob { H2DSeg seg[5]; uint32_t off_add; };   // seg[1] holds 32-bit CSR offsets: off_add is added to each (block relative -> contig stream)

__global__ __launch_bounds__(256) void gd_h2d_kernel(H2DJob job)
{
    const uint64_t gid = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    const uint64_t gsz = (uint64_t)gridDim.x * 256u;
#pragma unroll 1
    for (int s = 0; s < 5; ++s) {
        char* const dst = static_cast<char*>(job.seg[s].dst);
        const char* const src = static_cast<const char*>(job.seg[s].src);
        const uint64_t n = job.seg[s].bytes;
        if (n == 0) continue;
        if (s == 1) {                                          // CSR offsets: 4-byte elements, rebased on the way
            const uint32_t add = job.off_add;
            const uint64_t ne = n >> 2;
            const uint32_t* __restrict__ const s1 = reinterpret_cast<const uint32_t*>(src);
            uint32_t* __restrict__ const d1 = reinterpret_cast<uint32_t*>(dst);
            uint64_t h4 = ((16u - (uint64_t)(reinterpret_cast<uintptr_t>(dst) & 15u)) & 15u) >> 2;
            h4 = h4 < ne ? h4 : ne;
            for (uint64_t i = gid; i < h4; i += gsz) d1[i] = s1[i] + add;
            const uint64_t nv = (ne - h4) >> 2;
            if ((reinterpret_cast<uintptr_t>(s1 + h4) & 15u) == 0) {
                const uint4* __restrict__ const s4 = reinterpret_cast<const uint4*>(s1 + h4);
                uint4* __restrict__ const d4 = reinterpret_cast<uint4*>(d1 + h4);
                for (uint64_t i = gid; i < nv; i += gsz) {
                    uint4 v = s4[i];
                    v.x += add; v.y += add; v.z += add; v.w += add;
                    d4[i] = v;
                }
                for (uint64_t k = h4 + (nv << 2) + gid; k < ne; k += gsz) d1[k] = s1[k] + add;
            } else {
                for (uint64_t k = h4 + gid; k < ne; k += gsz) d1[k] = s1[k] + add;
            }
            continue;
        }
        // the staging arrays start 16-byte aligned; the destination is wherever the contig's array ends: bytes up
        // to its next 16-byte boundary one by one, then 16-byte vectors when the source is aligned there too (a
        // multiple of 16 bytes appended so far: every block but a contig's last), else 4-byte or single bytes
        uint64_t head = (16u - (uint64_t)(reinterpret_cast<uintptr_t>(dst) & 15u)) & 15u;
        head = head < n ? head : n;
        for (uint64_t i = gid; i < head; i += gsz) dst[i] = src[i];
        const uint64_t body = n - head;
        const char* const sb = src + head;
        char* const db = dst + head;
        if ((reinterpret_cast<uintptr_t>(sb) & 15u) == 0) {
            const uint64_t nv = body >> 4;
            const uint4* __restrict__ const s4 = reinterpret_cast<const uint4*>(sb);
            uint4* __restrict__ const d4 = reinterpret_cast<uint4*>(db);
            uint64_t i = gid;
            for (; i + 3 * gsz < nv; i += 4 * gsz) {          // four independent 16-byte reads in flight per lane
                const uint4 a = s4[i], b = s4[i + gsz], c2 = s4[i + 2 * gsz], d = s4[i + 3 * gsz];
                d4[i] = a; d4[i + gsz] = b; d4[i + 2 * gsz] = c2; d4[i + 3 * gsz] = d;
            }
            for (; i < nv; i += gsz) d4[i] = s4[i];
            for (uint64_t k = (nv << 4) + gid; k < body; k += gsz) db[k] = sb[k];
        } else if ((reinterpret_cast<uintptr_t>(sb) & 3u) == 0) {
            const uint64_t nv = body >> 2;
            const uint32_t* __restrict__ const s1 = reinterpret_cast<const uint32_t*>(sb);
            uint32_t* __restrict__ const d1 = reinterpret_cast<uint32_t*>(db);
            for (uint64_t i = gid; i < nv; i += gsz) d1[i] = s1[i];
            for (uint64_t k = (nv << 2) + gid; k < body; k += gsz) db[k] = sb[k];
        } else {
            for (uint64_t k = gid; k < body; k += gsz) db[k] = sb[k];
        }
    }
}

}  // namespace gd
