// gd_bamdecode.hpp -- BAM records -> the engine's SoA arrays, on the device.
//
// Second stage of the device-side BAM read (after gd_inflate.hpp; replaces the decode every
// `samtools depth` child performs, /root/reference/depth/depth.go:45).  BAM records are
// self-delimiting only forwards (block_size prefix), so a stream cannot be entered at an
// arbitrary byte -- but the .bai linear index stores the virtual offset of a record start
// for every 16 kb of reference (SAMv1 section 5.2: ioffset).  Those are the anchors: one
// lane walks the records from one anchor to the next, so a chromosome offers thousands of
// independent walks.
//   gd_bam_count_kernel    per segment: records and CIGAR ops (CG:B,I tags resolved), first /
//                          last position, a sortedness flag; stops at the contig's end
//   gd_bam_extract_kernel  the same walk, writing pos / flag / mapq / cigar_off / cigar at the
//                          segment's base (exclusive prefix sums of the counts, host side)
// Only the fields `samtools depth -Q q` consults are extracted.
#pragma once

namespace gd {

struct BamSegJob {
    const uint8_t* data;           // inflated bytes
    uint64_t n_bytes;
    const uint64_t* seg_beg;       // [n_seg] byte offset of the first record of the segment
    const uint64_t* seg_end;       // [n_seg] where the walk stops (next anchor, or n_bytes)
    int32_t  tid;                  // records of another reference end the contig
    int32_t  n_ref;                // references of the file: a sorted BAM goes on with tid < refID < n_ref, or -1
    uint32_t n_seg;
    // count pass
    uint32_t* n_rec;               // [n_seg]
    uint64_t* n_ops;               // [n_seg]
    int32_t*  first_pos;           // [n_seg] (0x7fffffff when empty)
    int32_t*  last_pos;            // [n_seg]
    uint32_t* flags;               // [n_seg] bit0 unsorted inside, bit1 corrupt record, bit2 walk overran seg_end,
                                   // bit3 a record of another reference ended the walk before seg_end, bit4 ... and
                                   // the reference resumes within the next 64 records
    // extract pass
    const uint64_t* rec_base;      // [n_seg] first record index of the segment
    const uint64_t* op_base;       // [n_seg] first op index
    int32_t*  pos;
    uint16_t* flag;
    uint8_t*  mapq;
    uint32_t* cigar_off;           // [n_records + 1]; [n_records] is written by the host
    uint32_t* cigar;
};

__device__ __forceinline__ uint32_t ld32(const uint8_t* p)
{
    uint32_t v;
    __builtin_memcpy(&v, p, 4);    // records are not aligned
    return v;
}

// The CIGAR of the record body r (block_size bytes): the stored one, or the CG:B,I tag's when the
// stored one is the <l_seq>S<ref_len>N placeholder (SAMv1 4.2.2).  false: corrupt.
__device__ __forceinline__ bool bam_record_cigar(const uint8_t* r, uint32_t block_size, const uint8_t** cg_out,
                                                 uint32_t* n_out)
{
    const uint32_t l_read_name = r[8];
    uint32_t n_cigar = (uint32_t)r[12] | ((uint32_t)r[13] << 8);
    const uint32_t l_seq = ld32(r + 16);
    if (32ull + l_read_name + 4ull * n_cigar > block_size) return false;
    const uint8_t* cg = r + 32 + l_read_name;
    if (n_cigar == 2 && (ld32(cg) & 0xf) == 4 && (ld32(cg) >> 4) == l_seq && (ld32(cg + 4) & 0xf) == 3) {
        const uint8_t* end = r + block_size;
        const uint8_t* t = cg + 8 + (l_seq + 1) / 2 + l_seq;
        while (t + 3 <= end) {
            const uint8_t t0 = t[0], t1 = t[1], ty = t[2];
            t += 3;
            uint32_t sz = 0;
            if (ty == 'A' || ty == 'c' || ty == 'C') sz = 1;
            else if (ty == 's' || ty == 'S') sz = 2;
            else if (ty == 'i' || ty == 'I' || ty == 'f') sz = 4;
            else if (ty == 'Z' || ty == 'H') { while (t < end && *t) ++t; ++t; continue; }
            else if (ty == 'B') {
                if (t + 5 > end) break;
                const uint8_t sub = t[0];
                const uint32_t cnt = ld32(t + 1);
                t += 5;
                const uint32_t es = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : 4;
                if (t0 == 'C' && t1 == 'G' && sub == 'I' && t + 4ull * cnt <= end) { cg = t; n_cigar = cnt; break; }
                t += (uint64_t)es * cnt;
                continue;
            } else break;
            t += sz;
        }
    }
    *cg_out = cg;
    *n_out = n_cigar;
    return true;
}

template <bool EXTRACT>
__global__ __launch_bounds__(64) void gd_bam_walk_kernel(BamSegJob j)
{
    const uint32_t s = blockIdx.x * 64 + threadIdx.x;
    if (s >= j.n_seg) return;
    uint64_t off = j.seg_beg[s];
    const uint64_t stop = j.seg_end[s];
    uint32_t nrec = 0, fl = 0;
    uint64_t nops = 0;
    int32_t first = 0x7fffffff, last = -0x7fffffff;
    uint64_t ri = EXTRACT ? j.rec_base[s] : 0, oi = EXTRACT ? j.op_base[s] : 0;
    while (off < stop) {
        if (off + 36 > j.n_bytes) { fl |= 2u; break; }
        const uint8_t* r = j.data + off + 4;
        const uint32_t block_size = ld32(j.data + off);
        if (block_size < 32 || off + 4 + block_size > j.n_bytes) { fl |= 2u; break; }
        const int32_t ref_id = (int32_t)ld32(r);
        if (ref_id != j.tid) {
            // the contig's records end here -- in a sorted BAM.  `samtools depth -r` stops at the first record of
            // another reference and so does this walk; that the reference does not RESUME is checked over the next
            // records up to the segment's end (at most 64: one damaged refID is caught, a contig's true end costs
            // nothing), and by the host across segments (bit 3)
            fl |= 8u;
            if (!(ref_id == -1 || (ref_id > j.tid && ref_id < j.n_ref))) { fl |= 2u; break; }   // not a refID a sorted BAM holds here: a damaged record (or a walk out of step)
            if (!EXTRACT) {
                uint64_t o2 = off;
                for (int k = 0; k < 64 && o2 < stop; ++k) {
                    if (o2 + 36 > j.n_bytes) break;
                    const uint32_t bs = ld32(j.data + o2);
                    if (bs < 32 || o2 + 4 + bs > j.n_bytes) break;
                    if ((int32_t)ld32(j.data + o2 + 4) == j.tid) { fl |= 16u; break; }
                    o2 += 4ull + bs;
                }
            }
            break;
        }
        const int32_t pos = (int32_t)ld32(r + 4);
        // POS -1 is BAM's "no position": a record filed under the reference but not placed on it (what `samtools
        // depth` drops through the 0x4 flag such a record carries) is not part of the contig's stream
        if (pos < 0) { off += 4ull + block_size; continue; }
        const uint8_t* cg;
        uint32_t nc;
        if (!bam_record_cigar(r, block_size, &cg, &nc)) { fl |= 2u; break; }
        if (nrec && pos < last) fl |= 1u;
        if (nrec == 0) first = pos;
        last = pos;
        if (EXTRACT) {
            j.pos[ri] = pos;
            j.mapq[ri] = r[9];
            j.flag[ri] = (uint16_t)((uint32_t)r[14] | ((uint32_t)r[15] << 8));
            j.cigar_off[ri] = (uint32_t)oi;
            for (uint32_t k = 0; k < nc; ++k) j.cigar[oi + k] = ld32(cg + 4 * (uint64_t)k);
            ++ri;
            oi += nc;
        }
        ++nrec;
        nops += nc;
        off += 4ull + block_size;
    }
    if (off > stop) fl |= 4u;                                // an anchor that is not a record start
    if (!EXTRACT) {
        j.n_rec[s] = nrec;
        j.n_ops[s] = nops;
        j.first_pos[s] = first;
        j.last_pos[s] = last;
        j.flags[s] = fl;
    }
}

}  // namespace gd
