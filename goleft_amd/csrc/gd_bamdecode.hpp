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
//   gd_bam_extract_tab_kernel  (round 6) the counting walk leaves WHERE every record starts and how many ops
//                          its segment holds in front of it; the extraction is then a thread per record -- no
//                          chain, no LDS, no barrier -- and the second walk is not run
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
    // the record table (null: none).  Written by the count pass, read by gd_bam_extract_tab_kernel: two words per record,
    // {start - seg_beg[s], ops of the segment in front of it}, segment s at entry tab_base[s] (room for (seg_end - seg_beg)
    // / 36 + 1 records: a record is 36 bytes at least; the host keeps segments under 4 GB)
    uint32_t* tab;
    const uint64_t* tab_base;      // [n_seg]
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

// ONE WAVE PER SEGMENT.  Records are self-delimiting only forwards, so finding where they start is a chain of
// dependent reads.  Round 3 gave each segment ONE LANE, which followed that chain with a load from HBM per record and
// then extracted the record itself -- 3 300 records in a row per segment of a 30x chromosome, every lane of a wave in
// a different place.  Here the wave stages 3 KB of the stream in LDS with coalesced 16-byte loads, lane 0 follows the
// block_size chain THERE (an LDS round trip per record) and notes up to 32 record starts, and then every lane takes one
// record -- CIGAR resolved (CG:B,I), fields extracted -- with the round's records' memory latencies in flight at once.
// Output positions inside a round come from the op counts the lanes leave in LDS.  Measured with the page-locked walk
// tables of gd_api_ingest.inc (the two changes were made together): the counting walk of a chromosome-sized range
// 17 -> 10 ms, the extracting one 18 -> 6 ms (scope iii).
// LDS: 3 KB of stream + 0.5 KB of tables.  Not more: the walks run BESIDE the inflate kernel of the next range, whose six
// workgroups per CU leave 4 KB of LDS -- a walk workgroup that needs more waits for an inflate workgroup to retire.
constexpr int BW_WIN = 3072;           // bytes of the stream staged per round
constexpr int BW_REC = 32;             // records per round at most (one lane each)

template <bool EXTRACT>
__global__ __launch_bounds__(64) void gd_bam_walk_kernel(BamSegJob j)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_win[BW_WIN];
    __shared__ uint64_t s_off[BW_REC];     // stream offsets of the round's records
    __shared__ uint32_t s_nc[BW_REC];      // their op counts
    __shared__ uint32_t s_bad[BW_REC];
    __shared__ uint64_t s_wbase;           // where the next window begins (16-byte aligned)
    __shared__ uint32_t s_n, s_done;       // records of this round; the segment is finished
    const uint32_t s = blockIdx.x;
    if (s >= j.n_seg) return;
    const uint32_t lane = threadIdx.x;
    const uint64_t stop = j.seg_end[s];
    // lane 0's walk state
    uint64_t off = j.seg_beg[s];
    uint32_t nrec = 0, fl = 0;
    uint64_t nops = 0;
    int32_t first = 0x7fffffff, last = -0x7fffffff;
    // every lane's copy of where the round's output begins
    uint64_t ri = EXTRACT ? j.rec_base[s] : 0, oi = EXTRACT ? j.op_base[s] : 0;
    if (lane == 0) { s_wbase = off & ~15ull; s_done = off < stop ? 0u : 1u; s_n = 0; }
    __syncthreads();
    while (s_done == 0u) {
        // ---- (1) the window: [wbase, wbase + BW_WIN), bytes past the stream read as zero ----
        const uint64_t wbase = s_wbase;
        for (uint32_t i = lane; i < BW_WIN / 16; i += 64) {
            const uint64_t a = wbase + 16ull * i;
            uint32_t w[4] = {0, 0, 0, 0};
            if (a + 16 <= j.n_bytes) __builtin_memcpy(w, j.data + a, 16);
            else if (a < j.n_bytes) __builtin_memcpy(w, j.data + a, (size_t)(j.n_bytes - a));
            __builtin_memcpy(s_win + 16u * i, w, 16);
        }
        __syncthreads();
        // ---- (2) lane 0: the block_size chain, in LDS ----
        if (lane == 0) {
            uint32_t k = 0, done = 0;
            while (k < (uint32_t)BW_REC) {
                if (off >= stop) { done = 1; break; }
                const uint64_t rel = off - wbase;
                if (rel + 16 > (uint64_t)BW_WIN) break;                     // block_size, refID and POS of the record lie in the next window
                if (off + 36 > j.n_bytes) { fl |= 2u; done = 1; break; }
                const uint32_t block_size = ld32(s_win + rel);
                if (block_size < 32 || off + 4 + block_size > j.n_bytes) { fl |= 2u; done = 1; break; }
                const int32_t ref_id = (int32_t)ld32(s_win + rel + 4);
                if (ref_id != j.tid) {
                    // the contig's records end here -- in a sorted BAM.  `samtools depth -r` stops at the first record of
                    // another reference and so does this walk; that the reference does not RESUME is checked over the next
                    // records up to the segment's end (at most 64: one damaged refID is caught, a contig's true end costs
                    // nothing), and by the host across segments (bit 3)
                    fl |= 8u;
                    done = 1;
                    if (!(ref_id == -1 || (ref_id > j.tid && ref_id < j.n_ref))) { fl |= 2u; break; }   // not a refID a sorted BAM holds here: a damaged record (or a walk out of step)
                    if (!EXTRACT) {
                        uint64_t o2 = off;
                        for (int q = 0; q < 64 && o2 < stop; ++q) {
                            if (o2 + 36 > j.n_bytes) break;
                            const uint32_t bs = ld32(j.data + o2);
                            if (bs < 32 || o2 + 4 + bs > j.n_bytes) break;
                            if ((int32_t)ld32(j.data + o2 + 4) == j.tid) { fl |= 16u; break; }
                            o2 += 4ull + bs;
                        }
                    }
                    break;
                }
                const int32_t pos = (int32_t)ld32(s_win + rel + 8);
                // POS -1 is BAM's "no position": a record filed under the reference but not placed on it (what `samtools
                // depth` drops through the 0x4 flag such a record carries) is not part of the contig's stream
                if (pos >= 0) {
                    if (nrec && pos < last) fl |= 1u;
                    if (nrec == 0) first = pos;
                    last = pos;
                    ++nrec;
                    s_off[k++] = off;
                }
                off += 4ull + block_size;
            }
            if (!done && off >= stop) done = 1;
            s_n = k;
            s_done = done;
            s_wbase = off & ~15ull;
        }
        __syncthreads();
        // ---- (3) one record per lane: its CIGAR (the stored one or the CG tag's) ----
        const uint32_t n = s_n;
        const uint8_t* r = nullptr;
        const uint8_t* cg = nullptr;
        uint32_t nc = 0;
        if (lane < n) {
            const uint64_t o = s_off[lane];
            r = j.data + o + 4;
            // the counting walk needs the op count alone, and the record's fixed fields and stored CIGAR are nearly always
            // inside the window it was found in: read there, it costs no trip to memory (block_size, refID, POS and
            // l_read_name lie inside by the chain's own condition); the CG:B,I placeholder and records that leave the
            // window go the long way
            const uint32_t rel = (uint32_t)(o - wbase);
            const uint32_t l_read_name = s_win[rel + 12u];
            bool ok, near = false;
            if (!EXTRACT && rel + 44u + l_read_name <= (uint32_t)BW_WIN) {
                const uint32_t block_size = ld32(s_win + rel);
                const uint32_t n_cigar = (uint32_t)s_win[rel + 16u] | ((uint32_t)s_win[rel + 17u] << 8);
                const uint32_t c0 = ld32(s_win + rel + 36u + l_read_name), c1 = ld32(s_win + rel + 40u + l_read_name);
                const bool placeholder = n_cigar == 2u && (c0 & 0xfu) == 4u && (c0 >> 4) == ld32(s_win + rel + 20u) && (c1 & 0xfu) == 3u;
                if (!placeholder) { near = true; ok = 32ull + l_read_name + 4ull * n_cigar <= block_size; nc = n_cigar; }
            }
            if (!near) ok = bam_record_cigar(r, ld32(j.data + o), &cg, &nc);
            s_bad[lane] = ok ? 0u : 1u;
            s_nc[lane] = ok ? nc : 0u;
        }
        __syncthreads();
        uint32_t bad = 0;
        uint64_t before = 0, total = 0;                                     // ops of the round's records in front of this lane's; of all
        for (uint32_t q = 0; q < n; ++q) {
            bad |= s_bad[q];
            if (q < lane) before += s_nc[q];
            total += s_nc[q];
        }
        if (bad) {                                                          // a corrupt record: the host refuses the file
            if (lane == 0) { fl |= 2u; s_done = 1; }
        } else if (!EXTRACT && j.tab && lane < n) {
            uint32_t* const e = j.tab + 2ull * (j.tab_base[s] + ri + lane);
            e[0] = (uint32_t)(s_off[lane] - j.seg_beg[s]);
            e[1] = (uint32_t)(oi + before);
        } else if (EXTRACT && lane < n) {
            const uint64_t rr = ri + lane, oo = oi + before;
            j.pos[rr] = (int32_t)ld32(r + 4);
            j.mapq[rr] = r[9];
            j.flag[rr] = (uint16_t)((uint32_t)r[14] | ((uint32_t)r[15] << 8));
            j.cigar_off[rr] = (uint32_t)oo;
            for (uint32_t q = 0; q < nc; ++q) j.cigar[oo + q] = ld32(cg + 4 * (uint64_t)q);
        }
        ri += n;
        oi += total;
        if (lane == 0) nops += total;
        __syncthreads();                                                    // (s_done above; and nobody still reads what the next round overwrites)
    }
    if (lane == 0 && !EXTRACT) {
        if (off > stop) fl |= 4u;                                // an anchor that is not a record start
        j.n_rec[s] = nrec;
        j.n_ops[s] = nops;
        j.first_pos[s] = first;
        j.last_pos[s] = last;
        j.flags[s] = fl;
    }
}

// A THREAD PER RECORD over the table the counting walk left: a workgroup per anchor segment (its record count, bases and
// table entry come from the walk's tables), the fields and the CIGAR (the stored one or the CG tag's, found again) copied
// to the contig's arrays.  The records of neighbouring threads are neighbours in the stream and in the output.
__global__ __launch_bounds__(256) void gd_bam_extract_tab_kernel(BamSegJob j)
{
    const uint32_t s = blockIdx.x;
    if (s >= j.n_seg) return;
    const uint32_t n = j.n_rec[s];
    const uint64_t beg = j.seg_beg[s], rb = j.rec_base[s], ob = j.op_base[s];
    const uint32_t* const tab = j.tab + 2ull * j.tab_base[s];
    for (uint32_t k = threadIdx.x; k < n; k += blockDim.x) {
        const uint64_t o = beg + tab[2u * k];
        const uint8_t* const r = j.data + o + 4;
        const uint8_t* cg = nullptr;
        uint32_t nc = 0;
        (void)bam_record_cigar(r, ld32(j.data + o), &cg, &nc);      // (the walk has seen it succeed)
        const uint64_t rr = rb + k, oo = ob + tab[2u * k + 1u];
        j.pos[rr] = (int32_t)ld32(r + 4);
        j.mapq[rr] = r[9];
        j.flag[rr] = (uint16_t)((uint32_t)r[14] | ((uint32_t)r[15] << 8));
        j.cigar_off[rr] = (uint32_t)oo;
        for (uint32_t q = 0; q < nc; ++q) j.cigar[oo + q] = ld32(cg + 4 * (uint64_t)q);
    }
}

}  // namespace gd
