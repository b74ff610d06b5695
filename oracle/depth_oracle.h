/*
 * depth_oracle.h -- CPU restatement of the `goleft depth` hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under goleft_amd/ may include, link or
 * call this.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker.
 *
 * PARITY STATUS: "parity unpinned" at per-base granularity.  The reference
 * (brentp/goleft @ v0.2.6) does not compute per-base depth itself; it shells
 * out to an external, un-vendored, un-pinned `samtools depth`
 * (/root/reference/depth/depth.go:45, .travis.yml:13).  The reference holds no
 * golden vectors for this path (depth/test/cmp.py:12 compares window means
 * against a live samtools with tolerance 0.5).  Neither Go nor samtools exist
 * in the build image, so the oracle is a restatement of
 *   - samtools >= 1.13 `depth -Q q` counting semantics (published behaviour,
 *     bam2depth.c), and
 *   - depth/depth.go:73-100,122-159,181-234,238-364 (restated line by line).
 * It is pinned only against (i) an independent brute-force Python counter
 * (oracle/pyoracle.py) and (ii) the survey-derived known answers for
 * depth/test/t.bam, via tests/golden/.
 */
#ifndef GOLEFT_DEPTH_ORACLE_H
#define GOLEFT_DEPTH_ORACLE_H

#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

/* samtools depth default read filter: UNMAP|SECONDARY|QCFAIL|DUP. */
#define GDO_DEFAULT_FLAG_MASK 0x704u

/* One contig's decoded records, coordinate sorted.  BAM CIGAR encoding
 * (len<<4|op, op in MIDNSHP=X = 0..8). */
typedef struct {
    const int32_t*  pos;        /* 0-based leftmost reference position */
    const uint16_t* flag;
    const uint8_t*  mapq;
    const uint32_t* cigar_off;  /* n+1 CSR offsets into cigar[] */
    const uint32_t* cigar;
    size_t          n;
} gdo_reads;

/* Per-base depth over the 0-based half-open region [start,end) as
 * `samtools depth -Q q -r chr:start+1-end` (>= 1.13) would count it:
 * a read is dropped if flag & flag_mask or mapq < q; M,=,X add one to each
 * covered reference position inside the region; D,N advance the reference
 * without counting; I,S,H,P do not consume reference.  out has end-start
 * entries (zero filled here). */
void gdo_perbase(const gdo_reads* r, int q, uint32_t flag_mask,
                 int64_t start, int64_t end, int32_t* out);

/* Same result through +1/-1 boundary marks and one prefix sum -- the shape of
 * samtools' no-base-quality fast path; used as the timed CPU baseline. */
void gdo_perbase_diff(const gdo_reads* r, int q, uint32_t flag_mask,
                      int64_t start, int64_t end, int32_t* out);

/* depth/depth.go:223-234 getCovClass.  0 NO_COVERAGE 1 LOW_COVERAGE
 * 2 CALLABLE 3 EXCESSIVE_COVERAGE. */
int gdo_cov_class(int depth, int mincov, int maxmeandepth);
const char* gdo_cov_class_name(int cls);

/* depth/depth.go:73-94 chromStartEndFromLine.  Returns 0 on success.
 * chrom is written NUL terminated into chrom[cap]. */
int gdo_chrom_start_end(const char* line, size_t len, char* chrom, size_t cap,
                        long* start, long* end);

/* depth/depth.go:122-159 genCommands tiling for one contig: step is
 * max(1, 10000000/W)*W; regions are [i, min(i+step,len)) 0-based half-open.
 * Writes up to cap (start,end) pairs; returns the number of tiles. */
size_t gdo_tiles(long length, int windowsize, long* starts, long* ends, size_t cap);
long gdo_step(int windowsize);

/* depth/depth.go:238-364 callback, restated line by line.  The per-base text
 * stream samtools would print for the region (positions with depth > 0 only)
 * is replaced by the depth vector of the region (depth[i] is the depth at
 * 0-based position region_start+i, region_end-region_start entries; entries
 * past the contig end must be 0).  Appends rows to the two FILE*s exactly as
 * the reference writes its tmp.depth.bed / tmp.callable.bed (no --stats). */
void gdo_callback(const char* chrom, long region_start, long region_end,
                  const int32_t* depth, int windowsize, int mincov,
                  int maxmeandepth, FILE* depth_bed, FILE* callable_bed);

/* ctypes convenience: same as gdo_callback, appending to two files. */
int gdo_callback_append(const char* chrom, long region_start, long region_end,
                        const int32_t* depth, int windowsize, int mincov,
                        int maxmeandepth, const char* depth_path,
                        const char* callable_path);

#ifdef __cplusplus
}
#endif
#endif
