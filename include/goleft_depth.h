/*
 * goleft_depth.h -- C ABI of the MI355X-native per-base depth engine.
 *
 * This is the drop-in boundary for ONE path of brentp/goleft: the `depth`
 * subcommand (reference paths below are relative to /root/reference).
 * The reference has no FFI for this path: its seam is a subprocess per
 * 10 Mb tile -- `samtools depth -Q q -d D -r chr:s-e bam` spawned through
 * gargs `process.Runner` (depth/depth.go:45, :392-394) whose text output is
 * re-parsed by the `callback` closure (depth/depth.go:238-364).  This library
 * replaces that subprocess, the text pipe and the per-line parse; the host
 * keeps flags, tiling and BED formatting (INTEGRATION.md shows the cgo stub).
 *
 * Conventions
 *   - plain C: opaque context, plain pointers and sizes, no C++/torch types;
 *   - every call returns 0 (GD_OK) or a negative gd_status; the library never
 *     exits or aborts (the Go host folds failures into its exit code the way
 *     depth/depth.go:395-399 does);
 *   - the library never retains caller host pointers after a call returns
 *     (cgo rule); staging memory handed out by gd_acquire is library-owned
 *     pinned memory; device pointers given to gd_adopt_device stay owned by
 *     the caller and must outlive the context or the next gd_reset;
 *   - all results are integers (int32 per base, int64 window sums, int32
 *     window minima, {start,end,class} runs); `%.4g` and BED text stay on the
 *     host so results are bit-exact by construction;
 *   - coordinates are 0-based half-open; CIGARs use the BAM encoding
 *     (len<<4 | op, op in MIDNSHP=X = 0..8), i.e. biogo `sam.CigarOp` memory.
 *   - calls on one context must be serialised by the caller; distinct
 *     contexts (one per device / per producer thread) are independent.
 */
#ifndef GOLEFT_DEPTH_H
#define GOLEFT_DEPTH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GD_ABI_VERSION 15

typedef enum {
    GD_OK = 0,
    GD_E_INVALID = -1,   /* bad argument */
    GD_E_NOMEM = -2,     /* host or device allocation failed */
    GD_E_HIP = -3,       /* a HIP runtime call failed (see gd_last_error) */
    GD_E_STATE = -4,     /* call out of order (e.g. results before compute) */
    GD_E_RANGE = -5,     /* tid / coordinate out of range */
    GD_E_NODEVICE = -6,  /* no usable gfx950 device */
    GD_E_UNSORTED = -7,  /* records not coordinate sorted within a contig */
    GD_E_CAPACITY = -8   /* caller buffer too small (needed size reported) */
} gd_status;

/* Coverage classes, depth/depth.go:223-234 getCovClass. */
enum { GD_NO_COVERAGE = 0, GD_LOW_COVERAGE = 1, GD_CALLABLE = 2, GD_EXCESSIVE_COVERAGE = 3 };

/* samtools depth default read filter UNMAP|SECONDARY|QCFAIL|DUP. */
#define GD_DEFAULT_FLAG_MASK 0x704u

typedef struct gd_ctx gd_ctx;

/* Mirrors the `dargs` fields that reach the arithmetic (depth/depth.go:27-41;
 * defaults depth/depth.go:164-167). */
typedef struct {
    int32_t  window_size;     /* --windowsize, default 250 */
    int32_t  min_mapq;        /* -Q, default 1 (samtools depth -Q) */
    int32_t  min_cov;         /* --mincov, default 4 */
    int32_t  max_mean_depth;  /* --maxmeandepth, default 0 (EXCESSIVE class off) */
    uint32_t flag_mask;       /* reads with flag & mask are dropped; 0x704 */
    int32_t  max_span_hint;   /* expected max reference span of a read (0 = default 512);
                                 only a performance hint: the engine verifies it on device
                                 and transparently re-runs with the observed maximum */
    int64_t  step;            /* callable runs are split at multiples of this
                                 (depth/depth.go:48,:132; quirk Q1); 0 = derive
                                 max(1,10000000/W)*W */
} gd_params;

/* One block of decoded records, SoA.  For gd_acquire the pointers are pinned
 * host memory owned by the library; for gd_adopt_device they are device
 * pointers owned by the caller. */
typedef struct {
    int32_t*  pos;        /* [reads_cap]   0-based leftmost position */
    uint16_t* flag;       /* [reads_cap]   BAM FLAG */
    uint8_t*  mapq;       /* [reads_cap]   MAPQ */
    uint32_t* cigar_off;  /* [reads_cap+1] CSR offsets into cigar[], block relative */
    uint32_t* cigar;      /* [ops_cap]     BAM-encoded CIGAR ops */
    size_t    reads_cap;
    size_t    ops_cap;
    int32_t   slot;       /* ring slot (library use) */
} gd_batch;

typedef struct {
    int32_t start;        /* 0-based */
    int32_t end;          /* exclusive */
    int32_t cls;          /* GD_NO_COVERAGE .. GD_EXCESSIVE_COVERAGE */
} gd_run;

typedef struct {
    uint64_t n_reads;        /* records resident on the device */
    uint64_t n_ops;          /* CIGAR ops resident on the device */
    uint64_t n_ref_bases;    /* sum of contig lengths computed */
    uint64_t n_windows;
    uint64_t n_tiles;        /* LDS tiles (= workgroups of the tile kernel) */
    uint64_t n_runs;         /* callable runs found */
    int32_t  tile_positions; /* reference positions per LDS tile */
    int32_t  lookback;       /* look-back span in use */
    int32_t  max_span_seen;  /* largest reference span among kept reads */
    int32_t  reruns;         /* times the last gd_compute re-ran (span / run capacity) */
    int32_t  path;           /* GD_PATH_TILE / _SCATTER / _CHUNK: what the last gd_compute ran */
    int32_t  n_slow_tiles;   /* tile path with GD_OPT_FAST_KERNEL: tiles that took the generic kernel instead */
    uint64_t n_canonical_ops; /* (ABI 13: ops of canonical CIGARs; always 0 since ABI 14 -- every path reads the original ops) */
    int32_t  tile_kernel;    /* GD_TK_*: the kernel that did the per-base arithmetic of the last gd_compute */
    int32_t  reserved_;
    uint64_t n_deletions;    /* long-read path: entries of the deletion lists the tile kernel read (8 bytes each) */
} gd_stats;

/* gd_stats.tile_kernel */
enum { GD_TK_NONE = 0,
       GD_TK_GENERIC = 1,      /* gd_tile_kernel: any tile shape, CIGARs in any form */
       GD_TK_FAST = 2,         /* (ABI 13: the straight-line kernel on canonical records; not reported since ABI 14) */
       GD_TK_FAST_RAW = 3,     /* gd_tile_fast_kernel on the records as they arrived (+ the slow list) */
       GD_TK_LONG = 4,         /* gd_ltile2_kernel (long-read path) */
       GD_TK_SCATTER = 5,      /* gd_expand_scatter_kernel + gd_scan_kernel */
       GD_TK_SUMS_STREAM = 6,  /* (ABI 13: the streaming sums kernel over canonical records; not reported since ABI 14) */
       GD_TK_TILE_SUMS = 7,    /* gd_tile_sums_kernel (GD_OUT_SUMS_ONLY otherwise) */
       GD_TK_SUMS_STREAM_RAW = 8 };  /* gd_sums_stream_kernel over the records as they arrived */

/* Kernel ids for gd_kernel_ms.  Tile path: PREP, TILE, RUNS.  Long-read path: PREP, TILE (the long-read tile
 * kernel), RUNS; CKPT = its deletion lists + tile indexes, built when the records arrive (like NORM: summed over
 * the contigs since gd_set_profiling was last called, not cleared by gd_compute).
 * Scatter path: PREP (zero-fill + init), EXPAND (CIGAR expand + scatter), SCAN
 * (in-place scan + window / class reductions), RUNS. */
enum { GD_K_PREP = 0, GD_K_TILE = 1, GD_K_RUNS = 2, GD_K_EXPAND = 3, GD_K_SCAN = 4, GD_K_CKPT = 5,
       GD_K_SEQSTATS = 6,   /* the kernel of the last gd_seq_stats */
       GD_K_MDFLAGS = 7,    /* the kernel of the last gd_md_flags */
       GD_K_INFLATE = 8,    /* the kernel of the last gd_inflate_bgzf */
       GD_K_NORM = 9,       /* CIGAR normalisation (gd_adopt_device, gd_ingest_finish, or the first gd_compute
                               after gd_commit): summed over the contigs normalised since gd_set_profiling
                               was last called; not cleared by gd_compute */
       GD_K_COUNT = 10 };

/* Device algorithm of gd_compute.  All are bit exact; they differ in cost.
 *   TILE     one workgroup per 4096-position tile re-walks the CIGARs of the
 *            reads that start within one maximum read span before it (LDS
 *            difference array, fused scan): the short-read path, ~6.5 HBM bytes
 *            per base;
 *   CHUNK    the same tile shape for long reads (ONT/PacBio, spliced): one pass
 *            over the CIGARs leaves a reference-position checkpoint every 64 ops
 *            and every read's end; a tile then expands only the 64-op chunks that
 *            reach it (LDS marks, no global atomics, any read span);
 *   SCATTER  every CIGAR op is expanded once and scattered with global integer
 *            atomics, then one in-place scan pass: no look-back at all, kept for
 *            pathological span distributions and as a cross-check;
 *   AUTO     CHUNK when records average more than 6 CIGAR ops or a read spans
 *            more than 32768 reference bases, else TILE (default). */
enum { GD_PATH_AUTO = 0, GD_PATH_TILE = 1, GD_PATH_SCATTER = 2, GD_PATH_CHUNK = 3 };

const char* gd_strerror(int status);
int         gd_abi_version(void);
int         gd_device_count(int* n);

/* Create a context bound to one HIP device (hipSetDevice is re-issued inside
 * every entry point, so calls may come from any OS thread / goroutine). */
int  gd_create(int device_id, gd_ctx** out);
void gd_destroy(gd_ctx* ctx);
const char* gd_last_error(const gd_ctx* ctx);

/* Optional: run all work on a caller-provided hipStream_t (passed as void*). */
int gd_set_stream(gd_ctx* ctx, void* hip_stream);

int gd_set_params(gd_ctx* ctx, const gd_params* p);
int gd_default_params(gd_params* p);
/* Choose the device algorithm (GD_PATH_*); default GD_PATH_AUTO. */
int gd_set_path(gd_ctx* ctx, int path);

/* Switches (defaults are what the measurements in DESIGN.md chose; results are bit identical under every setting).  A C
 * library behind cgo takes no behaviour from the process environment: these are calls.
 * (Numbers 1, 2, 4, 11 belonged to tile shapes and to "canonical records", measured slower / without a caller: retired.) */
enum { GD_OPT_NT_STORES = 3,        /* 1 (default): non-temporal per-base stores; 0: plain */
       GD_OPT_FAST_KERNEL = 5,      /* 1 (default): ordinary tiles run the straight-line tile kernel, the rest the
                                       generic one; 0: the generic kernel for every tile */
       GD_OPT_COPY_THREADS = 6,     /* host threads filling the staging buffer of gd_ingest_feed: 1 (default) .. 16 */
       GD_OPT_PUSH_THREADS = 7,     /* host threads of gd_push copying into a pinned ring block: 16 (default), 1 .. 64
                                       (one core moves ~11 GB/s into pinned memory; the link takes five times that) */
       GD_OPT_H2D_KERNEL = 8,       /* how a committed staging block reaches HBM: 1 (default) one kernel whose workgroups
                                       read the page-locked block over the link (all five arrays in one launch; n > 1:
                                       with n workgroups), 0: five hipMemcpyAsync through a DMA engine */
       GD_OPT_PUSH_CHUNK = 9,       /* records per staging block of gd_push: 2^20 (default), 4096 .. 2^24 */
       GD_OPT_BAM_REFS = 10,        /* gd_ingest_*: number of references in the BAM header (0, the default: unknown).  The
                                       record walk takes the first record of ANOTHER reference as the end of a contig's
                                       records only when its refID is one a sorted BAM can hold there (greater than the
                                       contig's and below this number, or -1); anything else is a damaged record */
       GD_OPT_INGEST_CRC = 12,      /* 1 (default): gd_ingest_* checks the CRC32 of every BGZF member after inflating it, as htslib
                                       does; 0: the file is trusted (a second pass over the inflated bytes is saved) */
       GD_OPT_INGEST_DMA = 13,      /* gd_ingest_feed*: how a staged piece crosses the link: 1 (default) a copy command on one stream;
                                       0: a copy kernel on a high-priority stream (what the CLI uses, with GD_OPT_INGEST_CU_SPLIT).
                                       (2 .. 4, slices of a piece on several streams, measured slower: measurement builds only) */
       GD_OPT_INGEST_INDEX = 14,    /* 1 (default): records are indexed as they arrive -- gd_adopt_device's check pass, a pass over
                                       every committed block once it has landed, the device BAM read's write pass: a position
                                       index (first read at or past every 64th position: the prep kernel looks its tiles' read
                                       ranges up instead of searching) and the largest reference span of any record (the first
                                       gd_compute starts with the right look-back instead of learning it; still verified).
                                       0: neither (the prep kernel searches, the look-back starts at max_span_hint or 512) */
       GD_OPT_INGEST_RANGE_HINT = 18, /* bytes of the LARGEST range gd_ingest_begin will be given (0, the default: unknown): the range
                                       buffers are allocated for it on first use instead of growing -- freeing and allocating --
                                       whenever a later range is larger than the ones before */
       GD_OPT_INGEST_CU_SPLIT = 19,   /* with GD_OPT_INGEST_DMA = 0: every n-th CU (2 .. 64; 0, the default: off) runs only the copy kernel that
                                       pulls staged pieces over the link, the other CUs only the inflate launches (CU-masked
                                       streams); set before the first gd_ingest_begin */
       GD_OPT_INFLATE_KERNEL = 20,    /* which kernel inflates BGZF members: 0 (default) a lane per member (gd_inflate.hpp); 1: a workgroup
                                       per member with the member's output in LDS (gd_inflate_wave.hpp, round 6; members it does not take
                                       go to the other kernel on the same stream).  Both produce zlib's bytes.  The second one moves a
                                       sixth of the bytes through memory and is the slower of the two on an MI355X (DESIGN.md 3.5 has the
                                       measurements and why); it is kept as a second implementation the tests compare the first with.
                                       (ABI 14 had a measurement switch with this number that produced wrong bytes: gone) */
       GD_OPT_INGEST_COPY_GRID = 25,  /* with GD_OPT_INGEST_DMA = 0: workgroups (of 256 threads) of the copy kernel that pulls a staged piece
                                       over the link: 16 (default), 1 .. 4096.  The link needs ~100 KB of reads in flight; a large grid
                                       keeps MEGABYTES of reads to host memory outstanding, and every other kernel on the device then
                                       runs 2.6 times slower (measured: inflate launches 62.7 ms with 512 workgroups, 24.1 ms behind a
                                       copy engine; the genome read 2.11 s with 512, 1.52 - 1.70 s with 8 .. 32, 1.76 s with 4 where
                                       the link starts to starve) */
       GD_OPT_COMMIT_CHECK = 23,      /* where the records of a gd_commit are checked (coordinate order, no negative position, CSR
                                       offsets non-decreasing): 0 (default) on the host, inside gd_commit -- a second pass by CPU
                                       threads over a block the producer has just written; 1: by the pass that indexes the block
                                       on the device once it has landed (it runs anyway), for blocks of 4096 records or more.
                                       gd_commit then returns before the verdict exists: it comes from gd_check_commits, or
                                       from the next gd_compute, which refuses to run on records that failed (the context
                                       needs a gd_reset then).  For producers that are short of CPU -- a decoder's threads --
                                       and do not need the verdict block by block; gdh_produce_in_place runs this way */

       /* ---- MEASUREMENT BUILDS ONLY (the library compiled with -DGD_MEASURE) ----
        * Each of these was built, measured neutral or worse on an MI355X and left at its default (HISTORY.md has the numbers).
        * A release library accepts the default value and answers GD_E_INVALID to any other; it also contains none of the
        * timing code of such builds (section cycle counters in the inflate kernels, marks in gd_ingest_begin). */
       GD_OPT_INGEST_PIECE_STREAMS = 15, /* whole staged pieces alternate over this many streams: 1 (default) .. 4 */
       GD_OPT_INFLATE_LDS_PAD = 16,   /* bytes of LDS a lane-per-member inflate workgroup claims on top of its own (an occupancy limiter): 0 */
       GD_OPT_INGEST_HYBRID = 17,     /* with several piece streams: the pieces of every stream but the first leave through a copy kernel: 0 */
       GD_OPT_INGEST_BATCHES = 21,    /* inflate launches per fed range: 8 */
       GD_OPT_INGEST_WALK_CUS = 22 }; /* with GD_OPT_INGEST_CU_SPLIT: the record walks on the copy kernel's CUs: 0 */
int gd_set_option(gd_ctx* ctx, int option, int64_t value);
/* The value an option has now (a library that scopes a setting puts it back afterwards). */
int gd_get_option(gd_ctx* ctx, int option, int64_t* value);

/* What is built FROM the records: nothing on the short-read tile path and for the streaming sums (their kernels read the
 * records as they arrived); on the long-read path (GD_PATH_CHUNK, or GD_PATH_AUTO and more than 6 CIGAR ops per record)
 * deletion lists, read records and tile indexes -- by the first gd_compute that needs them, in one pass over the original
 * CIGARs, kept until the records change.  gd_drop_derived forgets them: the state right after the records arrived
 * (measurement: one gd_compute from there is what one `goleft depth` run pays).  gd_rebuild_derived builds the ones the
 * selected contigs currently hold again, into the device block they already occupy -- the work a fresh input costs,
 * without the allocator's.  (ABI 13 also had gd_normalize / gd_canonical_cigars: canonical records, removed.) */
int gd_drop_derived(gd_ctx* ctx);
int gd_rebuild_derived(gd_ctx* ctx);

/* Which results gd_compute materialises in HBM.  GD_OUT_PERBASE (default): the
 * int32 per-base vector (12.4 GB for a human genome), needed by gd_perbase,
 * gd_device_perbase and the --bed region reductions.  Without it only window
 * sums/minima and class runs are produced -- all `goleft depth` prints for a whole
 * genome, and what a cohort (depthwed) needs; tile and chunk paths only. */
enum { GD_OUT_PERBASE = 1,
       /* Window sums ONLY (no per-base vector, no minima, no class runs): all that depth.bed's mean
        * column and the depthwed matrix need.  The short-read path then makes ONE streaming pass over the
        * records as they arrived, every read adding its overlap with the one or two windows it touches -- no tiles,
        * no per-base scan at all (window_size >= 32; smaller windows and the long-read path silently run the
        * regular windows-only kernels).  gd_callable and minima report GD_E_STATE.  Excludes
        * GD_OUT_PERBASE. */
       GD_OUT_SUMS_ONLY = 2 };
int gd_set_outputs(gd_ctx* ctx, unsigned flags);

/* Reference sequence table (@SQ LN of the BAM header / .fai lengths,
 * depth/depth.go:134-149).  Drops all records and results.  Lengths are 0 .. GD_MAX_CONTIG_LENGTH (BAM itself
 * stops at 2^31 - 1; the last 64 Ki positions are given up so that tile arithmetic stays in 32 bits), else
 * GD_E_RANGE. */
#define GD_MAX_CONTIG_LENGTH 0x7fff0000LL
int gd_set_contigs(gd_ctx* ctx, int n_contigs, const int64_t* lengths);

/* Restrict computation to a subset of contigs (--chrom, depth/depth.go:145).
 * n == 0 selects all. */
int gd_select_contigs(gd_ctx* ctx, int n, const int32_t* tids);

/* ---- record ingest: pinned ring buffers -> HBM (replaces the BGZF/BAM read
 * that each `samtools depth` child performs) --------------------------------*/

/* Borrow a pinned staging block with at least the given capacities.  Waits
 * for the block's previous copy if that is still in flight.  A producer may
 * hold up to three blocks at a time (its threads write block k+1 while
 * gd_commit validates and sends block k -- gdh_produce_in_place does); blocks
 * go back to the ring in the order they were handed out, and GD_E_STATE
 * answers a gd_acquire that comes round to a block still held.  gd_commit
 * with n_reads 0 gives a block back unused; gd_reset gives all of them back. */
int gd_acquire(gd_ctx* ctx, size_t reads_cap, size_t ops_cap, gd_batch* out);

/* Append n_reads records (n_ops ops) of contig tid from a block obtained from
 * gd_acquire; the H2D copy is asynchronous on the copy stream and the block
 * returns to the ring when it completes.  Records of one contig must be
 * committed in coordinate order (GD_E_UNSORTED); a negative position is
 * GD_E_RANGE (a placed BAM record has POS >= 0).  The order of the commits is
 * the order of the records; whether it succeeds or not, the block is no longer
 * the caller's afterwards (a second commit of it is GD_E_STATE). */
int gd_commit(gd_ctx* ctx, const gd_batch* b, int32_t tid, size_t n_reads, size_t n_ops);

/* With GD_OPT_COMMIT_CHECK = 1: waits for the committed blocks to land and returns what their checks found --
 * GD_OK, GD_E_UNSORTED, GD_E_RANGE (a negative position) or GD_E_INVALID (CSR offsets) -- for everything
 * committed since the last call (or the last gd_compute / gd_reset).  Nothing to report and nothing waited
 * for when no block has been checked on the device. */
int gd_check_commits(gd_ctx* ctx);

/* Optional: room for n_reads more records / n_ops more CIGAR ops of contig tid in one step.  A producer that knows
 * its totals (the .bai metadata pseudo-bin holds a reference's mapped-record count) spares the device arrays their
 * growth by doubling, each step of which waits for the copies in flight. */
int gd_reserve(gd_ctx* ctx, int32_t tid, size_t n_reads, size_t n_ops);

/* Convenience: copy records from ordinary host memory (copies before
 * returning; never keeps the pointers). */
int gd_push(gd_ctx* ctx, int32_t tid, const int32_t* pos, const uint16_t* flag,
            const uint8_t* mapq, const uint32_t* cigar_off, const uint32_t* cigar,
            size_t n_reads, size_t n_ops);

/* Use records already resident in HBM (zero copy).  Replaces any records of
 * that contig.  The call waits for the device (whatever stream produced the
 * arrays) and checks them the way gd_commit checks a host block -- positions in
 * coordinate order (GD_E_UNSORTED) and not negative (GD_E_RANGE), CSR offsets non-decreasing from 0 and ending
 * inside the op array (GD_E_INVALID); one pass over pos / cigar_off.  Arrays aligned to 16 (pos, cigar_off),
 * 8 (flag) and 4 (mapq) bytes -- anything hipMalloc or a tensor allocator returns -- get the straight-line
 * tile kernel; others the generic one. */
int gd_adopt_device(gd_ctx* ctx, int32_t tid, const gd_batch* dev, size_t n_reads, size_t n_ops);

/* Drop records and results, keep contigs/params/allocations. */
int gd_reset(gd_ctx* ctx);

/* ---- compute -------------------------------------------------------------*/

/* Per-base depth for every selected contig, fused with the per-window
 * sum/min reduction and the coverage-class run-length encoding (what
 * `samtools depth` + callback compute per tile).  Synchronous: returns when
 * results are ready. */
int gd_compute(gd_ctx* ctx);
/* The same in two halves: gd_compute_launch enqueues every kernel and the read-back copies on the context's
 * stream and returns without waiting; gd_compute_finish waits, verifies (re-running synchronously when the
 * look-back / capacity checks ask for it) and publishes the results.  Between the two the host is free -- e.g.
 * to issue the collective of the PREVIOUS step while this one's kernels run (bench.py --gpus N).  Exactly one
 * compute may be in flight per context; until gd_compute_finish, result calls and calls that change the job
 * (records, contigs, parameters, path, outputs, options, gd_ingest_*, gd_md_*) are refused with GD_E_STATE --
 * gd_compute_launch starts overwriting the result arrays.  gd_set_export may be called in between (it takes
 * effect with the next launch). */
int gd_compute_launch(gd_ctx* ctx);
int gd_compute_finish(gd_ctx* ctx);

/* ---- results -------------------------------------------------------------*/

/* depth[start..end) of contig tid into host memory. */
int gd_perbase(gd_ctx* ctx, int32_t tid, int64_t start, int64_t end, int32_t* out);

/* Window sums (and optionally minima; mins may be NULL) for W-anchored
 * windows of contig tid: window k covers [k*W, min((k+1)*W, len)).
 * *n receives the window count; GD_E_CAPACITY if cap is too small. */
int gd_windows(gd_ctx* ctx, int32_t tid, int64_t* sums, int32_t* mins, size_t cap, size_t* n);

/* Coverage-class runs of contig tid in coordinate order, split at multiples
 * of params.step (reference quirk Q1) and nowhere else. */
int gd_callable(gd_ctx* ctx, int32_t tid, gd_run* out, size_t cap, size_t* n);

/* --bed mode (depth/depth.go:103-120): reductions over an arbitrary region of
 * the already computed per-base vector.  Windows stay anchored at absolute
 * multiples of W and are clipped to [start,end) (depth/depth.go:297-298);
 * positions past the contig end have depth 0. */
int gd_region_windows(gd_ctx* ctx, int32_t tid, int64_t start, int64_t end,
                      int64_t* sums, int32_t* mins, size_t cap, size_t* n);
int gd_region_callable(gd_ctx* ctx, int32_t tid, int64_t start, int64_t end,
                       gd_run* out, size_t cap, size_t* n);
/* The same two reductions for MANY regions in one call (a --bed file: one launch per reduction for
 * the whole batch instead of several launches, allocations and synchronisations per row).  Region r
 * is [start[r], end[r]) on contig tid[r].  Its W-anchored windows are sums/mins[win_off[r] ..
 * win_off[r+1]) and its class runs runs[run_off[r] .. run_off[r+1]); win_off and run_off have
 * n_regions + 1 entries and are written by the call.  mins may be NULL.  GD_E_CAPACITY when
 * cap_windows or cap_runs is too small: win_off[n_regions] / run_off[n_regions] then hold what is
 * needed (a windows shortfall is reported first, with run_off zeroed). */
int gd_regions(gd_ctx* ctx, size_t n_regions, const int32_t* tid, const int64_t* start, const int64_t* end,
               int64_t* sums, int32_t* mins, size_t cap_windows, size_t* win_off,
               gd_run* runs, size_t cap_runs, size_t* run_off);

/* ---- depthwed on device (depthwed/depthwed.go; BASELINE.json config 4) -----
 * A cohort is loaded as n_samples x n_ctg contigs of ONE context (tids[s*n_ctg+j]
 * = engine contig holding sample s, reference contig j; the n_samples contigs of
 * one j must have equal length).  After gd_compute, builds the sites x samples
 * matrix `goleft depthwed -s size` would print from the samples' depth.bed files:
 * rows = groups of ceil(size/W) consecutive windows per contig (the last group of
 * a contig may be shorter), cell = sum over the group's windows of
 * int(0.5 + "%.4g"(window_sum/window_len)) -- the text round trip is reproduced
 * exactly in arithmetic, no text is formatted.  cells is row major
 * [n_rows][n_samples]; row_ctg/row_start/row_end (each may be NULL) describe the
 * rows.  *n_rows receives the row count; GD_E_CAPACITY if cap_rows is too small. */
int gd_depthwed(gd_ctx* ctx, int n_samples, int n_ctg, const int32_t* tids, int64_t size,
                int64_t* cells, int32_t* row_ctg, int64_t* row_start, int64_t* row_end,
                size_t cap_rows, size_t* n_rows);
/* Same matrix, left in HBM for the next consumer (the normalisation / CNV steps that read
 * depthwed output, e.g. dcnv/dcnv.go:153-186): *d_cells is a device pointer to
 * [n_rows][n_samples] int64, valid until the next gd_depthwed* / gd_destroy. */
int gd_depthwed_device(gd_ctx* ctx, int n_samples, int n_ctg, const int32_t* tids, int64_t size,
                       const int64_t** d_cells, size_t* n_rows);

/* ---- `--stats` columns on device (depth/depth.go:191-200, :244-252) ----------
 * The reference appends "%.3g" of GC, CpG and masked fraction of each window's
 * reference bases (faidx.Stats, an external module: PARITY UNPINNED, semantics
 * restated in oracle/pyoracle.py::seq_stats).  gd_seq_load copies ONE contig's
 * bases (FASTA line breaks removed) into HBM, replacing the previous one; the
 * pointer is not retained.  gd_seq_stats then returns, for each window
 * [start[k], end[k]) clipped to the contig: n_gc = bases in "GCgc", n_masked =
 * lower-case bases, n_cpg = C/c followed by G/g (the following base may lie past
 * the window, not past the contig).  Integer counts only: the divisions by
 * (end - start) and the "%.3g" stay on the host.  Does not need gd_compute. */
int gd_seq_load(gd_ctx* ctx, const uint8_t* seq, int64_t len);
int gd_seq_stats(gd_ctx* ctx, size_t n_windows, const int64_t* start, const int64_t* end,
                 uint32_t* n_gc, uint32_t* n_cpg, uint32_t* n_masked);
/* The same plus what the other plausible reading of faidx.Stats needs (goleft_depth_host.h GDH_STATS_*; the
 * caller picks the contract, the library only counts): n_acgt = bases in "ACGTacgt" (a denominator that
 * skips N and IUPAC codes), n_masked_acgt = bases in "acgt"; either may be NULL.  line_bases > 0 (bases per
 * FASTA line, the .fai LINEBASES column): a C that is the last base of a line -- or of the window -- does not
 * start a CpG: what a scan of the window's raw, line-broken bytes sees; 0: the sequence is one line and the
 * base after the window counts. */
int gd_seq_stats_ex(gd_ctx* ctx, size_t n_windows, const int64_t* start, const int64_t* end, int32_t line_bases,
                    uint32_t* n_gc, uint32_t* n_cpg, uint32_t* n_masked, uint32_t* n_acgt, uint32_t* n_masked_acgt);

/* ---- multidepth on device (multidepth/multidepth.go) ----------------------------
 * The S samples are S equally long contigs tids[0..S) of ONE context, computed by
 * gd_compute with -Q = multidepth's Q (per-base output kept).  gd_md_flags replaces
 * the multi-file `samtools depth` text stream and its parse (:203-207, :148-161)
 * plus sufficientDepth (:163-171) with two bitmaps over the contig (bit x of word
 * x/32, little endian): any = some sample has depth > 0 (the positions samtools
 * prints), suf = MORE THAN min_samples samples have depth >= min_cov.  The block
 * state machine (:217-258) then runs on the host over the bitmaps.  gd_md_sums
 * returns, for each block [start, end) and sample, the reference's running sum
 * `dps[i] += float64(d) / 1000.` over the block's suf sites in position order
 * (means, :270-277) as the exact IEEE double; sums is row major [n_blocks][S] and
 * uses the samples and bitmaps of the last gd_md_flags.  GD_E_CAPACITY if n_words
 * < ceil(len/32). */
int gd_md_flags(gd_ctx* ctx, int n_samples, const int32_t* tids, int32_t min_cov, int32_t min_samples,
                uint32_t* any_bits, uint32_t* suf_bits, size_t n_words);
int gd_md_sums(gd_ctx* ctx, size_t n_blocks, const int64_t* start, const int64_t* end, double* sums);
/* The same bitmaps with the samples brought in GROUPS, so that only one group's per-base vectors have to be
 * resident (gd_select_contigs + gd_compute per group; the reference bounds memory with 5 Mb position chunks,
 * :114,126): gd_md_begin(len) zeroes the per-position accumulators, gd_md_accumulate adds a group (the tids of
 * the last gd_compute), gd_md_finish turns them into `any` / `suf` (copied out when the pointers are non-NULL)
 * and makes them the bitmaps gd_md_blocks and gd_md_sums_group use.  At most 65535 samples. */
int gd_md_begin(gd_ctx* ctx, int64_t len);
int gd_md_accumulate(gd_ctx* ctx, int n_samples, const int32_t* tids, int32_t min_cov);
int gd_md_finish(gd_ctx* ctx, int32_t min_samples, uint32_t* any_bits, uint32_t* suf_bits, size_t n_words);
/* Bitmaps made elsewhere become the current ones (tests; a host with its own depth source). */
int gd_md_load_flags(gd_ctx* ctx, const uint32_t* any_bits, const uint32_t* suf_bits, int64_t len);
/* The block state machine on the device (aggregate + splitBlocks, multidepth.go:188-268, for every chunk of
 * `chunk` positions, in chunk order -- what `multidepth -p 1` prints): blocks [start, end) over the current
 * bitmaps.  *n_blocks receives their number; GD_E_CAPACITY (with *n_blocks set) when cap is too small.  The
 * quirks of the reference are kept: nothing is reported before a chunk's first insufficient site, the last
 * cache of a chunk skips the min_size test, neighbouring chunks may report overlapping blocks. */
int gd_md_blocks(gd_ctx* ctx, int64_t chunk, int32_t max_skip, int32_t min_size, int32_t window,
                 int64_t* starts, int64_t* ends, size_t cap, size_t* n_blocks);
/* gd_md_sums for the samples tids[0..n_samples) of the last gd_compute (one group): sums is [n_blocks][n_samples]. */
int gd_md_sums_group(gd_ctx* ctx, int n_samples, const int32_t* tids, size_t n_blocks, const int64_t* start,
                     const int64_t* end, double* sums);

/* ---- BGZF inflate on device (the first stage of the BAM read of depth/depth.go:45) ----
 * A BGZF file is a sequence of independent <= 64 KiB DEFLATE streams ("members").  The
 * caller lists them (payload offset/length inside data, ISIZE, and where each member's
 * bytes go in out); one GPU lane inflates one member and checks it against crc[m], the
 * CRC32 of the gzip trailer (crc may be NULL: not checked).  status[m] is 0 or a decoder
 * error code (corrupt stream, ISIZE mismatch, 18 = CRC32 mismatch).  data, out and status
 * are host buffers. */
int gd_inflate_bgzf(gd_ctx* ctx, const uint8_t* data, size_t n_bytes, size_t n_members,
                    const uint64_t* in_off, const uint32_t* in_len, const uint64_t* out_off,
                    const uint32_t* out_len, const uint32_t* crc, uint8_t* out, size_t out_bytes,
                    uint32_t* status);

/* ---- the whole BAM read of one contig on the device -------------------------------------
 * data: a byte range of the BAM file that begins at a BGZF member boundary (its file offset
 * is base_coffset) and holds every record of the BAM reference ref_id, which become the records
 * of engine contig tid (the same number for `goleft depth`; multidepth maps one reference of S
 * files onto S contigs).  anchors: virtual file offsets
 * (coffset << 16 | uoffset, as stored in the .bai) of record starts inside the range,
 * strictly ascending, anchors[0] = the contig's first record -- the .bai linear index
 * provides one per 16 kb of reference (SAMv1 5.2).  The device inflates the members (one
 * lane each), one wave per anchor walks the records up to the next anchor (a record of
 * another reference ends the walk; it counts them and notes where each starts), a thread
 * per record then extracts, and {pos, flag, mapq, CIGAR (CG:B,I resolved)} become
 * the contig's record arrays in HBM, replacing what it held -- the state gd_push / gd_commit
 * would have left, without any decode on the host.  Every member's CRC32 is verified.
 * Errors: GD_E_INVALID (corrupt member or record, CRC mismatch, anchor that is not a record
 * start), GD_E_UNSORTED. */
int gd_ingest_bgzf(gd_ctx* ctx, int32_t tid, int32_t ref_id, const uint8_t* data, size_t n_bytes,
                   uint64_t base_coffset, const uint64_t* anchors, size_t n_anchors, uint64_t* n_records);
/* The same read as a stream, so that file I/O overlaps the device work: list the range's
 * members first (gd_bgzf_members, headers and trailers only), announce them
 * (gd_ingest_begin: allocations), then feed the range's bytes in order in pieces of any
 * size -- each piece is copied into page-locked staging (the pointer is not retained),
 * uploaded asynchronously, and every member that is complete on the device is inflated
 * behind the copy while the caller reads the next piece -- and finish with the anchors.
 * gd_ingest_bgzf is begin + one feed + finish.  gd_ingest_abort drops an unfinished read.
 * A fed range may hold several references (a BAM with thousands of small contigs: one inflate
 * pass has a latency floor of ~0.04 s whatever its size): gd_ingest_decode is gd_ingest_finish
 * without the release, so call it once per reference of the range (each with that reference's
 * anchors and its own contig), then gd_ingest_release -- or gd_ingest_finish for the last one.
 * Three ranges may be pending at a time: once a range is completely fed, gd_ingest_begin / _feed of
 * the NEXT range may run before the first is decoded, so that the last inflate launches of one
 * range (latency bound) overlap the upload of the next -- and with a third, the decode of the oldest
 * never holds the upload up; gd_ingest_decode, _finish and _release
 * always act on the oldest pending range, gd_ingest_feed on the newest.  A fourth gd_ingest_begin is
 * refused (GD_E_STATE, nothing changes); gd_ingest_begin while the newest range is only partly fed
 * abandons what is pending, like gd_ingest_abort (which drops everything).  An error while feeding or
 * decoding drops everything pending. */
int gd_bgzf_members(const uint8_t* data, size_t n_bytes, size_t cap, uint64_t* member_off, uint32_t* member_size,
                    uint16_t* header_size, uint32_t* isize, uint32_t* crc, size_t* n_members);
int gd_ingest_begin(gd_ctx* ctx, uint64_t n_bytes, uint64_t base_coffset, size_t n_members,
                    const uint64_t* member_off, const uint32_t* member_size, const uint16_t* header_size,
                    const uint32_t* isize, const uint32_t* crc);
int gd_ingest_feed(gd_ctx* ctx, const uint8_t* bytes, size_t n);
/* gd_ingest_feed with the bytes taken from an open file: n bytes from `offset` of `fd` are read (pread on the
 * context's worker threads, GD_OPT_PUSH_THREADS of them) straight into the page-locked staging buffers -- none
 * of the page faults of a mapping, no second copy.  The read runs on a thread of the context and the call returns
 * at once (fd must stay open): gd_ingest_decode / _release of the OLDEST pending range and gd_ingest_begin of
 * the next one proceed meanwhile; every other gd_ingest_* call waits for the read first and reports its error
 * (GD_E_INVALID: the file ended early or could not be read). */
int gd_ingest_feed_fd(gd_ctx* ctx, int fd, uint64_t offset, size_t n);
int gd_ingest_finish(gd_ctx* ctx, int32_t tid, int32_t ref_id, const uint64_t* anchors, size_t n_anchors,
                     uint64_t* n_records);
int gd_ingest_decode(gd_ctx* ctx, int32_t tid, int32_t ref_id, const uint64_t* anchors, size_t n_anchors,
                     uint64_t* n_records);
int gd_ingest_release(gd_ctx* ctx);
/* A reference read in SEVERAL ranges (passes cut inside a chromosome at .bai anchors: buffers of one pass instead of
 * one chromosome, and only the last pass's inflate and record walks left without an upload to hide behind): the
 * oldest pending range holds the reference's records from anchors[0] up to `end_anchor` -- the virtual offset of the
 * first record that belongs to the NEXT part (a record start the index knows, inside a member of this range; 0: the
 * part runs to the range's end, as gd_ingest_decode).  flags: GD_PART_APPEND -- the records are appended to what the
 * contig holds (parts must arrive in file order; coordinate order is checked across parts, and a reference that ended
 * in an earlier part must not resume), else they replace it; GD_PART_RELEASE -- the range is released afterwards.
 * expect_scale (>= 1; 0: unknown): how many times this part's share the whole reference is expected to be (its byte
 * range over this part's), so that the FIRST part allocates the contig's arrays once for all parts (they grow
 * geometrically when the estimate was short). */
enum { GD_PART_APPEND = 1, GD_PART_RELEASE = 2 };
int gd_ingest_decode_part(gd_ctx* ctx, int32_t tid, int32_t ref_id, const uint64_t* anchors, size_t n_anchors,
                          uint64_t end_anchor, unsigned flags, double expect_scale, uint64_t* n_records);
/* Where the device BAM read of this context has spent its wall clock so far, in seconds (measurement only):
 * out[0] reading into the staging buffers, [1] waiting for a staging buffer to leave for the device, [2] gd_ingest_begin
 * (allocations, member table), [3] decode: waiting for the inflate launches, [4] decode: the counting walk, [5] decode:
 * allocating the contig's arrays, [6] decode: the extraction (a thread per record over the walk's table).  Fills min(n, 7) values. */
int gd_ingest_timing(gd_ctx* ctx, double* out, size_t n);
int gd_ingest_abort(gd_ctx* ctx);
/* Page-locked host memory for the byte range handed to gd_ingest_bgzf (read the file
 * straight into it: the H2D copy then runs at PCIe speed instead of through a bounce
 * buffer).  Plain pageable memory works too, only slower. */
int gd_host_alloc(gd_ctx* ctx, size_t bytes, void** out);
int gd_host_free(gd_ctx* ctx, void* p);

/* Device-side views of the results (for RCCL gathers and zero-copy
 * consumers).  Pointers stay valid until the next gd_compute/gd_reset. */
int gd_device_perbase(gd_ctx* ctx, int32_t tid, const int32_t** dptr, int64_t* len);
int gd_device_windows(gd_ctx* ctx, const int64_t** d_sums, const int32_t** d_mins,
                      size_t* n_total);
/* Window offset of contig tid inside the concatenated window arrays. */
int gd_window_offset(gd_ctx* ctx, int32_t tid, size_t* off, size_t* n);
/* Ordered run boundaries {pos, cls | tid<<2} of all contigs. */
int gd_device_runs(gd_ctx* ctx, const int32_t** d_bounds, size_t* n_bounds);

/* Packed export block for a merge step (the reference's merge loop, depth/depth.go:394-421, run by
 * one rank for N workers): when device_buf is not NULL, every gd_compute also writes -- before its
 * single synchronisation, no extra launch latency -- the int64 words
 *     [n_bounds][sums: max_windows][mins: ceil(max_windows / 2) words of int32 pairs][bounds: cap_bounds]
 * into that caller-owned device buffer (e.g. an RCCL send buffer), so the exchange is one collective
 * on one buffer with no host round trip for sizes.  n_bounds is the true boundary count (it may
 * exceed cap_bounds: the receiver must check); the window arrays hold the computed contigs in
 * ascending tid order; a boundary is {int32 pos, int32 cls | index_among_computed_contigs << 2}.
 * GD_E_CAPACITY from gd_compute if the job has more windows than max_windows.  NULL switches it off. */
int gd_set_export(gd_ctx* ctx, void* device_buf, int64_t max_windows, int64_t cap_bounds);
/* Everything this context enqueues from now on waits for `hip_event` (a hipEvent_t the caller recorded on a
 * stream of its own, passed as void*): how a caller that hands the export block to a collective tells the
 * engine "the collective that was still reading this buffer is over" without blocking the host
 * (hipStreamWaitEvent on the context's stream). */
int gd_wait_event(gd_ctx* ctx, void* hip_event);

/* ---- the exchange step: export blocks to one root over RCCL (xGMI between the GPUs of a node) ----------------
 * One context per GPU, one rank per context -- in one process (a worker thread per device, the counterpart of the
 * reference's `-p` pool, depth/depth.go:392-394) or one process per GPU.  The root owns the BED files like the
 * reference's merge loop (depth/depth.go:394-421) and receives every rank's export block (gd_set_export: window sums,
 * minima, ordered class-run boundaries; the boundary count travels in word 0) with ONE grouped send / receive.
 *   gd_comm_unique_id   on one rank; the 128 bytes reach the others by whatever means the host has (a Go channel,
 *                       a file, MPI, torch.distributed)
 *   gd_comm_init        collective over the `world` contexts; librccl is opened here, on first use (dlopen): a
 *                       single-GPU run never loads it.  GD_E_NODEVICE: no RCCL on this machine
 *   gd_gather_export    after a gd_compute: `words` int64 words -- send == NULL: the export block (words == 0: all
 *                       of it) -- to `root`, which receives rank r's at recv + r * words (every rank passes the same
 *                       `words`).  Asynchronous: issued on the context's copy stream behind the compute that filled
 *                       the block, so it runs under the kernels of the next gd_compute; with two export buffers
 *                       alternating (gd_set_export before each compute) nothing is written into a buffer a gather
 *                       may still read -- the compute stream waits for the gather before the last one
 *   gd_gather_wait      the host waits until every gather issued so far has landed */
#define GD_COMM_ID_BYTES 128
int gd_comm_unique_id(void* id, size_t bytes);
/* Which RCCL the calls above use: its path into `out` (NUL terminated, cut to `cap`); *shared = 1 when the process had one
 * mapped already -- a host that imported torch has torch's own copy, and that one is used: never two RCCLs in one process --
 * 0 when the library opened librccl.so.1 by name.  GD_E_NODEVICE: none on this machine.  Local and cheap: every rank can
 * ask BEFORE any rank enters the collective gd_comm_init (a rank that cannot open RCCL would leave the others waiting there). */
int gd_comm_library(char* out, size_t cap, int* shared);
int gd_comm_init(gd_ctx* ctx, int rank, int world, const void* id, size_t bytes);
int gd_comm_destroy(gd_ctx* ctx);
int gd_gather_export(gd_ctx* ctx, const int64_t* send, int64_t* recv, size_t words, int root);
int gd_gather_wait(gd_ctx* ctx);
/* Device memory for a host that has no HIP binding of its own (cgo): the send / receive buffers of gd_gather_export, and
 * a synchronous read of them (waits for the context's streams first). */
int gd_device_alloc(gd_ctx* ctx, size_t bytes, void** out);
int gd_device_free(gd_ctx* ctx, void* p);
int gd_device_read(gd_ctx* ctx, void* dst_host, const void* src_device, size_t bytes);

/* ---- measurement ---------------------------------------------------------*/
int gd_get_stats(gd_ctx* ctx, gd_stats* out);
/* HIP-event timing of the kernels of the last gd_compute, recorded on the
 * stream the kernels ran on.  Enable before gd_compute. */
int gd_set_profiling(gd_ctx* ctx, int on);
int gd_kernel_ms(gd_ctx* ctx, int kernel_id, float* ms);
/* Host wall clock of the last gd_compute (or launch + finish pair), seconds[0 .. n): [0] contig table + device
 * allocations (what only a context's FIRST compute of a job size pays), [1] enqueueing the launches, [2] the wait
 * for the device incl. verification, re-runs and publishing, [3] their total.  What the reference's counterpart
 * spends per tile on fork/exec and pipes (depth/depth.go:392-394) is here one number per genome. */
int gd_compute_timing(gd_ctx* ctx, double* seconds, int n);

#ifdef __cplusplus
}
#endif
#endif /* GOLEFT_DEPTH_H */
