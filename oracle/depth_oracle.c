/*
 * depth_oracle.c -- CPU restatement of the `goleft depth` hot path.
 * TEST INFRASTRUCTURE ONLY; see depth_oracle.h for the parity status
 * ("parity unpinned": the reference's arithmetic is an external samtools).
 *
 * Every function cites the reference lines it follows
 * (paths relative to /root/reference).
 */
#include "depth_oracle.h"

#include <stdlib.h>
#include <string.h>

static long lmin(long a, long b) { return a < b ? a : b; }
static long lmax(long a, long b) { return a > b ? a : b; }

/* ---- samtools depth counting semantics (external; call site depth/depth.go:45,
 * args bound at :116-117 and :152-153: `-Q q -d maxmean+2500 -r region`).
 * -d is accepted and ignored by samtools >= 1.13; this restatement is uncapped. */

static int op_consumes_ref(unsigned op)
{
    /* M=0 I=1 D=2 N=3 S=4 H=5 P=6 '='=7 X=8 */
    return op == 0 || op == 2 || op == 3 || op == 7 || op == 8;
}

static int op_is_counted(unsigned op) { return op == 0 || op == 7 || op == 8; }

void gdo_perbase(const gdo_reads* r, int q, uint32_t flag_mask,
                 int64_t start, int64_t end, int32_t* out)
{
    if (end <= start) return;
    memset(out, 0, (size_t)(end - start) * sizeof(int32_t));
    for (size_t i = 0; i < r->n; i++) {
        if (r->flag[i] & flag_mask) continue;
        if ((int)r->mapq[i] < q) continue;
        int64_t cur = r->pos[i];
        if (cur >= end) continue; /* sorted input could break; keep brute force */
        for (uint32_t k = r->cigar_off[i]; k < r->cigar_off[i + 1]; k++) {
            unsigned op = r->cigar[k] & 0xf;
            int64_t len = r->cigar[k] >> 4;
            if (op_is_counted(op)) {
                for (int64_t p = cur; p < cur + len; p++)
                    if (p >= start && p < end) out[p - start]++;
            }
            if (op_consumes_ref(op)) cur += len;
        }
    }
}

void gdo_perbase_diff(const gdo_reads* r, int q, uint32_t flag_mask,
                      int64_t start, int64_t end, int32_t* out)
{
    if (end <= start) return;
    size_t L = (size_t)(end - start);
    memset(out, 0, L * sizeof(int32_t));
    for (size_t i = 0; i < r->n; i++) {
        if (r->flag[i] & flag_mask) continue;
        if ((int)r->mapq[i] < q) continue;
        int64_t cur = r->pos[i];
        if (cur >= end) break; /* coordinate sorted */
        for (uint32_t k = r->cigar_off[i]; k < r->cigar_off[i + 1]; k++) {
            unsigned op = r->cigar[k] & 0xf;
            int64_t len = r->cigar[k] >> 4;
            if (op_is_counted(op) && len > 0) {
                int64_t s = cur < start ? start : cur;
                int64_t e = cur + len > end ? end : cur + len;
                if (s < e) {
                    out[s - start] += 1;
                    if ((size_t)(e - start) < L) out[e - start] -= 1;
                }
            }
            if (op_consumes_ref(op)) cur += len;
        }
    }
    int32_t acc = 0;
    for (size_t p = 0; p < L; p++) {
        acc += out[p];
        out[p] = acc;
    }
}

/* ---- depth/depth.go:223-234 getCovClass */
int gdo_cov_class(int depth, int mincov, int maxmeandepth)
{
    if (depth == 0) return 0;
    if (depth < mincov) return 1;
    if (maxmeandepth > 0 && depth >= maxmeandepth) return 3;
    return 2;
}

const char* gdo_cov_class_name(int cls)
{
    switch (cls) {
    case 0: return "NO_COVERAGE";
    case 1: return "LOW_COVERAGE";
    case 2: return "CALLABLE";
    case 3: return "EXCESSIVE_COVERAGE";
    }
    return "";
}

/* ---- depth/depth.go:73 regexp "(.+?)[:\t](\d+)([\-\t])(\d+).*?" and
 * :75-94 chromStartEndFromLine.  Leftmost match, lazy chrom: the earliest
 * separator position i >= 1 at which `[:\t] digits [-\t] digit` matches. */
static int is_digit(char c) { return c >= '0' && c <= '9'; }

int gdo_chrom_start_end(const char* line, size_t len, char* chrom, size_t cap,
                        long* start, long* end)
{
    for (size_t i = 1; i < len; i++) {
        if (line[i] != ':' && line[i] != '\t') continue;
        size_t j = i + 1;
        while (j < len && is_digit(line[j])) j++;
        if (j == i + 1) continue;         /* (\d+) needs one digit */
        /* \d+ is greedy but may backtrack; the separator is not a digit, so
         * only the full digit run can be followed by it. */
        if (j >= len || (line[j] != '-' && line[j] != '\t')) continue;
        size_t k = j + 1;
        while (k < len && is_digit(line[k])) k++;
        if (k == j + 1) continue;
        if (i + 1 > cap) return -1;
        memcpy(chrom, line, i);
        chrom[i] = 0;
        char buf[32];
        size_t n1 = j - (i + 1);
        size_t n2 = k - (j + 1);
        if (n1 >= sizeof buf || n2 >= sizeof buf) return -1;
        memcpy(buf, line + i + 1, n1); buf[n1] = 0;
        long istart = strtol(buf, NULL, 10);
        if (line[j] == '-') istart--;       /* :86-88 chr:s-e is 1-based */
        memcpy(buf, line + j + 1, n2); buf[n2] = 0;
        long iend = strtol(buf, NULL, 10);
        *start = lmax(istart, 0);           /* :93 */
        *end = iend;
        return 0;
    }
    return -1;                              /* :77-79 log.Fatal */
}

/* ---- depth/depth.go:48,:132 step; :150-154 tile loop */
long gdo_step(int windowsize)
{
    long step = 10000000;
    return lmax(1, step / windowsize) * windowsize;
}

size_t gdo_tiles(long length, int windowsize, long* starts, long* ends, size_t cap)
{
    long step = gdo_step(windowsize);
    size_t n = 0;
    for (long i = 0; i < length; i += step) {
        /* region "%s:%d-%d", i+1, min(i+step,length)  == 0-based [i, min(..)) */
        if (n < cap) {
            starts[n] = i;
            ends[n] = lmin(i + step, length);
        }
        n++;
    }
    return n;
}

/* ---- depth/depth.go:181-189 mean */
typedef struct {
    double sum; /* float64 accumulation of ints, :186 */
    long   len; /* len(depthCache) */
} dcache;

static double mean_of(const dcache* c, long l)
{
    if (c->len == 0 || l == 0) return 0;
    return c->sum / (double)l;
}

/* ---- depth/depth.go:238-364 callback */
void gdo_callback(const char* chrom, long regionStart, long regionEnd,
                  const int32_t* depthv, int W, int mincov, int maxmean,
                  FILE* fhHD, FILE* fhCA)
{
    dcache depthCache = {0.0, 0};
    long depth = 0, pos = 0;                                   /* :255 */
    long lastWindow = lmax(0, regionStart / W);                /* :263 */
    long cache0 = regionStart - 1, cache1 = regionStart - 1;   /* :264-266 */
    int lastCovClass = -1;                                     /* "" */

    /* :282-325 one iteration per line samtools prints: covered positions only */
    for (long p = regionStart; p < regionEnd; p++) {
        if (depthv[p - regionStart] <= 0) continue;
        pos = p;                                               /* :287 (pos--) */
        depth = depthv[p - regionStart];
        if (pos / W != lastWindow) {                           /* :293 */
            long thisWindow = pos / W;
            for (long iw = lastWindow; iw < thisWindow; iw++) {
                long s = lmax(regionStart, iw * W);
                long e = lmin(regionEnd, (iw + 1) * W);
                fprintf(fhHD, "%s\t%ld\t%ld\t%.4g\n", chrom, s, e,
                        mean_of(&depthCache, e - s));          /* :301 */
                depthCache.sum = 0; depthCache.len = 0;
            }
            lastWindow = thisWindow;
        }
        depthCache.sum += (double)depth; depthCache.len++;     /* :306 */
        int covClass = gdo_cov_class((int)depth, mincov, maxmean);
        if (covClass != lastCovClass || pos != cache1 + 1) {   /* :310 */
            if (lastCovClass != -1)
                fprintf(fhCA, "%s\t%ld\t%ld\t%s\n", chrom, cache0, cache1 + 1,
                        gdo_cov_class_name(lastCovClass));
            if (pos != cache1 + 1)                             /* :315 */
                fprintf(fhCA, "%s\t%ld\t%ld\t%s\n", chrom, cache1 + 1, pos,
                        "NO_COVERAGE");
            lastCovClass = covClass;
            cache0 = pos; cache1 = pos;
        } else {
            cache1 = pos;
        }
    }
    if (cache0 != -1 && lastCovClass != -1)                    /* :326 */
        fprintf(fhCA, "%s\t%ld\t%ld\t%s\n", chrom, cache0, cache1 + 1,
                gdo_cov_class_name(lastCovClass));
    if (depthCache.len > 0) {                                  /* :329 */
        long s = pos / W * W;
        if (s < regionEnd) {
            long s2 = lmax(s, regionStart);
            long e = lmin(regionEnd, s2 + W);
            fprintf(fhHD, "%s\t%ld\t%ld\t%.4g\n", chrom, s2, e,
                    mean_of(&depthCache, e - s2));
            depthCache.sum = 0; depthCache.len = 0;
            pos = e;                                           /* :338 */
        }
    }
    if (cache1 + 1 < regionEnd) {                              /* :343 */
        if (cache1 != -1)
            fprintf(fhCA, "%s\t%ld\t%ld\tNO_COVERAGE\n", chrom, cache1 + 1, regionEnd);
        else
            fprintf(fhCA, "%s\t%ld\t%ld\tNO_COVERAGE\n", chrom, regionStart, regionEnd);
        for (long ds = lmax(regionStart, pos) / W * W;
             ds < regionEnd && pos < regionEnd; ds += W) {     /* :351 */
            long de = lmin(regionEnd, ds + W);
            long s = lmax(ds, regionStart);
            fprintf(fhHD, "%s\t%ld\t%ld\t%.4g\n", chrom, s, de,
                    mean_of(&depthCache, de - s));
            depthCache.sum = 0; depthCache.len = 0;
        }
    }
}

/* Convenience for ctypes callers: append one region's rows to two paths. */
int gdo_callback_append(const char* chrom, long regionStart, long regionEnd,
                        const int32_t* depthv, int W, int mincov, int maxmean,
                        const char* depth_path, const char* callable_path)
{
    FILE* hd = fopen(depth_path, "a");
    if (!hd) return -1;
    FILE* ca = fopen(callable_path, "a");
    if (!ca) { fclose(hd); return -1; }
    gdo_callback(chrom, regionStart, regionEnd, depthv, W, mincov, maxmean, hd, ca);
    fclose(hd);
    fclose(ca);
    return 0;
}
