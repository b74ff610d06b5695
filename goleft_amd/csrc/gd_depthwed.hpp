// gd_depthwed.hpp -- sites x samples matrix of `goleft depthwed` on the device.
//
// The reference reads N `*.depth.bed` files in lockstep and, per group of
// consecutive windows spanning >= SIZE bases (or up to the chromosome end),
// prints the SUM over the group's rows of int(0.5 + mean_text)
// (/root/reference/depthwed/depthwed.go:117-157, :103).  Here the N samples are
// N sets of contigs of one engine (their integer window sums are already in
// HBM), the text round trip is replaced by gd_depthwed_cell (gd_round4g.hpp)
// and one thread produces one (row, sample) cell.
#pragma once

#include "gd_round4g.hpp"

namespace gd {

struct WedJob {
    const int64_t* win_sum;     // concatenated window sums of the last gd_compute
    const int64_t* off;         // [n_samples * n_ctg] window offset of (sample, contig)
    const int64_t* nwin;        // [n_ctg] windows per contig
    const int64_t* clen;        // [n_ctg] contig length
    const int64_t* row_beg;     // [n_ctg + 1] first matrix row of each contig
    int64_t* cells;             // [n_rows * n_samples], row major
    int32_t n_samples, n_ctg;
    int64_t n_rows;
    int64_t W, group;           // windows per group = ceil(size / W)
};

__global__ __launch_bounds__(256) void gd_depthwed_kernel(WedJob j)
{
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= j.n_rows * j.n_samples) return;
    const int64_t row = gid / j.n_samples;
    const int s = (int)(gid - row * j.n_samples);
    int lo = 0, hi = j.n_ctg;                       // contig of this row
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (j.row_beg[mid] <= row) lo = mid; else hi = mid;
    }
    const int64_t r = row - j.row_beg[lo];
    const int64_t w0 = r * j.group;
    const int64_t w1 = w0 + j.group < j.nwin[lo] ? w0 + j.group : j.nwin[lo];
    const int64_t* ws = j.win_sum + j.off[(int64_t)s * j.n_ctg + lo];
    const int64_t L = j.clen[lo];
    int64_t cell = 0;
    for (int64_t w = w0; w < w1; ++w) {
        const int64_t e = (w + 1) * j.W < L ? (w + 1) * j.W : L;
        cell += gd_depthwed_cell(ws[w], e - w * j.W);
    }
    j.cells[gid] = cell;
}

}  // namespace gd
