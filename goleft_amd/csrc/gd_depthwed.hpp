// gd_depthwed.hpp -- sites x samples matrix of `goleft depthwed` on the device.
//
// The reference reads N `*.depth.bed` files in lockstep and, per group of
// consecutive windows spanning >= SIZE bases (or up to the chromosome end),
// prints the SUM over the group's rows of int(0.5 + mean_text)
// (/root/reference/depthwed/depthwed.go:117-157, :103).  Here the N samples are
// N sets of contigs of one engine (their integer window sums are already in
// HBM), the text round trip is replaced by gd_depthwed_cell (gd_round4g.hpp)
// and one lane produces one (row, sample) cell.
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

// One workgroup per tile of 64 matrix rows x 64 samples.  The window sums are sample-major (a sample's windows
// are contiguous) and the matrix is row-major (a row's samples are contiguous): a thread per cell in matrix order
// read 32 bytes here and 32 bytes a whole sample further on (half of every line fetched for nothing, two 64-bit
// divisions per cell to find out where it was).  Here a wave takes 16 of the tile's samples, its lanes are the 64
// rows -- together they read one contiguous stretch of that sample's sums -- and the tile goes through LDS so that
// the stores are whole rows of 64 cells.
constexpr int WED_TILE = 64;

__global__ __launch_bounds__(256) void gd_depthwed_kernel(WedJob j)
{
    __shared__ int64_t s_tile[WED_TILE][WED_TILE + 1];
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int64_t row0 = (int64_t)blockIdx.x * WED_TILE;
    const int s0 = (int)blockIdx.y * WED_TILE;
    const int64_t row = row0 + lane;
    const bool in_rows = row < j.n_rows;
    int lo = 0;
    int64_t w0 = 0, w1 = 0, L = 0;
    if (in_rows) {
        int hi = j.n_ctg;                           // contig of this row
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (j.row_beg[mid] <= row) lo = mid; else hi = mid;
        }
        w0 = (row - j.row_beg[lo]) * j.group;
        w1 = w0 + j.group < j.nwin[lo] ? w0 + j.group : j.nwin[lo];
        L = j.clen[lo];
    }
    for (int k = 0; k < WED_TILE / 4; ++k) {
        const int sl = wv * (WED_TILE / 4) + k;     // sample of the tile
        const int s = s0 + sl;
        if (s >= j.n_samples) break;                // wave uniform
        int64_t cell = 0;
        if (in_rows) {
            const int64_t* ws = j.win_sum + j.off[(int64_t)s * j.n_ctg + lo];
            for (int64_t w = w0; w < w1; ++w) {
                const int64_t e = (w + 1) * j.W < L ? (w + 1) * j.W : L;
                cell += gd_depthwed_cell(ws[w], e - w * j.W);
            }
        }
        s_tile[lane][sl] = cell;
    }
    __syncthreads();
    const int s = s0 + lane;
    if (s < j.n_samples)
        for (int k = 0; k < WED_TILE / 4; ++k) {
            const int rl = wv * (WED_TILE / 4) + k;
            if (row0 + rl >= j.n_rows) break;
            j.cells[(row0 + rl) * j.n_samples + s] = s_tile[rl][lane];
        }
}

}  // namespace gd
