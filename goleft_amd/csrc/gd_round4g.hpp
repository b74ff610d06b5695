// gd_round4g.hpp -- the integer a `depthwed` cell receives from one depth.bed row,
// computed from the window's integer sum without formatting any text.
//
// Reference chain (paths relative to /root/reference):
//   goleft depth prints   mean = float64(sum) / float64(len)   with "%.4g"
//                         (depth/depth.go:181-189 mean, :301 / :335 / :356 Fprintf);
//   depthwed parses that text back and adds   int(0.5 + mean_text)
//                         (depthwed/depthwed.go:96 ParseFloat, :103).
// Both conversions are correctly rounded (Go strconv), so the cell is a pure
// function of q = fl(sum/len):
//   D    = q rounded to 4 significant decimal digits, ties to even ON THE EXACT
//          BINARY VALUE of q (what %.4g prints, fixed or exponent form alike);
//   cell = floor(D + 0.5)   -- D has at most 4 significant digits, so D + 0.5 is
//          never within rounding distance of an integer unless D = n + 0.5 exactly,
//          and the double sum 0.5 + fl(D) truncates to the same integer.
// Everything below is exact: two-product / FMA remainders decide the decimal
// rounding, no table of decimal strings, no snprintf.  Compiles for host and
// device (the host instance is the unit-tested one, tests/test_depthwed.py).
#pragma once

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define GD_HD __host__ __device__ inline
#else
#define GD_HD inline
#endif

// q (>= 1000) rounded to 4 significant digits; exact integer result.
GD_HD int64_t gd_round4g_big(double q)
{
    // largest k with 10^(k+3) <= q; powers of ten up to 1e22 are exact doubles
    double P = 1.0;                      // 10^k
    while (q >= P * 1e4) P *= 10.0;      // P * 1e4 is exact while P <= 1e18
    double d0 = rint(q / P);             // within +-1 of the answer
    const double r = fma(-d0, P, q);     // exact: d0*P has < 53 bits, |r| <= P
    const double half = P * 0.5;
    const bool odd = fmod(d0, 2.0) != 0.0;
    if (r > half || (r == half && odd)) d0 += 1.0;
    else if (r < -half || (r == -half && odd)) d0 -= 1.0;
    return (int64_t)d0 * (int64_t)P;
}

// The depthwed cell of one window: int(0.5 + parse(fmt("%.4g", sum/len))).
GD_HD int64_t gd_depthwed_cell(int64_t sum, int64_t len)
{
    if (sum <= 0 || len <= 0) return 0;              // mean() returns 0 for an empty window
    const double q = (double)sum / (double)len;      // depth/depth.go:188 (sum < 2^53: exact operands)
    if (q < 0.1) return 0;                           // D <= 0.1000 -> 0 (0.1 here is the double just above 1/10)
    if (q >= 1000.0) return gd_round4g_big(q);
    // 0.1 <= q < 1000: D = d / 10^j with d = q * 10^j rounded half-even, j in 1..4
    int j;
    double P;
    if (q >= 100.0) { j = 1; P = 10.0; }
    else if (q >= 10.0) { j = 2; P = 100.0; }
    else if (q >= 1.0) { j = 3; P = 1000.0; }
    else { j = 4; P = 10000.0; }
    (void)j;
    const double hi = q * P;
    const double lo = fma(q, P, -hi);                // q * P == hi + lo exactly
    double d0 = rint(hi);                            // ties of hi to even
    const double r = hi - d0;                        // exact, |r| <= 0.5
    if (r == 0.5) { if (lo > 0.0) d0 += 1.0; }       // true value just above the tie
    else if (r == -0.5) { if (lo < 0.0) d0 -= 1.0; } // just below
    // (|r| < 0.5: |lo| <= ulp(hi)/2 cannot reach the tie; r == +-0.5 with lo == 0 is an
    //  exact tie and rint already chose the even neighbour)
    const int64_t d = (int64_t)d0;                   // 1000 .. 10000
    const int64_t Pi = (int64_t)P;
    return (d + Pi / 2) / Pi;                        // floor(D + 0.5), D = d / 10^j
}
