// mat_perm.h -- MatPerm::calc (reference src/MatrixPermanent.cpp:41-112): permanent of an n x n matrix by the Nijenhuis-Wilf form
// of Ryser's formula, subsets visited in Gray-code order:  perm(A) = (-1)^n 2 sum_{k=1}^{2^(n-1)} s_k prod_i x_i(k),
// x_i(1) = A(i, n-1) - row_sum_i / 2, x_i(k) = x_i(k-1) + z A(i, j) for the column j whose bit flips at step k, s_k = -s_(k-1).
// One matrix per wavefront, batched.
//
// mat_perm_kernel_t<NP, R> (n >= 10): the 2^(n-1) steps are cut into 64 R contiguous ranges of per = 2^(n-1) / (64 R) steps (a power
// of two), R per lane, walked side by side.  Because every range starts at a multiple of `per`, the bit that flips at the i-th step
// of a range is ctz(i) FOR EVERY RANGE -- uniform over the wavefront -- so the column A(., j) is a broadcast operand, and half of all
// steps flip bit 0, a quarter bit 1: those two columns stay in registers, the others are read from LDS (transposed, two rows per
// read) by one step in four.  x lives in registers (NP = n rounded up to a multiple of four; the padding rows hold x = 1 and a zero
// column: a multiplication by 1.0 is exact), the row loops are unrolled, the R ranges give the multiply chain of a step R-fold
// instruction-level parallelism.  Per subset the arithmetic is the reference's, operation for operation (z A(i, j) is exact, so the
// fused multiply-add equals its multiply + add; the product runs over i in ascending order).  What differs is the ORDER OF THE
// OUTER SUM: the reference adds the 2^(n-1) terms one after the other; here each (lane, range) adds its own terms in order, the R
// ranges of a lane are added in range order, and the 64 lanes by the wave reduction's butterfly (wave_sum) -- a few ulp of the sum
// for well-conditioned inputs; the reference's own known answers (test/MatrixPermanentTest.hpp:55-87, integers) come out exact.
// Round 3's kernel (x[24] indexed by a run-time n: in scratch memory; one range per lane; LDS read per row and step) reached 0.77
// TFLOP/s-equivalent at n = 20; it stays as the form for n < 10, where there are too few steps to cut.
#pragma once
#include "common.h"

// Small matrices (n < 10): one range per lane, run-time n.
__global__ __launch_bounds__(64) void mat_perm_kernel(const double *A, int n, int batch, double *out) {
  const int b = blockIdx.x;
  if (b >= batch) return;
  const int lane = threadIdx.x & 63;
  __shared__ double sA[24 * 24];
  const double *Ab = A + (size_t)b * n * n;
  for (int t = lane; t < n * n; t += 64) sA[t] = Ab[t];
  __syncthreads();
  // x_i(0) = A(i,n-1) - row_sum_i / 2; subset index k (1-based in the reference) has gray code g(k-1)
  const unsigned long long total = 1ull << (n - 1);  // number of subsets (k = 1 .. 2^(n-1))
  const unsigned long long per = (total + 63) / 64;
  const unsigned long long k0 = per * lane;          // 0-based subset index
  const unsigned long long k1 = (k0 + per < total) ? k0 + per : total;
  double x[24];
  double acc = 0.0;
  if (k0 < total) {
    const unsigned long long g0 = k0 ^ (k0 >> 1);
    for (int r = 0; r < n; r++) {
      double rs = 0;
      for (int c = 0; c < n; c++) rs += sA[r * n + c];
      double v = sA[r * n + (n - 1)] - 0.5 * rs;
      for (int c = 0; c < n - 1; c++) if ((g0 >> c) & 1ull) v += sA[r * n + c];
      x[r] = v;
    }
    unsigned long long g = g0;
    for (unsigned long long k = k0; k < k1; k++) {
      if (k != k0) {
        const int j = __builtin_ctzll(k);  // bit flipped between gray(k-1) and gray(k)
        const double z = ((g >> j) & 1ull) ? -1.0 : 1.0;
        g ^= (1ull << j);
        for (int r = 0; r < n; r++) x[r] += z * sA[r * n + j];
      }
      double prod = 1.0;
      for (int r = 0; r < n; r++) prod *= x[r];
      // sign: s = -1 for k=1 (0-based 0), alternating
      acc += ((k & 1ull) ? 1.0 : -1.0) * prod;
    }
  }
  acc = wave_sum(acc);
  if (lane == 0) {
    double ret = 2 * acc;
    if (n % 2 != 0) ret *= -1;
    out[b] = ret;
  }
}

template <int NP, int R>
__global__ __launch_bounds__(64) void mat_perm_kernel_t(const double *__restrict__ A, const int n, const int batch, double *__restrict__ out) {
  const int b = blockIdx.x;
  if (b >= batch) return;
  const int lane = threadIdx.x & 63;
  __shared__ __align__(16) double sAT[24 * NP];       // transposed and padded: sAT[j * NP + r] = A(r, j), 0 for r >= n
  const double *Ab = A + (size_t)b * n * n;
  for (int t = lane; t < 24 * NP; t += 64) {
    const int j = t / NP, r = t - j * NP;
    sAT[t] = (j < n && r < n) ? Ab[r * n + j] : 0.0;
  }
  __syncthreads();
  const unsigned per = (1u << (n - 1)) / (64u * R);   // steps per range (power of two, >= 4: n >= 8 + log2 R)
  double x[R][NP], c0[NP], c1[NP], acc[R];
  unsigned g[R];                                       // Gray code of the range's current subset
#pragma unroll
  for (int r = 0; r < NP; r++) { c0[r] = sAT[0 * NP + r]; c1[r] = sAT[1 * NP + r]; }
#pragma unroll
  for (int q = 0; q < R; q++) {
    const unsigned k0 = per * (unsigned)(lane * R + q);     // 0-based index of the range's first subset
    g[q] = k0 ^ (k0 >> 1);
#pragma unroll
    for (int r = 0; r < NP; r++) {
      double v = 1.0;
      if (r < n) {
        double rs = 0;
        for (int c = 0; c < n; c++) rs += sAT[c * NP + r];            // row sum in column order, as the reference
        v = sAT[(n - 1) * NP + r] - 0.5 * rs;
        for (int c = 0; c < n - 1; c++) if ((g[q] >> c) & 1u) v += sAT[c * NP + r];
      }
      x[q][r] = v;
    }
    double prod = 1.0;
#pragma unroll
    for (int r = 0; r < NP; r++) prod *= x[q][r];
    acc[q] = ((k0 & 1u) ? 1.0 : -1.0) * prod;                          // s = -1 at the first subset, alternating
  }
  // one step of every range: flip bit j (uniform), column `col` (uniform values), sign of the term sgn (uniform: the step's parity)
  auto step = [&](const int j, const double (&col)[NP], const double sgn) {
#pragma unroll
    for (int q = 0; q < R; q++) {
      const double z = ((g[q] >> j) & 1u) ? -1.0 : 1.0;
      g[q] ^= 1u << j;
      double prod = 1.0;
#pragma unroll
      for (int r = 0; r < NP; r++) {
        x[q][r] = __builtin_fma(z, col[r], x[q][r]);
        prod *= x[q][r];
      }
      acc[q] += sgn * prod;
    }
  };
  // steps i = 1 .. per - 1 of every range in blocks of four: bits 0, 1, 0, ctz(i) >= 2.  (k0 is a multiple of four: the term's sign
  // depends on i only.)
  for (unsigned i4 = 0; i4 < per; i4 += 4) {
    step(0, c0, 1.0);                                   // i = i4 + 1 (odd index: +)
    step(1, c1, -1.0);                                  // i = i4 + 2
    step(0, c0, 1.0);                                   // i = i4 + 3
    if (i4 + 4 < per) {
      const int j = __builtin_ctz(i4 + 4);
      double cj[NP];
#pragma unroll
      for (int r = 0; r < NP; r += 2) {
        const double2 v = *reinterpret_cast<const double2 *>(&sAT[j * NP + r]);
        cj[r] = v.x; cj[r + 1] = v.y;
      }
      step(j, cj, -1.0);                                // i = i4 + 4
    }
  }
  double tot = acc[0];
#pragma unroll
  for (int q = 1; q < R; q++) tot += acc[q];
  tot = wave_sum(tot);
  if (lane == 0) {
    double ret = 2 * tot;
    if (n % 2 != 0) ret *= -1;
    out[b] = ret;
  }
}
#ifndef MATPERM_RANGES
#define MATPERM_RANGES 2
#endif
// launch on the current device's default stream; returns hipGetLastError()
static inline hipError_t mat_perm_launch(const double *dA, int n, int batch, double *dO) {
  if (n < 10) mat_perm_kernel<<<batch, 64>>>(dA, n, batch, dO);
  else if (n <= 12) mat_perm_kernel_t<12, MATPERM_RANGES><<<batch, 64>>>(dA, n, batch, dO);
  else if (n <= 16) mat_perm_kernel_t<16, MATPERM_RANGES><<<batch, 64>>>(dA, n, batch, dO);
  else if (n <= 20) mat_perm_kernel_t<20, MATPERM_RANGES><<<batch, 64>>>(dA, n, batch, dO);
  else mat_perm_kernel_t<24, MATPERM_RANGES><<<batch, 64>>>(dA, n, batch, dO);
  return hipGetLastError();
}
