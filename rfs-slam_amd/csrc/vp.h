// vp.h -- Victoria Park model on the device: 3-D landmarks (x, y, trunk diameter), measurements
// (range, bearing, diameter).  Reference: src/MeasurementModel_VictoriaPark.cpp:75-291 (measure, inverseMeasure,
// probabilityOfDetection(2), setLaserScan), include/KalmanFilter_VictoriaPark.hpp:56-73 (innovation: wrap FIRST, then
// gates), 3x3 algebra in Eigen's closed forms (cofactor inverse, brute-force determinant).
//
// HBM layout: slab[particle][11 planes][cap]: W, WP, MX, MY, MD, SXX, SXY, SXD, SYY, SYD, SDD (packed symmetric 3x3).
// Kernels keep the structure of the 2-D ones (one wavefront per particle): vp_update_map (gate bitmask -> dense survivor
// list -> emit), vp_weighting (rank sort, eval points, intensity products, L table, shared partition code), vp_merge
// (sequential-greedy, mixture staged in LDS) with optional fused prune.  Victoria Park maps are small (tens of
// Gaussians, ~10 measurements), so these favour clarity over the last bit of speed.
#pragma once
#include "common.h"
#include "weighting.h"

enum Plane3 { P3_W = 0, P3_WP, P3_MX, P3_MY, P3_MD, P3_SXX, P3_SXY, P3_SXD, P3_SYY, P3_SYD, P3_SDD, P3_COUNT };

__device__ __forceinline__ double *plane3(double *slab, int cap, int particle, int pl) {
  return slab + ((size_t)particle * P3_COUNT + pl) * (size_t)cap;
}

struct Ent3 {
  double w, x, y, d;
  double xx, xy, xd, yy, yd, dd;
};
__device__ __forceinline__ void load_ent3(const double *slab, int cap, int i, int m, Ent3 &e, bool withW) {
  const double *b = slab + (size_t)i * P3_COUNT * cap;
  if (withW) e.w = b[(size_t)P3_W * cap + m];
  e.x = b[(size_t)P3_MX * cap + m]; e.y = b[(size_t)P3_MY * cap + m]; e.d = b[(size_t)P3_MD * cap + m];
  e.xx = b[(size_t)P3_SXX * cap + m]; e.xy = b[(size_t)P3_SXY * cap + m]; e.xd = b[(size_t)P3_SXD * cap + m];
  e.yy = b[(size_t)P3_SYY * cap + m]; e.yd = b[(size_t)P3_SYD * cap + m]; e.dd = b[(size_t)P3_SDD * cap + m];
}
__device__ __forceinline__ void full3(const Ent3 &e, double S[9]) {
  S[0] = e.xx; S[1] = e.xy; S[2] = e.xd;
  S[3] = e.xy; S[4] = e.yy; S[5] = e.yd;
  S[6] = e.xd; S[7] = e.yd; S[8] = e.dd;
}

// Eigen determinant_impl<.,3> (bruteforce_det3_helper) and compute_inverse<.,3> (cofactors), row-major m[9].
__device__ __forceinline__ double det3(const double *m) {
  const double a = m[0] * (m[4] * m[8] - m[5] * m[7]);
  const double b = m[1] * (m[3] * m[8] - m[5] * m[6]);
  const double c = m[2] * (m[3] * m[7] - m[4] * m[6]);
  return a - b + c;
}
__device__ __forceinline__ double cof3(const double *m, int i, int j) {
  const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
  return m[3 * i1 + j1] * m[3 * i2 + j2] - m[3 * i1 + j2] * m[3 * i2 + j1];
}
__device__ __forceinline__ void inv3(const double *m, double *r) {
  const double c0 = cof3(m, 0, 0), c1 = cof3(m, 1, 0), c2 = cof3(m, 2, 0);
  const double d = (c0 * m[0] + c1 * m[3]) + c2 * m[6];
  const double invdet = 1.0 / d;
  r[0] = c0 * invdet; r[1] = c1 * invdet; r[2] = c2 * invdet;
  r[3] = cof3(m, 0, 1) * invdet; r[4] = cof3(m, 1, 1) * invdet; r[5] = cof3(m, 2, 1) * invdet;
  r[6] = cof3(m, 0, 2) * invdet; r[7] = cof3(m, 1, 2) * invdet; r[8] = cof3(m, 2, 2) * invdet;
}
// (e^T * Sinv) * e, RandomVec::mahalanobisDist2 order
__device__ __forceinline__ double md2_3(const double *Si, double e0, double e1, double e2) {
  const double t0 = (e0 * Si[0] + e1 * Si[3]) + e2 * Si[6];
  const double t1 = (e0 * Si[1] + e1 * Si[4]) + e2 * Si[7];
  const double t2 = (e0 * Si[2] + e1 * Si[5]) + e2 * Si[8];
  return (t0 * e0 + t1 * e1) + t2 * e2;
}

// measure(): :104-151 -- pose mean only (covariance dropped), heading - pi/2, 2-D model on (x,y), diameter passthrough,
// S = blockdiag(S2, Sdd + R33 + r^2 Slb), H = blockdiag(H2, 1); always "true".
struct VPMeas {
  double z0, z1, z2;
  double h00, h01, h10, h11;
  double S[9];
};
__device__ __forceinline__ void vp_measure(const Params &P, double px, double py, double pth, double mx, double my, double md, double sxx, double sxy,
                                           double syy, double sdd, VPMeas &o) {
  PoseReg tp;
  tp.x = px; tp.y = py; tp.th = pth - RFS_PI / 2;
#pragma unroll
  for (int k = 0; k < 9; k++) tp.P[k] = 0.0;
  MeasOut mo;
  rb_measure(P, tp, mx, my, sxx, sxy, syy, mo);  // P.R holds the 2x2 range-bearing block of R
  o.z0 = mo.z0; o.z1 = mo.z1; o.z2 = md;
  o.h00 = mo.h00; o.h01 = mo.h01; o.h10 = mo.h10; o.h11 = mo.h11;
  o.S[0] = mo.s00; o.S[1] = mo.s01; o.S[2] = 0.0;
  o.S[3] = mo.s10; o.S[4] = mo.s11; o.S[5] = 0.0;
  o.S[6] = 0.0; o.S[7] = 0.0; o.S[8] = sdd + P.R9[8] + (mo.z0 * mo.z0) * P.Slb;
}

// probabilityOfDetection2(): :202-265.  scan: nScan beams (half-degree steps); beams past the scan count as visible.
__device__ double vp_pd2(const Params &P, const double *scan, int nScan, double px, double py, double pth, double mx, double my, double md,
                         bool &close) {
  close = false;
  VPMeas o;
  vp_measure(P, px, py, pth, mx, my, md, 0.0, 0.0, 0.0, 0.0, o);  // only z is used
  const double dist = o.z0, angle = o.z1;
  if (angle > P.bmax || angle < P.bmin || dist < P.rmin || dist > P.rmax) return 0.0;
  const double rad = o.z2 / 2;
  const double gamma = atan(rad / o.z0);
  const int maxNumPoints = (int)floor(2 * gamma * 720.0 / (2 * RFS_PI));
  if (P.nPd > maxNumPoints && maxNumPoints >= 0 && P.PdTable[maxNumPoints] == 0) return 0.0;
  if (P.nPd > maxNumPoints && maxNumPoints >= 0 && P.PdTable[maxNumPoints] < P.bufferPd) close = true;
  int minb = (int)ceil((angle - gamma) * 720.0 / (2 * RFS_PI));
  int maxb = minb + maxNumPoints;
  while (minb >= 720) minb -= 720;
  while (minb < 0) minb += 720;
  while (maxb >= 720) maxb -= 720;
  while (maxb < 0) maxb += 720;
  int numPoints = 0;
  const double minrange = dist - rad - 6 * 0.03;
  if ((maxb - minb + 720) % 720 > 0) {
    for (int b = minb; b != maxb; b = (b + 1) % 720) {
      const double s = (b < nScan) ? scan[b] : 0.0;
      if (s > minrange || s == 0) numPoints++;
    }
  }
  if (numPoints >= P.nPd) numPoints = P.nPd - 1;
  if (P.PdTable[numPoints] == 0) close = false;
  return P.PdTable[numPoints];
}
// probabilityOfDetection(): :153-199 -- the maximum of probabilityOfDetection2 over the landmark and its copies shifted
// sideways by i * 2 * diameter, i = 1, 2, ... while (i - 1) * 2 * diameter < max(3 sigma_perp, 0.2); `angle` formed as the
// reference writes it; closeToLimit = (min == 0 && max > 0), else what the last call (the unshifted landmark) left.
// Evaluated for the 64 landmarks of a wave at once.  The number of shifted copies differs wildly between landmarks (a fresh,
// thin, far landmark: 3 sigma of lateral uncertainty over twice its diameter -- dozens; an established one: one), so a lane
// looping over its own copies leaves the wave waiting for its worst landmark.  Here the (landmark, copy) evaluations of all
// lanes are laid end to end and dealt out to the lanes; minimum and maximum (exact in any order: the values come from the
// Pd table) are folded with LDS atomics on the bit patterns (the values are >= 0).  `ws`: VP_PD_SCRATCH_BYTES of LDS per wave.
#define VP_PD_SCRATCH_BYTES (64 * (5 * 8 + 2 * 8) + 65 * 4 + 4)
__device__ double vp_pd_wave(const Params &P, const double *scan, int nScan, double px, double py, double pth, const Ent3 &e, bool act, bool &close,
                             unsigned char *ws) {
  const int lane = threadIdx.x & 63;
  double *lx = reinterpret_cast<double *>(ws), *ly = lx + 64, *ld = ly + 64, *lp0 = ld + 64, *lp1 = lp0 + 64;
  unsigned long long *lmn = reinterpret_cast<unsigned long long *>(lp1 + 64), *lmx = lmn + 64;
  int *pre = reinterpret_cast<int *>(lmx + 64);
  VPMeas o;
  vp_measure(P, px, py, pth, e.x, e.y, e.d, 0.0, 0.0, 0.0, 0.0, o);
  const double angle = atan2(o.z1, o.z0) + pth;  // sic (:165-166)
  const double p0 = -sin(angle), p1 = cos(angle);
  const double r0 = p0 * e.xx + p1 * e.xy, r1 = p0 * e.xy + p1 * e.yy;
  double sd = r0 * p0 + r1 * p1;
  sd = 3 * sqrt(sd);
  sd = (sd < 0.2) ? 0.2 : sd;   // std::max(sd, 0.2) as the reference has it: a NaN (indefinite covariance) stays NaN -> no shifted copies
  // n = how many i = 1, 2, ... satisfy (i - 1) * (2 d) < sd, the loop condition evaluated as the reference writes it
  int n = 0;
  if (act) {
    const double step = 2 * e.d;
    if (!(sd == sd)) {
      n = 0;                                               // NaN: the loop condition is false from the start
    } else if (!(step > 0)) {
      n = ((0.0 * step) < sd) ? 100000 : 0;               // non-positive diameter: the reference never terminates
    } else {
      double g = floor(sd / step);
      if (g > 100000.0) g = 100000.0;
      n = (int)g;                                          // near ceil(sd / step): settle it with the exact condition
      while (n > 0 && !((n - 1) * step < sd)) n--;
      while (n < 100000 && (n * step < sd)) n++;
    }
  }
  lx[lane] = e.x; ly[lane] = e.y; ld[lane] = e.d; lp0[lane] = p0; lp1[lane] = p1;
  lmn[lane] = 0x7ff0000000000000ull;                       // +inf
  lmx[lane] = 0ull;                                        // pd >= 0
  const int cnt = 2 * n;
  const int off = wave_excl_scan(cnt, lane);
  const int total = __builtin_amdgcn_readlane(off + cnt, 63);
  pre[lane] = off;
  if (lane == 63) pre[64] = total;
  wave_sync();
  for (int t = lane; t < total; t += 64) {
    int l = 0;                                             // owner: the last landmark whose block starts at or before t
#pragma unroll
    for (int st = 32; st >= 1; st >>= 1) l += (pre[l + st] <= t) ? st : 0;
    const int j = t - pre[l];
    const int i = (j >> 1) + 1;
    const double dd = ld[l];
    const double sh = i * 2 * dd;
    const double sg = (j & 1) ? -1.0 : 1.0;
    bool c2;
    const double p = (j & 1) ? vp_pd2(P, scan, nScan, px, py, pth, lx[l] - sh * lp0[l], ly[l] - sh * lp1[l], dd, c2)
                             : vp_pd2(P, scan, nScan, px, py, pth, lx[l] + sh * lp0[l], ly[l] + sh * lp1[l], dd, c2);
    (void)sg;
    const unsigned long long bits = (unsigned long long)__double_as_longlong(p);
    atomicMin(&lmn[l], bits);
    atomicMax(&lmx[l], bits);
  }
  wave_sync();
  double mn = __longlong_as_double((long long)lmn[lane]), mx = __longlong_as_double((long long)lmx[lane]);
  const double p = vp_pd2(P, scan, nScan, px, py, pth, e.x, e.y, e.d, close);   // the unshifted landmark last: `close` is its verdict
  mn = fmin(mn, p); mx = fmax(mx, p);
  if (mn == 0 && mx > 0) close = true;
  wave_sync();                                             // (the scratch is reused by the next pass)
  return mx;
}

// Probe for tests (rfsgpu_vp_probe_pd): Pd and the near-limit flag of every Gaussian of one particle, as the kernels see them.
__global__ __launch_bounds__(64) void vp_probe_pd_kernel(Buffers B, Params P, int cur, int slot, double *pdOut, int *closeOut) {
  __shared__ __align__(16) unsigned char ws[(VP_PD_SCRATCH_BYTES + 15) & ~15];
  const int lane = threadIdx.x;
  const int n = B.count[slot];
  const double px = B.pose[3 * slot], py = B.pose[3 * slot + 1], pth = B.pose[3 * slot + 2];
  for (int c0 = 0; c0 < n; c0 += 64) {
    const int m = c0 + lane;
    Ent3 e;
    e.w = 0; e.x = 10; e.y = 10; e.d = 1; e.xx = 1; e.xy = 0; e.xd = 0; e.yy = 1; e.yd = 0; e.dd = 1;
    if (m < n) load_ent3(B.slab[cur], B.cap, slot, m, e, false);
    bool close = false;
    const double pd = vp_pd_wave(P, B.scan, B.nScan, px, py, pth, e, m < n, close, ws);
    if (m < n) { pdOut[m] = pd; closeOut[m] = close ? 1 : 0; }
  }
}

// Landmark-level quantities of KalmanFilter::correct for the 3-D model.
struct LmKF3 {
  double zx0, zx1, zx2;
  double Si[9];
  double factor;   // sqrt((2pi)^3 |S|)
  double K[9];
  double p[6];     // updated covariance (symmetrised), packed xx, xy, xd, yy, yd, dd
};
__device__ void lm_precompute3(const Params &P, double px, double py, double pth, const Ent3 &e, LmKF3 &k) {
  VPMeas o;
  vp_measure(P, px, py, pth, e.x, e.y, e.d, e.xx, e.xy, e.yy, e.dd, o);
  k.zx0 = o.z0; k.zx1 = o.z1; k.zx2 = o.z2;
  inv3(o.S, k.Si);
  k.factor = sqrt(P.twoPiPowD * det3(o.S));
  double Pm[9];
  full3(e, Pm);
  // T = P * H^T, H = blockdiag(H2, 1): H^T = [h00 h10 0; h01 h11 0; 0 0 1]
  double T[9];
#pragma unroll
  for (int r = 0; r < 3; r++) {
    T[3 * r + 0] = (Pm[3 * r + 0] * o.h00 + Pm[3 * r + 1] * o.h01) + Pm[3 * r + 2] * 0.0;
    T[3 * r + 1] = (Pm[3 * r + 0] * o.h10 + Pm[3 * r + 1] * o.h11) + Pm[3 * r + 2] * 0.0;
    T[3 * r + 2] = (Pm[3 * r + 0] * 0.0 + Pm[3 * r + 1] * 0.0) + Pm[3 * r + 2] * 1.0;
  }
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int c = 0; c < 3; c++) k.K[3 * r + c] = (T[3 * r + 0] * k.Si[c] + T[3 * r + 1] * k.Si[3 + c]) + T[3 * r + 2] * k.Si[6 + c];
  // (I - K H) P
  double KH[9];
#pragma unroll
  for (int r = 0; r < 3; r++) {
    KH[3 * r + 0] = (k.K[3 * r + 0] * o.h00 + k.K[3 * r + 1] * o.h10) + k.K[3 * r + 2] * 0.0;
    KH[3 * r + 1] = (k.K[3 * r + 0] * o.h01 + k.K[3 * r + 1] * o.h11) + k.K[3 * r + 2] * 0.0;
    KH[3 * r + 2] = (k.K[3 * r + 0] * 0.0 + k.K[3 * r + 1] * 0.0) + k.K[3 * r + 2] * 1.0;
  }
  double A[9], Q[9];
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int c = 0; c < 3; c++) A[3 * r + c] = ((r == c) ? 1.0 : 0.0) - KH[3 * r + c];
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int c = 0; c < 3; c++) Q[3 * r + c] = (A[3 * r + 0] * Pm[c] + A[3 * r + 1] * Pm[3 + c]) + A[3 * r + 2] * Pm[6 + c];
  k.p[0] = (Q[0] + Q[0]) / 2; k.p[1] = (Q[1] + Q[3]) / 2; k.p[2] = (Q[2] + Q[6]) / 2;
  k.p[3] = (Q[4] + Q[4]) / 2; k.p[4] = (Q[5] + Q[7]) / 2; k.p[5] = (Q[8] + Q[8]) / 2;
}
// KalmanFilter_VictoriaPark::calculateInnovation (:56-73): wrap, then range gate, then bearing gate.
__device__ __forceinline__ bool vp_gate(const Params &P, const LmKF3 &k, double z0, double z1, double &nu0, double &nu1) {
  nu0 = z0 - k.zx0;
  nu1 = wrap_pi(z1 - k.zx1);
  const bool g0 = !((P.kfRange > 0) & (fabs(nu0) > P.kfRange));
  const bool g1 = !((P.kfBearing > 0) & (fabs(nu1) > P.kfBearing));
  return g0 & g1;
}
__device__ __forceinline__ double vp_value(const Params &P, const LmKF3 &k, double pdw, double z0, double z1, double z2) {
  const double e0 = z0 - k.zx0, e1 = z1 - k.zx1, e2 = z2 - k.zx2;  // RAW difference (KalmanFilter.hpp:317-320)
  const double md2 = md2_3(k.Si, e0, e1, e2);
  if (md2 > P.newGaussMd2) return 0.0;
  const double lik = gauss_from_md2(md2, k.factor);
  if (lik == 0.0) return 0.0;
  return pdw * lik;
}

// LDS per wave: survivor list (value, packed (m,z)) + per-landmark segment + stored Pd + final normalisers.
__host__ __device__ inline size_t vp_update_lds_bytes_per_wave(int cap) { return (((size_t)cap * (8 + 4 + 4 + 8) + RFSGPU_MAX_Z * 8 + VP_PD_SCRATCH_BYTES) + 15) & ~(size_t)15; }

// RBPHDFilter::updateMap for the Victoria Park model (same phases as phd_update_map_kernel).
template <int WPB>
__global__ __launch_bounds__(WPB * 64) void vp_update_map_kernel(Buffers B, Params P, int cur, int nZ) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  double *sZ = reinterpret_cast<double *>(smem_raw);                 // [3*MAX_Z]
  double *sScan = sZ + 3 * RFSGPU_MAX_Z;                              // [RFSGPU_VP_MAX_SCAN]
  const int wave = threadIdx.x >> 6;
  const int lane = threadIdx.x & 63;
  for (int t = threadIdx.x; t < 3 * nZ; t += WPB * 64) sZ[t] = B.Z[t];
  for (int t = threadIdx.x; t < B.nScan; t += WPB * 64) sScan[t] = B.scan[t];
  __syncthreads();
  const int i = __builtin_amdgcn_readfirstlane(blockIdx.x * WPB + wave);
  if (i >= B.N) return;
  const int cap = B.cap;
  unsigned char *wb = smem_raw + (3 * RFSGPU_MAX_Z + RFSGPU_VP_MAX_SCAN) * 8 + (size_t)wave * vp_update_lds_bytes_per_wave(cap);
  double *sV = reinterpret_cast<double *>(wb);
  double *sPd = sV + cap;
  double *sCol = sPd + cap;
  unsigned *sMZ = reinterpret_cast<unsigned *>(sCol + RFSGPU_MAX_Z);
  unsigned *sSeg = sMZ + cap;  // (start << 9) | (close << 8) | count

  const int nM = B.count[i];
  const unsigned long long zmask = (nZ >= 64) ? ~0ull : ((1ull << nZ) - 1ull);
  if (nM == 0) {
    if (lane == 0) { B.unusedMask[i] = zmask; B.nInFov[i] = 0; }
    return;
  }
  double *slab = B.slab[cur];
  const double px = B.pose[3 * i], py = B.pose[3 * i + 1], pth = B.pose[3 * i + 2];
  const int nPass = (nM + 63) >> 6;
  const int room = cap - nM;
  int nFov = 0, nSurv = 0;
  double wsum = 0.0, cs = P.vpClutter;
  bool overflow = false;

  for (int p = 0; p < nPass; p++) {
    const int m = p * 64 + lane;
    const bool act = m < nM;
    Ent3 e;
    e.w = 0; e.x = 10; e.y = 10; e.d = 1; e.xx = 1; e.xy = 0; e.xd = 0; e.yy = 1; e.yd = 0; e.dd = 1;
    if (act) load_ent3(slab, cap, i, m, e, true);
    bool close = false;
    double pd = vp_pd_wave(P, sScan, B.nScan, px, py, pth, e, act, close, reinterpret_cast<unsigned char *>(sSeg + cap));
    if (!act) { pd = 0.0; close = false; }
    if (close) pd = 1;  // RBPHDFilter.hpp:604-606
    const bool fov = act && (pd != 0);
    const double pdw = pd * e.w;
    nFov += __popcll(__ballot(fov));
    if (P.useCluster) wsum += act ? e.w : 0.0;
    LmKF3 k;
    lm_precompute3(P, px, py, pth, e, k);
    unsigned long long surv = 0;
    if (fov) {
      for (int z = 0; z < nZ; z++) {
        double nu0, nu1;
        if (vp_gate(P, k, sZ[3 * z], sZ[3 * z + 1], nu0, nu1) && vp_value(P, k, pdw, sZ[3 * z], sZ[3 * z + 1], sZ[3 * z + 2]) != 0.0) surv |= 1ull << z;
      }
    }
    const int cnt = __popcll(surv);
    const int off = wave_excl_scan(cnt, lane);
    const int total = __builtin_amdgcn_readlane(off + cnt, 63);
    if (act) { sSeg[m] = ((unsigned)(nSurv + off) << 9) | (close ? 256u : 0u) | (unsigned)cnt; sPd[m] = pd; }
    int pos = nSurv + off;
    for (unsigned long long g = surv; g; g &= g - 1) {
      const int z = __builtin_ctzll(g);
      if (pos < room) {
        sV[pos] = vp_value(P, k, pdw, sZ[3 * z], sZ[3 * z + 1], sZ[3 * z + 2]);
        sMZ[pos] = ((unsigned)m << 8) | (unsigned)z;
      } else {
        overflow = true;
      }
      pos++;
    }
    wave_sync();
    {
      const int lo = nSurv, hi = (nSurv + total < room) ? nSurv + total : room;
      for (int sIdx = lo; sIdx < hi; sIdx++) {
        const unsigned mz = sMZ[sIdx];
        const double v = sV[sIdx];
        if ((int)(mz & 0xffu) == lane) cs += v;  // landmark order == the reference's summation order
      }
    }
    nSurv += total;
  }
  if (__ballot(overflow) != 0ull || nSurv > room) {
    if (lane == 0) atomicOr(B.err, ERRBIT_CAPACITY);
    nSurv = room < 0 ? 0 : (nSurv > room ? room : nSurv);
  }
  sCol[lane] = cs;
  wave_sync();

  int outBase = nM;
  unsigned long long used = 0;
  for (int s0 = 0; s0 < nSurv; s0 += 64) {
    const int sIdx = s0 + lane;
    const bool act = sIdx < nSurv;
    unsigned mz = 0;
    double v = 0.0;
    if (act) { mz = sMZ[sIdx]; v = sV[sIdx]; }
    const int m = (int)(mz >> 8), z = (int)(mz & 0xffu);
    const double wn = act ? v / sCol[z] : 0.0;
    if (act && wn != 0.0) used |= (1ull << z);
    const bool keep = act && (wn > 0.0);
    const unsigned long long km = __ballot(keep);
    if (keep) {
      const int pos = outBase + __popcll(km & ((1ull << lane) - 1ull));
      Ent3 e;
      load_ent3(slab, cap, i, m, e, false);
      LmKF3 k;
      lm_precompute3(P, px, py, pth, e, k);
      double nu0, nu1;
      vp_gate(P, k, sZ[3 * z], sZ[3 * z + 1], nu0, nu1);
      const double nu2 = sZ[3 * z + 2] - k.zx2;
      plane3(slab, cap, i, P3_W)[pos] = wn;
      plane3(slab, cap, i, P3_WP)[pos] = 0.0;
      plane3(slab, cap, i, P3_MX)[pos] = e.x + ((k.K[0] * nu0 + k.K[1] * nu1) + k.K[2] * nu2);
      plane3(slab, cap, i, P3_MY)[pos] = e.y + ((k.K[3] * nu0 + k.K[4] * nu1) + k.K[5] * nu2);
      plane3(slab, cap, i, P3_MD)[pos] = e.d + ((k.K[6] * nu0 + k.K[7] * nu1) + k.K[8] * nu2);
      plane3(slab, cap, i, P3_SXX)[pos] = k.p[0];
      plane3(slab, cap, i, P3_SXY)[pos] = k.p[1];
      plane3(slab, cap, i, P3_SXD)[pos] = k.p[2];
      plane3(slab, cap, i, P3_SYY)[pos] = k.p[3];
      plane3(slab, cap, i, P3_SYD)[pos] = k.p[4];
      plane3(slab, cap, i, P3_SDD)[pos] = k.p[5];
    }
    outBase += __popcll(km);
  }
  for (int m = lane; m < nM; m += 64) {
    double *pW = plane3(slab, cap, i, P3_W), *pWP = plane3(slab, cap, i, P3_WP);
    const double w = pW[m];
    const unsigned seg = sSeg[m];
    const bool close = (seg >> 8) & 1u;
    const double pd = sPd[m];
    double w_k = (1 - pd) * w;
    if (close && w > P.birthW) {
      const int st = (int)(seg >> 9), c = (int)(seg & 0xffu);
      double rowsum = 0.0;
      for (int q = st; q < st + c && q < nSurv; q++) rowsum += sV[q] / sCol[sMZ[q] & 0xffu];
      const double delta_w = pd * w - rowsum;
      if (delta_w > 0) { w_k += delta_w; if (w_k > 1) w_k = 1; }
    }
    pWP[m] = w;
    pW[m] = w_k;
  }
  used = wave_or_u64(used);
  if (lane == 0) {
    B.count[i] = outBase;
    B.unusedMask[i] = (~used) & zmask;
    B.nInFov[i] = nFov;
  }
  if (P.useCluster) {
    const double s = RFS_DENORM_MIN + wave_sum_dpp(wsum);
    double prod = 1.0;
    for (int z = 0; z < nZ; z++) prod *= readlane_f64(cs, z);
    if (lane == 0) B.weight[i] = exp(s) * prod * B.weight[i];
  }
}

// RBPHDFilter::importanceWeighting for the Victoria Park model (same steps as phd_weight_multifeature_kernel).
__host__ __device__ inline size_t vp_weight_lds_bytes_per_wave(int cap, int evalCap, int nZ) {
  return (weight_lds_bytes_per_wave(cap, evalCap, nZ) + (size_t)evalCap * 16 * 8 + VP_PD_SCRATCH_BYTES + 15) & ~(size_t)15;
}
template <int WPB>
__global__ __launch_bounds__(WPB * 64) void vp_weighting_kernel(Buffers B, Params P, int src, int dst, int nZ, int evalCap, MurtyQueue Q) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  double *sZ = reinterpret_cast<double *>(smem_raw);
  double *sScan = sZ + 3 * RFSGPU_MAX_Z;
  const int wave = threadIdx.x >> 6;
  const int lane = threadIdx.x & 63;
  for (int t = threadIdx.x; t < 3 * nZ; t += WPB * 64) sZ[t] = B.Z[t];
  for (int t = threadIdx.x; t < B.nScan; t += WPB * 64) sScan[t] = B.scan[t];
  __syncthreads();
  const int i = __builtin_amdgcn_readfirstlane(blockIdx.x * WPB + wave);
  if (i >= B.N) return;
  const int cap = B.cap;
  WeightLDS s;
  // evX/evY hold x,y; the diameter of the evaluation points goes into evZ's spare slot via a separate array below
  unsigned char *wbase = smem_raw + (3 * RFSGPU_MAX_Z + RFSGPU_VP_MAX_SCAN) * 8 + (size_t)wave * vp_weight_lds_bytes_per_wave(cap, evalCap, nZ);
  carve_weight_lds(wbase, cap, evalCap, nZ, s);
  double *evD = reinterpret_cast<double *>(wbase + weight_lds_bytes_per_wave(cap, evalCap, nZ));  // [evalCap][16]: d, z_exp(3), Si(9), factor
  unsigned char *pdScratch = wbase + weight_lds_bytes_per_wave(cap, evalCap, nZ) + (size_t)evalCap * 16 * 8;          // vp_pd_wave
  const int N = B.count[i];
  const double *sl = B.slab[src];
  double *dl = B.slab[dst];
  const double *qW = sl + ((size_t)i * P3_COUNT + P3_W) * cap, *qWP = sl + ((size_t)i * P3_COUNT + P3_WP) * cap;
  const double px = B.pose[3 * i], py = B.pose[3 * i + 1], pth = B.pose[3 * i + 2];

  int nEvalPoints = ((unsigned)P.evalCount > (unsigned)N) ? N : P.evalCount;
  if (nEvalPoints == 0) {
    for (int pl = 0; pl < P3_COUNT; pl++)
      for (int m = lane; m < N; m += 64) (dl + ((size_t)i * P3_COUNT + pl) * cap)[m] = (sl + ((size_t)i * P3_COUNT + pl) * cap)[m];
    if (lane == 0) B.weight[i] = RFS_DENORM_MIN;
    return;
  }
  // rank sort (exact fp64 form: Victoria Park mixtures are small)
  for (int m = lane; m < N; m += 64) s.keys[m] = qW[m];
  wave_sync();
  for (int m = lane; m < N; m += 64) {
    const double wm = s.keys[m];
    int rank = 0;
    for (int j = 0; j < N; j++) {
      const double wj = s.keys[j];
      rank += ((wj > wm) | ((wj == wm) & (j < m))) ? 1 : 0;
    }
    s.perm[rank] = m;
  }
  wave_sync();
  for (int r = lane; r < N; r += 64) {
    const int m = s.perm[r];
    for (int pl = 0; pl < P3_COUNT; pl++) (dl + ((size_t)i * P3_COUNT + pl) * cap)[r] = (sl + ((size_t)i * P3_COUNT + pl) * cap)[m];
  }
  // evaluation points
  int nE = 0;
  {
    const int limit = nEvalPoints < RFSGPU_MAX_EVAL ? nEvalPoints : RFSGPU_MAX_EVAL;
    bool done = false;
    for (int c0 = 0; c0 < N && !done; c0 += 64) {
      const int r = c0 + lane;
      bool below = true, cand = false;
      Ent3 e;
      double pd = 0;
      e.w = 0; e.x = 10; e.y = 10; e.d = 1; e.xx = 1; e.xy = 0; e.xd = 0; e.yy = 1; e.yd = 0; e.dd = 1;
      if (r < N) {
        const int m = s.perm[r];
        below = s.keys[m] < P.evalMinW;
        load_ent3(sl, cap, i, m, e, false);
      }
      {
        bool close;
        const bool want = (r < N) && !below;
        pd = vp_pd_wave(P, sScan, B.nScan, px, py, pth, e, want, close, pdScratch);
        if (!want) pd = 0;
        cand = pd > 0;
      }
      const unsigned long long belowMask = __ballot(below);
      const unsigned long long valid = belowMask ? ((1ull << __builtin_ctzll(belowMask)) - 1ull) : ~0ull;
      if (belowMask) done = true;
      const unsigned long long candMask = __ballot(cand) & valid;
      const int need = limit - nE;
      const int before = __popcll(candMask & ((1ull << lane) - 1ull));
      if (((candMask >> lane) & 1ull) && before < need) {
        const int ev = nE + before;
        s.evX[ev] = e.x; s.evY[ev] = e.y; evD[16 * ev] = e.d;
        s.evPd[ev] = pd;
        s.evLog1mPd[ev] = log(1 - pd);
      }
      int got = __popcll(candMask);
      if (got >= need) { got = need; done = true; }
      nE += got;
    }
    if (nE == limit && nEvalPoints > limit && lane == 0) atomicOr(B.err, ERRBIT_EVALPTS);
  }
  wave_sync();
  // weight sums + intensity products
  double sumPrev = 0.0, sumCur = 0.0;
  for (int m = lane; m < N; m += 64) { sumPrev += qWP[m]; sumCur += s.keys[m]; }
  sumPrev = wave_sum_dpp(sumPrev);
  sumCur = wave_sum_dpp(sumCur);
  double prodBefore = 1.0, prodAfter = 1.0;
  for (int e0 = 0; e0 < nE; e0 += 4) {
    double accB[4], accA[4];
#pragma unroll
    for (int t = 0; t < 4; t++) { accB[t] = 0.0; accA[t] = 0.0; }
    for (int m = lane; m < N; m += 64) {
      Ent3 g;
      load_ent3(sl, cap, i, m, g, false);
      const double w = s.keys[m], wp = qWP[m];
      double Sm[9], Si[9];
      full3(g, Sm);
      inv3(Sm, Si);
      const double factor = sqrt(P.twoPiPowD * det3(Sm));
#pragma unroll
      for (int t = 0; t < 4; t++) {
        const int ev = (e0 + t < nE) ? e0 + t : e0;
        const double lik = gauss_from_md2(md2_3(Si, s.evX[ev] - g.x, s.evY[ev] - g.y, evD[16 * ev] - g.d), factor);
        accB[t] += wp * lik;
        accA[t] += w * lik;
      }
    }
#pragma unroll
    for (int t = 0; t < 4; t++)
      if (e0 + t < nE) {
        prodBefore *= (RFS_DENORM_MIN + wave_sum_dpp(accB[t]));
        prodAfter *= (RFS_DENORM_MIN + wave_sum_dpp(accA[t]));
      }
  }
  // likelihood table
  if (lane < nE) {
    VPMeas o;
    vp_measure(P, px, py, pth, s.evX[lane], s.evY[lane], evD[16 * lane], 0.0, 0.0, 0.0, 0.0, o);  // evalPt_copy.setCov(Zero)
    double *zz = evD + 16 * lane;
    zz[1] = o.z0; zz[2] = o.z1; zz[3] = o.z2;
    inv3(o.S, zz + 4);
    zz[13] = sqrt(P.twoPiPowD * det3(o.S));
  }
  wave_sync();
  for (int idx = lane; idx < nE * nZ; idx += 64) {
    const int ev = idx / nZ, n = idx - ev * nZ;
    const double *zz = evD + 16 * ev;
    const double md2 = md2_3(zz + 4, sZ[3 * n] - zz[1], sZ[3 * n + 1] - zz[2], sZ[3 * n + 2] - zz[3]);
    double Lv = gauss_from_md2(md2, zz[13]) * s.evPd[ev];
    if (md2 > P.weightingMd2) Lv = 0.0;
    s.L[idx] = Lv;
  }
  wave_sync();
  const double l = rfs_partitions_wave(s, nE, nZ, P.vpClutter, lane, i, Q, B.err, P.exactPartitions);
  const double ml = l / P.vpExpClutter;  // clutterIntensityIntegral (:287-290)
  const double overall = ml * prodBefore / prodAfter * exp(sumCur - sumPrev);
  if (lane == 0) B.weight[i] = overall * B.weight[i];
}

// LDS per wave for the merge: 42 B per Gaussian -- position (x, y, d), prefilter bound and weight as doubles, a u16 survivor list.
// (Until r02 every entry was staged with its covariance AND the inverse, 130 B: 25 KB per wave at cap 192, six waves per CU, and
// the 5000 particles of configs[3] took 3.3 rounds.  Now the distance prefilter below runs on the LDS copy and everything the
// exact test needs comes from the slab for the few pairs that survive it; all 5000 waves are resident at once.)
__host__ __device__ inline size_t vp_merge_lds_bytes_per_wave(int cap) { return (((size_t)cap * (5 * 8 + 2)) + 15) & ~(size_t)15; }

// Necessary condition for a pair to pass GaussianMixture::merge's test with covariance S: e^T S^-1 e >= |e|^2 / lambda_max(S) >=
// |e|^2 / tr(S) for a positive definite S, so md2 <= t^2 implies |e|^2 <= t^2 tr(S).  The bound carries a 1e-6 relative margin for
// the rounding of the computed inverse / md2; a covariance that is not positive definite (Sylvester), not finite, or so
// ill-conditioned that the computed md2 may be off by more than that margin gets an infinite bound -- always fully tested -- so
// the prefilter never changes a decision of the exact test.
__device__ __forceinline__ double merge_bound3(double t2, const Ent3 &e) {
  double S[9];
  full3(e, S);
  const double tr = (e.xx + e.yy) + e.dd, m2 = e.xx * e.yy - e.xy * e.xy, det = det3(S);
  const bool sane = (e.xx > 0.0) && (m2 > 0.0) && (det > 1e-9 * tr * tr * tr) && (tr < 1.0e100);
  return sane ? t2 * tr * (1.0 + 1e-6) : __builtin_huge_val();
}

// GaussianMixture::merge for 3-D Gaussians: the exact sequential-greedy scan (lanes test 64 candidates j at once against
// the current state of a, lowest passing lane merged, lanes above it re-tested), optional fused prune.  Merged rows are
// written back in place (a row is never read again once the scan has passed it); the prune reads the slab.
template <int WPB, bool FUSE_PRUNE>
__global__ __launch_bounds__(WPB * 64) void vp_merge_kernel(Buffers B, Params P, int cur, int dst) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const int wave = threadIdx.x >> 6;
  const int lane = threadIdx.x & 63;
  const int i = __builtin_amdgcn_readfirstlane(blockIdx.x * WPB + wave);
  if (i >= B.N) return;
  const int cap = B.cap;
  double *sb = reinterpret_cast<double *>(smem_raw + (size_t)wave * vp_merge_lds_bytes_per_wave(cap));
  double *sX = sb, *sY = sb + cap, *sD = sb + 2 * (size_t)cap, *sBnd = sb + 3 * (size_t)cap, *sW = sb + 4 * (size_t)cap;
  unsigned short *sIdx = reinterpret_cast<unsigned short *>(sb + 5 * (size_t)cap);
  const int N = B.count[i];
  double *slab = B.slab[cur];
  const double t2 = P.mergeT2, f = P.mergeInfl;
  unsigned hole = 0;
  for (int m = lane, sidx = 0; m < N; m += 64, sidx++) {
    Ent3 e;
    load_ent3(slab, cap, i, m, e, true);
    sX[m] = e.x; sY[m] = e.y; sD[m] = e.d;
    sBnd[m] = merge_bound3(t2, e);
    sW[m] = e.w;
    if (e.w < 0) hole |= 1u << sidx;
  }
  wave_sync();
  // inverse of a stored covariance as the reference's test reads it: Eigen's cofactor inverse of the full matrix, of which the
  // scan (like the r01 kernel, which kept exactly these six numbers per entry) uses the upper triangle mirrored
  auto inverse_of = [&](const Ent3 &e, double I9[9]) {
    double Sm[9], Si[9];
    full3(e, Sm);
    inv3(Sm, Si);
    I9[0] = Si[0]; I9[1] = Si[1]; I9[2] = Si[2]; I9[3] = Si[1]; I9[4] = Si[4]; I9[5] = Si[5]; I9[6] = Si[2]; I9[7] = Si[5]; I9[8] = Si[8];
  };
  // Rows that can merge at all: a row's first merge needs a partner that passes the exact test against the row's INITIAL state,
  // hence the distance prefilter on the initial states; a row without such a partner goes through the reference's scan
  // unchanged, so it is skipped outright (its record is never fetched).  On a Victoria Park map that is all but a few rows.
  unsigned rowFlag = 0;
  for (int m = lane, sidx = 0; m < N; m += 64, sidx++) {
    if ((hole >> sidx) & 1u) continue;
    const double mx = sX[m], my = sY[m], md = sD[m], mb = sBnd[m];
    bool c = false;
    for (int j = m + 1; j < N && !c; j++) {
      const double e0 = sX[j] - mx, e1 = sY[j] - my, e2 = sD[j] - md;
      c = !(((e0 * e0 + e1 * e1) + e2 * e2) > fmax(mb, sBnd[j])) && !(sW[j] < 0);
    }
    if (c) rowFlag |= 1u << sidx;
  }
  bool anyMerge = false;
  for (int r0 = 0; r0 < N; r0 += 64)
  for (unsigned long long rows = __ballot((rowFlag >> (r0 >> 6)) & 1u); rows; rows &= rows - 1ull) {
    const int a = r0 + __builtin_ctzll(rows);
    Ent3 ea;
    ea.w = 0;
    load_ent3(slab, cap, i, a, ea, false);                       // (wave-uniform address: one broadcast load per plane)
    const unsigned ownerHole = (unsigned)__builtin_amdgcn_readlane((int)hole, a & 63);
    if ((ownerHole >> (a >> 6)) & 1u) continue;
    double ax = ea.x, ay = ea.y, ad = ea.d, aw = sW[a], ab = sBnd[a];
    double aS[9], aI[9];
    full3(ea, aS);
    inverse_of(ea, aI);
    bool changed = false;
    for (int c0 = (a + 1) & ~63; c0 < N; c0 += 64) {
      const int j = c0 + lane;
      const int slot = c0 >> 6;
      bool live = (j > a) && (j < N) && !((hole >> slot) & 1u);
      int floorLane = 0;
      while (true) {
        bool pass = false;
        if (live && lane >= floorLane) {
          const double e0 = sX[j] - ax, e1 = sY[j] - ay, e2 = sD[j] - ad;
          if (!(((e0 * e0 + e1 * e1) + e2 * e2) > fmax(ab, sBnd[j]))) {   // (NaN distances and infinite bounds fall through to the exact test)
            bool far = md2_3(aI, e0, e1, e2) > t2;
            if (far) {
              Ent3 ej;
              load_ent3(slab, cap, i, j, ej, false);
              double jI[9];
              inverse_of(ej, jI);
              far = md2_3(jI, -e0, -e1, -e2) > t2;
            }
            pass = !far && ((aw + sW[j]) != 0.0);
          }
        }
        const unsigned long long pm = __ballot(pass);
        if (pm == 0ull) break;
        const int l = __builtin_ctzll(pm);
        const int jj = c0 + l;
        Ent3 eb;
        load_ent3(slab, cap, i, jj, eb, false);                  // uniform
        const double w1 = aw, w2 = sW[jj];
        const double bx[3] = {eb.x, eb.y, eb.d};
        double bS[9];
        full3(eb, bS);
        const double wm = w1 + w2;
        const double axv[3] = {ax, ay, ad};
        double xm[3], d1[3], d2[3];
#pragma unroll
        for (int k = 0; k < 3; k++) { xm[k] = (axv[k] * w1 + bx[k] * w2) / wm; d1[k] = xm[k] - axv[k]; d2[k] = xm[k] - bx[k]; }
        double nS[9];
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
          for (int c = 0; c < 3; c++) nS[3 * r + c] = (w1 * (aS[3 * r + c] + (f * d1[r]) * d1[c]) + w2 * (bS[3 * r + c] + (f * d2[r]) * d2[c])) / wm;
        ax = xm[0]; ay = xm[1]; ad = xm[2]; aw = wm;
        // packed symmetric storage: keep the upper triangle (the lower one is its mirror up to rounding of f*d_r*d_c order)
        nS[3] = nS[1]; nS[6] = nS[2]; nS[7] = nS[5];
#pragma unroll
        for (int k = 0; k < 9; k++) aS[k] = nS[k];
        inv3(aS, aI);
        // the reference inverts the full matrix; mirror its symmetric reads
        aI[3] = aI[1]; aI[6] = aI[2]; aI[7] = aI[5];
        {
          Ent3 em;
          em.w = aw; em.x = ax; em.y = ay; em.d = ad;
          em.xx = aS[0]; em.xy = aS[1]; em.xd = aS[2]; em.yy = aS[4]; em.yd = aS[5]; em.dd = aS[8];
          ab = merge_bound3(t2, em);
        }
        changed = true;
        if (lane == l) { hole |= 1u << slot; live = false; }
        floorLane = l + 1;
        if (floorLane >= 64) break;
      }
    }
    if (changed) {
      anyMerge = true;
      sW[a] = aw;   // (uniform store; the position / bound of a are never read again: a is behind the scan)
      if (lane == 0) {
        plane3(slab, cap, i, P3_W)[a] = aw;
        plane3(slab, cap, i, P3_MX)[a] = ax; plane3(slab, cap, i, P3_MY)[a] = ay; plane3(slab, cap, i, P3_MD)[a] = ad;
        plane3(slab, cap, i, P3_SXX)[a] = aS[0]; plane3(slab, cap, i, P3_SXY)[a] = aS[1]; plane3(slab, cap, i, P3_SXD)[a] = aS[2];
        plane3(slab, cap, i, P3_SYY)[a] = aS[4]; plane3(slab, cap, i, P3_SYD)[a] = aS[5]; plane3(slab, cap, i, P3_SDD)[a] = aS[8];
      }
    }
  }
  // in-place updates of merged rows (global memory, lane 0) -> visible to the wave's other lanes
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  wave_sync();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  if (!FUSE_PRUNE) {
    if (!anyMerge) return;
    for (int m = lane, sidx = 0; m < N; m += 64, sidx++)
      if ((hole >> sidx) & 1u) plane3(slab, cap, i, P3_W)[m] = -1.0;
    return;
  }
  for (int m = lane, sidx = 0; m < N; m += 64, sidx++)
    if ((hole >> sidx) & 1u) sW[m] = -1.0;
  wave_sync();
  double *dl = B.slab[dst];
  const double t = P.pruneT;
  int nSurv = 0;
  for (int c0 = 0; c0 < N; c0 += 64) {
    const int m = c0 + lane;
    const double wm = (m < N) ? sW[m] : -1.0;
    const bool keep = (wm >= t) && (wm >= 0.0);
    const unsigned long long km = __ballot(keep);
    if (keep) sIdx[nSurv + __popcll(km & ((1ull << lane) - 1ull))] = (unsigned short)m;
    nSurv += __popcll(km);
  }
  wave_sync();
  for (int q = lane; q < nSurv; q += 64) {
    const int m = sIdx[q];
    const double wm = sW[m];
    Ent3 e;
    load_ent3(slab, cap, i, m, e, false);   // the survivor's record first (independent loads in flight while the rank is counted)
    int rank = 0;
    for (int q2 = 0; q2 < nSurv; q2++) {
      const int j2 = sIdx[q2];
      const double wj = sW[j2];
      rank += ((wj > wm) | ((wj == wm) & (j2 < m))) ? 1 : 0;
    }
    plane3(dl, cap, i, P3_W)[rank] = wm;
    plane3(dl, cap, i, P3_WP)[rank] = 0.0;
    plane3(dl, cap, i, P3_MX)[rank] = e.x; plane3(dl, cap, i, P3_MY)[rank] = e.y; plane3(dl, cap, i, P3_MD)[rank] = e.d;
    plane3(dl, cap, i, P3_SXX)[rank] = e.xx; plane3(dl, cap, i, P3_SXY)[rank] = e.xy; plane3(dl, cap, i, P3_SXD)[rank] = e.xd;
    plane3(dl, cap, i, P3_SYY)[rank] = e.yy; plane3(dl, cap, i, P3_SYD)[rank] = e.yd; plane3(dl, cap, i, P3_SDD)[rank] = e.dd;
  }
  if (lane == 0) B.count[i] = nSurv;
}
