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
#ifndef VP_PD_OWN_MAX
#define VP_PD_OWN_MAX 512      // pool items whose landmark is looked up in a byte table (0: every item by binary search over the block starts)
#endif
#define VP_PD_SCRATCH_BYTES (64 * (5 * 8 + 2 * 8) + 65 * 4 + 4 + 64 * 4 + VP_PD_OWN_MAX)
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
  int *lcl = pre + 66;                                     // the unshifted landmark's near-limit verdict
  lx[lane] = e.x; ly[lane] = e.y; ld[lane] = e.d; lp0[lane] = p0; lp1[lane] = p1;
  lmn[lane] = 0x7ff0000000000000ull;                       // +inf
  lmx[lane] = 0ull;                                        // pd >= 0
  lcl[lane] = 0;
  // the pool: 2 n shifted copies + the landmark itself (r3: the unshifted evaluation used to be a separate round of all lanes --
  // with 40 landmarks of one copy pair each that was 2 + 1 rounds of probabilityOfDetection2 where the pool needs 2)
  const int cnt = act ? 2 * n + 1 : 0;
  const int off = wave_excl_scan(cnt, lane);
  const int total = __builtin_amdgcn_readlane(off + cnt, 63);
  pre[lane] = off;
  if (lane == 63) pre[64] = total;
  // (round 6) the owner of the first VP_PD_OWN_MAX items from a byte table the landmarks fill: the binary search below is six
  // DEPENDENT LDS reads per item
  [[maybe_unused]] unsigned char *own = reinterpret_cast<unsigned char *>(lcl + 64);
  if (VP_PD_OWN_MAX > 0)
    for (int k = 0; k < cnt && off + k < VP_PD_OWN_MAX; k++) own[off + k] = (unsigned char)lane;
  wave_sync();
  for (int t = lane; t < total; t += 64) {
    int l = 0;                                             // owner: the last landmark whose block starts at or before t
    if (VP_PD_OWN_MAX > 0 && t < VP_PD_OWN_MAX) l = own[t];
    else {
#pragma unroll
      for (int st = 32; st >= 1; st >>= 1) l += (pre[l + st] <= t) ? st : 0;
    }
    const int j = t - pre[l];
    const bool self = (t + 1 == pre[l + 1]);               // the last item of a landmark's block: the landmark where it is
    const int i = (j >> 1) + 1;
    const double dd = ld[l];
    const double sh = self ? 0.0 : i * 2 * dd;
    bool c2;
    double qx, qy;
    if (self) { qx = lx[l]; qy = ly[l]; }
    else if (j & 1) { qx = lx[l] - sh * lp0[l]; qy = ly[l] - sh * lp1[l]; }
    else { qx = lx[l] + sh * lp0[l]; qy = ly[l] + sh * lp1[l]; }
    const double p = vp_pd2(P, scan, nScan, px, py, pth, qx, qy, dd, c2);
    if (self) lcl[l] = c2 ? 1 : 0;                         // `close` is the unshifted landmark's verdict (the reference's last call)
    const unsigned long long bits = (unsigned long long)__double_as_longlong(p);
    atomicMin(&lmn[l], bits);
    atomicMax(&lmx[l], bits);
  }
  wave_sync();
  const double mn = __longlong_as_double((long long)lmn[lane]), mx = __longlong_as_double((long long)lmx[lane]);
  close = lcl[lane] != 0;
  if (mn == 0 && mx > 0) close = true;
  wave_sync();                                             // (the scratch is reused by the next pass)
  return act ? mx : 0.0;
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
// Candidate mask of one landmark for the innovation gates of vp_gate: a sweep over the measurement set in fp32 with the thresholds widened by a
// bound on the fp32 rounding of the operands, the difference and the 2-pi reduction (the 2-D kernel's gate_candidates, update_map.h, one
// measurement at a time from the fp64 copy in LDS): it can only let extra pairs through; every candidate then goes through the exact gate.
__device__ __forceinline__ unsigned long long vp_gate_candidates(const Params &P, const double zx0d, const double zx1d, const int nZ, const double *sZ,
                                                                 const float zrMax, const float zbMax) {
  const float zx0 = (float)zx0d, zx1 = (float)zx1d;
  const float inf = __builtin_huge_valf();
  float thrR = (P.kfRange > 0) ? (float)P.kfRange * (1.f + 1e-6f) + 2.5e-7f * (zrMax + fabsf(zx0)) + 1e-30f : inf;
  float thrB = (P.kfBearing > 0) ? (float)P.kfBearing * (1.f + 1e-6f) + 1e-6f * (zbMax + 3.2f) + 1e-6f : inf;
  if (!(zbMax < 50.f) || !(fabsf(zx1) < 50.f)) thrB = inf;       // far outside any sensible bearing range (or NaN): the exact test decides
  if (!(zrMax < 1.0e30f)) thrR = inf;
  unsigned long long m = 0ull;
#pragma unroll 4
  for (int z = 0; z < nZ; z++) {
    const float e0 = (float)sZ[3 * z] - zx0;
    float w = (float)sZ[3 * z + 1] - zx1;
    w = w - __builtin_rintf(w * 0.15915494309189533577f) * 6.2831853071795864769f;
    const bool c = ((int)!(fabsf(e0) > thrR) & (int)!(fabsf(w) > thrB)) != 0;    // !(x > t) form: a NaN is a candidate
    m |= c ? (1ull << z) : 0ull;
  }
  return m;
}
__device__ __forceinline__ double vp_value(const Params &P, const LmKF3 &k, double pdw, double z0, double z1, double z2) {
  const double e0 = z0 - k.zx0, e1 = z1 - k.zx1, e2 = z2 - k.zx2;  // RAW difference (KalmanFilter.hpp:317-320)
  const double md2 = md2_3(k.Si, e0, e1, e2);
  if (md2 > P.newGaussMd2) return 0.0;
  const double lik = gauss_from_md2(md2, k.factor);
  if (lik == 0.0) return 0.0;
  return pdw * lik;
}

// LDS shared by the waves of a workgroup: the measurement set (3 nZ doubles) and the laser scan (nScan doubles) -- sized by what
// is there (r02 reserved 64 measurements and 720 beams: 7.3 KB per workgroup where a Victoria Park scan has 361 beams and ~12
// measurements; the LDS block is what decides how many particles a CU holds).
__host__ __device__ inline size_t vp_shared_lds_bytes(int nZ, int nScan) { return (((size_t)(3 * nZ + nScan) * 8) + 15) & ~(size_t)15; }
// LDS per wave: survivor list (value, packed (m,z)) + per-landmark segment + stored Pd + final normalisers.
__host__ __device__ inline size_t vp_update_lds_bytes_per_wave(int cap) { return (((size_t)cap * (8 + 4 + 4 + 8) + RFSGPU_MAX_Z * 8 + VP_PD_SCRATCH_BYTES) + 15) & ~(size_t)15; }

// RBPHDFilter::updateMap for the Victoria Park model (same phases as phd_update_map_kernel).
// One wavefront takes particle i through the map update.  sZ / sScan: the measurement set and the laser scan, staged in LDS by
// the caller (shared by the waves of a workgroup); wb: vp_update_lds_bytes_per_wave(cap) bytes of LDS of this wave.
// pdCache (the fused step; may be null): receives, per landmark, the Pd-table index of its probabilityOfDetection (254: not a table value, evaluate again; 255: exactly 0
// by an early return) -- the weighting phase asks for the same number again when it picks evaluation points among landmarks
// the update has not moved.  Returns the number of landmarks before the update.
__device__ int vp_update_map_particle(const Buffers &B, const Params &P, int cur, int nZ, int i, int lane, const double *sZ, const double *sScan,
                                      unsigned char *wb, unsigned char *pdCache = nullptr) {
  const int cap = B.cap;
  double *sV = reinterpret_cast<double *>(wb);
  double *sPd = sV + cap;
  double *sCol = sPd + cap;
  unsigned *sMZ = reinterpret_cast<unsigned *>(sCol + RFSGPU_MAX_Z);
  unsigned *sSeg = sMZ + cap;  // (start << 9) | (close << 8) | count

  const int nM = B.count[i];
  const unsigned long long zmask = (nZ >= 64) ? ~0ull : ((1ull << nZ) - 1ull);
  if (nM == 0) {
    if (lane == 0) { B.unusedMask[i] = zmask; B.nInFov[i] = 0; }
    return 0;
  }
  double *slab = B.slab[cur];
  const double px = B.pose[3 * i], py = B.pose[3 * i + 1], pth = B.pose[3 * i + 2];
  // largest |range| and |bearing| of the measurement set: the rounding bounds of the fp32 gate sweep (vp_gate_candidates)
  const float zrMax = wave_max_f32((lane < nZ) ? fabsf((float)sZ[3 * lane]) : 0.f), zbMax = wave_max_f32((lane < nZ) ? fabsf((float)sZ[3 * lane + 1]) : 0.f);
  const int nPass = (nM + 63) >> 6;
  const int room = cap - nM;
  DBG_T(0, 0);
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
    if (pdCache && act) {          // (the value is a table entry or the 0.0 of an early return: kept as an index, one byte)
      int idx = (pd == 0.0) ? 255 : 254;   // 255: exactly 0; 254: a value that is no table entry (NaN entries, ...): NOT cached, re-evaluated later
      for (int k = 0; k < P.nPd; k++) idx = (P.PdTable[k] == pd) ? k : idx;
      pdCache[m] = (unsigned char)idx;
    }
    if (p == 0) DBG_T(0, 1);
    RFS_CUT(101);
    if (close) pd = 1;  // RBPHDFilter.hpp:604-606
    const bool fov = act && (pd != 0);
    const double pdw = pd * e.w;
    nFov += __popcll(__ballot(fov));
    if (P.useCluster) wsum += act ? e.w : 0.0;
    LmKF3 k;
    lm_precompute3(P, px, py, pth, e, k);
    if (p == 0) DBG_T(0, 2);
    RFS_CUT(102);
    unsigned long long surv = 0;
    if (fov) {
      // gates and Mahalanobis distance for every measurement, the Gaussian only for the pairs inside the gate (one loop for both made
      // every trip pay for the exp and the division as soon as ONE lane's pair had passed)
      unsigned long long gated = 0;
      for (unsigned long long g = vp_gate_candidates(P, k.zx0, k.zx1, nZ, sZ, zrMax, zbMax); g; g &= g - 1) {
        const int z = __builtin_ctzll(g);
        double nu0, nu1;
        if (vp_gate(P, k, sZ[3 * z], sZ[3 * z + 1], nu0, nu1) &&
            !(md2_3(k.Si, sZ[3 * z] - k.zx0, sZ[3 * z + 1] - k.zx1, sZ[3 * z + 2] - k.zx2) > P.newGaussMd2)) gated |= 1ull << z;
      }
      for (unsigned long long g = gated; g; g &= g - 1) {
        const int z = __builtin_ctzll(g);
        if (vp_value(P, k, pdw, sZ[3 * z], sZ[3 * z + 1], sZ[3 * z + 2]) != 0.0) surv |= 1ull << z;
      }
    }
    if (p == 0) DBG_T(0, 3);
    RFS_CUT(103);
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
  DBG_T(0, 4);
  RFS_CUT(104);

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
  DBG_T(0, 5);
  RFS_CUT(105);
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
  DBG_T(0, 6);
  RFS_CUT(106);
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
  return nM;
}
template <int WPB>
__global__ __launch_bounds__(WPB * 64) void vp_update_map_kernel(Buffers B, Params P, int cur, int nZ) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  double *sZ = reinterpret_cast<double *>(smem_raw);                 // [3 nZ]
  double *sScan = sZ + 3 * nZ;                                        // [nScan]
  const int wave = threadIdx.x >> 6;
  const int lane = threadIdx.x & 63;
  for (int t = threadIdx.x; t < 3 * nZ; t += WPB * 64) sZ[t] = B.Z[t];
  for (int t = threadIdx.x; t < B.nScan; t += WPB * 64) sScan[t] = B.scan[t];
  __syncthreads();
  const int i = __builtin_amdgcn_readfirstlane(blockIdx.x * WPB + wave);
  if (i >= B.N) return;
  vp_update_map_particle(B, P, cur, nZ, i, lane, sZ, sScan, smem_raw + vp_shared_lds_bytes(nZ, B.nScan) + (size_t)wave * vp_update_lds_bytes_per_wave(B.cap));
}

// RBPHDFilter::importanceWeighting for the Victoria Park model (same steps as phd_weight_multifeature_kernel).
// LDS of the weighting phase, per wave.  Its own layout since r3 (r02 took the 2-D kernel's WeightLDS -- with the float keys
// and evaluation-point tables this model never touches -- and added its own arrays behind it: 15.2 KB at cap 192, which
// held a CU to 8 waves once the three phases shared one allocation).  Early part: keys, sort permutation, evaluation points
// (x, y, Pd, log(1 - Pd)) and their 16-double records (diameter, z_exp, S^-1, factor); late part, written only after the
// evaluation points are chosen: likelihood table, component labels / masks, partition likelihoods -- the Pd scratch of the
// selection (vp_pd_wave) lies over it.
__host__ __device__ inline size_t vp_weight_lds_late_bytes(int evalCap, int nZ) {
  size_t late = (size_t)evalCap * nZ * 8 + 64 * 4 * 2 + 128 * 8 * 2 + 128 * 8;
  if (late < (size_t)VP_PD_SCRATCH_BYTES) late = VP_PD_SCRATCH_BYTES;
  return late;
}
// The evaluation-point tables + the late part double as the scratch of the tie-order replay (stdsort_replay.h) right after the rank
// sort; with few evaluation points and a large gm_capacity they would be smaller than what the replay cannot do without -- the
// range is then padded up to that (such a configuration used to be accepted at create and to fail with a capacity
// error in the middle of a run).
__host__ __device__ inline size_t vp_weight_scratch_bytes(int cap, int evalCap, int nZ) {
  const size_t a = (size_t)evalCap * 8 * 4 + (size_t)evalCap * 16 * 8 + vp_weight_lds_late_bytes(evalCap, nZ), b = (ss_must_bytes(cap) + 7) & ~(size_t)7;
  return a > b ? a : b;
}
__host__ __device__ inline size_t vp_weight_lds_bytes_per_wave(int cap, int evalCap, int nZ) {
  return ((size_t)cap * 8 + (size_t)((cap + 63) & ~63) * 4 + vp_weight_scratch_bytes(cap, evalCap, nZ) + 15) & ~(size_t)15;
}
__device__ __forceinline__ void carve_vp_weight_lds(unsigned char *base, int cap, int evalCap, int nZ, WeightLDS &s, double *&evD, unsigned char *&pdScratch) {
  unsigned char *p = base;
  s.keys = (double *)p; p += (size_t)cap * 8;
  s.evX = (double *)p; p += (size_t)evalCap * 8;
  s.evY = (double *)p; p += (size_t)evalCap * 8;
  s.evPd = (double *)p; p += (size_t)evalCap * 8;
  s.evLog1mPd = (double *)p; p += (size_t)evalCap * 8;
  evD = (double *)p; p += (size_t)evalCap * 16 * 8;
  pdScratch = p;                                            // (over the late part)
  s.L = (double *)p; p += (size_t)evalCap * nZ * 8;
  s.compRows = (unsigned long long *)p; p += 128 * 8;
  s.compCols = (unsigned long long *)p; p += 128 * 8;
  s.partLik = (double *)p; p += 128 * 8;
  s.labR = (int *)p; p += 64 * 4;
  s.labC = (int *)p; p += 64 * 4;
  s.perm = (int *)(base + (size_t)cap * 8 + vp_weight_scratch_bytes(cap, evalCap, nZ));
  s.fkeys = nullptr; s.evIdx = nullptr; s.evZ = nullptr; s.isum = nullptr;    // (2-D kernel only)
}
// One wavefront takes particle i through the weighting.  permOut == nullptr: the mixture sorted by weight is written to slab
// `dst` (the stand-alone kernel; the merge kernel then works on that copy).  permOut != nullptr (the fused step): the sorted
// order is handed on as a permutation in LDS (rank -> storage index, u16[cap]) and nothing is written to `dst`.
__device__ void vp_weight_particle(const Buffers &B, const Params &P, int src, int dst, int nZ, int evalCap, const MurtyQueue &Q, int i, int lane,
                                   const double *sZ, const double *sScan, unsigned char *wbase, unsigned short *permOut,
                                   const unsigned char *pdCache = nullptr, int nCached = 0) {
  const int cap = B.cap;
  WeightLDS s;
  double *evD;                  // [evalCap][16]: d, z_exp(3), Si(9), factor
  unsigned char *pdScratch;     // vp_pd_wave
  carve_vp_weight_lds(wbase, cap, evalCap, nZ, s, evD, pdScratch);
  const int N = B.count[i];
  const double *sl = B.slab[src];
  double *dl = B.slab[dst];
  const double *qW = sl + ((size_t)i * P3_COUNT + P3_W) * cap, *qWP = sl + ((size_t)i * P3_COUNT + P3_WP) * cap;
  const double px = B.pose[3 * i], py = B.pose[3 * i + 1], pth = B.pose[3 * i + 2];

  int nEvalPoints = ((unsigned)P.evalCount > (unsigned)N) ? N : P.evalCount;
  if (nEvalPoints == 0) {
    if (permOut) {
      for (int m = lane; m < N; m += 64) permOut[m] = (unsigned short)m;
    } else {
      for (int pl = 0; pl < P3_COUNT; pl++)
        for (int m = lane; m < N; m += 64) (dl + ((size_t)i * P3_COUNT + pl) * cap)[m] = (sl + ((size_t)i * P3_COUNT + pl) * cap)[m];
    }
    if (lane == 0) B.weight[i] = RFS_DENORM_MIN;
    return;
  }
  // rank sort (exact fp64 form: Victoria Park mixtures are small)
  DBG_T(16, 0);
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
  {  // equal weights in the order std::sort leaves them (stdsort_replay.h): positions in the upper halves of s.perm's words, the
     // rest of the scratch over the evaluation-point tables and the late part, which are written only after this
    StdSortScratch ss;
    ss.pos = reinterpret_cast<unsigned short *>(s.perm) + 1;
    ss.posStride = 2;
    unsigned char *sbuf = reinterpret_cast<unsigned char *>(s.evX);
    const size_t sbytes = vp_weight_scratch_bytes(cap, evalCap, nZ);
    if (!ss_carve(ss, sbuf, sbytes, N, false)) {
      if (N > SS_THRESHOLD && lane == 0) atomicOr(B.err, ERRBIT_CAPACITY);   // (cannot happen: the range is sized for N <= gm_capacity, vp_weight_scratch_bytes)
    } else {
      ss_correct_tie_order<1>([&](int e) { return s.keys[e]; }, [&](int r) { return (int)(unsigned short)s.perm[r]; },
                              [&](int r, unsigned short e) { s.perm[r] = (int)e; }, [](unsigned *) {}, N, N, ss, lane, [&]() { wave_sync(); });
    }
  }
  if (permOut) {
    for (int r = lane; r < N; r += 64) permOut[r] = (unsigned short)s.perm[r];
  } else {
    for (int r = lane; r < N; r += 64) {
      const int m = s.perm[r];
      for (int pl = 0; pl < P3_COUNT; pl++) (dl + ((size_t)i * P3_COUNT + pl) * cap)[r] = (sl + ((size_t)i * P3_COUNT + pl) * cap)[m];
    }
  }
  // evaluation points
  DBG_T(16, 1);
  RFS_CUT(110);
  int nE = 0;
  {
    const int limit = nEvalPoints < RFSGPU_MAX_EVAL ? nEvalPoints : RFSGPU_MAX_EVAL;
    bool done = false;
    // 16 ranks at a time (r02: 64): the selection stops at the first weight below the threshold or with `limit` points, i.e.
    // within the first few ranks, and every rank looked at costs its probabilityOfDetection (Pd of the landmark and of its
    // shifted copies, vp_pd_wave); the pooled evaluation of a 16-rank group is one round of the wave.  Same points, same order.
    // With the map update's Pd cache (the fused step) only Gaussians the update has created need an evaluation -- the top ranks,
    // a dozen: one group of 64 ranks, one round.
    const int G = pdCache ? 64 : 16;
    for (int c0 = 0; c0 < N && !done; c0 += G) {
      const int r = (lane < G) ? c0 + lane : N;
      bool below = true, cand = false;
      Ent3 e;
      double pd = 0;
      e.w = 0; e.x = 10; e.y = 10; e.d = 1; e.xx = 1; e.xy = 0; e.xd = 0; e.yy = 1; e.yd = 0; e.dd = 1;
      bool cached = false;
      if (r < N) {
        const int m = s.perm[r];
        below = s.keys[m] < P.evalMinW;
        load_ent3(sl, cap, i, m, e, false);
        if (pdCache && m < nCached) {       // a landmark the update left where it was: same pose, same mean, same covariance, same scan
          const int idx = pdCache[m];
          if (idx != 254) {
            pd = (idx == 255) ? 0.0 : P.PdTable[idx];
            cached = true;
          }
        }
      }
      {
        bool close;
        const bool want = (r < N) && !below;
        if (__ballot(want && !cached) != 0ull) {
          const double v = vp_pd_wave(P, sScan, B.nScan, px, py, pth, e, want && !cached, close, pdScratch);
          if (!cached) pd = v;
        }
        if (!want) pd = 0;
        cand = pd > 0;
      }
      const unsigned long long belowMask = __ballot(below && (lane < G) && (c0 + lane < N));   // (ranks that exist)
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
  DBG_T(16, 2);
  RFS_CUT(111);
  double sumPrev = 0.0, sumCur = 0.0;
  for (int m = lane; m < N; m += 64) { sumPrev += qWP[m]; sumCur += s.keys[m]; }
  sumPrev = wave_sum_dpp(sumPrev);
  sumCur = wave_sum_dpp(sumCur);
  double prodBefore = 1.0, prodAfter = 1.0;
  // (eight evaluation points per sweep of the mixture -- r02: four: every sweep inverts each Gaussian's covariance again)
  for (int e0 = 0; e0 < nE; e0 += 8) {
    double accB[8], accA[8];
#pragma unroll
    for (int t = 0; t < 8; t++) { accB[t] = 0.0; accA[t] = 0.0; }
    for (int m = lane; m < N; m += 64) {
      Ent3 g;
      load_ent3(sl, cap, i, m, g, false);
      const double w = s.keys[m], wp = qWP[m];
      double Sm[9], Si[9];
      full3(g, Sm);
      inv3(Sm, Si);
      // (exp(.) * (1 / factor) instead of exp(.) / factor, as in the 2-D kernel: one division per Gaussian instead of one per
      //  Gaussian and evaluation point; <= 1.5 ulp in a term of the intensity sum -- DESIGN, deviation 4)
      const double rfactor = 1.0 / sqrt(P.twoPiPowD * det3(Sm));
#pragma unroll
      for (int t = 0; t < 8; t++) {
        const int ev = (e0 + t < nE) ? e0 + t : e0;
        const double lik = gauss_from_md2_r(md2_3(Si, s.evX[ev] - g.x, s.evY[ev] - g.y, evD[16 * ev] - g.d), rfactor);
        accB[t] += wp * lik;
        accA[t] += w * lik;
      }
    }
#pragma unroll
    for (int t = 0; t < 8; t++)
      if (e0 + t < nE) {
        prodBefore *= (RFS_DENORM_MIN + wave_sum_dpp(accB[t]));
        prodAfter *= (RFS_DENORM_MIN + wave_sum_dpp(accA[t]));
      }
  }
  // likelihood table
  DBG_T(16, 3);
  RFS_CUT(112);
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
  DBG_T(16, 4);
  RFS_CUT(113);
  const double l = rfs_partitions_wave(s, nE, nZ, P.vpClutter, lane, i, Q, B.err, P.exactPartitions);
  DBG_T(16, 5);
  RFS_CUT(114);
  const double ml = l / P.vpExpClutter;  // clutterIntensityIntegral (:287-290)
  const double overall = ml * prodBefore / prodAfter * exp(sumCur - sumPrev);
  if (lane == 0) B.weight[i] = overall * B.weight[i];
}
template <int WPB>
__global__ __launch_bounds__(WPB * 64) void vp_weighting_kernel(Buffers B, Params P, int src, int dst, int nZ, int evalCap, MurtyQueue Q) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  double *sZ = reinterpret_cast<double *>(smem_raw);
  double *sScan = sZ + 3 * nZ;
  const int wave = threadIdx.x >> 6;
  const int lane = threadIdx.x & 63;
  for (int t = threadIdx.x; t < 3 * nZ; t += WPB * 64) sZ[t] = B.Z[t];
  for (int t = threadIdx.x; t < B.nScan; t += WPB * 64) sScan[t] = B.scan[t];
  __syncthreads();
  const int i = __builtin_amdgcn_readfirstlane(blockIdx.x * WPB + wave);
  if (i >= B.N) return;
  vp_weight_particle(B, P, src, dst, nZ, evalCap, Q, i, lane, sZ, sScan,
                     smem_raw + vp_shared_lds_bytes(nZ, B.nScan) + (size_t)wave * vp_weight_lds_bytes_per_wave(B.cap, evalCap, nZ), nullptr);
}

// LDS per wave for the merge: 42 B per Gaussian -- position (x, y, d), prefilter bound and weight as doubles, a u16 survivor list.
// (Until r02 every entry was staged with its covariance AND the inverse, 130 B: 25 KB per wave at cap 192, six waves per CU, and
// the 5000 particles of configs[3] took 3.3 rounds.  Now the distance prefilter below runs on the LDS copy and everything the
// exact test needs comes from the slab for the few pairs that survive it; all 5000 waves are resident at once.)
// Two layouts share the block: the row-by-row scan's (mixtures longer than a wavefront: 42 B per Gaussian of capacity) and the
// lane-parallel scan's (at most 64 Gaussians: the five arrays at stride 64, the survivor list, the six covariance planes and an
// fp32 copy of position + prefilter radius for the candidate sweeps: 6.8 KB whatever the capacity).
#define VP_MERGE_LANE_LDS_BYTES ((5 * 64 + 16 + 6 * 64) * 8 + 4 * 64 * 4 + 16)
__host__ __device__ inline size_t vp_merge_lds_bytes_per_wave(int cap) {
  size_t a = (size_t)cap * (5 * 8 + 2 + 2);    // five fp64 arrays, the survivor list and the prune's order (u16 each)
  if (a < (size_t)VP_MERGE_LANE_LDS_BYTES) a = VP_MERGE_LANE_LDS_BYTES;
  return (a + 15) & ~(size_t)15;
}

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

// sqrt of a prefilter bound as an fp32 number that is not smaller (2^-20 covers the conversion's rounding); a NaN bound -- the
// fp64 comparison `!(d2 > bound)` then keeps everything -- becomes +inf.
__device__ __forceinline__ float vp_radius_f32(double bound) {
  if (!(bound == bound)) return __builtin_huge_valf();
  return (float)(sqrt(bound) * (1.0 + 0x1p-20));
}
// One merge of GaussianMixture::merge (include/GaussianMixture.hpp:443-468) on the running state of row a: b is absorbed.
// State: position, weight, full covariance aS (upper triangle mirrored), its inverse aI as the test reads it, prefilter bound ab.
__device__ __forceinline__ void merge3_step(double t2, double f, const Ent3 &eb, double w2, double &ax, double &ay, double &ad, double &aw, double aS[9],
                                            double aI[9], double &ab) {
  const double w1 = aw;
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
  Ent3 em;
  em.w = aw; em.x = ax; em.y = ay; em.d = ad;
  em.xx = aS[0]; em.xy = aS[1]; em.xd = aS[2]; em.yy = aS[4]; em.yd = aS[5]; em.dd = aS[8];
  ab = merge_bound3(t2, em);
}

// GaussianMixture::merge for 3-D Gaussians: the exact sequential-greedy scan (lanes test 64 candidates j at once against
// the current state of a, lowest passing lane merged, lanes above it re-tested), optional fused prune.  Merged rows are
// written back in place (a row is never read again once the scan has passed it); the prune reads the slab.
// `perm` (the fused step): row r of the scan is entry perm[r] of the slab -- the mixture in weight order without a sorted copy;
// nullptr: the slab itself is in scan order (the stand-alone kernel after vp_weighting_kernel's sorted copy).
#ifndef VP_LANE_SCAN
#define VP_LANE_SCAN 1   // tuning knob (tools/variant_bench.py): 0 = the row-by-row scan for every mixture
#endif
template <bool FUSE_PRUNE>
__device__ void vp_merge_particle(const Buffers &B, const Params &P, int cur, int dst, int i, int lane, unsigned char *wmem, const unsigned short *perm) {
  const int cap = B.cap;
  double *sb = reinterpret_cast<double *>(wmem);
  auto at = [&](int r) -> int { return perm ? (int)perm[r] : r; };   // storage index of scan row r
  const int N = B.count[i];
  const bool laneMode = VP_LANE_SCAN && N <= 64;
  const size_t st = laneMode ? 64 : (size_t)cap;       // stride of the per-Gaussian arrays (see vp_merge_lds_bytes_per_wave)
  double *sX = sb, *sY = sb + st, *sD = sb + 2 * st, *sBnd = sb + 3 * st, *sW = sb + 4 * st;
  unsigned short *sIdx = reinterpret_cast<unsigned short *>(sb + 5 * st);
  double *sCov = sb + 5 * 64 + 16;                      // [6][64]                      (lane mode only)
  float *fX = reinterpret_cast<float *>(sCov + 6 * 64), *fY = fX + 64, *fD = fY + 64, *fR = fD + 64;   // fp32 position, prefilter radius
  double *slab = B.slab[cur];
  const double t2 = P.mergeT2, f = P.mergeInfl;
  unsigned hole = 0;
  double posMax = 0.0;     // largest |coordinate| of the mixture (lane mode: error bound of the fp32 sweep)
  DBG_T(32, 0);
  for (int m = lane, sidx = 0; m < N; m += 64, sidx++) {
    Ent3 e;
    load_ent3(slab, cap, i, at(m), e, true);
    sX[m] = e.x; sY[m] = e.y; sD[m] = e.d;
    sBnd[m] = merge_bound3(t2, e);
    sW[m] = e.w;
    if (e.w < 0) hole |= 1u << sidx;
    if (laneMode) {
      sCov[m] = e.xx; sCov[64 + m] = e.xy; sCov[128 + m] = e.xd; sCov[192 + m] = e.yy; sCov[256 + m] = e.yd; sCov[320 + m] = e.dd;
      fX[m] = (float)e.x; fY[m] = (float)e.y; fD[m] = (float)e.d;
      fR[m] = vp_radius_f32(sBnd[m]);
      posMax = fmax(posMax, fmax(fmax(fabs(e.x), fabs(e.y)), fabs(e.d)));
    }
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
  DBG_T(32, 1);
  RFS_CUT(120);
  if (N > 64 || !VP_LANE_SCAN)      // (mixtures of at most one wavefront take the lane-parallel scan below, which needs no row list)
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
  DBG_T(32, 2);
  // the reference's scan of ONE row, by the whole wave: 64 candidates at a time against the current state of a
  auto serial_row = [&](const int a) {
    Ent3 ea;
    ea.w = 0;
    load_ent3(slab, cap, i, at(a), ea, false);                       // (wave-uniform address: one broadcast load per plane)
    const unsigned ownerHole = (unsigned)__builtin_amdgcn_readlane((int)hole, a & 63);
    if ((ownerHole >> (a >> 6)) & 1u) return;
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
              load_ent3(slab, cap, i, at(j), ej, false);
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
        load_ent3(slab, cap, i, at(jj), eb, false);                  // uniform
        merge3_step(t2, f, eb, sW[jj], ax, ay, ad, aw, aS, aI, ab);
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
        const int sa = at(a);
        plane3(slab, cap, i, P3_W)[sa] = aw;
        plane3(slab, cap, i, P3_MX)[sa] = ax; plane3(slab, cap, i, P3_MY)[sa] = ay; plane3(slab, cap, i, P3_MD)[sa] = ad;
        plane3(slab, cap, i, P3_SXX)[sa] = aS[0]; plane3(slab, cap, i, P3_SXY)[sa] = aS[1]; plane3(slab, cap, i, P3_SXD)[sa] = aS[2];
        plane3(slab, cap, i, P3_SYY)[sa] = aS[4]; plane3(slab, cap, i, P3_SYD)[sa] = aS[5]; plane3(slab, cap, i, P3_SDD)[sa] = aS[8];
      }
    }
  };
  if (N <= 64 && VP_LANE_SCAN) {
    // ---- lane-parallel scan (r3).  Row a's scan depends on the rest of the mixture in ONE way only: which of the entries
    // above it earlier rows have already absorbed -- entries above a are in their initial state when the reference reaches a.
    // So every lane runs its own row's scan against the initial mixture (j ascending, state updated at each merge: the very
    // sequence of the reference), and the rows are then validated in order: a row whose absorbed set avoids everything
    // absorbed before it did exactly what the reference does and is committed; a row that collides (two rows after the same
    // entry) is redone by the whole wave against the true state (serial_row).  On a Victoria Park map the clusters -- a tree,
    // its updated copies -- are disjoint, so every row commits; the serial form cost one wave-wide pass per row with a
    // partner (40 k of a particle's 165 k cycles at configs[3], and 14 k more for the row list).
    const int a = lane;
    const unsigned long long holes0 = __ballot((hole & 1u) != 0);
    const bool isRow = (a < N) && !((holes0 >> a) & 1ull);
    // (every entry the scan touches comes from the LDS copy made above -- the same doubles the slab holds: a merge inside the
    //  candidate loop that waits for ten dependent global loads is what made the first lane-parallel form as slow as the serial one)
    auto staged = [&](int m, Ent3 &e) {
      e.x = sX[m]; e.y = sY[m]; e.d = sD[m];
      e.xx = sCov[m]; e.xy = sCov[64 + m]; e.xd = sCov[128 + m]; e.yy = sCov[192 + m]; e.yd = sCov[256 + m]; e.dd = sCov[320 + m];
    };
    Ent3 ea;
    ea.w = 0; ea.x = 10; ea.y = 10; ea.d = 1; ea.xx = 1; ea.xy = 0; ea.xd = 0; ea.yy = 1; ea.yd = 0; ea.dd = 1;
    if (isRow) staged(a, ea);
    double ax = ea.x, ay = ea.y, ad = ea.d, aw = isRow ? sW[a] : 0.0, ab = isRow ? sBnd[a] : 0.0;
    double aS[9], aI[9];
    full3(ea, aS);
    inverse_of(ea, aI);
    unsigned long long absorbed = 0ull;
    // Candidates by bit mask: a uniform sweep over j (LDS broadcasts, no divergence, unrolled) marks the entries above the row
    // that pass the distance prefilter against the row's CURRENT state; the lowest marked entry then goes through the exact test
    // (all rows that have one, at once) and, if it merges, the row's mask above it is rebuilt from the new state by another sweep.
    // The sweep runs in fp32 on the staged copies and keeps a SUPERSET of what the fp64 prefilter |e|^2 <= max(bound_a, bound_j)
    // keeps (whatever it lets through is decided by the exact fp64 test anyway): with u = 2^-24, every fp32 difference is
    // within eta = 2^-22 max|coordinate| of the true one, so |e| <= sqrt(B) implies |e_f| <= sqrt(B) + 2 eta; the radii are
    // rounded up, the product carries 2^-18 for its own roundings; NaN / inf (positions, bounds) compare as "keep".
    const float etaAll = wave_max_f32((float)(posMax * (1.0 + 0x1p-20))) * (2.0f * 0x1p-22f * (1.0f + 0x1p-20f));
    // Two entries per packed instruction (v_pk_*: the same fp32 operations with the same roundings as one by one), the verdicts shifted
    // into two 32-bit words from the top entry down -- bit k of the pair of words <-> entry 2 (jStart / 2) + k --, `j > from`, `j < N`
    // and the holes applied to the finished mask.  (r3: one entry per trip, the bit set through a 64-bit select: 16 instructions
    // per entry against 9.)
    typedef float vpf2_t __attribute__((ext_vector_type(2)));
    auto sweep = [&](const int from, const bool want, const int jStart) -> unsigned long long {   // entries j > from (>= jStart: uniform, <= from + 1) passing the prefilter
      const float fax = (float)ax, fay = (float)ay, fad = (float)ad;
      const float ra = vp_radius_f32(ab);
      const float eta2 = fmaxf(etaAll, fmaxf(fmaxf(fabsf(fax), fabsf(fay)), fabsf(fad)) * (2.0f * 0x1p-22f * (1.0f + 0x1p-20f)));
      const vpf2_t vx = {fax, fax}, vy = {fay, fay}, vd = {fad, fad}, ve = {eta2, eta2}, vk = {1.0f + 0x1p-18f, 1.0f + 0x1p-18f};
      const vpf2_t *pX = reinterpret_cast<const vpf2_t *>(fX), *pY = reinterpret_cast<const vpf2_t *>(fY);
      const vpf2_t *pD = reinterpret_cast<const vpf2_t *>(fD), *pR = reinterpret_cast<const vpf2_t *>(fR);
      auto pair_bits = [&](const int q) -> unsigned {   // entries 2q (bit 0) and 2q + 1 (bit 1)
        const vpf2_t e0 = pX[q] - vx, e1 = pY[q] - vy, e2 = pD[q] - vd;
        vpf2_t R = pR[q];
        R.x = fmaxf(ra, R.x); R.y = fmaxf(ra, R.y);
        R = R + ve;
        const vpf2_t d2 = __builtin_elementwise_fma(e0, e0, __builtin_elementwise_fma(e1, e1, e2 * e2));
        const vpf2_t thr = (R * R) * vk;
        return (!(d2.x > thr.x) ? 1u : 0u) | (!(d2.y > thr.y) ? 2u : 0u);
      };
      const int q0 = jStart >> 1, qEnd = (N + 1) >> 1, qMid = (q0 + 16 < qEnd) ? q0 + 16 : qEnd;
      unsigned lo = 0u, hi = 0u;
#pragma unroll 2
      for (int q = qEnd - 1; q >= qMid; q--) hi = (hi << 2) | pair_bits(q);
#pragma unroll 2
      for (int q = qMid - 1; q >= q0; q--) lo = (lo << 2) | pair_bits(q);
      unsigned long long m = (((unsigned long long)hi << 32) | lo) << (2 * q0);
      m &= ~((2ull << (from & 63)) - 1ull);                                       // j > from
      m &= (N >= 64) ? ~0ull : ((1ull << N) - 1ull);                              // j < N (the last pair may reach one past the mixture)
      return want ? (m & ~holes0) : 0ull;
    };
    unsigned long long cand = sweep(a, isRow, 1);
    // FIND, then MERGE TOGETHER: every row first walks its own marked entries (ascending) through the exact test until one
    // passes -- cheap, and rows differ in how many they reject; then all rows that found one merge at once and rebuild their
    // masks at once.  A merge is a dozen fp64 divisions and a sweep is a pass over the mixture: paying them once per ROUND
    // (the largest number of merges any row makes: one or two) instead of once per row or once per absorbed entry is the point.
    while (true) {
      int jm = -1;
      while (__ballot(cand != 0ull && jm < 0) != 0ull) {
        if (cand != 0ull && jm < 0) {
          const int j = __builtin_ctzll(cand);
          cand &= cand - 1ull;
          const double e0 = sX[j] - ax, e1 = sY[j] - ay, e2 = sD[j] - ad;
          bool far = md2_3(aI, e0, e1, e2) > t2;
          if (far) {
            Ent3 ej;
            staged(j, ej);
            double jI[9];
            inverse_of(ej, jI);
            far = md2_3(jI, -e0, -e1, -e2) > t2;
          }
          if (!far && ((aw + sW[j]) != 0.0)) jm = j;
        }
      }
      if (__ballot(jm >= 0) == 0ull) break;
      if (jm >= 0) {
        Ent3 eb;
        staged(jm, eb);
        merge3_step(t2, f, eb, sW[jm], ax, ay, ad, aw, aS, aI, ab);
        absorbed |= 1ull << jm;
      }
      // (wave-wide; rows that found nothing are finished: cand == 0.  Only entries above the LOWEST entry absorbed in this round
      //  can matter to anyone -- the mixture is in weight order and a fresh Gaussian's partner is the faded landmark it came
      //  from, near the end of the order: the rebuild usually looks at a handful of entries, not at all of them.)
      const int jLow = (int)wave_min_u32((jm >= 0) ? (unsigned)jm : 0xffffu) + 1;
      const unsigned long long fresh = sweep(jm, jm >= 0, jLow);
      if (jm >= 0) cand = fresh;
    }
    // ordered validation
    DBG_T(32, 5);
#ifdef RFS_PROFILE
    int dbgConf = 0, dbgRows = 0, dbgAbs = 0;
#endif
    unsigned long long holes = holes0, commit = 0ull;
    for (unsigned long long rm = __ballot(absorbed != 0ull); rm; rm &= rm - 1ull) {
      const int r = __builtin_ctzll(rm);
      if ((holes >> r) & 1ull) continue;                       // an earlier row absorbed r itself
      const unsigned long long m = readlane_u64(absorbed, r);
#ifdef RFS_PROFILE
      dbgRows++; dbgAbs += __popcll(m); if ((m & holes) != 0ull) dbgConf++;
#endif
      if ((m & holes) == 0ull) {
        holes |= m;
        commit |= 1ull << r;
      } else {                                                 // collision: the true scan of r skips entries that are gone
        hole = (hole & ~1u) | (unsigned)((holes >> lane) & 1ull);
        serial_row(r);
        holes = __ballot((hole & 1u) != 0);
      }
    }
    DBG_T(32, 6);
#ifdef RFS_PROFILE
    if (B.dbg && i == 7 && lane == 0) { B.dbg[40] = dbgRows; B.dbg[41] = dbgConf; B.dbg[42] = dbgAbs; B.dbg[43] = N; }
#endif
    hole = (hole & ~1u) | (unsigned)((holes >> lane) & 1ull);
    if ((commit >> lane) & 1ull) {
      const int sa = at(a);
      sW[a] = aw;
      plane3(slab, cap, i, P3_W)[sa] = aw;
      plane3(slab, cap, i, P3_MX)[sa] = ax; plane3(slab, cap, i, P3_MY)[sa] = ay; plane3(slab, cap, i, P3_MD)[sa] = ad;
      plane3(slab, cap, i, P3_SXX)[sa] = aS[0]; plane3(slab, cap, i, P3_SXY)[sa] = aS[1]; plane3(slab, cap, i, P3_SXD)[sa] = aS[2];
      plane3(slab, cap, i, P3_SYY)[sa] = aS[4]; plane3(slab, cap, i, P3_SYD)[sa] = aS[5]; plane3(slab, cap, i, P3_SDD)[sa] = aS[8];
    }
    if (commit != 0ull) anyMerge = true;
  } else {
    for (int r0 = 0; r0 < N; r0 += 64)
      for (unsigned long long rows = __ballot((rowFlag >> (r0 >> 6)) & 1u); rows; rows &= rows - 1ull) serial_row(r0 + __builtin_ctzll(rows));
  }
  // in-place updates of merged rows (global memory) -> visible to the wave's other lanes
  DBG_T(32, 3);
  RFS_CUT(121);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  wave_sync();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  if (!FUSE_PRUNE) {
    if (!anyMerge) return;
    for (int m = lane, sidx = 0; m < N; m += 64, sidx++)
      if ((hole >> sidx) & 1u) plane3(slab, cap, i, P3_W)[at(m)] = -1.0;
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
  // ranks among the survivors (ties by index) -> sOrder, equal weights into std::sort's order (stdsort_replay.h; the sort of
  // GaussianMixture::prune runs over the whole list, merged-away entries -- weight 0 -- included), then the compaction
  unsigned short *sOrder = sIdx + st;                       // [nSurv]   (room: see vp_merge_lds_bytes_per_wave)
  for (int q = lane; q < nSurv; q += 64) {
    const int m = sIdx[q];
    const double wm = sW[m];
    int rank = 0;
    for (int q2 = 0; q2 < nSurv; q2++) {
      const int j2 = sIdx[q2];
      const double wj = sW[j2];
      rank += ((wj > wm) | ((wj == wm) & (j2 < m))) ? 1 : 0;
    }
    sOrder[rank] = (unsigned short)m;
  }
  wave_sync();
  {
    StdSortScratch ss;                                      // positions and bounds of the merge are dead: 4 x st doubles
    auto key0 = [&](int e) { const double w = sW[e]; return w < 0.0 ? 0.0 : w; };
    auto is_rest = [&](int e) { const double w = sW[e]; return !((w >= t) && (w >= 0.0)); };
    auto rest_group = [&](unsigned *T) {
      for (int e = lane; e < N; e += 64) {
        if (!is_rest(e)) continue;
        const double k = key0(e);
        int g = nSurv;
        for (int e2 = 0; e2 < N; e2++) g += (is_rest(e2) && key0(e2) > k) ? 1 : 0;
        T[e] = ((unsigned)g << 16) | (unsigned)g;
      }
    };
    if (ss_carve(ss, reinterpret_cast<unsigned char *>(sb), 4 * st * sizeof(double), N, true))
      ss_correct_tie_order<1>(key0, [&](int r) { return (int)sOrder[r]; }, [&](int r, unsigned short e) { sOrder[r] = e; }, rest_group, N, nSurv, ss, lane,
                              [&]() { wave_sync(); });
    else if (N > SS_THRESHOLD && lane == 0) atomicOr(B.err, ERRBIT_CAPACITY);   // (cannot happen: 32 st bytes against ~6.2 N + 124 with N <= st)
  }
  for (int rank = lane; rank < nSurv; rank += 64) {
    const int m = sOrder[rank];
    const double wm = sW[m];
    Ent3 e;
    load_ent3(slab, cap, i, at(m), e, false);
    plane3(dl, cap, i, P3_W)[rank] = wm;
    plane3(dl, cap, i, P3_WP)[rank] = 0.0;
    plane3(dl, cap, i, P3_MX)[rank] = e.x; plane3(dl, cap, i, P3_MY)[rank] = e.y; plane3(dl, cap, i, P3_MD)[rank] = e.d;
    plane3(dl, cap, i, P3_SXX)[rank] = e.xx; plane3(dl, cap, i, P3_SXY)[rank] = e.xy; plane3(dl, cap, i, P3_SXD)[rank] = e.xd;
    plane3(dl, cap, i, P3_SYY)[rank] = e.yy; plane3(dl, cap, i, P3_SYD)[rank] = e.yd; plane3(dl, cap, i, P3_SDD)[rank] = e.dd;
  }
  if (lane == 0) B.count[i] = nSurv;
  DBG_T(32, 4);
}
template <int WPB, bool FUSE_PRUNE>
__global__ __launch_bounds__(WPB * 64) void vp_merge_kernel(Buffers B, Params P, int cur, int dst) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const int wave = threadIdx.x >> 6;
  const int lane = threadIdx.x & 63;
  const int i = __builtin_amdgcn_readfirstlane(blockIdx.x * WPB + wave);
  if (i >= B.N) return;
  vp_merge_particle<FUSE_PRUNE>(B, P, cur, dst, i, lane, smem_raw + (size_t)wave * vp_merge_lds_bytes_per_wave(B.cap), nullptr);
}

// ---- the whole Victoria Park step in ONE launch (RBPHDFilter::update body, include/RBPHDFilter.hpp:469-520) -------------------
// One wavefront owns a particle and takes it through updateMap -> importanceWeighting -> merge + prune; WPB particles share a
// workgroup (and the LDS copies of the measurement set and the laser scan).  As three kernels every phase re-read the 11-plane
// slab from HBM and the weighting wrote a sorted copy of it that the merge read back (r02: 138 MB of traffic per update at
// 5000 particles x 40-50 Gaussians against 57.8 MB algorithmic); here the map update's output is still in cache when the
// weighting reads it, the sorted order is a u16 permutation in LDS (as in the 2-D fused step) and the mixture is written once
// more only by the prune's compaction into the other slab.  The phases are the device functions of the stand-alone kernels:
// same arithmetic in the same order, bit-identical results.  useWeighting == 0: SC-PHD (the weight comes out of the map update,
// the mixture is not sorted).
__host__ __device__ inline size_t vp_step_lds_bytes_per_wave(int cap, int evalCap, int nZ) {
  size_t a = vp_update_lds_bytes_per_wave(cap);
  const size_t b = vp_weight_lds_bytes_per_wave(cap, evalCap, nZ), c = vp_merge_lds_bytes_per_wave(cap);
  if (b > a) a = b;
  if (c > a) a = c;
  return ((a + 15) & ~(size_t)15) + (((size_t)cap * 2 + 15) & ~(size_t)15) + (((size_t)cap + 15) & ~(size_t)15);   // + the permutation + the Pd cache
}
#ifndef VP_STEP_WAVES_PER_EU
#define VP_STEP_WAVES_PER_EU 0   // 0: the compiler's own register count (132 VGPRs with -disable-machine-licm: three waves per SIMD)
#endif
template <int WPB>
__global__ __launch_bounds__(WPB * 64)
#if VP_STEP_WAVES_PER_EU > 0
__attribute__((amdgpu_waves_per_eu(VP_STEP_WAVES_PER_EU)))
#endif
void vp_step_fused_kernel(Buffers B, Params P, int cur, int nZ, int evalCap, int useWeighting, MurtyQueue Q, ZArg zarg, float *stepCost, const int *stepOrder) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  double *sZ = reinterpret_cast<double *>(smem_raw);
  double *sScan = sZ + 3 * nZ;
  const int wave = threadIdx.x >> 6;
  const int lane = threadIdx.x & 63;
  for (int t = threadIdx.x; t < 3 * nZ; t += WPB * 64) sZ[t] = zarg.v[t];     // the measurement set rides in the kernel arguments
  for (int t = threadIdx.x; t < B.nScan; t += WPB * 64) sScan[t] = B.scan[t];
  __syncthreads();
  // stepOrder (may be null): launch slot -> particle.  A launch of 5000 one-wave particles on ~3000 wave slots ends with the waves that
  // started last; with the particles that took longest in the PREVIOUS step first, the tail is made of short ones (round 6; stepCost
  // receives this step's duration per particle, 100 MHz ticks).  Which particle a slot works on changes nothing in any result.
  const int slot = __builtin_amdgcn_readfirstlane(blockIdx.x * WPB + wave);
  if (slot >= B.N) return;
  const int i = stepOrder ? __builtin_amdgcn_readfirstlane(stepOrder[slot]) : slot;
  const long long tStart = stepCost ? (long long)wall_clock64() : 0ll;
  const size_t per = vp_step_lds_bytes_per_wave(B.cap, evalCap, nZ);
  unsigned char *wmem = smem_raw + vp_shared_lds_bytes(nZ, B.nScan) + (size_t)wave * per;
  unsigned char *sPdIdx = wmem + per - (((size_t)B.cap + 15) & ~(size_t)15);
  unsigned short *sPerm = reinterpret_cast<unsigned short *>(sPdIdx - (((size_t)B.cap * 2 + 15) & ~(size_t)15));
#ifdef RFS_PROFILE
  long long *fd = B.dbg ? B.dbg + 64 + 4 * (size_t)B.N + 4 * (size_t)i : nullptr;   // per particle: start | after update | after weighting | end (100 MHz ticks)
  if (fd && lane == 0) {
    const unsigned hw = __builtin_amdgcn_s_getreg((16 - 1) << 11 | 4), xcc = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 20);
    fd[0] = (long long)((wall_clock64() & 0xfffffffffffull) | ((unsigned long long)(hw & 0xffffu) << 44) | ((unsigned long long)(xcc & 0xfu) << 60));
  }
#endif
  const int nBefore = vp_update_map_particle(B, P, cur, nZ, i, lane, sZ, sScan, wmem, sPdIdx);
#ifdef RFS_PROFILE
  if (fd && lane == 0) fd[1] = (long long)wall_clock64();
#endif
  // (one wave: the slab rows it wrote are its own; order the global writes before the reads of the next phase)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  wave_sync();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  const unsigned short *perm = nullptr;
  if (useWeighting) {
    vp_weight_particle(B, P, cur, cur ^ 1, nZ, evalCap, Q, i, lane, sZ, sScan, wmem, sPerm, sPdIdx, nBefore);
    wave_sync();
    perm = sPerm;
  }
#ifdef RFS_PROFILE
  if (fd && lane == 0) fd[2] = (long long)wall_clock64();
#endif
  vp_merge_particle<true>(B, P, cur, cur ^ 1, i, lane, wmem, perm);
  if (stepCost && lane == 0) stepCost[i] = (float)((long long)wall_clock64() - tStart);
#ifdef RFS_PROFILE
  if (fd && lane == 0) fd[3] = (long long)wall_clock64();
#endif
}

