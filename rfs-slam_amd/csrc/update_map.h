// update_map.h -- phd_update_map: RBPHDFilter::updateMap (reference include/RBPHDFilter.hpp:543-725) with
// KalmanFilter::correct (include/KalmanFilter.hpp:261-342), KalmanFilter_RngBrg::calculateInnovation
// (src/KalmanFilter_RngBrg.cpp:52-65) and MeasurementModel_RngBrg::{measure, probabilityOfDetection}
// fused into one kernel.
//
// One 64-lane wavefront per particle; lanes stride over that particle's landmarks (pass*64 + lane).
// The measurement set is staged once per block in LDS.  The nM x nZ weight table of the reference never
// exists: pass 1 keeps, per landmark, a 64-bit mask of the measurements that survive the gates (LDS) and
// accumulates the per-measurement normalisers in the reference's summation order (clutter, then landmarks in
// index order: ballot + readlane in lane order); pass 2 recomputes the few surviving pairs, normalises,
// appends the new Gaussians in (m,z) row-major order through a wave prefix sum, and writes the
// missed-detection weights.  HBM traffic per particle = one read of the mixture per pass + the appended
// records + the weight planes.
#pragma once
#include "common.h"

// Landmark-level quantities of KalmanFilter::correct that are shared by all measurements.
struct LmKF {
  double zx0, zx1;            // expected measurement
  double i00, i01, i10, i11;  // S^-1
  double factor;              // sqrt((2pi)^2 |S|)
  double k00, k01, k10, k11;  // Kalman gain
  double p00, p01, p11;       // updated covariance (symmetrised)
  bool ok;                    // measure() returned true
};

__device__ __forceinline__ void lm_precompute(const Params &P, const PoseReg &pr, double mx, double my, double sxx, double sxy, double syy,
                                              LmKF &k, double &range) {
  MeasOut mo;
  rb_measure(P, pr, mx, my, sxx, sxy, syy, mo);
  range = mo.range;
  k.ok = mo.inRange;
  k.zx0 = mo.z0;
  k.zx1 = mo.z1;
  double det;
  inv2(mo.s00, mo.s01, mo.s10, mo.s11, k.i00, k.i01, k.i10, k.i11, det);
  k.factor = pdf_factor2(det);
  // K = (P * H^T) * S^-1
  double t00 = sxx * mo.h00 + sxy * mo.h01, t01 = sxx * mo.h10 + sxy * mo.h11;
  double t10 = sxy * mo.h00 + syy * mo.h01, t11 = sxy * mo.h10 + syy * mo.h11;
  k.k00 = t00 * k.i00 + t01 * k.i10;
  k.k01 = t00 * k.i01 + t01 * k.i11;
  k.k10 = t10 * k.i00 + t11 * k.i10;
  k.k11 = t10 * k.i01 + t11 * k.i11;
  // P+ = (I - K H) P, then (P+ + P+^T)/2
  double kh00 = k.k00 * mo.h00 + k.k01 * mo.h10, kh01 = k.k00 * mo.h01 + k.k01 * mo.h11;
  double kh10 = k.k10 * mo.h00 + k.k11 * mo.h10, kh11 = k.k10 * mo.h01 + k.k11 * mo.h11;
  double a00 = 1.0 - kh00, a01 = 0.0 - kh01, a10 = 0.0 - kh10, a11 = 1.0 - kh11;
  double q00 = a00 * sxx + a01 * sxy, q01 = a00 * sxy + a01 * syy;
  double q10 = a10 * sxx + a11 * sxy, q11 = a10 * sxy + a11 * syy;
  k.p00 = (q00 + q00) / 2;
  k.p01 = (q01 + q10) / 2;
  k.p11 = (q11 + q11) / 2;
}

// One (landmark, measurement) pair: returns Pd*w*likelihood, or 0 when any gate rejects it
// (include/KalmanFilter.hpp:311-338 + include/RBPHDFilter.hpp:622-632).  nu = wrapped innovation.
__device__ __forceinline__ double pair_weight(const Params &P, const LmKF &k, double pdw, double z0, double z1, double &nu0, double &nu1) {
  if (!k.ok) return 0.0;
  double e0 = z0 - k.zx0, e1 = z1 - k.zx1;
  if (P.kfRange > 0 && fabs(e0) > P.kfRange) return 0.0;
  double w1 = wrap_pi(e1);
  if (P.kfBearing > 0 && fabs(w1) > P.kfBearing) return 0.0;
  nu0 = e0;
  nu1 = w1;
  // likelihood uses the UNWRAPPED difference (KalmanFilter.hpp:317-320)
  double t0 = e0 * k.i00 + e1 * k.i10;
  double t1 = e0 * k.i01 + e1 * k.i11;
  double md2 = t0 * e0 + t1 * e1;
  if (md2 > P.newGaussMd2) return 0.0;
  double lik = gauss_from_md2(md2, k.factor);
  if (lik == 0.0) return 0.0;
  return pdw * lik;
}

template <int WPB>
__global__ __launch_bounds__(WPB * 64) void phd_update_map_kernel(Buffers B, Params P, int cur, int nZ) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  // LDS: Z[2*nZ] | per wave: colsum[64] | per wave: assoc[cap]
  double *sZ = reinterpret_cast<double *>(smem_raw);
  double *sCol = sZ + 2 * RFSGPU_MAX_Z;
  unsigned long long *sAssoc = reinterpret_cast<unsigned long long *>(sCol + WPB * RFSGPU_MAX_Z);

  const int wave = threadIdx.x >> 6;
  const int lane = threadIdx.x & 63;
  for (int t = threadIdx.x; t < 2 * nZ; t += WPB * 64) sZ[t] = B.Z[t];
  __syncthreads();

  const int i = __builtin_amdgcn_readfirstlane(blockIdx.x * WPB + wave);
  if (i >= B.N) return;
  double *colsum = sCol + wave * RFSGPU_MAX_Z;
  unsigned long long *assoc = sAssoc + (size_t)wave * B.cap;

  const int nM = B.count[i];
  const unsigned long long zmask = (nZ >= 64) ? ~0ull : ((1ull << nZ) - 1ull);
  if (nM == 0) {  // :559-564
    if (lane == 0) {
      B.unusedMask[i] = zmask;
      B.nInFov[i] = 0;
    }
    return;
  }
  double *slab = B.slab[cur];
  double *pW = plane(slab, B.cap, i, PL_W), *pWP = plane(slab, B.cap, i, PL_WP);
  double *pMX = plane(slab, B.cap, i, PL_MX), *pMY = plane(slab, B.cap, i, PL_MY);
  double *pSXX = plane(slab, B.cap, i, PL_SXX), *pSXY = plane(slab, B.cap, i, PL_SXY), *pSYY = plane(slab, B.cap, i, PL_SYY);

  DBG_T(0, 0);
  PoseReg pr;
  load_pose(B, P, i, pr);

  const int nPass = (nM + 63) >> 6;
  int nFov = 0;
  double wsum = 0.0;  // SC-PHD: sum of prior weights

  // ---------------- pass 1: gates, association masks, per-measurement normalisers ----------------
  for (int p = 0; p < nPass; p++) {
    const int m = p * 64 + lane;
    const bool act = m < nM;
    double w = 0, mx = 0, my = 0, sxx = 1, sxy = 0, syy = 1;
    if (act) { w = pW[m]; mx = pMX[m]; my = pMY[m]; sxx = pSXX[m]; sxy = pSXY[m]; syy = pSYY[m]; }
    LmKF k;
    double range;
    lm_precompute(P, pr, mx, my, sxx, sxy, syy, k, range);
    bool close;
    double pd = rb_pd(P, range, close);
    if (close) pd = 1;  // :604-606
    const bool fov = act && (pd != 0);
    const double pdw = pd * w;
    nFov += __popcll(__ballot(fov));
    if (P.useCluster) wsum += act ? w : 0.0;
    unsigned long long mymask = 0;
    for (int z = 0; z < nZ; z++) {
      double nu0, nu1;
      double v = fov ? pair_weight(P, k, pdw, sZ[2 * z], sZ[2 * z + 1], nu0, nu1) : 0.0;
      unsigned long long hit = __ballot(v != 0.0);
      if (v != 0.0) mymask |= (1ull << z);
      double cs = (p == 0) ? P.clutter : colsum[z];
      while (hit) {  // reference order: sum = clutter; for m: sum += W[m][z]
        int l = __builtin_ctzll(hit);
        hit &= hit - 1;
        cs += readlane_f64(v, l);
      }
      colsum[z] = cs;  // every lane stores the same value: plain per-thread store->load ordering, no cross-lane hazard
    }
    if (act) assoc[m] = mymask;
  }
  __builtin_amdgcn_wave_barrier();
  DBG_T(0, 1);

  // ---------------- pass 2: normalise, append new Gaussians, missed-detection weights ----------------
  int base = nM;
  unsigned long long used = 0;
  bool overflow = false;
  for (int p = 0; p < nPass; p++) {
    const int m = p * 64 + lane;
    const bool act = m < nM;
    double w = 0, mx = 0, my = 0, sxx = 1, sxy = 0, syy = 1;
    if (act) { w = pW[m]; mx = pMX[m]; my = pMY[m]; sxx = pSXX[m]; sxy = pSXY[m]; syy = pSYY[m]; }
    LmKF k;
    double range;
    lm_precompute(P, pr, mx, my, sxx, sxy, syy, k, range);
    bool close;
    double pd = rb_pd(P, range, close);
    if (close) pd = 1;
    const double pdw = pd * w;
    unsigned long long mymask = act ? assoc[m] : 0ull;
    // loop A: count survivors (normalised weight > 0), row sum, used flags
    int cnt = 0;
    double rowsum = 0.0;
    for (unsigned long long mm = mymask; mm; mm &= mm - 1) {
      int z = __builtin_ctzll(mm);
      double nu0, nu1;
      double v = pair_weight(P, k, pdw, sZ[2 * z], sZ[2 * z + 1], nu0, nu1);
      double wn = v / colsum[z];
      rowsum += wn;
      if (wn != 0.0) used |= (1ull << z);
      if (wn > 0.0) cnt++;
    }
    const int off = wave_excl_scan(cnt, lane);
    const int total = __shfl(off + cnt, 63, 64);
    // loop B: emit in (m, z) row-major order
    int pos = base + off;
    for (unsigned long long mm = mymask; mm; mm &= mm - 1) {
      int z = __builtin_ctzll(mm);
      double nu0 = 0, nu1 = 0;
      double v = pair_weight(P, k, pdw, sZ[2 * z], sZ[2 * z + 1], nu0, nu1);
      double wn = v / colsum[z];
      if (wn > 0.0) {
        if (pos < B.cap) {
          pW[pos] = wn;
          pWP[pos] = 0.0;  // addGaussian: weight_prev = 0 (GaussianMixture.hpp:267-284)
          pMX[pos] = mx + (k.k00 * nu0 + k.k01 * nu1);
          pMY[pos] = my + (k.k10 * nu0 + k.k11 * nu1);
          pSXX[pos] = k.p00;
          pSXY[pos] = k.p01;
          pSYY[pos] = k.p11;
        } else {
          overflow = true;
        }
        pos++;
      }
    }
    base += total;
    // missed detection (:686-706); setWeight keeps the old weight in w_prev
    if (act) {
      double w_k = (1 - pd) * w;
      if (close && w > P.birthW) {
        double delta_w = pd * w - rowsum;
        if (delta_w > 0) {
          w_k += delta_w;
          if (w_k > 1) w_k = 1;
        }
      }
      pWP[m] = w;
      pW[m] = w_k;
    }
  }
  DBG_T(0, 2);
  used = wave_or_u64(used);
  if (__ballot(overflow) != 0ull) {
    if (lane == 0) atomicOr(B.err, ERRBIT_CAPACITY);
    if (base > B.cap) base = B.cap;
  }
  if (lane == 0) {
    B.count[i] = base;
    B.unusedMask[i] = (~used) & zmask;  // :709-720
    B.nInFov[i] = nFov;
  }
  if (P.useCluster) {  // :570-580, :652-668
    double s = RFS_DENORM_MIN + wave_sum(wsum);
    double prod = 1.0;
    for (int z = 0; z < nZ; z++) prod *= colsum[z];
    if (lane == 0) B.weight[i] = exp(s) * prod * B.weight[i];
  }
}
