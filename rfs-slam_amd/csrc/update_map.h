// update_map.h -- phd_update_map: RBPHDFilter::updateMap (reference include/RBPHDFilter.hpp:543-725) with
// KalmanFilter::correct (include/KalmanFilter.hpp:261-342), KalmanFilter_RngBrg::calculateInnovation
// (src/KalmanFilter_RngBrg.cpp:52-65) and MeasurementModel_RngBrg::{measure, probabilityOfDetection}
// fused into one kernel.
//
// Two forms of the same per-particle routine (see the comments above each): phd_update_map_particle, one wavefront per
// particle (the stand-alone kernel), and phd_update_map_block, a workgroup per particle (inside the fused step kernel).
// The measurement set is staged once per workgroup in LDS and, where the index is wave-uniform, read through the scalar
// cache.  The nM x nZ weight table of the reference never exists: phase 1 builds each landmark's innovation-gate bitmask,
// puts only the gated pairs through the Mahalanobis gate and the Gaussian, and writes the survivors into a dense
// (m, z)-row-major list in LDS (list position == output slot); lane z folds the survivors of measurement z into its
// normaliser in the reference's summation order (clutter, then landmarks ascending).  Phase 2 is dense over the list:
// normalise, recompute the landmark's KF quantities, emit.  Phase 3: missed-detection weights + near-limit heuristic.
// HBM traffic per particle = one read of the mixture + the appended records + the weight planes.
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

// LDS per wave: survivor list (value, packed (m,z)) + per-landmark segment (start, count) + final normalisers.
__host__ __device__ inline size_t update_map_lds_bytes_per_wave(int cap) {
  return (size_t)cap * (8 + 4 + 4) + RFSGPU_MAX_Z * 8;
}

// Gates of KalmanFilter_RngBrg::calculateInnovation only (cheap part of pair_weight).
__device__ __forceinline__ bool pair_gate(const Params &P, const LmKF &k, double z0, double z1) {
  const double e0 = z0 - k.zx0;
  if (P.kfRange > 0 && fabs(e0) > P.kfRange) return false;
  const double w1 = wrap_pi(z1 - k.zx1);
  if (P.kfBearing > 0 && fabs(w1) > P.kfBearing) return false;
  return true;
}
// Mahalanobis distance of the pair as pair_value forms it (same expression, same bits)
__device__ __forceinline__ double pair_md2(const LmKF &k, double z0, double z1) {
  const double e0 = z0 - k.zx0, e1 = z1 - k.zx1;  // UNWRAPPED difference
  const double t0 = e0 * k.i00 + e1 * k.i10;
  const double t1 = e0 * k.i01 + e1 * k.i11;
  return t0 * e0 + t1 * e1;
}
// Likelihood part for a pair that passed the innovation gates: Pd*w*lik or 0 (KalmanFilter.hpp:317-326, RBPHDFilter.hpp:622-632).
__device__ __forceinline__ double pair_value(const Params &P, const LmKF &k, double pdw, double z0, double z1) {
  const double e0 = z0 - k.zx0, e1 = z1 - k.zx1;  // UNWRAPPED difference
  const double t0 = e0 * k.i00 + e1 * k.i10;
  const double t1 = e0 * k.i01 + e1 * k.i11;
  const double md2 = t0 * e0 + t1 * e1;
  if (md2 > P.newGaussMd2) return 0.0;
  const double lik = gauss_from_md2(md2, k.factor);
  if (lik == 0.0) return 0.0;
  return pdw * lik;
}

// ---- innovation-gate prefilter in packed fp32 ------------------------------------------------------------------------------
// The gates of KalmanFilter_RngBrg::calculateInnovation (src/KalmanFilter_RngBrg.cpp:52-65) reject almost every
// (landmark, measurement) pair, and evaluating them in fp64 for all nM x nZ pairs was the largest single item of the map
// update.  The sweep over the measurement set therefore runs in fp32, two measurements per packed instruction (v_pk_*), with
// thresholds widened by a bound on the fp32 rounding of the operands, the difference and the 2-pi reduction: it can only
// let extra pairs through, never drop one.  Every candidate then goes through the exact fp64 gate (pair_gate) before its
// likelihood is evaluated, so results are unchanged.
typedef float f2_t __attribute__((ext_vector_type(2)));

// Stages the measurement set into the LDS head (RFS_Z_LDS_BYTES): doubles, the fp32 copy, and the two maxima the prefilter's
// error bound needs.  All threads of the block call it; the caller synchronises afterwards.
// `zsrc(t)` yields double t of the set (a device buffer, or the kernel-argument block indexed in place -- taking the
// address of a by-value kernel argument would make the compiler copy it to scratch).
template <class ZSrc>
__device__ __forceinline__ void stage_measurements_lds(unsigned char *smem_raw, ZSrc zsrc, int nZ, int tid, int nThreads) {
  double *sZ = reinterpret_cast<double *>(smem_raw);
  float *sZf = reinterpret_cast<float *>(smem_raw + 2 * RFSGPU_MAX_Z * 8);
  for (int t = tid; t < 2 * nZ; t += nThreads) sZ[t] = zsrc(t);
  if (tid < 64) {
    float ar = 0.f, ab = 0.f;
    const int z = tid < nZ ? tid : 0;
    const float zr = nZ > 0 ? (float)zsrc(2 * z) : 0.f, zb = nZ > 0 ? (float)zsrc(2 * z + 1) : 0.f;
    sZf[tid] = zr;                       // (entries beyond nZ repeat measurement 0: the sweep reads whole pairs)
    sZf[RFSGPU_MAX_Z + tid] = zb;
    ar = wave_max_f32(fabsf(zr));
    ab = wave_max_f32(fabsf(zb));
    if (tid == 0) { sZf[2 * RFSGPU_MAX_Z] = ar; sZf[2 * RFSGPU_MAX_Z + 1] = ab; }
  }
}

// Candidate mask of one landmark: bit z set if measurement z MAY pass both innovation gates.
__device__ __forceinline__ unsigned long long gate_candidates(const Params &P, const LmKF &k, const bool live, const int nZ, const float *sZf) {
  const float zx0 = (float)k.zx0, zx1 = (float)k.zx1;
  const float zrMax = sZf[2 * RFSGPU_MAX_Z], zbMax = sZf[2 * RFSGPU_MAX_Z + 1];
  const float inf = __builtin_huge_valf();
  // |computed - exact| <= 2^-23 (|a| + |b|) for the rounded operands and difference; the bearing also carries the fp32 2-pi
  // reduction (|k| <= (zbMax + pi) / 2pi + 1 multiples of a constant that is off by < 2e-7)
  float thrR = (P.kfRange > 0) ? (float)P.kfRange * (1.f + 1e-6f) + 2.5e-7f * (zrMax + fabsf(zx0)) + 1e-30f : inf;
  float thrB = (P.kfBearing > 0) ? (float)P.kfBearing * (1.f + 1e-6f) + 1e-6f * (zbMax + 3.2f) + 1e-6f : inf;
  if (!(zbMax < 50.f)) thrB = inf;       // far outside any sensible bearing range (or NaN): the exact test decides
  if (!(zrMax < 1.0e30f)) thrR = inf;
  const f2_t vx0 = {zx0, zx0}, vx1 = {zx1, zx1};
  const f2_t twoPi = {6.2831853071795864769f, 6.2831853071795864769f}, inv2Pi = {0.15915494309189533577f, 0.15915494309189533577f};
  unsigned lo = 0, hi = 0;
#pragma unroll 1
  for (int half = 0; half < 2; half++) {
    const int zBase = 32 * half;
    if (zBase >= nZ) break;
    unsigned bits = 0;
    const f2_t *pr = reinterpret_cast<const f2_t *>(sZf + zBase), *pb = reinterpret_cast<const f2_t *>(sZf + RFSGPU_MAX_Z + zBase);
    const int nPairs = (min(nZ - zBase, 32) + 1) >> 1;
#pragma unroll 4
    for (int q = 0; q < nPairs; q++) {
      const f2_t zr = pr[q], zb = pb[q];     // uniform address: LDS broadcast reads of two measurements each
      const f2_t e0 = zr - vx0;
      f2_t w = zb - vx1;
      f2_t kk = w * inv2Pi;
      kk.x = __builtin_rintf(kk.x); kk.y = __builtin_rintf(kk.y);
      w = w - kk * twoPi;
      // !(x > t) form: a NaN is a candidate
      const bool c0 = ((int)!(fabsf(e0.x) > thrR) & (int)!(fabsf(w.x) > thrB)) != 0;
      const bool c1 = ((int)!(fabsf(e0.y) > thrR) & (int)!(fabsf(w.y) > thrB)) != 0;
      bits |= (c0 ? 1u : 0u) << (2 * q);
      bits |= (c1 ? 2u : 0u) << (2 * q);
    }
    if (half == 0) lo = bits; else hi = bits;
  }
  unsigned long long m = ((unsigned long long)hi << 32) | lo;
  m &= (nZ >= 64) ? ~0ull : ((1ull << nZ) - 1ull);
  return live ? m : 0ull;
}

// New Gaussian of a surviving (landmark, measurement) pair: KalmanFilter::correct's mean (include/KalmanFilter.hpp:311-316)
// with the wrapped innovation, and the landmark's updated covariance.
__device__ __forceinline__ void emit_updated(const LmKF &k, double mx, double my, double z0, double z1, double *pMX, double *pMY, double *pSXX,
                                             double *pSXY, double *pSYY, int pos) {
  const double nu0 = z0 - k.zx0;
  const double nu1 = wrap_pi(z1 - k.zx1);
  pMX[pos] = mx + (k.k00 * nu0 + k.k01 * nu1);
  pMY[pos] = my + (k.k10 * nu0 + k.k11 * nu1);
  pSXX[pos] = k.p00;
  pSXY[pos] = k.p01;
  pSYY[pos] = k.p11;
}

#define UPDMAP_KEEP 4  // survivor values a lane keeps in registers between the likelihood loop and the list write

// Structure (one wavefront per particle):
//  phase 1, per pass of 64 landmarks: KF quantities once per landmark; the packed-fp32 sweep over the measurements gives each
//    landmark's candidate mask; only the set bits (a few per landmark) go through the exact innovation gates, the Mahalanobis
//    gate and the Gaussian; survivors are written into a dense (m,z)-row-major list in LDS through a wave prefix sum -- list
//    position == output slot of the new Gaussian, whose mean and covariance are written to the slab right there (the lane
//    still holds the landmark's gain and updated covariance); lane z then folds this pass's survivors of measurement z into
//    its normaliser in landmark order, i.e. in the reference's summation order (clutter first, then m ascending).
//  phase 2, dense over the survivor list: normalise, write the weights.  (A new Gaussian whose normalised weight is not
//    positive -- only through inf / NaN arithmetic -- is dropped by an in-place compaction of the appended block.)
//  phase 3: missed-detection weights (+ near-limit heuristic from the landmark's list segment), unused mask.
#ifndef UPDMAP_WAVES_PER_EU
#define UPDMAP_WAVES_PER_EU 2
#endif

// In-place, order-preserving removal of the appended Gaussians whose normalised weight (sV[q]) is not > 0; executed by ONE
// wave (the rare path of phase 2).  Returns the number kept.
__device__ __forceinline__ int compact_appended(const double *sV, int nM, int nSurv, int lane, double *pW, double *pWP, double *pMX, double *pMY,
                                                double *pSXX, double *pSXY, double *pSYY) {
  int out = 0;
  for (int s0 = 0; s0 < nSurv; s0 += 64) {
    const int q = s0 + lane;
    const bool keep = (q < nSurv) && (sV[q] > 0.0);
    const unsigned long long km = __ballot(keep);
    double v0 = 0, v1 = 0, v2 = 0, v3 = 0, v4 = 0, v5 = 0;
    if (keep) { v0 = sV[q]; v1 = pMX[nM + q]; v2 = pMY[nM + q]; v3 = pSXX[nM + q]; v4 = pSXY[nM + q]; v5 = pSYY[nM + q]; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    wave_sync();
    if (keep) {
      const int d = nM + out + __popcll(km & ((1ull << lane) - 1ull));
      pW[d] = v0; pWP[d] = 0.0; pMX[d] = v1; pMY[d] = v2; pSXX[d] = v3; pSXY[d] = v4; pSYY[d] = v5;
    }
    out += __popcll(km);
  }
  return out;
}

// One particle, one wavefront: `lane` of the calling wave, `smemZ` the workgroup's staged measurement set (RFS_Z_LDS_BYTES), `wb`
// this wave's LDS block (update_map_lds_bytes_per_wave).  The stand-alone kernel's form.
__device__ __forceinline__ void phd_update_map_particle(const Buffers &B, const Params &P, const int cur, const int nZ, const int i,
                                                        const int lane, const unsigned char *smemZ, unsigned char *wb) {
  const double *sZ = reinterpret_cast<const double *>(smemZ);
  const float *sZf = reinterpret_cast<const float *>(smemZ + 2 * RFSGPU_MAX_Z * 8);
  const int cap = B.cap;
  double *sV = reinterpret_cast<double *>(wb);                 // [cap] survivor values Pd*w*lik, later the normalised weights
  double *sCol = sV + cap;                                      // [MAX_Z] final normalisers
  unsigned *sMZ = reinterpret_cast<unsigned *>(sCol + RFSGPU_MAX_Z);  // [cap] (m << 8) | z
  unsigned *sSeg = sMZ + cap;                                   // [cap] per landmark: (start << 8) | count

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
  double *pW = plane(slab, cap, i, PL_W), *pWP = plane(slab, cap, i, PL_WP);
  double *pMX = plane(slab, cap, i, PL_MX), *pMY = plane(slab, cap, i, PL_MY);
  double *pSXX = plane(slab, cap, i, PL_SXX), *pSXY = plane(slab, cap, i, PL_SXY), *pSYY = plane(slab, cap, i, PL_SYY);

  DBG_T(0, 0);
  PoseReg pr;
  load_pose(B, P, i, pr);

  const int nPass = (nM + 63) >> 6;
  const int room = cap - nM;  // survivors that still fit as new Gaussians
  int nFov = 0;
  double wsum = 0.0;           // SC-PHD: sum of prior weights
  double cs = P.clutter;       // lane z: normaliser of measurement z (reference: sum = clutter; sum += W[m][z] ...)
  int nSurv = 0;
  bool overflow = false;

  // ---------------- phase 1 ----------------
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
    if (p == 0) DBG_T(0, 4);
    const unsigned long long cand = gate_candidates(P, k, fov && k.ok, nZ, sZf);
    if (p == 0) DBG_T(0, 5);
    // exact innovation gates + Mahalanobis gate + likelihood for the candidates
    unsigned long long surv = 0;
    double keepV[UPDMAP_KEEP];
#pragma unroll
    for (int t = 0; t < UPDMAP_KEEP; t++) keepV[t] = 0.0;
    int cnt = 0;
    // two loops, each as long as its busiest lane: the candidates of the fp32 sweep (a handful per landmark) through the exact
    // innovation gates and the Mahalanobis gate -- distance only --, then the few that are left (seldom more than one per landmark)
    // through the Gaussian: as one loop every trip paid for the exp and the division because SOME lane's candidate had passed
    unsigned long long gated = 0;
    for (unsigned long long g = cand; g; g &= g - 1) {
      const int z = __builtin_ctzll(g);
      const double z0 = sZ[2 * z], z1 = sZ[2 * z + 1];
      if (!pair_gate(P, k, z0, z1)) continue;
      if (pair_md2(k, z0, z1) > P.newGaussMd2) continue;
      gated |= 1ull << z;
    }
    for (unsigned long long g = gated; g; g &= g - 1) {
      const int z = __builtin_ctzll(g);
      const double z0 = sZ[2 * z], z1 = sZ[2 * z + 1];
      const double v = pair_value(P, k, pdw, z0, z1);
      if (v != 0.0) {
        surv |= (1ull << z);
#pragma unroll
        for (int t = 0; t < UPDMAP_KEEP; t++) keepV[t] = (cnt == t) ? v : keepV[t];
        cnt++;
      }
    }
    if (p == 0) DBG_T(0, 6);
    const int off = wave_excl_scan(cnt, lane);
    const int total = __builtin_amdgcn_readlane(off + cnt, 63);
    if (act) sSeg[m] = ((unsigned)(nSurv + off) << 8) | (unsigned)cnt;
    {
      int pos = nSurv + off, q = 0;
      for (unsigned long long g = surv; g; g &= g - 1, q++, pos++) {
        const int z = __builtin_ctzll(g);
        if (pos < room) {
          const double z0 = sZ[2 * z], z1 = sZ[2 * z + 1];
          double v = keepV[0];
#pragma unroll
          for (int t = 1; t < UPDMAP_KEEP; t++) v = (q == t) ? keepV[t] : v;
          if (q >= UPDMAP_KEEP) v = pair_value(P, k, pdw, z0, z1);
          sV[pos] = v;
          sMZ[pos] = ((unsigned)m << 8) | (unsigned)z;
          emit_updated(k, mx, my, z0, z1, pMX, pMY, pSXX, pSXY, pSYY, nM + pos);
        } else {
          overflow = true;
        }
      }
    }
    wave_sync();
    if (p == 0) DBG_T(0, 7);
    // lane z folds this pass's survivors of measurement z into its normaliser, in landmark order
    {
      const int lo = nSurv, hi = (nSurv + total < room) ? nSurv + total : room;
      int sIdx = lo;
      for (; sIdx + 4 <= hi; sIdx += 4) {  // broadcast reads, four in flight; the adds stay in list (= landmark) order
        const unsigned mz0 = sMZ[sIdx], mz1 = sMZ[sIdx + 1], mz2 = sMZ[sIdx + 2], mz3 = sMZ[sIdx + 3];
        const double v0 = sV[sIdx], v1 = sV[sIdx + 1], v2 = sV[sIdx + 2], v3 = sV[sIdx + 3];
        if ((int)(mz0 & 0xffu) == lane) cs += v0;
        if ((int)(mz1 & 0xffu) == lane) cs += v1;
        if ((int)(mz2 & 0xffu) == lane) cs += v2;
        if ((int)(mz3 & 0xffu) == lane) cs += v3;
      }
      for (; sIdx < hi; sIdx++) {
        const unsigned mz = sMZ[sIdx];
        const double v = sV[sIdx];
        if ((int)(mz & 0xffu) == lane) cs += v;
      }
    }
    nSurv += total;
    if (p == 0) DBG_T(0, 8);
  }
  DBG_T(0, 1);
  if (__ballot(overflow) != 0ull || nSurv > room) {
    if (lane == 0) atomicOr(B.err, ERRBIT_CAPACITY);
    nSurv = room < 0 ? 0 : (nSurv > room ? room : nSurv);
  }
  sCol[lane] = cs;
  wave_sync();

  // ---------------- phase 2: normalise, write the weights of the appended Gaussians ----------------
  unsigned long long used = 0;
  bool dropAny = false;
  for (int s0 = 0; s0 < nSurv; s0 += 64) {
    const int sIdx = s0 + lane;
    const bool act = sIdx < nSurv;
    double wn = 0.0;
    int z = 0;
    if (act) { z = (int)(sMZ[sIdx] & 0xffu); wn = sV[sIdx] / sCol[z]; sV[sIdx] = wn; }
    if (act && wn != 0.0) used |= (1ull << z);
    if (act) { pW[nM + sIdx] = wn; pWP[nM + sIdx] = 0.0; }  // addGaussian: weight_prev = 0 (GaussianMixture.hpp:267-284)
    dropAny |= act && !(wn > 0.0);  // :677
  }
  int outBase = nM + nSurv;
  if (__ballot(dropAny) != 0ull) {
    wave_sync();
    outBase = nM + compact_appended(sV, nM, nSurv, lane, pW, pWP, pMX, pMY, pSXX, pSXY, pSYY);
  }
  wave_sync();

  // ---------------- phase 3: missed-detection weights (:686-706); setWeight keeps the old weight in w_prev ----------------
  for (int m = lane; m < nM; m += 64) {
    const double w = pW[m];
    const double dx = pMX[m] - pr.x, dy = pMY[m] - pr.y;
    bool close;
    double pd = rb_pd(P, sqrt(dx * dx + dy * dy), close);
    if (close) pd = 1;
    double w_k = (1 - pd) * w;
    if (close && w > P.birthW) {
      const unsigned seg = sSeg[m];
      const int st = (int)(seg >> 8), c = (int)(seg & 0xffu);
      double rowsum = 0.0;
      for (int q = st; q < st + c && q < nSurv; q++) rowsum += sV[q];  // (already divided by the normaliser)
      const double delta_w = pd * w - rowsum;
      if (delta_w > 0) {
        w_k += delta_w;
        if (w_k > 1) w_k = 1;
      }
    }
    pWP[m] = w;
    pW[m] = w_k;
  }
  DBG_T(0, 2);
  used = wave_or_u64(used);
  if (lane == 0) {
    B.count[i] = outBase;
    B.unusedMask[i] = (~used) & zmask;  // :709-720
    B.nInFov[i] = nFov;
  }
  if (P.useCluster) {  // :570-580, :652-668
    const double s = RFS_DENORM_MIN + wave_sum_dpp(wsum);
    double prod = 1.0;
    for (int z = 0; z < nZ; z++) prod *= readlane_f64(cs, z);
    if (lane == 0) B.weight[i] = exp(s) * prod * B.weight[i];
  }
}

// LDS of the workgroup form below: the single-wave layout + cross-wave scratch.
__host__ __device__ inline size_t update_map_block_lds_bytes(int cap) {
  return update_map_lds_bytes_per_wave(cap) + 128;
}

// The same map update by a workgroup of WPP waves (used inside the fused step kernel, where the particle's workgroup has
// more than one wave): passes cover WPP*64 landmarks; the survivor list keeps its (m, z) order through a cross-wave
// offset exchange, wave 0 folds the normalisers in list order (the reference's summation order), phases 2 and 3 are
// split over all threads.  Bit-identical to phd_update_map_particle.
template <int WPP>
__device__ __forceinline__ void phd_update_map_block(const Buffers &B, const Params &P, const int cur, const int nZ, const int i, const int tid,
                                                     const unsigned char *smemZ, unsigned char *wb, const bool phasePrio = false) {
  constexpr int NT = WPP * 64;
  const double *sZ = reinterpret_cast<const double *>(smemZ);
  const float *sZf = reinterpret_cast<const float *>(smemZ + 2 * RFSGPU_MAX_Z * 8);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int cap = B.cap;
  double *sV = reinterpret_cast<double *>(wb);                 // [cap] survivor values Pd*w*lik, later the normalised weights
  double *sCol = sV + cap;                                      // [MAX_Z] final normalisers
  unsigned *sMZ = reinterpret_cast<unsigned *>(sCol + RFSGPU_MAX_Z);  // [cap] (m << 8) | z
  unsigned *sSeg = sMZ + cap;                                   // [cap] per landmark: (start << 8) | count
  int *sTot = reinterpret_cast<int *>(sSeg + cap);              // [2][8] survivors per wave of a pass (double-buffered)
  int *sMisc = sTot + 16;                                       // [0] overflow flag [1] landmarks in FOV [2] a new Gaussian has to be dropped
  unsigned *sUsed = reinterpret_cast<unsigned *>(sMisc + 4);    // [2] used-measurement mask (lo, hi)

  const int nM = B.count[i];
  const unsigned long long zmask = (nZ >= 64) ? ~0ull : ((1ull << nZ) - 1ull);
  if (nM == 0) {  // :559-564
    if (tid == 0) {
      B.unusedMask[i] = zmask;
      B.nInFov[i] = 0;
    }
    return;
  }
  double *slab = B.slab[cur];
  double *pW = plane(slab, cap, i, PL_W), *pWP = plane(slab, cap, i, PL_WP);
  double *pMX = plane(slab, cap, i, PL_MX), *pMY = plane(slab, cap, i, PL_MY);
  double *pSXX = plane(slab, cap, i, PL_SXX), *pSXY = plane(slab, cap, i, PL_SXY), *pSYY = plane(slab, cap, i, PL_SYY);

  PoseReg pr;
  load_pose(B, P, i, pr);
  if (tid < 6) { if (tid < 4) sMisc[tid] = 0; else sUsed[tid - 4] = 0u; }

  const int nPass = (nM + NT - 1) / NT;
  const int room = cap - nM;  // survivors that still fit as new Gaussians
  int nFov = 0;
  double cs = P.clutter;       // wave 0, lane z: normaliser of measurement z (reference: sum = clutter; sum += W[m][z] ...)
  int nSurv = 0;
  bool overflow = false;

  // (step_fused.h: issue priority falls from phase to phase; raised here rather than at the kernel's first instruction,
  //  where the scheduling barrier it forms made the register allocator spill 84 B per lane)
  //  Inside this phase, where all workgroups of a CU start together, the waves in the last hardware slot of their SIMD
  //  (HW_ID.wave_id: the last arrivals, which the arbiter ranks lowest) get the level above the others.
#ifndef STEP_MAP_PRIO
#define STEP_MAP_PRIO 2
#endif
  if (phasePrio) { if ((__builtin_amdgcn_s_getreg((4 - 1) << 11 | 4) & 0xfu) >= 3u) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(STEP_MAP_PRIO); }
  // ---------------- phase 1 ----------------
  for (int p = 0; p < nPass; p++) {
    const int m = p * NT + tid;
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
    RFS_CUT(1);
    const unsigned long long cand = gate_candidates(P, k, fov && k.ok, nZ, sZf);
    RFS_CUT(2);
    unsigned long long surv = 0;
    double keepV[UPDMAP_KEEP];
#pragma unroll
    for (int t = 0; t < UPDMAP_KEEP; t++) keepV[t] = 0.0;
    int cnt = 0;
    // two loops, each as long as its busiest lane: the candidates of the fp32 sweep (a handful per landmark) through the exact
    // innovation gates and the Mahalanobis gate -- distance only --, then the few that are left (seldom more than one per landmark)
    // through the Gaussian: as one loop every trip paid for the exp and the division because SOME lane's candidate had passed
    unsigned long long gated = 0;
    for (unsigned long long g = cand; g; g &= g - 1) {
      const int z = __builtin_ctzll(g);
      const double z0 = sZ[2 * z], z1 = sZ[2 * z + 1];
      if (!pair_gate(P, k, z0, z1)) continue;
      if (pair_md2(k, z0, z1) > P.newGaussMd2) continue;
      gated |= 1ull << z;
    }
    for (unsigned long long g = gated; g; g &= g - 1) {
      const int z = __builtin_ctzll(g);
      const double z0 = sZ[2 * z], z1 = sZ[2 * z + 1];
      const double v = pair_value(P, k, pdw, z0, z1);
      if (v != 0.0) {
        surv |= (1ull << z);
#pragma unroll
        for (int t = 0; t < UPDMAP_KEEP; t++) keepV[t] = (cnt == t) ? v : keepV[t];
        cnt++;
      }
    }
    RFS_CUT(3);
    const int off = wave_excl_scan(cnt, lane);
    const int totalW = __builtin_amdgcn_readlane(off + cnt, 63);
    int *tot = sTot + 8 * (p & 1);
    if (lane == 0) tot[wave] = totalW;
    __syncthreads();
    int before = 0, total = 0;
#pragma unroll
    for (int w2 = 0; w2 < WPP; w2++) { const int t = tot[w2]; before += (w2 < wave) ? t : 0; total += t; }
    if (act) sSeg[m] = ((unsigned)(nSurv + before + off) << 8) | (unsigned)cnt;
    {
      int pos = nSurv + before + off, q = 0;
      for (unsigned long long g = surv; g; g &= g - 1, q++, pos++) {
        const int z = __builtin_ctzll(g);
        if (pos < room) {
          const double z0 = sZ[2 * z], z1 = sZ[2 * z + 1];
          double v = keepV[0];
#pragma unroll
          for (int t = 1; t < UPDMAP_KEEP; t++) v = (q == t) ? keepV[t] : v;
          if (q >= UPDMAP_KEEP) v = pair_value(P, k, pdw, z0, z1);
          sV[pos] = v;
          sMZ[pos] = ((unsigned)m << 8) | (unsigned)z;
          emit_updated(k, mx, my, z0, z1, pMX, pMY, pSXX, pSXY, pSYY, nM + pos);
        } else {
          overflow = true;
        }
      }
    }
    __syncthreads();
    RFS_CUT(4);
    if (wave == 0) {  // lane z folds this pass's survivors of measurement z into its normaliser, in landmark order
      const int lo = nSurv, hi = (nSurv + total < room) ? nSurv + total : room;
      int sIdx = lo;
      for (; sIdx + 4 <= hi; sIdx += 4) {  // broadcast reads, four in flight; the adds stay in list (= landmark) order
        const unsigned mz0 = sMZ[sIdx], mz1 = sMZ[sIdx + 1], mz2 = sMZ[sIdx + 2], mz3 = sMZ[sIdx + 3];
        const double v0 = sV[sIdx], v1 = sV[sIdx + 1], v2 = sV[sIdx + 2], v3 = sV[sIdx + 3];
        if ((int)(mz0 & 0xffu) == lane) cs += v0;
        if ((int)(mz1 & 0xffu) == lane) cs += v1;
        if ((int)(mz2 & 0xffu) == lane) cs += v2;
        if ((int)(mz3 & 0xffu) == lane) cs += v3;
      }
      for (; sIdx < hi; sIdx++) {
        const unsigned mz = sMZ[sIdx];
        const double v = sV[sIdx];
        if ((int)(mz & 0xffu) == lane) cs += v;
      }
    }
    nSurv += total;
    RFS_CUT(5);
  }
  RFS_CUT(6);
  if (__ballot(overflow) != 0ull && lane == 0) atomicOr(&sMisc[0], 1);
  if (lane == 0) atomicAdd(&sMisc[1], nFov);
  if (wave == 0) sCol[lane] = cs;
  __syncthreads();
  if (sMisc[0] != 0 || nSurv > room) {
    if (tid == 0) atomicOr(B.err, ERRBIT_CAPACITY);
    nSurv = room < 0 ? 0 : (nSurv > room ? room : nSurv);
  }

  // ---------------- phase 2: normalise in place, write the weights of the appended Gaussians ----------------
  {
    unsigned long long used = 0;
    bool dropAny = false;
    for (int s0 = wave * 64; s0 < nSurv; s0 += NT) {
      const int sIdx = s0 + lane;
      const bool act = sIdx < nSurv;
      double wn = 0.0;
      int z = 0;
      if (act) { z = (int)(sMZ[sIdx] & 0xffu); wn = sV[sIdx] / sCol[z]; sV[sIdx] = wn; }
      if (act && wn != 0.0) used |= (1ull << z);
      if (act) { pW[nM + sIdx] = wn; pWP[nM + sIdx] = 0.0; }  // addGaussian: weight_prev = 0 (GaussianMixture.hpp:267-284)
      dropAny |= act && !(wn > 0.0);  // :677
    }
    used = wave_or_u64(used);
    if (lane == 0) { atomicOr(&sUsed[0], (unsigned)(used & 0xffffffffull)); atomicOr(&sUsed[1], (unsigned)(used >> 32)); }
    if (__ballot(dropAny) != 0ull && lane == 0) atomicOr(&sMisc[2], 1);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __syncthreads();
  int outBase = nM + nSurv;
  if (sMisc[2] != 0) {  // rare: a normalised weight that is not positive (inf / NaN arithmetic) -- drop it, keep the order
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (wave == 0) {
      const int kept = compact_appended(sV, nM, nSurv, lane, pW, pWP, pMX, pMY, pSXX, pSXY, pSYY);
      if (lane == 0) sMisc[3] = kept;
    }
    __syncthreads();
    outBase = nM + sMisc[3];
  }

  RFS_CUT(7);
  // ---------------- phase 3: missed-detection weights (:686-706); setWeight keeps the old weight in w_prev ----------------
  for (int m = tid; m < nM; m += NT) {
    const double w = pW[m];
    const double dx = pMX[m] - pr.x, dy = pMY[m] - pr.y;
    bool close;
    double pd = rb_pd(P, sqrt(dx * dx + dy * dy), close);
    if (close) pd = 1;
    double w_k = (1 - pd) * w;
    if (close && w > P.birthW) {
      const unsigned seg = sSeg[m];
      const int st = (int)(seg >> 8), c = (int)(seg & 0xffu);
      double rowsum = 0.0;
      for (int q = st; q < st + c && q < nSurv; q++) rowsum += sV[q];  // (already divided by the normaliser)
      const double delta_w = pd * w - rowsum;
      if (delta_w > 0) {
        w_k += delta_w;
        if (w_k > 1) w_k = 1;
      }
    }
    pWP[m] = w;
    pW[m] = w_k;
  }
  if (tid == 0) {
    B.count[i] = outBase;
    const unsigned long long used = (unsigned long long)sUsed[0] | ((unsigned long long)sUsed[1] << 32);
    B.unusedMask[i] = (~used) & zmask;  // :709-720
    B.nInFov[i] = sMisc[1];
  }
  if (P.useCluster) __syncthreads();  // (phase 3 left the prior weights in w_prev)
  if (P.useCluster && wave == 0) {  // :570-580, :652-668 -- sum of the prior weights in the single-wave order
    double wsum = 0.0;
    for (int m = lane; m < nM; m += 64) wsum += pWP[m];
    const double s2 = RFS_DENORM_MIN + wave_sum_dpp(wsum);
    double prod = 1.0;
    for (int z = 0; z < nZ; z++) prod *= readlane_f64(cs, z);
    if (lane == 0) B.weight[i] = exp(s2) * prod * B.weight[i];
  }
}

template <int WPB>
__global__ __launch_bounds__(WPB * 64) __attribute__((amdgpu_waves_per_eu(UPDMAP_WAVES_PER_EU, UPDMAP_WAVES_PER_EU))) void phd_update_map_kernel(Buffers B, Params P, int cur, int nZ, const double *__restrict__ Zg) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const int wave = threadIdx.x >> 6;
  const int lane = threadIdx.x & 63;
  stage_measurements_lds(smem_raw, [&](int t) { return Zg[t]; }, nZ, (int)threadIdx.x, WPB * 64);
  __syncthreads();
  const int i = __builtin_amdgcn_readfirstlane(blockIdx.x * WPB + wave);
  if (i >= B.N) return;
  phd_update_map_particle(B, P, cur, nZ, i, lane, smem_raw, smem_raw + RFS_Z_LDS_BYTES + (size_t)wave * update_map_lds_bytes_per_wave(B.cap));
}

// The stand-alone kernel in the workgroup form: WPP waves per particle (phd_update_map_block, bit-identical to the single-wave
// routine).  One wave per particle leaves a 2000-particle launch at two waves per SIMD; with two waves per particle the phase
// hides its latencies as it does inside the fused step.
template <int WPP>
__global__ __launch_bounds__(WPP * 64) __attribute__((amdgpu_waves_per_eu(UPDMAP_WAVES_PER_EU, 4))) void phd_update_map_block_kernel(Buffers B, Params P, int cur, int nZ, const double *__restrict__ Zg) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  stage_measurements_lds(smem_raw, [&](int t) { return Zg[t]; }, nZ, (int)threadIdx.x, WPP * 64);
  __syncthreads();
  phd_update_map_block<WPP>(B, P, cur, nZ, (int)blockIdx.x, (int)threadIdx.x, smem_raw, smem_raw + RFS_Z_LDS_BYTES);
}

