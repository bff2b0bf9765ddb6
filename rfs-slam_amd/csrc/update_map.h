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

// Structure (one wavefront per particle):
//  phase 1, per pass of 64 landmarks: KF quantities once per landmark; a cheap sweep over the measurements builds each
//    landmark's innovation-gate bitmask; only the set bits (a few per landmark) go through the Mahalanobis gate and the
//    Gaussian; survivors are written into a dense (m,z)-row-major list in LDS through a wave prefix sum -- list position
//    == output slot of the new Gaussian; lane z then folds this pass's survivors of measurement z into its normaliser
//    in landmark order, i.e. in the reference's summation order (clutter first, then m ascending).
//  phase 2, dense over the survivor list (all 64 lanes busy): normalise, recompute the landmark's KF quantities, emit.
//  phase 3: missed-detection weights (+ near-limit heuristic from the landmark's list segment), unused mask.
#ifndef UPDMAP_GATE_BATCH
#define UPDMAP_GATE_BATCH 8
#endif
#ifndef UPDMAP_WAVES_PER_EU
#define UPDMAP_WAVES_PER_EU 2
#endif
// One particle, one wavefront: `lane` of the calling wave, `sZ` the workgroup's LDS copy of the measurement set, `wb` this
// wave's LDS block (update_map_lds_bytes_per_wave).  Shared by the stand-alone kernel and the fused step kernel.
template <int GB>  // measurements whose gates are evaluated per trip (scalar loads up front, branch-free): 8 in the stand-alone
                   // kernel, 4 inside the fused step kernel, whose 128-VGPR budget it shares
__device__ __forceinline__ void phd_update_map_particle(const Buffers &B, const Params &P, const int cur, const int nZ,
                                                        const double *__restrict__ Zg, const int i, const int lane, const double *sZ,
                                                        unsigned char *wb) {
  const int cap = B.cap;
  double *sV = reinterpret_cast<double *>(wb);                 // [cap] survivor values Pd*w*lik
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
    // innovation gates for every measurement (cheap), as a bitmask
    unsigned long long gate = 0;
    {
      // Eight measurements per trip: their 16 doubles are fetched with wide scalar loads up front (uniform
      // addresses), the gate arithmetic is branch-free; the rare bearing difference beyond one wrap is redone exactly.
      const bool live = fov && k.ok;
      const bool useR = P.kfRange > 0, useB = P.kfBearing > 0;
      for (int z0 = 0; z0 < nZ; z0 += GB) {
        double zr[GB], zb[GB];
#pragma unroll
        for (int u = 0; u < GB; u++) {
          const int zz = (z0 + u < nZ) ? z0 + u : nZ - 1;
          zr[u] = Zg[2 * zz];
          zb[u] = Zg[2 * zz + 1];
        }
        bool redo = false;
#pragma unroll
        for (int u = 0; u < GB; u++) {
          const double e0 = zr[u] - k.zx0;
          double w1 = zb[u] - k.zx1;
          w1 = (w1 > RFS_PI) ? w1 - 2 * RFS_PI : w1;
          w1 = (w1 < -RFS_PI) ? w1 + 2 * RFS_PI : w1;
          // bitwise (non-short-circuit) logic on purpose: no branches, the 8 chains interleave
          redo = ((int)redo | (int)(w1 > RFS_PI) | (int)(w1 < -RFS_PI)) != 0;
          const int outR = (int)useR & (int)(fabs(e0) > P.kfRange), outB = (int)useB & (int)(fabs(w1) > P.kfBearing);
          const bool g = (outR | outB) == 0;
          gate |= ((int)live & (int)g & (int)(z0 + u < nZ)) ? (1ull << (z0 + u)) : 0ull;
        }
        if (__ballot(redo) != 0ull) {  // some bearing difference needs more than one wrap step: exact loop form
          for (int u = 0; u < GB && z0 + u < nZ; u++) {
            const bool g = pair_gate(P, k, zr[u], zb[u]);
            const unsigned long long bit = 1ull << (z0 + u);
            gate = (live && g) ? (gate | bit) : (gate & ~bit);
          }
        }
      }
    }
    if (p == 0) DBG_T(0, 5);
    // Mahalanobis gate + likelihood only for the set bits
    unsigned long long surv = 0;
    for (unsigned long long g = gate; g; g &= g - 1) {
      const int z = __builtin_ctzll(g);
      if (pair_value(P, k, pdw, sZ[2 * z], sZ[2 * z + 1]) != 0.0) surv |= (1ull << z);
    }
    if (p == 0) DBG_T(0, 6);
    const int cnt = __popcll(surv);
    const int off = wave_excl_scan(cnt, lane);
    const int total = __builtin_amdgcn_readlane(off + cnt, 63);
    if (act) sSeg[m] = ((unsigned)(nSurv + off) << 8) | (unsigned)cnt;
    {
      int pos = nSurv + off;
      for (unsigned long long g = surv; g; g &= g - 1) {
        const int z = __builtin_ctzll(g);
        if (pos < room) {
          sV[pos] = pair_value(P, k, pdw, sZ[2 * z], sZ[2 * z + 1]);
          sMZ[pos] = ((unsigned)m << 8) | (unsigned)z;
        } else {
          overflow = true;
        }
        pos++;
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

  // ---------------- phase 2: dense over survivors ----------------
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
    const bool keep = act && (wn > 0.0);  // :677
    const unsigned long long km = __ballot(keep);
    if (keep) {
      const int pos = outBase + __popcll(km & ((1ull << lane) - 1ull));
      const double mx = pMX[m], my = pMY[m], sxx = pSXX[m], sxy = pSXY[m], syy = pSYY[m];
      LmKF k;
      double range;
      lm_precompute(P, pr, mx, my, sxx, sxy, syy, k, range);
      const double nu0 = sZ[2 * z] - k.zx0;
      const double nu1 = wrap_pi(sZ[2 * z + 1] - k.zx1);
      pW[pos] = wn;
      pWP[pos] = 0.0;  // addGaussian: weight_prev = 0 (GaussianMixture.hpp:267-284)
      pMX[pos] = mx + (k.k00 * nu0 + k.k01 * nu1);
      pMY[pos] = my + (k.k10 * nu0 + k.k11 * nu1);
      pSXX[pos] = k.p00;
      pSXY[pos] = k.p01;
      pSYY[pos] = k.p11;
    }
    outBase += __popcll(km);
  }

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
      for (int q = st; q < st + c && q < nSurv; q++) rowsum += sV[q] / sCol[sMZ[q] & 0xffu];
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
  return update_map_lds_bytes_per_wave(cap) + (size_t)(cap / 64 + 1) * 4 + 128;
}

// The same map update by a workgroup of WPP waves (used inside the fused step kernel, where the particle's workgroup has
// more than one wave): passes cover WPP*64 landmarks; the survivor list keeps its (m, z) order through a cross-wave
// offset exchange, wave 0 folds the normalisers in list order (the reference's summation order), phases 2 and 3 are
// split over all threads.  Bit-identical to phd_update_map_particle.
template <int WPP, int GB>
__device__ __forceinline__ void phd_update_map_block(const Buffers &B, const Params &P, const int cur, const int nZ,
                                                     const double *__restrict__ Zg, const int i, const int tid, const double *sZ,
                                                     unsigned char *wb, const bool phasePrio = false) {
  constexpr int NT = WPP * 64;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int cap = B.cap;
  double *sV = reinterpret_cast<double *>(wb);                 // [cap] survivor values Pd*w*lik, later the normalised weights
  double *sCol = sV + cap;                                      // [MAX_Z] final normalisers
  unsigned *sMZ = reinterpret_cast<unsigned *>(sCol + RFSGPU_MAX_Z);  // [cap] (m << 8) | z
  unsigned *sSeg = sMZ + cap;                                   // [cap] per landmark: (start << 8) | count
  int *sKeep = reinterpret_cast<int *>(sSeg + cap);             // [cap/64 + 1] new Gaussians per 64-survivor chunk
  int *sTot = sKeep + (cap / 64 + 1);                           // [2][8] survivors per wave of a pass (double-buffered)
  int *sMisc = sTot + 16;                                       // [0] overflow flag [1] landmarks in FOV
  unsigned *sUsed = reinterpret_cast<unsigned *>(sMisc + 2);    // [2] used-measurement mask (lo, hi)

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
  if (tid < 4) { if (tid < 2) sMisc[tid] = 0; else sUsed[tid - 2] = 0u; }

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
  if (phasePrio) { if ((__builtin_amdgcn_s_getreg((4 - 1) << 11 | 4) & 0xfu) >= 3u) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(2); }
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
    unsigned long long gate = 0;
    {
      const bool live = fov && k.ok;
      const bool useR = P.kfRange > 0, useB = P.kfBearing > 0;
      for (int z0 = 0; z0 < nZ; z0 += GB) {
        double zr[GB], zb[GB];
#pragma unroll
        for (int u = 0; u < GB; u++) {
          const int zz = (z0 + u < nZ) ? z0 + u : nZ - 1;
          zr[u] = Zg[2 * zz];
          zb[u] = Zg[2 * zz + 1];
        }
        bool redo = false;
#pragma unroll
        for (int u = 0; u < GB; u++) {
          const double e0 = zr[u] - k.zx0;
          double w1 = zb[u] - k.zx1;
          w1 = (w1 > RFS_PI) ? w1 - 2 * RFS_PI : w1;
          w1 = (w1 < -RFS_PI) ? w1 + 2 * RFS_PI : w1;
          redo = ((int)redo | (int)(w1 > RFS_PI) | (int)(w1 < -RFS_PI)) != 0;
          const int outR = (int)useR & (int)(fabs(e0) > P.kfRange), outB = (int)useB & (int)(fabs(w1) > P.kfBearing);
          const bool g = (outR | outB) == 0;
          gate |= ((int)live & (int)g & (int)(z0 + u < nZ)) ? (1ull << (z0 + u)) : 0ull;
        }
        if (__ballot(redo) != 0ull) {  // some bearing difference needs more than one wrap step: exact loop form
          for (int u = 0; u < GB && z0 + u < nZ; u++) {
            const bool g = pair_gate(P, k, zr[u], zb[u]);
            const unsigned long long bit = 1ull << (z0 + u);
            gate = (live && g) ? (gate | bit) : (gate & ~bit);
          }
        }
      }
    }
    unsigned long long surv = 0;
    for (unsigned long long g = gate; g; g &= g - 1) {
      const int z = __builtin_ctzll(g);
      if (pair_value(P, k, pdw, sZ[2 * z], sZ[2 * z + 1]) != 0.0) surv |= (1ull << z);
    }
    const int cnt = __popcll(surv);
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
      int pos = nSurv + before + off;
      for (unsigned long long g = surv; g; g &= g - 1) {
        const int z = __builtin_ctzll(g);
        if (pos < room) {
          sV[pos] = pair_value(P, k, pdw, sZ[2 * z], sZ[2 * z + 1]);
          sMZ[pos] = ((unsigned)m << 8) | (unsigned)z;
        } else {
          overflow = true;
        }
        pos++;
      }
    }
    __syncthreads();
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
  }
  if (__ballot(overflow) != 0ull && lane == 0) atomicOr(&sMisc[0], 1);
  if (lane == 0) atomicAdd(&sMisc[1], nFov);
  if (wave == 0) sCol[lane] = cs;
  __syncthreads();
  if (sMisc[0] != 0 || nSurv > room) {
    if (tid == 0) atomicOr(B.err, ERRBIT_CAPACITY);
    nSurv = room < 0 ? 0 : (nSurv > room ? room : nSurv);
  }

  // ---------------- phase 2a: normalise in place, count the new Gaussians of every 64-survivor chunk ----------------
  {
    unsigned long long used = 0;
    for (int s0 = wave * 64; s0 < nSurv; s0 += NT) {
      const int sIdx = s0 + lane;
      const bool act = sIdx < nSurv;
      double wn = 0.0;
      int z = 0;
      if (act) { z = (int)(sMZ[sIdx] & 0xffu); wn = sV[sIdx] / sCol[z]; sV[sIdx] = wn; }
      if (act && wn != 0.0) used |= (1ull << z);
      const unsigned long long km = __ballot(act && (wn > 0.0));  // :677
      if (lane == 0) sKeep[s0 >> 6] = __popcll(km);
    }
    used = wave_or_u64(used);
    if (lane == 0) { atomicOr(&sUsed[0], (unsigned)(used & 0xffffffffull)); atomicOr(&sUsed[1], (unsigned)(used >> 32)); }
  }
  __syncthreads();
  // ---------------- phase 2b: emit, dense over the survivors ----------------
  int outBase = nM;
  {
    const int nChunks = (nSurv + 63) >> 6;
    int myBase = nM, c = 0;
    for (int s0 = wave * 64; s0 < nSurv; s0 += NT) {
      for (; c < (s0 >> 6); c++) myBase += sKeep[c];
      const int sIdx = s0 + lane;
      const bool act = sIdx < nSurv;
      unsigned mz = 0;
      double wn = 0.0;
      if (act) { mz = sMZ[sIdx]; wn = sV[sIdx]; }
      const int m = (int)(mz >> 8), z = (int)(mz & 0xffu);
      const bool keep = act && (wn > 0.0);
      const unsigned long long km = __ballot(keep);
      if (keep) {
        const int pos = myBase + __popcll(km & ((1ull << lane) - 1ull));
        const double mx = pMX[m], my = pMY[m], sxx = pSXX[m], sxy = pSXY[m], syy = pSYY[m];
        LmKF k;
        double range;
        lm_precompute(P, pr, mx, my, sxx, sxy, syy, k, range);
        const double nu0 = sZ[2 * z] - k.zx0;
        const double nu1 = wrap_pi(sZ[2 * z + 1] - k.zx1);
        pW[pos] = wn;
        pWP[pos] = 0.0;  // addGaussian: weight_prev = 0 (GaussianMixture.hpp:267-284)
        pMX[pos] = mx + (k.k00 * nu0 + k.k01 * nu1);
        pMY[pos] = my + (k.k10 * nu0 + k.k11 * nu1);
        pSXX[pos] = k.p00;
        pSXY[pos] = k.p01;
        pSYY[pos] = k.p11;
      }
    }
    for (int c2 = 0; c2 < nChunks; c2++) outBase += sKeep[c2];
  }

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
  // The measurement set is wave-uniform and read-only: it is read through the scalar cache (s_load into SGPRs, which
  // VALU instructions take as operands directly) where the index is uniform, and from an LDS copy where lanes index
  // it independently (phases 1b/2).
  double *sZ = reinterpret_cast<double *>(smem_raw);  // [2*MAX_Z], block-shared
  const int wave = threadIdx.x >> 6;
  const int lane = threadIdx.x & 63;
  for (int t = threadIdx.x; t < 2 * nZ; t += WPB * 64) sZ[t] = Zg[t];
  __syncthreads();
  const int i = __builtin_amdgcn_readfirstlane(blockIdx.x * WPB + wave);
  if (i >= B.N) return;
  phd_update_map_particle<UPDMAP_GATE_BATCH>(B, P, cur, nZ, Zg, i, lane, sZ, smem_raw + 2 * RFSGPU_MAX_Z * 8 + (size_t)wave * update_map_lds_bytes_per_wave(B.cap));
}
