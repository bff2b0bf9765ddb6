// birth.h -- map part of RBPHDFilter::predict for BOTH models, with the complete addBirthGaussians logic
// (reference include/RBPHDFilter.hpp:1000-1084): unused measurements are consumed back to front; each is matched against
// the particle's birth candidates (measure() + raw Mahalanobis distance <= SupportDist^2 -> single-measurement KF correct
// (KalmanFilter.hpp:209-259) + support count), otherwise inverseMeasure() creates a candidate that is born immediately
// (CountThreshold == 1 or few landmarks in the FOV) or queued; then the promotion / expiry loop (:1062-1080) including the
// effect of its `it++` on end(): with libstdc++'s circular list that wraps to begin(), so when the LAST candidate is erased
// while others remain, the survivors are visited again.  Then StaticProcessModel::staticStep, Sigma += Q, on every
// Gaussian (ProcessModel.hpp:195-208) -- births get Q added where they are created.
// One wavefront per particle: the candidate list is staged in LDS, the support distance of a measurement to the candidates is
// evaluated 64 at a time across the lanes; the walk itself keeps the reference's order.  All lanes do the Sigma += Q sweep.  (The 2-D immediate-birth case keeps its lane-parallel kernel in merge_prune.h.)
#pragma once
#include "common.h"
#include "vp.h"

// ---- birth-state inheritance after a resampling: RBPHDFilter.hpp:1005-1011 as written -------------------------------------
//   for i = 0 .. N-1:  if (resampleOccured_) { i_prev = particle[i].idParent_;  if (i_prev != i) { unused_measurements_[i] =
//   unused_measurements_[i_prev];  birthGaussians_[i] = birthGaussians_[i_prev]; } }   ... then slot i's own birth step.
// The arrays are per SLOT and i_prev is an ID, so what a slot receives depends on where the walk is: a HIGHER slot i_prev still
// holds what it held before this predict (it may be overwritten later, by its own copy), a LOWER one has already been through
// its birth step (empty unused list, candidates one check older and possibly promoted away).  On the device:
//   level 0 = slots that keep their lists (i_prev == i) or copy from a higher slot.  Their sources are read before anything is
//             written: kernel <0> copies source lists into a staging area indexed by the DESTINATION, kernel <1> commits them
//             (a source may itself be a destination).  Then the birth launch for level 0.
//   level L = slots whose i_prev is a lower slot of level L-1: kernel <2> copies the (final) lists of i_prev, then the birth
//             launch for level L.  Two slots of one level never read each other.
// FOV counts (nLandmarksInFOV_) are never copied by the reference, so they are not here either.
struct BirthLists {
  unsigned long long *unused;
  int *count, *sup, *chk;
  double *mean, *cov;
};
__device__ inline void birth_lists_copy(const BirthLists &S, int s, const BirthLists &Dst, int d) {
  const int nc = S.count[s];
  const size_t sb = (size_t)s * RFSGPU_MAX_CANDIDATES, db = (size_t)d * RFSGPU_MAX_CANDIDATES;
  for (int t = threadIdx.x; t < nc * 3; t += blockDim.x) Dst.mean[db * 3 + t] = S.mean[sb * 3 + t];
  for (int t = threadIdx.x; t < nc * 6; t += blockDim.x) Dst.cov[db * 6 + t] = S.cov[sb * 6 + t];
  for (int t = threadIdx.x; t < nc; t += blockDim.x) { Dst.sup[db + t] = S.sup[sb + t]; Dst.chk[db + t] = S.chk[sb + t]; }
  if (threadIdx.x == 0) { Dst.count[d] = nc; Dst.unused[d] = S.unused[s]; }
}
template <int PHASE>
__global__ __launch_bounds__(128) void birth_inherit_kernel(Buffers B, BirthLists T, const int *parent, const int *level, int L) {
  const int i = blockIdx.x;
  const int p = parent[i];
  if (p == i || level[i] != L) return;
  BirthLists live{B.unusedMask, B.candCount, B.candSup, B.candChk, B.candMean, B.candCov};
  if (PHASE == 0) birth_lists_copy(live, p, T, i);        // level 0, p > i: the source as it is before this predict
  else if (PHASE == 1) birth_lists_copy(T, i, live, i);
  else birth_lists_copy(live, p, live, i);                // level >= 1, p < i: the source after its own birth step
}

template <int D>
struct Cand {
  double x[3];
  double S[6];  // packed xx, xy, xd, yy, yd, dd (2-D uses xx, xy, yy = S[0], S[1], S[3])
};

template <int D>
__device__ void cand_load(const Buffers &B, int i, int c, Cand<D> &k) {
  const double *m = B.candMean + ((size_t)i * RFSGPU_MAX_CANDIDATES + c) * 3;
  const double *s = B.candCov + ((size_t)i * RFSGPU_MAX_CANDIDATES + c) * 6;
  for (int t = 0; t < 3; t++) k.x[t] = m[t];
  for (int t = 0; t < 6; t++) k.S[t] = s[t];
}
template <int D>
__device__ void cand_store(const Buffers &B, int i, int c, const Cand<D> &k) {
  double *m = B.candMean + ((size_t)i * RFSGPU_MAX_CANDIDATES + c) * 3;
  double *s = B.candCov + ((size_t)i * RFSGPU_MAX_CANDIDATES + c) * 6;
  for (int t = 0; t < 3; t++) m[t] = k.x[t];
  for (int t = 0; t < 6; t++) s[t] = k.S[t];
}

// d2 = z_exp.mahalanobisDist2(z) with z_exp ~ N(h(x, candidate), S)   (:1029-1031)
template <int D>
__device__ double cand_support_md2(const Params &P, const PoseReg &pr, const Cand<D> &k, const double *z) {
  if (D == 2) {
    MeasOut mo;
    rb_measure(P, pr, k.x[0], k.x[1], k.S[0], k.S[1], k.S[3], mo);
    double i00, i01, i10, i11, det;
    inv2(mo.s00, mo.s01, mo.s10, mo.s11, i00, i01, i10, i11, det);
    const double e0 = z[0] - mo.z0, e1 = z[1] - mo.z1;
    const double t0 = e0 * i00 + e1 * i10, t1 = e0 * i01 + e1 * i11;
    return t0 * e0 + t1 * e1;
  } else {
    VPMeas o;
    vp_measure(P, pr.x, pr.y, pr.th, k.x[0], k.x[1], k.x[2], k.S[0], k.S[1], k.S[3], k.S[5], o);
    double Si[9];
    inv3(o.S, Si);
    return md2_3(Si, z[0] - o.z0, z[1] - o.z1, z[2] - o.z2);
  }
}

// kfs_[0].correct(x, z, candidate, candidate)  (KalmanFilter.hpp:209-259): in place; unchanged when measure() or the
// innovation gate rejects.
template <int D>
__device__ void cand_correct(const Params &P, const PoseReg &pr, Cand<D> &k, const double *z) {
  if (D == 2) {
    MeasOut mo;
    rb_measure(P, pr, k.x[0], k.x[1], k.S[0], k.S[1], k.S[3], mo);
    if (!mo.inRange) return;
    const double e0 = z[0] - mo.z0;
    if (P.kfRange > 0 && fabs(e0) > P.kfRange) return;  // KalmanFilter_RngBrg: range gate before the wrap
    const double w1 = wrap_pi(z[1] - mo.z1);
    if (P.kfBearing > 0 && fabs(w1) > P.kfBearing) return;
    double i00, i01, i10, i11, det;
    inv2(mo.s00, mo.s01, mo.s10, mo.s11, i00, i01, i10, i11, det);
    const double sxx = k.S[0], sxy = k.S[1], syy = k.S[3];
    const double t00 = sxx * mo.h00 + sxy * mo.h01, t01 = sxx * mo.h10 + sxy * mo.h11;
    const double t10 = sxy * mo.h00 + syy * mo.h01, t11 = sxy * mo.h10 + syy * mo.h11;
    const double k00 = t00 * i00 + t01 * i10, k01 = t00 * i01 + t01 * i11;
    const double k10 = t10 * i00 + t11 * i10, k11 = t10 * i01 + t11 * i11;
    const double kh00 = k00 * mo.h00 + k01 * mo.h10, kh01 = k00 * mo.h01 + k01 * mo.h11;
    const double kh10 = k10 * mo.h00 + k11 * mo.h10, kh11 = k10 * mo.h01 + k11 * mo.h11;
    const double a00 = 1.0 - kh00, a01 = 0.0 - kh01, a10 = 0.0 - kh10, a11 = 1.0 - kh11;
    const double q00 = a00 * sxx + a01 * sxy, q01 = a00 * sxy + a01 * syy;
    const double q10 = a10 * sxx + a11 * sxy, q11 = a10 * sxy + a11 * syy;
    k.x[0] = k.x[0] + (k00 * e0 + k01 * w1);
    k.x[1] = k.x[1] + (k10 * e0 + k11 * w1);
    k.S[0] = (q00 + q00) / 2; k.S[1] = (q01 + q10) / 2; k.S[3] = (q11 + q11) / 2;
  } else {
    Ent3 e;
    e.x = k.x[0]; e.y = k.x[1]; e.d = k.x[2];
    e.xx = k.S[0]; e.xy = k.S[1]; e.xd = k.S[2]; e.yy = k.S[3]; e.yd = k.S[4]; e.dd = k.S[5];
    LmKF3 kf;
    lm_precompute3(P, pr.x, pr.y, pr.th, e, kf);
    double nu0, nu1;
    if (!vp_gate(P, kf, z[0], z[1], nu0, nu1)) return;
    const double nu2 = z[2] - kf.zx2;
    k.x[0] = e.x + ((kf.K[0] * nu0 + kf.K[1] * nu1) + kf.K[2] * nu2);
    k.x[1] = e.y + ((kf.K[3] * nu0 + kf.K[4] * nu1) + kf.K[5] * nu2);
    k.x[2] = e.d + ((kf.K[6] * nu0 + kf.K[7] * nu1) + kf.K[8] * nu2);
    for (int t = 0; t < 6; t++) k.S[t] = kf.p[t];
  }
}

// inverseMeasure(): src/MeasurementModel_RngBrg.cpp:117-136 / src/MeasurementModel_VictoriaPark.cpp:75-102
template <int D>
__device__ void cand_inverse(const Params &P, const PoseReg &pr, const double *z, Cand<D> &k) {
  const double th = (D == 2) ? pr.th : pr.th - RFS_PI / 2;
  const double a = th + z[1];
  const double ca = cos(a), sa = sin(a);
  const double h00 = ca, h01 = -z[0] * sa, h10 = sa, h11 = z[0] * ca;
  const double t00 = h00 * P.R[0] + h01 * P.R[2], t01 = h00 * P.R[1] + h01 * P.R[3];
  const double t10 = h10 * P.R[0] + h11 * P.R[2], t11 = h10 * P.R[1] + h11 * P.R[3];
  for (int t = 0; t < 3; t++) k.x[t] = 0.0;
  for (int t = 0; t < 6; t++) k.S[t] = 0.0;
  k.x[0] = pr.x + z[0] * ca;
  k.x[1] = pr.y + z[0] * sa;
  k.S[0] = t00 * h00 + t01 * h01;
  k.S[1] = t00 * h10 + t01 * h11;
  k.S[3] = t10 * h10 + t11 * h11;
  if (D == 3) { k.x[2] = z[2]; k.S[5] = P.R9[8]; }
}

// GaussianMixture::addGaussian(candidate, birthWeight, true) + this predict's Sigma += Q on it.
template <int D>
__device__ bool birth_append(const Buffers &B, const Params &P, int cur, int i, int &n, const Cand<D> &k) {
  if (n >= B.cap) return false;
  double *slab = B.slab[cur];
  if (D == 2) {
    plane(slab, B.cap, i, PL_W)[n] = P.birthW;
    plane(slab, B.cap, i, PL_WP)[n] = 0.0;
    plane(slab, B.cap, i, PL_MX)[n] = k.x[0];
    plane(slab, B.cap, i, PL_MY)[n] = k.x[1];
    plane(slab, B.cap, i, PL_SXX)[n] = k.S[0] + P.Qlm[0];
    plane(slab, B.cap, i, PL_SXY)[n] = k.S[1] + P.Qlm[1];
    plane(slab, B.cap, i, PL_SYY)[n] = k.S[3] + P.Qlm[2];
  } else {
    plane3(slab, B.cap, i, P3_W)[n] = P.birthW;
    plane3(slab, B.cap, i, P3_WP)[n] = 0.0;
    plane3(slab, B.cap, i, P3_MX)[n] = k.x[0];
    plane3(slab, B.cap, i, P3_MY)[n] = k.x[1];
    plane3(slab, B.cap, i, P3_MD)[n] = k.x[2];
    for (int t = 0; t < 6; t++) plane3(slab, B.cap, i, P3_SXX + t)[n] = k.S[t] + P.Qlm6[t];
  }
  n++;
  return true;
}

// the same without the capacity check / count (the wave-parallel walk keeps n uniform)
template <int D>
__device__ void birth_write(const Buffers &B, const Params &P, int cur, int i, int n, const Cand<D> &k) {
  int nn = n;
  birth_append<D>(B, P, cur, i, nn, k);
}

// LDS image of one particle's candidate list (struct of arrays, list order = index): 80 B per candidate.
template <int D>
struct CandLDS {
  double x[3][RFSGPU_MAX_CANDIDATES];
  double S[6][RFSGPU_MAX_CANDIDATES];
  int sup[RFSGPU_MAX_CANDIDATES], chk[RFSGPU_MAX_CANDIDATES];
  __device__ void get(int c, Cand<D> &k) const {
#pragma unroll
    for (int t = 0; t < 3; t++) k.x[t] = x[t][c];
#pragma unroll
    for (int t = 0; t < 6; t++) k.S[t] = S[t][c];
  }
  __device__ void put(int c, const Cand<D> &k) {
#pragma unroll
    for (int t = 0; t < 3; t++) x[t][c] = k.x[t];
#pragma unroll
    for (int t = 0; t < 6; t++) S[t][c] = k.S[t];
  }
};

// Two kernels share the birth step.  predict_map_general_kernel: one wavefront per particle, candidate c on lane c -- the fast form,
// for every particle whose list stays within the 64 lanes during the call (candidates + unused measurements <= 64: almost all of
// them); it also does the static step of EVERY particle.  predict_map_long_kernel: the same walk on a list staged in LDS
// (<= RFSGPU_MAX_CANDIDATES), 64 candidates at a time, for the particles the first kernel left alone.  (r02: the LDS form alone,
// for everybody, made a birth predict of the Victoria Park run 103 us where the lane form takes 47: 20 KB of LDS per wave.)
template <int D, int WPB>
__global__ __launch_bounds__(WPB * 64) void predict_map_general_kernel(Buffers B, Params P, int cur, int addBirth, int nZprev, BirthLevel LV) {
  const int wave = threadIdx.x >> 6;
  const int lane = threadIdx.x & 63;
  const int i = __builtin_amdgcn_readfirstlane(blockIdx.x * WPB + wave);
  if (i >= B.N) return;
  const int cap = B.cap;
  const int nOld = B.count[i];
  addBirth = addBirth && LV.mine(i);
  // A list that may outgrow the lanes during this call (candidates + unused measurements > 64) is left to predict_map_long_kernel,
  // which the host launches right after this one; the static step below is done here for every particle.
  const bool longList = addBirth && (B.candCount[i] + ((nZprev > 0) ? __popcll(B.unusedMask[i]) : 0) > 64);
  if (addBirth && !longList) {
    // Candidate c of the particle lives on lane c (at most 64 here): loaded once, kept in registers, stored once.
    // The reference's walk over the unused measurements and over the list stays serial where its order is observable
    // (first matching candidate in list order, list order kept by erase, the ++end() wrap); what is per candidate --
    // the support distance of a measurement to every candidate -- runs across the lanes.
    int n = nOld;
    int nc = B.candCount[i];
    const unsigned nfov = (unsigned)B.nInFov[i];
    PoseReg pr;
    load_pose(B, P, i, pr);
    bool fail = false, listFull = false;
    int *supG = B.candSup + (size_t)i * RFSGPU_MAX_CANDIDATES, *chkG = B.candChk + (size_t)i * RFSGPU_MAX_CANDIDATES;
    Cand<D> k;
    for (int t = 0; t < 3; t++) k.x[t] = 0.0;
    for (int t = 0; t < 6; t++) k.S[t] = 0.0;
    int sup = 0, chk = 0;
    if (lane < nc) { cand_load<D>(B, i, lane, k); sup = supG[lane]; chk = chkG[lane]; }
    unsigned long long um = (nZprev > 0) ? B.unusedMask[i] : 0ull;
    while (um) {  // back to front (:1013-1017)
      const int zi = 63 - __builtin_clzll(um);
      um &= ~(1ull << zi);
      const double *z = B.Z + (size_t)D * zi;
      double d2 = 1.0e300;
      if (lane < nc) d2 = cand_support_md2<D>(P, pr, k, z);
      const unsigned long long hit = __ballot(lane < nc && d2 <= P.birthSupportD2);
      if (hit != 0ull) {                                   // the first candidate in list order that supports it
        if (lane == __builtin_ctzll(hit)) { cand_correct<D>(P, pr, k, z); sup++; }
      } else {
        Cand<D> kn;
        cand_inverse<D>(P, pr, z, kn);                     // (the same values on every lane)
        if (P.birthCountThr == 1u || nfov <= P.birthCurThr) {
          if (n < cap) { if (lane == 0) birth_write<D>(B, P, cur, i, n, kn); n++; }
          else fail = true;
        } else if (nc < 64) {   // (always: nc + unused <= 64 on this path)
          if (lane == nc) { k = kn; sup = 1; chk = 0; }
          nc++;
        } else {
          listFull = true;
        }
      }
    }
    // promotion / expiry (:1062-1080) with the ++end() wrap
    int kk = 0;
    while (kk < nc) {
      if (lane == kk) chk++;
      bool atEnd = false;
      for (;;) {
        const unsigned supk = (unsigned)__builtin_amdgcn_readlane(sup, kk), chkk = (unsigned)__builtin_amdgcn_readlane(chk, kk);
        if (!(supk >= P.birthCountThr || chkk > P.birthCheckThr || nfov <= P.birthCurThr)) break;
        if (supk >= P.birthCountThr || nfov <= P.birthCurThr) {
          if (n < cap) { if (lane == kk) birth_write<D>(B, P, cur, i, n, k); n++; }
          else fail = true;
        }
        {  // erase(it): the tail moves down one lane, list order kept
          const int from = (lane >= kk && lane < 63) ? lane + 1 : lane;
#pragma unroll
          for (int t = 0; t < 3; t++) k.x[t] = __shfl(k.x[t], from, 64);
#pragma unroll
          for (int t = 0; t < 6; t++) k.S[t] = __shfl(k.S[t], from, 64);
          sup = __shfl(sup, from, 64);
          chk = __shfl(chk, from, 64);
        }
        nc--;
        if (kk < nc) { if (lane == kk) chk++; }
        else { atEnd = true; break; }
      }
      kk = atEnd ? 0 : kk + 1;
    }
    if (lane < nc) { cand_store<D>(B, i, lane, k); supG[lane] = sup; chkG[lane] = chk; }
    if (lane == 0) {
      B.unusedMask[i] = 0ull;
      B.candCount[i] = nc;
      B.count[i] = n;
      if (fail) atomicOr(B.err, ERRBIT_CAPACITY);
      if (listFull) atomicOr(B.err, ERRBIT_BIRTHLIST);
    }
  }
  // staticStep on the pre-existing Gaussians
  if (!LV.doStatic) return;
  double *slab = B.slab[cur];
  if (D == 2) {
    double *pSXX = plane(slab, cap, i, PL_SXX), *pSXY = plane(slab, cap, i, PL_SXY), *pSYY = plane(slab, cap, i, PL_SYY);
    for (int m = lane; m < nOld; m += 64) { pSXX[m] += P.Qlm[0]; pSXY[m] += P.Qlm[1]; pSYY[m] += P.Qlm[2]; }
  } else {
    for (int t = 0; t < 6; t++) {
      double *p = plane3(slab, cap, i, P3_SXX + t);
      for (int m = lane; m < nOld; m += 64) p[m] += P.Qlm6[t];
    }
  }
}

template <int D, int WPB>
__global__ __launch_bounds__(WPB * 64) void predict_map_long_kernel(Buffers B, Params P, int cur, int nZprev, BirthLevel LV) {
  __shared__ CandLDS<D> sCand[WPB];
  const int wave = threadIdx.x >> 6;
  const int lane = threadIdx.x & 63;
  const int i = __builtin_amdgcn_readfirstlane(blockIdx.x * WPB + wave);
  if (i >= B.N) return;
  if (!LV.mine(i)) return;
  const int cap = B.cap;
  const int nOld = B.count[i];
  // (only the particles predict_map_general_kernel left alone: candidates + unused measurements > one per lane)
  if (B.candCount[i] + ((nZprev > 0) ? __popcll(B.unusedMask[i]) : 0) <= 64) return;
  {
    // The particle's candidate list (<= RFSGPU_MAX_CANDIDATES) is staged in LDS in list order.  The reference's walk over
    // the unused measurements and over the list stays serial where its order is observable (first matching candidate in
    // list order, list order kept by erase, the ++end() wrap); what is per candidate -- the support distance of a
    // measurement to every candidate -- runs across the lanes, 64 candidates at a time.  (Until r02 the list lived on the
    // lanes, one candidate each, which capped it at 64: Victoria Park with artificial clutter and CheckCountThreshold 10
    // keeps ~60 +- 20 candidates per particle, and about half of all seeds overflowed at 5000 particles.)
    CandLDS<D> &L = sCand[wave];
    int n = nOld;
    int nc = B.candCount[i];
    const unsigned nfov = (unsigned)B.nInFov[i];
    PoseReg pr;
    load_pose(B, P, i, pr);
    bool fail = false, listFull = false;
    int *supG = B.candSup + (size_t)i * RFSGPU_MAX_CANDIDATES, *chkG = B.candChk + (size_t)i * RFSGPU_MAX_CANDIDATES;
    for (int c = lane; c < nc; c += 64) {
      Cand<D> k;
      for (int t = 0; t < 3; t++) k.x[t] = 0.0;
      for (int t = 0; t < 6; t++) k.S[t] = 0.0;
      cand_load<D>(B, i, c, k);
      L.put(c, k);
      L.sup[c] = supG[c];
      L.chk[c] = chkG[c];
    }
    wave_sync();
    unsigned long long um = (nZprev > 0) ? B.unusedMask[i] : 0ull;
    while (um) {  // back to front (:1013-1017)
      const int zi = 63 - __builtin_clzll(um);
      um &= ~(1ull << zi);
      const double *z = B.Z + (size_t)D * zi;
      int first = -1;                                       // the first candidate in list order that supports it
      for (int c0 = 0; c0 < nc && first < 0; c0 += 64) {
        const int c = c0 + lane;
        double d2 = 1.0e300;
        if (c < nc) {
          Cand<D> k;
          for (int t = 0; t < 3; t++) k.x[t] = 0.0;
          for (int t = 0; t < 6; t++) k.S[t] = 0.0;
          L.get(c, k);
          d2 = cand_support_md2<D>(P, pr, k, z);
        }
        const unsigned long long hit = __ballot(c < nc && d2 <= P.birthSupportD2);
        if (hit != 0ull) first = c0 + __builtin_ctzll(hit);
      }
      if (first >= 0) {
        if (lane == 0) {
          Cand<D> k;
          for (int t = 0; t < 3; t++) k.x[t] = 0.0;
          for (int t = 0; t < 6; t++) k.S[t] = 0.0;
          L.get(first, k);
          cand_correct<D>(P, pr, k, z);
          L.put(first, k);
          L.sup[first]++;
        }
        wave_sync();
      } else {
        Cand<D> kn;
        cand_inverse<D>(P, pr, z, kn);                     // (the same values on every lane)
        if (P.birthCountThr == 1u || nfov <= P.birthCurThr) {
          if (n < cap) { if (lane == 0) birth_write<D>(B, P, cur, i, n, kn); n++; }
          else fail = true;
        } else if (nc < RFSGPU_MAX_CANDIDATES) {
          if (lane == 0) { L.put(nc, kn); L.sup[nc] = 1; L.chk[nc] = 0; }
          nc++;
          wave_sync();
        } else {
          listFull = true;
        }
      }
    }
    // promotion / expiry (:1062-1080) with the ++end() wrap
    int kk = 0;
    while (kk < nc) {
      if (lane == 0) L.chk[kk]++;
      wave_sync();
      bool atEnd = false;
      for (;;) {
        const unsigned supk = (unsigned)L.sup[kk], chkk = (unsigned)L.chk[kk];
        if (!(supk >= P.birthCountThr || chkk > P.birthCheckThr || nfov <= P.birthCurThr)) break;
        if (supk >= P.birthCountThr || nfov <= P.birthCurThr) {
          if (n < cap) {
            if (lane == 0) {
              Cand<D> k;
              for (int t = 0; t < 3; t++) k.x[t] = 0.0;
              for (int t = 0; t < 6; t++) k.S[t] = 0.0;
              L.get(kk, k);
              birth_write<D>(B, P, cur, i, n, k);
            }
            n++;
          } else fail = true;
        }
        // erase(it): the tail moves down one place, list order kept (64 entries at a time, ascending: every lane reads its
        // source before any lane of the same step writes, and a step only writes places the earlier steps have read)
        for (int c0 = kk; c0 < nc - 1; c0 += 64) {
          const int c = c0 + lane;
          const bool mv = c < nc - 1;
          Cand<D> k;
          for (int t = 0; t < 3; t++) k.x[t] = 0.0;
          for (int t = 0; t < 6; t++) k.S[t] = 0.0;
          int su = 0, ch = 0;
          if (mv) { L.get(c + 1, k); su = L.sup[c + 1]; ch = L.chk[c + 1]; }
          wave_sync();
          if (mv) { L.put(c, k); L.sup[c] = su; L.chk[c] = ch; }
          wave_sync();
        }
        nc--;
        if (kk < nc) { if (lane == 0) L.chk[kk]++; wave_sync(); }
        else { atEnd = true; break; }
      }
      kk = atEnd ? 0 : kk + 1;
    }
    for (int c = lane; c < nc; c += 64) {
      Cand<D> k;
      for (int t = 0; t < 3; t++) k.x[t] = 0.0;
      for (int t = 0; t < 6; t++) k.S[t] = 0.0;
      L.get(c, k);
      cand_store<D>(B, i, c, k);
      supG[c] = L.sup[c];
      chkG[c] = L.chk[c];
    }
    if (lane == 0) {
      B.unusedMask[i] = 0ull;
      B.candCount[i] = nc;
      B.count[i] = n;
      if (fail) atomicOr(B.err, ERRBIT_CAPACITY);
      if (listFull) atomicOr(B.err, ERRBIT_BIRTHLIST);
    }
  }
}

// StaticProcessModel::staticStep (include/ProcessModel.hpp:195-208), Sigma += Q on every Gaussian, for a RUN of predicts in one
// launch: a predict that adds no births (every odometry message but the first after an update) is a static step and nothing
// else, and the additions of a run made one after the other inside one kernel give the same bits as one launch per predict.
// Every step of the run carries its own Q (the drivers scale it with the message interval).  One wavefront per particle.
#define STATIC_RUN_MAX 16
struct StaticRun {
  int n;
  double q[STATIC_RUN_MAX][6];   // packed: 2-D xx, xy, yy in [0..2]; 3-D xx, xy, xd, yy, yd, dd
};
template <int D, int WPB>
__global__ __launch_bounds__(WPB * 64) void static_steps_kernel(Buffers B, int cur, StaticRun R) {
  const int wave = threadIdx.x >> 6;
  const int lane = threadIdx.x & 63;
  const int i = __builtin_amdgcn_readfirstlane(blockIdx.x * WPB + wave);
  if (i >= B.N) return;
  const int cap = B.cap, n = B.count[i];
  double *slab = B.slab[cur];
  constexpr int NC = (D == 2) ? 3 : 6;
  for (int t = 0; t < NC; t++) {
    double *p = (D == 2) ? plane(slab, cap, i, PL_SXX + t) : plane3(slab, cap, i, P3_SXX + t);
    for (int m = lane; m < n; m += 64) {
      double a = p[m];
      for (int r = 0; r < R.n; r++) a += R.q[r][t];
      p[m] = a;
    }
  }
}
