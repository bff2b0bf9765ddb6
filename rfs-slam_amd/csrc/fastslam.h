// fastslam.h -- FastSLAM 1.0 map update on the RB-PHD engine's state (reference include/FastSLAM.hpp:424-706, one
// data-association hypothesis; several hypotheses per particle: fastslam_mh.h).  A particle's mixture is its landmark map, a Gaussian's weight the landmark's log-odds of
// existence, the birth-candidate lists are the landmark candidates.  Both measurement models (template parameter D).
//
// Three kernels per update (one wavefront per particle):
//  fs_associate_update   in-range landmarks (ballot-compacted, :440-449) -> the log-likelihood table (:468-481) kept
//                        SPARSE: only entries above the floor are stored (the reference fills an nMZ x nMZ table with the
//                        floor) -> CostMatrix::reduce (src/CostMatrix.cpp:263-340): a pairing that is the only possibility
//                        of its row and of its column is fixed; what remains ambiguous (rows and columns with competing
//                        possibilities) falls apart into connected components, each solved by inspection (1 x n, n x 1) or
//                        by the Hungarian method (murty.h) on a small dense block padded with the floor -> Kalman correction of the associated landmarks, existence log-odds (:573-604),
//                        particle weight *= exp(sum of the associated log-likelihoods) (:696-697).
//                        The reference runs the Hungarian method on the whole reduced table, rows and columns without any
//                        possibility included; those can only take floor-valued cells, which never trigger an update
//                        (:584), so the associations that matter are the maximum-weight matching of the ambiguous part --
//                        identical unless two associations tie exactly.
//  gm_prune<holes off>   GaussianMixture::prune(mapExistencePruneThreshold) (:611-612).
//  fs_new_landmarks      unassociated measurements -> candidates / new landmarks and the promotion loop (:615-690), a short
//                        serial walk on lane 0 (cf. birth.h).
#pragma once
#define FS_LANE_CANDIDATES 64   // landmark candidates per particle on the FastSLAM path (one per lane); the storage stride is RFSGPU_MAX_CANDIDATES
#include "common.h"
#include "murty.h"
#include "birth.h"

struct FsParams {
  double prior;        // landmarkExistencePrior_
  double minLog;       // minLogMeasurementLikelihood_
  double lockW;        // landmarkLockWeight_
  double pfa;          // clutterIntensityIntegral(nZ) / nZ   (:561-562)
  double newW;         // log(prior / (1 - prior))            (:614)
  double supportD2;    // landmarkCandidateMeasurementSupportDist_^2
  unsigned countThr, curThr, checkThr;  // landmarkCandidateMeasurementCountThreshold_ / CurrentMeasurementCountThreshold_ / CheckThreshold_
};

#define FS_AMBIG_MAX 64    // rows / columns of ONE connected component of competing associations (the in-kernel Hungarian, MURTY_N)
#define FS_AMBIG_ROWS 256  // rows with competing associations per particle, over all components
#define FS_SMALL 12        // components up to this size are solved in LDS, larger ones in the particle's HBM scratch

// per-particle scratch of the Hungarian method in HBM: the dense block (sized with room for per-row / per-column work arrays)
__host__ __device__ inline size_t fs_arena_bytes_n(int n) {
  return ((size_t)n * n * 8 + 3 * (size_t)n * 8 + 6 * (size_t)n * 4 + 8 * (size_t)n + 63) & ~(size_t)63;
}
__host__ __device__ inline size_t fs_arena_bytes() { return fs_arena_bytes_n(FS_AMBIG_MAX); }
__device__ inline void fs_arena_carve(unsigned char *base, MurtyArena &A, unsigned char *&soln, int n) {
  unsigned char *p = base;
  A.Ct = (double *)p; p += (size_t)n * n * 8;
  A.lx = (double *)p; p += n * 8;
  A.ly = (double *)p; p += n * 8;
  A.slack = (double *)p; p += n * 8;
  A.xy = (int *)p; p += n * 4;
  A.yx = (int *)p; p += n * 4;
  A.p = (int *)p; p += 2 * n * 4;
  A.queue = (int *)p; p += 2 * n * 4;
  A.S = p; p += n;
  A.T = p; p += n;
  A.NS = p; p += n;
  A.xq = p; p += n;
  A.yq = p; p += n;
  soln = p;
  A.nodeScore = nullptr; A.nodeParent = nullptr; A.heap = nullptr; A.nodeId = nullptr; A.nodeA = nullptr;
}

// LDS per wave: in-range list (index, Pd), assignment, row segment, log-weight contribution; sparse table (value, (row<<8)|z)
__host__ __device__ inline size_t fs_lds_bytes_per_wave(int cap) {
  return (size_t)cap * (2 + 8 + 2 + 4 + 8) + (size_t)2 * cap * (8 + 4) + 64 * 4 + (size_t)FS_AMBIG_ROWS * (2 + 8) + fs_arena_bytes_n(FS_SMALL) + FS_AMBIG_MAX * 2 + 64 + 64;
}

// Per-landmark quantities of one table row for either model (D = 2: range-bearing, D = 3: Victoria Park).
template <int D>
struct FsRow {
  bool valid;      // measure() returned true (:472)
  double pd;
  bool close;
  // D == 2
  MeasOut mo;
  double i00, i01, i10, i11, factor;
  // D == 3
  LmKF3 k3;
  double lf;       // log(factor)
};
template <int D>
__device__ __forceinline__ void fs_row(const Buffers &B, const Params &P, const PoseReg &pr, const double *slab, int cap, int i, int m, bool act,
                                       FsRow<D> &r, unsigned char *pdScratch = nullptr) {  // pdScratch: VP_PD_SCRATCH_BYTES of the wave's LDS (D == 3)
  if (D == 2) {
    double mx = 0, my = 0, sxx = 1, sxy = 0, syy = 1;
    if (act) {
      mx = plane((double *)slab, cap, i, PL_MX)[m]; my = plane((double *)slab, cap, i, PL_MY)[m];
      sxx = plane((double *)slab, cap, i, PL_SXX)[m]; sxy = plane((double *)slab, cap, i, PL_SXY)[m]; syy = plane((double *)slab, cap, i, PL_SYY)[m];
    }
    rb_measure(P, pr, mx, my, sxx, sxy, syy, r.mo);
    r.valid = r.mo.inRange;
    r.pd = rb_pd(P, r.mo.range, r.close);
    double det;
    inv2(r.mo.s00, r.mo.s01, r.mo.s10, r.mo.s11, r.i00, r.i01, r.i10, r.i11, det);
    r.factor = pdf_factor2(det);
    r.lf = log(r.factor);
  } else {
    Ent3 e;
    e.w = 0; e.x = 10; e.y = 10; e.d = 1; e.xx = 1; e.xy = 0; e.xd = 0; e.yy = 1; e.yd = 0; e.dd = 1;
    if (act) load_ent3(slab, cap, i, m, e, false);
    r.valid = true;  // MeasurementModel_VictoriaPark::measure always returns true
    // all 64 lanes call this together: the shifted-copy evaluations are shared out over the wave (vp.h)
    r.pd = vp_pd_wave(P, B.scan, B.nScan, pr.x, pr.y, pr.th, e, act, r.close, pdScratch);
    if (!act) { r.pd = 0.0; r.close = false; }
    lm_precompute3(P, pr.x, pr.y, pr.th, e, r.k3);
    r.factor = r.k3.factor;
    r.lf = log(r.factor);
  }
}
// One table cell: fmax(floor, log(N(z; z_exp, S)))  (:476-477; evalGaussianLikelihood uses the RAW difference, NaN -> 0)
template <int D>
__device__ __forceinline__ double fs_cell_d(const FsRow<D> &r, const double *z, double lim) {
  double md2;
  if (D == 2) {
    const double e0 = z[0] - r.mo.z0, e1 = z[1] - r.mo.z1;
    const double t0 = e0 * r.i00 + e1 * r.i10, t1 = e0 * r.i01 + e1 * r.i11;
    md2 = t0 * e0 + t1 * e1;
  } else {
    md2 = md2_3(r.k3.Si, z[0] - r.k3.zx0, z[1] - r.k3.zx1, z[2] - r.k3.zx2);
  }
  if (-0.5 * md2 - r.lf < lim - 1.0) return lim;  // cheap bound: clearly below the floor
  double l = exp(-0.5 * md2) / r.factor;
  if (l != l) l = 0.0;
  return fmax(lim, log(l));
}

// kfs_.correct(pose, Z[z], lm, lm) (KalmanFilter.hpp:209-259) on landmark m of particle slot i, in place; true when the
// correction was performed (measure() valid and the innovation inside the gates).
template <int D>
__device__ __forceinline__ bool fs_kf_correct(const Params &P, const PoseReg &pr, double *slab, int cap, int i, int m, const double *sZ, int z) {
  bool upd = false;
  if (D == 2) {
    double *pMX = plane(slab, cap, i, PL_MX), *pMY = plane(slab, cap, i, PL_MY);
    double *pSXX = plane(slab, cap, i, PL_SXX), *pSXY = plane(slab, cap, i, PL_SXY), *pSYY = plane(slab, cap, i, PL_SYY);
    const double mx = pMX[m], my = pMY[m], sxx = pSXX[m], sxy = pSXY[m], syy = pSYY[m];
    MeasOut mo;
    rb_measure(P, pr, mx, my, sxx, sxy, syy, mo);
    const double e0 = sZ[2 * z] - mo.z0;
    const double w1 = wrap_pi(sZ[2 * z + 1] - mo.z1);
    const bool gateR = !(P.kfRange > 0 && fabs(e0) > P.kfRange), gateB = !(P.kfBearing > 0 && fabs(w1) > P.kfBearing);
    if (mo.inRange && gateR && gateB) {
      double i00, i01, i10, i11, det;
      inv2(mo.s00, mo.s01, mo.s10, mo.s11, i00, i01, i10, i11, det);
      const double t00 = sxx * mo.h00 + sxy * mo.h01, t01 = sxx * mo.h10 + sxy * mo.h11;
      const double t10 = sxy * mo.h00 + syy * mo.h01, t11 = sxy * mo.h10 + syy * mo.h11;
      const double k00 = t00 * i00 + t01 * i10, k01 = t00 * i01 + t01 * i11;
      const double k10 = t10 * i00 + t11 * i10, k11 = t10 * i01 + t11 * i11;
      const double kh00 = k00 * mo.h00 + k01 * mo.h10, kh01 = k00 * mo.h01 + k01 * mo.h11;
      const double kh10 = k10 * mo.h00 + k11 * mo.h10, kh11 = k10 * mo.h01 + k11 * mo.h11;
      const double a00 = 1.0 - kh00, a01 = 0.0 - kh01, a10 = 0.0 - kh10, a11 = 1.0 - kh11;
      const double q00 = a00 * sxx + a01 * sxy, q01 = a00 * sxy + a01 * syy;
      const double q10 = a10 * sxx + a11 * sxy, q11 = a10 * sxy + a11 * syy;
      pMX[m] = mx + (k00 * e0 + k01 * w1);
      pMY[m] = my + (k10 * e0 + k11 * w1);
      pSXX[m] = (q00 + q00) / 2;
      pSXY[m] = (q01 + q10) / 2;
      pSYY[m] = (q11 + q11) / 2;
      upd = true;
    }
  } else {
    Ent3 e;
    load_ent3(slab, cap, i, m, e, false);
    LmKF3 kf;
    lm_precompute3(P, pr.x, pr.y, pr.th, e, kf);
    double nu0, nu1;
    if (vp_gate(P, kf, sZ[3 * z], sZ[3 * z + 1], nu0, nu1)) {  // KalmanFilter_VictoriaPark::calculateInnovation
      const double nu2 = sZ[3 * z + 2] - kf.zx2;
      plane3(slab, cap, i, P3_MX)[m] = e.x + ((kf.K[0] * nu0 + kf.K[1] * nu1) + kf.K[2] * nu2);
      plane3(slab, cap, i, P3_MY)[m] = e.y + ((kf.K[3] * nu0 + kf.K[4] * nu1) + kf.K[5] * nu2);
      plane3(slab, cap, i, P3_MD)[m] = e.d + ((kf.K[6] * nu0 + kf.K[7] * nu1) + kf.K[8] * nu2);
      for (int t = 0; t < 6; t++) plane3(slab, cap, i, P3_SXX + t)[m] = kf.p[t];
      upd = true;
    }
  }
  return upd;
}
// The existence log-odds of one in-range landmark after the association step (:586-603).
__device__ __forceinline__ double fs_existence_step(const FsParams &F, double w, double pd, bool upd) {
  double pe;
  if (upd) {
    pe = ((1 - pd) * F.pfa * F.prior + pd * F.prior) / (F.pfa + (1 - F.pfa) * pd * F.prior);
  } else {
    pe = ((1 - pd) * F.prior) / ((1 - F.prior) + (1 - pd) * F.prior);
    if (w > F.lockW) pe = 0.5;
  }
  return w + log(pe / (1 - pe));
}

template <int WPB, int D>
__global__ __launch_bounds__(WPB * 64) void fs_associate_update_kernel(Buffers B, Params P, FsParams F, int cur, int nZ, unsigned char *arena) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  __shared__ __align__(16) unsigned char sPdScratch[WPB][(D == 3) ? ((VP_PD_SCRATCH_BYTES + 15) & ~15) : 16];
  double *sZ = reinterpret_cast<double *>(smem_raw);
  const int wave = threadIdx.x >> 6;
  const int lane = threadIdx.x & 63;
  for (int t = threadIdx.x; t < D * nZ; t += WPB * 64) sZ[t] = B.Z[t];
  __syncthreads();
  const int i = __builtin_amdgcn_readfirstlane(blockIdx.x * WPB + wave);
  if (i >= B.N) return;
  const int cap = B.cap, LCAP = 2 * cap;
  unsigned char *wb = smem_raw + 3 * RFSGPU_MAX_Z * 8 + (size_t)wave * fs_lds_bytes_per_wave(cap);
  double *sPd = reinterpret_cast<double *>(wb);                    // [cap] Pd of in-range row k
  double *sC = sPd + cap;                                           // [cap] log-weight contribution of row k
  double *sMV = sC + cap;                                           // [LCAP] table values above the floor, (row, z) order
  unsigned *sMZ = reinterpret_cast<unsigned *>(sMV + LCAP);         // [LCAP] (row << 8) | z
  unsigned *sSeg = sMZ + LCAP;                                      // [cap] (start << 8) | count of row k's cells
  int *sColCnt = reinterpret_cast<int *>(sSeg + cap);               // [64] cells above the floor per measurement
  unsigned short *sIdx = reinterpret_cast<unsigned short *>(sColCnt + 64);  // [cap] mixture index of row k
  short *sDa = reinterpret_cast<short *>(sIdx + cap);               // [cap] associated measurement of row k, or -1
  unsigned short *sAR = reinterpret_cast<unsigned short *>(sDa + cap);      // [FS_AMBIG_ROWS] rows with competing associations
  // 8-byte aligned from here: cap*(8+8) + LCAP*12 + cap*4 + 256 + cap*4 + ROWS*2 is a multiple of 8 (cap is a multiple of 64)
  unsigned long long *sRM = reinterpret_cast<unsigned long long *>(sAR + FS_AMBIG_ROWS);  // [FS_AMBIG_ROWS] their measurement masks
  double *sHL = reinterpret_cast<double *>(sRM + FS_AMBIG_ROWS);            // Hungarian scratch for small components
  unsigned short *sRows = reinterpret_cast<unsigned short *>(reinterpret_cast<unsigned char *>(sHL) + fs_arena_bytes_n(FS_SMALL));  // [FS_AMBIG_MAX]
  unsigned char *sParent = reinterpret_cast<unsigned char *>(sRows + FS_AMBIG_MAX);                                          // [64]

  const int nM = B.count[i];
  const unsigned long long zmask = (nZ >= 64) ? ~0ull : ((1ull << nZ) - 1ull);
  const unsigned long long lt = (1ull << lane) - 1ull;
  double *slab = B.slab[cur];
  double *pW = slab + ((size_t)i * B.npl + 0) * cap, *pWP = slab + ((size_t)i * B.npl + 1) * cap;  // planes 0 / 1 in both layouts
  PoseReg pr;
  load_pose(B, P, i, pr);
  const double lim = F.minLog;

  // ---- A. in-range rows and the cells of the table above the floor ----
  int nIn = 0, nList = 0;
  bool overflow = false;
  sColCnt[lane] = 0;
  wave_sync();
  for (int c0 = 0; c0 < nM; c0 += 64) {
    const int m = c0 + lane;
    const bool act = m < nM;
    FsRow<D> row;
    fs_row<D>(B, P, pr, slab, cap, i, m, act, row, sPdScratch[wave]);
    const double pd = row.pd;
    const bool inR = act && (pd != 0 || row.close);  // :446
    unsigned long long cells = 0;
    if (inR && row.valid)
      for (int z = 0; z < nZ; z++)
        if (fs_cell_d<D>(row, sZ + D * z, lim) > lim) cells |= 1ull << z;
    const unsigned long long im = __ballot(inR);
    const int k = nIn + __popcll(im & lt);
    const int cnt = __popcll(cells);
    const int off = wave_excl_scan(cnt, lane);
    const int total = __builtin_amdgcn_readlane(off + cnt, 63);
    if (inR) {
      sIdx[k] = (unsigned short)m;
      sPd[k] = pd;
      sSeg[k] = ((unsigned)(nList + off) << 8) | (unsigned)cnt;
      sDa[k] = -1;
      int pos = nList + off;
      for (unsigned long long g = cells; g; g &= g - 1) {
        const int z = __builtin_ctzll(g);
        if (pos < LCAP) {
          sMV[pos] = fs_cell_d<D>(row, sZ + D * z, lim);
          sMZ[pos] = ((unsigned)k << 8) | (unsigned)z;
          atomicAdd(&sColCnt[z], 1);
        } else {
          overflow = true;
        }
        pos++;
      }
    }
    nIn += __popcll(im);
    nList += total;
  }
  if (__ballot(overflow) != 0ull) {
    if (lane == 0) atomicOr(B.err, ERRBIT_CAPACITY);
    nList = LCAP;
  }
  wave_sync();

  // ---- B. CostMatrix::reduce: fix what is unambiguous, collect the ambiguous rows (with their column masks) ----
  int nRa = 0;
  bool tooBig = false;
  for (int k0 = 0; k0 < nIn; k0 += 64) {
    const int k = k0 + lane;
    bool amb = false;
    unsigned long long rmask = 0;
    if (k < nIn) {
      const unsigned seg = sSeg[k];
      const int st = (int)(seg >> 8), cnt = (int)(seg & 0xffu);
      if (cnt == 1) {
        const int z = (int)(sMZ[st] & 0xffu);
        if (sColCnt[z] == 1) sDa[k] = (short)z;  // the only possibility of the row and of the column
        else amb = true;
      } else if (cnt > 1) {
        amb = true;
      }
      if (amb)
        for (int q = st; q < st + cnt && q < nList; q++) rmask |= 1ull << (sMZ[q] & 0xffu);
    }
    const unsigned long long am = __ballot(amb);
    if (amb) {
      const int a = nRa + __popcll(am & lt);
      if (a < FS_AMBIG_ROWS) { sAR[a] = (unsigned short)k; sRM[a] = rmask; }
      else tooBig = true;
    }
    nRa += __popcll(am);
  }
  if (__ballot(tooBig) != 0ull) {
    if (lane == 0) atomicOr(B.err, ERRBIT_MURTY);  // more competing rows than the kernel lists: refuse
    nRa = 0;
  }
  wave_sync();

  // ---- C. the ambiguous part falls apart into connected components (rows linked by shared measurements); each is a
  //         small assignment problem: dense block padded with the floor -> the wave-parallel Hungarian method
  //         (hungarian_wave.h: same traversal order, tolerances and tie-breaks as the reference's solver).  The component
  //         walk itself is wave-uniform: every lane follows the same scalar control flow over the LDS lists, lane 0 does
  //         the (tiny) serial bookkeeping, lanes x < nR fill and read row x of the block. ----
  if (nRa > 0) {
    unsigned char *parent = sParent;  // (LDS: private arrays indexed at run time would live in scratch memory)
    if (lane == 0) {  // union-find over the <= 64 measurement columns, left fully compressed
      for (int z = 0; z < 64; z++) parent[z] = (unsigned char)z;
      auto find = [&](int z) { while (parent[z] != z) { parent[z] = parent[parent[z]]; z = parent[z]; } return z; };
      for (int a = 0; a < nRa; a++) {
        const unsigned long long m = sRM[a];
        const int r0 = find(__builtin_ctzll(m));
        for (unsigned long long g = m & (m - 1); g; g &= g - 1) {
          const int r1 = find(__builtin_ctzll(g));
          if (r1 != r0) parent[r1] = (unsigned char)r0;
        }
      }
      for (int z = 0; z < 64; z++) parent[z] = (unsigned char)find(z);
    }
    wave_sync();
    unsigned long long colAmb = 0;
    for (int a = 0; a < nRa; a++) colAmb |= sRM[a];
    double *CtG = reinterpret_cast<double *>(arena + (size_t)i * fs_arena_bytes());   // [64 x 64] in the particle's HBM scratch
    double *CtL = sHL;                                                                // [FS_SMALL x FS_SMALL] in LDS
    for (unsigned long long roots = colAmb; roots; roots &= roots - 1) {
      const int r = __builtin_ctzll(roots);
      if (parent[r] != r) continue;
      unsigned long long cmask = 0;
      for (unsigned long long g = colAmb; g; g &= g - 1) { const int z = __builtin_ctzll(g); if (parent[z] == r) cmask |= 1ull << z; }
      unsigned short *rows = sRows;
      int nR = 0;
      bool big = false;
      wave_sync();  // (the previous component's readers of `rows` are done)
      for (int a = 0; a < nRa; a++)
        if (sRM[a] & cmask) { if (nR < FS_AMBIG_MAX) { if (lane == 0) rows[nR] = (unsigned short)a; nR++; } else big = true; }
      wave_sync();
      const int nC = __popcll(cmask);
      if (big) { if (lane == 0) atomicOr(B.err, ERRBIT_MURTY); continue; }  // a component beyond the in-kernel Hungarian's size: refuse
      if (nR == 1) {  // one landmark, several measurements: the best cell (what the Hungarian optimum is, ties aside)
        if (lane == 0) {
          const unsigned seg = sSeg[sAR[rows[0]]];
          const int st = (int)(seg >> 8), cnt = (int)(seg & 0xffu);
          double bv = lim;
          int bz = -1;
          for (int q = st; q < st + cnt && q < nList; q++)
            if (sMV[q] > bv) { bv = sMV[q]; bz = (int)(sMZ[q] & 0xffu); }
          if (bz >= 0) sDa[sAR[rows[0]]] = (short)bz;
        }
        continue;
      }
      if (nC == 1) {  // several landmarks, one measurement: the landmark with the best cell takes it
        if (lane == 0) {
          double bv = lim;
          int bx = -1;
          for (int x = 0; x < nR; x++) {
            const unsigned seg = sSeg[sAR[rows[x]]];
            const double v = sMV[seg >> 8];  // (the row's only cell)
            if (v > bv) { bv = v; bx = x; }
          }
          if (bx >= 0) sDa[sAR[rows[bx]]] = (short)__builtin_ctzll(cmask);
        }
        continue;
      }
      const int nA = nR > nC ? nR : nC;
      double *Ct = (nA <= FS_SMALL) ? CtL : CtG;
      for (int t = lane; t < nA * nA; t += 64) Ct[t] = lim;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      wave_sync();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      if (lane < nR) {
        const unsigned seg = sSeg[sAR[rows[lane]]];
        const int st = (int)(seg >> 8), cnt = (int)(seg & 0xffu);
        for (int q = st; q < st + cnt && q < nList; q++) {
          const int z = (int)(sMZ[q] & 0xffu);
          Ct[lane * nA + __popcll(cmask & ((1ull << z) - 1ull))] = sMV[q];
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      wave_sync();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      double cost;
      int xy = -1;
      if (hungarian_wave(Ct, nA, nA, xy, &cost, nullptr)) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        if (lane < nR) {
          const int bcol = xy;
          if (bcol >= 0 && bcol < nC && Ct[lane * nA + bcol] > lim) sDa[sAR[rows[lane]]] = (short)nth_bit(cmask, bcol);
        }
      } else {
        if (lane == 0) atomicOr(B.err, ERRBIT_MURTY);  // the reference would leave the particle untouched (:511-515); refused loudly here
      }
    }
  }
  wave_sync();

  // ---- D. Kalman correction of the associated landmarks, existence log-odds (:573-604) ----
  unsigned long long used = 0;
  int nUpd = 0;
  for (int k0 = 0; k0 < nIn; k0 += 64) {
    const int k = k0 + lane;
    bool upd = false;
    double val = 0.0;
    if (k < nIn) {
      const int m = sIdx[k];
      const int z = sDa[k];
      const double pd = sPd[k];
      const double w = pW[m];
      if (z >= 0) {
        const unsigned seg = sSeg[k];
        const int st = (int)(seg >> 8), cnt = (int)(seg & 0xffu);
        bool found = false;
        for (int q = st; q < st + cnt && q < nList; q++)
          if ((int)(sMZ[q] & 0xffu) == z) { val = sMV[q]; found = true; }
        if (found) upd = fs_kf_correct<D>(P, pr, slab, cap, i, m, sZ, z);  // likelihoodTable[m][z] > floor (:584)
      }
      if (upd) used |= 1ull << z;
      pWP[m] = w;
      pW[m] = fs_existence_step(F, w, pd, upd);
      sC[k] = upd ? val : 0.0;
    }
    nUpd += __popcll(__ballot(upd));
  }
  used = wave_or_u64(used);
  wave_sync();
  if (lane == 0) {
    double logw = 0.0;  // the reference adds the associated cells in landmark order (:588)
    for (int k = 0; k < nIn; k++) logw += sC[k];  // (rows without an update hold 0)
    B.weight[i] = B.weight[i] * exp(logw);
    B.unusedMask[i] = (~used) & zmask;
    B.nInFov[i] = nUpd;
  }
}

// GaussianMixture::addGaussian(candidate, w, true) (:267-284) -- no process noise here (the update adds none)
template <int D>
__device__ inline bool fs_append(const Buffers &B, int cur, int i, int &n, const Cand<D> &k, double w) {
  if (n >= B.cap) return false;
  double *slab = B.slab[cur];
  if (D == 2) {
    plane(slab, B.cap, i, PL_W)[n] = w;
    plane(slab, B.cap, i, PL_WP)[n] = 0.0;
    plane(slab, B.cap, i, PL_MX)[n] = k.x[0];
    plane(slab, B.cap, i, PL_MY)[n] = k.x[1];
    plane(slab, B.cap, i, PL_SXX)[n] = k.S[0];
    plane(slab, B.cap, i, PL_SXY)[n] = k.S[1];
    plane(slab, B.cap, i, PL_SYY)[n] = k.S[3];
  } else {
    plane3(slab, B.cap, i, P3_W)[n] = w;
    plane3(slab, B.cap, i, P3_WP)[n] = 0.0;
    plane3(slab, B.cap, i, P3_MX)[n] = k.x[0];
    plane3(slab, B.cap, i, P3_MY)[n] = k.x[1];
    plane3(slab, B.cap, i, P3_MD)[n] = k.x[2];
    for (int t = 0; t < 6; t++) plane3(slab, B.cap, i, P3_SXX + t)[n] = k.S[t];
  }
  n++;
  return true;
}

// New landmarks from the measurements no landmark took (:615-690): one wavefront per particle, landmark candidate c on lane c
// (as in birth.h: the support distance of a measurement to all candidates is one lane-parallel evaluation, the walk keeps the
// reference's order -- measurements in index order, first supporting candidate in list order, the promotion loop once per
// unassociated measurement with its ++end() wrap).
#define FS_NEWLM_WPB 4
template <int D>
__global__ __launch_bounds__(64 * FS_NEWLM_WPB) void fs_new_landmarks_kernel(Buffers B, Params P, FsParams F, int cur, int nZ) {
  const int wave = threadIdx.x >> 6;
  const int lane = threadIdx.x & 63;
  const int i = __builtin_amdgcn_readfirstlane(blockIdx.x * FS_NEWLM_WPB + wave);
  if (i >= B.N) return;
  int n = B.count[i];
  int nc = B.candCount[i];
  const int cap = B.cap;
  const unsigned nfov = (unsigned)B.nInFov[i];
  PoseReg pr;
  load_pose(B, P, i, pr);
  bool fail = false, listFull = false;
  int *supG = B.candSup + (size_t)i * RFSGPU_MAX_CANDIDATES, *chkG = B.candChk + (size_t)i * RFSGPU_MAX_CANDIDATES;
  Cand<D> k;
  for (int t = 0; t < 3; t++) k.x[t] = 0.0;
  for (int t = 0; t < 6; t++) k.S[t] = 0.0;
  int sup = 0, chk = 0;
  if (nc > FS_LANE_CANDIDATES) { listFull = true; nc = FS_LANE_CANDIDATES; }   // (a list imported longer than this path holds: refused loudly)
  if (lane < nc) { cand_load<D>(B, i, lane, k); sup = supG[lane]; chk = chkG[lane]; }
  const unsigned long long um = B.unusedMask[i];
  for (int zi = 0; zi < nZ; zi++) {  // measurements in index order (:615)
    if (!((um >> zi) & 1ull)) continue;
    const double *z = B.Z + (size_t)D * zi;
    double d2 = 1.0e300;
    if (lane < nc) d2 = cand_support_md2<D>(P, pr, k, z);
    const unsigned long long hit = __ballot(lane < nc && d2 <= F.supportD2);
    if (hit != 0ull) {
      if (lane == __builtin_ctzll(hit)) { cand_correct<D>(P, pr, k, z); sup++; }
    } else {
      Cand<D> kn;
      cand_inverse<D>(P, pr, z, kn);
      if (F.countThr == 1u || nfov <= F.curThr) {
        if (n < cap) { if (lane == 0) { int nn = n; fs_append<D>(B, cur, i, nn, kn, F.newW); } n++; }
        else fail = true;
      } else if (nc < FS_LANE_CANDIDATES) {   // (FastSLAM's landmark candidates still live one per lane)
        if (lane == nc) { k = kn; sup = 1; chk = 0; }
        nc++;
      } else {
        listFull = true;
      }
    }
    // the promotion loop runs once per unassociated measurement (:656-688), with the ++end() wrap of libstdc++'s list
    int kk = 0;
    while (kk < nc) {
      if (lane == kk) chk++;
      bool atEnd = false;
      for (;;) {
        const unsigned supk = (unsigned)__builtin_amdgcn_readlane(sup, kk), chkk = (unsigned)__builtin_amdgcn_readlane(chk, kk);
        if (!(supk >= F.countThr || chkk > F.checkThr || nfov <= F.curThr)) break;
        if (supk >= F.countThr || nfov <= F.curThr) {
          if (n < cap) { if (lane == kk) { int nn = n; fs_append<D>(B, cur, i, nn, k, F.newW * chk); } n++; }
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
