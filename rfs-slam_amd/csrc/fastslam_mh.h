// fastslam_mh.h -- multi-hypothesis FastSLAM (reference include/FastSLAM.hpp:492-556 with config.maxNDataAssocHypotheses_
// > 1): every particle keeps the k best data associations Murty's algorithm finds within maxDataAssocLogLikelihoodDiff_ of
// the best, and is copied once per extra hypothesis (ParticleFilter::copyParticle); resampleWithMapCopy later brings the
// set back to its initial size.
//
// Unlike the single-hypothesis path (fastslam.h), WHICH assignments come 2nd, 3rd, ... depends on the whole reduced table,
// floor-valued cells included (assignments that differ only in which unmatched landmark takes which unmatched measurement
// tie exactly and all count as hypotheses).  So this path follows the reference literally: the dense nMZ x nMZ table,
// CostMatrix::reduce with rows / columns without any possibility kept in the reduced table, and Murty's ranked
// enumeration (murty.h: the same Hungarian method, node pool and heap discipline as the oracle's restatement) on it.
// nMZ = max(landmarks in range, measurements) <= 64 (MURTY_N); beyond that the update refuses loudly.
//
//  fs_mh_associate   one workgroup of two wavefronts per particle: wave 0's lanes build the table rows and reduce it (row /
//                    column counters on lanes); Murty runs in murty.h's block form (children of an expansion shared out
//                    over the waves, the wave-parallel Hungarian solver of hungarian_wave.h inside).
//                    Leaves per particle: the in-range list, the table, nH and the nH assignments (HBM).
//  (host)            slots of the copies, in particle order: pi[h] = nParticles_ - h after each particle's copies (:543-556)
//  fs_mh_copy        one workgroup per new slot: the source particle's map, counters, pose, weight / nH (+ candidates when
//                    the previous update resampled, :551-553)
//  fs_mh_apply       one wavefront per (slot, hypothesis): KF correction, existence log-odds, particle weight (:559-604,696)
//  then gm_prune / fs_new_landmarks of fastslam.h over the grown set.
#pragma once
#include "fastslam.h"

#define FSMH_N MURTY_N          // max table dimension
#define FSMH_MAX_HYP 16         // max config.maxNDataAssocHypotheses_ handled
#define FSMH_NODES (1 + FSMH_MAX_HYP * FSMH_N)
#define FSMH_LDS_N 32           // Murty sub-problems up to this dimension are solved in an LDS tile (8 KB per wave)
#define FSMH_WAVES 2            // wavefronts per particle in fs_mh_associate: the children of a Murty expansion are shared out

// Per-particle HBM block: table T, reduced table Cr, Murty's arena, and the results.
struct FsMhLayout {
  size_t offT, offCr, offCt, offWork, offNodeScore, offNodeParent, offHeap, offNodeId, offNodeA, offRes, offDa, offIdx, offPd, offHdr, total;
};
__host__ __device__ inline FsMhLayout fs_mh_layout() {
  FsMhLayout L;
  size_t o = 0;
  L.offT = o; o += (size_t)FSMH_N * FSMH_N * 8;
  L.offCr = o; o += (size_t)FSMH_N * FSMH_N * 8;
  L.offCt = o; o += (size_t)FSMH_WAVES * FSMH_N * FSMH_N * 8;   // one per wave of the particle's workgroup
  L.offWork = o; o += (size_t)3 * FSMH_N * 8 + 6 * FSMH_N * 4 + 8 * FSMH_N;   // lx ly slack | xy yx p queue | flags
  o = (o + 7) & ~(size_t)7;
  L.offNodeScore = o; o += (size_t)FSMH_NODES * 8;
  L.offNodeParent = o; o += (size_t)FSMH_NODES * 2;
  L.offHeap = o; o += (size_t)FSMH_NODES * 2;
  L.offNodeId = o; o += (size_t)FSMH_NODES;
  L.offNodeA = o; o += (size_t)FSMH_NODES * FSMH_N;
  L.offRes = o; o += (size_t)FSMH_MAX_HYP * FSMH_N;           // Murty's assignments, one row of the reduced table per byte
  o = (o + 7) & ~(size_t)7;
  L.offDa = o; o += (size_t)FSMH_MAX_HYP * FSMH_N * 2;   // short da[h][row]
  L.offIdx = o; o += (size_t)FSMH_N * 2;                  // mixture index of row k
  o = (o + 7) & ~(size_t)7;
  L.offPd = o; o += (size_t)FSMH_N * 8;                   // Pd of row k
  L.offHdr = o; o += 16;                                   // int nIn, nMZ, nH, pad
  L.total = (o + 63) & ~(size_t)63;
  return L;
}
__device__ inline void fs_mh_carve(unsigned char *base, const FsMhLayout &L, MurtyArena &A) {
  A.Ct = (double *)(base + L.offCt);
  unsigned char *p = base + L.offWork;
  A.lx = (double *)p; p += FSMH_N * 8;
  A.ly = (double *)p; p += FSMH_N * 8;
  A.slack = (double *)p; p += FSMH_N * 8;
  A.xy = (int *)p; p += FSMH_N * 4;
  A.yx = (int *)p; p += FSMH_N * 4;
  A.p = (int *)p; p += 2 * FSMH_N * 4;
  A.queue = (int *)p; p += 2 * FSMH_N * 4;
  A.S = p; p += FSMH_N;
  A.T = p; p += FSMH_N;
  A.NS = p; p += FSMH_N;
  A.xq = p; p += FSMH_N;
  A.yq = p; p += FSMH_N;
  A.nodeScore = (double *)(base + L.offNodeScore);
  A.nodeParent = (short *)(base + L.offNodeParent);
  A.heap = (short *)(base + L.offHeap);
  A.nodeId = base + L.offNodeId;
  A.nodeA = base + L.offNodeA;
}

// errBits: ERRBIT_MURTY when the table is larger than FSMH_N or Murty runs out of nodes.
template <int D>
__global__ __launch_bounds__(64 * FSMH_WAVES) __attribute__((amdgpu_waves_per_eu(D == 2 ? 4 : 3)))   // 2-D: <= 128 VGPRs, 8 workgroups per CU, 2048 particles at once
void fs_mh_associate_kernel(Buffers B, Params P, FsParams F, int cur, int nZ, int kmax, double maxDiff,
                                                                        unsigned char *arena) {
  __shared__ double sZ[3 * RFSGPU_MAX_Z];
  __shared__ __align__(16) unsigned char sPdScratch[(D == 3) ? ((VP_PD_SCRATCH_BYTES + 15) & ~15) : 16];
  __shared__ double sTile[FSMH_WAVES][FSMH_LDS_N * FSMH_LDS_N];
  __shared__ double sScore[FSMH_N];
  __shared__ double sBest;
  __shared__ int sCtl[8], sNRed, sNH;
  __shared__ unsigned char sPushed[FSMH_N];
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  const int i = blockIdx.x;
  for (int t = threadIdx.x; t < D * nZ; t += 64 * FSMH_WAVES) sZ[t] = B.Z[t];
  if (threadIdx.x == 0) { sNRed = 0; sNH = 0; }
  __syncthreads();
#ifdef RFS_PROFILE
  long long *fd = B.dbg ? B.dbg + 64 + 4 * (size_t)B.N + 4 * (size_t)blockIdx.x : nullptr;
  if (fd && threadIdx.x == 0) fd[0] = (long long)wall_clock64();
#endif
  const FsMhLayout L = fs_mh_layout();
  unsigned char *base = arena + (size_t)i * L.total;
  double *T = (double *)(base + L.offT);
  unsigned short *gIdx = (unsigned short *)(base + L.offIdx);
  double *gPd = (double *)(base + L.offPd);
  int *hdr = (int *)(base + L.offHdr);
  double *Cr = (double *)(base + L.offCr);
  unsigned char *outR = base + L.offRes;
  short *da = (short *)(base + L.offDa);
  const unsigned long long lt = (1ull << lane) - 1ull;
  // wave 0's state across the Murty stage
  int nIn = 0, nMZ = 0, aFixed = -1, jRed = 0, nRed = 0;
  unsigned long long redI = 0;
  bool tooBig = false;

  if (wave == 0) {   // table, CostMatrix::reduce: one wave
    const int cap = B.cap;
    const int nM = B.count[i];
    const double *slab = B.slab[cur];
    PoseReg pr;
    load_pose(B, P, i, pr);
    const double lim = F.minLog;
    // in-range count first (the table dimension is needed before rows can be written)
    for (int c0 = 0; c0 < nM; c0 += 64) {
      const int m = c0 + lane;
      FsRow<D> row;
      fs_row<D>(B, P, pr, slab, cap, i, m, m < nM, row, sPdScratch);
      nIn += __popcll(__ballot((m < nM) && (row.pd != 0 || row.close)));
    }
    nMZ = nIn > nZ ? nIn : nZ;
    tooBig = nMZ > FSMH_N;
    if (!tooBig) {
      for (int t = lane; t < nMZ * nMZ; t += 64) T[t] = lim;  // :458-465
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      wave_sync();
      int k0 = 0;
      for (int c0 = 0; c0 < nM; c0 += 64) {
        const int m = c0 + lane;
        const bool act = m < nM;
        FsRow<D> row;
        fs_row<D>(B, P, pr, slab, cap, i, m, act, row, sPdScratch);
        const bool inR = act && (row.pd != 0 || row.close);
        const unsigned long long im = __ballot(inR);
        if (inR) {
          const int k = k0 + __popcll(im & lt);
          gIdx[k] = (unsigned short)m;
          gPd[k] = row.pd;
          if (row.valid)
            for (int z = 0; z < nZ; z++) T[k * nMZ + z] = fs_cell_d<D>(row, sZ + D * z, lim);  // :468-481
        }
        k0 += __popcll(im);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      wave_sync();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");

      // ---- CostMatrix::reduce (src/CostMatrix.cpp:263-340) ----
      // Row x's counters on lane x, column y's on lane y.  The serial scan fixes (x, y) exactly when y is row x's only match
      // and x is column y's only match (a pair seen while either count is already 2 is never fixed, and one fixed early is
      // undone by the final count checks), so counts + the single match index decide it.
      const bool inT = lane < nMZ;
      int nMatchI = 0, nMatchJ = 0, yStar = 0, xStar = 0;
      for (int x = 0; x < nMZ; x++) {
        const unsigned long long mk = __ballot(inT && T[x * nMZ + lane] > lim);
        if (lane == x) { nMatchI = __popcll(mk); yStar = mk ? __builtin_ctzll(mk) : 0; }
        if ((mk >> lane) & 1ull) { if (nMatchJ == 0) xStar = x; nMatchJ++; }
      }
      const int nJatStar = __shfl(nMatchJ, yStar, 64), nIatStar = __shfl(nMatchI, xStar, 64);
      aFixed = (inT && nMatchI == 1 && nJatStar == 1) ? yStar : -1;
      const int aRev = (inT && nMatchJ == 1 && nIatStar == 1) ? xStar : -1;
      redI = __ballot(inT && aFixed == -1);
      const unsigned long long redJ = __ballot(inT && aRev == -1);
      nRed = __popcll(redI);
      const int iRed = murty_kth_bit(redI, lane);     // meaningful on lanes < nRed
      jRed = murty_kth_bit(redJ, lane);
      if (nRed == 1) {                                                              // a 1 x 1 remainder is assigned (:325-331)
        if (lane == __builtin_ctzll(redI)) aFixed = __builtin_ctzll(redJ);
        nRed = 0;
      }
      if (nRed > 0) {
        for (int x = 0; x < nRed; x++) {
          const int ix = __builtin_amdgcn_readlane(iRed, x);
          if (lane < nRed) Cr[x * nRed + lane] = T[ix * nMZ + jRed];
        }
        if (lane == 0) sNRed = nRed;
      }
    }
  }
  __threadfence_block();
  __syncthreads();
#ifdef RFS_PROFILE
  if (fd && threadIdx.x == 0) fd[1] = (long long)wall_clock64();
#endif

  // ---- Murty's k best assignments of the reduced table (FastSLAM.hpp:506-541; src/MurtyAlgorithm.cpp:137-320 with
  //      realAssign_n{R,C}_ == n, i.e. no setRealAssignmentBlock): the whole workgroup, murty.h's block form ----
  const int nRedS = __builtin_amdgcn_readfirstlane(sNRed);
  bool okM = true;
  if (nRedS > 0) {
    MurtyArena A;
    fs_mh_carve(base, L, A);
    murty_kbest_block<FSMH_WAVES, FSMH_LDS_N>(
        Cr, nRedS, nRedS - 1, nRedS, FSMH_NODES, kmax, A, okM, sTile[wave], sCtl, sScore, sPushed, wave,
        [&](double s) {                                     // the best assignment (:506-518)
          sBest = s;
          if (0.0 >= maxDiff) return true;                  // (best - best >= maxDiff: only with maxDiff <= 0)
          for (int r = 0; r < nRedS; r++) outR[r] = A.nodeA[r];
          sNH = 1;
          return false;
        },
        [&](double st, int top) {                           // the next one, while it is within maxDiff of the best (:520-540)
          if (sBest - st >= maxDiff) return true;
          const int h = sNH;
          for (int r = 0; r < nRedS; r++) outR[h * FSMH_N + r] = A.nodeA[(size_t)top * FSMH_N + r];
          sNH = h + 1;
          return false;
        });
  }
  __threadfence_block();
  __syncthreads();
#ifdef RFS_PROFILE
  if (fd && threadIdx.x == 0) { fd[2] = (long long)wall_clock64(); fd[3] = sNRed; }
#endif

  if (wave == 0) {
    int nH = 0;
    if (tooBig) {
      if (lane == 0) atomicOr(B.err, ERRBIT_MURTY);
      nIn = 0; nMZ = 0;
    } else if (nRed == 0) {  // :498-505
      if (lane < nIn) da[lane] = (short)aFixed;
      nH = 1;
    } else {
      nH = __builtin_amdgcn_readfirstlane(sNH);
      if (!okM) { if (lane == 0) atomicOr(B.err, ERRBIT_MURTY); nH = 0; }
      const int myRedRow = __popcll(redI & lt);                                   // reduced row of table row `lane`
      for (int h = 0; h < nH; h++) {  // :525-540
        const int res = (lane < nRed) ? outR[h * FSMH_N + lane] : 0;
        const int z_o = __shfl(jRed, res, 64);
        const int val = (z_o < nZ) ? z_o : -2;
        const int mine = __shfl(val, myRedRow, 64);
        if (lane < nIn) da[h * FSMH_N + lane] = (short)((aFixed != -1) ? aFixed : mine);
      }
    }
    if (lane == 0) { hdr[0] = nIn; hdr[1] = nMZ; hdr[2] = nH; }
  }
}

// One workgroup per NEW slot: ParticleFilter::copyParticle (ParticleFilter.hpp:273-294) of slot src -> slot dst.
__global__ __launch_bounds__(256) void fs_mh_copy_kernel(Buffers B, int cur, const int *dstSlot, const int *srcSlot, int copyCand, int poseCovStride) {
  const int d = dstSlot[blockIdx.x], s = srcSlot[blockIdx.x];
  const int n = B.count[s];
  for (int pl = 0; pl < B.npl; pl++) {
    const double *q = B.slab[cur] + ((size_t)s * B.npl + pl) * (size_t)B.cap;
    double *o = B.slab[cur] + ((size_t)d * B.npl + pl) * (size_t)B.cap;
    for (int m = threadIdx.x; m < n; m += blockDim.x) o[m] = q[m];
  }
  if (threadIdx.x == 0) {
    B.count[d] = n;
    B.nInFov[d] = B.nInFov[s];
    B.unusedMask[d] = B.unusedMask[s];
    for (int t = 0; t < 3; t++) B.pose[3 * d + t] = B.pose[3 * s + t];
    if (poseCovStride)
      for (int t = 0; t < 9; t++) B.poseCov[(size_t)poseCovStride * d + t] = B.poseCov[(size_t)poseCovStride * s + t];
  }
  if (copyCand) {  // landmarkCandidates_[pi[h]] = landmarkCandidates_[pi[0]] only when the previous update resampled (:551-553)
    const int nc = B.candCount[s];
    for (int t = threadIdx.x; t < nc * 3; t += blockDim.x) B.candMean[(size_t)d * RFSGPU_MAX_CANDIDATES * 3 + t] = B.candMean[(size_t)s * RFSGPU_MAX_CANDIDATES * 3 + t];
    for (int t = threadIdx.x; t < nc * 6; t += blockDim.x) B.candCov[(size_t)d * RFSGPU_MAX_CANDIDATES * 6 + t] = B.candCov[(size_t)s * RFSGPU_MAX_CANDIDATES * 6 + t];
    for (int t = threadIdx.x; t < nc; t += blockDim.x) {
      B.candSup[(size_t)d * RFSGPU_MAX_CANDIDATES + t] = B.candSup[(size_t)s * RFSGPU_MAX_CANDIDATES + t];
      B.candChk[(size_t)d * RFSGPU_MAX_CANDIDATES + t] = B.candChk[(size_t)s * RFSGPU_MAX_CANDIDATES + t];
    }
    if (threadIdx.x == 0) B.candCount[d] = nc;
  }
}
// weight of every hypothesis slot of a multiplied particle := weight / nH (:545-548)
// phase 0: the copies read the source's undivided weight; phase 1 (a later launch): the sources divide their own
__global__ void fs_mh_split_weights_kernel(double *weight, const int *slotSrc, const int *slotNH, int nSlots, int phase) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= nSlots) return;
  const int nh = slotNH[s], src = slotSrc[s];
  if (nh > 1 && ((phase == 0) == (src != s))) weight[s] = weight[src] / nh;
}

// One wavefront per slot: the update of one particle under one hypothesis (:559-604, :696-697).
template <int D>
__global__ __launch_bounds__(64) void fs_mh_apply_kernel(Buffers B, Params P, FsParams F, int cur, int nZ, const int *slotSrc, const int *slotHyp,
                                                       const unsigned char *arena) {
  __shared__ double sZ[3 * RFSGPU_MAX_Z];
  __shared__ double sC[FSMH_N];
  const int lane = threadIdx.x;
  const int i = blockIdx.x;
  const int src = slotSrc[i], h = slotHyp[i];
  if (h < 0) return;  // no hypothesis: the particle is left untouched
  for (int t = lane; t < D * nZ; t += 64) sZ[t] = B.Z[t];
  wave_sync();
  const FsMhLayout L = fs_mh_layout();
  const unsigned char *base = arena + (size_t)src * L.total;
  const double *T = (const double *)(base + L.offT);
  const unsigned short *gIdx = (const unsigned short *)(base + L.offIdx);
  const double *gPd = (const double *)(base + L.offPd);
  const int *hdr = (const int *)(base + L.offHdr);
  const short *da = (const short *)(base + L.offDa) + (size_t)h * FSMH_N;
  const int nIn = hdr[0], nMZ = hdr[1];
  const int cap = B.cap;
  double *slab = B.slab[cur];
  double *pW = slab + ((size_t)i * B.npl + 0) * cap, *pWP = slab + ((size_t)i * B.npl + 1) * cap;
  PoseReg pr;
  load_pose(B, P, i, pr);
  const double lim = F.minLog;
  const unsigned long long zmask = (nZ >= 64) ? ~0ull : ((1ull << nZ) - 1ull);
  bool upd = false;
  int z = -1;
  const int k = lane;
  if (k < nIn) {
    const int m = gIdx[k];
    z = da[k];
    const double pd = gPd[k];
    const double w = pW[m];
    double val = 0.0;
    if (z < nZ && z >= 0) {
      val = T[k * nMZ + z];
      if (val > lim) upd = fs_kf_correct<D>(P, pr, slab, cap, i, m, sZ, z);  // :584-585
    }
    pWP[m] = w;
    pW[m] = fs_existence_step(F, w, pd, upd);
    sC[k] = upd ? val : 0.0;
  }
  const unsigned long long um = __ballot(upd);
  unsigned long long used = upd ? (1ull << z) : 0ull;
  used = wave_or_u64(used);
  wave_sync();
  if (lane == 0) {
    double logw = 0.0;
    for (int q = 0; q < nIn; q++) logw += sC[q];
    B.weight[i] = B.weight[i] * exp(logw);
    B.unusedMask[i] = (~used) & zmask;
    B.nInFov[i] = __popcll(um);
  }
}
