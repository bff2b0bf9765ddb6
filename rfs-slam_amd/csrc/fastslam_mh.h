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
//  fs_mh_associate   one wavefront per particle: lanes build the table rows, reduce it (row / column counters on lanes) and
//                    run Murty with the wave-parallel Hungarian solver of hungarian_wave.h.
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
#define FSMH_LDS_N 48           // Murty sub-problems up to this dimension are solved in an LDS tile (18 KB: 8 wavefronts per CU)

// Per-particle HBM block: table T, reduced table Cr, Murty's arena, and the results.
struct FsMhLayout {
  size_t offT, offCr, offCt, offWork, offNodeScore, offNodeParent, offHeap, offNodeId, offNodeA, offRes, offDa, offIdx, offPd, offHdr, total;
};
__host__ __device__ inline FsMhLayout fs_mh_layout() {
  FsMhLayout L;
  size_t o = 0;
  L.offT = o; o += (size_t)FSMH_N * FSMH_N * 8;
  L.offCr = o; o += (size_t)FSMH_N * FSMH_N * 8;
  L.offCt = o; o += (size_t)FSMH_N * FSMH_N * 8;
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

// Murty::findNextBest driven like FastSLAM.hpp:506-541: up to kmax assignments of the n x n table C (maximisation), stopping
// at the first whose score is maxDiff or more below the best.  out[h * FSMH_N + row] = column.  Returns nH (uniform).
// (src/MurtyAlgorithm.cpp:137-320 with realAssign_n{R,C}_ == n, i.e. no setRealAssignmentBlock: murty.h's wave-per-problem
// pieces with the dummy-range rule switched off.)
__device__ __forceinline__ int fs_mh_kbest(double *C, int n, int kmax, double maxDiff, MurtyArena &A, unsigned char *out, unsigned char *queue,
                                           double *ldsTile, long long *prof = nullptr) {
  const int lane = threadIdx.x & 63;
  int nNodes = 0, heapLen = 0;
  int a0;
  double best;
  if (!murty_root_wave(C, n, A, a0, best, queue)) return 0;  // rank -1 on the first call: no hypothesis (:511-515)
  if (lane < n) out[lane] = (unsigned char)a0;
  nNodes = 1;
  heapLen = 1;
  if (0.0 >= maxDiff) return 0;  // best - best >= maxDiff (only with maxDiff <= 0)
  int nH = 1;
  while (nH < kmax) {
    if (heapLen == 0) break;  // rank == -1
    if (!murty_expand_wave<FSMH_LDS_N>(C, n, n - 1, n, FSMH_NODES, A, nNodes, heapLen, queue, ldsTile, prof)) return -1;
    if (heapLen == 0) break;
    int top;
    const double s = murty_top_wave(A, top);
    if (best - s >= maxDiff) break;  // :520-523
    if (lane < n) out[nH * FSMH_N + lane] = A.nodeA[(size_t)top * FSMH_N + lane];
    nH++;
  }
  return nH;
}

// errBits: ERRBIT_MURTY when the table is larger than FSMH_N or Murty runs out of nodes.
template <int D>
__global__ __launch_bounds__(64) void fs_mh_associate_kernel(Buffers B, Params P, FsParams F, int cur, int nZ, int kmax, double maxDiff,
                                                           unsigned char *arena) {
  __shared__ double sZ[3 * RFSGPU_MAX_Z];
  __shared__ unsigned char sQueue[2 * FSMH_N];
  __shared__ __align__(16) unsigned char sPdScratch[(D == 3) ? ((VP_PD_SCRATCH_BYTES + 15) & ~15) : 16];
  __shared__ double sTile[FSMH_LDS_N * FSMH_LDS_N];
  const int lane = threadIdx.x;
  const int i = blockIdx.x;
  for (int t = lane; t < D * nZ; t += 64) sZ[t] = B.Z[t];
  wave_sync();
  const FsMhLayout L = fs_mh_layout();
  unsigned char *base = arena + (size_t)i * L.total;
  double *T = (double *)(base + L.offT);
  unsigned short *gIdx = (unsigned short *)(base + L.offIdx);
  double *gPd = (double *)(base + L.offPd);
  int *hdr = (int *)(base + L.offHdr);
  const int cap = B.cap;
  const int nM = B.count[i];
  const double *slab = B.slab[cur];
  PoseReg pr;
  load_pose(B, P, i, pr);
  const double lim = F.minLog;
  const unsigned long long lt = (1ull << lane) - 1ull;

  // in-range count first (the table dimension is needed before rows can be written)
  int nIn = 0;
  for (int c0 = 0; c0 < nM; c0 += 64) {
    const int m = c0 + lane;
    FsRow<D> row;
    fs_row<D>(B, P, pr, slab, cap, i, m, m < nM, row, sPdScratch);
    nIn += __popcll(__ballot((m < nM) && (row.pd != 0 || row.close)));
  }
  const int nMZ = nIn > nZ ? nIn : nZ;
  if (nMZ > FSMH_N) {
    if (lane == 0) { atomicOr(B.err, ERRBIT_MURTY); hdr[0] = 0; hdr[1] = 0; hdr[2] = 0; }
    return;
  }
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
  // Row x's counters on lane x, column y's on lane y.  The serial scan fixes (x, y) exactly when y is row x's only match and
  // x is column y's only match (a pair seen while either count is already 2 is never fixed, and one fixed early is undone
  // by the final count checks), so counts + the single match index decide it.
  short *da = (short *)(base + L.offDa);
  const bool inT = lane < nMZ;
  int nMatchI = 0, nMatchJ = 0, yStar = 0, xStar = 0;
  for (int x = 0; x < nMZ; x++) {
    const unsigned long long mk = __ballot(inT && T[x * nMZ + lane] > lim);
    if (lane == x) { nMatchI = __popcll(mk); yStar = mk ? __builtin_ctzll(mk) : 0; }
    if ((mk >> lane) & 1ull) { if (nMatchJ == 0) xStar = x; nMatchJ++; }
  }
  const int nJatStar = __shfl(nMatchJ, yStar, 64), nIatStar = __shfl(nMatchI, xStar, 64);
  int aFixed = (inT && nMatchI == 1 && nJatStar == 1) ? yStar : -1;
  const int aRev = (inT && nMatchJ == 1 && nIatStar == 1) ? xStar : -1;
  const unsigned long long redI = __ballot(inT && aFixed == -1), redJ = __ballot(inT && aRev == -1);
  int nRed = __popcll(redI);
  const int iRed = murty_kth_bit(redI, lane), jRed = murty_kth_bit(redJ, lane);     // meaningful on lanes < nRed
  if (nRed == 1) {                                                              // a 1 x 1 remainder is assigned (:325-331)
    if (lane == __builtin_ctzll(redI)) aFixed = __builtin_ctzll(redJ);
    nRed = 0;
  }
  int nH = 0;
  if (nRed == 0) {  // :498-505
    if (lane < nIn) da[lane] = (short)aFixed;
    nH = 1;
  } else {
    double *Cr = (double *)(base + L.offCr);
    for (int x = 0; x < nRed; x++) {
      const int ix = __builtin_amdgcn_readlane(iRed, x);
      if (lane < nRed) Cr[x * nRed + lane] = T[ix * nMZ + jRed];
    }
    MurtyArena A;
    fs_mh_carve(base, L, A);
    unsigned char *outR = base + L.offRes;
#ifdef RFS_PROFILE
    long long prof[9] = {(long long)__builtin_readcyclecounter(), 0, 0, nRed, 0, 0, 0, 0, 0};
    nH = fs_mh_kbest(Cr, nRed, kmax, maxDiff, A, outR, sQueue, sTile, prof);
    if (B.dbg && lane == 0) {
      long long *o = B.dbg + 64 + 4 * (size_t)i;
      o[0] = (long long)__builtin_readcyclecounter() - prof[0]; o[1] = prof[1]; o[2] = prof[2]; o[3] = prof[3];
      if (i == 7) printf("particle 7: children %lld dims %lld main-loop steps %lld bfs iterations %lld label updates %lld; solver cycles %lld of which main loop %lld\n",
                         prof[2], prof[7], prof[4], prof[5], prof[6], prof[1], prof[8]);
    }
#else
    nH = fs_mh_kbest(Cr, nRed, kmax, maxDiff, A, outR, sQueue, sTile);
#endif
    if (nH < 0) { if (lane == 0) atomicOr(B.err, ERRBIT_MURTY); nH = 0; }
    const int myRedRow = __popcll(redI & lt);                                   // reduced row of table row `lane`
    for (int h = 0; h < nH; h++) {  // :525-540
      const int res = (lane < nRed) ? outR[h * FSMH_N + lane] : 0;              // own store in fs_mh_kbest
      const int z_o = __shfl(jRed, res, 64);
      const int val = (z_o < nZ) ? z_o : -2;
      const int mine = __shfl(val, myRedRow, 64);
      if (lane < nIn) da[h * FSMH_N + lane] = (short)((aFixed != -1) ? aFixed : mine);
    }
  }
  if (lane == 0) { hdr[0] = nIn; hdr[1] = nMZ; hdr[2] = nH; }
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
