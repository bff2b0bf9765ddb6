// weighting.h -- phd_weight_multifeature: RBPHDFilter::importanceWeighting (reference
// include/RBPHDFilter.hpp:728-819) + rfsMeasurementLikelihood (:821-997) + CostMatrixGeneral::partition
// (src/CostMatrix.cpp:92-157) + the <=8 partial-assignment enumeration that the reference drives through
// PermutationLexicographic (src/PermutationLexicographic.cpp:38-96).
//
// One workgroup of WPP waves per particle (phd_weight_particle).  Steps: (1) rank-sort the mixture by weight (weight
// desc, index asc; fp32 first pass, exact fp64 completion) on all threads and write the sorted mixture to the other slab
// (the order merge needs); (2) pick the evaluation points (wave 0) while the last wave sums the weights; then two
// concurrent strands -- (3) intensity products before/after the update on waves 1.., lanes over Gaussians, 8 evaluation
// points at a time in registers; (4) likelihood table L (nE x nZ) in LDS, (5) bipartite connected components by
// min-label propagation over 64-bit adjacency masks, numbered like BGL's DFS discovery order (by smallest vertex), incl.
// the reference's zero-partition merge and its partition-indexing quirk, (6) one lane per partition sums the partial
// assignments, on wave 0; partitions with nR+nC > 8 go to the Murty work queue (murty.h).  (7) the particle weight.
#pragma once
#include "common.h"
#include "stdsort_replay.h"

// Partitions too large for the in-kernel enumeration (nR + nC > 8) are handed to murty.h through this queue.
struct MurtyJob {
  int particle;
  int nR, nC;
  int slot;         // index into partLik of that particle
  // extended (nR+nC)^2 log-likelihood matrix is stored in the job's scratch area
};
struct MurtyQueue {
  int *count;          // [1] number of jobs
  MurtyJob *jobs;      // [maxJobs]
  double *mats;        // [maxJobs][MURTY_MAXN*MURTY_MAXN]
  double *results;     // [maxJobs]
  int maxJobs;
  int *order;          // [maxJobs] job indices, largest extended dimension first (murty_order_kernel); may be null
};
#define MURTY_MAXN 64

// Sparse intensity sums (phd_weight_particle, step 3b): per evaluation point a list of <= WEIGHT_SPARSE_SEG Gaussians (u16) + its own
// term (f64) + four 32-bit words (x, y, threshold, count); 16 evaluation points in flight per workgroup (one wave x 16 or two waves x 8).
#define WEIGHT_SPARSE_SEG 64
#define WEIGHT_SPARSE_LDS_BYTES (16 * (WEIGHT_SPARSE_SEG * 2 + 8 + 4 * 4))
struct WeightLDS {
  double *keys;                // [cap]
  int *perm;                   // [cap rounded up to 64] (doubles as the sorted-chunk buffer of the rank sort)
  float *fkeys;                // [cap rounded up to 64]
  double *evX, *evY, *evPd;    // [evalCap]
  double *evLog1mPd;           // [evalCap]
  int *evIdx;                  // [evalCap] sorted position of the evaluation point
  double *evZ;                 // [evalCap][7] z_exp0, z_exp1, i00, i01, i10, i11, factor
  double *L;                   // [evalCap][nZ]
  int *labR, *labC;            // [64]
  unsigned long long *compRows, *compCols;  // [128]
  double *partLik;             // [128]
  double *isum;                // [2 evalCap] intensity sums of the evaluation points (before / after), split mode
  unsigned char *sparse;       // [WEIGHT_SPARSE_LDS_BYTES] pair lists + fp32 evaluation points / thresholds of the sparse intensity sums
};

__host__ __device__ inline size_t weight_lds_bytes_per_wave(int cap, int evalCap, int nZ) {
  size_t b = 0;
  b += (size_t)cap * 8;            // keys
  b += (size_t)((cap + 63) & ~63) * 4;   // perm
  b += (size_t)((cap + 63) & ~63) * 4;   // fkeys (float keys for the rank sort, whole 64-entry chunks)
  b += (size_t)evalCap * 8 * 4;    // evX evY evPd evLog1mPd
  b += (size_t)evalCap * 4;        // evIdx
  b += (size_t)evalCap * 7 * 8;    // evZ
  b += (size_t)evalCap * nZ * 8;   // L
  b += 64 * 4 * 2;                 // labR labC
  b += 128 * 8 * 2;                // compRows compCols
  b += 128 * 8;                    // partLik
  b += (size_t)evalCap * 8 * 2;    // isum
  b += WEIGHT_SPARSE_LDS_BYTES;    // sparse
  return (b + 15) & ~(size_t)15;
}

__device__ __forceinline__ void carve_weight_lds(unsigned char *base, int cap, int evalCap, int nZ, WeightLDS &s) {
  unsigned char *p = base;
  s.keys = (double *)p; p += (size_t)cap * 8;
  s.evX = (double *)p; p += (size_t)evalCap * 8;
  s.evY = (double *)p; p += (size_t)evalCap * 8;
  s.evPd = (double *)p; p += (size_t)evalCap * 8;
  s.evLog1mPd = (double *)p; p += (size_t)evalCap * 8;
  s.evZ = (double *)p; p += (size_t)evalCap * 7 * 8;
  s.L = (double *)p; p += (size_t)evalCap * nZ * 8;
  s.compRows = (unsigned long long *)p; p += 128 * 8;
  s.compCols = (unsigned long long *)p; p += 128 * 8;
  s.partLik = (double *)p; p += 128 * 8;
  s.isum = (double *)p; p += (size_t)evalCap * 8 * 2;
  s.sparse = p; p += WEIGHT_SPARSE_LDS_BYTES;
  s.perm = (int *)p; p += (size_t)((cap + 63) & ~63) * 4;
  s.fkeys = (float *)p; p += (size_t)((cap + 63) & ~63) * 4;
  s.evIdx = (int *)p; p += (size_t)evalCap * 4;
  s.labR = (int *)p; p += 64 * 4;
  s.labC = (int *)p; p += 64 * 4;
}

__device__ __forceinline__ int nth_bit(unsigned long long m, int k) {  // index of the k-th (0-based) set bit
  for (int t = 0; t < k; t++) m &= m - 1;
  return __builtin_ctzll(m);
}

// Sum over all partial assignments of one partition (rows = eval points, cols = measurements), r + c <= 8:
//   sum over matchings M of  prod_{(i,j) in M} L[i][j] * prod_{rows i unmatched} (1 - Pd_i) * clutter^(#unmatched cols)
// The reference walks the assignments in lexicographic order and forms each term as exp(sum of logs) with the log table
// floored at -1000 (include/RBPHDFilter.hpp:907-917, 961-988); a floored cell makes its term exp(<= -1000 + ...) == 0 exactly,
// and a cell is floored exactly when L == 0 (a positive double has log >= -745), i.e. when the product form gives 0 as well.
// Here the same sum is built by a subset recurrence over the SMALLER side of the partition (min(r, c) <= 4, so 16 states,
// all register-resident with static indexing): the items of the larger side are taken one at a time,
//   f'[S] = f[S] * u + sum_{b in S} f[S - b] * a[b],   u = the item's "left unmatched" factor, a[b] = its L against small item b,
// and at the end every state is closed with the unmatched factors of the small side.  No exp, no log, no data-dependent
// control flow: one lane per partition costs <= 7 items x ~50 fused multiply-adds, against up to 209 terms x (exp + walk) of
// the enumeration.  The terms are the reference's up to the rounding of exp(log a + log b) vs a * b and the order of the
// additions (a few ulp, against the 1e-9 tolerance on particle weights).
__device__ double enumerate_partition(const WeightLDS &s, int nZ, unsigned long long rmask, unsigned long long cmask, double clutter) {
  const int r = __popcll(rmask), c = __popcll(cmask);
  const bool colsSmall = c <= r;
  const unsigned long long smallMask = colsSmall ? cmask : rmask;
  unsigned long long largeMask = colsSmall ? rmask : cmask;
  const int k = colsSmall ? c : r;
  int sm[4];
  double h[4];  // the small item's factor when it stays unmatched
  {
    unsigned long long mm = smallMask;
#pragma unroll
    for (int b = 0; b < 4; b++) {
      sm[b] = (b < k) ? __builtin_ctzll(mm) : 0;
      h[b] = (b < k) ? (colsSmall ? clutter : 1.0 - s.evPd[sm[b]]) : 1.0;
      if (b < k) mm &= mm - 1;
    }
  }
  double f[16];
  f[0] = 1.0;
#pragma unroll
  for (int S = 1; S < 16; S++) f[S] = 0.0;
  for (; largeMask; largeMask &= largeMask - 1) {
    const int idx = __builtin_ctzll(largeMask);
    const double u = colsSmall ? 1.0 - s.evPd[idx] : clutter;
    double a[4];
#pragma unroll
    for (int b = 0; b < 4; b++) a[b] = (b < k) ? (colsSmall ? s.L[idx * nZ + sm[b]] : s.L[sm[b] * nZ + idx]) : 0.0;
#pragma unroll
    for (int S = 15; S >= 1; S--) {  // descending: f[S - b] is still the previous item's value
      double acc = f[S] * u;
#pragma unroll
      for (int b = 0; b < 4; b++)
        if ((S >> b) & 1) acc += f[S ^ (1 << b)] * a[b];
      f[S] = acc;
    }
    f[0] *= u;
  }
  double lik = 0.0;
#pragma unroll
  for (int S = 0; S < 16; S++) {
    double g = f[S];
#pragma unroll
    for (int b = 0; b < 4; b++)
      if (!((S >> b) & 1)) g *= h[b];
    lik += g;
  }
  return lik;
}

// The same sum for a partition of ANY size whose smaller side has k <= RFS_EXACT_MAX_SMALL items, by the whole wave: the 2^k
// subset states are spread over the wave -- the low 3 bits of a state index pick one of 8 registers, the higher bits the lane --
// so an item of the larger side costs 8 k fused multiply-adds per lane plus (k - 3) lane exchanges per register.  This is the
// EXACT (untruncated) multi-feature likelihood of the partition; the reference hands partitions with r + c > 8 to Murty's
// ranked enumeration and stops after the 200 best assignments (include/RBPHDFilter.hpp:920-959), which is what the default
// (bug-compatible) mode reproduces through murty.h.  Selected by rfsgpu_set_partition_mode(RFSGPU_PARTITION_EXACT).
#define RFS_EXACT_MAX_SMALL 9
__device__ double partition_exact_wave(const WeightLDS &s, int nZ, unsigned long long rmask, unsigned long long cmask, double clutter, int lane) {
  const int r = __popcll(rmask), c = __popcll(cmask);
  const bool colsSmall = c <= r;
  const unsigned long long smallMask = colsSmall ? cmask : rmask;
  unsigned long long largeMask = colsSmall ? rmask : cmask;
  const int k = colsSmall ? c : r;                 // <= RFS_EXACT_MAX_SMALL (caller)
  int sm[RFS_EXACT_MAX_SMALL];
  double h[RFS_EXACT_MAX_SMALL];
  {
    unsigned long long mm = smallMask;
#pragma unroll
    for (int b = 0; b < RFS_EXACT_MAX_SMALL; b++) {
      sm[b] = (b < k) ? __builtin_ctzll(mm) : 0;
      h[b] = (b < k) ? (colsSmall ? clutter : 1.0 - s.evPd[sm[b]]) : 1.0;
      if (b < k) mm &= mm - 1;
    }
  }
  double f[8];
#pragma unroll
  for (int q = 0; q < 8; q++) f[q] = (lane == 0 && q == 0) ? 1.0 : 0.0;
  for (; largeMask; largeMask &= largeMask - 1) {
    const int idx = __builtin_ctzll(largeMask);
    const double u = colsSmall ? 1.0 - s.evPd[idx] : clutter;
    double a[RFS_EXACT_MAX_SMALL];
#pragma unroll
    for (int b = 0; b < RFS_EXACT_MAX_SMALL; b++) a[b] = (b < k) ? (colsSmall ? s.L[idx * nZ + sm[b]] : s.L[sm[b] * nZ + idx]) : 0.0;
    double g[8];
#pragma unroll
    for (int q = 0; q < 8; q++) {
      double acc = f[q] * u;
#pragma unroll
      for (int b = 0; b < 3; b++)
        if ((q >> b) & 1) acc += f[q ^ (1 << b)] * a[b];
      g[q] = acc;
    }
#pragma unroll
    for (int b = 3; b < RFS_EXACT_MAX_SMALL; b++) {
      if (b >= k) break;                            // (uniform)
      const bool has = (lane >> (b - 3)) & 1;
#pragma unroll
      for (int q = 0; q < 8; q++) {
        const double partner = __shfl_xor(f[q], 1 << (b - 3), 64);   // state with bit b cleared lives in lane ^ (1 << (b - 3))
        g[q] += has ? partner * a[b] : 0.0;
      }
    }
#pragma unroll
    for (int q = 0; q < 8; q++) f[q] = g[q];
  }
  double gl = 1.0;                                  // unmatched small items of the lane bits
#pragma unroll
  for (int b = 3; b < RFS_EXACT_MAX_SMALL; b++)
    if (!((lane >> (b - 3)) & 1)) gl *= h[b];
  double part = 0.0;
#pragma unroll
  for (int q = 0; q < 8; q++) {
    double gq = f[q];
#pragma unroll
    for (int b = 0; b < 3; b++)
      if (!((q >> b) & 1)) gq *= h[b];
    part += gq;
  }
  return wave_sum_dpp(part * gl);
}

// Steps 5-6 of the particle weight, model-independent: connected components of the bipartite graph (rows = evaluation
// points, columns = measurements) of the likelihood table s.L (nE x nZ, already gated, incl. Pd), the reference's
// zero-partition merge + partition-indexing quirk, one lane per partition for the <= 8 enumeration, Murty jobs for the
// rest.  Returns the product over the visited partitions (RBPHDFilter.hpp:865-990), before the clutter-integral division.
__device__ double rfs_partitions_wave(const WeightLDS &s, int nE, int nZ, double clutter, int lane, int particle, MurtyQueue Q, int *err,
                                      int exactMode = 0, long long *dbgp = nullptr) {
  // ---- 5. connected components of the bipartite graph (rows = eval points, cols = measurements) ----
#ifdef RFS_PROFILE
#define PART_T(k) do { if (dbgp && particle == 7 && lane == 0) dbgp[k] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define PART_T(k) do { } while (0)
#endif
  PART_T(27);
  unsigned long long myRow = 0, myCol = 0;
  if (lane < nE) for (int n = 0; n < nZ; n++) if (s.L[lane * nZ + n] != 0.0) myRow |= 1ull << n;
  if (lane < nZ) for (int e = 0; e < nE; e++) if (s.L[e * nZ + lane] != 0.0) myCol |= 1ull << e;
  int labR = lane, labC = nE + lane;  // label = smallest vertex index reachable (rows first, then columns)
  s.labR[lane] = labR;
  s.labC[lane] = labC;
  wave_sync();
  for (int it = 0; it < 130; it++) {
    int nr = labR;
    for (unsigned long long mm = myRow; mm; mm &= mm - 1) { int v = s.labC[__builtin_ctzll(mm)]; nr = v < nr ? v : nr; }
    bool ch = nr != labR;
    labR = nr;
    s.labR[lane] = labR;
    wave_sync();
    int nc = labC;
    for (unsigned long long mm = myCol; mm; mm &= mm - 1) { int v = s.labR[__builtin_ctzll(mm)]; nc = v < nc ? v : nc; }
    ch = ch || (nc != labC);
    labC = nc;
    s.labC[lane] = labC;
    wave_sync();
    if (__ballot(ch) == 0ull) break;
  }
  PART_T(28);
  // (the reference's log table, :907-917, is formed only where Murty needs it: see enumerate_partition)
  const unsigned long long rootR = __ballot(lane < nE && labR == lane);
  const unsigned long long rootC = __ballot(lane < nZ && labC == nE + lane);
  const int nRootR = __popcll(rootR);
  const int ncc = nRootR + __popcll(rootC);
  // component id = rank of its smallest vertex (== BGL DFS discovery order)
  auto comp_of = [&](int label) -> int {
    return label < nE ? __popcll(rootR & ((1ull << label) - 1ull)) : nRootR + __popcll(rootC & ((1ull << (label - nE)) - 1ull));
  };
  s.compRows[lane] = 0; s.compRows[lane + 64] = 0;
  s.compCols[lane] = 0; s.compCols[lane + 64] = 0;
  wave_sync();
  if (lane < nE) atomicOr(&s.compRows[comp_of(labR)], 1ull << lane);
  if (lane < nZ) atomicOr(&s.compCols[comp_of(labC)], 1ull << lane);
  wave_sync();
  // zero partitions (no rows or no cols) are merged into the first one (src/CostMatrix.cpp:126-144)
  unsigned long long zr = 0, zc = 0;
  bool z0 = false, z1 = false;
  if (lane < ncc) { z0 = (s.compRows[lane] == 0 || s.compCols[lane] == 0); if (z0) { zr |= s.compRows[lane]; zc |= s.compCols[lane]; } }
  if (lane + 64 < ncc) { z1 = (s.compRows[lane + 64] == 0 || s.compCols[lane + 64] == 0); if (z1) { zr |= s.compRows[lane + 64]; zc |= s.compCols[lane + 64]; } }
  const unsigned long long zeroLo = __ballot(z0), zeroHi = __ballot(z1);
  const int nZero = __popcll(zeroLo) + __popcll(zeroHi);
  const int combined = zeroLo ? __builtin_ctzll(zeroLo) : (zeroHi ? 64 + __builtin_ctzll(zeroHi) : -1);
  const unsigned long long mergedRows = wave_or_u64(zr), mergedCols = wave_or_u64(zc);
  const int nPartitions = ncc - (nZero > 0 ? nZero - 1 : 0);  // caller still indexes components [0, nPartitions) -- quirk kept

  PART_T(29);
  // ---- 6. one lane per partition ----
  for (int p0 = 0; p0 < nPartitions; p0 += 64) {
    const int p = p0 + lane;
    bool wantExact = false;
    unsigned long long rmask = 0, cmask = 0;
    if (p < nPartitions) {
    rmask = s.compRows[p]; cmask = s.compCols[p];
    double pl;
    if (p == combined) {  // all landmarks mis-detected, all measurements outliers (:891-900; Pd, not 1-Pd)
      rmask = mergedRows;
      cmask = mergedCols;
      pl = 1.0;
      for (unsigned long long mm = rmask; mm; mm &= mm - 1) pl *= s.evPd[__builtin_ctzll(mm)];
      for (unsigned long long mm = cmask; mm; mm &= mm - 1) pl *= clutter;
    } else if (__popcll(rmask) + __popcll(cmask) <= 8) {
      pl = enumerate_partition(s, nZ, rmask, cmask, clutter);
    } else if (exactMode && min(__popcll(rmask), __popcll(cmask)) <= RFS_EXACT_MAX_SMALL) {
      wantExact = true;   // the whole wave takes it below
      pl = 1.0;
    } else {
      // Murty-200 (:920-959): queue the extended matrix; the factor is multiplied in by murty_kernel
      pl = 1.0;
      int job = Q.count ? atomicAdd(Q.count, 1) : Q.maxJobs;
      const int nR = __popcll(rmask), nC = __popcll(cmask), n = nR + nC;
      const double logc = log(clutter);
      if (job < Q.maxJobs && n <= MURTY_MAXN) {
        MurtyJob J;
        J.particle = particle; J.nR = nR; J.nC = nC; J.slot = p;
        Q.jobs[job] = J;
        double *M = Q.mats + (size_t)job * MURTY_MAXN * MURTY_MAXN;
        for (int a = 0; a < n; a++)
          for (int b = 0; b < n; b++) {
            double v;
            if (a < nR && b < nC) {  // log table with the -1000 floor (:907-917)
              v = s.L[nth_bit(rmask, a) * nZ + nth_bit(cmask, b)];
              if (v == 0.0) v = -1000.0;
              else { v = log(v); if (v < -1000.0) v = -1000.0; }
            }
            else if (a < nR) v = (a == b - nC) ? s.evLog1mPd[nth_bit(rmask, a)] : -1000.0;
            else if (b < nC) v = (a - nR == b) ? logc : -1000.0;
            else v = 0.0;
            M[a * n + b] = v;
          }
      } else {
        atomicOr(err, ERRBIT_MURTY);
        if (job < Q.maxJobs) {   // the slot is reserved: leave a skip-job (dimension 0) in it, never stale or uninitialised contents
          MurtyJob J;
          J.particle = particle; J.nR = 0; J.nC = 0; J.slot = p;
          Q.jobs[job] = J;
        }
      }
    }
    s.partLik[p] = pl;
    }
    // exact mode: the large partitions of this batch, one after the other, by the whole wave
    for (unsigned long long em = __ballot(wantExact); em; em &= em - 1) {
      const int src = __builtin_ctzll(em);
      const double v = partition_exact_wave(s, nZ, readlane_u64(rmask, src), readlane_u64(cmask, src), clutter, lane);
      if (lane == src) s.partLik[p] = v;
    }
  }
  wave_sync();
  PART_T(30);
#ifdef RFS_PROFILE
  if (dbgp && particle == 7 && lane == 0) dbgp[31] = nPartitions;
#endif
  double l = 1.0;
  for (int p = 0; p < nPartitions; p++) l *= s.partLik[p];
  return l;
}

// Cross-wave scratch of the multi-wave kernel: a few doubles / ints after the per-particle LDS block.
#define WEIGHT_SCRATCH_BYTES 64

// ---- rank sort by buckets ---------------------------------------------------------------------------------------------------
// rank(m) = #{ j : key_j > key_m  or (key_j == key_m and j < m) }  -- the order of sortByWeight with ties by index.
// Counting that against all entries (or against sorted 64-entry chunks, the form below this one) was 15 % of the fused
// step's vector instructions at 2000 x 200.  Here the keys' order-preserving u64 images are cut into NB equal buckets over the
// range of their upper words (an LDS histogram, one wave-wide scan, a scatter -- the merge phase's grid build in one
// dimension), and an entry is compared only with the members of its own bucket: rank = entries in the buckets ahead + the
// members of its bucket that are ahead of it.  Exact for every input (the u64 image is a total order consistent with the
// fp64 one; -0 is folded into +0); a mixture whose keys crowd into one bucket (many equal weights) makes the walks long, and
// the caller then falls back to the chunk form (returns false).
__device__ __forceinline__ unsigned long long sort_key_u64(double w) {
  const long long b = __double_as_longlong(w + 0.0);
  return (unsigned long long)b ^ ((unsigned long long)(b >> 63) | 0x8000000000000000ull);
}
__host__ __device__ inline int rank_sort_log_buckets(int cap) { return cap >= 344 ? 10 : (cap >= 176 ? 9 : (cap >= 90 ? 8 : 7)); }  // NB*2 + 16 + cap*2 <= cap*8
#ifndef RANK_SORT_MAX_BUCKET
#define RANK_SORT_MAX_BUCKET 32
#endif
// (Measured and dropped, r04 -- profiles/r04h_ab_rank_sort_bucket_order.txt: the last loop taking the entries in BUCKET order, as the
//  merge's candidate scan takes them in grid order: fused step 121.7 -> 123.2 us at configs[1] -- the walks are short here, and the
//  extra list read and the scattered write of the permutation cost more than the lanes' better agreement saves.)
// keyAt(m): key of entry m < N (LDS reads; ties rank by m); hist: (NB / 2 + 4) words of LDS scratch for NB = 1 << logNB buckets; order: [N]
// u16 of LDS scratch; xscr: 2 WPP + 1 ints of LDS scratch; the ranks of entries tid + NT * k go to rl[k] (N <= NS * NT).
// All threads of the block call it; the caller synchronises before it reuses the scratch.
// tied (out, per thread): one of this thread's entries shares its key with another entry -- what decides whether the order of equal
// keys has to be corrected to std::sort's (stdsort_replay.h); undefined when the function returns false.
template <int WPP, int NS, class KeyAt, class Sync>
__device__ __forceinline__ bool bucket_rank_sort(KeyAt keyAt, const int N, const int tid, unsigned *hist, unsigned short *order, const int logNB, int *xscr,
                                                 int (&rl)[NS], Sync block_sync, bool &tied) {
  constexpr int NT = WPP * 64;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int NB = 1 << logNB;
  auto cell_at = [&](int e) -> unsigned { return (hist[e >> 1] >> (16 * (e & 1))) & 0xffffu; };
  unsigned hmin = 0xffffffffu, hmax = 0u;
#pragma unroll
  for (int k = 0; k < NS; k++) {
    const int m = tid + NT * k;
    if (m < N) {
      const unsigned h = (unsigned)(sort_key_u64(keyAt(m)) >> 32);
      hmin = min(hmin, h);
      hmax = max(hmax, h);
    }
  }
  hmin = wave_min_u32(hmin);
  hmax = wave_max_u32(hmax);
  if (lane == 0) { xscr[wave] = (int)hmin; xscr[WPP + wave] = (int)hmax; }
  for (int c = tid; c <= NB / 2; c += NT) hist[c] = 0u;
  block_sync();
#pragma unroll
  for (int w2 = 0; w2 < WPP; w2++) { hmin = min(hmin, (unsigned)xscr[w2]); hmax = max(hmax, (unsigned)xscr[WPP + w2]); }
  // bucket of an upper word h: (hmax - h) scaled onto [0, NB) in fp32 -- conversion, product with a positive constant and
  // truncation are all monotone, which is all the ranks need; the clamp covers the rounding at the far end
  const float bscale = (float)NB * (1.f - 1e-6f) * __builtin_amdgcn_rcpf((float)(hmax - hmin) + 1.f);
  auto bucket_of = [&](unsigned h) -> int { return min((int)((float)(hmax - h) * bscale), NB - 1); };
#pragma unroll
  for (int k = 0; k < NS; k++) {
    const int m = tid + NT * k;
    if (m < N) {
      const int e = bucket_of((unsigned)(sort_key_u64(keyAt(m)) >> 32)) + 1;  // largest keys first; counts shifted by one entry
      atomicAdd(&hist[e >> 1], 1u << (16 * (e & 1)));
    }
  }
  block_sync();
  if (wave == 0) {  // exclusive scan over the NB + 1 half-word entries; lane l owns words [WPL l, WPL (l + 1)), the last entry gets the total
    const int WPL = NB >> 7;
    int tot = 0;
    unsigned mx = (lane == 63) ? (hist[NB / 2] & 0xffffu) : 0u;                     // (the last bucket's count sits in the extra entry)
    for (int k = 0; k < WPL; k++) {
      const unsigned v = hist[WPL * lane + k];
      tot += (int)(v & 0xffffu) + (int)(v >> 16);
      mx = max(mx, max(v & 0xffffu, v >> 16));
    }
    int off = wave_excl_scan(tot, lane);
    mx = wave_max_u32(mx);
    for (int k = 0; k < WPL; k++) {
      const unsigned v = hist[WPL * lane + k];
      const unsigned lo = (unsigned)off;
      off += (int)(v & 0xffffu);
      const unsigned hi = (unsigned)off;
      off += (int)(v >> 16);
      hist[WPL * lane + k] = lo | (hi << 16);
    }
    if (lane == 63) hist[NB / 2] = (unsigned)off;
    if (lane == 0) xscr[2 * WPP] = (int)mx;
  }
  block_sync();
  if (xscr[2 * WPP] > RANK_SORT_MAX_BUCKET) return false;
#pragma unroll
  for (int k = 0; k < NS; k++) {
    const int m = tid + NT * k;
    if (m < N) {
      const int e = bucket_of((unsigned)(sort_key_u64(keyAt(m)) >> 32)) + 1;
      const unsigned pos = (atomicAdd(&hist[e >> 1], 1u << (16 * (e & 1))) >> (16 * (e & 1))) & 0xffffu;
      order[pos] = (unsigned short)m;
    }
  }
  block_sync();   // entry c of the histogram is now the start of bucket c, entry c + 1 its end
#pragma unroll
  for (int k = 0; k < NS; k++) {
    const int m = tid + NT * k;
    rl[k] = 0;
    if (m < N) {
      const unsigned long long u = sort_key_u64(keyAt(m));
      const int b = bucket_of((unsigned)(u >> 32));
      const int st = (int)cell_at(b), en = (int)cell_at(b + 1);
      int r = st, same = 0;
      for (int q = st; q < en; q++) {
        const int j = order[q];
        const unsigned long long uj = sort_key_u64(keyAt(j));
        r += ((uj > u) | ((uj == u) & (j < m))) ? 1 : 0;
        same += (uj == u) ? 1 : 0;
      }
      rl[k] = r;
      tied |= same > 1;       // (the entry itself is in its bucket)
    }
  }
  return true;
}

#ifndef WEIGHT_W0_SHARE_NUM
#define WEIGHT_W0_SHARE_NUM 0   // split mode: wave 0 takes the intensity sums over the last NUM / 5 of the mixture's chunks after its own strand.
#endif                          // Measured twice (r02k at C2a: weighting phase 36 -> 40 us; r04: fused step 121.4 -> 124.4 / 126.5 us with 1 / 2): off
#ifndef WEIGHT_SPARSE_INTENSITY
#define WEIGHT_SPARSE_INTENSITY 1   // intensity sums over the pairs within reach only (phd_weight_particle, step 3b); 0: the dense loop
#endif
#ifndef WEIGHT_SPARSE_MIN_N
#define WEIGHT_SPARSE_MIN_N 128     // mixtures up to two 64-entry chunks: the dense loop is as cheap
#endif
#ifndef WEIGHT_SPARSE_GROUP
#define WEIGHT_SPARSE_GROUP 16      // evaluation points per sweep over the mixture (one or two waves per particle; 8 with three)
#endif
#define WEIGHT_SPARSE_BITS 66.0f    // a pair is listed when one of its terms can be within 2^-66 of its sum (fp32 estimate of the log2 term, +-0.5)
#define WEIGHT_SPARSE_PRIOR_BITS 24.0f               // G: the sum before the update is taken to be >= 2^-G x the evaluation point's own term after it (checked)
#define WEIGHT_SPARSE_PRIOR_SCALE 5.9604644775390625e-08   // 2^-G
#ifndef WEIGHT_EVAL_GROUP
#define WEIGHT_EVAL_GROUP 8  // evaluation points whose sums a wave keeps in registers per pass over the mixture
#endif
#ifndef WEIGHT_WAVES_PER_EU
#define WEIGHT_WAVES_PER_EU 4  // <= 128 VGPRs: with 2 waves per particle all ~2000 particles of C2 are resident at once
#endif
// One workgroup of WPP waves per particle.  The entry-parallel steps (rank sort, sorted write-out, likelihood table) use
// all threads; the intensity sums split the evaluation points between the waves (groups of 8); the serial steps
// (evaluation-point selection, components, partition enumeration) run on wave 0 while wave 1 takes the weight sums.
// permOut (LDS, [cap] u16) != null: instead of writing the weight-sorted mixture to the other slab (what the stand-alone kernel
// does -- the order GaussianMixture::merge then walks), only the sorting permutation is left there for the merge phase of the
// fused step kernel, which reads the slab through it; `dst` is then unused.
template <int WPP>
__device__ __forceinline__ void phd_weight_particle(const Buffers &B, const Params &P, const int src, const int dst, const int nZ, const int evalCap,
                                                    const MurtyQueue &Q, const int i, const int tid, unsigned char *smem_raw,
                                                    unsigned short *permOut = nullptr) {
  constexpr int NT = WPP * 64;
  double *sZ = reinterpret_cast<double *>(smem_raw);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  auto block_sync = [&]() { if (WPP == 1) wave_sync(); else __syncthreads(); };
  if (!permOut) for (int t = tid; t < 2 * nZ; t += NT) sZ[t] = B.Z[t];   // (inside the fused step kernel the staged set of the map update is still there)
  WeightLDS s;
  const size_t perBytes = weight_lds_bytes_per_wave(B.cap, evalCap, nZ);
  carve_weight_lds(smem_raw + RFS_Z_LDS_BYTES, B.cap, evalCap, nZ, s);
  double *sScr = reinterpret_cast<double *>(smem_raw + RFS_Z_LDS_BYTES + perBytes);  // [0] sumPrev [1] sumCur
  int *sScrI = reinterpret_cast<int *>(sScr + 4);                                           // [0] nE [1] missing-rank flag

  const int N = B.count[i];
  const double *sl = B.slab[src];
  double *dl = B.slab[dst];
  const double *qW = plane((double *)sl, B.cap, i, PL_W), *qWP = plane((double *)sl, B.cap, i, PL_WP);
  const double *qMX = plane((double *)sl, B.cap, i, PL_MX), *qMY = plane((double *)sl, B.cap, i, PL_MY);
  const double *qSXX = plane((double *)sl, B.cap, i, PL_SXX), *qSXY = plane((double *)sl, B.cap, i, PL_SXY),
               *qSYY = plane((double *)sl, B.cap, i, PL_SYY);

  // nEvalPoints (:735-745); evalCount = -1 => all (unsigned compare)
  int nEvalPoints = ((unsigned)P.evalCount > (unsigned)N) ? N : P.evalCount;
  if (nEvalPoints == 0) {
    // weight := denorm_min, mixture NOT sorted (:742-745): copy through unchanged
    if (permOut) {
      for (int m = tid; m < N; m += NT) permOut[m] = (unsigned short)m;
    } else {
      for (int pl = 0; pl < PL_COUNT; pl++) {
        const double *q = plane((double *)sl, B.cap, i, pl);
        double *d = plane(dl, B.cap, i, pl);
        for (int m = tid; m < N; m += NT) d[m] = q[m];
      }
    }
    if (tid == 0) B.weight[i] = RFS_DENORM_MIN;
    return;
  }

  DBG_TB(16, 0);
#ifdef RFS_PROFILE
  const long long dbgT0 = (long long)__builtin_readcyclecounter();
#endif
  // ---- 1. sort by weight: rank = #{ j : w_j > w_m  or (w_j == w_m and j < m) } ----  (exact, fp64, ties by index)
  // Chunk-wise: every 64-entry chunk (one wave's entries: its keys are LDS broadcast reads) is rank-sorted on its own --
  // 64 compares per entry; a wave whose ranks do not add up to 0 + 1 + ... has tied keys and redoes its chunk with the
  // index tie-break (births share one weight, so ties are ordinary in a running filter) -- then an entry's rank is its
  // chunk rank plus, per other chunk, the number of keys ahead of it there: a 7-probe binary search in that sorted chunk,
  // counting ">=" in chunks of lower indices and ">" in chunks of higher ones.  64 + 7 N/64 probes per entry, not N.
  double *sorted = reinterpret_cast<double *>(s.perm);   // perm + fkeys: 8 bytes per entry; perm is written after the searches
  const int nChunks = (N + 63) >> 6;
  for (int m = tid; m < N; m += NT) s.keys[m] = qW[m];
  block_sync();
  constexpr int NS = 8;                                    // entries per thread held in registers
  bool tiedKeys = false;                                   // some thread has seen two equal weights (all-pairs form: not looked for, assumed)
  if (N <= NS * NT) {
    int rl[NS];
    const int logNB = rank_sort_log_buckets(B.cap);
    unsigned *hist = reinterpret_cast<unsigned *>(s.perm);  // histogram + bucket-ordered list inside the 8 B per entry of perm + fkeys
    if (!bucket_rank_sort<WPP, NS>([&](int m) { return s.keys[m]; }, N, tid, hist, reinterpret_cast<unsigned short *>(hist + (1 << logNB) / 2 + 4), logNB,
                                   s.labR, rl, block_sync, tiedKeys)) {
    tiedKeys = true;   // (crowded buckets ARE equal weights)
    block_sync();
#pragma unroll
    for (int k = 0; k < NS; k++) {
      const int m = tid + NT * k;                          // (m & 63) == lane: the wave's 64 entries are chunk m >> 6
      const int base = m & ~63;
      rl[k] = 0;
      if (base < N) {                                      // wave-uniform
        const int len = min(64, N - base);
        const double km = (m < N) ? s.keys[m] : 0.0;
        const double *ck = s.keys + base;
        int r = 0;
#pragma unroll 8
        for (int j = 0; j < len; j++) r += (ck[j] > km) ? 1 : 0;
        if (wave_sum_i_dpp((m < N) ? r : 0) != len * (len - 1) / 2) {   // tied keys in this chunk
          r = 0;
          for (int j = 0; j < len; j++) { const double kj = ck[j]; r += ((kj > km) | ((kj == km) & (j < lane))) ? 1 : 0; }
        }
        rl[k] = r;
        if (m < N) sorted[base + r] = km;
        else sorted[m] = -1.7976931348623157e308;            // tail of the last chunk: never ahead of anything
      }
    }
    block_sync();
    DBG_TB(16, 9);
    RFS_CUT(10);
#pragma unroll
    for (int k = 0; k < NS; k++) {
      const int m = tid + NT * k;
      if ((m & ~63) < N) {
        const int c = m >> 6;
        const double km = (m < N) ? s.keys[m] : 0.0;
        int rank = rl[k];
        // chunks of lower indices count ">=", chunks of higher ones ">" (two uniform loops: no per-probe select); the last
        // chunk's tail holds -DBL_MAX sentinels, so no probe needs a length check; cnt + st - 1 <= 63 by construction
        auto search4 = [&](int bFirst, int bEnd, auto pred) {
          for (int b0 = bFirst; b0 < bEnd; b0 += 4) {      // four searches in flight
            const double *p0 = sorted + 64 * b0, *p1 = sorted + 64 * min(b0 + 1, bEnd - 1), *p2 = sorted + 64 * min(b0 + 2, bEnd - 1),
                         *p3 = sorted + 64 * min(b0 + 3, bEnd - 1);
            int c0 = 0, c1 = 0, c2 = 0, c3 = 0;
#pragma unroll
            for (int st = 32; st >= 1; st >>= 1) {
              c0 += pred(p0[c0 + st - 1]) ? st : 0;
              c1 += pred(p1[c1 + st - 1]) ? st : 0;
              c2 += pred(p2[c2 + st - 1]) ? st : 0;
              c3 += pred(p3[c3 + st - 1]) ? st : 0;
            }
            c0 += pred(p0[c0]) ? 1 : 0;
            c1 += pred(p1[c1]) ? 1 : 0;
            c2 += pred(p2[c2]) ? 1 : 0;
            c3 += pred(p3[c3]) ? 1 : 0;
            rank += c0 + ((b0 + 1 < bEnd) ? c1 : 0) + ((b0 + 2 < bEnd) ? c2 : 0) + ((b0 + 3 < bEnd) ? c3 : 0);
          }
        };
        search4(0, c, [&](double v) { return v >= km; });
        search4(c + 1, nChunks, [&](double v) { return v > km; });
        rl[k] = rank;
      }
    }
    }
    {   // the workgroup's verdict on equal weights rides on the barrier that is due anyway
      const bool any = __ballot(tiedKeys) != 0ull;
      if (lane == 0) s.labR[8 + wave] = any ? 1 : 0;       // (xscr of the sort: words 0 .. 2 WPP)
    }
    block_sync();                                          // all ranks known: the sort's scratch is dead, perm may be written
    tiedKeys = false;
#pragma unroll
    for (int w2 = 0; w2 < WPP; w2++) tiedKeys |= s.labR[8 + w2] != 0;
#pragma unroll
    for (int k = 0; k < NS; k++) {
      const int m = tid + NT * k;
      if (m < N) s.perm[rl[k]] = m;
    }
  } else {  // more entries than the register slots cover: all-pairs ranking
    tiedKeys = true;
    for (int m = tid; m < N; m += NT) {
      const double wm = s.keys[m];
      int rank = 0;
      for (int j = 0; j < N; j++) {
        const double wj = s.keys[j];
        rank += ((wj > wm) | ((wj == wm) & (j < m))) ? 1 : 0;
      }
      s.perm[rank] = m;                                    // (perm does not alias keys)
    }
  }
  DBG_TB(16, 10);
  RFS_CUT(11);
  block_sync();
  // ---- 1b. equal weights in the order std::sort leaves them (stdsort_replay.h): s.perm holds the stable order (ties by index); the
  // partition phase of libstdc++'s introsort is replayed on 32-bit words (group | rank) in the dead float-key array, the ranks'
  // positions go to the unused upper halves of s.perm's words, runs of tied ranks are reordered by position.  Nothing happens for
  // mixtures of <= 16 entries or without equal weights (the flag comes out of the rank sort).
  StdSortScratch ss;
  {
    ss.T = reinterpret_cast<unsigned *>(s.fkeys);                              // [N] words   (4 N <= 4 cap64 bytes; the float keys are dead)
    ss.pos = reinterpret_cast<unsigned short *>(s.perm) + 1;                   // upper half of s.perm[r]
    ss.posStride = 2;
    unsigned char *cbuf = reinterpret_cast<unsigned char *>(s.compRows);       // compRows | compCols | partLik: 3072 B, written only later
    ss.eq = reinterpret_cast<unsigned long long *>(cbuf);                      // [<= 32]
    ss.stack = reinterpret_cast<unsigned *>(cbuf + 256);                       // [<= 28]
    const int LC = ss_list_cap(N);
    if ((size_t)LC * 4 <= 3072 - 256 - 112) {                                  // N <= 1350: the stopper lists fit
      ss.Ll = reinterpret_cast<unsigned short *>(cbuf + 256 + 112);
      ss.Rl = ss.Ll + LC;
    } else {
      ss.Ll = nullptr; ss.Rl = nullptr;                                        // (lane 0 replays serially)
    }
  }
  auto keyOf = [&](int e) { return s.keys[e]; };
  auto entryOf = [&](int r) { return (int)(unsigned short)s.perm[r]; };
  auto setEntryOf = [&](int r, unsigned short e) { s.perm[r] = (int)e; };
  // The correction is a chain of dependent LDS round trips that ONE wave walks (measured at C2b, where every mixture holds a run
  // of tied birth weights: +10 us per step when the whole workgroup waits for it).  In the fused step it therefore runs on wave 0
  // alone, BESIDE the other waves' intensity strand, which is the longer one by more than that: wave 0 first looks whether a tie
  // can touch the evaluation points at all (an eq bit among the ranks whose weight reaches importanceWeightingEvalPointGuassianWeight);
  // if not -- the ordinary case: ties sit at the birth weight -- it picks the evaluation points from the stable order, releases
  // the other waves, and corrects the order afterwards; otherwise the correction comes first.  Same results either way.
  constexpr bool SPLITTABLE = WPP > 1 && WEIGHT_W0_SHARE_NUM == 0;
  const bool splitEarly = SPLITTABLE && (size_t)B.cap * 4 >= 128 * sizeof(double);
  const bool overlap = splitEarly && permOut != nullptr && tiedKeys && N > SS_THRESHOLD;
  if (!overlap) {
    ss_correct_tie_order<WPP>(keyOf, entryOf, setEntryOf, [](unsigned *) {}, N, N, ss, tid, block_sync, tiedKeys);
    // sorted mixture -> other slab (or just the permutation, for the fused step's merge phase)
    if (permOut) {
      for (int r = tid; r < N; r += NT) permOut[r] = (unsigned short)s.perm[r];
    } else
    for (int r = tid; r < N; r += NT) {
      const int m = s.perm[r];
      // all gathers first (independent loads in flight together), then the coalesced stores
      const double v0 = s.keys[m], v1 = qWP[m], v2 = qMX[m], v3 = qMY[m], v4 = qSXX[m], v5 = qSXY[m], v6 = qSYY[m];
      plane(dl, B.cap, i, PL_W)[r] = v0;
      plane(dl, B.cap, i, PL_WP)[r] = v1;
      plane(dl, B.cap, i, PL_MX)[r] = v2;
      plane(dl, B.cap, i, PL_MY)[r] = v3;
      plane(dl, B.cap, i, PL_SXX)[r] = v4;
      plane(dl, B.cap, i, PL_SXY)[r] = v5;
      plane(dl, B.cap, i, PL_SYY)[r] = v6;
    }
  }
  DBG_TB(16, 8);
  RFS_CUT(12);

  DBG_TB(16, 1);
  PoseReg pr;
  load_pose(B, P, i, pr);

  // ---- 2. evaluation points: first <= nEvalPoints sorted entries with w >= minW and Pd > 0 (:747-762) ----  (wave 0)
  auto select_eval_points = [&]() {
    int nE = 0;
    bool evalOverflow = false;
    const int limit = nEvalPoints < RFSGPU_MAX_EVAL ? nEvalPoints : RFSGPU_MAX_EVAL;
    bool done = false;
    for (int c0 = 0; c0 < N && !done; c0 += 64) {
      const int r = c0 + lane;
      bool below = true, cand = false;
      double mx = 0, my = 0, pd = 0;
      int mOf = 0;
      if (r < N) {
        const int m = (int)(unsigned short)s.perm[r];          // (the upper halves may hold the correction's scratch)
        mOf = m;
        below = s.keys[m] < P.evalMinW;
        mx = qMX[m];
        my = qMY[m];
        double dx = mx - pr.x, dy = my - pr.y;
        bool close;
        pd = rb_pd(P, sqrt(dx * dx + dy * dy), close);
        cand = pd > 0;
      }
      unsigned long long belowMask = __ballot(below);
      unsigned long long valid = belowMask ? ((1ull << __builtin_ctzll(belowMask)) - 1ull) : ~0ull;
      if (belowMask) done = true;
      unsigned long long candMask = __ballot(cand) & valid;
      const int need = limit - nE;
      const int before = __popcll(candMask & ((1ull << lane) - 1ull));
      if (((candMask >> lane) & 1ull) && before < need) {
        const int e = nE + before;
        s.evX[e] = mx;
        s.evY[e] = my;
        s.evPd[e] = pd;
        s.evLog1mPd[e] = log(1 - pd);
        s.evIdx[e] = r | (mOf << 16);    // sorted position | the Gaussian's index
      }
      int got = __popcll(candMask);
      if (got >= need) { got = need; done = true; }
      nE += got;
    }
    // more evaluation points requested than the device path holds: refuse loudly (conservative)
    if (nE == limit && nEvalPoints > limit) evalOverflow = true;
    if (evalOverflow && lane == 0) atomicOr(B.err, ERRBIT_EVALPTS);
    if (lane == 0) sScrI[0] = nE;
  };
  // ---- 3a. weight sums (:765-775) ----  (the last wave, alongside step 2)
  auto weight_sums = [&]() {
    double sumPrev = 0.0, sumCur = 0.0;
    for (int m = lane; m < N; m += 64) { sumPrev += qWP[m]; sumCur += s.keys[m]; }
    sumPrev = wave_sum_dpp(sumPrev);
    sumCur = wave_sum_dpp(sumCur);
    if (lane == 0) { sScr[0] = sumPrev; sScr[1] = sumCur; }
  };
  if (!overlap) {
    if (wave == 0) select_eval_points();
    if (wave == WPP - 1) weight_sums();
    block_sync();
  } else if (wave == 0) {
    // ranks whose weight reaches the evaluation points' threshold: the leading rCut ranks of the sorted order
    int rCut = 0;
    for (int m0 = 0; m0 < N; m0 += 64) rCut += __popcll(__ballot(m0 + lane < N && !(s.keys[(m0 + lane < N) ? m0 + lane : 0] < P.evalMinW)));
    bool released = false;
    auto wsync = [&]() { wave_sync(); };
    auto after_eq = [&]() {             // (the eq words are complete here)
      bool touch = false;
      for (int c = 0; 64 * c < rCut; c++) {
        unsigned long long w = ss.eq[c];
        if (64 * c + 63 >= rCut) w &= (rCut - 64 * c >= 64) ? ~0ull : ((1ull << (rCut - 64 * c)) - 1ull);   // ranks < rCut
        touch |= w != 0ull;
      }
      if (!touch) { select_eval_points(); block_sync(); released = true; }
    };
    ss_correct_tie_order<1>(keyOf, entryOf, setEntryOf, [](unsigned *) {}, N, N, ss, lane, wsync, true, after_eq);
    if (!released) { select_eval_points(); block_sync(); }
    for (int r = lane; r < N; r += 64) permOut[r] = (unsigned short)s.perm[r];
  } else {
    if (wave == WPP - 1) weight_sums();
    block_sync();
  }
  const int nE = sScrI[0];

  DBG_TB(16, 2);
  RFS_CUT(13);
  // ---- 3b / 4-6: two independent strands after the evaluation points are known ----
  //   intensity at the evaluation points (:776-800): groups of EG points in registers, a pass over the mixture per group;
  //   likelihood table (:847-863) -> partitions -> assignment sums (:865-990).
  // With one wave they run one after the other.  With several, wave 0 takes the table and the (largely serial) partition
  // work while the other waves share the intensity groups: neither strand reads what the other writes (the sums go to the
  // rank-sort permutation's storage, free since step 2).
  constexpr int EG = WEIGHT_EVAL_GROUP;
  const bool split = WPP > 1 && (size_t)B.cap * 4 >= 128 * sizeof(double);  // (the sort's scratch, perm + fkeys = 8 B per entry, must hold 4 x 64 doubles)
  // where the sums go: the permutation's storage (4 x 64 doubles, free since step 2) -- except while the tie-order correction is still
  // using it on wave 0 (the overlapped form above): then their own 2 evalCap doubles
  // (only then: with the sums always in their own region the fused step at configs[1] was 0.8 us slower, r04 A/B)
  const bool ownSums = split && WEIGHT_W0_SHARE_NUM == 0 && overlap;
  const int sumStride = ownSums ? evalCap : 64;
  double *sumB = ownSums ? s.isum : reinterpret_cast<double *>(split ? (void *)s.perm : (void *)s.compRows), *sumA = sumB + sumStride;
  double *sumB0 = sumA + 64, *sumA0 = sumB0 + 64;   // (sharing knob only) wave 0's partial sums over ITS share of the mixture
  const int iw = split ? wave - 1 : wave, nIw = split ? WPP - 1 : WPP;  // this wave's share of the intensity groups
  // In split mode wave 0 can also take the intensity sums over the last WEIGHT_W0_SHARE_NUM / 5 of the mixture's 64-entry chunks once
  // it is through with its own strand; the two partial sums of an evaluation point are then added in a fixed order.
  const int nChunksI = (N + 63) >> 6;
  const int w0Chunks = (split && nChunksI >= 3) ? (WEIGHT_W0_SHARE_NUM * nChunksI) / 5 : 0;
  const int mSplit = (split && w0Chunks > 0) ? 64 * (nChunksI - w0Chunks) : N;   // waves >= 1: entries [0, mSplit); wave 0: [mSplit, N)
  // groups e0 = EG * first, step EG * stride; entries [mLo, mHi); results of evaluation point e to outB[e] / outA[e]
  auto intensity = [&](const int mLo, const int mHi, const int first, const int stride, double *outB, double *outA) {
    for (int e0 = EG * first; e0 < nE; e0 += EG * stride) {
      double accB[EG], accA[EG];
#pragma unroll
      for (int t = 0; t < EG; t++) { accB[t] = 0.0; accA[t] = 0.0; }
      // the evaluation points are re-read from LDS (broadcast) inside the pair loop: two LDS reads per pair cost less
      // than the 4*EG registers that holding them would take from the accumulators' budget (128 VGPRs at 4 waves/SIMD)
      const double *gx = s.evX + e0, *gy = s.evY + e0;
      const int nG = (nE - e0 < EG) ? nE - e0 : EG;
      for (int m = mLo + lane; m < mHi; m += 64) {
        const double w = s.keys[m], wp = qWP[m], mx = qMX[m], my = qMY[m];
        const double sxx = qSXX[m], sxy = qSXY[m], syy = qSYY[m];
        double i00, i01, i10, i11, det;
        inv2(sxx, sxy, sxy, syy, i00, i01, i10, i11, det);
        // 1 / sqrt((2 pi)^2 det) once per Gaussian: the pair loop multiplies instead of dividing (<= 1.5 ulp from the
        // reference's exp(.)/factor, far inside the 1e-9 weight tolerance) -- a division per pair costs as much as the exp
        const double rfac = 1.0 / pdf_factor2(det);
        int opaque = 0;
        asm volatile("" : "+v"(opaque));  // keeps the (loop-invariant) LDS reads below inside the loop
#pragma unroll
        for (int t = 0; t < EG; t++) {
          const int tt = ((t < nG) ? t : 0) + opaque;
          const double d0 = gx[tt] - mx, d1 = gy[tt] - my;
          const double t0 = d0 * i00 + d1 * i01, t1 = d0 * i01 + d1 * i11;  // (i10 == i01: Sigma is stored symmetric)
          const double md2 = t0 * d0 + t1 * d1;
          double lik = (md2 > 1500.0) ? 0.0 : rfs_exp(-0.5 * md2) * rfac;  // exactly 0 beyond 1500 (gauss_from_md2)
          lik = (lik != lik) ? 0.0 : lik;                                // NaN -> 0 (include/RandomVec.hpp:417-434)
          accB[t] += wp * lik;
          accA[t] += w * lik;
        }
      }
#pragma unroll
      for (int t = 0; t < EG; t++) {
        const double b = wave_sum_dpp(accB[t]), a = wave_sum_dpp(accA[t]);
        if (lane == 0 && e0 + t < nE) { outB[e0 + t] = b; outA[e0 + t] = a; }
      }
    }
  };
  // ---- the same sums over the pairs that can reach them (WEIGHT_SPARSE_INTENSITY) ----
  // Of the nE x N (evaluation point, Gaussian) pairs only a few per cent contribute to a sum at all: a Gaussian ten of its own standard
  // deviations away adds less than 2^-64 of the sum's largest term, i.e. nothing that survives the fp64 addition.  So the dense fp64 loop
  // above (distance, exp, two multiply-adds for every pair) is replaced by
  //   sweep   packed fp32 over all pairs: log2 of the pair's terms, v = c_m - md2 log2(e) / 2 with c_m = log2(weight / factor) after (w) and
  //           before (w_prev) the update.  An evaluation point IS the mean of a Gaussian of the mixture, so its sum after the update is at
  //           least that Gaussian's own term, ownA_e (md2 = 0); the sum before the update is ASSUMED to be at least 2^-G ownA_e (the parent
  //           landmark of an updated copy sits inside the innovation gate of it: typically 2^-12) and the assumption is checked afterwards.  A
  //           pair is LISTED when vA > log2(ownA_e) - BITS or vB + G > log2(ownA_e) - BITS (BITS = 64 + 2 for the fp32 rounding of v, bounded
  //           below) -- ballot + mbcnt compaction into a per-evaluation-point list in LDS;
  //   exact   the listed pairs in fp64 with the dense loop's own formulas (inv2, pdf_factor2, rfs_exp), one evaluation point per 16-lane
  //           row, summed per lane in list order (Gaussian index ascending) and over the row by a fixed DPP tree;
  //   check   the sum before the update over the listed pairs (a lower bound of the full sum) reaches 2^-G ownA_e -- otherwise, or when a
  //           list overflows, that evaluation point is summed by the dense loop.
  // What is dropped: terms below 2^-64 of their sum, at most N of them -- together below 2^-55 of the sum (N <= 512), half an ulp; the
  // sums differ from the dense loop's by the order of the additions only.  A Gaussian whose fp32 image cannot be trusted (covariance not
  // positive definite or correlated beyond 0.9995, non-finite or negative weights, a standard deviation so small against the coordinates
  // that the fp32 difference could move v by half a bit) is listed for every evaluation point.  The result of an evaluation point does not
  // depend on the group it is in or the wave that takes it (1, 2 or 3 waves per particle give the same bits).
  constexpr int SG = (WPP >= 3) ? 8 : WEIGHT_SPARSE_GROUP;    // evaluation points in flight per wave
  constexpr int SEG = WEIGHT_SPARSE_SEG;
  auto intensity_dense_one = [&](const int e, double *outB, double *outA) {   // one evaluation point, lanes over the mixture
    double aB = 0.0, aA = 0.0;
    const double gx = s.evX[e], gy = s.evY[e];
    for (int m = lane; m < N; m += 64) {
      const double w = s.keys[m], wp = qWP[m], mx = qMX[m], my = qMY[m];
      double i00, i01, i10, i11, det;
      inv2(qSXX[m], qSXY[m], qSXY[m], qSYY[m], i00, i01, i10, i11, det);
      const double rfac = 1.0 / pdf_factor2(det);
      const double d0 = gx - mx, d1 = gy - my;
      const double t0 = d0 * i00 + d1 * i01, t1 = d0 * i01 + d1 * i11;
      const double md2 = t0 * d0 + t1 * d1;
      double lik = (md2 > 1500.0) ? 0.0 : rfs_exp(-0.5 * md2) * rfac;
      lik = (lik != lik) ? 0.0 : lik;
      aB += wp * lik;
      aA += w * lik;
    }
    aB = wave_sum_dpp(aB);
    aA = wave_sum_dpp(aA);
    if (lane == 0) { outB[e] = aB; outA[e] = aA; }
  };
  auto intensity_sparse = [&](const int first, const int stride, double *outB, double *outA, unsigned char *scr) {
    unsigned short *list = reinterpret_cast<unsigned short *>(scr);                    // [SG][SEG]
    double *ownAd = reinterpret_cast<double *>(scr + SG * SEG * 2);                     // [SG] 2^-G x the evaluation point's own term
    float *gxf = reinterpret_cast<float *>(ownAd + SG), *gyf = gxf + SG;                 // evaluation points relative to the pose
    float *nthA = gyf + SG;                                                            // BITS - log2(own term)
    int *cntL = reinterpret_cast<int *>(nthA + SG);
    const float ninf = -__builtin_huge_valf();
    for (int e0 = SG * first; e0 < nE; e0 += SG * stride) {
      const int nG = (nE - e0 < SG) ? nE - e0 : SG;
      wave_sync();                                   // (the previous group's lists have been read)
#ifdef RFS_PROFILE
      if (B.dbg && i == 7 && lane == 0) B.dbg[9] = (long long)__builtin_readcyclecounter();
#endif
      float gs = 0.f;
      if (lane < SG) {
        const int e = e0 + ((lane < nG) ? lane : 0);
        const float x = (float)(s.evX[e] - pr.x), y = (float)(s.evY[e] - pr.y);
        gxf[lane] = x; gyf[lane] = y;
        gs = __builtin_fmaxf(__builtin_fabsf(x), __builtin_fabsf(y));
        const int me = (int)((unsigned)s.evIdx[e] >> 16);                        // the Gaussian the point is the mean of
        const double sxx = qSXX[me], sxy = qSXY[me], syy = qSYY[me];
        const double own = s.keys[me] / pdf_factor2(sxx * syy - sxy * sxy);
        const bool ok = (own > 0.0) && (own < 1.0e300);                         // (a NaN / degenerate own term: list everything -> dense)
        int ex;
        const float mant = (float)__builtin_frexp(own, &ex);
        nthA[lane] = ok ? WEIGHT_SPARSE_BITS - (__builtin_amdgcn_logf(mant) + (float)ex) : 3.0e38f;
        ownAd[lane] = ok ? own * WEIGHT_SPARSE_PRIOR_SCALE : __builtin_huge_val();
      }
      const float gScale = 2.f * wave_max_f32(gs);   // (NaN coordinates: every pair of that point drops out, as its dense terms are 0)
      wave_sync();
      const f2_t *gx2 = reinterpret_cast<const f2_t *>(gxf), *gy2 = reinterpret_cast<const f2_t *>(gyf), *na2 = reinterpret_cast<const f2_t *>(nthA);
      int cnt[SG];
#pragma unroll
      for (int t = 0; t < SG; t++) cnt[t] = 0;
      bool anyB = false;
      int zero = 0;
      asm volatile("" : "+v"(zero));                  // one vector base per array, the pairs at immediate offsets
      for (int m0 = 0; m0 < N; m0 += 64) {
        // fp32 image of the lane's Gaussian: position relative to the pose, J = -log2(e)/2 Sigma^-1, c = log2(weight / factor)
        const int m = m0 + lane;
        const bool act = m < N;
        const int mm = act ? m : 0;
        const double w = s.keys[mm], wp = qWP[mm];
        const float a = (float)qSXX[mm], b = (float)qSXY[mm], c = (float)qSYY[mm];
        float fx = (float)(qMX[mm] - pr.x), fy = (float)(qMY[mm] - pr.y);
        const float ac = a * c;
        const float det = __builtin_fmaf(-b, b, ac);
        const float rdet = __builtin_amdgcn_rcpf(det);
        const float kr = -0.72134752044448170368f * rdet;
        float j00 = c * kr, j01 = -b * kr, j11 = a * kr;
        const float base = -2.65149612947231879804f - 0.5f * __builtin_amdgcn_logf(det);   // -log2(2 pi) - log2(det) / 2
        int exA, exB;                                  // log2 of a double of any magnitude (weights far below fp32's range keep their order)
        const float mantA = (float)__builtin_frexp(w, &exA), mantB = (float)__builtin_frexp(wp, &exB);
        const float cA = __builtin_amdgcn_logf(mantA) + (float)exA + base;
        const float cB = __builtin_amdgcn_logf(mantB) + (float)exB + base + WEIGHT_SPARSE_PRIOR_BITS;
        float cS = __builtin_fmaxf(cA, cB);
        // |dv| <= 2 |K| sqrt(md2 tr(Sigma^-1)) dd with dd <= 2^-24 S (S >= |g| + |m| per coordinate) and md2 <= 136 at the threshold:
        // tr(Sigma^-1) S^2 < 6e10 keeps it below a quarter of a bit; det > 1e-3 a c bounds the cancellation in det (relative 2^-13: 0.01 bit)
        const float S = gScale + __builtin_fabsf(fx) + __builtin_fabsf(fy);
        const bool sane = (a > 0.f) & (c > 0.f) & (det > 1e-3f * ac) & ((a + c) * rdet * S * S < 6.0e10f) & (w >= 0.0) & (wp >= 0.0) & (S < 1.0e6f);
        if (!sane) { j00 = 0.f; j01 = 0.f; j11 = 0.f; fx = 0.f; fy = 0.f; cS = 3.0e38f; }   // listed for every (finite) evaluation point
        if (!act) cS = ninf;
        anyB |= __ballot(act && !(wp == 0.0)) != 0ull;
        const f2_t vfx = {fx, fx}, vfy = {fy, fy}, v00 = {j00, j00}, v01 = {j01, j01}, v11 = {j11, j11}, vc = {cS, cS};
        bool sg[SG];
#pragma unroll
        for (int q = 0; q < SG / 2; q++) {
          const f2_t d0 = gx2[q + zero] - vfx, d1 = gy2[q + zero] - vfy;    // (uniform addresses: LDS broadcast reads)
          const f2_t t0 = d0 * v00 + d1 * v01, t1 = d0 * v01 + d1 * v11;
          const f2_t h = t0 * d0 + t1 * d1;
          const f2_t u = (h + vc) + na2[q + zero];
          sg[2 * q] = u.x > 0.f;
          sg[2 * q + 1] = u.y > 0.f;
        }
#pragma unroll
        for (int t = 0; t < SG; t++) {
          const unsigned long long mk = __ballot(sg[t]);
          const int c0 = cnt[t], pc = __popcll(mk);
          if (c0 + pc <= SEG) {                      // (wave-uniform)
            const int pos = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mk >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mk, 0u));
            if (sg[t]) list[t * SEG + c0 + pos] = (unsigned short)m;
          }
          cnt[t] = c0 + pc;
        }
      }
      unsigned over = 0;                              // evaluation points whose list overflowed: dense
#ifdef RFS_PROFILE
      if (B.dbg && i == 7 && lane == 0) {
        B.dbg[10] = (long long)__builtin_readcyclecounter();
        long long tot = 0, trips = 0;
        for (int t = 0; t < SG; t++) tot += cnt[t];
        for (int q4 = 0; q4 < SG / 4; q4++) { int mx4 = 0; for (int r = 0; r < 4; r++) mx4 = cnt[4 * q4 + r] > mx4 ? cnt[4 * q4 + r] : mx4; trips += (mx4 + 15) / 16; }
        B.dbg[12] = tot; B.dbg[13] = trips; B.dbg[15] = nE;
      }
#endif
#pragma unroll
      for (int t = 0; t < SG; t++) {
        if (lane == t) cntL[t] = (cnt[t] > SEG) ? 0 : cnt[t];
        if (cnt[t] > SEG && t < nG) over |= 1u << t;
      }
      wave_sync();
      // ---- exact terms of the listed pairs: one evaluation point per 16-lane row ----
#pragma unroll 1
      for (int eq = 0; eq < SG / 4; eq++) {
        if (4 * eq >= nG) break;
        const int eL = eq * 4 + (lane >> 4);
        const int c = cntL[eL];
        const int cMax = (int)wave_max_u32((unsigned)c);
        const int e = e0 + ((eL < nG) ? eL : 0);
        const double gx = s.evX[e], gy = s.evY[e];
        double aB = 0.0, aA = 0.0;
        // one listed pair: the dense loop's term with 1 / sqrt(det) from v_rsq_f64 + two Newton steps standing in for its two divisions and
        // square root (1 / det = r^2, 1 / factor = r / 2 pi: <= 3 ulp on the term, as far inside the 1e-9 weight tolerance as the dense loop's 1.5)
        auto term = [&](const int slot, double &tB, double &tA) {
          const bool v = slot < c;
          int m = 0;
          if (v) m = (int)list[eL * SEG + slot];
          const double w = s.keys[m], wp = qWP[m], mx = qMX[m], my = qMY[m];
          const double sxx = qSXX[m], sxy = qSXY[m], syy = qSYY[m];
          const double det = sxx * syy - sxy * sxy;
          double r = __builtin_amdgcn_rsq(det);
          r = __builtin_fma(0.5 * r, __builtin_fma(-det * r, r, 1.0), r);
          r = __builtin_fma(0.5 * r, __builtin_fma(-det * r, r, 1.0), r);
          const double invdet = r * r, rfac = r * 0.15915494309189533577;
          const double d0 = gx - mx, d1 = gy - my;
          const double t0 = (d0 * syy - d1 * sxy) * invdet, t1 = (d1 * sxx - d0 * sxy) * invdet;   // Sigma^-1 d (inv2's entries x 1 / det)
          const double md2 = t0 * d0 + t1 * d1;
          double lik = (md2 > 1500.0) ? 0.0 : rfs_exp(-0.5 * md2) * rfac;
          lik = (lik != lik) ? 0.0 : lik;
          tB = v ? wp * lik : 0.0;
          tA = v ? w * lik : 0.0;
        };
        for (int j0 = 0; j0 < cMax; j0 += 32) {       // two blocks of 16 per trip: two independent chains in flight
          double b0, a0, b1, a1;
          term(j0 + (lane & 15), b0, a0);
          term(j0 + 16 + (lane & 15), b1, a1);
          aB += b0; aA += a0;
          aB += b1; aA += a1;
        }
        aB += dpp_f64<0xb1, 0xf>(aB); aA += dpp_f64<0xb1, 0xf>(aA);     // quad_perm [1,0,3,2]
        aB += dpp_f64<0x4e, 0xf>(aB); aA += dpp_f64<0x4e, 0xf>(aA);     // quad_perm [2,3,0,1]
        aB += dpp_f64<0x114, 0xf>(aB); aA += dpp_f64<0x114, 0xf>(aA);   // row_shr:4
        aB += dpp_f64<0x118, 0xf>(aB); aA += dpp_f64<0x118, 0xf>(aA);   // row_shr:8 -> lane 15 of the row
        const bool lead = (lane & 15) == 15 && eL < nG;
        if (lead) { outB[e0 + eL] = aB; outA[e0 + eL] = aA; }
        // the assumption behind the lists: the sum before the update reaches 2^-G of the point's own term (the listed part of it already does)
        const unsigned long long bad = __ballot(lead && anyB && !(aB >= ownAd[eL]));
#pragma unroll
        for (int r = 0; r < 4; r++) over |= ((bad >> (16 * r + 15)) & 1ull) ? (1u << (eq * 4 + r)) : 0u;
      }
#ifdef RFS_PROFILE
      if (B.dbg && i == 7 && lane == 0) { B.dbg[11] = (long long)__builtin_readcyclecounter(); B.dbg[14] = __popc(over); }
      if (B.dbg && lane == 0 && over) atomicAdd((unsigned long long *)&B.dbg[8], (unsigned long long)__popc(over));   // all particles: dense fall-backs since the buffer was cleared
#endif
      while (over) {                                  // (wave-uniform)
        const int t = __builtin_ctz(over);
        over &= over - 1;
        intensity_dense_one(e0 + t, outB, outA);
      }
    }
  };
#ifdef RFS_STOP_AT
  if ((RFS_STOP_AT == 14 || RFS_STOP_AT == 15) && split && wave > 0) __builtin_amdgcn_endpgm();   // wave 0's strand alone
#endif
  const bool sparseI = WEIGHT_SPARSE_INTENSITY && (split || WPP == 1) && w0Chunks == 0 && N > WEIGHT_SPARSE_MIN_N && N <= 65535 && !P.denseIntensity;
  if (!split || wave > 0) {
    if (sparseI) intensity_sparse(iw, nIw, sumB, sumA, s.sparse + ((split && WPP >= 3) ? (size_t)(wave - 1) * (WEIGHT_SPARSE_LDS_BYTES / 2) : 0));
    else intensity(0, mSplit, iw, nIw, sumB, sumA);
  }
  if (!split) block_sync();

  DBG_TB(16, 3);
  double l = 1.0;
  if (!split || wave == 0) {
    // ---- 4. likelihood table L[e][n] = N(z_n; h(x, e), S_e) * Pd_e, gated (:847-863) ----
    const int t0i = split ? lane : tid, tN = split ? 64 : NT;
    if (t0i < nE) {
      MeasOut mo;
      rb_measure(P, pr, s.evX[t0i], s.evY[t0i], 0.0, 0.0, 0.0, mo);  // evalPt_copy.setCov(Zero)
      double i00, i01, i10, i11, det;
      inv2(mo.s00, mo.s01, mo.s10, mo.s11, i00, i01, i10, i11, det);
      double *z = s.evZ + 7 * t0i;
      z[0] = mo.z0; z[1] = mo.z1; z[2] = i00; z[3] = i01; z[4] = i10; z[5] = i11; z[6] = pdf_factor2(det);
    }
    if (split) wave_sync(); else block_sync();
    auto table_md2 = [&](int idx, int &e) -> double {
      e = idx / nZ;
      const int n = idx - e * nZ;
      const double *z = s.evZ + 7 * e;
      const double d0 = sZ[2 * n] - z[0], d1 = sZ[2 * n + 1] - z[1];
      const double t0 = d0 * z[2] + d1 * z[4], t1 = d0 * z[3] + d1 * z[5];
      return t0 * d0 + t1 * d1;
    };
    if (split || WPP == 1) {
      // One wave fills the table.  Almost every cell fails the Mahalanobis gate (:855-859) and is 0: the sweep computes
      // md2 only and collects the few cells inside the gate; the Gaussian (exp) is then evaluated densely over that list.
      int *list = s.labR;                       // [128]: labR + labC, written only later by the component search
      int nList = 0;
      for (int i0 = 0; i0 < nE * nZ; i0 += 64) {
        const int idx = i0 + lane;
        bool in = false;
        if (idx < nE * nZ) {
          int e;
          const double md2 = table_md2(idx, e);
          in = !(md2 > P.weightingMd2);         // (a NaN md2 stays in: its likelihood is 0 through the NaN guard)
          s.L[idx] = 0.0;
        }
        const unsigned long long m = __ballot(in);
        const int pos = nList + __popcll(m & ((1ull << lane) - 1ull));
        if (in) {
          if (pos < 128) list[pos] = idx;
          else {                                 // list full: evaluate on the spot
            int e;
            const double md2 = table_md2(idx, e);
            s.L[idx] = gauss_from_md2(md2, s.evZ[7 * e + 6]) * s.evPd[e];
          }
        }
        nList += __popcll(m);
      }
      wave_sync();
      nList = nList < 128 ? nList : 128;
      for (int q = lane; q < nList; q += 64) {
        const int idx = list[q];
        int e;
        const double md2 = table_md2(idx, e);
        s.L[idx] = gauss_from_md2(md2, s.evZ[7 * e + 6]) * s.evPd[e];
      }
    } else {
      for (int idx = t0i; idx < nE * nZ; idx += tN) {
        int e;
        const double md2 = table_md2(idx, e);
        double Lv = gauss_from_md2(md2, s.evZ[7 * e + 6]) * s.evPd[e];
        if (md2 > P.weightingMd2) Lv = 0.0;
        s.L[idx] = Lv;
      }
    }
    if (split) wave_sync(); else block_sync();
  }
  double prodBefore = 1.0, prodAfter = 1.0;
  if (!split) {
    if (wave != 0) return;  // the rest is wave 0's
    // the products over the evaluation points, in order (the sums were left in LDS by the waves)
    for (int e = 0; e < nE; e++) {
      prodBefore *= (RFS_DENORM_MIN + sumB[e]);
      prodAfter *= (RFS_DENORM_MIN + sumA[e]);
    }
    wave_sync();  // (sumA / sumB live in the component scratch that step 5 reuses)
  }
  DBG_TB(16, 4);
  RFS_CUT(14);
#ifdef RFS_PROFILE
  const long long dbgT4 = (long long)__builtin_readcyclecounter();
#endif
  if (wave == 0) {
    // ---- 5./6. partition the table, sum the assignments of every partition (shared with the 3-D kernel) ----
#ifdef RFS_PROFILE
    l = rfs_partitions_wave(s, nE, nZ, P.clutter, lane, i, Q, B.err, P.exactPartitions, B.dbg);
#else
    l = rfs_partitions_wave(s, nE, nZ, P.clutter, lane, i, Q, B.err, P.exactPartitions);
#endif
  }
  RFS_CUT(15);
  if (split) {
    if (wave == 0 && w0Chunks > 0) intensity(mSplit, N, 0, 1, sumB0, sumA0);   // wave 0's share of the mixture
    block_sync();          // the intensity sums of the other waves are complete
    if (wave != 0) return;
    for (int e = 0; e < nE; e++) {
      const double b = (w0Chunks > 0) ? sumB[e] + sumB0[e] : sumB[e], a = (w0Chunks > 0) ? sumA[e] + sumA0[e] : sumA[e];
      prodBefore *= (RFS_DENORM_MIN + b);
      prodAfter *= (RFS_DENORM_MIN + a);
    }
  }
  const double sumPrev = sScr[0], sumCur = sScr[1];
  const double sensingArea = 2 * RFS_PI * (P.rmax - P.rmin);
  const double ml = l / (P.clutter * sensingArea);  // clutterIntensityIntegral (src/MeasurementModel_RngBrg.cpp:175-178)

  DBG_TB(16, 6);
  // ---- 7. overall weight (:806-811) ----
  const double overall = ml * prodBefore / prodAfter * exp(sumCur - sumPrev);
  if (lane == 0) {
    const double wnew = overall * B.weight[i];
    B.weight[i] = wnew;
  }
  DBG_TB(16, 7);
#ifdef RFS_PROFILE
  if (B.dbg && tid == 0) {
    long long *d = B.dbg + 64 + 4 * (size_t)i;
    const long long t = (long long)__builtin_readcyclecounter();
    d[0] = t - dbgT0; d[1] = t - dbgT4; d[2] = nE; d[3] = N;
  }
#endif
}

template <int WPP>
__global__ __launch_bounds__(WPP * 64) __attribute__((amdgpu_waves_per_eu(WPP == 1 ? 2 : WEIGHT_WAVES_PER_EU)))
void phd_weight_multifeature_kernel(Buffers B, Params P, int src, int dst, int nZ, int evalCap, MurtyQueue Q) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  phd_weight_particle<WPP>(B, P, src, dst, nZ, evalCap, Q, (int)blockIdx.x, (int)threadIdx.x, smem_raw);
}
