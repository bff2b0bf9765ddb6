// merge_prune.h -- gm_merge: GaussianMixture::merge (reference include/GaussianMixture.hpp:394-475) and
// gm_prune: GaussianMixture::prune + sortByWeight (:477-534).
//
// gm_merge keeps the reference's sequential-greedy semantics exactly: for i ascending, j ascending > i, j is
// absorbed into i as soon as md2_i(j) <= t^2 or md2_j(i) <= t^2, and i's mean/covariance change before j+1 is
// tested.  How that becomes parallel work (grid-based candidate lists against the initial states, speculative row
// replay with ordered validation) is described at gm_merge_particle below.  Holes (absorbed Gaussians; landmark == NULL,
// weight 0 in the reference) are written back with weight -1 so that gm_prune drops them.
//
// gm_prune: rank-sort the survivors (w >= threshold) by (weight desc, index asc) and compact them into the
// other slab.  The reference keeps exactly the sorted prefix with w >= t (binary search + linear walk).
#pragma once
#include "common.h"
#include "weighting.h"  // wave_sync

// LDS per particle, 28 B per entry: position (x, y) and prefilter RADIUS as fp32 (the radius' sign is the liveness flag),
// weight as fp64 (prune's sort key, the merge's w_a + w_j test), row record (u32), grid-sorted index (u16), slack (u16); plus
// the 32x32 spatial grid (one cursor array, two 16-bit cursors per word) and the list of possible partners (u32 each).
// Everything the EXACT tests need (means, covariances) is read from the slab (HBM / L2) for the few pairs that survive the
// distance prefilter; the fp32 copies only feed the prefilter, whose thresholds carry the fp32 rounding bound, so it can
// let extra pairs through but never drop one.  (Until r02 the entries were 40 B -- x, y, bound as fp64 -- which set the
// fused kernel's LDS block and with it how many particles of the configs[2] shard are resident.)
// The grid has (1 << GL) x (1 << GL) cells; GL is a template parameter of the merge (5, or 6 where the host finds that the 6 KB of
// extra cursors do not cost a resident workgroup -- large mixtures, where cells of extent / 32 are several merge radii wide and
// the 3 x 3 neighbourhood of an entry holds a multiple of the entries it needs to).  Results do not depend on GL: the grid only
// proposes pairs, the exact tests decide, and a row that could meet partners beyond its listed ones falls back to the
// reference's scan.  (A run-time GL was tried first: its cell arithmetic cost 3 us at configs[1].)
#define MERGE_LOG_CELLS 10  // the fused prune's rank sort reuses the grid's cursor array as a 1024-bucket histogram (fits either grid)
__host__ __device__ constexpr int merge_cells(int gridLog) { return 1 << (2 * gridLog); }
#define MERGE_PAIR_CAP(cap) ((cap) > 320 ? (cap) : 320)  // listed partners per particle: ~0.7 per entry on dense maps
#define MERGE_ROW_SLOTS 8   // prefilter survivors one row can list; a row with more is replayed by the sequential scan
#ifndef MERGE_SCAN_SHIFTED_RADIUS
#define MERGE_SCAN_SHIFTED_RADIUS 1   // r06: the scan's rounding allowance is folded into the LDS radius once per entry (see the grid scatter)
#endif
// (Measured and dropped, r04 -- profiles/r04a_ab_merge_stage.txt: phase 1b leaving the replay's operands -- mean and covariance of
//  every passing partner and of its row -- in LDS, so that the one wave that replays reads LDS instead of dependent global loads:
//  fused step 125.2 -> 126.5 us at configs[1].  The loads the replay waits for are L2 hits issued together; the staging's stores
//  and slot atomics in the pair phase cost more than they save.
//  Also measured and dropped, r04 -- profiles/r04c_ab_flat_candidate_scan.txt: the candidate scan FLATTENED over (entry, neighbour)
//  items with every unordered pair visited once (own cell upwards, east cell, the three cells of the next row; items of all
//  entries laid end to end and dealt out in equal runs; survivors appended to an unordered list and grouped into the rows'
//  segments by a counting pass; bit-identical results, 200 fuzz cases clean): half the thread-per-entry form's vector
//  instructions on the busier wave, but eight more workgroup barriers, LDS atomics per far neighbour and 3 KB more LDS: fused
//  step 123.8-124.5 -> 125.0-125.2 us, stand-alone merge + prune 63.3 -> 62.2 us.  The phase is as long as its chain of dependent
//  LDS round trips and barriers, not as its instruction count.)
__host__ __device__ inline size_t merge_lds_bytes_per_wave(int cap, int gridLog = 5) {
  // entries: w (f64) + x, y, radius (f32) + row record (u32) + grid-sorted index (u16) + prefilter slack (u16); grid: CELLS/2+4 u32; pair list: u32
  return (((size_t)cap * (8 + 3 * 4 + 4 + 2 + 2)) + (size_t)(merge_cells(gridLog) / 2 + 4) * 4 + (size_t)MERGE_PAIR_CAP(cap) * 4 + 16 + 15) & ~(size_t)15;
}
// Row record (sRec[m]): [31:25] claim of the speculative round (lane, 0x7f = none) | [24] the row has a partner that
// passes the exact test against the initial states | [23:20] number of listed survivors (15 = not listable) |
// [19:0] offset of the row's contiguous segment in the pair list.
#define MERGE_REC_NOCLAIM 0xfe000000u
#ifndef MERGE_WALK_PREFETCH
#define MERGE_WALK_PREFETCH 1     // the replay's walk requests the next partner's operands before it merges the present one
#endif
#ifndef MERGE_VALIDATE_ROUNDS
#define MERGE_VALIDATE_ROUNDS 1   // phase 2 validates in sub-rounds (0: rows after the first conflict one by one, the form of rounds 2-5)
#endif
#define MERGE_REC_ISROW 0x01000000u
// + cross-wave reduction scratch ([waves][8] floats) when a workgroup of several waves works on one particle
__host__ __device__ inline size_t merge_lds_bytes_per_block(int cap, int wavesPerParticle, int gridLog = 5) {
  return merge_lds_bytes_per_wave(cap, gridLog) + (size_t)wavesPerParticle * 32;
}

// Necessary condition for a pair to pass the merge test: md2 = e^T S^-1 e >= |e|^2 / lambda_max(S) >= |e|^2 / tr(S),
// so d1 <= t^2 or d2 <= t^2 implies |e|^2 <= t^2 * max(tr S_a, tr S_j).  The bound carries a 1e-6 relative
// margin for rounding in the computed md2 / inverse; a non-PSD or non-finite covariance gets an infinite bound
// (always fully tested), so the prefilter never changes a decision of the exact test.
__device__ __forceinline__ double merge_bound(double t2, double xx, double xy, double yy) {
  const double tr = xx + yy;
  const double det = xx * yy - xy * xy;
  const bool sane = (det > 0.0) && (tr > 0.0) && (tr < 1.7e308);
  return sane ? t2 * tr * (1.0 + 1e-6) : __builtin_huge_val();
}
// The prefilter radius sqrt(bound) as an fp32 that is >= the exact value (sqrt and the conversion round by < 2e-7).
__device__ __forceinline__ float merge_radius_f32(double bound) {
  return __builtin_amdgcn_sqrtf((float)bound) * (1.f + 1e-6f) + 1e-37f;
}

// The exact pair test of GaussianMixture::merge (:434-447) for e = x_j - x_a:
//   d1 = e^T S_a^-1 e ; if d1 > t2: d2 = e^T S_j^-1 e (same value for -e); fail if d2 > t2 ; fail if w_a + w_j == 0.
// S_j is fetched from the slab only when d1 fails.
__device__ __forceinline__ bool merge_pair_passes(double e0, double e1, double a00, double a01, double a11, double aw, double jw,
                                                  const double *pSXX, const double *pSXY, const double *pSYY, int j, double t2) {
  const double u0 = e0 * a00 + e1 * a01, u1 = e0 * a01 + e1 * a11;
  bool far = (u0 * e0 + u1 * e1) > t2;
  if (far) {
    double j00, j01, j10, j11, det;
    const double xy = pSXY[j];
    inv2(pSXX[j], xy, xy, pSYY[j], j00, j01, j10, j11, det);
    const double g0 = -e0, g1 = -e1;
    const double v0 = g0 * j00 + g1 * j01, v1 = g0 * j01 + g1 * j11;
    far = (v0 * g0 + v1 * g1) > t2;
  }
  return !far && ((aw + jw) != 0.0);
}

// gm_merge (+ optional fused gm_prune).
// Phase 1 (parallel): every pair (a, j>a) is examined once against the INITIAL states -- which is exactly what the
//   reference's sequential scan sees the first time it meets a pair, because a Gaussian only changes while it is the
//   outer index.  Candidates come from a 32x32 uniform grid over the mixture's bounding box whose cell edges are >= the
//   largest prefilter radius, so every pair that can pass lies in adjacent cells; each entry writes the list of its
//   possible partners (higher indices only) and a slack bound for everything it did not list; the listed pairs are
//   tested exactly.  A non-finite bound makes the cell edge infinite: everything falls into one cell and the search
//   degrades to all pairs, still exact.
// Phase 2: rows with a passing partner are replayed with the exact greedy rule (merge the lowest passing j, update a,
//   re-test only j' > j against the new state, skip absorbed entries) -- speculatively one row per lane, validated in
//   ascending row order; see the comments in the kernel.
// FUSE_PRUNE: survivors (w >= pruneT, not absorbed) are rank-sorted by (weight desc, index asc) and compacted into
//   the other slab straight from here (GaussianMixture::prune :477-521), saving gm_prune's launch and re-read.
// `perm` (LDS, or null): the order the mixture is to be merged in, as a permutation of the slab's entries -- entry m of the
//   merge is slab entry perm[m].  The fused step kernel passes the weight-sorted order left by the weighting phase
//   (sortByWeight, include/RBPHDFilter.hpp:733), so that the sorted mixture is never written out and read back.
// One workgroup of WPP waves per particle: the entry- and pair-parallel phases (stage, grid, phase 1, prune) are spread
// over all WPP*64 threads, which is what fills the SIMDs at ~2000 particles; phase 2 is run by wave 0.
#ifndef MERGE_WAVES_PER_EU
#define MERGE_WAVES_PER_EU 4  // 128 VGPRs: with 2 waves per particle all ~2000 particles of C2 are resident at once
#endif
template <int WPP, bool FUSE_PRUNE, int GL = 5>
__device__ __forceinline__ void gm_merge_particle(const Buffers &B, const Params &P, const int cur, const int dst, const int i, const int tid,
                                                  unsigned char *smem_raw, const unsigned short *perm = nullptr) {
  constexpr int NT = WPP * 64;
  constexpr int MERGE_GX = 1 << GL, MERGE_GY = 1 << GL, MERGE_CELLS = MERGE_GX * MERGE_GY;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int cap = B.cap;
  unsigned char *wbase = smem_raw;
  float *sRed = reinterpret_cast<float *>(smem_raw + merge_lds_bytes_per_wave(cap, GL));  // [WPP][8] cross-wave reduction scratch
  auto block_sync = [&]() { if (WPP == 1) wave_sync(); else __syncthreads(); };
  double *sW = reinterpret_cast<double *>(wbase);                          // [cap] weight (exact)
  float *sX = reinterpret_cast<float *>(sW + cap), *sY = sX + cap;          // [cap] position, fp32 (prefilter only)
  float *sRad = sY + cap;                                                   // [cap] prefilter radius, fp32, >= exact; < 0: absorbed
  // grid cursors, 16 bits each, two per word: cell c's entries are sSorted[cell_at(c) .. cell_at(c + 1))
  unsigned *sCellStart = reinterpret_cast<unsigned *>(sRad + cap);          // [(CELLS + 1) halves]
  auto cell_at = [&](int e) -> unsigned { return (sCellStart[e >> 1] >> (16 * (e & 1))) & 0xffffu; };
  unsigned *sRec = sCellStart + MERGE_CELLS / 2 + 4;                        // [cap] row records (see MERGE_REC_*)
  unsigned *sPairs = sRec + cap;                                            // [PAIR_CAP] (a << 16) | (passes << 15) | (reserve << 14) | j
  unsigned *sPairCount = sPairs + MERGE_PAIR_CAP(cap);                      // [1] (+3 pad)
  unsigned short *sSorted = reinterpret_cast<unsigned short *>(sPairCount + 4);  // [cap] grid order; later: candidate rows, prune survivors
  unsigned short *sSlack = sSorted + cap;                                   // [cap] prefilter slack of the row (upper half of an fp32, rounded down)

  const int N = B.count[i];
  double *slab = B.slab[cur];
  double *pW = plane(slab, cap, i, PL_W), *pMX = plane(slab, cap, i, PL_MX), *pMY = plane(slab, cap, i, PL_MY);
  double *pSXX = plane(slab, cap, i, PL_SXX), *pSXY = plane(slab, cap, i, PL_SXY), *pSYY = plane(slab, cap, i, PL_SYY);
  const double t2 = P.mergeT2, f = P.mergeInfl;
  auto phys = [&](int m) -> int { return perm ? (int)perm[m] : m; };       // slab entry of merge entry m

  DBG_TB(32, 0);
#ifdef RFS_PROFILE
  const long long dbgT0 = (long long)__builtin_readcyclecounter();
  long long dbgT2 = 0, dbgT3 = 0;
  int dbgFallbacks = 0, dbgSlackN = 0, dbgUnlistN = 0, dbgSerial = 0, dbgRewalks = 0;
  long long dbgRewalkCycles = 0, dbgWalk1 = 0;
  int dbgRoundsN = 0;
  unsigned dbgPairs = 0;
  if (B.dbg && tid == 0) B.dbg[64 + 4 * (size_t)i + 3] = 0;
  __syncthreads();
#endif
  // ---- stage; hole flags live in per-lane registers: bit s of `hole` <=> entry s*64+lane is a hole ----
  unsigned hole = 0;
  float fxmin = 3.0e38f, fxmax = -3.0e38f, fymin = 3.0e38f, fymax = -3.0e38f, frad = 0.f, fabsmax = 0.f;
  for (int m = tid, sidx = 0; m < N; m += NT, sidx++) {
    const int pm = phys(m);
    const double w = pW[pm], mx = pMX[pm], my = pMY[pm];
    const double bnd = merge_bound(t2, pSXX[pm], pSXY[pm], pSYY[pm]);
    const float fx = (float)mx, fy = (float)my;
    sX[m] = fx; sY[m] = fy; sW[m] = w;
    sRec[m] = MERGE_REC_NOCLAIM;
    if (w < 0) { hole |= 1u << sidx; sRad[m] = -1.f; continue; }  // already absorbed (merge called twice); radius < 0 marks a hole
    const float rad = merge_radius_f32(bnd);
    sRad[m] = rad;
    fxmin = fminf(fxmin, fx); fxmax = fmaxf(fxmax, fx);
    fymin = fminf(fymin, fy); fymax = fmaxf(fymax, fy);
    fabsmax = fmaxf(fabsmax, fmaxf(fabsf(fx), fabsf(fy)));
    frad = fmaxf(frad, rad * 1.0001f);
  }
  for (int c = tid; c <= MERGE_CELLS / 2; c += NT) sCellStart[c] = 0u;
  fxmin = wave_min_f32(fxmin); fxmax = wave_max_f32(fxmax);
  fymin = wave_min_f32(fymin); fymax = wave_max_f32(fymax);
  frad = wave_max_f32(frad);
  fabsmax = wave_max_f32(fabsmax);
  if (WPP > 1 && lane == 0) { float *r = sRed + wave * 8; r[0] = fxmin; r[1] = fxmax; r[2] = fymin; r[3] = fymax; r[4] = frad; r[5] = fabsmax; }
  block_sync();

  DBG_TB(32, 1);
  RFS_CUT(20);
  // ---- phase 1: grid build ----
  if (WPP > 1) {
#pragma unroll
    for (int w2 = 0; w2 < WPP; w2++) {
      const float *r = sRed + w2 * 8;
      fxmin = fminf(fxmin, r[0]); fxmax = fmaxf(fxmax, r[1]); fymin = fminf(fymin, r[2]); fymax = fmaxf(fymax, r[3]); frad = fmaxf(frad, r[4]);
      fabsmax = fmaxf(fabsmax, r[5]);
    }
  }
  // |fp32 coordinate difference - exact difference| <= errAbs: two conversions (2^-24 |x| each) and one subtraction; NaN / inf
  // coordinates make it NaN / inf, and every comparison below then keeps the pair (the exact test decides)
  const float errAbs = 3.0e-7f * fabsmax + 1e-37f;
  // float rounding of the box / radius is covered by the 1e-3 relative slack on the cell edges; indices are clamped
  // (clamping is monotone, so adjacency is preserved for out-of-box values).  Cell edges are >= the largest prefilter
  // radius in both directions, so every pair that can pass lies in adjacent cells.
  const double x0 = (double)fxmin - 1e-3 * fabs((double)fxmin) - 1e-30, y0 = (double)fymin - 1e-3 * fabs((double)fymin) - 1e-30;
  const double spanx = (double)fxmax - x0, spany = (double)fymax - y0;
  // The 64 x 64 grid pays on sparse maps only (configs[2]'s shard: cells of two prefilter radii, fused step 254 -> 243 us).  Where
  // a cell of extent / 64 would be thinner than ~1.5 radii it is clamped to the radius, the rows' slack (cell edge - radius) drops
  // to zero and every row that merges falls back to the sequential scan (2000 x 400 Gaussians within 2.5 m: 319 -> 573 us): such a
  // particle uses the first 32 x 32 cells of the array only (gxe x gye cells in use, row stride MERGE_GX).
  int gxe = MERGE_GX, gye = MERGE_GY;
  if constexpr (GL == 6) {
#ifndef MERGE_FINE_MIN
#define MERGE_FINE_MIN 1.5
#endif
    if (!(spanx / MERGE_GX >= MERGE_FINE_MIN * (double)frad) || !(spany / MERGE_GY >= MERGE_FINE_MIN * (double)frad)) { gxe = MERGE_GX / 2; gye = MERGE_GY / 2; }
  }
  const double cellx = fmax((double)frad, spanx / gxe) * 1.001 + 1e-300, celly = fmax((double)frad, spany / gye) * 1.001 + 1e-300;
  const bool degenerate = !(cellx < 1.0e30) || !(celly < 1.0e30) || !(spanx == spanx) || !(spany == spany) || !(errAbs < 1.0e30f);  // inf / NaN -> a single cell
  const float x0f = (float)x0, y0f = (float)y0;
  const float invCx = degenerate ? 0.f : (float)(1.0 / cellx) * (1.f - 1e-6f), invCy = degenerate ? 0.f : (float)(1.0 / celly) * (1.f - 1e-6f);
  auto cell_of = [&](float x, float y, int &cx, int &cy) {
    int ix = (int)((x - x0f) * invCx), iy = (int)((y - y0f) * invCy);
    cx = ix < 0 ? 0 : (ix >= gxe ? gxe - 1 : ix);
    cy = iy < 0 ? 0 : (iy >= gye ? gye - 1 : iy);
    if (degenerate) { cx = 0; cy = 0; }
  };
  for (int m = tid, sidx = 0; m < N; m += NT, sidx++) {
    if ((hole >> sidx) & 1u) continue;
    int cx, cy;
    cell_of(sX[m], sY[m], cx, cy);
    const int e = cy * MERGE_GX + cx + 1;  // counts, shifted by one entry
    atomicAdd(&sCellStart[e >> 1], 1u << (16 * (e & 1)));  // (a half never overflows: counts <= cap < 65536)
  }
  block_sync();
  if (wave == 0) {  // entry[c + 1] := entries in cells < c (the cursor of cell c); the scatter below advances it to end(c) = start(c + 1)
    // exclusive scan over the CELLS + 1 half-word entries: lane l owns entries 16l .. 16l + 15 (8 words); entry CELLS (the
    // low half of the last word) receives the total
    constexpr int WPL = MERGE_CELLS / 128;  // words per lane
    int off;
    if constexpr (WPL <= 8) {
      unsigned wv[WPL];
      int tot = 0;
#pragma unroll
      for (int k = 0; k < WPL; k++) { wv[k] = sCellStart[WPL * lane + k]; tot += (int)(wv[k] & 0xffffu) + (int)(wv[k] >> 16); }
      off = wave_excl_scan(tot, lane);
      wave_sync();
#pragma unroll
      for (int k = 0; k < WPL; k++) {
        const unsigned lo = (unsigned)off;
        off += (int)(wv[k] & 0xffffu);
        const unsigned hi = (unsigned)off;
        off += (int)(wv[k] >> 16);
        sCellStart[WPL * lane + k] = lo | (hi << 16);
      }
    } else {  // 64 x 64: 32 words per lane, read twice rather than held in registers
      int tot = 0;
      for (int k = 0; k < WPL; k += 4) {
        const uint4 v = *reinterpret_cast<const uint4 *>(&sCellStart[WPL * lane + k]);
        tot += (int)(v.x & 0xffffu) + (int)(v.x >> 16) + (int)(v.y & 0xffffu) + (int)(v.y >> 16) + (int)(v.z & 0xffffu) + (int)(v.z >> 16) + (int)(v.w & 0xffffu) + (int)(v.w >> 16);
      }
      off = wave_excl_scan(tot, lane);
      for (int k = 0; k < WPL; k++) {   // (every lane rewrites only its own words)
        const unsigned v = sCellStart[WPL * lane + k];
        const unsigned lo = (unsigned)off;
        off += (int)(v & 0xffffu);
        const unsigned hi = (unsigned)off;
        off += (int)(v >> 16);
        sCellStart[WPL * lane + k] = lo | (hi << 16);
      }
    }
    if (lane == 63) sCellStart[MERGE_CELLS / 2] = (unsigned)off;  // entry CELLS = the cursor of the last cell... (see below)
  }
  block_sync();
  for (int m = tid, sidx = 0; m < N; m += NT, sidx++) {
    if ((hole >> sidx) & 1u) continue;
    int cx, cy;
    cell_of(sX[m], sY[m], cx, cy);
    const int e = cy * MERGE_GX + cx + 1;
    const unsigned pos = (atomicAdd(&sCellStart[e >> 1], 1u << (16 * (e & 1))) >> (16 * (e & 1))) & 0xffffu;
    sSorted[pos] = (unsigned short)m;
#if MERGE_SCAN_SHIFTED_RADIUS
    // Round 6: from here on sRad holds the prefilter radius WITH the fp32 rounding bound of a coordinate difference added
    // (r + 1.5 errAbs): the candidate scan's per-neighbour threshold is then max(r_a', r_j')^2 (1 + 1e-5) -- the value it used to
    // form from the two radii per neighbour (the shift is monotone, so the maximum commutes with it: the same fp32 number), three
    // vector instructions instead of six.  Every later reader wants an upper bound of the radius (the sequential scan's bound) or
    // takes the shift back out (the walk's rPass); the sign stays the liveness flag.
    sRad[m] = sRad[m] + 1.5f * errAbs;
#endif
  }
  if (tid == 0) *sPairCount = 0u;
  block_sync();
  DBG_TB(32, 8);
  RFS_CUT(21);
  // ---- phase 1: every pair (a, j > a) against the INITIAL states ----
  // 1a: each entry scans the 3x3 cells around it, four neighbours per trip, with the branch-free fp32 distance prefilter
  //     |x_j - x_a| <= max(r_a, r_j) + rounding bound.  Survivors with a higher index are collected in registers and written
  //     as ONE contiguous segment of the pair list (one atomicAdd per row).  The scan also keeps the row's SLACK: the
  //     smallest (distance - prefilter radius) among the neighbours that fail, bounded by what separates it from entries
  //     outside the 3x3 cells.  While the row later moves / grows by less than its slack, no other entry can start passing,
  //     so the segment stays the complete list of possible partners.
  // 1b: the list is processed with all lanes busy (one pair per lane): means and covariances of both entries are fetched
  //     together from the slab, the exact Mahalanobis test runs, and the verdict is kept in the pair's own word (bit 15)
  //     and in the row record.
  const int pairCap = MERGE_PAIR_CAP(cap);
  const float slackOut = (float)(fmin(cellx, celly)) * (1.f - 4e-6f) - frad * (1.f + 4e-6f) - 3.f * errAbs;
#ifndef MERGE_SCAN_GRID_ORDER
#define MERGE_SCAN_GRID_ORDER 1   // r04, profiles/r04d_ab_scan_grid_order.txt: fused step 124.0 -> 121.2-121.9 us at configs[1]
#endif
#ifndef MERGE_SCAN_PACK_ALIGNBIT
#define MERGE_SCAN_PACK_ALIGNBIT 1   // r06: a row's survivors in a four-register shift chain (v_alignbit) instead of variable 64-bit shifts + selects
#endif
  const int nLive = MERGE_SCAN_GRID_ORDER ? (int)cell_at(MERGE_CELLS) : N;     // (entries in the grid = the live ones)
  for (int t0 = tid, sidx = 0; t0 < nLive; t0 += NT, sidx++) {
    // MERGE_SCAN_GRID_ORDER: the threads take the entries in GRID order, so that the lanes of a wave work on neighbouring cells --
    // similar neighbour counts (the trips of four last as long as the busiest lane's), the same LDS words
    const int m = MERGE_SCAN_GRID_ORDER ? (int)sSorted[t0] : t0;
    if (!MERGE_SCAN_GRID_ORDER && ((hole >> sidx) & 1u)) continue;
    const float ax = sX[m], ay = sY[m], ar = sRad[m];   // (MERGE_SCAN_SHIFTED_RADIUS: ar, jr carry + 1.5 errAbs already)
    int cx, cy;
    cell_of(ax, ay, cx, cy);
    const int cxa = cx > 0 ? cx - 1 : 0, cxb = cx < gxe - 1 ? cx + 1 : gxe - 1;
    [[maybe_unused]] unsigned long long buf0 = 0ull, buf1 = 0ull;  // up to 8 survivors, 16 bits each
    [[maybe_unused]] unsigned sb0 = 0u, sb1 = 0u, sb2 = 0u, sb3 = 0u;   // MERGE_SCAN_PACK_ALIGNBIT: the same eight fields as a shift chain, newest in sb0's low half
    int nP = 0;
    float farE2 = 3.0e38f;   // nearest neighbour that fails the prefilter by a factor >= 2 in distance
    // the three cell rows are three contiguous ranges of the sorted list; they are walked as ONE sequence (same order as
    // row by row), so that only the last trip of four is partly empty instead of the last trip of every row
    const unsigned qs1 = cell_at(cy * MERGE_GX + cxa), n1 = cell_at(cy * MERGE_GX + cxb + 1) - qs1;
    unsigned qs0 = 0, n0 = 0, qs2 = 0, n2 = 0;
    if (cy > 0) { qs0 = cell_at((cy - 1) * MERGE_GX + cxa); n0 = cell_at((cy - 1) * MERGE_GX + cxb + 1) - qs0; }
    if (cy < gye - 1) { qs2 = cell_at((cy + 1) * MERGE_GX + cxa); n2 = cell_at((cy + 1) * MERGE_GX + cxb + 1) - qs2; }
    const unsigned n01 = n0 + n1, tot = n01 + n2;       // (tot >= 1: the entry itself)
#ifdef RFS_PROFILE
    if (B.dbg && i == 7) {   // (this run's numbers: the host clears the words before the launch it reports)
      atomicAdd((unsigned long long *)&B.dbg[58], (unsigned long long)tot);
      { const int k6 = (wave * 3 + (sidx < 2 ? sidx : 2)) % 6; atomicMax((unsigned long long *)&B.dbg[k6 == 0 ? 57 : (k6 == 1 ? 59 : 58 + k6)], (unsigned long long)((tot + 3u) >> 2)); }
    }
#endif
    {
      for (unsigned q = 0; q < tot; q += 4) {
        unsigned jj[4];
        float jx[4], jy[4], jr[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const unsigned t = (q + k < tot) ? q + k : tot - 1;
          const unsigned pos = (t < n0) ? qs0 + t : ((t < n01) ? qs1 + (t - n0) : qs2 + (t - n01));
          jj[k] = sSorted[pos];
        }
#pragma unroll
        for (int k = 0; k < 4; k++) { jx[k] = sX[jj[k]]; jy[k] = sY[jj[k]]; jr[k] = sRad[jj[k]]; }
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const float e0 = jx[k] - ax, e1 = jy[k] - ay;
          const float e2 = e0 * e0 + e1 * e1;
#if MERGE_SCAN_SHIFTED_RADIUS
          float thr;                                                 // max(ar, jr) of two radii that are numbers (or +inf): the raw instruction, no canonicalising pre-pass
          asm("v_max_f32 %0, %1, %2" : "=v"(thr) : "v"(ar), "v"(jr[k]));
#else
          const float thr = fmaxf(ar, jr[k]) + 1.5f * errAbs;       // exact |e| <= exact radius  =>  fp32 |e| <= thr (+ ulps)
#endif
          const float T = thr * thr * (1.f + 1e-5f);
          // higher index only (the entry plays `a`); NaN distances fall through to the exact test like the reference
          const bool cand = (q + k < tot) & (jj[k] > (unsigned)m);
          const bool c = cand & !(e2 > T);
          // neighbours within twice the prefilter radius are listed too, as RESERVE partners (bit 14): they cannot pass
          // now, but may once the row has merged and moved; everything farther bounds the row's slack from below
          const bool reserve = cand & !c & (e2 < 4.0f * T);
          const bool farFail = cand & !c & !reserve;
          farE2 = farFail ? fminf(farE2, e2) : farE2;  // exact distance - exact radius >= sqrt(e2) / 2 for these
          if (c | reserve) {
#if MERGE_SCAN_PACK_ALIGNBIT
            // (the order of a row's segment is free: every reader takes the lowest index first; fields pushed out of the top belong
            //  to a row with more than eight survivors, which is not listed at all)
            sb3 = __builtin_amdgcn_alignbit(sb3, sb2, 16);
            sb2 = __builtin_amdgcn_alignbit(sb2, sb1, 16);
            sb1 = __builtin_amdgcn_alignbit(sb1, sb0, 16);
            sb0 = (sb0 << 16) | (jj[k] | (reserve ? 0x4000u : 0u));
#else
            const unsigned long long v = (unsigned long long)(jj[k] | (reserve ? 0x4000u : 0u)) << (16 * (nP & 3));
            if (nP < 4) buf0 |= v; else if (nP < 8) buf1 |= v;
#endif
            nP++;
          }
        }
      }
    }
    unsigned rec = MERGE_REC_NOCLAIM;
    if (nP > 0) {
      unsigned base = 0;
      bool listed = nP <= MERGE_ROW_SLOTS;
      if (listed) {
        base = atomicAdd(sPairCount, (unsigned)nP);
        listed = base + (unsigned)nP <= (unsigned)pairCap;
      }
      if (listed) {
#if MERGE_SCAN_PACK_ALIGNBIT
        const unsigned hi = (unsigned)m << 16;
        const unsigned fl[MERGE_ROW_SLOTS] = {sb0 & 0xffffu, sb0 >> 16, sb1 & 0xffffu, sb1 >> 16, sb2 & 0xffffu, sb2 >> 16, sb3 & 0xffffu, sb3 >> 16};
#pragma unroll
        for (int k = 0; k < MERGE_ROW_SLOTS; k++)
          if (k < nP) sPairs[base + k] = hi | fl[k];                                              // index | reserve flag
#else
        for (int k = 0; k < nP; k++) {
          const unsigned j = (unsigned)(((k < 4 ? buf0 : buf1) >> (16 * (k & 3))) & 0xffffull);  // index | reserve flag
          sPairs[base + k] = ((unsigned)m << 16) | j;
        }
#endif
        rec |= ((unsigned)nP << 20) | base;
      } else {
        rec |= MERGE_REC_ISROW | (15u << 20);  // too crowded to list: the sequential scan handles this row
        if (nP <= MERGE_ROW_SLOTS)
          for (int k = 0; k < nP; k++)
            if (base + k < (unsigned)pairCap) sPairs[base + k] = 0xffffffffu;  // reserved but unused words: skipped by 1b
      }
    }
    sRec[m] = rec;
    float slackMin = __builtin_amdgcn_sqrtf(farE2) * (0.5f - 4e-6f);
    slackMin = degenerate ? 0.f : fminf(slackMin, slackOut);
    sSlack[m] = (slackMin > 0.f) ? (unsigned short)(__float_as_uint(slackMin) >> 16) : (unsigned short)0;
  }
  block_sync();
  RFS_CUT(22);
  {
    const int nPairs = (int)min(*sPairCount, (unsigned)pairCap);
    for (int p0 = 0; p0 < nPairs; p0 += NT) {
      const int pi = p0 + tid;
      if (pi < nPairs) {
        const unsigned pr2 = sPairs[pi];
        const int a = (int)(pr2 >> 16), j = (int)(pr2 & 0x3fffu);
        if (pr2 != 0xffffffffu && !(pr2 & 0x4000u)) {  // (unused words and reserve partners need no test)
          const int pa = phys(a), pj = phys(j);
          // both entries up front: ten independent loads in flight
          const double axx = pSXX[pa], axy = pSXY[pa], ayy = pSYY[pa];
          const double jxx = pSXX[pj], jxy = pSXY[pj], jyy = pSYY[pj];
          const double e0 = pMX[pj] - pMX[pa], e1 = pMY[pj] - pMY[pa];
          double a00, a01, a10, a11, det;
          inv2(axx, axy, axy, ayy, a00, a01, a10, a11, det);
          const double u0 = e0 * a00 + e1 * a01, u1 = e0 * a01 + e1 * a11;
          bool far = (u0 * e0 + u1 * e1) > t2;
          if (far) {
            double j00, j01, j10, j11, jdet;
            inv2(jxx, jxy, jxy, jyy, j00, j01, j10, j11, jdet);
            const double g0 = -e0, g1 = -e1;
            const double v0 = g0 * j00 + g1 * j01, v1 = g0 * j01 + g1 * j11;
            far = (v0 * g0 + v1 * g1) > t2;
          }
          if (!far && ((sW[a] + sW[j]) != 0.0)) {
            sPairs[pi] = pr2 | 0x8000u;
            atomicOr(&sRec[a], MERGE_REC_ISROW);
          }
        }
      }
    }
  }
  block_sync();

  DBG_TB(32, 2);
  RFS_CUT(23);
#ifndef MERGE_P2_PRIO
#define MERGE_P2_PRIO 0
#endif
  __builtin_amdgcn_s_setprio(MERGE_P2_PRIO);   // (step_fused.h: the last level of the fused step's falling issue priority; a no-op elsewhere)
#ifdef RFS_PROFILE
  dbgT2 = (long long)__builtin_readcyclecounter();
  dbgPairs = *sPairCount;
#endif
  // ---- phase 2: replay the rows that have a candidate, with the exact greedy rule ----
  // Liveness lives in LDS from here on: sRad[j] < 0  <=>  j has been absorbed.
  //
  // Speculative lane-parallel replay + ordered validation.  Up to 64 candidate rows at a time, one per lane, are
  // replayed independently against the states every entry had when the round started: the lane walks its row's listed
  // partners in ascending index order (each met once, with the row's state at that moment), merging as the reference
  // would; before the row's first merge the verdicts of phase 1b are reused.  A row only reads entries with a higher
  // index, and an entry's mean/covariance change only while it is the outer row, so the sole cross-row hazard is an
  // entry absorbed by an EARLIER row of the same round.  Rows are therefore validated in ascending order: a row that
  // was itself absorbed is dropped; a row that absorbed an entry already taken walks again with the committed holes
  // visible (round 6: all such rows at once, in sub-rounds -- see MERGE_VALIDATE_ROUNDS below; rounds 2-5: one by one);
  // everything else commits as computed.  A row that outgrows its slack (unlisted entries might start passing) or its
  // list goes to the sequential scan by the whole wave (the reference's own scan).
  bool anyMerge = false;
#ifdef RFS_PROFILE
  int dbgRows = 0, dbgMerges = 0, dbgChunks = 0;
#endif
  if (wave == 0) {  // ======== phase 2 is wave 0's; the other waves wait at the barrier below ========
  unsigned short *sRows = sSorted;                                      // [<= cap] candidate rows, ascending (the grid is done)
  unsigned short *sSpec = reinterpret_cast<unsigned short *>(sCellStart);  // [64][8] absorbed entries per lane
  int nRowsTotal = 0;
  for (int c0 = 0; c0 < N; c0 += 64) {
    const int m = c0 + lane;
    const bool isRow = (m < N) && (sRec[m] & MERGE_REC_ISROW);
    const unsigned long long rm = __ballot(isRow);
    wave_sync();  // (sRows aliases sSorted: nothing reads the grid order any more)
    if (isRow) sRows[nRowsTotal + __popcll(rm & ((1ull << lane) - 1ull))] = (unsigned short)m;
    nRowsTotal += __popcll(rm);
  }
  wave_sync();
  DBG_TB(32, 9);

  // the reference's sequential scan for ONE row, executed by the whole wave (fallback + exactness anchor)
  auto seq_replay = [&](const int a) {
#ifdef RFS_PROFILE
    dbgFallbacks++;
#endif
    // nothing below the row's lowest listed partner can pass before the first merge, and that merge is at or above it
    int j0 = a + 1;
    {
      const unsigned rec = sRec[a];
      const int n = (int)((rec >> 20) & 15u);
      if (n != 15 && n > 0) {
        const unsigned base = rec & 0xfffffu;
        unsigned lo = 0x3fffu;
        for (int k = 0; k < n; k++) lo = min(lo, sPairs[base + k] & 0x3fffu);
        j0 = (int)lo;
      }
    }
    const int pa = phys(a);
    double ax = pMX[pa], ay = pMY[pa], aw = sW[a];
    double axx = pSXX[pa], axy = pSXY[pa], ayy = pSYY[pa];
    double ab = merge_bound(t2, axx, axy, ayy);
    double a00, a01, a10, a11, adet;
    inv2(axx, axy, axy, ayy, a00, a01, a10, a11, adet);
    bool changed = false;
    int floorLane = j0 & 63;
    for (int c0 = j0 & ~63; c0 < N; c0 += 64) {
      const int j = c0 + lane;
#ifdef RFS_PROFILE
      dbgChunks++;
#endif
      double jx = 0, jy = 0, jw = 0, jb = -1.0;
      int pj = 0;
      if (j > a && j < N) {
        pj = phys(j);
        jx = pMX[pj]; jy = pMY[pj]; jw = sW[j];
        const float r = sRad[j];
        jb = (r < 0.f) ? -1.0 : (double)r * (double)r * (1.0 + 1e-5);   // >= the entry's exact bound (a shifted radius is only larger)
      }
      bool live = !(jb < 0.0);
      while (true) {
        bool pass = false;
        if (live && lane >= floorLane) {
          const double e0 = jx - ax, e1 = jy - ay;
          if (!((e0 * e0 + e1 * e1) > fmax(ab, jb))) pass = merge_pair_passes(e0, e1, a00, a01, a11, aw, jw, pSXX, pSXY, pSYY, pj, t2);
        }
        const unsigned long long pm = __ballot(pass);
        if (pm == 0ull) break;
        const int l = __builtin_ctzll(pm);
        const int jj = c0 + l;
        const int pjj = phys(jj);
        // merge jj into a (GaussianMixture.hpp:444-471), wave-uniform arithmetic
        const double w1 = aw, w2 = sW[jj];
        const double x2 = pMX[pjj], y2 = pMY[pjj], bxx = pSXX[pjj], bxy = pSXY[pjj], byy = pSYY[pjj];
        const double wm = w1 + w2;
        const double xm = (ax * w1 + x2 * w2) / wm, ym = (ay * w1 + y2 * w2) / wm;
        const double d10 = xm - ax, d11 = ym - ay, d20 = xm - x2, d21 = ym - y2;
        const double nxx = (w1 * (axx + (f * d10) * d10) + w2 * (bxx + (f * d20) * d20)) / wm;
        const double nxy = (w1 * (axy + (f * d10) * d11) + w2 * (bxy + (f * d20) * d21)) / wm;
        const double nyy = (w1 * (ayy + (f * d11) * d11) + w2 * (byy + (f * d21) * d21)) / wm;
        ax = xm; ay = ym; axx = nxx; axy = nxy; ayy = nyy; aw = wm;
        inv2(axx, axy, axy, ayy, a00, a01, a10, a11, adet);
        ab = merge_bound(t2, axx, axy, ayy);
        changed = true;
#ifdef RFS_PROFILE
        dbgMerges++;
#endif
        if (lane == l) { sRad[jj] = -1.f; live = false; }
        floorLane = l + 1;
        if (floorLane >= 64) break;
      }
      floorLane = 0;
    }
    if (changed) {
      anyMerge = true;
      sW[a] = aw;  // uniform store; the other LDS fields of a are never read again (a is behind the scan)
      if (lane == 0) { pW[pa] = aw; pMX[pa] = ax; pMY[pa] = ay; pSXX[pa] = axx; pSXY[pa] = axy; pSYY[pa] = ayy; }
    }
    wave_sync();
  };

  // (Measured and dropped, round 6 -- profiles/r06b_ab_component_replay.txt, r06b_component_replay.patch: the replay by CONNECTED
  //  COMPONENTS of the listed-pair graph, one component per lane walking its rows in the reference's order, no claims / validation /
  //  re-walks.  Exact -- the full GPU suite and 1500 fuzz cases pass with it -- and 19 % slower: fused step 137.3 against 115.0 us at
  //  configs[1].  The components are small (12-25 per particle, the largest 4-7 rows, tools/merge_components_study.py) but the label
  //  propagation takes ~4 trips over the pair list (7.8 k cycles) and the busiest lane walks 3 rows one after the other (28 k) where the
  //  speculative pass below walks all rows at once (15 k) and validates in 11 k.)
  for (int r0 = 0; r0 < nRowsTotal; r0 += 64) {
    const int cnt = (nRowsTotal - r0 < 64) ? nRowsTotal - r0 : 64;
    // ---- speculative replay, one row per lane ----
    const int a = (lane < cnt) ? (int)sRows[r0 + lane] : 0;
    const int pa = phys(a);
    const bool active = (lane < cnt) && !(sRad[a] < 0.f);
    int nAbs = 0;
    bool ovf = false;
    [[maybe_unused]] int dbgWhy = 0;  // (profile builds report why rows fell back to the sequential scan)
    double ax = 0, ay = 0, aw = 0, axx = 1, axy = 0, ayy = 1;
    // the walk of ONE row over its listed partners, by the calling lane, against the liveness it sees now
    auto walk = [&]() {
      nAbs = 0;
      ovf = false;
      const unsigned rec = sRec[a];
      const int n = (int)((rec >> 20) & 15u);
      if (n == 15) {
        ovf = true;  // not listable
        dbgWhy |= 1;
      } else {
        const unsigned base = rec & 0xfffffu;
        // the row's partners: (passes << 15) | j; absorbed ones are dropped here (liveness is fixed during the round)
        unsigned it[MERGE_ROW_SLOTS];
#pragma unroll
        for (int k = 0; k < MERGE_ROW_SLOTS; k++) it[k] = (k < n) ? (sPairs[base + k] & 0xffffu) : 0xffffffffu;
#pragma unroll
        for (int k = 0; k < MERGE_ROW_SLOTS; k++)
          if (k < n && sRad[it[k] & 0x3fffu] < 0.f) it[k] = 0xffffffffu;
        ax = pMX[pa]; ay = pMY[pa]; aw = sW[a];
        axx = pSXX[pa]; axy = pSXY[pa]; ayy = pSYY[pa];
        const float slack = __uint_as_float((unsigned)sSlack[a] << 16);
        const float rPass = MERGE_SCAN_SHIFTED_RADIUS ? ((sRad[a] - 1.5f * errAbs) - 1.2e-7f * sRad[a]) * (1.f - 1e-5f)   // (the shift taken back out, less the two roundings it cost)
                                                       : sRad[a] * (1.f - 1e-5f);     // <= the row's exact initial radius
        float shift = 0.f;
        bool changed = false;
        double a00 = 0, a01 = 0, a11 = 0;
#if MERGE_WALK_PREFETCH
        // The partner this step works on was picked -- and its operands requested -- during the step before: while the row is in its
        // initial state the lowest listed partner that PASSED in phase 1b (the others change nothing), from its first merge on the
        // lowest listed one above the last.  A step that is worked on at all leaves the row changed or finds it so, hence the partner
        // after it is always "the lowest listed above it": its loads (L2 hits, ~700 cycles with the SIMDs as empty as this phase leaves
        // them) are in flight during the five divisions of the merge instead of after them.
        auto pick = [&](const unsigned above, const bool passOnly) -> unsigned {
          unsigned best = 0xffffffffu;
#pragma unroll
          for (int k = 0; k < MERGE_ROW_SLOTS; k++) {
            const unsigned j = it[k] & 0x3fffu;
            const bool c = (it[k] != 0xffffffffu) & (j > above) & (!passOnly | ((it[k] & 0x8000u) != 0u)) & (j < (best & 0x3fffu) || best == 0xffffffffu);
            best = c ? it[k] : best;
          }
          return best;
        };
        unsigned nb = pick((unsigned)a, true);
        double njw = 0, njx = 0, njy = 0, njxx = 1, njxy = 0, njyy = 1;
        auto fetch = [&]() {
          if (nb != 0xffffffffu) {
            const unsigned j = nb & 0x3fffu;
            const int pj = phys((int)j);
            njw = sW[j]; njx = pMX[pj]; njy = pMY[pj];
            njxx = pSXX[pj]; njxy = pSXY[pj]; njyy = pSYY[pj];
          }
        };
        fetch();
        for (int step = 0; step < MERGE_ROW_SLOTS; step++) {
          if (nb == 0xffffffffu) break;
          const unsigned j = nb & 0x3fffu;
          const double jw = njw, jx = njx, jy = njy, jxx = njxx, jxy = njxy, jyy = njyy;
          nb = pick(j, false);
          fetch();
          bool pass = true;                    // (initial state: picked because it passed)
#else
        unsigned cur = (unsigned)a;
        for (int step = 0; step < MERGE_ROW_SLOTS; step++) {
          // lowest listed partner above `cur`
          unsigned best = 0xffffffffu;
#pragma unroll
          for (int k = 0; k < MERGE_ROW_SLOTS; k++) {
            const unsigned j = it[k] & 0x3fffu;
            const bool c = (it[k] != 0xffffffffu) & (j > cur) & (j < (best & 0x3fffu) || best == 0xffffffffu);
            best = c ? it[k] : best;
          }
          if (best == 0xffffffffu) break;
          const unsigned j = best & 0x3fffu;
          cur = j;
          bool pass = (best & 0x8000u) != 0u;  // verdict of phase 1b: valid while the row is in its initial state
          if (!changed && !pass) continue;
          const int pj = phys((int)j);
          const double jw = sW[j], jx = pMX[pj], jy = pMY[pj];
          const double jxx = pSXX[pj], jxy = pSXY[pj], jyy = pSYY[pj];
#endif
          if (changed) {
            const double e0 = jx - ax, e1 = jy - ay;
            const double u0 = e0 * a00 + e1 * a01, u1 = e0 * a01 + e1 * a11;
            bool far = (u0 * e0 + u1 * e1) > t2;
            if (far) {
              double j00, j01, j10, j11, jdet;
              inv2(jxx, jxy, jxy, jyy, j00, j01, j10, j11, jdet);
              const double g0 = -e0, g1 = -e1;
              const double v0 = g0 * j00 + g1 * j01, v1 = g0 * j01 + g1 * j11;
              far = (v0 * g0 + v1 * g1) > t2;
            }
            pass = !far && ((aw + jw) != 0.0);
            if (!pass) continue;
          }
          if (nAbs >= 8) { ovf = true; dbgWhy |= 4; break; }
          sSpec[lane * 8 + nAbs] = (unsigned short)j;
          nAbs++;
          const double w1 = aw, w2 = jw;
          const double wm = w1 + w2;
          const double xm = (ax * w1 + jx * w2) / wm, ym = (ay * w1 + jy * w2) / wm;
          const double d10 = xm - ax, d11 = ym - ay, d20 = xm - jx, d21 = ym - jy;
          const double nxx = (w1 * (axx + (f * d10) * d10) + w2 * (jxx + (f * d20) * d20)) / wm;
          const double nxy = (w1 * (axy + (f * d10) * d11) + w2 * (jxy + (f * d20) * d21)) / wm;
          const double nyy = (w1 * (ayy + (f * d11) * d11) + w2 * (jyy + (f * d21) * d21)) / wm;
          ax = xm; ay = ym; axx = nxx; axy = nxy; ayy = nyy; aw = wm;
          changed = true;
          // Could an entry outside the list pass now?  Not while the row has moved / grown by less than its slack.
          const double ab = merge_bound(t2, axx, axy, ayy);
          shift += __builtin_amdgcn_sqrtf((float)(d10 * d10 + d11 * d11)) * (1.f + 4e-6f) + 1e-30f;
          const float grow = __builtin_amdgcn_sqrtf((float)ab) * (1.f + 4e-6f) - rPass;
          if (!(slack > shift + fmaxf(grow, 0.f))) { ovf = true; dbgWhy |= 2; break; }
          double a10, adet;
          inv2(axx, axy, axy, ayy, a00, a01, a10, a11, adet);
        }
      }
    };
#ifdef RFS_PROFILE
    const long long wk0 = (long long)__builtin_readcyclecounter();
#endif
    if (active) walk();
    wave_sync();
#ifdef RFS_PROFILE
    dbgWalk1 += (long long)__builtin_readcyclecounter() - wk0;
    dbgRoundsN++;
#endif
#ifndef MERGE_P2_BOOST
#define MERGE_P2_BOOST 22       // rows left to validate one by one from which the workgroup's issue priority goes up (0: never)
#define MERGE_P2_BOOST_PRIO 2
#endif
#if MERGE_VALIDATE_ROUNDS
    // Validation in sub-rounds (round 6).  `pending`: the rows of this round that are not final yet.  Per sub-round: the pending lanes
    // claim what they absorbed (the lane in the top bits of the entry's record, lowest lane wins); `conflict`: the lane lost a claim
    // (or could not finish its row).  A row is absorbed exactly when the (lowest) lane claiming it is itself alive: alive(l) =
    // !claimed(a_l) || !alive(claimer(a_l)), claimer < l -- resolved by iterating to the fixed point (chain depth, usually 1-2 trips).
    // Before the first lane that is alive AND in conflict this is exact: an alive lane without conflict holds the lowest claim on
    // everything it absorbed, and any other claimer of those entries would be alive-and-in-conflict itself or dead.  Those rows commit
    // together.  Then, instead of validating the rest one by one (rounds 2-5: p50 11 rows and 3 one-lane re-walks of ~2.5 k cycles
    // each per particle, the slowest particles 26 and 11 -- and a launch ends with those), the rest goes through the same procedure
    // again: rows absorbed by now drop out, EVERY row that absorbed an entry that has gone since walks again -- all of them at once,
    // against the present liveness --, the claims are cleared and made again by the pending lanes only.  A walk depends on the liveness
    // only through the entries it absorbed (an entry it met and left is as good as a hole), so a row none of whose entries has gone has
    // the walk it would have now; the lowest pending lane, re-walked with every earlier row final, wins all its claims: each sub-round
    // settles at least that row, and as many more as have no real overlap (chain depth instead of row count).  A row that cannot be
    // finished from its list is replayed by the whole wave (the reference's scan) when it is the lowest pending one.
    bool pendLane = active;
    unsigned long long commit = 0ull;
    const unsigned long long actm = __ballot(active);
    while (true) {
      if (pendLane)
        for (int k = 0; k < nAbs; k++) {
          const unsigned e = sSpec[lane * 8 + k];
          atomicMin(&sRec[e], ((unsigned)lane << 25) | (sRec[e] & 0x01ffffffu));
        }
      wave_sync();
      DBG_TB(32, 10);
      bool conflict = ovf;
      bool claimedRow = false;
      unsigned claimer = 0;
      if (pendLane) {
        claimer = sRec[a] >> 25;
        claimedRow = claimer != 0x7fu;
        claimer &= 63u;
        for (int k = 0; k < nAbs; k++) conflict |= (sRec[sSpec[lane * 8 + k]] >> 25) != (unsigned)lane;
      }
      const unsigned long long pendm = __ballot(pendLane);
      unsigned long long alivem = __ballot(pendLane & !claimedRow);
      for (int it = 0; it < 64; it++) {
        const unsigned long long nm = __ballot(pendLane & (!claimedRow | !((alivem >> claimer) & 1ull)));
        if (nm == alivem) break;
        alivem = nm;
      }
      const unsigned long long cm = __ballot(conflict & pendLane) & alivem;
#ifdef RFS_PROFILE
      dbgSlackN += __popcll(__ballot(pendLane && (dbgWhy & 2) != 0));
      dbgUnlistN += __popcll(__ballot(pendLane && (dbgWhy & 1) != 0));
      if (B.dbg && i == 7 && pendm == actm) {
        const int w1 = __popcll(__ballot(dbgWhy & 1)), w2 = __popcll(__ballot(dbgWhy & 2)), w4 = __popcll(__ballot(dbgWhy & 4)), wc = __popcll(__ballot(conflict & active & !ovf));
        if (lane == 0) { B.dbg[52] = w1; B.dbg[53] = w2; B.dbg[54] = w4; B.dbg[55] = wc; B.dbg[56] = __popcll(actm); }
      }
      dbgSerial++;
#endif
      const int firstDirty = cm ? __builtin_ctzll(cm) : 64;
      // (a workgroup that finds itself with a long tail takes the map update's issue priority back, see MERGE_P2_BOOST)
      if (MERGE_P2_BOOST > 0 && firstDirty < 64 && __popcll(pendm >> firstDirty) >= MERGE_P2_BOOST) __builtin_amdgcn_s_setprio(MERGE_P2_BOOST_PRIO);
      const bool commitNow = pendLane && lane < firstDirty && ((alivem >> lane) & 1ull) && nAbs > 0;
      if (commitNow) {
        for (int k = 0; k < nAbs; k++) sRad[sSpec[lane * 8 + k]] = -1.f;
      }
      commit |= __ballot(commitNow);
#ifdef RFS_PROFILE
      dbgRows += __popcll(alivem & ((firstDirty < 64) ? ((1ull << firstDirty) - 1ull) : ~0ull));
#endif
      if (firstDirty >= 64) break;                 // every pending row was clean: the round is done
      wave_sync();
      // lane firstDirty is alive and every row before it is final
      pendLane = pendLane && lane >= firstDirty;
      if ((__ballot(ovf) >> firstDirty) & 1ull) {
        seq_replay(__builtin_amdgcn_readlane(a, firstDirty));   // the row could not be finished from its list: the reference's scan, exactly
        if (lane == firstDirty) pendLane = false;
      }
      if (pendLane && sRad[a] < 0.f) pendLane = false;   // absorbed by a row that is final: the row no longer exists
      bool stale = false;
      if (pendLane)
        for (int k = 0; k < nAbs; k++) stale |= sRad[sSpec[lane * 8 + k]] < 0.f;
#ifdef RFS_PROFILE
      const long long rw0 = (long long)__builtin_readcyclecounter();
      dbgRewalks += __popcll(__ballot(stale));
#endif
      if (stale) walk();                            // every earlier row's holes are visible; rows that overlap a PENDING row come round again
#ifdef RFS_PROFILE
      dbgRewalkCycles += (long long)__builtin_readcyclecounter() - rw0;
#endif
      if (__ballot(pendLane) == 0ull) break;
      for (int m = lane; m < N; m += 64) sRec[m] |= MERGE_REC_NOCLAIM;   // claims are per sub-round
      wave_sync();
    }
#else
    // claims: the lane in the top bits of the record of every entry a lane absorbed (lowest lane wins)
    for (int k = 0; k < nAbs; k++) {
      const unsigned e = sSpec[lane * 8 + k];
      atomicMin(&sRec[e], ((unsigned)lane << 25) | (sRec[e] & 0x01ffffffu));
    }
    wave_sync();
    DBG_TB(32, 10);
    // Validation.  `conflict`: the lane lost the claim on something it absorbed (or could not finish its row).
    // A row is absorbed exactly when the (lowest) lane claiming it is itself alive: alive(l) = !claimed(a_l) ||
    // !alive(claimer(a_l)), claimer < l -- resolved by iterating to the fixed point (chain depth, usually 1-2 trips).
    // Before the first lane that is alive AND in conflict this is exact: an alive lane without conflict holds the
    // lowest claim on everything it absorbed, and any other claimer of those entries would be alive-and-in-conflict
    // itself or dead.  Those rows commit together.  From that lane on, rows are validated one by one in ascending
    // order, and replayed by the whole wave when an absorbed entry is gone.
    bool conflict = ovf;
    bool claimedRow = false;
    unsigned claimer = 0;
    if (active) {
      claimer = sRec[a] >> 25;
      claimedRow = claimer != 0x7fu;
      claimer &= 63u;
      for (int k = 0; k < nAbs; k++) conflict |= (sRec[sSpec[lane * 8 + k]] >> 25) != (unsigned)lane;
    }
    const unsigned long long actm = __ballot(active);
    unsigned long long alivem = __ballot(active & !claimedRow);
    for (int it = 0; it < 64; it++) {
      const unsigned long long nm = __ballot(active & (!claimedRow | !((alivem >> claimer) & 1ull)));
      if (nm == alivem) break;
      alivem = nm;
    }
    const unsigned long long cm = __ballot(conflict & active) & alivem;
#ifdef RFS_PROFILE
    dbgSlackN += __popcll(__ballot((dbgWhy & 2) != 0));
    dbgUnlistN += __popcll(__ballot((dbgWhy & 1) != 0));
    if (B.dbg && i == 7) {
      const int w1 = __popcll(__ballot(dbgWhy & 1)), w2 = __popcll(__ballot(dbgWhy & 2)), w4 = __popcll(__ballot(dbgWhy & 4)), wc = __popcll(__ballot(conflict & active & !ovf));
      if (lane == 0) { B.dbg[52] = w1; B.dbg[53] = w2; B.dbg[54] = w4; B.dbg[55] = wc; B.dbg[56] = __popcll(actm); }
    }
#endif
    const int firstDirty = cm ? __builtin_ctzll(cm) : 64;
    // A launch ends with its slowest workgroups, and those are the particles whose replay validates many rows one by one (the replay's
    // length explains 0.92 of the merge phase's spread, tools/tail_study.py; a particle is slow launch after launch).  The replay runs at
    // the lowest issue priority (step_fused.h); a workgroup that finds itself with a long serial tail takes the map update's level back.
    if (MERGE_P2_BOOST > 0 && firstDirty < 64 && __popcll(actm >> firstDirty) >= MERGE_P2_BOOST) __builtin_amdgcn_s_setprio(MERGE_P2_BOOST_PRIO);
    const bool commitNow = active && lane < firstDirty && ((alivem >> lane) & 1ull) && nAbs > 0;
    if (commitNow) {
      for (int k = 0; k < nAbs; k++) sRad[sSpec[lane * 8 + k]] = -1.f;
    }
    unsigned long long commit = __ballot(commitNow);
#ifdef RFS_PROFILE
    dbgRows += __popcll(alivem & ((firstDirty < 64) ? ((1ull << firstDirty) - 1ull) : ~0ull));
#endif
    wave_sync();
    for (int l = firstDirty; l < cnt; l++) {
      if (!((actm >> l) & 1ull)) continue;
      const int al = __builtin_amdgcn_readlane(a, l);
      if (sRad[al] < 0.f) continue;  // absorbed by an earlier row of this round: the row no longer exists
#ifdef RFS_PROFILE
      dbgRows++;
      dbgSerial++;
#endif
      const int nl = __builtin_amdgcn_readlane(nAbs, l);
      const bool ovl = (__ballot(ovf) >> l) & 1ull;
      bool taken = false;
      int mine = 0;
      if (lane < nl) { mine = sSpec[l * 8 + lane]; taken = sRad[mine] < 0.f; }
      if (ovl) {
        seq_replay(al);  // the row could not be finished from its list: the reference's scan, exactly
      } else if (__ballot(taken) != 0ull) {
        // An entry this row absorbed is gone.  Every earlier row is final now, so the row's own walk, redone by its
        // lane against the present liveness, is final too (unless it outgrows its list: then the full scan).
#ifdef RFS_PROFILE
        const long long rw0 = (long long)__builtin_readcyclecounter();
#endif
        if (lane == l) walk();
        wave_sync();
#ifdef RFS_PROFILE
        dbgRewalks++;
        dbgRewalkCycles += (long long)__builtin_readcyclecounter() - rw0;
#endif
        if ((__ballot(ovf) >> l) & 1ull) {
          seq_replay(al);
        } else {
          const int nl2 = __builtin_amdgcn_readlane(nAbs, l);
          if (lane < nl2) sRad[sSpec[l * 8 + lane]] = -1.f;
          if (nl2 > 0) commit |= 1ull << l;
          wave_sync();
        }
      } else {
        if (lane < nl) sRad[mine] = -1.f;
        if (nl > 0) commit |= 1ull << l;
        wave_sync();
      }
    }
#endif
    if ((commit >> lane) & 1ull) {  // committed rows publish their merged state
      anyMerge = true;
      sW[a] = aw;
      pW[pa] = aw; pMX[pa] = ax; pMY[pa] = ay; pSXX[pa] = axx; pSXY[pa] = axy; pSYY[pa] = ayy;
    }
    anyMerge = __ballot(anyMerge) != 0ull;
    // claims are per round
    if (r0 + 64 < nRowsTotal)
      for (int m = lane; m < N; m += 64) sRec[m] |= MERGE_REC_NOCLAIM;
    wave_sync();
    DBG_TB(32, 11);
  }
#ifdef RFS_PROFILE
  if (B.dbg && i == 7 && lane == 0) { B.dbg[48] = dbgRows; B.dbg[49] = dbgMerges; B.dbg[50] = dbgChunks; B.dbg[51] = N; }
#endif
  if (lane == 0) sRed[0] = anyMerge ? 1.f : 0.f;
#ifdef RFS_PROFILE
  dbgT3 = (long long)__builtin_readcyclecounter();
#endif
  }  // ======== end of wave 0's phase 2 ========
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // in-place updates of merged rows (global) -> visible to the workgroup
  block_sync();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  anyMerge = sRed[0] != 0.f;
  // hole flags back into the per-thread registers used by the write-back / prune below
  for (int m = tid, sidx = 0; m < N; m += NT, sidx++)
    if (sRad[m] < 0.f) hole |= 1u << sidx;

  if (!FUSE_PRUNE) {
    if (!anyMerge) return;
    for (int m = tid, sidx = 0; m < N; m += NT, sidx++)
      if ((hole >> sidx) & 1u) pW[phys(m)] = -1.0;
    return;
  }

  DBG_TB(32, 3);
  RFS_CUT(24);
#ifdef MERGE_PRUNE_PRIO
  __builtin_amdgcn_s_setprio(MERGE_PRUNE_PRIO);
#endif
  // ---- fused prune: keep w >= t (not absorbed), order (weight desc, index asc), compact into the other slab ----
  for (int m = tid, sidx = 0; m < N; m += NT, sidx++)
    if ((hole >> sidx) & 1u) sW[m] = -1.0;
  block_sync();
  double *dl = B.slab[dst];
  const double t = P.pruneT;
  // survivors are compacted into sSorted (ascending index) so that ranks only need the survivors' keys
  int nSurv = 0;
  if (wave == 0) {
    for (int c0 = 0; c0 < N; c0 += 64) {
      const int m = c0 + lane;
      const double wm = (m < N) ? sW[m] : -1.0;
      const bool keep = (wm >= t) && (wm >= 0.0);
      const unsigned long long km = __ballot(keep);
      if (keep) sSorted[nSurv + __popcll(km & ((1ull << lane) - 1ull))] = (unsigned short)m;
      nSurv += __popcll(km);
    }
    if (lane == 0) *sPairCount = (unsigned)nSurv;
  }
  block_sync();
  nSurv = (int)*sPairCount;
  auto put = [&](const int rank, const int m) {
    const int pm = phys(m);
    const double vx = pMX[pm], vy = pMY[pm], vxx = pSXX[pm], vxy = pSXY[pm], vyy = pSYY[pm];
    plane(dl, cap, i, PL_W)[rank] = sW[m];
    plane(dl, cap, i, PL_WP)[rank] = 0.0;
    plane(dl, cap, i, PL_MX)[rank] = vx;
    plane(dl, cap, i, PL_MY)[rank] = vy;
    plane(dl, cap, i, PL_SXX)[rank] = vxx;
    plane(dl, cap, i, PL_SXY)[rank] = vxy;
    plane(dl, cap, i, PL_SYY)[rank] = vyy;
  };
  // ranks among the survivors through the weighting phase's bucket sort (weighting.h): the grid's cursor array is the
  // histogram (1024 buckets, inside either grid's cursor array), the pair list holds the bucket order, ties rank by list position (= ascending index).
  // The order goes to sOrder (rank -> entry) first: survivors of equal weight are then put into the order std::sort leaves them in
  // (stdsort_replay.h; GaussianMixture::prune sorts the WHOLE list, merged-away entries -- weight 0 -- included, :477-534), and only
  // then is the compacted mixture written.
  unsigned short *sOrder = sSlack;                                              // [nSurv] (the rows' slacks are dead)
  constexpr int NS = 8;
  bool ranked = false, tiedKeys = false;
  if (nSurv <= NS * NT) {
    int rl[NS];
    ranked = bucket_rank_sort<WPP, NS>([&](int q) { return sW[sSorted[q]]; }, nSurv, tid, sCellStart, reinterpret_cast<unsigned short *>(sPairs),
                                       MERGE_LOG_CELLS, reinterpret_cast<int *>(sRed), rl, block_sync, tiedKeys);
    if (ranked) {
      const bool any = __ballot(tiedKeys) != 0ull;
      if (lane == 0) reinterpret_cast<int *>(sRed)[4 * WPP + wave] = any ? 1 : 0;   // (behind the sort's 2 WPP + 1 words of scratch)
#pragma unroll
      for (int k = 0; k < NS; k++) {
        const int q = tid + NT * k;
        if (q < nSurv) sOrder[rl[k]] = sSorted[q];
      }
    }
  }
  if (!ranked) {  // crowded buckets (many equal weights) or more survivors than the register slots cover: all pairs
    for (int q = tid; q < nSurv; q += NT) {
      const int m = sSorted[q];
      const double wm = sW[m];
      int rank = 0;
      for (int q2 = 0; q2 < nSurv; q2++) {
        const int j2 = sSorted[q2];
        const double wj = sW[j2];
        rank += (wj > wm || (wj == wm && j2 < m)) ? 1 : 0;
      }
      sOrder[rank] = (unsigned short)m;
    }
  }
  block_sync();
  if (ranked) {
    tiedKeys = false;
#pragma unroll
    for (int w2 = 0; w2 < WPP; w2++) tiedKeys |= reinterpret_cast<int *>(sRed)[4 * WPP + w2] != 0;
  } else tiedKeys = true;
  {
    StdSortScratch ss;                              // the fp32 positions / radii of the merge are dead: 12 B per entry
    ss.T = reinterpret_cast<unsigned *>(sX);        // [N] words (group | entry)
    ss.pos = reinterpret_cast<unsigned short *>(sY);  // [N]        | Ll [N / 2 + 1]   (3 N + 2 <= 4 cap bytes)
    ss.posStride = 1;
    ss.Ll = ss.pos + N;
    ss.Rl = reinterpret_cast<unsigned short *>(sRad);                           // [N / 2 + 1] | eq | stack   (N + 2 + cap / 8 + 112 <= 4 cap bytes)
    ss.eq = reinterpret_cast<unsigned long long *>(reinterpret_cast<unsigned char *>(sRad) + ((((size_t)(N / 2 + 1) * 2) + 7) & ~(size_t)7));
    ss.stack = reinterpret_cast<unsigned *>(ss.eq + ((N + 63) >> 6));
    auto key0 = [&](int e) { const double w = sW[e]; return w < 0.0 ? 0.0 : w; };          // merged-away entries: weight 0 in the reference
    auto is_rest = [&](int e) { const double w = sW[e]; return !((w >= t) && (w >= 0.0)); };  // not a survivor
    // the entries below the survivors: group = nSurv + the number of such entries with a larger key (equal keys share a group)
    auto rest_group = [&](unsigned *T) {
      for (int e = tid; e < N; e += NT) {
        if (!is_rest(e)) continue;
        const double k = key0(e);
        int g = nSurv;
        for (int e2 = 0; e2 < N; e2++) g += (is_rest(e2) && key0(e2) > k) ? 1 : 0;
        T[e] = ((unsigned)g << 16) | (unsigned)g;
      }
    };
    ss_correct_tie_order<WPP>(key0, [&](int r) { return (int)sOrder[r]; }, [&](int r, unsigned short e) { sOrder[r] = e; }, rest_group, N, nSurv, ss, tid,
                              block_sync, tiedKeys);
  }
  for (int r = tid; r < nSurv; r += NT) put(r, sOrder[r]);
  if (tid == 0) B.count[i] = nSurv;
  DBG_TB(32, 4);
#ifdef RFS_PROFILE
  if (B.dbg && tid == 0) {
    long long *d = B.dbg + 64 + 4 * (size_t)i;
    d[0] = (long long)__builtin_readcyclecounter() - dbgT0; d[1] = ((dbgT3 - dbgT2) & 0xffffffll) | ((dbgWalk1 & 0xffffffll) << 24) | ((long long)dbgRoundsN << 48); d[2] = (long long)(dbgFallbacks | (dbgSlackN << 8) | (dbgUnlistN << 16)) | ((long long)min(dbgSerial, 255) << 24) | ((long long)min(dbgRewalks, 255) << 32) | (min(dbgRewalkCycles, 0xfffffll) << 40); atomicAdd((unsigned long long *)&d[3], (unsigned long long)N | ((unsigned long long)min(dbgPairs, 65535u) << 16));
  }
#endif
}

template <int WPP, bool FUSE_PRUNE>
__global__ __launch_bounds__(WPP * 64) __attribute__((amdgpu_waves_per_eu(MERGE_WAVES_PER_EU))) void gm_merge_kernel(Buffers B, Params P, int cur, int dst) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  gm_merge_particle<WPP, FUSE_PRUNE>(B, P, cur, dst, (int)blockIdx.x, (int)threadIdx.x, smem_raw);
}

// LDS per wave: keys[cap] doubles + the order / std::sort-replay scratch (u16 each: order [cap], T [cap], pos [cap], two stopper
// lists [cap / 2 + 1]; eq words, stack)
__host__ __device__ inline size_t gm_prune_lds_bytes_per_wave(int cap) {
  // keys f64 | order u16 | T u32 | pos u16 | two stopper lists u16 [cap / 2 + 2] | eq words | stack
  return (((size_t)cap * 8 + (size_t)cap * 2 + (size_t)cap * 4 + (size_t)cap * 2 + (size_t)(cap / 2 + 2) * 2 * 2 + (size_t)((cap + 63) / 64) * 8 + 128) + 15) & ~(size_t)15;
}
// NEG_IS_HOLE: the RB-PHD mixtures mark merged-away entries with w = -1; FastSLAM's log-odds weights are legitimately negative.
// GaussianMixture::prune (include/GaussianMixture.hpp:477-534): std::sort of the whole list by weight, the sorted prefix with
// w >= t stays.  Ranks by counting (ties by index), then equal weights in std::sort's order (stdsort_replay.h).
template <int WPB, bool NEG_IS_HOLE = true>
__global__ __launch_bounds__(WPB * 64) void gm_prune_kernel(Buffers B, Params P, int src, int dst) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const int wave = threadIdx.x >> 6;
  const int lane = threadIdx.x & 63;
  const int i = __builtin_amdgcn_readfirstlane(blockIdx.x * WPB + wave);
  if (i >= B.N) return;
  const int cap = B.cap;
  unsigned char *base = smem_raw + (size_t)wave * gm_prune_lds_bytes_per_wave(cap);
  double *keys = reinterpret_cast<double *>(base);
  unsigned short *sOrder = reinterpret_cast<unsigned short *>(keys + cap);
  const int N = B.count[i];
  const double *sl = B.slab[src];
  double *dl = B.slab[dst];
  const double *qW = sl + ((size_t)i * B.npl + 0) * cap;  // plane 0 is the weight in every layout
  const double t = P.pruneT;
  for (int m = lane; m < N; m += 64) keys[m] = qW[m];
  wave_sync();
  int kept = 0;
  for (int m = lane; m < N; m += 64) {
    const double wm = keys[m];
    const bool keep = (wm >= t) && (!NEG_IS_HOLE || wm >= 0.0);  // holes carry -1
    if (keep) {
      int rank = 0;
      for (int j = 0; j < N; j++) {
        const double wj = keys[j];
        rank += (wj > wm || (wj == wm && j < m)) ? 1 : 0;  // everything ranked ahead of a survivor also survives
      }
      sOrder[rank] = (unsigned short)m;
      kept++;
    }
  }
  kept = wave_sum_i(kept);
  wave_sync();
  {
    StdSortScratch ss;
    ss.T = reinterpret_cast<unsigned *>(sOrder + cap);
    ss.pos = reinterpret_cast<unsigned short *>(ss.T + cap);
    ss.posStride = 1;
    ss.Ll = ss.pos + cap;
    ss.Rl = ss.Ll + (cap / 2 + 2);
    ss.eq = reinterpret_cast<unsigned long long *>(base + (size_t)cap * 16 + (size_t)(cap / 2 + 2) * 4);
    ss.stack = reinterpret_cast<unsigned *>(ss.eq + (cap + 63) / 64);
    auto key0 = [&](int e) { const double w = keys[e]; return (NEG_IS_HOLE && w < 0.0) ? 0.0 : w; };
    auto is_rest = [&](int e) { const double w = keys[e]; return !((w >= t) && (!NEG_IS_HOLE || w >= 0.0)); };
    auto rest_group = [&](unsigned *T) {
      for (int e = lane; e < N; e += 64) {
        if (!is_rest(e)) continue;
        const double k = key0(e);
        int g = kept;
        for (int e2 = 0; e2 < N; e2++) g += (is_rest(e2) && key0(e2) > k) ? 1 : 0;
        T[e] = ((unsigned)g << 16) | (unsigned)g;
      }
    };
    ss_correct_tie_order<1>(key0, [&](int r) { return (int)sOrder[r]; }, [&](int r, unsigned short e) { sOrder[r] = e; }, rest_group, N, kept, ss, lane,
                            [&]() { wave_sync(); });
  }
  for (int r = lane; r < kept; r += 64) {
    const int m = sOrder[r];
    for (int pl = 0; pl < B.npl; pl++)
      (dl + ((size_t)i * B.npl + pl) * cap)[r] = (sl + ((size_t)i * B.npl + pl) * cap)[m];
  }
  if (lane == 0) B.count[i] = kept;
}

// ---- small kernels ---------------------------------------------------------------------------------

// {sum w, sum w^2} of this shard; one block, deterministic tree.
__global__ __launch_bounds__(1024) void weight_sums_kernel(const double *w, int N, double *out2) {
  __shared__ double s0[16], s1[16];
  double a = 0, b = 0;
  for (int k = threadIdx.x; k < N; k += 1024) { double v = w[k]; a += v; b += v * v; }
  a = wave_sum(a); b = wave_sum(b);
  if ((threadIdx.x & 63) == 0) { s0[threadIdx.x >> 6] = a; s1[threadIdx.x >> 6] = b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double x = 0, y = 0;
    for (int k = 0; k < 16; k++) { x += s0[k]; y += s1[k]; }
    out2[0] = x; out2[1] = y;
  }
}
// sumDev: nParts pairs {sum w, sum w^2} (one per shard sharing the normalisation); the divisor is their sum, in order
__global__ void normalize_kernel(double *w, int N, double sum, const double *sumDev, int nParts) {
  double sdiv = sum;
  if (sumDev) {
    sdiv = sumDev[0];
    for (int p = 1; p < nParts; p++) sdiv += sumDev[2 * p];
  }
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < N) w[k] = w[k] / sdiv;
}
__global__ void set_weights_kernel(double *w, int N, double v) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < N) w[k] = v;
}

// Resample copy: slot k takes slot src[k]'s mixture (Particle::copy -> GaussianMixture copy ctor).
// One block per destination slot; sources are slots that keep themselves, so in-place is hazard-free.
__global__ __launch_bounds__(256) void resample_gather_kernel(Buffers B, int cur, const int *srcSlot, int poseCovStride, int mapOnly) {
  const int k = blockIdx.x;
  const int s = srcSlot[k];
  if (s == k) return;
  const int n = B.count[s];
  double *slab = B.slab[cur];
  for (int pl = 0; pl < B.npl; pl++) {
    const double *q = slab + ((size_t)s * B.npl + pl) * (size_t)B.cap;
    double *d = slab + ((size_t)k * B.npl + pl) * (size_t)B.cap;
    for (int m = threadIdx.x; m < n; m += blockDim.x) d[m] = q[m];
  }
  // the pose travels with the particle (Particle::copy, include/Particle.hpp:218-223)
  if (threadIdx.x < 3) B.pose[3 * (size_t)k + threadIdx.x] = B.pose[3 * (size_t)s + threadIdx.x];
  if (poseCovStride == 9 && threadIdx.x < 9) B.poseCov[9 * (size_t)k + threadIdx.x] = B.poseCov[9 * (size_t)s + threadIdx.x];
  if (mapOnly) {   // RFSGPU_INHERIT_REFERENCE / _EXTERNAL: the per-slot birth state stays where it is (birth.h, birth_inherit_kernel)
    if (threadIdx.x == 0) B.count[k] = n;
    return;
  }
  // RFSGPU_INHERIT_EAGER: birth bookkeeping travels with the particle: unused list, FOV count, candidate list
  const int nc = B.candCount[s];
  for (int t = threadIdx.x; t < nc * 3; t += blockDim.x) B.candMean[(size_t)k * RFSGPU_MAX_CANDIDATES * 3 + t] = B.candMean[(size_t)s * RFSGPU_MAX_CANDIDATES * 3 + t];
  for (int t = threadIdx.x; t < nc * 6; t += blockDim.x) B.candCov[(size_t)k * RFSGPU_MAX_CANDIDATES * 6 + t] = B.candCov[(size_t)s * RFSGPU_MAX_CANDIDATES * 6 + t];
  for (int t = threadIdx.x; t < nc; t += blockDim.x) {
    B.candSup[(size_t)k * RFSGPU_MAX_CANDIDATES + t] = B.candSup[(size_t)s * RFSGPU_MAX_CANDIDATES + t];
    B.candChk[(size_t)k * RFSGPU_MAX_CANDIDATES + t] = B.candChk[(size_t)s * RFSGPU_MAX_CANDIDATES + t];
  }
  if (threadIdx.x == 0) {
    B.count[k] = n;
    B.unusedMask[k] = B.unusedMask[s];
    B.nInFov[k] = B.nInFov[s];
    B.candCount[k] = nc;
  }
}

// ---- cross-shard migration of whole particles (multi-GPU resampling) ----------------------------------------
// A particle leaves / enters a shard as one fixed-size packed ROW in a device buffer: header (16 doubles: count, FOV count,
// unused mask, pose, pose covariance, candidate count) | npl planes of `cap` doubles (the first `count` of each are live) |
// the birth-candidate block.  The rows of all migrants of a step sit back to back, so that the transport (RCCL send/recv
// between processes, hipMemcpyPeerAsync inside one process) moves device memory to device memory and the host only ever
// sees slot indices.  What Particle::copy + RBPHDFilter.hpp:1005-1011 carry: pose, mixture, unused list, FOV count, candidates.
// rowCand: candidates a row has room for -- RFSGPU_MAX_CANDIDATES for a filter whose configuration keeps candidate lists, 0
// for one that does not (the 2-D simulator's CountThreshold == 1: 20 KB per migrant that would only ever carry zeros).
#define RFSGPU_ROW_HEADER_DOUBLES 16
__host__ __device__ inline size_t slab_row_bytes(int npl, int cap, int rowCand) {
  return (size_t)RFSGPU_ROW_HEADER_DOUBLES * 8 + (size_t)npl * cap * 8 + (size_t)rowCand * (3 + 6) * 8 + (size_t)rowCand * 2 * 4;
}
template <bool EXPORT>
__global__ __launch_bounds__(256) void slab_rows_kernel(Buffers B, int cur, const int *slots, unsigned char *rows, int poseCovStride, int rowCand, int mapOnly) {
  const int s = slots[blockIdx.x];
  unsigned char *row = rows + (size_t)blockIdx.x * slab_row_bytes(B.npl, B.cap, rowCand);
  double *hdr = reinterpret_cast<double *>(row);
  double *pl = hdr + RFSGPU_ROW_HEADER_DOUBLES;
  double *cm = pl + (size_t)B.npl * B.cap, *cc = cm + (size_t)rowCand * 3;
  int *cs = reinterpret_cast<int *>(cc + (size_t)rowCand * 6), *ck = cs + rowCand;
  double *slab = B.slab[cur] + (size_t)s * B.npl * (size_t)B.cap;
  const size_t cb = (size_t)s * RFSGPU_MAX_CANDIDATES;
  if (EXPORT) {
    const int n = B.count[s];
    int nc = mapOnly ? 0 : B.candCount[s];   // (mapOnly: the row carries what Particle::copy carries -- pose + mixture)
    if (nc > rowCand) {   // a list the row has no room for (the configuration says there is none): refused loudly, never dropped silently
      if (threadIdx.x == 0) atomicOr(B.err, ERRBIT_BIRTHLIST);
      nc = rowCand;
    }
    for (int p = 0; p < B.npl; p++)
      for (int m = threadIdx.x; m < n; m += blockDim.x) pl[(size_t)p * B.cap + m] = slab[(size_t)p * B.cap + m];
    for (int t = threadIdx.x; t < nc * 3; t += blockDim.x) cm[t] = B.candMean[cb * 3 + t];
    for (int t = threadIdx.x; t < nc * 6; t += blockDim.x) cc[t] = B.candCov[cb * 6 + t];
    for (int t = threadIdx.x; t < nc; t += blockDim.x) { cs[t] = B.candSup[cb + t]; ck[t] = B.candChk[cb + t]; }
    if (threadIdx.x < 3) hdr[3 + threadIdx.x] = B.pose[3 * (size_t)s + threadIdx.x];
    if (threadIdx.x < 9) hdr[6 + threadIdx.x] = B.poseCov[(size_t)poseCovStride * s + threadIdx.x];
    if (threadIdx.x == 0) {
      hdr[0] = (double)n;
      hdr[1] = (double)B.nInFov[s];
      hdr[2] = __longlong_as_double((long long)B.unusedMask[s]);
      hdr[15] = (double)nc;
    }
  } else {
    const int n = (int)hdr[0], nc = mapOnly ? 0 : (int)hdr[15];
    for (int p = 0; p < B.npl; p++)
      for (int m = threadIdx.x; m < n; m += blockDim.x) slab[(size_t)p * B.cap + m] = pl[(size_t)p * B.cap + m];
    for (int t = threadIdx.x; t < nc * 3; t += blockDim.x) B.candMean[cb * 3 + t] = cm[t];
    for (int t = threadIdx.x; t < nc * 6; t += blockDim.x) B.candCov[cb * 6 + t] = cc[t];
    for (int t = threadIdx.x; t < nc; t += blockDim.x) { B.candSup[cb + t] = cs[t]; B.candChk[cb + t] = ck[t]; }
    if (threadIdx.x < 3) B.pose[3 * (size_t)s + threadIdx.x] = hdr[3 + threadIdx.x];
    if (poseCovStride == 9 && threadIdx.x < 9) B.poseCov[9 * (size_t)s + threadIdx.x] = hdr[6 + threadIdx.x];
    if (threadIdx.x == 0) {
      B.count[s] = n;
      if (!mapOnly) {
        B.nInFov[s] = (int)hdr[1];
        B.unusedMask[s] = (unsigned long long)__double_as_longlong(hdr[2]);
        B.candCount[s] = nc;
      }
    }
  }
}

// Number of valid Gaussians per particle (holes left by gm_merge carry w < 0 until gm_prune drops them).
__global__ __launch_bounds__(256) void valid_count_kernel(Buffers B, int cur, int *out) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= B.N) return;
  const int lane = threadIdx.x & 63;
  const int n = B.count[i];
  const double *w = B.slab[cur] + ((size_t)i * B.npl + 0) * B.cap;
  int c = 0;
  for (int m = lane; m < n; m += 64) c += (w[m] >= 0.0) ? 1 : 0;
  c = wave_sum_i(c);
  if (lane == 0) out[i] = c;
}

// rfsgpu_restore_state: copy the saved live entries back (one block per particle).
template <int NPL>
__global__ __launch_bounds__(256) void restore_state_kernel(Buffers B, int cur, const double *snapSlab, const double *snapWeight,
                                                            const int *snapCount, const int *snapFov, const unsigned long long *snapUnused) {
  const int k = blockIdx.x;
  const int n = snapCount[k];
  // every plane's element of an entry is loaded before the first is stored: NPL (7 or 11) loads in flight per thread
  const double *q0 = snapSlab + (size_t)k * NPL * (size_t)B.cap;
  double *d0 = B.slab[cur] + (size_t)k * NPL * (size_t)B.cap;
  for (int m = threadIdx.x; m < n; m += blockDim.x) {
    double v[NPL];
#pragma unroll
    for (int pl = 0; pl < NPL; pl++) v[pl] = q0[(size_t)pl * B.cap + m];
#pragma unroll
    for (int pl = 0; pl < NPL; pl++) d0[(size_t)pl * B.cap + m] = v[pl];
  }
  if (threadIdx.x == 0) {
    B.count[k] = n;
    B.weight[k] = snapWeight[k];
    B.nInFov[k] = snapFov[k];
    B.unusedMask[k] = snapUnused[k];
  }
}

// Map part of RBPHDFilter::predict (:415-442): birth Gaussians from the previous update's unused
// measurements at the current (pre-propagation) pose -- immediate-birth branch of addBirthGaussians
// (:1000-1084) with MeasurementModel_RngBrg::inverseMeasure (src/MeasurementModel_RngBrg.cpp:117-136) --
// then StaticProcessModel::staticStep, Sigma += Q (include/ProcessModel.hpp:195-208).
// One particle's share of it, by NT threads (tid 0 ... NT-1; the births are lane work of the first wavefront, the static step is
// spread over all threads).  birthPose: the poses the births happen at -- the ones the previous update used (the reference adds
// the birth Gaussians before it propagates the particles, :424-431).  The stand-alone kernel below and the head of the fused
// cycle (step_fused.h, round 5) both run THIS function: same expressions, same bits.
template <int NT>
__device__ __forceinline__ void predict_map_particle(const Buffers &B, const Params &P, const int cur, const int i, const int tid, bool addBirth,
                                                     const int nZprev, const double *birthPose, const bool doStatic) {
  const int lane = tid & 63;
  const int cap = B.cap;
  double *slab = B.slab[cur];
  double *pW = plane(slab, cap, i, PL_W), *pWP = plane(slab, cap, i, PL_WP), *pMX = plane(slab, cap, i, PL_MX),
         *pMY = plane(slab, cap, i, PL_MY);
  double *pSXX = plane(slab, cap, i, PL_SXX), *pSXY = plane(slab, cap, i, PL_SXY), *pSYY = plane(slab, cap, i, PL_SYY);
  int n = B.count[i];
  const int nOld = n;  // staticStep below covers the pre-existing Gaussians; births get Q added where they are created
  // Several waves per particle (the fused cycle's head): EVERY wave must hold the pre-birth count before the first wave publishes
  // count + births below -- a wave that arrived late would otherwise take the new count for nOld, add Q to the newborn Gaussians a
  // second time and race with the first wave's stores to them.  The barrier waits for outstanding loads (vmcnt / lgkmcnt 0) first.
  // The call is block-uniform in the fused kernel; the one-wave stand-alone kernel needs nothing.
  if constexpr (NT > 64) __syncthreads();
  if (addBirth && nZprev > 0 && tid < 64) {
    const unsigned long long um = B.unusedMask[i];
    const bool immediate = (P.birthCountThr == 1u) || ((unsigned)B.nInFov[i] <= P.birthCurThr);
    if (um != 0ull && !immediate) {
      if (lane == 0) atomicOr(B.err, ERRBIT_BIRTHLIST);  // candidate-list mode is not on the device path yet
    } else if (um != 0ull) {
      // unused measurements are consumed back to front (:1013-1017): highest index first
      const int nb = __popcll(um);
      const bool mine = (um >> lane) & 1ull;
      const int rankFromTop = __popcll(um >> lane) - 1;  // 0 for the highest set bit
      if (mine) {
        const int pos = n + rankFromTop;
        if (pos < cap) {
          const double px = birthPose[3 * i], py = birthPose[3 * i + 1], pth = birthPose[3 * i + 2];
          const double zr = B.Z[2 * lane], zb = B.Z[2 * lane + 1];
          const double a = pth + zb;
          const double ca = cos(a), sa = sin(a);
          // Hinv = [ca, -zr*sa; sa, zr*ca];  cov = Hinv * R * Hinv^T
          const double h00 = ca, h01 = -zr * sa, h10 = sa, h11 = zr * ca;
          const double t00 = h00 * P.R[0] + h01 * P.R[2], t01 = h00 * P.R[1] + h01 * P.R[3];
          const double t10 = h10 * P.R[0] + h11 * P.R[2], t11 = h10 * P.R[1] + h11 * P.R[3];
          pW[pos] = P.birthW;
          pWP[pos] = 0.0;
          pMX[pos] = px + zr * ca;
          pMY[pos] = py + zr * sa;
          // birth covariance, then this predict's staticStep on it: (Hinv R Hinv^T) + Q
          pSXX[pos] = (t00 * h00 + t01 * h01) + P.Qlm[0];
          pSXY[pos] = (t00 * h10 + t01 * h11) + P.Qlm[1];
          pSYY[pos] = (t10 * h10 + t11 * h11) + P.Qlm[2];
        }
      }
      if (n + nb > cap) {
        if (lane == 0) atomicOr(B.err, ERRBIT_CAPACITY);
        n = cap;
      } else {
        n += nb;
      }
      if (lane == 0) { B.count[i] = n; B.unusedMask[i] = 0ull; }
    }
  }
  if (!doStatic) return;
  for (int m = tid; m < nOld; m += NT) {
    pSXX[m] += P.Qlm[0];
    pSXY[m] += P.Qlm[1];
    pSYY[m] += P.Qlm[2];
  }
}
template <int WPB>
__global__ __launch_bounds__(WPB * 64) void predict_map_kernel(Buffers B, Params P, int cur, int addBirth, int nZprev, BirthLevel LV) {
  const int wave = threadIdx.x >> 6;
  const int lane = threadIdx.x & 63;
  const int i = __builtin_amdgcn_readfirstlane(blockIdx.x * WPB + wave);
  if (i >= B.N) return;
  predict_map_particle<64>(B, P, cur, i, lane, addBirth && LV.mine(i), nZprev, B.pose, LV.doStatic != 0);
}

