// hungarian_wave.h -- HungarianMethod::run (reference include/HungarianMethod.hpp:91-587, maximise) with one WAVEFRONT per
// problem instead of one thread.
//
// The reference's solver is a serial search whose tie-breaks are observable: with exactly tied assignment scores (FastSLAM's
// floor-valued table cells tie all the time) WHICH optimal assignment comes out depends on the order rows / columns are
// scanned in, and Murty's ranked enumeration on top of it inherits that.  So the traversal is kept exactly: the same step
// sequence, the same "first index" / "last index" choices, the same tolerances (1e-14 / 1e-12), the same in-place offset
// subtract / re-add on the table.  What changes is who runs the O(n) inner loops: row x's state (lx, xy, S, the BFS marks)
// lives in lane x's registers, column y's state (ly, slack, yx, T, NS) in lane y's, a scan over y or x is one wave
// instruction plus a ballot / min-reduction (min and max are exact in any order), and a table row is one coalesced 64-lane
// load.  n <= 64.  Every scalar decision is made from ballots and readlanes, so control flow is wave-uniform.
//
// All 64 lanes call hungarian_wave together.  C: n x n, leading dimension ld, LDS or global (lane y only ever touches
// column y; the table is published to the other lanes on return).  (`queue` is unused: the BFS queue lives in registers.)
// On return lane x < n holds xy[x] (the column assigned to row x); *cost (uniform) = sum_x C[x][xy[x]] added in row order.
#pragma once
#include "common.h"

// min / max over the wave by DPP (exact in any order).  bound_ctrl off + old = own value: a lane without a valid partner, or in
// a row the step does not address, combines with itself.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_keep_f64(double v) {
  const int l = __double2loint(v), h = __double2hiint(v);
  const int lo = __builtin_amdgcn_update_dpp(l, l, CTRL, ROW_MASK, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(h, h, CTRL, ROW_MASK, 0xf, false);
  return __hiloint2double(hi, lo);
}
// (v_min_f64 / v_max_f64 as they are: fmin() / fmax() put a canonicalising v_max_f64 v, v, v in front of every step -- six more
//  fp64 instructions in a chain that is all latency.  They differ from fmin / fmax for signalling NaNs only, which arithmetic
//  does not produce.)
__device__ __forceinline__ double raw_min_f64(double a, double b) { double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ double raw_max_f64(double a, double b) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ double wave_min_f64(double v) {
  v = raw_min_f64(v, dpp_keep_f64<0xb1, 0xf>(v));   // quad_perm [1,0,3,2]
  v = raw_min_f64(v, dpp_keep_f64<0x4e, 0xf>(v));   // quad_perm [2,3,0,1]
  v = raw_min_f64(v, dpp_keep_f64<0x114, 0xf>(v));  // row_shr:4
  v = raw_min_f64(v, dpp_keep_f64<0x118, 0xf>(v));  // row_shr:8
  v = raw_min_f64(v, dpp_keep_f64<0x142, 0xa>(v));  // row_bcast:15
  v = raw_min_f64(v, dpp_keep_f64<0x143, 0xc>(v));  // row_bcast:31 -> lane 63
  return readlane_f64(v, 63);
}
__device__ __forceinline__ double wave_max_f64(double v) {
  v = raw_max_f64(v, dpp_keep_f64<0xb1, 0xf>(v));
  v = raw_max_f64(v, dpp_keep_f64<0x4e, 0xf>(v));
  v = raw_max_f64(v, dpp_keep_f64<0x114, 0xf>(v));
  v = raw_max_f64(v, dpp_keep_f64<0x118, 0xf>(v));
  v = raw_max_f64(v, dpp_keep_f64<0x142, 0xa>(v));
  v = raw_max_f64(v, dpp_keep_f64<0x143, 0xc>(v));
  return readlane_f64(v, 63);
}
// sum of v over lanes 0..n-1 added in lane order (the reference's serial loops), uniform result
__device__ __forceinline__ double wave_ordered_sum(double v, int n) {
  double acc = 0;
  for (int r = 0; r < n; r++) acc += readlane_f64(v, r);
  return acc;
}

// SCRATCH (Murty's children: the table is a private LDS tile nobody reads afterwards, and the caller computes the score from the
// job's table itself): offset and greedy start run ROW-parallel -- lane x walks row x for its minimum, then for its offset-free
// maximum and the LAST column holding it (what the reference's `>=` scan keeps) -- instead of one wave-wide max-reduction per
// row, and the end of the solve neither re-adds the offset nor sums the cost.  Same lx, xy, yx out of the start; *cost = 0.
// lxRaw (may be null): on success lane x < n receives the row's dual variable for the table AS GIVEN (lx + offset: with the column
// duals the solve ends with, lx + ly >= C - offset and equality on the assignment) -- what hungarian_warm_wave below starts a
// child's solve from.
template <bool SCRATCH = false>
__device__ __forceinline__ bool hungarian_wave(double *C, int ld, int n, int &xyOut, double *cost, unsigned char *queue, long long *prof = nullptr,
                                               double *lxRaw = nullptr) {
  const int lane = threadIdx.x & 63;
  const bool in = lane < n;
  double *Ccol = C + lane;           // column `lane`
  // row-indexed state (lane = x) and column-indexed state (lane = y)
  double lx = 0, ly = 0, slack = 0;
  int xy = -1, yx = -1;
  bool S = false, T = false, NS = false;
  int px = -1, py = -1;              // BFS parents: p[x], p[y + n]
  bool xq = false, yq = false;
  double offset = 0;
  if constexpr (SCRATCH) {
    double *Crow = C + lane * ld;    // row `lane`
    double mn = 0;
    if (in)
      for (int j = 0; j < n; j++) mn = raw_min_f64(mn, Crow[j]);
    offset = wave_min_f64(mn);
    double m = -1.0;                 // (every real cell is >= 0 once the offset is gone)
    int yy = 0;
    if (in)
      for (int j = 0; j < n; j++) {
        const double v = Crow[j] - offset;
        Crow[j] = v;
        if (v >= m) { m = v; yy = j; }
      }
    lx = in ? m : 0.0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // the rows written above are read by columns from here on
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    for (int x = 0; x < n; x++) {    // step 1 (:162-190): the rows claim their columns in row order
      const int yyx = __builtin_amdgcn_readlane(yy, x);
      const int x_t = __builtin_amdgcn_readlane(yx, yyx);
      bool keep = true;
      if (x_t != -1) {
        const double mx = readlane_f64(lx, x), c_t = readlane_f64(lx, x_t);   // c_t == C[x_t][yyx]: yyx is row x_t's maximum
        keep = mx > c_t;
        if (keep && lane == x_t) xy = -1;
      }
      if (keep) {
        if (lane == x) xy = yyx;
        if (lane == yyx) yx = x;
      }
    }
  } else {

  // offset = min(0, min C); C -= offset (:128-160)
  double mn = 0;
  if (in)
    for (int x = 0; x < n; x++) mn = fmin(mn, Ccol[x * ld]);
  offset = wave_min_f64(mn);
  if (in)
    for (int x = 0; x < n; x++) Ccol[x * ld] -= offset;

  // step 1 (:162-190): greedy start, row by row
  for (int x = 0; x < n; x++) {
    const double v = in ? Ccol[x * ld] : -1.0;                  // every real cell is >= 0 now
    const double m = wave_max_f64(v);
    const unsigned long long eq = __ballot(in && v == m);
    const int yy = 63 - __builtin_clzll(eq);                     // `>=` keeps the LAST maximum
    if (lane == x) { lx = m; xy = yy; }
    const int x_t = __builtin_amdgcn_readlane(yx, yy);
    if (x_t != -1) {
      const double c_t = readlane_f64(lx, x_t);                  // == C[x_t][yy]: yy is row x_t's maximum
      if (m > c_t) {
        if (lane == x_t) xy = -1;
        if (lane == yy) yx = x;
      } else if (lane == x) {
        xy = -1;
      }
    } else if (lane == yy) {
      yx = x;
    }
  }

  }
  bool pickFreeVertex = true;
  int root = 0;
#ifdef RFS_PROFILE
  const long long tMain = (long long)__builtin_readcyclecounter();
  if (prof) prof[7] += n;
#endif
  for (int guard = 0; guard < 8 * 64 * 64; guard++) {
#ifdef RFS_PROFILE
    if (prof) prof[4]++;
#endif
    if (pickFreeVertex) {  // step 2
      S = false; T = false; NS = false;
      const unsigned long long fr = __ballot(in && xy == -1);
      if (fr == 0) {
        if (lxRaw) *lxRaw = lx + offset;
        if constexpr (SCRATCH) {
          *cost = 0;
          xyOut = xy;
          return true;
        }
        if (offset != 0 && in)
          for (int x = 0; x < n; x++) Ccol[x * ld] = Ccol[x * ld] + offset;
        const double mine = in ? Ccol[yx * ld] : 0.0;            // C[x][xy[x]] sits with the lane of column xy[x]
        double c = 0;
        for (int x = 0; x < n; x++) c += readlane_f64(mine, __builtin_amdgcn_readlane(xy, x));
        *cost = c;
        xyOut = xy;
#ifdef RFS_PROFILE
        if (prof) prof[8] += (long long)__builtin_readcyclecounter() - tMain;
#endif
        // the caller's other lanes may read any cell next: publish the rewritten table
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        return true;
      }
      root = __builtin_ctzll(fr);
      if (lane == root) S = true;
      const double lxr = readlane_f64(lx, root);
      if (in) {
        slack = lxr + ly - Ccol[root * ld];
        if (fabs(slack) < 1e-14) { slack = 0; NS = true; }
      }
    }
    // step 3
    if (__ballot(in && (NS != T)) == 0) {
#ifdef RFS_PROFILE
      if (prof) prof[6]++;
#endif
      const double a = wave_min_f64((in && !T) ? slack : 1.7976931348623157e308);
      if (S) lx -= a;
      if (T) ly += a;
      if (in) {
        if (!T) slack -= a;
        if (slack == 0) NS = true;
      }
    }
    // step 4
    const unsigned long long cand = __ballot(in && NS && !T);
    if (cand == 0) return false;
    const int y = __builtin_ctzll(cand);
    int x_t = __builtin_amdgcn_readlane(yx, y);
    if (x_t == -1) {
      // augmenting path root -> y by breadth-first search over tight edges (:420-523), same visiting order.  The queue
      // (<= 2n node ids) lives in two registers spread over the lanes: entry k on lane k & 63 of q0 (k < 64) or q1.
      const int target = y + n;
      int qh = 0, qt = 1;
      int q0 = root, q1 = 0;
      xq = (lane == root); yq = false;
      px = -1; py = -1;
      bool found = false;
      // The reference dequeues until the target column comes out; its path back only follows parents that were set before
      // the target was DISCOVERED (p[] is written once per node), so the search can stop at the discovery: same xy / yx.
      while (qh < qt) {
#ifdef RFS_PROFILE
        if (prof) prof[5]++;
#endif
        int t = (qh < 64) ? __builtin_amdgcn_readlane(q0, qh) : __builtin_amdgcn_readlane(q1, qh - 64);
        qh++;
        if (t < n) {
          const double lxt = readlane_f64(lx, t);
          const int xyt = __builtin_amdgcn_readlane(xy, t);
          const bool push = in && fabs(lxt + ly - Ccol[t * ld]) < 1e-12 && !yq && xyt != lane;
          const unsigned long long pm = __ballot(push);
          if (push) { yq = true; py = t; }
          if ((pm >> y) & 1ull) { found = true; break; }
          for (unsigned long long g = pm; g; g &= g - 1ull) {      // enqueue in ascending column order
            const int v = __builtin_ctzll(g) + n;
            if (lane == (qt & 63)) { if (qt < 64) q0 = v; else q1 = v; }
            qt++;
          }
        } else {
          t -= n;
          const int x = __builtin_amdgcn_readlane(yx, t);        // the only row that can pass `yx[t] == x`
          if (x >= 0) {
            const double lxx = readlane_f64(lx, x);
            const bool tight = ((__ballot(in && fabs(lxx + ly - Ccol[x * ld]) < 1e-12) >> t) & 1ull) != 0;   // lane t's verdict
            const bool Sx = (__ballot(S) >> x) & 1ull, xqx = (__ballot(xq) >> x) & 1ull;
            if (tight && Sx && !xqx) {
              if (lane == x) { xq = true; px = t + n; }
              if (lane == (qt & 63)) { if (qt < 64) q0 = x; else q1 = x; }
              qt++;
            }
          }
        }
      }
      if (found) {
        int t = target;
        while (t != root) {
          if (t >= n) {
            const int xt = __builtin_amdgcn_readlane(py, t - n);
            if (lane == xt) xy = t - n;
            if (lane == t - n) yx = xt;
            t = xt;
          } else {
            t = __builtin_amdgcn_readlane(px, t);
          }
        }
      }
      if (!found) return false;
      pickFreeVertex = true;
    } else {
      if (lane == x_t) S = true;
      if (lane == y) T = true;
      const double lxt = readlane_f64(lx, x_t);
      if (in) {
        const double d = lxt + ly - Ccol[x_t * ld];
        if (fabs(d) < 1e-14) NS = true;
        if (d < slack) slack = d;
      }
      pickFreeVertex = false;
    }
  }
  return false;
}

// ---- one augmentation from the parent's dual variables (round 6) -------------------------------------------------------------
// A child of a Murty expansion is its parent's sub-problem with rows dropped (they stay with the parent's columns) and a few cells
// of its FIRST row forbidden (the parent's own choice there, plus what the chain of ancestors created at the same row forbids).
// The parent's optimal assignment restricted to the child's rows is therefore a matching that misses exactly row 0, and the
// parent's dual variables, restricted likewise, are still feasible for the child (lowering cells keeps lx + ly >= C): ONE shortest
// augmenting path from row 0 -- the Hungarian method's last phase -- ends at the child's optimum.  A solve from scratch runs ~25
// dependent trips of ~200 instructions at dimension 15; this one runs as many trips as the path has rows (one to a
// handful) of ~60.
// What it returns is AN optimal assignment of the child's table.  Where several are optimal to within the arithmetic (rounding of
// the duals: a few 1e-16 of the cell magnitude per update) it need not be the one the reference's solver would pick; the k-best
// SCORES, which is all rfsMeasurementLikelihood sums (include/RBPHDFilter.hpp:948-959), do not depend on that choice, the ranked
// ASSIGNMENTS of the FastSLAM path would -- so only the RB-PHD partition sums use it (murty.h, WARM), everything else keeps
// hungarian_wave.
// Reduced indices throughout: row r / column y of the child's table Ct (leading dimension ld, n <= 64 rows and columns), lane r
// holds row r's state, lane y column y's.  In: lxIn (row duals; row 0's is the parent's dual of that row), xyIn (column of row r in
// the parent's matching, -1 for row 0), lyIn / yxIn (column duals -- C[parent's row][y] - lx of that row, computed by the caller
// from the UNCONSTRAINED table -- and the matched row, -1 for the one free column).  Out: xyOut, lxOut.  False if no augmenting path
// is found within n trips (NaN cells): the caller falls back to hungarian_wave.  Ct is only read.
__device__ __forceinline__ bool hungarian_warm_wave(const double *Ct, int ld, int n, double lxIn, int xyIn, double lyIn, int yxIn, int &xyOut, double &lxOut) {
  const int lane = threadIdx.x & 63;
  const bool in = lane < n;
  const double INF = 1.7976931348623157e308;
  double lx = lxIn, ly = lyIn;
  int xy = xyIn, yx = yxIn;
  bool S = (lane == 0), T = false;
  int from = 0;                                    // row through which column `lane` got its present slack
  double slack = in ? (readlane_f64(lx, 0) + ly - Ct[lane]) : INF;
  for (int trip = 0; trip < n; trip++) {
    const double d = wave_min_f64((in && !T) ? slack : INF);
    const unsigned long long at = __ballot(in && !T && slack == d);
    if (at == 0ull) return false;                  // (NaN slacks)
    const int y = __builtin_ctzll(at);
    // dual update: rows of the tree down, its columns up, the others' slacks down -- slack[y] becomes 0
    if (S) lx -= d;
    if (T) ly += d; else slack -= d;
    const int x = __builtin_amdgcn_readlane(yx, y);
    if (x < 0) {                                   // a free column: flip the path y <- from[y] <- (its old column) <- ...
      int yy = y;
      for (int g = 0; g <= n; g++) {
        const int xx = __builtin_amdgcn_readlane(from, yy);
        const int yPrev = __builtin_amdgcn_readlane(xy, xx);
        if (lane == xx) xy = yy;
        if (lane == yy) yx = xx;
        if (xx == 0) break;
        yy = yPrev;
      }
      xyOut = xy;
      lxOut = lx;
      return true;
    }
    if (lane == y) T = true;
    if (lane == x) S = true;
    const double lxx = readlane_f64(lx, x);
    if (in && !T) {
      const double s2 = lxx + ly - Ct[x * ld + lane];
      if (s2 < slack) { slack = s2; from = x; }
    }
  }
  return false;
}
