// hungarian_quad.h -- HungarianMethod::run (reference include/HungarianMethod.hpp:91-587, maximise) with a QUARTER of a
// wavefront (16 lanes = one DPP row) per problem: four independent problems of dimension <= 16 per wave.
//
// Murty's sub-problems at configs[4] have dimension 9-15: hungarian_wave.h spends a whole wave's issue slot per instruction
// on 9-15 live lanes.  Here a quarter q = lane >> 4 owns one problem and the four quarters of a wave run as four virtual
// solvers under ordinary SIMT divergence (a quarter is always active or inactive as a whole, so ballots, DPP row steps and
// LDS broadcasts stay inside it).  Same step sequence, same first / last index choices, same tolerances as hungarian_wave
// (and therefore as the reference): the assignment that comes out is the same one, tie for tie.
//
// What is different in the representation: everything hungarian_wave reads with v_readlane from a data-dependent lane is
// kept REPLICATED in the quarter's 16 lanes instead (identical values computed from ballots): the assignments xy / yx as
// sixteen nibbles of a 64-bit word + a validity mask, the marks S / T / NS and the BFS marks as 16-bit masks.  The only
// per-lane state is what is updated per lane: lx (row ql), ly and slack (column ql); lx has a mirror in LDS (`lxs`) for the
// reads of a data-dependent row.  The BFS queue and the BFS parents are byte arrays in LDS.
//
// The solver is RESUMABLE: hq_start() sets a problem up (offset, greedy start), hq_trip() runs one trip of the main loop
// (steps 2-4 with the whole augmenting-path search if the trip has one) and says whether the problem is finished, so that a
// quarter can take its next problem while the wave's other quarters are in the middle of theirs (murty.h, quad solver).
#pragma once
#include "hungarian_wave.h"

#define HQ_N 16
struct alignas(8) HQScratch {       // per quarter, LDS
  double tile[HQ_N * HQ_N];         // the problem's table, row-major, leading dimension 16; lane ql only touches column ql
  double lxs[HQ_N];                 // mirror of lx
  double tp[HQ_N];                  // (murty: the parent's terms / the child's terms, for the row-ordered sums)
  unsigned char bq[2 * HQ_N];       // BFS queue: node ids (rows 0..n-1, columns n..2n-1)
  unsigned char pys[HQ_N], pxs[HQ_N];   // BFS parents: of column y (a row), of row x (a column + n)
  unsigned char ap[HQ_N], cr[HQ_N], jas[HQ_N];   // (murty: parent assignment, column remap, child's assignment in job columns)
};

struct HQState {                    // per lane; every member but lx / ly / slack is the same in the quarter's 16 lanes
  double lx, ly, slack;
  unsigned long long xyP, yxP;      // nibble x = column of row x / nibble y = row of column y
  unsigned xyV, yxV;                // which nibbles hold an assignment (a clear bit is the reference's -1)
  unsigned S, T, NS;
  int root, n, trips;
  bool pick;
};

// stores of one lane -> loads of the quarter's other lanes (LDS).  The LDS operations of one wave execute in program order,
// so nothing has to be waited for: this only keeps the COMPILER from moving a load of another lane's cell over the store.
// (A workgroup-scope fence here also waits for the wave's global loads -- those of another quarter's next child, murty.h.)
__device__ __forceinline__ void hq_sync() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  asm volatile("" ::: "memory");
}
__device__ __forceinline__ unsigned hq_ballot(bool p, int qshift) { return (unsigned)(__ballot(p) >> qshift) & 0xffffu; }
__device__ __forceinline__ int hq_nib(unsigned long long p, int i) { return (int)((p >> (4 * i)) & 15ull); }
__device__ __forceinline__ unsigned long long hq_nib_set(unsigned long long p, int i, int v) {
  const int s = 4 * i;
  return (p & ~(15ull << s)) | ((unsigned long long)(unsigned)v << s);
}
// min / max over the 16 lanes of a DPP row, result in every lane (exact in any order): xor 1, xor 2, mirror in 8, mirror in 16
__device__ __forceinline__ double hq_min(double v) {
  v = fmin(v, dpp_keep_f64<0xb1, 0xf>(v));    // quad_perm [1,0,3,2]
  v = fmin(v, dpp_keep_f64<0x4e, 0xf>(v));    // quad_perm [2,3,0,1]
  v = fmin(v, dpp_keep_f64<0x141, 0xf>(v));   // row_half_mirror
  v = fmin(v, dpp_keep_f64<0x140, 0xf>(v));   // row_mirror
  return v;
}
__device__ __forceinline__ double hq_max(double v) {
  v = fmax(v, dpp_keep_f64<0xb1, 0xf>(v));
  v = fmax(v, dpp_keep_f64<0x4e, 0xf>(v));
  v = fmax(v, dpp_keep_f64<0x141, 0xf>(v));
  v = fmax(v, dpp_keep_f64<0x140, 0xf>(v));
  return v;
}

// offset subtraction (:128-160) and the greedy start (:162-190) on sc.tile (n x n); all 16 lanes of the quarter call it.
__device__ __forceinline__ void hq_start(HQState &h, HQScratch &sc, const int n, const int ql, const int qshift) {
  const bool in = ql < n;
  double *Ccol = sc.tile + ql;
  h.lx = 0; h.ly = 0; h.slack = 0;
  h.xyP = 0; h.yxP = 0; h.xyV = 0; h.yxV = 0;
  h.S = 0; h.T = 0; h.NS = 0;
  h.root = 0; h.n = n; h.trips = 0; h.pick = true;
  double mn = 0;
  if (in)
    for (int x = 0; x < n; x++) mn = fmin(mn, Ccol[x * HQ_N]);
  const double offset = hq_min(mn);
  if (offset != 0 && in)
    for (int x = 0; x < n; x++) Ccol[x * HQ_N] -= offset;
  for (int x = 0; x < n; x++) {
    const double v = in ? Ccol[x * HQ_N] : -1.0;                 // every real cell is >= 0 now
    const double m = hq_max(v);
    const unsigned eq = hq_ballot(in && v == m, qshift);
    const int yy = 31 - __builtin_clz(eq);                        // `>=` keeps the LAST maximum
    if (ql == x) { h.lx = m; sc.lxs[x] = m; }
    h.xyP = hq_nib_set(h.xyP, x, yy);
    h.xyV |= 1u << x;
    if ((h.yxV >> yy) & 1u) {
      const int x_t = hq_nib(h.yxP, yy);
      hq_sync();
      const double c_t = sc.lxs[x_t];                             // == C[x_t][yy]: yy is row x_t's maximum
      if (m > c_t) {
        h.xyV &= ~(1u << x_t);
        h.yxP = hq_nib_set(h.yxP, yy, x);
      } else {
        h.xyV &= ~(1u << x);
      }
    } else {
      h.yxP = hq_nib_set(h.yxP, yy, x);
      h.yxV |= 1u << yy;
    }
  }
  hq_sync();
}

// One trip of the main loop.  0: go on; 1: finished, the assignment is in h.xyP (all rows valid); 2: no assignment.
__device__ __forceinline__ int hq_trip(HQState &h, HQScratch &sc, const int ql, const int qshift) {
  const int n = h.n;
  const bool in = ql < n;
  const unsigned inMask = (1u << n) - 1u;
  const double *Ccol = sc.tile + ql;
  if (++h.trips > 8 * 64 * 64) return 2;
  if (h.pick) {  // step 2
    h.S = 0; h.T = 0; h.NS = 0;
    const unsigned fr = inMask & ~h.xyV;
    if (fr == 0) return 1;
    h.root = __builtin_ctz(fr);
    h.S = 1u << h.root;
    const double lxr = sc.lxs[h.root];
    bool z = false;
    if (in) {
      h.slack = lxr + h.ly - Ccol[h.root * HQ_N];
      if (fabs(h.slack) < 1e-14) { h.slack = 0; z = true; }
    }
    h.NS = hq_ballot(z, qshift);
  }
  // step 3
  if (((h.NS ^ h.T) & inMask) == 0) {
    const bool inT = (h.T >> ql) & 1u;
    const double a = hq_min((in && !inT) ? h.slack : 1.7976931348623157e308);
    if ((h.S >> ql) & 1u) { h.lx -= a; sc.lxs[ql] = h.lx; }
    if (inT) h.ly += a;
    if (in && !inT) h.slack -= a;
    h.NS |= hq_ballot(in && h.slack == 0, qshift);
    hq_sync();
  }
  // step 4
  const unsigned cand = h.NS & ~h.T & inMask;
  if (cand == 0) return 2;
  const int y = __builtin_ctz(cand);
  if (!((h.yxV >> y) & 1u)) {
    // augmenting path root -> y by breadth-first search over tight edges (:420-523), same visiting order; stops when the
    // target column is discovered (hungarian_wave.h explains why that gives the same xy / yx)
    const int root = h.root;
    int qh = 0, qt = 1;
    unsigned xq = 1u << root, yq = 0;
    if (ql == 0) sc.bq[0] = (unsigned char)root;
    // lane ql: which rows are tight with column ql (lx / ly do not change during the search) -- 2 n independent LDS reads up
    // front instead of two dependent ones per dequeued node
    unsigned tm = 0;
    if (in)
      for (int t = 0; t < n; t++) tm |= (fabs(sc.lxs[t] + h.ly - Ccol[t * HQ_N]) < 1e-12) ? (1u << t) : 0u;
    hq_sync();
    bool found = false;
    while (qh < qt) {
      int t = sc.bq[qh];
      qh++;
      if (t < n) {
        const int xyt = ((h.xyV >> t) & 1u) ? hq_nib(h.xyP, t) : -1;
        const bool push = ((tm >> t) & 1u) && !((yq >> ql) & 1u) && xyt != ql;
        const unsigned pm = hq_ballot(push, qshift);
        if (push) {
          sc.pys[ql] = (unsigned char)t;
          sc.bq[qt + __popc(pm & ((1u << ql) - 1u))] = (unsigned char)(ql + n);   // enqueued in ascending column order
        }
        yq |= pm;
        if ((pm >> y) & 1u) { found = true; break; }
        qt += __popc(pm);
        hq_sync();
      } else {
        t -= n;
        if ((h.yxV >> t) & 1u) {
          const int x = hq_nib(h.yxP, t);                        // the only row that can pass `yx[t] == x`
          const unsigned tight = hq_ballot((tm >> x) & 1u, qshift);   // bit t: lane t's verdict
          if (((tight >> t) & 1u) && ((h.S >> x) & 1u) && !((xq >> x) & 1u)) {
            xq |= 1u << x;
            if (ql == 0) { sc.pxs[x] = (unsigned char)(t + n); sc.bq[qt] = (unsigned char)x; }
            qt++;
            hq_sync();
          }
        }
      }
    }
    if (!found) return 2;
    hq_sync();
    int t = y + n;
    while (t != root) {
      if (t >= n) {
        const int xt = sc.pys[t - n];
        h.xyP = hq_nib_set(h.xyP, xt, t - n); h.xyV |= 1u << xt;
        h.yxP = hq_nib_set(h.yxP, t - n, xt); h.yxV |= 1u << (t - n);
        t = xt;
      } else {
        t = sc.pxs[t];
      }
    }
    h.pick = true;
  } else {
    const int x_t = hq_nib(h.yxP, y);
    h.S |= 1u << x_t;
    h.T |= 1u << y;
    const double lxt = sc.lxs[x_t];
    bool z = false;
    if (in) {
      const double d = lxt + h.ly - Ccol[x_t * HQ_N];
      z = fabs(d) < 1e-14;
      if (d < h.slack) h.slack = d;
    }
    h.NS |= hq_ballot(z, qshift);
    h.pick = false;
  }
  return 0;
}
