// murty_quad_solver.h -- TEST SUPPORT (not part of the shipped library): the solver wave of the quarter-wave Hungarian variant
// (hungarian_quad.h), included by csrc/murty.h only in builds with -DMURTY_QUAD=1 (rfs-slam_amd/build.py: build_quad_variant).
// Measured slower than the shipped one-wave solvers; kept because the bit-identity test
// (tests/test_gpu_parity.py::test_quarter_wave_hungarian_variant_returns_the_same_bits) exercises the search with a second solver.
#pragma once
#include "hungarian_quad.h"
// ---- quad solver wave: four 16-lane virtual solvers (hungarian_quad.h), each with its own mailbox ----------------------
// Quarter q of solver wave w serves mailbox v = 4 (w - 1) + q + 1.  A quarter is a little state machine -- idle (look at the
// mailbox once), solving (one trip of the Hungarian main loop per turn) -- so that a quarter starts its next child while the
// other quarters are in the middle of theirs; the wave leaves when the search has ended and every quarter is idle.
// The child (murty_solve_child / murty_child_wave above, leading dimension 16): table, negative constraints, solver, score
// terms added in row order, assignment in job columns -- the same arithmetic in the same order.
struct MurtyQuadTask {
  int t, e, X, c, nn, nFree, colRemap, aPar, ppX;
  unsigned excl;
  double fixedScore;
};
// first half: the three loads that depend on nothing but the node (issued, not waited for: the wave's other quarters run
// their turn in between)
__device__ __forceinline__ void murty_quad_fetch(MurtyQuadTask &k, const int n, MurtyArena &A, const int ql) {
  k.ppX = A.nodeId[k.X];
  k.aPar = (ql < n) ? (int)A.nodeA[(size_t)k.X * MURTY_N + ql] : 0;
  k.excl = A.nodeExcl[k.X];
}
// second half: LDS and registers only.  sC: the job's table (n x n) in LDS.  The negative constraints (src/MurtyAlgorithm.cpp:247-265)
// all fall into the child's first row: child c > 0 must not repeat the node's own choice for that row; child 0 -- whose
// first row is the row the node itself was created at -- must not repeat the node's, its parent's, and the choices of the
// ancestors above for as long as they were created at that row too: the set the search keeps per node (nodeExcl).  Lane dj of the constraint walk = the lane whose free column IS the excluded one; the dummy-range rule on the
// REDUCED column index as there.
__device__ __forceinline__ bool murty_quad_begin(MurtyQuadTask &k, HQState &h, HQScratch &sc, const double *sC, const int n, const int realNC, const int ql,
                                                 const int qshift) {
  const double bigNumber = 10000.0;
  const int nn = k.ppX + k.c, nFree = n - nn;
  const int aPar = k.aPar;
  const double termPar = (ql < n) ? sC[ql * n + aPar] : 0.0;
  sc.ap[ql] = (unsigned char)aPar;
  sc.tp[ql] = termPar;
  hq_sync();
  double fixedScore = 0;
  unsigned used = 0;
  for (int r = 0; r < nn; r++) { fixedScore += sc.tp[r]; used |= 1u << sc.ap[r]; }   // rows 0..nn-1 fixed to the parent's choice
  const unsigned freeCols = ((1u << n) - 1u) & ~used;
  const int colRemap = (ql < nFree) ? murty_kth_bit((unsigned long long)freeCols, ql) : 0;
  k.nn = nn; k.nFree = nFree; k.colRemap = colRemap; k.fixedScore = fixedScore;
  const unsigned excl = (k.c == 0) ? k.excl : (1u << sc.ap[nn]);
  const bool hit = ql < nFree && ((excl >> colRemap) & 1u);
  const bool dummy = hq_ballot(hit && ql >= realNC, qshift) != 0;
  if (ql < nFree) {
    for (int r = 0; r < nFree; r++) {
      double v = sC[(nn + r) * n + colRemap];
      if (r == 0 && (hit || (dummy && ql >= realNC))) v = -bigNumber;
      sc.tile[r * HQ_N + ql] = v;
    }
  }
  hq_sync();
  if (hq_ballot(ql < nFree && sc.tile[ql] != -bigNumber, qshift) == 0) return false;   // the constraint row is reduced row 0
  hq_start(h, sc, nFree, ql, qshift);
  return true;
}
// the solved (or failed) child -> table slot e, mailbox v
__device__ __forceinline__ void murty_quad_finish(const MurtyQuadTask &k, const HQState &h, HQScratch &sc, const double *sC, const int n, const bool okH,
                                                  MurtySpec *spec, const int v, const int ql) {
  double sAcc = 0;
  int aNew = k.aPar;
  if (okH) {
    const int aTmp = hq_nib(h.xyP, ql);
    sc.cr[ql] = (unsigned char)k.colRemap;
    hq_sync();
    const int ja = (ql < k.nFree) ? (int)sc.cr[aTmp] : 0;
    const double term = (ql < k.nFree) ? sC[(k.nn + ql) * n + ja] : 0.0;
    sc.tp[ql] = term;
    sc.jas[ql] = (unsigned char)ja;
    hq_sync();
    for (int r = 0; r < k.nFree; r++) sAcc += sc.tp[r];
    sAcc += k.fixedScore;
    if (ql >= k.nn) aNew = sc.jas[ql - k.nn];
  }
  spec->a[k.e][ql] = (unsigned char)aNew;
  if (ql == 0) { spec->score[k.e] = sAcc; spec->pushed[k.e] = okH ? 1 : 0; }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  if (ql == 0) { murty_flag_store(&spec->ready[k.e], 1); murty_flag_store(&spec->doneSeq[v], k.t); }
}
__device__ __forceinline__ void murty_quad_solver_wave(const double *C, const int n, const int realNC, MurtyArena &A, const int wave, MurtySpec *spec,
                                                       HQScratch *quadScratch, double *sC) {
  const int lane = threadIdx.x & 63, q = lane >> 4, ql = lane & 15, qshift = lane & 48;
  const int v = 4 * (wave - 1) + q + 1;
  HQScratch &sc = quadScratch[4 * (wave - 1) + q];
  // the job's table -> LDS (after the root's solve, which rewrites it in place; every solver wave writes the same values and
  // reads them after its own stores)
  for (int t = lane; t < n * n; t += 64) sC[t] = C[t];
  hq_sync();
  HQState h;
  MurtyQuadTask k;
  k.t = 0;
  int state = 0, seen = 0;   // 0 idle, 1 solving, 2 gone
#ifdef RFS_PROFILE
  long long pc[6] = {0, 0, 0, 0, 0, 0};   // cycles: begin, trips, finish; counts: turns, trips of this quarter, children of this quarter
#define HQ_T0 const long long hqT0 = (long long)__builtin_readcyclecounter()
#define HQ_ADD(i) pc[i] += (long long)__builtin_readcyclecounter() - hqT0
#else
#define HQ_T0 do { } while (0)
#define HQ_ADD(i) do { } while (0)
#endif
  for (;;) {
#ifdef RFS_PROFILE
    pc[3]++;
#endif
    if (state == 0) {
      const int t = murty_flag_load(&spec->taskSeq[v]);
      if (t != seen) {
        k.t = t;
        k.e = spec->taskSlot[v];
        k.X = spec->taskNode[v];
        k.c = spec->taskC[v];
        murty_quad_fetch(k, n, A, ql);
        state = 3;
      } else if (murty_flag_load(&spec->quit)) {
        state = 2;
      }
    } else if (state == 3) {
      HQ_T0;
      if (murty_quad_begin(k, h, sc, sC, n, realNC, ql, qshift)) state = 1;
      else { murty_quad_finish(k, h, sc, sC, n, false, spec, v, ql); seen = k.t; state = 0; }
      HQ_ADD(0);
#ifdef RFS_PROFILE
      pc[5]++;
#endif
    } else if (state == 1) {
      HQ_T0;
      const int r = hq_trip(h, sc, ql, qshift);
      HQ_ADD(1);
#ifdef RFS_PROFILE
      pc[4]++;
#endif
      if (r != 0) { HQ_T0; murty_quad_finish(k, h, sc, sC, n, r == 1, spec, v, ql); seen = k.t; state = 0; HQ_ADD(2); }
    }
    if (__ballot(state != 2) == 0ull) break;
    if (__ballot(state == 1 || state == 3) == 0ull) __builtin_amdgcn_s_sleep(MURTY_SOLVER_SLEEP);
  }
#ifdef RFS_PROFILE
  if (ql == 0 && (blockIdx.x & 255) == 7)
    printf("quad solver block %d wave %d quarter %d (n %d): children %lld, trips %lld, turns %lld; cycles begin %lld, trips %lld, finish %lld\n", (int)blockIdx.x, wave, q, n,
           pc[5], pc[4], pc[3], pc[0], pc[1], pc[2]);
#endif
}

