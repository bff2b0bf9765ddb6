// murty.h -- Murty k-best (k <= 200) sums for measurement-likelihood partitions with nR + nC > 8:
// the loop at reference include/RBPHDFilter.hpp:920-959 over Murty::findNextBest
// (src/MurtyAlgorithm.cpp:141-336) with HungarianMethod::run (include/HungarianMethod.hpp:91-587) as the
// inner solver.  The truncation to the 200 best assignments is observable (SURVEY §7 hard part 1), so this
// follows the reference's algorithm step for step: same partition tree, same negative-constraint walk (incl.
// the dummy-column test on the REDUCED column index, :256-262), same tolerances (1e-14 / 1e-12) and in-place
// offset add/subtract in the Hungarian solver, same binary-heap discipline as std::priority_queue.
//
// Jobs are produced by phd_weight_multifeature_kernel (weighting.h) into a device queue; murty_jobs_kernel consumes
// them -- one workgroup of MURTY_JOB_WAVES (four) wavefronts per job: the tree search stays serial, the Hungarian solver inside it
// (hungarian_wave.h) and the sub-problem / assignment bookkeeping run across the lanes, the children of an expansion across
// the waves -- and multiplies each job's partition likelihood into its
// particle's weight in partition order.  Node pool, heap and root table live in HBM (per-job arena), sub-problem tables
// in an LDS tile.  (The one-thread form of the solver that round 1 still carried for FastSLAM's small blocks is gone: every caller uses hungarian_wave.)
#pragma once
#include "common.h"
#include "weighting.h"
#include "hungarian_wave.h"

#define MURTY_N 64             /* max extended dimension nR + nC handled on the device */
#define MURTY_KBEST 200
#define MURTY_MAX_NODES 6401   /* 1 root + <= 200 expansions x <= 32 children */
#define MURTY_WARM_N 16        /* jobs up to this extended dimension (the small form) keep their nodes' dual variables */
#ifndef MURTY_WARM
#define MURTY_WARM 1           /* children of the RB-PHD partition sums start from their parent's duals: one augmentation instead of a solve (hungarian_wave.h) */
#endif
#ifndef MURTY_JOB_WAVES
// measured at configs[4] (1918 jobs of dimension 9-15), barrier form: 1 wave 26.8 ms, 2: 16.5, 3: 15.4, 4: 15.1; search wave + solvers
// (murty_kbest_async) at the compiler's 92 VGPRs (5 waves per SIMD): 3: 8.9, 4: 6.3, 5: 7.8, 6: 8.9, 8: 8.4; capped at 64 VGPRs
// (MURTY_WAVES_PER_EU 8: 100 B of scratch per lane, 1280 six-wave workgroups on the GPU at once): 4: 6.6, 5: 5.7, 6: 5.5, 7: 8.6, 8: 5.6
#define MURTY_JOB_WAVES 5   // round 6, children solved by ONE augmentation from their parent's duals (MURTY_WARM): the search wave's own bookkeeping is what a pop
                            // costs now, and solver waves that mostly poll their mailboxes only take issue slots from it -- configs[4], kernel ms (profiles/r06d_*):
                            // 8 waves 1.80 (2.10 with the solves from scratch), 6: 1.73, 5: 1.61, 4: 1.52; with two peeked heap positions instead of five 4: 1.43,
                            // 5: 1.35, 6: 1.39; then with the open nodes in an unsorted array scanned by the wave (MURTY_ARGQ) 5: 1.07, 4: 1.20, 6: 1.16, 3: 1.51.  (round 5, solves from scratch: 6: 2.96 ms, 8: 2.50, 10: 3.1, 12: 5.3; 16 table slots + peeks: 8: 2.16)
#endif
#define MURTY_CT_WAVES (MURTY_JOB_WAVES > 4 ? MURTY_JOB_WAVES : 4)   /* (the multi-hypothesis FastSLAM search uses up to four waves on the same arena) */

#if defined(MURTY_WARM_CHECK)
__device__ unsigned long long g_murtyWarmChecks = 0ull, g_murtyWarmBad = 0ull;
#endif
struct MurtyScratch {
  unsigned char *arena;   // [nArenas][jobBytes]: scratch that is live only while a workgroup works on a job -> one per WORKGROUP
  size_t jobBytes;
  int nArenas;            // = the largest grid murty_jobs_kernel is launched with (r2: one per queue slot, 5 GB at 8192 slots)
};

struct MurtyArena {
  double *Ct;        // [N*N]
  double *lx, *ly, *slack;   // [N]
  int *xy, *yx, *p, *queue;  // [N],[N],[2N],[2N]
  unsigned char *S, *T, *NS, *xq, *yq;  // [N]
  // node pool
  double *nodeScore;       // [MAX_NODES]
  short *nodeParent;       // [MAX_NODES]
  unsigned char *nodeId;   // [MAX_NODES]
  unsigned char *nodeA;    // [MAX_NODES][N]
  short *heap;             // [MAX_NODES]
  unsigned short *nodeExcl;   // [MAX_NODES] (small form, n <= 16): the columns child 0 of the node must not take in its first row
  double *nodeLx;             // [MAX_NODES][MURTY_WARM_N] (small form, round 6): the node's row duals for the job's table -- where its children's solves start (hungarian_warm_wave)
};

__host__ __device__ inline size_t murty_job_bytes() {
  size_t b = 0;
  b += (size_t)MURTY_CT_WAVES * MURTY_N * MURTY_N * 8;  // Ct, one per wave of the job's workgroup
  b += 3 * MURTY_N * 8;                    // lx ly slack
  b += (1 + 1 + 2 + 2) * MURTY_N * 4;      // xy yx p queue
  b += 5 * MURTY_N;                        // flags
  b += (size_t)MURTY_MAX_NODES * 8;        // score
  b += (size_t)MURTY_MAX_NODES * 2;        // parent
  b += (size_t)MURTY_MAX_NODES;            // id
  b += (size_t)MURTY_MAX_NODES * MURTY_N;  // assignments
  b += (size_t)MURTY_MAX_NODES * 2;        // heap
  b += (size_t)MURTY_MAX_NODES * 2 + 2;    // nodeExcl
  b += (size_t)MURTY_MAX_NODES * MURTY_WARM_N * 8 + 8;   // nodeLx
  return (b + 63) & ~(size_t)63;
}

__device__ inline void murty_carve(unsigned char *base, MurtyArena &A) {
  unsigned char *p = base;
  A.Ct = (double *)p; p += (size_t)MURTY_CT_WAVES * MURTY_N * MURTY_N * 8;
  A.lx = (double *)p; p += MURTY_N * 8;
  A.ly = (double *)p; p += MURTY_N * 8;
  A.slack = (double *)p; p += MURTY_N * 8;
  A.nodeScore = (double *)p; p += (size_t)MURTY_MAX_NODES * 8;
  A.xy = (int *)p; p += MURTY_N * 4;
  A.yx = (int *)p; p += MURTY_N * 4;
  A.p = (int *)p; p += 2 * MURTY_N * 4;
  A.queue = (int *)p; p += 2 * MURTY_N * 4;
  A.nodeParent = (short *)p; p += (size_t)MURTY_MAX_NODES * 2;
  A.heap = (short *)p; p += (size_t)MURTY_MAX_NODES * 2;
  A.S = p; p += MURTY_N;
  A.T = p; p += MURTY_N;
  A.NS = p; p += MURTY_N;
  A.xq = p; p += MURTY_N;
  A.yq = p; p += MURTY_N;
  A.nodeId = p; p += MURTY_MAX_NODES;
  A.nodeA = p; p += (size_t)MURTY_MAX_NODES * MURTY_N;
  A.nodeExcl = (unsigned short *)(((size_t)p + 1) & ~(size_t)1);
  A.nodeLx = (double *)(((size_t)(A.nodeExcl + MURTY_MAX_NODES) + 7) & ~(size_t)7);
}

// std::priority_queue<MurtyNode*, vector, MurtyNodeCompare> == libstdc++ push_heap / pop_heap on scores.
__device__ inline void heap_push(short *h, int &len, short v, const double *score) {
  int hole = len++;
  int parent = (hole - 1) / 2;
  while (hole > 0 && score[h[parent]] < score[v]) {
    h[hole] = h[parent];
    hole = parent;
    parent = (hole - 1) / 2;
  }
  h[hole] = v;
}
__device__ inline short heap_pop(short *h, int &len, const double *score) {
  const short top = h[0];
  len--;
  if (len == 0) return top;
  const short value = h[len];
  int hole = 0, second = 0;
  while (second < (len - 1) / 2) {
    second = 2 * (second + 1);
    if (score[h[second]] < score[h[second - 1]]) second--;
    h[hole] = h[second];
    hole = second;
  }
  if ((len & 1) == 0 && second == (len - 2) / 2) {
    second = 2 * (second + 1);
    h[hole] = h[second - 1];
    hole = second - 1;
  }
  int parent = (hole - 1) / 2;
  while (hole > 0 && score[h[parent]] < score[value]) {
    h[hole] = h[parent];
    hole = parent;
    parent = (hole - 1) / 2;
  }
  h[hole] = value;
  return top;
}

// The same heap with its first MURTY_HEAP_LDS positions (node id + a copy of the node's score) in LDS, the rest in the arena as
// above: libstdc++'s pop_heap walks the hole down to a leaf before it pushes the last element back up -- every level two ids and
// then their two scores, dependent loads -- and with everything in global memory that walk was most of the ~14 k cycles a pop
// cost the searching wave beside its solves (configs[4], dimension 12).  Same comparisons in the same order.
#ifndef MURTY_HEAP_LDS
#define MURTY_HEAP_LDS 512
#endif
struct MurtyHeap {
  short *lid;            // [MURTY_HEAP_LDS] LDS
  double *lsc;           // [MURTY_HEAP_LDS] LDS
  short *gid;            // arena (positions >= MURTY_HEAP_LDS)
  const double *gscore;  // arena: score by node id
};
__device__ __forceinline__ short mheap_id(const MurtyHeap &H, int pos) { return pos < MURTY_HEAP_LDS ? H.lid[pos] : H.gid[pos]; }
__device__ __forceinline__ double mheap_sc(const MurtyHeap &H, int pos) { return pos < MURTY_HEAP_LDS ? H.lsc[pos] : H.gscore[H.gid[pos]]; }
__device__ __forceinline__ void mheap_set(const MurtyHeap &H, int pos, short id, double sc) {
  if (pos < MURTY_HEAP_LDS) { H.lid[pos] = id; H.lsc[pos] = sc; }
  else H.gid[pos] = id;
}
__device__ inline void mheap_push(const MurtyHeap &H, int &len, short v, double sv) {   // sv == gscore[v], already stored there
  int hole = len++;
  int parent = (hole - 1) / 2;
  while (hole > 0 && mheap_sc(H, parent) < sv) {
    mheap_set(H, hole, mheap_id(H, parent), mheap_sc(H, parent));
    hole = parent;
    parent = (hole - 1) / 2;
  }
  mheap_set(H, hole, v, sv);
}
__device__ inline short mheap_pop(const MurtyHeap &H, int &len) {
  const short top = mheap_id(H, 0);
  len--;
  if (len == 0) return top;
  const short value = mheap_id(H, len);
  const double vs = mheap_sc(H, len);
  int hole = 0, second = 0;
  while (second < (len - 1) / 2) {
    second = 2 * (second + 1);
    if (mheap_sc(H, second) < mheap_sc(H, second - 1)) second--;
    mheap_set(H, hole, mheap_id(H, second), mheap_sc(H, second));
    hole = second;
  }
  if ((len & 1) == 0 && second == (len - 2) / 2) {
    second = 2 * (second + 1);
    mheap_set(H, hole, mheap_id(H, second - 1), mheap_sc(H, second - 1));
    hole = second - 1;
  }
  int parent = (hole - 1) / 2;
  while (hole > 0 && mheap_sc(H, parent) < vs) {
    mheap_set(H, hole, mheap_id(H, parent), mheap_sc(H, parent));
    hole = parent;
    parent = (hole - 1) / 2;
  }
  mheap_set(H, hole, value, vs);
  return top;
}

// ---- the same store as an UNSORTED array scanned by the whole wave (round 6, the RB-PHD partition sums of the small form) ----
// With the children's solves down to one augmentation (MURTY_WARM) a pop's cost was lane 0's walk through the binary heap: pop_heap's
// sift-down and push_heap's sift-up are chains of dependent LDS reads that one lane runs while 63 idle (3.6 k + 2.9 k cycles of a
// 14 k-cycle pop on an idle GPU, five to ten times that when every SIMD holds six such waves; profiles/r06i_*).  Here the entries
// lie in the same arrays in arrival order, a push appends, a pop removes by moving the last entry into the hole, and the best three
// come out of ONE pass in which every lane looks at len / 64 entries and the wave reduces (DPP maxima, no memory in the chain).
// Which of several entries with EXACTLY equal scores comes out first differs from std::priority_queue's order: the partition sum
// adds equal terms then, in a different order -- the same sum (the ranked ASSIGNMENTS would differ: FastSLAM keeps the heap).
struct MqTop { int id[3]; int pos0; double sc0; };
__device__ __forceinline__ void mq_push(const MurtyHeap &H, int &len, int id, double sc) {   // lane 0's stores; len is uniform
  if ((threadIdx.x & 63) == 0) mheap_set(H, len, (short)id, sc);
  len++;
}
__device__ __forceinline__ void mq_remove(const MurtyHeap &H, int &len, int pos) {          // lane 0's stores
  len--;
  if ((threadIdx.x & 63) == 0 && pos != len) mheap_set(H, pos, mheap_id(H, len), mheap_sc(H, len));
}
// the three best entries (ids; -1 where there are fewer), the best one's position and score; all lanes call
__device__ __forceinline__ void mq_top3(const MurtyHeap &H, int len, MqTop &T) {
  const int lane = threadIdx.x & 63;
  const double NONE = -1.7976931348623157e308;
  double b0 = NONE, b1 = NONE, b2 = NONE;     // this lane's three best, descending (ties: the earlier position first)
  int i0 = -1, i1 = -1, i2 = -1, p0 = -1;
  for (int pos = lane; pos < len; pos += 64) {
    const double sc = mheap_sc(H, pos);
    const int id = mheap_id(H, pos);
    if (sc > b0) { b2 = b1; i2 = i1; b1 = b0; i1 = i0; b0 = sc; i0 = id; p0 = pos; }
    else if (sc > b1) { b2 = b1; i2 = i1; b1 = sc; i1 = id; }
    else if (sc > b2) { b2 = sc; i2 = id; }
  }
#pragma unroll
  for (int r = 0; r < 3; r++) {
    const double m = wave_max_f64(i0 >= 0 ? b0 : NONE);
    const unsigned long long own = __ballot(i0 >= 0 && b0 == m);
    if (own == 0ull) { T.id[r] = -1; if (r == 0) { T.pos0 = -1; T.sc0 = NONE; } continue; }
    const int l = __builtin_ctzll(own);
    T.id[r] = __builtin_amdgcn_readlane(i0, l);
    if (r == 0) { T.pos0 = __builtin_amdgcn_readlane(p0, l); T.sc0 = m; }
    if (lane == l) { b0 = b1; i0 = i1; b1 = b2; i1 = i2; b2 = NONE; i2 = -1; }
  }
}

// ---- one WAVEFRONT per Murty problem (hungarian_wave.h as the inner solver) --------------------------------------------
// Row r's assignment lives on lane r, sub-problem tables are built a row per step with lane c writing column c, the node
// pool and the heap are lane 0's (scalars broadcast with readfirstlane).  Same partition tree, heap discipline,
// negative-constraint walk and score arithmetic as the serial restatement above (and the oracle's).

// k-th (0-based) set bit of m
__device__ __forceinline__ int murty_kth_bit(unsigned long long m, int k) {
  for (int i = 0; i < k; i++) m &= m - 1ull;
  return m ? __builtin_ctzll(m) : 0;
}
__device__ __forceinline__ void murty_publish() {  // stores of one lane -> loads of the wave's other lanes (global memory)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// One child of a Murty expansion: rows nn.. of C restricted to the free columns -> Ct (lane c writes column c), the negative
// constraints of the partition chain (src/MurtyAlgorithm.cpp:247-265, incl. the dummy-column range test on the REDUCED
// column index), then the solver.  False when the constraint row has no possibility left or the solver finds no
// assignment.  LDT = leading dimension of Ct.  realNC == n switches the dummy-range rule off (FastSLAM's use).
// MASK (jobs of extended dimension <= 16, murty_kbest_async's SMALL form): the walk over the partition chain is replaced by
// its result, kept per node by the search -- all of a child's negative constraints fall into its FIRST row (every chain step
// has curPart == nn), so they are a set of job columns: `excl`.  Lane dj of the walk = the lane whose free column is the
// excluded one (an excluded column is never one of the child's fixed columns: it is an ancestor's choice for row nn, and the
// ancestor shares the child's rows 0..nn-1).
// WARM (round 6; the RB-PHD partition sums of the small form): aPar / lxPar = the parent's assignment and row duals, job row r on lane
// r.  The child's solve is ONE augmentation from them (hungarian_warm_wave); lxNew (reduced row r on lane r) = the child's own duals.
template <int LDT, bool MASK = false, bool WARM = false>
__device__ __forceinline__ bool murty_child_wave(double *Ct, const double *C, int n, int nn, int nFree, int pn, int parent, int colRemap,
                                                 unsigned long long freeCols, int realNC, MurtyArena &A, int &aTmp, unsigned char *queue,
                                                 long long *prof, const unsigned excl = 0u, const int aPar = 0, const double lxPar = 0.0, double *lxNew = nullptr) {
  const double bigNumber = 10000.0;
  const int lane = threadIdx.x & 63;
  if (lane < nFree) {
#pragma unroll 4
    for (int r = 0; r < nFree; r++) Ct[r * LDT + lane] = C[(nn + r) * n + colRemap];
  }
  if constexpr (MASK) {
    const bool hit = lane < nFree && ((excl >> colRemap) & 1u);
    const bool dummy = __ballot(hit && lane >= realNC) != 0ull;
    if (hit || (dummy && lane >= realNC && lane < nFree)) Ct[lane] = -bigNumber;
  } else {
    int current = pn, curPart = nn;
    for (;;) {  // the walk is uniform (same loads on every lane)
      const int next = (current == pn) ? parent : (int)A.nodeParent[current];
      const int naCol = A.nodeA[(size_t)next * MURTY_N + curPart];
      const int di = curPart - nn;
      const int dj = __popcll(freeCols & ((1ull << naCol) - 1ull));
      if (lane == dj || (dj >= realNC && lane >= realNC && lane < nFree)) Ct[di * LDT + lane] = -bigNumber;
      current = next;
      if (current == 0) break;
      curPart = A.nodeId[current];
      if (curPart < nn) break;
    }
  }
  if (__ballot(lane < nFree && Ct[lane] != -bigNumber) == 0) return false;   // the constraint row is reduced row 0
  double s = 0;
#ifdef RFS_PROFILE
  const long long tH = (long long)__builtin_readcyclecounter();
#endif
  if constexpr (WARM) {
    // the parent's matching and duals in the child's reduced indices
    const bool inr = lane < nFree;
    const int jr = inr ? nn + lane : 0;                                  // job row of reduced row `lane`
    const double lxr = __shfl(lxPar, jr, 64);
    const int jcOfRow = __shfl(aPar, jr, 64);                            // the parent's column of that row (a free column: the row is not fixed)
    const int xyIn = (inr && lane > 0) ? __popcll(freeCols & ((1ull << jcOfRow) - 1ull)) : -1;
    // column side: the row the parent gave the job column to (the inverse of aPar, by a forward permute), its dual from tightness
    const int invA = __builtin_amdgcn_ds_permute(((lane < n) ? aPar : lane) << 2, lane);   // lane c: the row r with aPar[r] == c
    const int rowOfCol = __shfl(invA, inr ? colRemap : 0, 64);
    const double lxOfCol = __shfl(lxPar, rowOfCol, 64);
    const double lyIn = inr ? C[rowOfCol * n + colRemap] - lxOfCol : 0.0;  // (the UNCONSTRAINED cell: the parent's edge was tight)
    const int yxIn = (inr && rowOfCol > nn) ? rowOfCol - nn : -1;        // the column row nn held is the free one
    int aW = 0;
    double lxW = 0.0;
    bool okW = hungarian_warm_wave(Ct, LDT, nFree, lxr, xyIn, lyIn, yxIn, aW, lxW);
#if defined(MURTY_WARM_CHECK)
    {   // test builds (tools/murty_warm_check.py): the augmentation's assignment must be worth what the solve from scratch finds
      int aC = 0;
      double dummy = 0;
      const bool okC = hungarian_wave<true>(Ct, LDT, nFree, aC, &dummy, queue, nullptr);
      // (that solver has taken its offset out of the tile -- every cell moved by the same amount: both assignments are priced on the tile as it is now)
      const double tw = (okW && inr) ? Ct[lane * LDT + aW] : 0.0, tc = (okC && inr) ? Ct[lane * LDT + aC] : 0.0;
      double cw = 0, cc = 0;
      for (int r = 0; r < nFree; r++) { cw += readlane_f64(tw, r); cc += readlane_f64(tc, r); }
      if (lane == 0) {
        atomicAdd(&g_murtyWarmChecks, 1ull);
        if (okW != okC || fabs(cw - cc) > 1e-9 * (1.0 + fabs(cc))) {
          if (atomicAdd(&g_murtyWarmBad, 1ull) < 8ull) printf("MURTY WARM MISMATCH n %d nn %d nFree %d: warm %d %.17g, from scratch %d %.17g\n", n, nn, nFree, (int)okW, cw, (int)okC, cc);
        }
      }
    }
#endif
    if (okW) {
      aTmp = aW;
      if (lxNew) *lxNew = lxW;
#ifdef RFS_PROFILE
      if (prof) { prof[1] += (long long)__builtin_readcyclecounter() - tH; prof[2]++; }
#endif
      return true;
    }
    // (no path within nFree trips -- NaN cells: the solve from scratch decides, as before)
  }
  double lxCold = 0.0;
  const bool okH = hungarian_wave<true>(Ct, LDT, nFree, aTmp, &s, queue, prof, WARM ? &lxCold : nullptr);
  if (WARM && lxNew) *lxNew = lxCold;
#ifdef RFS_PROFILE
  if (prof) { prof[1] += (long long)__builtin_readcyclecounter() - tH; prof[2]++; }
#endif
  return okH;
}

// root: the solver on the full table (:147-158); node 0.  False when there is no assignment.
__device__ __forceinline__ bool murty_root_wave(double *C, int n, MurtyArena &A, int &a0, double &s, unsigned char *queue, double *lxRaw = nullptr) {
  const int lane = threadIdx.x & 63;
  if (!hungarian_wave(C, n, n, a0, &s, queue, nullptr, lxRaw)) return false;
  if (lane < n) A.nodeA[lane] = (unsigned char)a0;
  if (lane == 0) {
    A.nodeId[0] = 0;
    A.nodeParent[0] = -1;
    A.nodeScore[0] = s;
    int hl = 0;
    heap_push(A.heap, hl, 0, A.nodeScore);
  }
  murty_publish();   // node 0's assignment is read back lane-crossed (the caller's policies, the first expansion)
  return true;
}
#ifndef MURTY_LDS_N
#define MURTY_LDS_N 20   // sub-problems up to this dimension are solved in a 3.1 KB LDS tile per wave (larger ones in the job's arena); 24 until the heap moved into LDS
#endif

// Child c of node `par` (created at partition ppar): sub-problem, constraints, solution; aPar / termPar: the parent's
// assignment and its terms, row r on lane r.  pn: the child's node number (only compared against in the constraint walk; a
// child solved ahead of its parent's pop has none yet).  Out: pushed (a solution exists), its score and assignment.
// MASK: `exclNode` = what child 0 of `par` must not take in its first row (the search's per-node set); a child c > 0 must not
// take the parent's own choice for that row.
// (Measured and dropped, r04 -- profiles/r04f_ab_murty_solver_as_call.txt: this body as a real call (__noinline__), so that the solver
//  waves' path would get a register allocation of its own under the 64-VGPR cap: 360 B of stack per lane instead of 160 B of
//  spills, configs[4] 5.18 -> 8.5 ms per update.)
// WARM: lxPar = the parent's row duals (job row r on lane r); lxNew (may be null) receives the child's, in the same indexing.
template <int LDSN, bool MASK = false, bool WARM = false>
__device__ __forceinline__ void murty_solve_child(double *myTile, const double *C, const int n, const int realNC, MurtyArena &A, const int wave, const int par,
                                                  const int ppar, const int c, const int pn, const int aPar, const double termPar, bool &pushed,
                                                  double &sAcc, int &aNew, long long *prof, const unsigned exclNode = 0u, const double lxPar = 0.0,
                                                  double *lxNew = nullptr) {
  const int lane = threadIdx.x & 63;
  const int nn = ppar + c;
  double fixedScore = 0;
  for (int r = 0; r < nn; r++) fixedScore += readlane_f64(termPar, r);   // rows 0..nn-1 fixed to the parent's choice
  const unsigned long long usedCols = wave_or_u64((lane < nn) ? (1ull << aPar) : 0ull);
  const unsigned long long freeCols = ((n >= 64) ? ~0ull : ((1ull << n) - 1ull)) & ~usedCols;
  const int nFree = n - nn;
  const int colRemap = (lane < nFree) ? murty_kth_bit(freeCols, lane) : 0;
  pushed = false;
  sAcc = 0;
  aNew = aPar;
  int aTmp = 0;
  unsigned excl = 0u;
  if constexpr (MASK) excl = (c == 0) ? exclNode : (1u << __builtin_amdgcn_readlane(aPar, nn));
  double lxRed = 0.0;     // the child's row duals, reduced row r on lane r
  const bool okH = (nFree <= LDSN)
                       ? murty_child_wave<LDSN, MASK, WARM>(myTile, C, n, nn, nFree, pn, par, colRemap, freeCols, realNC, A, aTmp, nullptr, prof, excl, aPar, lxPar, &lxRed)
                       : murty_child_wave<MURTY_N, MASK, WARM>(A.Ct + (size_t)wave * MURTY_N * MURTY_N, C, n, nn, nFree, pn, par, colRemap, freeCols, realNC, A, aTmp, nullptr, prof, excl, aPar, lxPar, &lxRed);
  if constexpr (WARM) {
    if (lxNew) {   // back to job rows: rows nn .. n-1 are the child's, the fixed rows keep the parent's values (nothing reads them again)
      const double sh = __shfl(lxRed, (lane >= nn) ? lane - nn : 0, 64);
      *lxNew = (lane >= nn && lane < n) ? sh : lxPar;
    }
  }
  if (okH) {
    const int ja = __shfl((lane < nFree) ? colRemap : 0, (lane < nFree) ? aTmp : 0, 64);
    const double term = (lane < nFree) ? C[(nn + lane) * n + ja] : 0.0;
    for (int r = 0; r < nFree; r++) sAcc += readlane_f64(term, r);
    sAcc += fixedScore;
    const int jaShift = __shfl(ja, (lane >= nn) ? lane - nn : 0, 64);
    if (lane >= nn) aNew = jaShift;
    pushed = true;
  }
}

// One partition: sum of exp(score) over the <= 200 best assignments (RBPHDFilter.hpp:948-959).
// Murty's ranked enumeration by one WORKGROUP of W wavefronts: the children of an expansion -- independent sub-problems --
// are shared out among the waves (child c of the popped node to wave c mod W, each in its own LDS tile of LDSN x LDSN);
// wave 0 pops, and after a barrier pushes the children in partition order and looks at the next score, exactly as the
// one-wave form does.  What happens with the scores is the caller's: onRoot(score) and onTop(score, node) run on wave 0's
// lane 0 and return true to stop (onTop is called for the 2nd, 3rd, ... best, at most maxK - 1 times).
// ctl: [0] parent [1] its partition [2] nodes so far [3] heap length [4] stop [5] ok.
template <int W, int LDSN, class FRoot, class FTop>
__device__ __forceinline__ void murty_kbest_block(double *C, int n, int partitionMax, int realNC, int maxNodes, int maxK, MurtyArena &A, bool &ok,
                                                  double *myTile, int *ctl, double *sScore, unsigned char *sPushed, const int wave, FRoot onRoot,
                                                  FTop onTop) {
  const int lane = threadIdx.x & 63;
#ifdef RFS_PROFILE
  long long tp[5] = {0, 0, 0, 0, 0};
  long long hp[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};   // hungarian_wave's counters: [1] cycles [2] solves [4] main-loop trips [5] BFS dequeues [6] label updates [7] sum of n [8] main-loop cycles
  long long tq = (long long)__builtin_readcyclecounter();
#define MB_STAMP(i) do { const long long tn = (long long)__builtin_readcyclecounter(); tp[i] += tn - tq; tq = tn; } while (0)
#else
#define MB_STAMP(i) do { } while (0)
#endif
  if (wave == 0) {
    int a0;
    double s = 0;
    const bool okr = murty_root_wave(C, n, A, a0, s, nullptr);
    if (lane == 0) {
      ctl[2] = 1; ctl[3] = okr ? 1 : 0; ctl[5] = okr ? 1 : 0;
      ctl[4] = (!okr || onRoot(s)) ? 1 : 0;
    }
  }
  __threadfence_block();
  __syncthreads();
  MB_STAMP(4);
  // (values read back from LDS are wave-uniform, but only readfirstlane tells the compiler so: without it the whole search
  //  would be compiled as divergent control flow)
  for (int k = 1; k < maxK && __builtin_amdgcn_readfirstlane(ctl[4]) == 0; k++) {
    if (wave == 0 && lane == 0) {
      int hl = ctl[3];
      const int parent = heap_pop(A.heap, hl, A.nodeScore);
      ctl[0] = parent; ctl[1] = A.nodeId[parent]; ctl[3] = hl;
    }
    __threadfence_block();
    __syncthreads();
    MB_STAMP(0);
    const int parent = __builtin_amdgcn_readfirstlane(ctl[0]), pp = __builtin_amdgcn_readfirstlane(ctl[1]), nNodes = __builtin_amdgcn_readfirstlane(ctl[2]);
    const int cnt = partitionMax - pp;
    const bool poolFull = cnt > 0 && nNodes + cnt > maxNodes;
    if (!poolFull && cnt > 0) {
      const int aPar = (lane < n) ? A.nodeA[(size_t)parent * MURTY_N + lane] : 0;
      const double termPar = (lane < n) ? C[lane * n + aPar] : 0.0;
      for (int c = wave; c < cnt; c += W) {
        const int nn = pp + c, pn = nNodes + c;
        if (lane == 0) { A.nodeId[pn] = (unsigned char)nn; A.nodeParent[pn] = (short)parent; }
        bool pushed = false;
        double sAcc = 0;
        int aNew = aPar;
#ifdef RFS_PROFILE
        murty_solve_child<LDSN>(myTile, C, n, realNC, A, wave, parent, pp, c, pn, aPar, termPar, pushed, sAcc, aNew, hp);
#else
        murty_solve_child<LDSN>(myTile, C, n, realNC, A, wave, parent, pp, c, pn, aPar, termPar, pushed, sAcc, aNew, nullptr);
#endif
        if (lane < n) A.nodeA[(size_t)pn * MURTY_N + lane] = (unsigned char)aNew;
        if (lane == 0) { sPushed[c] = pushed ? 1 : 0; sScore[c] = sAcc; }
      }
    }
    MB_STAMP(1);
    __threadfence_block();
    __syncthreads();
    MB_STAMP(3);
    if (wave == 0 && lane == 0) {
      int stop = 0;
      if (poolFull) {
        ctl[5] = 0;
        stop = 1;
      } else {
        int hl = ctl[3];
        for (int c = 0; c < cnt; c++)
          if (sPushed[c]) {
            A.nodeScore[nNodes + c] = sScore[c];
            heap_push(A.heap, hl, (short)(nNodes + c), A.nodeScore);
          }
        ctl[2] = nNodes + (cnt > 0 ? cnt : 0);
        ctl[3] = hl;
        if (hl == 0) stop = 1;  // rank == -1
        else {
          const int top = A.heap[0];
          if (onTop(A.nodeScore[top], top)) stop = 1;
        }
      }
      ctl[4] = stop;
    }
    __threadfence_block();
    __syncthreads();
    MB_STAMP(2);
  }
#ifdef RFS_PROFILE
  if (lane == 0 && (blockIdx.x & 255) == 7) printf("murty block %d wave %d: n %d nodes %d; cycles root %lld, pop+barrier %lld, own children %lld, wait for the other waves %lld, push+top+barrier %lld\n", (int)blockIdx.x, wave, n, ctl[2], tp[4], tp[0], tp[1], tp[3], tp[2]);
  if (lane == 0 && (blockIdx.x & 255) == 7 && hp[2] > 0) printf("   wave %d solver: %lld solves, mean dimension %.1f, %lld cycles per solve (%lld in the main loop); per solve: %.1f main-loop trips, %.1f BFS dequeues, %.1f label updates -> %lld cycles per trip\n", wave, hp[2], (double)hp[7] / hp[2], hp[1] / hp[2], hp[8] / hp[2], (double)hp[4] / hp[2], (double)hp[5] / hp[2], (double)hp[6] / hp[2], hp[4] ? hp[8] / hp[4] : 0);
#endif
  ok = __builtin_amdgcn_readfirstlane(ctl[5]) != 0;
}

// ---- the same search with the solves taken off the pop's path --------------------------------------------------------------
// What a node expands to -- each child's sub-problem, its negative constraints along the parent chain, the solver's answer --
// depends on the node and its ancestors only, not on WHEN the node is popped.  Here wave 0 alone runs the search (pop, push,
// scores, stop rules: the sequence of murty_kbest_block, untouched), and the other waves of the workgroup are solvers with a
// mailbox each: wave 0 hands them the children of the node it has just popped AND, when solvers are free, the children of the
// nodes a coming pop is most likely to take (the heap's new top, then the better of the top's two children).  Answers are
// parked in a small table keyed (node, child); a pop whose children are in the table costs heap work only.  Node numbers,
// pushes and pops are exactly the serial ones: the table only replaces a solve by its own result.  No workgroup barrier inside
// the search -- flags in LDS (release / acquire at workgroup scope); wave 0 never waits for anything but a solve in flight, the
// solvers for nothing but a task or the end, so there is no cycle to wait in.

#ifndef MURTY_SPEC_SLOTS
#define MURTY_SPEC_SLOTS 16
#endif
#ifndef MURTY_SOLVER_SLEEP
#define MURTY_SOLVER_SLEEP 4   // x 64 cycles between two looks at the mailbox (1 ... 64 measured at configs[4]: 6.37-6.44 ms, no trend)
#endif
#ifndef MURTY_SEARCH_SLEEP
#define MURTY_SEARCH_SLEEP 2
#endif
#ifndef MURTY_PEEK
#define MURTY_PEEK 2   // heap positions whose children are solved ahead of their pop when solvers are free (round 6, five waves per job, warm solves: 1: 1.50 ms (4 waves), 2: 1.35, 3: 1.38; round 5, eight waves, solves from scratch: 2: 2.33 ms, 3: 2.25, 5: 2.16, 7: 2.49)
#endif
#define MURTY_VSOLVERS (MURTY_CT_WAVES - 1)   /* mailboxes 1..MURTY_VSOLVERS */
struct MurtySpec {
  double score[MURTY_SPEC_SLOTS];
  int ready[MURTY_SPEC_SLOTS];           // slot payload complete (solver: 1; wave 0 clears it when it hands the slot out)
  int taskSeq[MURTY_VSOLVERS + 1];       // wave 0 -> solver v: tasks posted so far
  int doneSeq[MURTY_VSOLVERS + 1];       // solver v -> wave 0: tasks finished so far
  int taskNode[MURTY_VSOLVERS + 1], taskC[MURTY_VSOLVERS + 1], taskSlot[MURTY_VSOLVERS + 1];
  int quit;
  int peekNode[8], peekPart[8];
  unsigned char pushed[MURTY_SPEC_SLOTS];
  unsigned char a[MURTY_SPEC_SLOTS][MURTY_N];
  double lx[MURTY_SPEC_SLOTS][MURTY_WARM_N];   // (small form, MURTY_WARM) the solved child's row duals, until its node has a number
};
__device__ __forceinline__ int murty_flag_load(const int *p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void murty_flag_store(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }

#define HQ_N 16                 // the small form's LDS table (murty_kbest_async<..., SMALL>) keeps this leading dimension

// One mailbox per solver wave.  (A quarter-wave solver variant -- four 16-lane Hungarian solvers per wave, rounds 2-5, test support --
// was bit-identical and slower; it is gone with round 6, when the children's solves became single augmentations.)
// SMALL (extended dimension <= 16): the job's table is read from a copy in LDS (`sC`, taken after the root's solve, which
// rewrites the table in place) and the children's negative constraints come from the per-node sets (`nodeExcl`) instead of the
// walk over the partition chain -- a child's set-up then waits for ONE round of global loads (the node's row, its partition
// index, its set) instead of three to five dependent ones.
template <int W, int LDSN, bool SMALL, class FRoot, class FTop>
__device__ __forceinline__ void murty_kbest_async(double *C, int n, int partitionMax, int realNC, int maxNodes, int maxK, MurtyArena &A, bool &ok,
                                                  double *myTile, int *ctl, double *sScore, unsigned char *sPushed, const int wave, MurtySpec *spec,
                                                  double *sC, const MurtyHeap H, FRoot onRoot, FTop onTop) {
  static_assert(W >= 2 && W <= MURTY_CT_WAVES, "one searching wave + at least one solver");
  constexpr int NS = W - 1;
  static_assert(NS <= MURTY_VSOLVERS && NS < 32, "mailboxes");
  // WARM (round 6): every node keeps its row duals (A.nodeLx), and a child's solve is one augmentation from its parent's
  // (hungarian_warm_wave) -- the small form of the RB-PHD partition sums only; scores, not ranked assignments, are what that path uses
  constexpr bool WARM = SMALL && (MURTY_WARM != 0);
  // ARGQ: the open nodes as an unsorted array scanned by the wave (mq_*) instead of lane 0's binary heap -- with WARM only (scores, not
  // ranked assignments; -DMURTY_ARGQ=0 keeps the heap for A/B)
#ifndef MURTY_ARGQ
#define MURTY_ARGQ 1
#endif
  constexpr bool ARGQ = WARM && (MURTY_ARGQ != 0);
  static_assert(!ARGQ || MURTY_PEEK <= 2, "mq_top3 yields the popped node and two peeks");
  static_assert(!WARM || HQ_N <= MURTY_WARM_N, "the small form's dimension fits the dual-variable records");
  const int lane = threadIdx.x & 63;
  if (wave == 0) {
    if (lane < MURTY_SPEC_SLOTS) spec->ready[lane] = 0;
    if (lane <= MURTY_VSOLVERS) { spec->taskSeq[lane] = 0; spec->doneSeq[lane] = 0; }
    if (lane == 0) spec->quit = 0;
    int a0;
    double s = 0;
    double lxRoot = 0.0;
    const bool okr = murty_root_wave(C, n, A, a0, s, nullptr, WARM ? &lxRoot : nullptr);
    if constexpr (WARM) { if (lane < MURTY_WARM_N) A.nodeLx[lane] = lxRoot; }
    if (lane == 0) {
      if (okr) { H.lid[0] = 0; H.lsc[0] = s; }      // (murty_root_wave pushed node 0 onto the arena's heap: position 0 lives in LDS here)
      if constexpr (SMALL) A.nodeExcl[0] = (unsigned short)(1u << a0);
      ctl[2] = 1; ctl[3] = okr ? 1 : 0; ctl[5] = okr ? 1 : 0;
      ctl[4] = (!okr || onRoot(s)) ? 1 : 0;
    }
  }
  __threadfence_block();
  __syncthreads();
  const double *Cs = C;   // where the children read the table from
  if constexpr (SMALL) {
    // (every wave writes the same values and reads them after its own stores; nobody writes the table after the root)
    for (int t = lane; t < n * n; t += 64) sC[t] = C[t];
    murty_publish();
    Cs = sC;
  }
#ifdef RFS_PROFILE
  long long hp[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  int dbgHit = 0, dbgPosted = 0, dbgDirect = 0, dbgSpec = 0, dbgPops = 0;
  long long dbgWait = 0;
  const long long dbgT0 = (long long)__builtin_readcyclecounter();
  long long sec[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, secT = dbgT0;   // search sections: pop + peeks, table look-up + posts, own solves, results from the table, push + stop rule
#define MS_STAMP(i) do { const long long tn = (long long)__builtin_readcyclecounter(); sec[i] += tn - secT; secT = tn; } while (0)
  long long *const prof = hp;
#else
  long long *const prof = nullptr;
#define MS_STAMP(i) do { } while (0)
#endif
  if (wave != 0) {
    {
      // ---- solver ----
      int seen = 0;
      for (;;) {
        int t;
        while ((t = murty_flag_load(&spec->taskSeq[wave])) == seen) {
          if (murty_flag_load(&spec->quit)) break;
          __builtin_amdgcn_s_sleep(MURTY_SOLVER_SLEEP);
        }
        t = __builtin_amdgcn_readfirstlane(t);
        if (t == seen) break;   // the search is over
        const int X = __builtin_amdgcn_readfirstlane(spec->taskNode[wave]), c = __builtin_amdgcn_readfirstlane(spec->taskC[wave]);
        const int e = __builtin_amdgcn_readfirstlane(spec->taskSlot[wave]);
        const int ppX = A.nodeId[X];
        const int aPar = (lane < n) ? A.nodeA[(size_t)X * MURTY_N + lane] : 0;
        unsigned exclX = 0u;
        if constexpr (SMALL) exclX = A.nodeExcl[X];
        const double termPar = (lane < n) ? Cs[lane * n + aPar] : 0.0;
        bool pushed = false;
        double sAcc = 0;
        int aNew = aPar;
        double lxPar = 0.0, lxNew = 0.0;
        if constexpr (WARM) lxPar = (lane < MURTY_WARM_N) ? A.nodeLx[(size_t)X * MURTY_WARM_N + lane] : 0.0;
        murty_solve_child<LDSN, SMALL, WARM>(myTile, Cs, n, realNC, A, wave, X, ppX, c, 0x7ffe, aPar, termPar, pushed, sAcc, aNew, prof, exclX, lxPar, WARM ? &lxNew : nullptr);
        if constexpr (WARM) { if (lane < MURTY_WARM_N) spec->lx[e][lane] = lxNew; }
        spec->a[e][lane] = (unsigned char)aNew;
        if (lane == 0) { spec->score[e] = sAcc; spec->pushed[e] = pushed ? 1 : 0; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) { murty_flag_store(&spec->ready[e], 1); murty_flag_store(&spec->doneSeq[wave], t); }
        seen = t;
      }
    }
  } else {
    // ---- the search ----
    int tNode = -1, tC = 0;   // lane e < MURTY_SPEC_SLOTS: the key of table slot e (node < 0: free)
    int fifo = 0;             // next slot to recycle when none is free
    int postedL = 0;          // lane v in 1..NS: tasks posted to mailbox v so far
    MqTop top;                // (ARGQ) the best three open nodes, found after the last push
    int qLen = __builtin_amdgcn_readfirstlane(ctl[3]);
    if constexpr (ARGQ) mq_top3(H, qLen, top);
    for (int k = 1; k < maxK && __builtin_amdgcn_readfirstlane(ctl[4]) == 0; k++) {
      if constexpr (ARGQ) {
        const int parent = top.id[0];
        mq_remove(H, qLen, top.pos0);
        if (lane == 0) {
          ctl[0] = parent; ctl[1] = A.nodeId[parent]; ctl[3] = qLen;
          const int b1 = top.id[1], b2 = top.id[2];     // what is left, best first: the nodes a coming pop is most likely to take
          spec->peekNode[0] = b1; spec->peekPart[0] = (b1 >= 0) ? (int)A.nodeId[b1] : 0;
          if (MURTY_PEEK > 1) { spec->peekNode[1] = b2; spec->peekPart[1] = (b2 >= 0) ? (int)A.nodeId[b2] : 0; }
        }
      } else
      if (lane == 0) {
        int hl = ctl[3];
        const int parent = mheap_pop(H, hl);
        ctl[0] = parent; ctl[1] = A.nodeId[parent]; ctl[3] = hl;
        // the nodes a coming pop is most likely to take: the new top, then the better of its two children, then the other one
        // (MURTY_PEEK > 3: the heap's first three levels, positions 0 ... 6, by score -- the exact top three and a good guess beyond)
        if constexpr (MURTY_PEEK <= 3) {
          const int b1 = (hl > 0) ? (int)mheap_id(H, 0) : -1;
          int b2 = (hl > 1) ? (int)mheap_id(H, 1) : -1, b3 = (hl > 2) ? (int)mheap_id(H, 2) : -1;
          if (b3 >= 0 && mheap_sc(H, 2) > mheap_sc(H, 1)) { const int tmp = b2; b2 = b3; b3 = tmp; }
          spec->peekNode[0] = b1; spec->peekPart[0] = (b1 >= 0) ? (int)A.nodeId[b1] : 0;
          spec->peekNode[1] = b2; spec->peekPart[1] = (b2 >= 0) ? (int)A.nodeId[b2] : 0;
          spec->peekNode[2] = b3; spec->peekPart[2] = (b3 >= 0) ? (int)A.nodeId[b3] : 0;
        } else {
          static_assert(MURTY_HEAP_LDS >= 7 && MURTY_PEEK <= 7, "the peeked positions live in the LDS part of the heap");
          int id[7];
          double sc[7];
#pragma unroll
          for (int q = 0; q < 7; q++) { id[q] = (q < hl) ? (int)H.lid[q] : -1; sc[q] = (q < hl) ? H.lsc[q] : -1.7976931348623157e308; }
#pragma unroll
          for (int a = 1; a < 7; a++)          // insertion sort by score, descending (position 0 is the maximum already)
#pragma unroll
            for (int b = a; b > 1; b--)
              if (sc[b] > sc[b - 1]) { const double ts = sc[b]; sc[b] = sc[b - 1]; sc[b - 1] = ts; const int ti = id[b]; id[b] = id[b - 1]; id[b - 1] = ti; }
#pragma unroll
          for (int q = 0; q < MURTY_PEEK; q++) { spec->peekNode[q] = id[q]; spec->peekPart[q] = (id[q] >= 0) ? (int)A.nodeId[id[q]] : 0; }
        }
      }
      MS_STAMP(5);    // (profile builds: lane 0's pop + peeks alone; section 0 below is then the fence and the control words)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // (also: the nodes pushed so far are visible to the solvers before any task names them)
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      const int parent = __builtin_amdgcn_readfirstlane(ctl[0]), pp = __builtin_amdgcn_readfirstlane(ctl[1]), nNodes = __builtin_amdgcn_readfirstlane(ctl[2]);
      const int cnt = partitionMax - pp;
      const bool poolFull = cnt > 0 && nNodes + cnt > maxNodes;
#ifdef RFS_PROFILE
      dbgPops++;
#endif
      MS_STAMP(0);
      if (!poolFull && cnt > 0) {
        // which children of this node are in the table (solved or being solved)?
        int mySlot = -1;
#pragma unroll
        for (int e = 0; e < MURTY_SPEC_SLOTS; e++) {
          const int en = __builtin_amdgcn_readlane(tNode, e), ec = __builtin_amdgcn_readlane(tC, e);
          if (en == parent && ec == lane) mySlot = e;
        }
#ifdef RFS_PROFILE
        dbgHit += __popcll(__ballot(lane < cnt && mySlot >= 0));
#endif
        // solvers without a task in flight
        unsigned freeW = (unsigned)__ballot(lane >= 1 && lane <= NS && murty_flag_load(&spec->doneSeq[(lane >= 1 && lane <= NS) ? lane : 0]) == postedL);
        // a table slot for (X, c): a free one, else the oldest solved entry that does not belong to this pop; never one in flight
        auto take_slot = [&](const int X, const int c) -> int {
          const int rdy = (lane < MURTY_SPEC_SLOTS) ? murty_flag_load(&spec->ready[lane]) : 0;
          const unsigned freeS = (unsigned)__ballot(lane < MURTY_SPEC_SLOTS && tNode < 0);
          const unsigned evict = (unsigned)__ballot(lane < MURTY_SPEC_SLOTS && tNode >= 0 && tNode != parent && rdy != 0);
          int e = -1;
          if (freeS) e = __builtin_ctz(freeS);
          else
            for (int t = 0; t < MURTY_SPEC_SLOTS; t++) {
              const int q = (fifo + t) & (MURTY_SPEC_SLOTS - 1);
              if ((evict >> q) & 1u) { e = q; fifo = (q + 1) & (MURTY_SPEC_SLOTS - 1); break; }
            }
          if (e >= 0 && lane == e) { tNode = X; tC = c; spec->ready[e] = 0; }
          return e;
        };
        auto post = [&](const int w, const int X, const int c, const int e) {
          if (lane == 0) { spec->taskNode[w] = X; spec->taskC[w] = c; spec->taskSlot[w] = e; }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
          __builtin_amdgcn_wave_barrier();
          if (lane == w) { postedL++; murty_flag_store(&spec->taskSeq[w], postedL); }
        };
        // this node's missing children: the first one stays with wave 0 (it has nothing else to do until they are all there),
        // the others go to free solvers
        const unsigned long long miss = __ballot(lane < cnt && mySlot < 0);
        unsigned long long direct = miss & (0ull - miss);
        for (unsigned long long g = miss & ~direct; g; g &= g - 1ull) {
          const int c = __builtin_ctzll(g);
          const int e = freeW ? take_slot(parent, c) : -1;
          if (e >= 0) {
            const int w = __builtin_ctz(freeW);
            freeW &= freeW - 1u;
            post(w, parent, c, e);
            if (lane == c) mySlot = e;
#ifdef RFS_PROFILE
            dbgPosted++;
#endif
          } else {
            direct |= 1ull << c;
          }
        }
        MS_STAMP(6);    // (look-up + this node's posts; section 1 below: the posts ahead of coming pops)
        // solvers still free: children of the nodes next in the heap
        for (int cand = 0; cand < MURTY_PEEK && freeW; cand++) {
          const int X = __builtin_amdgcn_readfirstlane(spec->peekNode[cand]);
          if (X < 0) continue;
          const int cntX = partitionMax - __builtin_amdgcn_readfirstlane(spec->peekPart[cand]);
          for (int c = 0; c < cntX && freeW; c++) {
            if (__ballot(lane < MURTY_SPEC_SLOTS && tNode == X && tC == c) != 0ull) continue;   // in the table already
            const int e = take_slot(X, c);
            if (e < 0) { freeW = 0; break; }
            const int w = __builtin_ctz(freeW);
            freeW &= freeW - 1u;
            post(w, X, c, e);
#ifdef RFS_PROFILE
            dbgSpec++;
#endif
          }
        }
        MS_STAMP(1);
        // the children nobody took: solved here
        const int aPar = (lane < n) ? A.nodeA[(size_t)parent * MURTY_N + lane] : 0;
        const double termPar = (lane < n) ? Cs[lane * n + aPar] : 0.0;
        unsigned exclPar = 0;
        if constexpr (SMALL) exclPar = A.nodeExcl[parent];
        double lxParent = 0.0;
        if constexpr (WARM) lxParent = (lane < MURTY_WARM_N) ? A.nodeLx[(size_t)parent * MURTY_WARM_N + lane] : 0.0;
        // (small form) what child 0 of the new node -- child c of `parent`, created at row nn = pp + c -- must not take in its
        // first row, which is row nn again: the constraint walk (src/MurtyAlgorithm.cpp:247-265) visits the node itself, its
        // parent, and goes on upwards for as long as the ancestor was created at the same row.  So: the node's own choice for
        // row nn, and either everything child 0 of the parent must not take (c == 0: same row) or the parent's choice for row nn.
        auto note_excl = [&](const int c, const int pn, const int aNew) {
          if constexpr (SMALL) {
            const unsigned ex = (1u << __builtin_amdgcn_readlane(aNew, pp + c)) | (c == 0 ? exclPar : (1u << __builtin_amdgcn_readlane(aPar, pp + c)));
            if (lane == 0) A.nodeExcl[pn] = (unsigned short)ex;
          }
        };
        MS_STAMP(7);    // (the parent's record loaded; section 2 below: node records written + wave 0's own solves)
        for (int c = 0; c < cnt; c++) {
          const int nn = pp + c, pn = nNodes + c;
          if (lane == 0) { A.nodeId[pn] = (unsigned char)nn; A.nodeParent[pn] = (short)parent; }
          if (!((direct >> c) & 1ull)) continue;
          bool pushed = false;
          double sAcc = 0;
          int aNew = aPar;
          double lxNew = 0.0;
          murty_solve_child<LDSN, SMALL, WARM>(myTile, Cs, n, realNC, A, 0, parent, pp, c, pn, aPar, termPar, pushed, sAcc, aNew, prof, exclPar, lxParent, WARM ? &lxNew : nullptr);
          if constexpr (WARM) { if (lane >= pp + c && lane < n) A.nodeLx[(size_t)pn * MURTY_WARM_N + lane] = lxNew; }   // (rows below pp + c stay with the parent's columns in every descendant: their duals are never read)
          if (lane < n) A.nodeA[(size_t)pn * MURTY_N + lane] = (unsigned char)aNew;
          if (lane == 0) { sPushed[c] = pushed ? 1 : 0; sScore[c] = sAcc; }
          note_excl(c, pn, aNew);
#ifdef RFS_PROFILE
          dbgDirect++;
#endif
        }
        MS_STAMP(2);
        // the others come out of the table
        for (int c = 0; c < cnt; c++) {
          if ((direct >> c) & 1ull) continue;
          const int e = __builtin_amdgcn_readlane(mySlot, c);
          const int pn = nNodes + c;
#ifdef RFS_PROFILE
          const long long tw = (long long)__builtin_readcyclecounter();
#endif
          while (!murty_flag_load(&spec->ready[e])) __builtin_amdgcn_s_sleep(MURTY_SEARCH_SLEEP);
#ifdef RFS_PROFILE
          dbgWait += (long long)__builtin_readcyclecounter() - tw;
#endif
          const int aNew = spec->a[e][lane];
          if constexpr (WARM) { if (lane >= pp + c && lane < n) A.nodeLx[(size_t)pn * MURTY_WARM_N + lane] = spec->lx[e][lane]; }
          if (lane < n) A.nodeA[(size_t)pn * MURTY_N + lane] = (unsigned char)aNew;
          if (lane == 0) { sPushed[c] = spec->pushed[e]; sScore[c] = spec->score[e]; }
          note_excl(c, pn, aNew);
          if (lane == e) tNode = -1;   // the slot is free again
        }
        MS_STAMP(3);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      MS_STAMP(8);      // (the fence ahead of the push)
      if constexpr (ARGQ) {
        int stop = 0;
        if (poolFull) {
          if (lane == 0) ctl[5] = 0;
          stop = 1;
        } else {
          for (int c = 0; c < cnt; c++)
            if (sPushed[c]) {      // (uniform: one LDS byte)
              const double sc = sScore[c];
              if (lane == 0 && qLen >= MURTY_HEAP_LDS) A.nodeScore[nNodes + c] = sc;   // (an entry beyond the LDS positions keeps its score in the arena; the others never read it)
              mq_push(H, qLen, nNodes + c, sc);
            }
          if (lane == 0) { ctl[2] = nNodes + (cnt > 0 ? cnt : 0); ctl[3] = qLen; }
          if (qLen == 0) stop = 1;  // rank == -1
          else {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");     // lane 0's appends -> the scan's reads
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            mq_top3(H, qLen, top);
            int st = 0;
            if (lane == 0) st = onTop(top.sc0, top.id[0]) ? 1 : 0;
            stop = __builtin_amdgcn_readfirstlane(st);
          }
        }
        if (lane == 0) ctl[4] = stop;
      } else
      if (lane == 0) {
        int stop = 0;
        if (poolFull) {
          ctl[5] = 0;
          stop = 1;
        } else {
          int hl = ctl[3];
          for (int c = 0; c < cnt; c++)
            if (sPushed[c]) {
              A.nodeScore[nNodes + c] = sScore[c];
              mheap_push(H, hl, (short)(nNodes + c), sScore[c]);
            }
          ctl[2] = nNodes + (cnt > 0 ? cnt : 0);
          ctl[3] = hl;
          if (hl == 0) stop = 1;  // rank == -1
          else {
            const int top = mheap_id(H, 0);
            if (onTop(mheap_sc(H, 0), top)) stop = 1;
          }
        }
        ctl[4] = stop;
      }
      MS_STAMP(9);      // (lane 0's pushes + stop rule; section 4 below: the fence behind them)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      MS_STAMP(4);
    }
    if (lane == 0) murty_flag_store(&spec->quit, 1);
#ifdef RFS_PROFILE
    if (lane == 0 && (blockIdx.x & 255) == 7)
      printf("murty block %d: n %d pops %d nodes %d; children from the table %d, posted at the pop %d, solved by wave 0 %d, ahead of a pop %d; wave 0 waited %lld of %lld cycles; its own solves: %lld, cycles in the solver %lld (main loop %lld), trips %lld, BFS dequeues %lld, label updates %lld, sum of dimensions %lld\n",
             (int)blockIdx.x, n, dbgPops, ctl[2], dbgHit, dbgPosted, dbgDirect, dbgSpec, dbgWait, (long long)__builtin_readcyclecounter() - dbgT0, hp[2], hp[1], hp[8], hp[4], hp[5], hp[6], hp[7]);
    if (lane == 0 && (blockIdx.x & 255) == 7)
      printf("murty sections block %d n %d pops %d: pop+peeks (lane 0) %lld | fence+control %lld | look-up + own posts %lld | posts ahead %lld | parent record %lld | records + own solves %lld | table results %lld | fence %lld | pushes+stop (lane 0) %lld | fence %lld cycles\n",
             (int)blockIdx.x, n, dbgPops, sec[5], sec[0], sec[6], sec[1], sec[7], sec[2], sec[3], sec[8], sec[9], sec[4]);
#endif
  }
  __threadfence_block();
  __syncthreads();
  ok = __builtin_amdgcn_readfirstlane(ctl[5]) != 0;
}

// One partition: sum of exp(score) over the <= 200 best assignments (RBPHDFilter.hpp:948-959).
template <int W>
__device__ __forceinline__ double murty_partition_sum_block(double *C, int n, int nR, int nC, MurtyArena &A, bool &ok, double *myTile, int *ctl,
                                                            double *sSum, double *sScore, unsigned char *sPushed, const int wave,
                                                            MurtySpec *spec = nullptr, double *sC = nullptr, short *heapId = nullptr,
                                                            double *heapSc = nullptr) {
  const double BIG_NEG = -1000.0;
  const int realNR = nR > n ? n : nR, realNC = nC > n ? n : nC;
  const int partitionMax = (realNR == n) ? n - 1 : realNR;
  if (threadIdx.x == 0) *sSum = 0.0;
  // Early end of the ranked enumeration (round 5; same sum, bit for bit).  The caller adds exp(score) over the assignments in
  // the order Murty returns them (RBPHDFilter.hpp:948-959), and the scores come out non-increasing: a child's assignment is
  // feasible for its parent's sub-problem, whose solution is that sub-problem's optimum.  Once a term is below 2^-56 of the
  // running sum, it and every later one is below half an ulp of it (a sum in [2^e, 2^(e+1)) has ulp 2^(e-52) >= 2^-53 of it; the
  // factor 4 to spare covers the solver's 1e-12 tolerances): each remaining addition rounds to no change, so the loop may end.
  // The premise -- every node's solution IS its sub-problem's optimum -- fails only if a solve picks a cell the negative
  // constraints have set to -bigNumber (-10000; the resulting duplicate would carry its true score, src/MurtyAlgorithm.cpp:247-265,
  // 300-311).  That cannot happen while the table's cells lie in [-1000, 1000]: from the sub-problem's unconstrained optimum
  // (score U) one swap of two rows' columns gives an allowed assignment scoring >= U - 2 (max - min) >= U - 4000, while any
  // assignment through a forbidden cell has modified score <= U + 1000 - 10000.  Cells are logs floored at BIG_NEG_NUM = -1000
  // (RBPHDFilter.hpp:907-940), so only log(1 - Pd) = -inf or a NaN can break the range: such a table runs all 200 calls.
  // Measured on configs[4]'s own jobs through the oracle (tools/murty_early_stop_study.py): 62 % of the calls go, every job stops
  // early (dimension 15: 200 -> 74 calls on average), no score ever increases.  RFS_MURTY_FULL_LOOP keeps the full loop (A/B).
  bool earlyStop = false;
#ifndef RFS_MURTY_FULL_LOOP
  {
    const int lane = threadIdx.x & 63;
    double mn = 0.0, mx = 0.0;
    bool bad = false;
    for (int t = lane; t < n * n; t += 64) { const double c = C[t]; mn = raw_min_f64(mn, c); mx = raw_max_f64(mx, c); bad |= !(c == c); }
    mn = wave_min_f64(mn); mx = wave_max_f64(mx);
    earlyStop = __ballot(bad) == 0ull && mn >= -1000.0 && mx <= 1000.0;
  }
#endif
  // (rfs_exp, common.h: relative error <= 1e-14, coefficients as scalar operands.  The library exp's eleven fp64 coefficients were
  //  materialised in vector registers ahead of the search loop and -- under the 64-VGPR cap -- spilled: 72 of the kernel's 140 bytes
  //  of scratch per lane, i.e. half of the 154 MB the launch wrote at configs[4]; MURTY_LIB_EXP=1 restores it.)
#ifndef MURTY_LIB_EXP
#define MURTY_LIB_EXP 0
#endif
  auto term_of = [](double s) { return MURTY_LIB_EXP ? exp(s) : rfs_exp(s); };
  auto onRoot = [&](double s) { if (s < BIG_NEG) return true; *sSum = term_of(s); return false; };
  auto onTop = [&](double st, int) {
    if (st < BIG_NEG) return true;
    const double t = term_of(st), sum = *sSum + t;
    *sSum = sum;
    return earlyStop && t < sum * 0x1p-56;
  };
  if constexpr (W >= 2) {
    if (spec) {
      const MurtyHeap H{heapId, heapSc, A.heap, A.nodeScore};
      if (sC && n <= HQ_N) {
        murty_kbest_async<W, MURTY_LDS_N, true>(C, n, partitionMax, realNC, MURTY_MAX_NODES, MURTY_KBEST, A, ok, myTile, ctl, sScore, sPushed, wave, spec, sC, H, onRoot, onTop);
        return *sSum;
      }
      murty_kbest_async<W, MURTY_LDS_N, false>(C, n, partitionMax, realNC, MURTY_MAX_NODES, MURTY_KBEST, A, ok, myTile, ctl, sScore, sPushed, wave, spec, nullptr, H, onRoot, onTop);
      return *sSum;
    }
  }
  murty_kbest_block<W, MURTY_LDS_N>(C, n, partitionMax, realNC, MURTY_MAX_NODES, MURTY_KBEST, A, ok, myTile, ctl, sScore, sPushed, wave, onRoot, onTop);
  return *sSum;
}


// One workgroup of MURTY_JOB_WAVES wavefronts per queued partition (jobs strided over the grid); the last workgroup to
// finish multiplies every particle's factors into its weight, in partition (slot) order.  Q.count[0] = number of jobs,
// Q.count[1] = finished-workgroup ticket.  With an empty queue (the common case: no partition above 8) every workgroup
// exits at once -- one empty launch, no host round trip.
#define MURTY_JOB_BLOCKS 2048
// Besides the Murty jobs this is the step's POST kernel: whoever finishes last (block 0 alone when the queue is empty -- the
// usual case: at the shipped 3-sigma gate Murty is never entered) multiplies the Murty factors into the particle weights,
// clears the queue for the next step and, when `sums` is given, leaves {sum w, sum w^2} of the shard there
// (ParticleFilter::normalizeWeights / N_eff, include/ParticleFilter.hpp:352-363, 405-415) and -- `normalize` != 0, a filter
// that lives on one GPU -- divides the weights by the sum right away.  One launch instead of three (Murty, sums, divide).
// preDiv (round 5, multi-GPU hosts): the all-reduced weight total of the PREVIOUS step, which has travelled over xGMI while this
// step's kernel ran -- the weights are divided by it HERE, before this step's sums are taken, instead of by a kernel of its own
// behind the collective at the end of the previous step: (w L) / T in place of (w / T) L, an ulp apart, and the collective is
// off the step's critical path.
__device__ __forceinline__ bool coll_wait(const int *word, int need);   // (below, with StepOut)
__device__ __forceinline__ void step_post_tail(double *weight, int N, double *sums, int normalize, const double *preDiv = nullptr, int *collSeq = nullptr,
                                               int collNeed = 0, int collPost = 0, int *err = nullptr) {
  if (!sums) return;
  __shared__ double sPd;
  if (collSeq) {      // event-free hand-over: wait for the previous collective's sequence number, then read its total coherently
    if (threadIdx.x == 0) {
      if (collNeed > 0 && !coll_wait(collSeq + 1, collNeed) && err) atomicOr(err, ERRBIT_COLLECTIVE);
      sPd = preDiv ? __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long *>(preDiv), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) : 1.0;
    }
    __syncthreads();
  }
  const double pd = collSeq ? sPd : (preDiv ? preDiv[0] : 1.0);
  __shared__ double sA[16], sB[16];   // (one entry per wave of the block, whatever it was launched with)
  __shared__ double sDiv;
  // (eight loads in flight per thread: one block sums the whole shard, and taken one at a time the ~16 dependent L2 round trips
  //  of a 2000-particle shard were most of this kernel's 7 us; the order of the additions is unchanged)
  constexpr int U = 8;
  double a = 0, b = 0;
  for (int k0 = threadIdx.x; k0 < N; k0 += U * (int)blockDim.x) {
    double v[U];
#pragma unroll
    for (int j = 0; j < U; j++) { const int k = k0 + j * (int)blockDim.x; v[j] = (k < N) ? weight[k] : 0.0; }
    if (preDiv) {
#pragma unroll
      for (int j = 0; j < U; j++) { const int k = k0 + j * (int)blockDim.x; v[j] = v[j] / pd; if (k < N) weight[k] = v[j]; }
    }
#pragma unroll
    for (int j = 0; j < U; j++) { a += v[j]; b += v[j] * v[j]; }
  }
  a = wave_sum_dpp(a); b = wave_sum_dpp(b);
  if ((threadIdx.x & 63) == 0) { sA[threadIdx.x >> 6] = a; sB[threadIdx.x >> 6] = b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double x = 0, y = 0;
    for (int k = 0; k < (int)(blockDim.x >> 6); k++) { x += sA[k]; y += sB[k]; }
    sums[0] = x; sums[1] = y;
    sDiv = x;
    if (collSeq) __hip_atomic_store(collSeq, collPost, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);   // the sums are out: the side stream's gate may pass
  }
  if (!normalize) return;
  __syncthreads();
  const double d = sDiv;
  for (int k0 = threadIdx.x; k0 < N; k0 += U * (int)blockDim.x) {
    double v[U];
#pragma unroll
    for (int j = 0; j < U; j++) { const int k = k0 + j * (int)blockDim.x; v[j] = (k < N) ? weight[k] : 0.0; }
#pragma unroll
    for (int j = 0; j < U; j++) { const int k = k0 + j * (int)blockDim.x; if (k < N) weight[k] = v[j] / d; }
  }
}
// Results of a step straight into a PINNED host buffer, by the post kernel's last workgroup (rfsgpu_update_io, round 5): the
// particle weights, the device error word, then -- behind a system-scope release -- the sequence number the host spins on.  No
// copy commands behind the step and no stream synchronisation: two blit kernels, their queue hand-overs and the wake-up of
// hipStreamSynchronize were 20 us of an update through the boundary at configs[1].
struct StepOut {
  double *hostW;     // [N] (nullptr: nothing to deliver)
  int *hostFlag;     // [0] error word, [1] sequence number
  int seq;
  const double *preDiv;   // device: {sum w, sum w^2} of the previous step over all shards, or nullptr (step_post_tail)
  // The same hand-over WITHOUT stream events (rfsgpu_step_async_trailing): collSeq[1] is raised to k by a one-thread kernel behind the
  // collective of step k on its side stream; the post kernel of step k + 1 waits for it HERE (it has been there for ~100 us), and raises
  // collSeq[0] to its own number once its sums are written, on which the side stream's gate kernel waits.  An event record and an event
  // wait are a marker and a barrier packet on the step's stream, ~4 us each at configs[1].
  int *collSeq;
  int collNeed, collPost;
};
// (bounded: a host that never runs the collective must not hang the device -- the waiter gives up after 0.5 s of the constant 100 MHz
//  clock and raises the collective-protocol bit, its own error bit and message, not Murty's)
#define COLL_WAIT_TICKS 50000000ll
__device__ __forceinline__ bool coll_wait(const int *word, int need) {
  const long long t0 = (long long)wall_clock64();
  for (;;) {
    if (__hip_atomic_load(word, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) >= need) return true;
    if ((long long)wall_clock64() - t0 > COLL_WAIT_TICKS) return false;
    __builtin_amdgcn_s_sleep(32);
  }
}
__global__ void coll_gate_kernel(const int *collSeq, int need, int *err) { if (threadIdx.x == 0 && !coll_wait(collSeq, need)) atomicOr(err, ERRBIT_COLLECTIVE); }
// rfsgpu_collective_probe: the hand-over played once with nothing at stake -- *verdict = need if the word arrived within `ticks`
#define COLL_PROBE_TICKS 20000000ll    // 0.2 s
__global__ void coll_probe_kernel(const int *word, int need, int *verdict, long long ticks) {
  if (threadIdx.x != 0) return;
  const long long t0 = (long long)wall_clock64();
  int seen = 0;
  for (;;) {
    if (__hip_atomic_load(word, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) >= need) { seen = need; break; }
    if ((long long)wall_clock64() - t0 > ticks) break;
    __builtin_amdgcn_s_sleep(32);
  }
  *verdict = seen;
}
__global__ void coll_spin_kernel(long long ticks) {     // (test hook: holds a stream back)
  if (threadIdx.x != 0) return;
  const long long t0 = (long long)wall_clock64();
  while ((long long)wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}
__global__ void coll_publish_kernel(int *word, int seq) { if (threadIdx.x == 0) __hip_atomic_store(word, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void step_post_out(const double *weight, int N, int *err, const StepOut &SO) {
  if (!SO.hostW) return;
  __threadfence();
  __syncthreads();     // every thread's weight writes (Murty factors, division) before anybody reads them back
  for (int k = threadIdx.x; k < N; k += blockDim.x) SO.hostW[k] = weight[k];
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) {
    SO.hostFlag[0] = __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&SO.hostFlag[1], SO.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
// Launch order of the NEXT step's particles from THIS step's measured durations (round 6; the Victoria Park step: one wavefront per
// particle, 5000 particles on ~3000 resident wave slots, i.e. two rounds whose second ends with whatever started last -- a third of the
// launch was tail).  With the particles that took longest first, the tail is made of short ones: 190.6 -> 163.7 us at configs[3] with the
// durations of the previous launch, 164.3 with eight classes of them (tools/vp_order_study.py; shortest first: 196.4).  A particle's cost
// repeats from step to step (r = 0.85 in a running filter, tools/vp_cost_repeat.py).  The step's post kernel sorts the particles into
// STEP_ORDER_CLASSES equal-width classes of duration, longest first, ONE CLASS PER WORKGROUP (blocks 1 .. STEP_ORDER_CLASSES, beside
// block 0's weight sums), in ONE pass over the durations: a thread reads its share, counts the particles of the classes ahead of its
// own block's and keeps a bit per particle of its own class; wave sums, one barrier, and the class is written where it starts --
// members in (wavefront, bit, lane) order, so the order array is a function of the durations alone.  No workgroup waits for another and
// nothing is zeroed between steps.  The class limits are the extrema of the PREVIOUS step's durations (`ext`, two pairs used in turn:
// the first class block writes the pair the next launch reads; all zero at the start = one class = the identity order): with this
// step's own extrema every block would have to read the durations twice.  What this costs on the post kernel's critical path at
// n = 5000 (profiles/r06t): the whole sort in block 1 with an LDS atomic per particle 6.8 us; a class per block with the extrema first
// and an atomic per ballot 10.6 us (2 fetch rounds x 3 passes); this form: see there.
#ifndef STEP_ORDER_CLASSES
#define STEP_ORDER_CLASSES 32
#endif
#define STEP_ORDER_QUADS 8         // 16-byte loads a thread has in flight at a time (a round: 32 durations)
#define STEP_ORDER_MAX_ROUNDS 2    // (a bit per particle in one 64-bit word: n <= 64 x blockDim, else the order is left as it is)
struct StepOrderArg {
  const float *cost;   // [n] ticks per particle, written by the step kernel (nullptr: no ordering)
  int *order;          // [n] launch slot -> particle, read by the next step kernel
  int n;
  float *ext;          // [4] {lo, hi} x 2: extrema of the durations, read at [2 parity], written at [2 (1 - parity)]
  int parity;
};
__device__ __forceinline__ void step_cost_order_class(const StepOrderArg &O, const int c) {
  __shared__ unsigned sLo, sHi;
  __shared__ int sBase, sMine[16];
  const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wave = tid >> 6, nw = nt >> 6;
  constexpr int PER_ROUND = 4 * STEP_ORDER_QUADS;
  const int rounds = (O.n + nt * PER_ROUND - 1) / (nt * PER_ROUND);
  if (rounds > STEP_ORDER_MAX_ROUNDS) return;    // (block-uniform; the order array keeps the permutation it holds)
  if (tid == 0) { sLo = 0xffffffffu; sHi = 0u; sBase = 0; }
  __syncthreads();
  const float fLo = O.ext[2 * O.parity], fHi = O.ext[2 * O.parity + 1];
  const float scale = (fHi > fLo) ? (float)STEP_ORDER_CLASSES / (fHi - fLo) : 0.f;
  unsigned lo = 0xffffffffu, hi = 0u;            // durations are positive floats: their bit patterns order like the values
  unsigned long long mine = 0ull;                // bit 32 r + 4 j + e: particle 4 ((r QUADS + j) nt + tid) + e
  int ahead = 0;
  const float4 *const cost4 = (const float4 *)O.cost;       // (the array is 16-byte aligned and four floats longer than its capacity)
  for (int r = 0; r < rounds; r++) {
    float4 v[STEP_ORDER_QUADS];
#pragma unroll
    for (int j = 0; j < STEP_ORDER_QUADS; j++) {
      const int k4 = (r * STEP_ORDER_QUADS + j) * nt + tid;
      v[j] = 4 * k4 < O.n ? cost4[k4] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int j = 0; j < STEP_ORDER_QUADS; j++) {
      const int k = 4 * ((r * STEP_ORDER_QUADS + j) * nt + tid);
      const float ve[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
#pragma unroll
      for (int e = 0; e < 4; e++) {
        if (k + e >= O.n) continue;
        const float x = fmaxf(ve[e], 0.f);
        int q = (int)((fHi - x) * scale);        // class 0 = the longest
        q = q < 0 ? 0 : (q >= STEP_ORDER_CLASSES ? STEP_ORDER_CLASSES - 1 : q);
        ahead += q < c ? 1 : 0;
        if (q == c) mine |= 1ull << (r * PER_ROUND + 4 * j + e);
        const unsigned b = __float_as_uint(x);
        lo = min(lo, b); hi = max(hi, b);
      }
    }
  }
  const int aheadW = wave_sum_i(ahead), mineW = wave_sum_i(__popcll(mine));
  if (lane == 0) { if (aheadW) atomicAdd(&sBase, aheadW); sMine[wave] = mineW; }
  if (c == 0) {                                  // (block-uniform) the limits the next launch's classes use
    lo = wave_min_u32(lo); hi = wave_max_u32(hi);
    if (lane == 0) { atomicMin(&sLo, lo); atomicMax(&sHi, hi); }
  }
  __syncthreads();
  if (c == 0 && tid == 0 && sLo <= sHi) { O.ext[2 * (1 - O.parity)] = __uint_as_float(sLo); O.ext[2 * (1 - O.parity) + 1] = __uint_as_float(sHi); }
  int total = 0, before = 0;
  for (int w = 0; w < nw; w++) { const int m = sMine[w]; total += m; before += w < wave ? m : 0; }
  if (total == 0) return;
  int at = sBase + before;                       // a wavefront's members behind those of the wavefronts before it
  const int nbits = rounds * PER_ROUND;
  for (int b = 0; b < nbits; b++) {
    const bool is = (mine >> b) & 1ull;
    const unsigned long long m = __ballot(is);
    if (m == 0ull) continue;                     // (wave-uniform)
    if (is) O.order[at + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u))] = 4 * ((b >> 2) * nt + tid) + (b & 3);
    at += __popcll(m);
  }
}

// Longest jobs first: a job's duration grows with its extended dimension (2.3 ms at 9, 10 ms at 15 at configs[4]), the jobs are
// queued in whatever order the particles' weighting phases reach them, and more jobs than resident workgroups means a second
// round -- in which a 10 ms job started after the first 2 ms ones have finished sets the launch's length.  One workgroup sorts the
// job indices by dimension, descending (counting sort; the order among equals is irrelevant: jobs are independent and their
// factors are multiplied into the weights in slot order afterwards).
__global__ __launch_bounds__(1024) void murty_order_kernel(MurtyQueue Q) {
  __shared__ int hist[MURTY_N + 2];
  const int nJobs = min(*Q.count, Q.maxJobs);
  for (int k = threadIdx.x; k < MURTY_N + 2; k += blockDim.x) hist[k] = 0;
  __syncthreads();
  for (int j = threadIdx.x; j < nJobs; j += blockDim.x) {
    const int n = min(Q.jobs[j].nR + Q.jobs[j].nC, MURTY_N + 1);
    atomicAdd(&hist[MURTY_N + 1 - n], 1);            // bucket 0 = the largest dimension
  }
  __syncthreads();
  if (threadIdx.x == 0) { int acc = 0; for (int k = 0; k < MURTY_N + 2; k++) { const int c = hist[k]; hist[k] = acc; acc += c; } }
  __syncthreads();
  for (int j = threadIdx.x; j < nJobs; j += blockDim.x) {
    const int n = min(Q.jobs[j].nR + Q.jobs[j].nC, MURTY_N + 1);
    Q.order[atomicAdd(&hist[MURTY_N + 1 - n], 1)] = j;
  }
}
#ifndef MURTY_WAVES_PER_EU
#define MURTY_WAVES_PER_EU 8   // <= 64 VGPRs: four eight-wave workgroups per CU (round 3: five six-wave ones instead of three; see MURTY_JOB_WAVES).  No scratch since the build runs with -disable-machine-licm
#endif
#ifndef MURTY_LIGHT_WAVES
#define MURTY_LIGHT_WAVES (MURTY_JOB_WAVES >= 8 ? 8 : 4)   // (round 5: eight -- the first update that queues partitions 5.8 -> 4.4 ms at configs[4], the empty-queue step of configs[1] unchanged at 121.4-122.3 us)
#endif
#ifndef MURTY_FIRST_BLOCKS
#define MURTY_FIRST_BLOCKS 2048   // workgroups of the light instance (a filter that has not shown Murty work yet): the full grid --
// measured (tools/murty_first_step.py, r04): the empty-queue step at configs[1] costs the same with 64 ... 2048 of these
// workgroups (138.2-139.0 us either way), while the FIRST step that queues partitions at configs[4] takes 81.7 ms on 64
// workgroups, 25.5 on 256, 15.8 on 512, 10.7 on 1024 and 9.2 on 2048 (steady state, the capped six-wave instance: 5.1-5.3)
#endif
// Two instances.  <MURTY_JOB_WAVES, MURTY_WAVES_PER_EU>: the one for filters that have shown Murty work.  <MURTY_LIGHT_WAVES, 0>
// (the compiler's own register count, no scratch): what a filter WITHOUT Murty work launches as its post kernel
// step after step -- the capped instance needs scratch memory set up for every wave it dispatches and costs 0.2 us more per
// step for nothing.  Both do the same thing with whatever the queue holds.
template <int W, int WAVES_PER_EU>
__global__ __launch_bounds__(64 * W) __attribute__((amdgpu_waves_per_eu(WAVES_PER_EU > 0 ? WAVES_PER_EU : 1, WAVES_PER_EU > 0 ? WAVES_PER_EU : 8)))
void murty_jobs_kernel(MurtyQueue Q, MurtyScratch MS, int *err, double *weight, int N, double *sums,
                                                                         int normalize, ZArg zarg, double *dZ, int nZdoubles, int *hostSeen, int ordered, StepOut SO,
                                                                         StepOrderArg SOrd) {
  // (a fused step carries the measurement set in its kernel arguments; the device copy the next predict reads is written here)
  if (blockIdx.x == 0 && dZ)
    for (int t = threadIdx.x; t < nZdoubles; t += blockDim.x) dZ[t] = zarg.v[t];
  if (SOrd.cost && blockIdx.x >= 1 && blockIdx.x <= STEP_ORDER_CLASSES)     // (every launch has >= 64 workgroups; block-uniform)
    step_cost_order_class(SOrd, (int)blockIdx.x - 1);
  const int nJobs = min(*Q.count, Q.maxJobs);
  if (nJobs == 0) {
    if (blockIdx.x == 0) { step_post_tail(weight, N, sums, normalize, SO.preDiv, SO.collSeq, SO.collNeed, SO.collPost, err); step_post_out(weight, N, err, SO); }
    return;
  }
  __shared__ double sTile[W][MURTY_LDS_N * MURTY_LDS_N];
  __shared__ double sScore[MURTY_N];
  __shared__ double sSum;
  __shared__ int sCtl[8];
  __shared__ unsigned char sPushed[MURTY_N];
#if !defined(MURTY_NO_SPEC)
  __shared__ MurtySpec sSpec;
  MurtySpec *const spec = &sSpec;
#else
  MurtySpec *const spec = nullptr;
#endif
  __shared__ double sHeapSc[MURTY_HEAP_LDS];   // the searching wave's heap, first positions (mheap_*)
  __shared__ short sHeapId[MURTY_HEAP_LDS];
#if !defined(MURTY_NO_SPEC) && !defined(MURTY_NO_SMALL)
  __shared__ double sJobC[HQ_N * HQ_N];   // the job's table for the small form (murty_kbest_async)
  double *const jobC = sJobC;
#else
  double *const jobC = nullptr;
#endif
  // (readfirstlane: tells the compiler the wave index is uniform, so that the whole search compiles to scalar control flow)
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  for (int jq = blockIdx.x; jq < nJobs; jq += gridDim.x) {
    const int j = (ordered && Q.order) ? Q.order[jq] : jq;
    const MurtyJob J = Q.jobs[j];
    const int n = J.nR + J.nC;
    double v = 1.0;
#ifdef RFS_PROFILE
    const long long dbgJob0 = (long long)wall_clock64();
#endif
#if defined(MURTY_ONLY_MIN) || defined(MURTY_ONLY_MAX)   // tuning aid: time one class of jobs alone (the others count as 1.0: WRONG weights)
#ifndef MURTY_ONLY_MIN
#define MURTY_ONLY_MIN 0
#endif
#ifndef MURTY_ONLY_MAX
#define MURTY_ONLY_MAX 64
#endif
    if (n < MURTY_ONLY_MIN || n > MURTY_ONLY_MAX) {
    } else
#endif
    if (n <= 0 || (unsigned)J.particle >= (unsigned)N) {
      // a queue slot its producer reserved but could not fill (partition beyond MURTY_MAXN: the error bit is already set): skipped
    } else if (n > MURTY_N || (int)blockIdx.x >= MS.nArenas) {
      if (threadIdx.x == 0) atomicOr(err, ERRBIT_MURTY);
    } else {
      MurtyArena A;
      murty_carve(MS.arena + (size_t)blockIdx.x * MS.jobBytes, A);
      bool ok;
      v = murty_partition_sum_block<W>(Q.mats + (size_t)j * MURTY_MAXN * MURTY_MAXN, n, J.nR, J.nC, A, ok, sTile[wave], sCtl, &sSum, sScore,
                                                     sPushed, wave, spec, jobC, sHeapId, sHeapSc);
      if (!ok && threadIdx.x == 0) atomicOr(err, ERRBIT_MURTY);
    }
    if (threadIdx.x == 0) Q.results[j] = v;
#ifdef RFS_PROFILE
    if (threadIdx.x == 0 && ((blockIdx.x & 63) == 7 || blockIdx.x >= 1900)) printf("murty job: block %d n %d started at tick %lld, ended at %lld\n", (int)blockIdx.x, n, dbgJob0, (long long)wall_clock64());
#endif
    __syncthreads();   // the control words are reused by the next job
  }
  __shared__ int isLast;
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) isLast = (atomicAdd(Q.count + 1, 1) == (int)gridDim.x - 1);
  __syncthreads();
  if (!isLast) return;
  __threadfence();
  if (threadIdx.x == 0) { Q.count[1] = 0; Q.count[0] = 0; }   // the queue is consumed: empty for the next step
#ifdef RFS_PROFILE
  if (threadIdx.x == 0) {
    int hist[MURTY_N + 1];
    for (int k = 0; k <= MURTY_N; k++) hist[k] = 0;
    for (int q = 0; q < nJobs; q++) { const int n = Q.jobs[q].nR + Q.jobs[q].nC; hist[n > MURTY_N ? MURTY_N : n]++; }
    printf("murty jobs %d; by extended dimension:", nJobs);
    for (int k = 0; k <= MURTY_N; k++) if (hist[k]) printf(" %d:%d", k, hist[k]);
    printf("\n");
  }
#endif
#ifdef RFS_PROFILE
  const long long dbgTail0 = (long long)wall_clock64();
#endif
#if defined(MURTY_WARM_CHECK)
  if (threadIdx.x == 0) printf("murty warm check: %llu child solves compared with the solve from scratch, %llu mismatches (running totals)\n", g_murtyWarmChecks, g_murtyWarmBad);
#endif
  if (threadIdx.x == 0 && hostSeen) *hostSeen = 1;            // (pinned host word: this filter does reach the Murty path -- see murty_launch)
  // Every particle's factors, multiplied in partition (slot) order.  The jobs of a particle are chained through a list first
  // (head per particle in the first job's arena, links in the order array -- both idle by now), so that a particle looks at
  // its own few jobs only: scanning the whole queue per particle was 2.6 ms of a 9 ms launch at configs[4] (1000 x 1918).
  if (Q.order && (size_t)N * sizeof(int) <= MS.jobBytes * (size_t)MS.nArenas) {
    int *head = reinterpret_cast<int *>(MS.arena), *next = Q.order;
    for (int i = threadIdx.x; i < N; i += blockDim.x) head[i] = -1;
    __threadfence();
    __syncthreads();
    for (int q = threadIdx.x; q < nJobs; q += blockDim.x) {
      const int pi = Q.jobs[q].particle;
      if ((unsigned)pi < (unsigned)N && Q.jobs[q].nR + Q.jobs[q].nC > 0) next[q] = atomicExch(&head[pi], q);   // (skip-jobs: see above)
      else next[q] = -1;
    }
    __threadfence();
    __syncthreads();
    for (int i = threadIdx.x; i < N; i += blockDim.x) {
      const int h = __builtin_nontemporal_load(&head[i]);
      if (h < 0) continue;
      int last = -1;
      double w = weight[i];
      while (true) {
        int best = -1, bestSlot = 1 << 30;
        for (int q = h; q >= 0; q = __builtin_nontemporal_load(&next[q])) {
          const int sl = Q.jobs[q].slot;
          if (sl > last && sl < bestSlot) { best = q; bestSlot = sl; }
        }
        if (best < 0) break;
        w *= __builtin_nontemporal_load(&Q.results[best]);
        last = bestSlot;
      }
      weight[i] = w;
    }
  } else {
    for (int i = threadIdx.x; i < N; i += blockDim.x) {
      int last = -1;
      double w = weight[i];
      bool any = false;
      while (true) {
        int best = -1, bestSlot = 1 << 30;
        for (int q = 0; q < nJobs; q++)
          if (Q.jobs[q].particle == i && Q.jobs[q].nR + Q.jobs[q].nC > 0 && Q.jobs[q].slot > last && Q.jobs[q].slot < bestSlot) { best = q; bestSlot = Q.jobs[q].slot; }
        if (best < 0) break;
        w *= __builtin_nontemporal_load(&Q.results[best]);
        last = bestSlot;
        any = true;
      }
      if (any) weight[i] = w;
    }
  }
  __syncthreads();
#ifdef RFS_PROFILE
  if (threadIdx.x == 0) printf("murty tail (last workgroup %d): factors into the weights %lld ticks of 10 ns, ends at tick %lld\n", (int)blockIdx.x, (long long)wall_clock64() - dbgTail0, (long long)wall_clock64());
#endif
  step_post_tail(weight, N, sums, normalize, SO.preDiv, SO.collSeq, SO.collNeed, SO.collPost, err);
  step_post_out(weight, N, err, SO);
}

// Start of a step: measurement set -> device buffer (read by every kernel of the step and by the next predict), Murty
// job counter := 0.
__global__ __launch_bounds__(256) void stage_step_kernel(ZArg z, double *dZ, int nDoubles, int *murtyCount) {
  const int t = threadIdx.x;
  if (t < nDoubles) dZ[t] = z.v[t];
  if (t == 0 && murtyCount) murtyCount[0] = 0;
}

static inline int murty_alloc(MurtyQueue &Q, MurtyScratch &MS, int N) {
  int maxJobs = 4 * N;   // (configs[4] queues 1.9 jobs per particle on average and up to ~2.2 on some seeds; 0.7 MB of arena per resident workgroup)
  if (maxJobs < 256) maxJobs = 256;
  if (maxJobs > 8192) maxJobs = 8192;
  Q.maxJobs = maxJobs;
  MS.jobBytes = murty_job_bytes();
  MS.nArenas = std::min(MURTY_JOB_BLOCKS, maxJobs);
  bool ok = true;
  ok &= hipMalloc(&Q.count, 2 * sizeof(int)) == hipSuccess;
  ok &= hipMalloc(&Q.jobs, (size_t)maxJobs * sizeof(MurtyJob)) == hipSuccess;
  ok &= hipMalloc(&Q.mats, (size_t)maxJobs * MURTY_MAXN * MURTY_MAXN * sizeof(double)) == hipSuccess;
  ok &= hipMalloc(&Q.results, (size_t)maxJobs * sizeof(double)) == hipSuccess;
  ok &= hipMalloc(&Q.order, (size_t)maxJobs * sizeof(int)) == hipSuccess;
  ok &= hipMalloc(&MS.arena, (size_t)MS.nArenas * MS.jobBytes) == hipSuccess;
  if (ok) ok &= hipMemset(Q.count, 0, 2 * sizeof(int)) == hipSuccess;
  if (ok) ok &= hipMemset(Q.jobs, 0, (size_t)maxJobs * sizeof(MurtyJob)) == hipSuccess;   // (an all-zero job is a skip-job)
  return ok ? 0 : 1;
}
static inline void murty_free(MurtyQueue &Q, MurtyScratch &MS) {
  hipFree(Q.count); hipFree(Q.jobs); hipFree(Q.mats); hipFree(Q.results); hipFree(Q.order); hipFree(MS.arena);
  Q = MurtyQueue{};
  MS = MurtyScratch{};
}
// The job count lives on the device: one launch, which is empty when no partition exceeded 8.
// `hostSeen` (pinned, device-visible): set by the kernel once any step has queued Murty jobs.  Until then the launch is the light
// instance (four waves, no register cap, no scratch set-up) on the same grid; a filter that has shown Murty work gets the capped
// eight-wave instance and the job ordering from the next step on.  Correct either way: jobs are strided over whatever grid there is.
static inline int murty_launch(MurtyQueue &Q, MurtyScratch &MS, Buffers &B, hipStream_t stream, double *sums = nullptr, int normalize = 0,
                               const ZArg *za = nullptr, int nZdoubles = 0, int *hostSeen = nullptr, const StepOut &SO = StepOut{nullptr, nullptr, 0, nullptr, nullptr, 0, 0},
                               const StepOrderArg &SOrd = StepOrderArg{nullptr, nullptr, 0, nullptr, 0}) {
  int blocks = std::min(MURTY_JOB_BLOCKS, Q.maxJobs);
  // (RFSGPU_MURTY_FIRST_BLOCKS: grid of the light instance, for A/B runs -- tools/murty_first_step.py)
  static const int firstBlocks = [] { const char *e = getenv("RFSGPU_MURTY_FIRST_BLOCKS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : MURTY_FIRST_BLOCKS; }();
  if (hostSeen && *hostSeen == 0) blocks = std::min(blocks, firstBlocks);
  static const ZArg none{};
  // (a filter that has shown Murty work -- or a caller without the pinned flag -- gets its jobs ordered, longest first; ~5 us,
  //  and only then)
  const int ordered = (Q.order && (!hostSeen || *hostSeen != 0)) ? 1 : 0;
  // (the step-order classes take one workgroup each after block 0; a grid without them leaves the order it has -- a permutation already)
  const StepOrderArg sord = blocks > STEP_ORDER_CLASSES ? SOrd : StepOrderArg{nullptr, nullptr, 0, nullptr, 0};
  if (ordered) murty_order_kernel<<<1, 1024, 0, stream>>>(Q);
  if (hostSeen && *hostSeen == 0)
    murty_jobs_kernel<MURTY_LIGHT_WAVES, 0><<<blocks, 64 * MURTY_LIGHT_WAVES, 0, stream>>>(Q, MS, B.err, B.weight, B.N, sums, normalize, za ? *za : none,
                                                                                         za ? B.Z : nullptr, nZdoubles, hostSeen, ordered, SO, sord);
  else
    murty_jobs_kernel<MURTY_JOB_WAVES, MURTY_WAVES_PER_EU><<<blocks, 64 * MURTY_JOB_WAVES, 0, stream>>>(Q, MS, B.err, B.weight, B.N, sums, normalize,
                                                                                                      za ? *za : none, za ? B.Z : nullptr, nZdoubles, hostSeen, ordered, SO, sord);
  return hipGetLastError() == hipSuccess ? 0 : 1;
}
