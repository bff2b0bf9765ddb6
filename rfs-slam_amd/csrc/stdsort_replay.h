// stdsort_replay.h -- the order std::sort leaves EQUAL keys in, reproduced on the device.
//
// GaussianMixture::sortByWeight (reference include/GaussianMixture.hpp:523-534) is `std::sort(gList_.begin(), gList_.end(),
// weightCompare)` with weightCompare(a, b) = a.weight > b.weight: an UNSTABLE sort, so Gaussians of exactly equal weight (every
// birth Gaussian enters with birthGaussianWeight, RBPHDFilter.hpp:1000-1084; merged-away entries all carry weight 0; the
// near-limit heuristic clamps weights to exactly 1, :692-703) come out in an order that is a property of the sort algorithm, and
// that order is observable: it picks the evaluation points among tied weights (:747-761), it is the order the greedy merge walks
// (GaussianMixture.hpp:394-416) and the order in which the next update appends its new Gaussians.  On the Victoria Park extract
// the filter's weights differ by factors between the two orders (tools/tie_order_study.py), so "ties by index" is not the
// reference's result.
//
// What is reproduced is libstdc++'s std::sort (bits/stl_algo.h: __sort -> __introsort_loop + __final_insertion_sort; unchanged
// from GCC 4.9 to 13), the standard library of the reference's only supported toolchain (CMakeLists.txt: GNU flags):
//   * __introsort_loop(first, last, depth = 2 floor(lg n)): while the range holds more than 16 elements: (depth exhausted: heap-
//     sort the range and stop;) move the median of {first + 1, middle, last - 1} to `first`, Hoare-partition (first, last)
//     around it (__unguarded_partition), recurse into the right part, continue with the left part;
//   * __final_insertion_sort: an insertion sort of the whole array whose comparisons are strict, i.e. it is STABLE with respect to the
//     arrangement the loop left.
// Hence   std::sort(a) == stable_sort(arrangement after the partition phase),   and equal keys end up in the order of their
// positions after the partition phase.  The device already has the stable order of the ORIGINAL arrangement (its rank sorts
// break ties by index); what this header adds is the partition phase replayed on an array T of 32-bit words, one per position
// (word = (group << 16) | stable rank; group = the stable rank of the first member of the entry's run of equal keys, i.e. an
// order- and equality-preserving 16-bit image of the key, so a comparison is one LDS read), and a fix-up that reorders every run
// of tied ranks by position in T.
//   * Only comparisons are replayed: no key is moved, no arithmetic happens, the result is exact by construction and is checked
//     against the real std::sort on the host (tests/test_stdsort_replay.py, the serial form below compiled with g++) and against
//     the oracle's std::sort on the device (every GPU parity test now runs the oracle in its reference mode).
//   * A sub-range whose final ranks hold no two equal keys (that matter to the caller) is not followed further: elements never
//     leave their sub-range, so its internal arrangement cannot influence the order of tied keys.  In a running filter the ties
//     sit at the birth weight and at 0, so after one or two partitions most of the array is dropped.
//   * A mixture of <= 16 entries, or without equal keys, needs nothing: std::sort is then the stable order.
// One wavefront replays a partition step in parallel (ss_partition_wave): the stoppers of the two scans are enumerated with
// ballots -- L[k] = k-th position from the left whose key is <= the pivot's, R[k] = k-th from the right whose key is >= it --
// the scan swaps (L[k], R[k]) exactly while L[k] < R[k], K swaps in all, all disjoint, and returns
// cut = min(L[K], R[K-1]) (L[0] when K = 0); see the derivation at ss_partition_lists, which the host test runs against the
// two-pointer loop.  What the correction costs is the chain of dependent LDS round trips that one wave walks (3 + one per 256
// positions per partition step); in the fused 2-D step it runs on wave 0 beside the other waves' intensity strand (weighting.h).
#pragma once
#if defined(__HIPCC__)
#define SS_HD __host__ __device__
#else
#define SS_HD
#endif

#define SS_THRESHOLD 16   // libstdc++ _S_threshold

SS_HD inline int ss_floor_lg(int n) { int k = 0; while (n > 1) { n >>= 1; k++; } return k; }   // std::__lg

// ---- serial form (one thread): the reference for the host test, the device's fallback, and the depth-limit heap sort --------
// get(p) -> entry at position p; put(p, e); gt(a, b) <=> key of entry a > key of entry b  (= weightCompare).

// __adjust_heap + __push_heap (bits/stl_heap.h) on positions f + [0, len)
template <class Get, class Put, class Gt>
SS_HD inline void ss_adjust_heap(Get get, Put put, Gt gt, const int f, int hole, const int len, const decltype(get(0)) value) {
  const int top = hole;
  int child = hole;
  while (child < (len - 1) / 2) {
    child = 2 * (child + 1);
    if (gt(get(f + child), get(f + child - 1))) child--;
    put(f + hole, get(f + child));
    hole = child;
  }
  if ((len & 1) == 0 && child == (len - 2) / 2) {
    child = 2 * (child + 1);
    put(f + hole, get(f + child - 1));
    hole = child - 1;
  }
  int parent = (hole - 1) / 2;
  while (hole > top && gt(get(f + parent), value)) {
    put(f + hole, get(f + parent));
    hole = parent;
    parent = (hole - 1) / 2;
  }
  put(f + hole, value);
}
// __partial_sort(first, last, last): __heap_select with an empty tail (= __make_heap) + __sort_heap
template <class Get, class Put, class Gt>
SS_HD inline void ss_heap_sort(Get get, Put put, Gt gt, const int f, const int l) {
  const int len = l - f;
  if (len < 2) return;
  for (int parent = (len - 2) / 2;; parent--) {
    ss_adjust_heap(get, put, gt, f, parent, len, get(f + parent));
    if (parent == 0) break;
  }
  for (int last = l; last - f > 1;) {
    --last;
    const auto value = get(last);
    put(last, get(f));
    ss_adjust_heap(get, put, gt, f, 0, last - f, value);
  }
}
// __move_median_to_first(first, first + 1, mid, last - 1): which of the three positions is swapped with `first`
template <class Get, class Gt>
SS_HD inline int ss_median_position(Get get, Gt gt, const int f, const int l) {
  const int a = f + 1, b = f + (l - f) / 2, c = l - 1;
  const auto ea = get(a), eb = get(b), ec = get(c);
  if (gt(ea, eb)) {
    if (gt(eb, ec)) return b;
    if (gt(ea, ec)) return c;
    return a;
  }
  if (gt(ea, ec)) return a;
  if (gt(eb, ec)) return c;
  return b;
}
// __unguarded_partition_pivot, the two-pointer loop as written
template <class Get, class Put, class Gt>
SS_HD inline int ss_partition_serial(Get get, Put put, Gt gt, const int f, const int l) {
  const int s = ss_median_position(get, gt, f, l);
  { const auto t = get(f); put(f, get(s)); put(s, t); }
  const auto pivot = get(f);
  int first = f + 1, last = l;
  while (true) {
    while (gt(get(first), pivot)) ++first;
    --last;
    while (gt(pivot, get(last))) --last;
    if (!(first < last)) return first;
    { const auto t = get(first); put(first, get(last)); put(last, t); }
    ++first;
  }
}
// The same partition step from the two stopper lists (the form the wavefront executes), serial, for the host test:
//   L = positions p in (f, l), ascending, with !gt(T[p], pivot)   (where the left scan can stop),
//   R = positions p in (f, l), descending, with !gt(pivot, T[p])  (where the right scan can stop).
// Induction over the loop's trips: before trip k the array differs from the initial one only at L[0..k) and R[0..k), the left
// pointer stands at L[k-1] + 1 and the right one at R[k-1].  The left scan stops at L[k] if L[k] < R[k-1] (untouched
// positions in between hold no stopper before L[k]), else at R[k-1] (which now holds a left stopper); the right scan stops at
// R[k] if R[k] > L[k-1], else at L[k-1].  A swap happens iff left < right, which in every case is L[k] < R[k]; L ascends and R
// descends, so the swaps are k = 0 .. K-1 with K = #{k : L[k] < R[k]}, they touch pairwise distinct positions whose contents
// are the initial ones, and the loop returns cut = L[K] if L[K] < R[K-1] (or K == 0) else R[K-1].  (L[0], R[0] exist: the two
// non-median elements of the median-of-three are sentinels for the unguarded scans.)
template <class Get, class Put, class Gt>
SS_HD inline int ss_partition_lists(Get get, Put put, Gt gt, const int f, const int l, unsigned short *Ll, unsigned short *Rl) {
  const int s = ss_median_position(get, gt, f, l);
  { const auto t = get(f); put(f, get(s)); put(s, t); }
  const auto pivot = get(f);
  // (the lists are kept to their first LC = (l - f) / 2 + 1 entries, as on the device: K <= (l - f - 1) / 2 < LC)
  const int LC = (l - f) / 2 + 1;
  int nL = 0, nR = 0;
  for (int p = f + 1; p < l && nL < LC; p++) if (!gt(get(p), pivot)) Ll[nL++] = (unsigned short)p;
  for (int p = l - 1; p > f && nR < LC; p--) if (!gt(pivot, get(p))) Rl[nR++] = (unsigned short)p;
  int K = 0;
  while (K < nL && K < nR && Ll[K] < Rl[K]) {
    const auto t = get(Ll[K]); put(Ll[K], get(Rl[K])); put(Rl[K], t);
    K++;
  }
  if (K == 0) return Ll[0];
  return (K < nL && Ll[K] < Rl[K - 1]) ? Ll[K] : Rl[K - 1];
}

// Ranges waiting on the replay's stack: first (11 bits) | last (12 bits) << 11 | depth << 23  (n <= 2048)
SS_HD inline unsigned ss_pack_range(int f, int l, int d) { return (unsigned)f | ((unsigned)l << 11) | ((unsigned)d << 23); }

// __introsort_loop on T[0, N), ranges in any order (they are disjoint); relevant(f, l) == false drops a range (see the header).
// stack: room for 2 floor(lg N) + 2 words.  USE_LISTS: partition through ss_partition_lists (host test of that form).
template <bool USE_LISTS, class Get, class Put, class Gt, class Rel>
SS_HD inline void ss_replay_serial(Get get, Put put, Gt gt, const int N, Rel relevant, unsigned *stack, unsigned short *Ll = nullptr,
                                   unsigned short *Rl = nullptr) {
  int sp = 0;
  stack[sp++] = ss_pack_range(0, N, 2 * ss_floor_lg(N));
  while (sp > 0) {
    const unsigned w = stack[--sp];
    const int f = (int)(w & 0x7ffu);
    int l = (int)((w >> 11) & 0xfffu), d = (int)(w >> 23);
    while (l - f > SS_THRESHOLD && relevant(f, l)) {
      if (d == 0) { ss_heap_sort(get, put, gt, f, l); break; }
      --d;
      const int cut = USE_LISTS ? ss_partition_lists(get, put, gt, f, l, Ll, Rl) : ss_partition_serial(get, put, gt, f, l);
      if (l - cut > SS_THRESHOLD) stack[sp++] = ss_pack_range(cut, l, d);
      l = cut;
    }
  }
}

#if defined(__HIPCC__)
// ---- wavefront form ------------------------------------------------------------------------------------------------------------
// The replay works on 32-bit words  (group << 16) | rank : `rank` = the entry's stable rank, `group` = the stable rank of the first member of its run of
// equal keys -- an order- and equality-preserving 16-bit image of the key (smaller group = larger weight) -- so that one LDS read
// brings what a comparison needs:  gt(a, b)  <=>  group(a) < group(b).  (r04, first version: a u16 index array with the keys looked
// up behind it -- two dependent LDS reads per comparison, the pivot re-read after the median swap, the cut read back from the lists:
// 8 + 2 chunks LDS round trips per partition step, one wave, everything else waiting: +10.5 us per step at C2b, where every
// mixture holds tied birth weights.  This form: 3 + chunks.)
struct StdSortScratch {
  unsigned *T;           // [N] arrangement: word at position p
  unsigned short *pos;   // [N * posStride] position, after the partition phase, of the entry that the stable order ranks r-th (index: r)
  int posStride;         // 1, or 2 when the array lives in the unused upper halves of an int[] permutation array
  unsigned short *Ll;    // [LC] left stoppers, ascending   (LC = N / 2 + 1; nullptr: no room -> lane 0 replays serially)
  unsigned short *Rl;    // [LC] right stoppers, descending
  unsigned long long *eq;  // [ceil(N / 64)] bit r of word r / 64: the keys of sorted ranks r - 1 and r are equal (and matter)
  unsigned *stack;       // [>= 26]
};
__host__ __device__ inline int ss_list_cap(int N) { return N / 2 + 1; }

// the part ss_carve cannot do without (a caller that provides the position array itself): eq words, stack, T
__host__ __device__ inline size_t ss_must_bytes(int N) { return (size_t)((N + 63) >> 6) * 8 + 112 + (size_t)N * 4 + 4; }
// Lay the scratch out in a byte range [buf, buf + bytes) (8-byte aligned): eq words | stack | T | pos (when ownPos; otherwise the
// caller has set S.pos / S.posStride) | the two stopper lists if they still fit (else the replay runs serially on lane 0).
// Returns false when even the mandatory part does not fit.
__device__ __forceinline__ bool ss_carve(StdSortScratch &S, unsigned char *buf, size_t bytes, const int N, const bool ownPos) {
  const size_t eqB = (size_t)((N + 63) >> 6) * 8, stackB = 112, tB = (size_t)N * 4 + ((N & 1) ? 4 : 0), posB = ((size_t)N * 2 + 7) & ~(size_t)7;
  const size_t must = eqB + stackB + tB + (ownPos ? posB : 0);
  if (must > bytes) return false;
  unsigned char *p = buf;
  S.eq = reinterpret_cast<unsigned long long *>(p); p += eqB;
  S.stack = reinterpret_cast<unsigned *>(p); p += stackB;
  S.T = reinterpret_cast<unsigned *>(p); p += tB;
  if (ownPos) { S.pos = reinterpret_cast<unsigned short *>(p); S.posStride = 1; p += posB; }
  const size_t listB = ((size_t)ss_list_cap(N) * 2 + 7) & ~(size_t)7;
  if (must + 2 * listB <= bytes) {
    S.Ll = reinterpret_cast<unsigned short *>(p); p += listB;
    S.Rl = reinterpret_cast<unsigned short *>(p);
  } else {
    S.Ll = nullptr; S.Rl = nullptr;
  }
  return true;
}

// relevant(f, l): some rank in (f, l) carries an eq bit, i.e. two equal keys end up inside [f, l).  eqReg: lane w holds eq word w
// (loaded once per replay: no LDS read here).  Uniform; all lanes call it.
__device__ __forceinline__ bool ss_relevant(const unsigned long long eqReg, const int f, const int l, const int lane) {
  const int lo = f + 1, hi = l - 1;          // ranks f + 1 .. l - 1
  unsigned long long m = 0ull;
  if (lo <= hi && lane >= (lo >> 6) && lane <= (hi >> 6)) {
    m = eqReg;
    if (lane == (lo >> 6)) m &= ~0ull << (lo & 63);
    if (lane == (hi >> 6)) m &= ~0ull >> (63 - (hi & 63));
  }
  return __ballot(m != 0ull) != 0ull;
}

// One partition step (__unguarded_partition_pivot) on T[f, l) by the calling wavefront; returns the cut (uniform).
// Dependent LDS round trips: the median's four words; one per 64 positions from either end; the stopper lists; the swapped words.
__device__ __forceinline__ int ss_partition_wave(const StdSortScratch &S, const int f, const int l, const int lane) {
  unsigned *const T = S.T;
  const int LC = ss_list_cap(l - f);
  // median of three to the front: the four words in one trip; the swap is written by lane 0 and NOT waited for -- the scans
  // below substitute the one word they could read stale (position s), and a barrier precedes the next reader of T
  const int a = f + 1, b = f + (l - f) / 2, c = l - 1;
  const unsigned wf = T[f], wa = T[a], wb = T[b], wc = T[c];
  const unsigned ga = wa >> 16, gb = wb >> 16, gc = wc >> 16;      // gt(x, y) <=> group(x) < group(y)
  int s;
  if (ga < gb) s = (gb < gc) ? b : ((ga < gc) ? c : a);
  else s = (ga < gc) ? a : ((gb < gc) ? c : b);
  const unsigned wp = (s == a) ? wa : ((s == b) ? wb : wc);        // the pivot's word, now at position f
  if (lane == 0) { T[f] = wp; T[s] = wf; }
  const unsigned gp = wp >> 16;
  // stoppers from both ends at once: trip t looks at positions f + 1 + (64 t + lane) and l - 1 - (64 t + lane).  Four trips'
  // words are loaded together (eight independent LDS reads in flight, with the median's four above: one latency for a range of
  // up to 256), then enumerated from registers.
  int nL = 0, nR = 0;
  const int n1 = l - f - 1;                  // positions f + 1 .. l - 1
  for (int t0 = 0; t0 < n1 && (nL < LC || nR < LC); t0 += 256) {
    unsigned wx[4], wy[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int q = t0 + 64 * u + lane;
      const bool in = q < n1;
      wx[u] = in ? T[f + 1 + q] : 0u;
      wy[u] = in ? T[l - 1 - q] : 0u;
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      if (t0 + 64 * u >= n1) break;          // (uniform)
      const int q = t0 + 64 * u + lane;
      const bool in = q < n1;
      const int pa = f + 1 + q, pd = l - 1 - q;
      const unsigned x = (pa == s) ? wf : wx[u], y = (pd == s) ? wf : wy[u];
      const bool sl = in && !((x >> 16) < gp), sr = in && !(gp < (y >> 16));   // left stops at !gt(x, pivot), right at !gt(pivot, y)
      const unsigned long long ml = __ballot(sl), mr = __ballot(sr);
      const unsigned long long below = (1ull << lane) - 1ull;
      const int il = nL + __popcll(ml & below), ir = nR + __popcll(mr & below);
      if (sl && il < LC) S.Ll[il] = (unsigned short)pa;
      if (sr && ir < LC) S.Rl[ir] = (unsigned short)pd;
      nL += __popcll(ml);
      nR += __popcll(mr);
    }
  }
  wave_sync();
  nL = nL < LC ? nL : LC;
  nR = nR < LC ? nR : LC;
  // the swaps: k < K  <=>  L[k] < R[k]  (a prefix; beyond the lists' capacity L[k] > R[k] anyway: 2 k >= l - f - 2).  The cut comes
  // out of the registers of the last trip: R[K - 1] sits on lane K - 1 - k0, L[K] on lane K - k0 (or in the list's next entry).
  const int nMin = nL < nR ? nL : nR;
  int K = 0, cut = -1;
  for (int k0 = 0; k0 < nMin || k0 == 0; k0 += 64) {
    const int k = k0 + lane;
    int pl = 0x7fffffff, pr = -1;
    if (k < nL) pl = S.Ll[k];
    if (k < nR) pr = S.Rl[k];
    const bool sw = (k < nMin) && pl < pr;
    const unsigned long long ms = __ballot(sw);
    if (sw) {
      const unsigned el = T[pl], er = T[pr];
      T[pl] = er;
      T[pr] = el;
    }
    const int cnt = __popcll(ms);
    K = k0 + cnt;
    if (cnt < 64) {                          // the prefix ends inside this trip: lane cnt holds L[K] (if it exists), lane cnt - 1 holds R[K - 1]
      const int lk = __builtin_amdgcn_readlane(pl, cnt);                       // 0x7fffffff: no such stopper
      int rk1 = (cnt > 0) ? __builtin_amdgcn_readlane(pr, cnt - 1) : -1;
      if (cnt == 0 && k0 > 0) rk1 = S.Rl[K - 1];                                // (the prefix ended exactly at the trip boundary)
      cut = (K == 0) ? lk : ((lk < rk1) ? lk : rk1);
      break;
    }
  }
  if (cut < 0) {                             // every one of the nMin pairs swapped and nMin is a multiple of 64
    const int rk1 = S.Rl[K - 1];
    const int lk = (K < nL) ? (int)S.Ll[K] : 0x7fffffff;
    cut = (lk < rk1) ? lk : rk1;
  }
  wave_sync();
  return cut;
}

// The whole partition phase on T[0, N) (already holding the words of the initial arrangement) by ONE wavefront, then pos[].
__device__ __forceinline__ void ss_replay_wave(const StdSortScratch &S, const int N, const int lane) {
  unsigned *const T = S.T;
  auto get = [&](int p) -> unsigned { return T[p]; };
  auto put = [&](int p, unsigned e) { T[p] = e; };
  auto gt = [&](unsigned x, unsigned y) -> bool { return (x >> 16) < (y >> 16); };
  if (S.Ll == nullptr) {   // no room for the stopper lists: lane 0 alone, the two-pointer loop (relevance from the same bits)
    if (lane == 0) {
      auto rel = [&](int f, int l) -> bool {
        for (int r = f + 1; r <= l - 1; r++) if ((S.eq[r >> 6] >> (r & 63)) & 1ull) return true;
        return false;
      };
      ss_replay_serial<false>(get, put, gt, N, rel, S.stack);
    }
  } else {
    const int nWords = (N + 63) >> 6;
    const unsigned long long eqReg = (lane < nWords) ? S.eq[lane] : 0ull;
    int sp = 0;
    unsigned top = ss_pack_range(0, N, 2 * ss_floor_lg(N));   // (the stack's top entry rides in a register)
    bool have = true;
    while (have) {
      const int f = (int)(top & 0x7ffu);
      int l = (int)((top >> 11) & 0xfffu), d = (int)(top >> 23);
      have = false;
      while (l - f > SS_THRESHOLD && ss_relevant(eqReg, f, l, lane)) {
        if (d == 0) {
          if (lane == 0) ss_heap_sort(get, put, gt, f, l);
          wave_sync();
          break;
        }
        --d;
        const int cut = ss_partition_wave(S, f, l, lane);
        if (l - cut > SS_THRESHOLD && ss_relevant(eqReg, cut, l, lane)) {
          if (lane == 0) S.stack[sp] = ss_pack_range(cut, l, d);
          sp++;
        }
        l = cut;
      }
      if (sp > 0) {
        wave_sync();
        --sp;
        top = S.stack[sp];
        have = true;
      }
    }
  }
  wave_sync();
  for (int p = lane; p < N; p += 64) S.pos[(T[p] & 0xffffu) * S.posStride] = (unsigned short)p;   // by rank (the word's low half)
  wave_sync();
}

// eq bits of 64 sorted ranks [c0, c0 + 64) by the calling wavefront: rank r is flagged when r < R (ranks that matter), r > 0 and
// key(rank r) == key(rank r - 1).  entryAt(r): entry at stable rank r.  In the same pass the ranks' words of the initial arrangement
// are written (position = entry index): group = start of the rank's run, found from the ballot in registers; a rank whose run began
// in an earlier chunk gets group 0xffff for now and is fixed once all eq words are known (ss_fix_groups).  Returns the word.
template <class KeyAt, class EntryAt>
__device__ __forceinline__ unsigned long long ss_eq_word_and_words(KeyAt keyAt, EntryAt entryAt, unsigned *T, const int c0, const int R, const int lane) {
  const int r = c0 + lane;
  bool e = false;
  int ent = 0;
  if (r < R) {
    ent = entryAt(r);
    if (r > 0) e = keyAt(ent) == keyAt(entryAt(r - 1));
  }
  const unsigned long long w = __ballot(e);
  if (r < R) {
    const unsigned long long z = ~w & (~0ull >> (63 - lane));                  // run starts at positions <= lane
    const unsigned g = z ? (unsigned)(c0 + 63 - __builtin_clzll(z)) : 0xffffu;
    T[ent] = (g << 16) | (unsigned)r;
  }
  return w;
}

// start of the run of equal keys that rank r belongs to (r itself when its key differs from its predecessor's): the highest
// clear eq bit at or below r -- one LDS read unless the run crosses a word boundary (a loop over the members would be one
// dependent read per member)
__device__ __forceinline__ int ss_run_start(const unsigned long long *eq, int r) {
  int w = r >> 6;
  unsigned long long z = ~eq[w] & (~0ull >> (63 - (r & 63)));     // clear bits at positions <= r
  while (z == 0ull) { w--; z = ~eq[w]; }                           // (bit 0 of word 0 is never set: the loop ends)
  return 64 * w + 63 - __builtin_clzll(z);
}
// one past the last rank of the run that rank r belongs to: the lowest clear eq bit above r (limit: R)
__device__ __forceinline__ int ss_run_end(const unsigned long long *eq, int r, const int R) {
  int q = r + 1;
  while (q < R) {
    const unsigned long long z = ~eq[q >> 6] & (~0ull << (q & 63));   // clear bits at positions >= q
    if (z != 0ull) { q = 64 * (q >> 6) + __builtin_ctzll(z); break; }
    q = 64 * ((q >> 6) + 1);
  }
  return q < R ? q : R;
}

// New rank of the entry at stable rank r: unchanged unless r lies in a run of equal keys, then run start + the number of run
// members whose position after the partition phase is smaller.  rp(q) = position of the entry at rank q (written by the replay
// itself: the words carry the rank), four per trip.
template <class RpAt>
__device__ __forceinline__ int ss_fixed_rank(const unsigned long long *eq, RpAt rp, const int r, const int R) {
  const bool mine = (eq[r >> 6] >> (r & 63)) & 1ull, next = (r + 1 < R) && ((eq[(r + 1) >> 6] >> ((r + 1) & 63)) & 1ull);
  if (!mine && !next) return r;
  const int r0 = ss_run_start(eq, r), r1 = ss_run_end(eq, r, R);
  const unsigned short myPos = rp(r);
  int ahead = 0, q = r0;
  for (; q + 4 <= r1; q += 4) {
    const unsigned short p0 = rp(q), p1 = rp(q + 1), p2 = rp(q + 2), p3 = rp(q + 3);
    ahead += (p0 < myPos) + (p1 < myPos) + (p2 < myPos) + (p3 < myPos);
  }
  for (; q < r1; q++) ahead += (rp(q) < myPos) ? 1 : 0;
  return r0 + ahead;
}

// The whole correction for one mixture, by all NT = WPP * 64 threads of the workgroup that owns it:
//   keyAt(e)      key of entry e (the caller maps merged-away entries to the reference's weight 0);
//   entryAt(r)    entry at rank r of the STABLE order (ties by index), r < R;  setEntry(r, e) installs the corrected order;
//   N             entries std::sort sees (the whole gList_);  R <= N: the ranks whose order matters (prune: the survivors).
// Entries that are NOT among the R leading ranks (prune: everything below the threshold, holes) need a word in T too: they all
// compare below the survivors and their mutual order is irrelevant EXCEPT that equal keys must compare equal and unequal ones in
// order -- their group is found by ranking them among themselves (keyRankOfRest: see the prune call sites; the weighting sort has
// R == N and never needs it).  restGroup(T) writes T[e] = (g << 16) | g for every entry e outside the leading ranks, g = any value in
// [R, N) that is monotone in the key (larger key -> smaller value, equal keys -> equal values).
// Returns false (nothing to do) for N <= 16 or when no two of the R leading keys are equal.  The caller's barrier separates the
// steps; S.T is reused as the staging area of the new order once pos[] exists.
struct SsNoHook { __device__ void operator()() const {} };
// afterEq: called (by every calling thread) once the eq words are complete and at least one is non-zero, before the replay.
template <int WPP, class KeyAt, class EntryAt, class SetEntry, class RestGroup, class Sync, class Hook = SsNoHook>
__device__ __forceinline__ bool ss_correct_tie_order(KeyAt keyAt, EntryAt entryAt, SetEntry setEntry, RestGroup restGroup, const int N, const int R,
                                                     const StdSortScratch &S, const int tid, Sync block_sync, const bool maybeTied = true,
                                                     Hook afterEq = Hook()) {
  constexpr int NT = WPP * 64;
#ifdef SS_TIE_ORDER_OFF   // (tuning aid: what the correction costs -- ties by index, NOT the reference's order)
  return false;
#endif
  // maybeTied (uniform over the workgroup): false when the caller's rank sort has already seen that no two keys are equal
  if (N <= SS_THRESHOLD || R < 2 || !maybeTied) return false;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int nWords = (N + 63) >> 6;
  for (int c = wave; c < nWords; c += WPP) {
    const unsigned long long w = ss_eq_word_and_words(keyAt, entryAt, S.T, 64 * c, R, lane);
    if (lane == 0) S.eq[c] = w;
  }
  if (R < N) restGroup(S.T);                 // (prune: the words of the entries below the survivors)
  block_sync();
  if (__ballot(lane < nWords && S.eq[lane < nWords ? lane : 0] != 0ull) == 0ull) return false;   // (uniform over the workgroup)
  afterEq();
  if (wave == 0) {
    // runs that cross a 64-rank boundary: their later members could not see the start in their own chunk
    for (int c0 = 64; c0 < R; c0 += 64) {
      const int r = c0 + lane;
      const unsigned long long wq = S.eq[c0 >> 6];
      if ((wq & 1ull) == 0ull) continue;                                        // (uniform: the chunk does not begin inside a run)
      if (r < R && (~wq & (~0ull >> (63 - lane))) == 0ull) {                    // no run start at or below this lane: still in that run
        const int ent = entryAt(r);
        S.T[ent] = ((unsigned)ss_run_start(S.eq, r) << 16) | (unsigned)r;
      }
    }
    wave_sync();
    ss_replay_wave(S, N, lane);
  }
  block_sync();
  unsigned short *stage = reinterpret_cast<unsigned short *>(S.T);     // (T is dead once the positions are known)
  auto rpAt = [&](int q) -> unsigned short { return S.pos[q * S.posStride]; };
  for (int r = tid; r < R; r += NT) stage[ss_fixed_rank(S.eq, rpAt, r, R)] = (unsigned short)entryAt(r);
  block_sync();
  for (int r = tid; r < R; r += NT) setEntry(r, stage[r]);
  block_sync();
  return true;
}
#endif  // __HIPCC__
