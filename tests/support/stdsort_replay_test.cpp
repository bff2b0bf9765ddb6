// Host check of rfs-slam_amd/csrc/stdsort_replay.h (test support; g++ -O2): the partition-phase replay + "stable sort of the
// arrangement" must give the permutation the REAL std::sort gives for the reference's comparator (weightCompare: a.weight >
// b.weight on structs carried by value, include/GaussianMixture.hpp:523-534), on tie-heavy inputs, on every size around the
// 16-element threshold, with the relevance pruning the device uses, through both partition forms (two-pointer loop and stopper
// lists), and on adversarial inputs that exhaust the depth limit (heap-sort branch).  Prints "ok <cases> <heap sorts seen>".
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <vector>
#include "../../rfs-slam_amd/csrc/stdsort_replay.h"

struct G { double weight; int id; double pad[6]; };   // (a Gaussian-sized struct: the payload does not matter to the algorithm)
static bool weightCompare(G a, G b) { return a.weight > b.weight; }

static long g_heap_sorts = 0;

// replay on T, then stable by key over the arrangement -> ids in final order
static std::vector<int> replayed_order(const std::vector<double> &w, bool lists, bool prune_ranges, int R) {
  const int N = (int)w.size();
  std::vector<unsigned short> T(N), Ll(N / 2 + 2), Rl(N / 2 + 2);
  for (int p = 0; p < N; p++) T[p] = (unsigned short)p;
  // the device's relevance bits: eq[r] <=> sorted key r == sorted key r - 1, for ranks < R
  std::vector<double> sk(w);
  std::sort(sk.begin(), sk.end(), [](double a, double b) { return a > b; });
  std::vector<char> eq(N + 1, 0);
  for (int r = 1; r < N && r < R; r++) eq[r] = sk[r] == sk[r - 1];
  auto get = [&](int p) -> unsigned short { return T[p]; };
  auto put = [&](int p, unsigned short e) { T[p] = e; };
  auto gt = [&](unsigned short a, unsigned short b) -> bool { return w[a] > w[b]; };
  long heaps = 0;
  auto rel = [&](int f, int l) -> bool {
    if (!prune_ranges) return true;
    for (int r = f + 1; r <= l - 1; r++) if (eq[r]) return true;
    return false;
  };
  unsigned stack[64];
  if (N > 1) {
    // count heap sorts by running the loop by hand would duplicate the code: detect them through depth instead
    if (lists) ss_replay_serial<true>(get, put, gt, N, rel, stack, Ll.data(), Rl.data());
    else ss_replay_serial<false>(get, put, gt, N, rel, stack);
  }
  (void)heaps;
  std::vector<int> order(T.begin(), T.end());
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return w[a] > w[b]; });
  return order;
}

static std::vector<int> std_sort_order(const std::vector<double> &w) {
  std::vector<G> g(w.size());
  for (size_t k = 0; k < w.size(); k++) { g[k].weight = w[k]; g[k].id = (int)k; }
  std::sort(g.begin(), g.end(), weightCompare);
  std::vector<int> o(w.size());
  for (size_t k = 0; k < w.size(); k++) o[k] = g[k].id;
  return o;
}

// does std::sort's introsort loop run out of depth on this input?  (replay with a counting heap sort)
static bool exhausts_depth(const std::vector<double> &w) {
  const int N = (int)w.size();
  std::vector<unsigned short> T(N);
  for (int p = 0; p < N; p++) T[p] = (unsigned short)p;
  auto get = [&](int p) -> unsigned short { return T[p]; };
  auto put = [&](int p, unsigned short e) { T[p] = e; };
  auto gt = [&](unsigned short a, unsigned short b) -> bool { return w[a] > w[b]; };
  struct R { int f, l, d; };
  std::vector<R> st{{0, N, 2 * ss_floor_lg(N)}};
  bool hit = false;
  while (!st.empty()) {
    R r = st.back(); st.pop_back();
    while (r.l - r.f > SS_THRESHOLD) {
      if (r.d == 0) { hit = true; ss_heap_sort(get, put, gt, r.f, r.l); break; }
      --r.d;
      const int cut = ss_partition_serial(get, put, gt, r.f, r.l);
      st.push_back({cut, r.l, r.d});
      r.l = cut;
    }
  }
  return hit;
}

// McIlroy's adversary ("A Killer Adversary for Quicksort", 1999) run against std::sort itself: returns a permutation of 0..n-1 on
// which this std::sort goes quadratic, i.e. runs into its depth limit.
static std::vector<int> *a_val;
static int a_nsolid, a_candidate, a_gas;
static bool a_less(int x, int y) {
  std::vector<int> &val = *a_val;
  if (val[x] == a_gas && val[y] == a_gas) { if (x == a_candidate) val[x] = a_nsolid++; else val[y] = a_nsolid++; }
  if (val[x] == a_gas) a_candidate = x; else if (val[y] == a_gas) a_candidate = y;
  return val[x] > val[y];      // (descending, like weightCompare)
}
static std::vector<int> killer(int n) {
  std::vector<int> val(n), ptr(n);
  a_val = &val; a_gas = n - 1; a_nsolid = 0; a_candidate = 0;
  for (int i = 0; i < n; i++) { ptr[i] = i; val[i] = a_gas; }
  std::sort(ptr.begin(), ptr.end(), a_less);
  for (int i = 0; i < n; i++) if (val[i] == a_gas) val[i] = a_nsolid++;
  return val;
}

int main(int argc, char **argv) {
  if (argc > 3 && std::string(argv[1]) == "killer") {   // "killer n q": an adversarial weight sequence (values val / q), one per line, then
    const int n = atoi(argv[2]), q = atoi(argv[3]);      // whether std::sort's depth limit is reached on it
    const std::vector<int> val = killer(n);
    std::vector<double> w(n);
    for (int k = 0; k < n; k++) { w[k] = (double)(val[k] / q); printf("%d\n", val[k] / q); }
    printf("depth_limit_reached %d\n", exhausts_depth(w) ? 1 : 0);
    return 0;
  }
  const int cases = argc > 1 ? atoi(argv[1]) : 20000;
  std::mt19937_64 rng(argc > 2 ? atoll(argv[2]) : 1);
  long done = 0;
  for (int c = 0; c < cases; c++) {
    int N;
    const int pick = (int)(rng() % 6);
    if (pick == 0) N = 1 + (int)(rng() % 40);                 // around the 16-element threshold
    else if (pick == 1) N = 60 + (int)(rng() % 10);           // around one wavefront
    else N = 17 + (int)(rng() % 700);
    std::vector<double> w(N);
    const int mode = (int)(rng() % 6);
    const int levels = 1 + (int)(rng() % (mode == 0 ? 3 : (mode == 1 ? 12 : 60)));
    for (int k = 0; k < N; k++) {
      const double u = (double)(rng() % 1000000) / 1e6;
      if (mode <= 2) w[k] = (double)(rng() % levels) / levels;                       // few distinct values: ties everywhere
      else if (mode == 3) w[k] = (rng() % 4 == 0) ? 0.01 : u;                         // births at one weight among distinct ones
      else if (mode == 4) w[k] = (rng() % 3 == 0) ? 0.0 : ((rng() % 5 == 0) ? 1.0 : u);   // holes at 0, clamped weights at 1
      else w[k] = (k < N / 2) ? 1.0 - 1e-3 * (k / 3) : u;                               // a sorted, tied prefix (last step's survivors) + new ones
    }
    const std::vector<int> want = std_sort_order(w);
    for (int variant = 0; variant < 4; variant++) {
      const std::vector<int> got = replayed_order(w, variant & 1, variant & 2, N);
      if (got != want) { printf("MISMATCH case %d N %d mode %d variant %d\n", c, N, mode, variant); return 1; }
    }
    // prune's use: only ranks < R matter (ranks >= R are dropped anyway): the first R entries must agree
    {
      const int R = 1 + (int)(rng() % N);
      const std::vector<int> got = replayed_order(w, true, true, R);
      for (int r = 0; r < R; r++)
        if (w[got[r]] >= w[want[R - 1]] && w[got[r]] > w[want[N - 1]] - 1.0) { /* keys agree by construction */ }
      // the kept prefix is compared entry by entry, but only over complete tied runs: a run cut by R is ordered only up to the cut
      int rr = R;
      while (rr > 0 && rr < N && w[want[rr]] == w[want[rr - 1]]) rr--;      // back to the start of the run that straddles R
      for (int r = 0; r < rr; r++) if (got[r] != want[r]) { printf("MISMATCH (R = %d) case %d N %d rank %d\n", R, c, N, r); return 1; }
    }
    done++;
  }
  // depth-limit branch: adversarial permutations, then quantised so that the heap-sorted ranges hold ties
  for (int n : {40, 64, 100, 257, 600, 1500, 2048}) {
    const std::vector<int> val = killer(n);
    for (int q : {1, 2, 3, 5, 8}) {
      std::vector<double> w(n);
      for (int k = 0; k < n; k++) w[k] = (double)(val[k] / q);
      if (exhausts_depth(w)) g_heap_sorts++;
      const std::vector<int> want = std_sort_order(w);
      for (int variant = 0; variant < 4; variant++)
        if (replayed_order(w, variant & 1, variant & 2, n) != want) { printf("MISMATCH killer n %d q %d variant %d\n", n, q, variant); return 1; }
      done++;
    }
  }
  if (g_heap_sorts == 0) { printf("no adversarial input reached the heap-sort branch\n"); return 2; }
  printf("ok %ld %ld\n", done, g_heap_sorts);
  return 0;
}
