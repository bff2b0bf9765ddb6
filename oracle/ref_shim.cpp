/*
 * ref_shim.cpp -- thin extern "C" wrapper around the two reference classes that compile from their
 * own sources without Eigen/Boost (probed: g++ -c on the unmodified files succeeds):
 *   rfs::PermutationLexicographic   (src/PermutationLexicographic.cpp)
 *   rfs::BruteForceLinearAssignment (src/BruteForceAssignment.cpp) -- the reference's own checker for
 *     Murty in src/examples/linearAssignment_MurtyAlgorithm.cpp:99-130.
 * TEST INFRASTRUCTURE ONLY.  Built by oracle/Makefile into oracle/_ref/librfs_ref.so straight from
 * /root/reference (sources are compiled where they lie; nothing is copied into this repo).
 * Everything else in the reference needs Eigen3/Boost, which this image lacks => unbuildable here.
 */
#include <cstring>
#include "PermutationLexicographic.hpp"
#include "BruteForceAssignment.hpp"

extern "C" {

/* all permutations produced by PermutationLexicographic(nM, nZ, includeClutter=true).next() */
int rfsref_permlex_all(unsigned nM, unsigned nZ, unsigned *out, int max_perm) {
  rfs::PermutationLexicographic pl(nM, nZ, true);
  unsigned *o = new unsigned[nM + nZ];
  int k = 0;
  while (pl.next(o) != 0) {
    if (k < max_perm) memcpy(out + (size_t)k * (nM + nZ), o, (nM + nZ) * sizeof(unsigned));
    k++;
  }
  delete[] o;
  return k;
}

/* ranked (best first) scores + assignments of every n! assignment of the n x n row-major score matrix C */
int rfsref_bruteforce(const double *C, int n, int kmax, double *scores, unsigned *assignments) {
  double **rows = new double *[n];
  double *copy = new double[(size_t)n * n];
  memcpy(copy, C, sizeof(double) * n * n);
  for (int i = 0; i < n; i++) rows[i] = copy + (size_t)i * n;
  rfs::BruteForceLinearAssignment bf;
  unsigned **a;
  double *s;
  unsigned cnt = bf.run(rows, n, a, s, true);
  int k = 0;
  for (; k < (int)cnt && k < kmax; k++) {
    scores[k] = s[k];
    if (assignments) memcpy(assignments + (size_t)k * n, a[k], n * sizeof(unsigned));
  }
  delete[] rows;
  delete[] copy;
  return (int)cnt;
}
}
