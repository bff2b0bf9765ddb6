// merge_prune.h -- gm_merge: GaussianMixture::merge (reference include/GaussianMixture.hpp:394-475) and
// gm_prune: GaussianMixture::prune + sortByWeight (:477-534).
//
// gm_merge keeps the reference's sequential-greedy semantics exactly: for i ascending, j ascending > i, j is
// absorbed into i as soon as md2_i(j) <= t^2 or md2_j(i) <= t^2, and i's mean/covariance change before j+1 is
// tested.  Wave-parallel form: the whole mixture (mean, packed covariance, its inverse, weight) is staged in
// LDS; for the current i the 64 lanes test 64 candidates j at once against i's CURRENT state, the lowest
// passing lane is merged (ballot + ctz), and only lanes above it are re-tested against the new state.
// Holes (absorbed Gaussians; landmark == NULL, weight 0 in the reference) are written back with weight -1 so
// that gm_prune drops them.
//
// gm_prune: rank-sort the survivors (w >= threshold) by (weight desc, index asc) and compact them into the
// other slab.  The reference keeps exactly the sorted prefix with w >= t (binary search + linear walk).
#pragma once
#include "common.h"
#include "weighting.h"  // wave_sync

#define MERGE_LDS_DOUBLES_PER_ENTRY 9

__host__ __device__ inline size_t merge_lds_bytes_per_wave(int cap) { return (size_t)cap * MERGE_LDS_DOUBLES_PER_ENTRY * 8; }

template <int WPB>
__global__ __launch_bounds__(WPB * 64) void gm_merge_kernel(Buffers B, Params P, int cur) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const int wave = threadIdx.x >> 6;
  const int lane = threadIdx.x & 63;
  const int i = __builtin_amdgcn_readfirstlane(blockIdx.x * WPB + wave);
  if (i >= B.N) return;
  const int cap = B.cap;
  double *base = reinterpret_cast<double *>(smem_raw) + (size_t)wave * cap * MERGE_LDS_DOUBLES_PER_ENTRY;
  double *sMX = base, *sMY = base + cap, *sXX = base + 2 * cap, *sXY = base + 3 * cap, *sYY = base + 4 * cap;
  double *sW = base + 5 * cap, *sI00 = base + 6 * cap, *sI01 = base + 7 * cap, *sI11 = base + 8 * cap;

  const int N = B.count[i];
  double *slab = B.slab[cur];
  double *pW = plane(slab, cap, i, PL_W), *pMX = plane(slab, cap, i, PL_MX), *pMY = plane(slab, cap, i, PL_MY);
  double *pSXX = plane(slab, cap, i, PL_SXX), *pSXY = plane(slab, cap, i, PL_SXY), *pSYY = plane(slab, cap, i, PL_SYY);

  // stage; hole flags live in per-lane registers: bit s of `hole` <=> entry s*64+lane is a hole
  unsigned hole = 0;
  for (int m = lane, sidx = 0; m < N; m += 64, sidx++) {
    const double w = pW[m], mx = pMX[m], my = pMY[m], xx = pSXX[m], xy = pSXY[m], yy = pSYY[m];
    double i00, i01, i10, i11, det;
    inv2(xx, xy, xy, yy, i00, i01, i10, i11, det);
    sMX[m] = mx; sMY[m] = my; sXX[m] = xx; sXY[m] = xy; sYY[m] = yy; sW[m] = w;
    sI00[m] = i00; sI01[m] = i01; sI11[m] = i11;
    if (w < 0) hole |= 1u << sidx;  // already a hole (merge called twice)
  }
  wave_sync();

  const double t2 = P.mergeT2, f = P.mergeInfl;
  bool anyMerge = false;
  for (int a = 0; a < N; a++) {
    // is a itself a hole?  owner lane a&63, slot a>>6
    const unsigned ownerHole = (unsigned)__builtin_amdgcn_readlane((int)hole, a & 63);
    if ((ownerHole >> (a >> 6)) & 1u) continue;
    // current state of Gaussian a (wave-uniform)
    double ax = sMX[a], ay = sMY[a], axx = sXX[a], axy = sXY[a], ayy = sYY[a], aw = sW[a];
    double a00 = sI00[a], a01 = sI01[a], a11 = sI11[a];
    bool changed = false;
    for (int c0 = (a + 1) & ~63; c0 < N; c0 += 64) {
      const int j = c0 + lane;
      const int slot = c0 >> 6;
      bool live = (j > a) && (j < N) && !((hole >> slot) & 1u);
      double jx = 0, jy = 0, jw = 0, j00 = 0, j01 = 0, j11 = 0;
      if (live) { jx = sMX[j]; jy = sMY[j]; jw = sW[j]; j00 = sI00[j]; j01 = sI01[j]; j11 = sI11[j]; }
      int floorLane = 0;  // only lanes >= floorLane are (re-)tested
      while (true) {
        bool pass = false;
        if (live && lane >= floorLane) {
          // d1 = md2 of x_j under (x_a, S_a); d2 = md2 of x_a under (x_j, S_j)   (:434-442)
          const double e0 = jx - ax, e1 = jy - ay;
          const double u0 = e0 * a00 + e1 * a01, u1 = e0 * a01 + e1 * a11;
          const double d1 = u0 * e0 + u1 * e1;
          bool far = d1 > t2;
          if (far) {
            const double g0 = -e0, g1 = -e1;
            const double v0 = g0 * j00 + g1 * j01, v1 = g0 * j01 + g1 * j11;
            const double d2 = v0 * g0 + v1 * g1;
            far = d2 > t2;
          }
          pass = !far && ((aw + jw) != 0.0);
        }
        const unsigned long long pm = __ballot(pass);
        if (pm == 0ull) break;
        const int l = __builtin_ctzll(pm);
        const int jj = c0 + l;
        // merge jj into a (:444-471), wave-uniform arithmetic
        const double w1 = aw, w2 = sW[jj];
        const double x2 = sMX[jj], y2 = sMY[jj], bxx = sXX[jj], bxy = sXY[jj], byy = sYY[jj];
        const double wm = w1 + w2;
        const double xm = (ax * w1 + x2 * w2) / wm, ym = (ay * w1 + y2 * w2) / wm;
        const double d10 = xm - ax, d11 = ym - ay, d20 = xm - x2, d21 = ym - y2;
        const double nxx = (w1 * (axx + (f * d10) * d10) + w2 * (bxx + (f * d20) * d20)) / wm;
        const double nxy = (w1 * (axy + (f * d10) * d11) + w2 * (bxy + (f * d20) * d21)) / wm;
        const double nyy = (w1 * (ayy + (f * d11) * d11) + w2 * (byy + (f * d21) * d21)) / wm;
        ax = xm; ay = ym; axx = nxx; axy = nxy; ayy = nyy; aw = wm;
        double i10, det;
        inv2(axx, axy, axy, ayy, a00, a01, i10, a11, det);
        changed = true;
        if (lane == l) { hole |= 1u << slot; live = false; }
        floorLane = l + 1;
        if (floorLane >= 64) break;
      }
    }
    if (changed) {
      anyMerge = true;
      // every lane stores the same values (uniform); visible to later reads of this wave
      sMX[a] = ax; sMY[a] = ay; sXX[a] = axx; sXY[a] = axy; sYY[a] = ayy; sW[a] = aw;
      sI00[a] = a00; sI01[a] = a01; sI11[a] = a11;
    }
  }
  wave_sync();
  if (!anyMerge) return;  // nothing changed: the slab is already right
  for (int m = lane, sidx = 0; m < N; m += 64, sidx++) {
    const bool h = (hole >> sidx) & 1u;
    pW[m] = h ? -1.0 : sW[m];
    if (!h) { pMX[m] = sMX[m]; pMY[m] = sMY[m]; pSXX[m] = sXX[m]; pSXY[m] = sXY[m]; pSYY[m] = sYY[m]; }
  }
}

// LDS per wave: keys[cap] doubles
template <int WPB>
__global__ __launch_bounds__(WPB * 64) void gm_prune_kernel(Buffers B, Params P, int src, int dst) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const int wave = threadIdx.x >> 6;
  const int lane = threadIdx.x & 63;
  const int i = __builtin_amdgcn_readfirstlane(blockIdx.x * WPB + wave);
  if (i >= B.N) return;
  const int cap = B.cap;
  double *keys = reinterpret_cast<double *>(smem_raw) + (size_t)wave * cap;
  const int N = B.count[i];
  const double *sl = B.slab[src];
  double *dl = B.slab[dst];
  const double *qW = plane((double *)sl, cap, i, PL_W);
  const double t = P.pruneT;
  for (int m = lane; m < N; m += 64) keys[m] = qW[m];
  wave_sync();
  int kept = 0;
  for (int m = lane; m < N; m += 64) {
    const double wm = keys[m];
    const bool keep = (wm >= t) && (wm >= 0.0);  // holes carry -1
    int rank = 0;
    if (keep) {
      for (int j = 0; j < N; j++) {
        const double wj = keys[j];
        rank += (wj > wm || (wj == wm && j < m)) ? 1 : 0;  // everything ranked ahead of a survivor also survives
      }
      for (int pl = 0; pl < PL_COUNT; pl++) plane(dl, cap, i, pl)[rank] = plane((double *)sl, cap, i, pl)[m];
      kept++;
    }
  }
  kept = wave_sum_i(kept);
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
__global__ void normalize_kernel(double *w, int N, double sum, const double *sumDev) {
  const double sdiv = sumDev ? sumDev[0] : sum;
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < N) w[k] = w[k] / sdiv;
}
__global__ void set_weights_kernel(double *w, int N, double v) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < N) w[k] = v;
}

// Resample copy: slot k takes slot src[k]'s mixture (Particle::copy -> GaussianMixture copy ctor).
// One block per destination slot; sources are slots that keep themselves, so in-place is hazard-free.
__global__ __launch_bounds__(256) void resample_gather_kernel(Buffers B, int cur, const int *srcSlot) {
  const int k = blockIdx.x;
  const int s = srcSlot[k];
  if (s == k) return;
  const int n = B.count[s];
  double *slab = B.slab[cur];
  for (int pl = 0; pl < PL_COUNT; pl++) {
    const double *q = plane(slab, B.cap, s, pl);
    double *d = plane(slab, B.cap, k, pl);
    for (int m = threadIdx.x; m < n; m += blockDim.x) d[m] = q[m];
  }
  if (threadIdx.x == 0) {
    B.count[k] = n;
    B.unusedMask[k] = B.unusedMask[s];
    B.nInFov[k] = B.nInFov[s];
  }
}

// Number of valid Gaussians per particle (holes left by gm_merge carry w < 0 until gm_prune drops them).
__global__ __launch_bounds__(256) void valid_count_kernel(Buffers B, int cur, int *out) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= B.N) return;
  const int lane = threadIdx.x & 63;
  const int n = B.count[i];
  const double *w = plane(B.slab[cur], B.cap, i, PL_W);
  int c = 0;
  for (int m = lane; m < n; m += 64) c += (w[m] >= 0.0) ? 1 : 0;
  c = wave_sum_i(c);
  if (lane == 0) out[i] = c;
}

// rfsgpu_restore_state: copy the saved live entries back (one block per particle).
__global__ __launch_bounds__(256) void restore_state_kernel(Buffers B, int cur, const double *snapSlab, const double *snapWeight,
                                                            const int *snapCount, const int *snapFov, const unsigned long long *snapUnused) {
  const int k = blockIdx.x;
  const int n = snapCount[k];
  for (int pl = 0; pl < PL_COUNT; pl++) {
    const double *q = snapSlab + ((size_t)k * PL_COUNT + pl) * (size_t)B.cap;
    double *d = plane(B.slab[cur], B.cap, k, pl);
    for (int m = threadIdx.x; m < n; m += blockDim.x) d[m] = q[m];
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
template <int WPB>
__global__ __launch_bounds__(WPB * 64) void predict_map_kernel(Buffers B, Params P, int cur, int addBirth, int nZprev) {
  const int wave = threadIdx.x >> 6;
  const int lane = threadIdx.x & 63;
  const int i = __builtin_amdgcn_readfirstlane(blockIdx.x * WPB + wave);
  if (i >= B.N) return;
  const int cap = B.cap;
  double *slab = B.slab[cur];
  double *pW = plane(slab, cap, i, PL_W), *pWP = plane(slab, cap, i, PL_WP), *pMX = plane(slab, cap, i, PL_MX),
         *pMY = plane(slab, cap, i, PL_MY);
  double *pSXX = plane(slab, cap, i, PL_SXX), *pSXY = plane(slab, cap, i, PL_SXY), *pSYY = plane(slab, cap, i, PL_SYY);
  int n = B.count[i];
  const int nOld = n;  // staticStep below covers the pre-existing Gaussians; births get Q added where they are created
  if (addBirth && nZprev > 0) {
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
          const double px = B.pose[3 * i], py = B.pose[3 * i + 1], pth = B.pose[3 * i + 2];
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
  for (int m = lane; m < nOld; m += 64) {
    pSXX[m] += P.Qlm[0];
    pSXY[m] += P.Qlm[1];
    pSYY[m] += P.Qlm[2];
  }
}

// MatPerm::calc (reference src/MatrixPermanent.cpp:41-112), Nijenhuis-Wilf / Gray-code Ryser.  One matrix per
// wave; the 2^(n-1) Gray-code steps are split into 64 contiguous ranges, one per lane: each lane jumps to
// its first subset directly (Gray code of the start index), then walks its range with the single-column
// updates of the reference; partial sums are wave-reduced.
__global__ __launch_bounds__(64) void mat_perm_kernel(const double *A, int n, int batch, double *out) {
  const int b = blockIdx.x;
  if (b >= batch) return;
  const int lane = threadIdx.x & 63;
  __shared__ double sA[24 * 24];
  const double *Ab = A + (size_t)b * n * n;
  for (int t = lane; t < n * n; t += 64) sA[t] = Ab[t];
  __syncthreads();
  // x_i(0) = A(i,n-1) - row_sum_i / 2; subset index k (1-based in the reference) has gray code g(k-1)
  const unsigned long long total = 1ull << (n - 1);  // number of subsets (k = 1 .. 2^(n-1))
  const unsigned long long per = (total + 63) / 64;
  const unsigned long long k0 = per * lane;          // 0-based subset index
  const unsigned long long k1 = (k0 + per < total) ? k0 + per : total;
  double x[24];
  double acc = 0.0;
  if (k0 < total) {
    const unsigned long long g0 = k0 ^ (k0 >> 1);
    for (int r = 0; r < n; r++) {
      double rs = 0;
      for (int c = 0; c < n; c++) rs += sA[r * n + c];
      double v = sA[r * n + (n - 1)] - 0.5 * rs;
      for (int c = 0; c < n - 1; c++) if ((g0 >> c) & 1ull) v += sA[r * n + c];
      x[r] = v;
    }
    unsigned long long g = g0;
    for (unsigned long long k = k0; k < k1; k++) {
      if (k != k0) {
        const int j = __builtin_ctzll(k);  // bit flipped between gray(k-1) and gray(k)
        const double z = ((g >> j) & 1ull) ? -1.0 : 1.0;
        g ^= (1ull << j);
        for (int r = 0; r < n; r++) x[r] += z * sA[r * n + j];
      }
      double prod = 1.0;
      for (int r = 0; r < n; r++) prod *= x[r];
      // sign: s = -1 for k=1 (0-based 0), alternating
      acc += ((k & 1ull) ? 1.0 : -1.0) * prod;
    }
  }
  acc = wave_sum(acc);
  if (lane == 0) {
    double ret = 2 * acc;
    if (n % 2 != 0) ret *= -1;
    out[b] = ret;
  }
}
